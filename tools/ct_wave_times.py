import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mopa_rl_amd import _lib
from mopa_rl_amd.kinematic_env import make_env
E=4096; dev=torch.device("cuda:0")
env = make_env("SawyerPushObstacle-v0", E, device=dev, seed=11, dynamics=True, contacts=True, max_episode_steps=1 << 30)
g = torch.Generator(device=dev); g.manual_seed(3)
acts = ((torch.rand(12, E, env.action_dim, generator=g, dtype=torch.float64, device=dev) * 2 - 1)).contiguous()
stats = torch.zeros(E, 4, dtype=torch.int32, device=dev)
_lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, stats.data_ptr()))
env.reset()
for t in range(10):
    env.step(acts[t]); torch.cuda.synchronize()
    s = stats.cpu().numpy()
    tw = s[::4, 3] / 100.0   # us per wave (first env of each wave)
    it = s[:, 1].reshape(-1, 4) / 75.0
    con = s[:, 0].reshape(-1, 4) / 75.0
    order = np.argsort(-tw)[:5]
    print(f"step {t}: wave time us: mean {tw.mean():.0f} p50 {np.median(tw):.0f} p99 {np.percentile(tw,99):.0f} max {tw.max():.0f}; slowest waves: " +
          "; ".join(f"w{w} {tw[w]:.0f}us it {np.round(it[w],1)} con {np.round(con[w],1)}" for w in order))
