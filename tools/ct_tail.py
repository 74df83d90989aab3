"""K7 tail probe: per-env solver sweeps per sub-step under the bench's uniform policy, by noslip setting.
   python tools/ct_tail.py [env] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from mopa_rl_amd import _lib
from mopa_rl_amd.kinematic_env import make_env

name = {"push": "SawyerPushObstacle-v0", "lift": "SawyerLiftObstacle-v0", "assembly": "SawyerAssemblyObstacle-v0"}[sys.argv[1] if len(sys.argv) > 1 else "lift"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
E = 4096
dev = torch.device("cuda:0")
for ns in (0, 5):
    env = make_env(name, E, device=dev, seed=seed, dynamics=True, contacts=True, max_episode_steps=250, contact_options=dict({"noslip_iterations": ns}, **({"maxpair": int(os.environ["CT_MAXPAIR"])} if os.environ.get("CT_MAXPAIR") else {})))
    env.reset()
    stats = torch.zeros(E, 4, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, stats.data_ptr()))
    g = torch.Generator(device=dev); g.manual_seed(seed + 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for t in range(steps):
        a = ((torch.rand(E, env.action_dim, generator=g, dtype=torch.float64, device=dev) * 2 - 1) * float(os.environ.get("CT_SCALE", "1"))).contiguous()
        ev0.record(); env.step(a); ev1.record(); torch.cuda.synchronize()
        s = stats.cpu().numpy().astype(np.float64)
        sw = s[:, 1] / env.dyn.nsub
        con = s[:, 0] / env.dyn.nsub
        top = np.argsort(-sw)[:4]
        print(f"{name} noslip {ns} step {t}: {ev0.elapsed_time(ev1):6.2f} ms; sweeps/sub-step mean {sw.mean():5.2f} p99 {np.percentile(sw, 99):5.1f} max {sw.max():5.1f}; "
              f"envs > 20: {(sw > 20).sum()}, > 40: {(sw > 40).sum()}; top envs {[int(x) for x in top]} sweeps {[round(float(x), 1) for x in sw[top]]} contacts {[round(float(x), 1) for x in con[top]]}", flush=True)
    env.close()
