"""Is the Lift contact step's time bimodal?  Repeats bench.env_dynamics_block's Lift section, per-step times by HIP events."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from mopa_rl_amd import _lib
from mopa_rl_amd.kinematic_env import make_env

dev = torch.device("cuda:0")
E = 4096
name = sys.argv[1] if len(sys.argv) > 1 else "SawyerLiftObstacle-v0"
for rep in range(4):
    g = torch.Generator(device=dev); g.manual_seed(100 + rep)
    env = make_env(name, E, device=dev, seed=11, dynamics=True, contacts=True, max_episode_steps=1 << 30)
    steps = 8
    acts = (torch.rand(steps + 2, E, env.action_dim, generator=g, dtype=torch.float64, device=dev) * 2 - 1).contiguous()
    env.reset()
    stats = torch.zeros(E, 4, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, stats.data_ptr()))
    ts = []
    for t in range(steps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(acts[t]); e1.record(); torch.cuda.synchronize()
        s = stats.cpu().numpy().astype(np.float64)
        ts.append((e0.elapsed_time(e1), s[:, 1].max() / env.dyn.nsub, int(s[:, 1].argmax()), s[:, 0].max() / env.dyn.nsub))
    print(f"rep {rep}: " + "  ".join(f"{a:.1f}ms/it{b:.1f}@{c}/c{d:.1f}" for a, b, c, d in ts), flush=True)
    env.close()
