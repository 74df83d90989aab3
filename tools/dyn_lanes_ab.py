import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
from mopa_rl_amd.kinematic_env import make_env
for E in (4096, 16384, 32768):
    for lanes in (16, 1):
        env = make_env("SawyerPushObstacle-v0", E, dynamics=True, dyn_lanes=lanes)
        env.reset()
        n = 10
        acts = (torch.rand(n + 3, E, env.action_dim, dtype=torch.float64, device=env.device) * 2 - 1).contiguous()
        for k in range(3): env.step(acts[k])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n): env.step(acts[3 + k])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"E={E} lanes/env={lanes}: {dt*1e3:.3f} ms/step {E/dt/1e6:.3f} M env-steps/s", flush=True)
        env.close()
