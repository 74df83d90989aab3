"""Per-phase time of asynchronous agent_step calls (target / interpolate / plan / execute; each mark synchronises the main
stream only) on the GPU box."""
import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
E = 4096
env = make_env("SawyerPushObstacle-v0", E, seed=5); env.reset()
ro = BatchMoPARollout(env, RolloutConfig(async_planner=True))
gen = torch.Generator(device=env.device); gen.manual_seed(1)
T = {}
ro.timing = T
times = []
for t in range(40):
    ac = torch.rand(E, 7, generator=gen, dtype=torch.float64, device=env.device) * 2 - 1
    if t == 8: T.clear()
    torch.cuda.current_stream().synchronize(); t0 = time.perf_counter()
    out = ro.agent_step(ac)
    torch.cuda.current_stream().synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    d = out["done"].bool() & out["stepped"]
    if bool(d.any()): env.reset(d)
print("ms per call (with phase syncs)", [round(x, 1) for x in times[8:]])
print({k: round(v / 32 * 1e3, 2) for k, v in T.items()})
