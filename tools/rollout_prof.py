import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
from mopa_rl_amd.kinematic_env import BatchKinematicPushEnv
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
import mopa_rl_amd.rollout as R
E = 4096
env = BatchKinematicPushEnv(E, seed=5); env.reset()
ro = BatchMoPARollout(env, RolloutConfig())
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T["fn:" + name] = T.get("fn:" + name, 0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
for n in ("_densify",): wrap(ro, n)
ro.timing = T
wrap(ro.bp, "plan"); wrap(R, "simple_interpolate_batch")
wrap(ro.main if hasattr(ro, "main") else ro, "plan") if False else None
gen = torch.Generator(device=env.device); gen.manual_seed(1)
for t in range(4):
    ac = torch.rand(E, 7, generator=gen, dtype=torch.float64, device=env.device) * 2 - 1
    if t == 1: T.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    ro.agent_step(ac)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 3
print("per agent step %.1f ms" % (tot * 1e3), {k: round(v / 3 * 1e3, 1) for k, v in T.items()})
