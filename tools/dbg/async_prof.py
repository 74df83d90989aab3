import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
E = 4096
env = make_env("SawyerPushObstacle-v0", E, seed=5); env.reset()
ro = BatchMoPARollout(env, RolloutConfig(async_planner=True))
gen = torch.Generator(device=env.device); gen.manual_seed(1)
import mopa_rl_amd.rollout as R
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for n in ("_rrt_advance", "_rrt_finish", "plan", "ik_displacement", "_rrt_launch", "_densify_cut", "_fallback_launch", "_merge_paths"): wrap(ro, n)
_v = ro._valid
def _valid_t(q):
    t0 = time.perf_counter(); r = _v(q); t1 = time.perf_counter(); r2 = r.cpu(); t2 = time.perf_counter()
    if t2 - t0 > 5e-3: print("  _valid slow: launch %.1f ms, cpu() %.1f ms, N=%d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(q)))
    return r
ro._valid = _valid_t
wrap(ro.bp, "pullback"); wrap(R, "simple_interpolate_batch"); wrap(env, "exec_trajectories"); wrap(env, "_launch")
times = []
for t in range(24):
    ac = torch.rand(E, 7, generator=gen, dtype=torch.float64, device=env.device) * 2 - 1
    if t == 4: T.clear()
    T0 = dict(T)
    torch.cuda.current_stream().synchronize(); t0 = time.perf_counter()
    out = ro.agent_step(ac)
    t1 = time.perf_counter()
    torch.cuda.current_stream().synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    if times[-1] > 18 and t > 3: print("slow call", t, round(times[-1], 1), "host part", round((t1 - t0) * 1e3, 1), {k: round((T[k] - T0.get(k, 0)) * 1e3, 1) for k in T if T[k] - T0.get(k, 0) > 1e-3})
    d = out["done"].bool() & out["stepped"]
    if bool(d.any()): env.reset(d)
print("ms per call", [round(x, 1) for x in times])
print({k: round(v / 20 * 1e3, 2) for k, v in T.items()}, "pending jobs", len(ro._jobs))
