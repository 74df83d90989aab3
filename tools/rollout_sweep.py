"""Asynchronous rollout with uniform random actions: ms per agent_step call and agent steps/s for given planner-launch
knobs (GPU box).  python tools/rollout_sweep.py <planner_workgroups> <planner_streams> <planner_job_cap> [env name]"""
import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
E = 4096
def run(wg, streams, cap, calls=80, env_name="SawyerPushObstacle-v0"):
    env = make_env(env_name, E, seed=5); env.reset()
    ro = BatchMoPARollout(env, RolloutConfig(async_planner=True, planner_workgroups=wg, planner_streams=streams, planner_job_cap=cap))
    gen = torch.Generator(device=env.device); gen.manual_seed(1)
    times, stepped = [], 0
    for t in range(calls):
        ac = torch.rand(E, env.action_dim if hasattr(env, "action_dim") else 7, generator=gen, dtype=torch.float64, device=env.device) * 2 - 1
        torch.cuda.current_stream().synchronize(); t0 = time.perf_counter()
        out = ro.agent_step(ac)
        torch.cuda.current_stream().synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        if t >= 20: stepped += int(out["stepped"].sum())
        d = out["done"].bool() & out["stepped"]
        if bool(d.any()): env.reset(d)
    tt = sum(times[20:])
    print(f"wg {wg:4d} streams {streams} cap {cap:5d}: {tt / (calls - 20):6.2f} ms/call, {stepped / tt:7.1f} k agent steps/s, max call {max(times[20:]):.1f} ms, median {np.median(times[20:]):.2f}", flush=True)
    ro.drain()
a = [int(x) for x in sys.argv[1:4]]
run(*a, env_name=sys.argv[4] if len(sys.argv) > 4 else "SawyerPushObstacle-v0")
