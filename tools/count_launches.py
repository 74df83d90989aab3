"""Device launches (kernels + memcpys / memsets) of one BatchMoPARollout.agent_step call (kinematic env, asynchronous planner,
4096 envs), by name: torch's profiler around `calls` steady-state calls of agent_step ALONE -- the actions are drawn before the
profiled region and episodes are not reset inside it (MOPA_COUNT_LOOP=1: with the bench loop's action draw + env.reset).
MOPA_BENCH_ROLLOUT=fused=0 counts the torch form.   python tools/count_launches.py [calls] [env]"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
name = sys.argv[2] if len(sys.argv) > 2 else "SawyerPushObstacle-v0"
over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("MOPA_BENCH_ROLLOUT", "").split(",") if kv)}
E = 4096
env = make_env(name, E, seed=21, max_episode_steps=250)
env.reset()
ro = BatchMoPARollout(env, RolloutConfig(async_planner=True, **over))
g = torch.Generator(device=env.device)
g.manual_seed(8)
ad = ro.ac_dim


loop = bool(os.environ.get("MOPA_COUNT_LOOP"))


def one(a=None):
    if a is None:
        a = (torch.rand(E, ad, generator=g, dtype=torch.float64, device=env.device) * 2 - 1)
    out = ro.agent_step(a)
    if loop or a is None:
        env.reset(out["done"].bool() & out["stepped"])


for _ in range(40):
    one()
acts = [(torch.rand(E, ad, generator=g, dtype=torch.float64, device=env.device) * 2 - 1) for _ in range(calls)]
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for k in range(calls):
        one(None if loop else acts[k])
    torch.cuda.synchronize()
c = Counter()
t = Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        c[ev.name] += 1
        t[ev.name] += ev.device_time
tot = sum(c.values())
lib_k = sum(v for k, v in c.items() if k.startswith("k_") or k.startswith("void k_"))
print(f"{tot / calls:.1f} device launches per agent_step call over {calls} calls ({'with' if loop else 'without'} the loop's action draw / env.reset; "
      f"fused={ro.cfg.fused}): {lib_k / calls:.1f} library kernels, {(tot - lib_k) / calls:.1f} torch kernels / copies; "
      f"{sum(t.values()) / calls:.0f} us of device time per call, side streams included")
for k, v in c.most_common(60):
    print(f"{v / calls:7.2f}  {t[k] / calls:8.1f} us  {k[:110]}")
