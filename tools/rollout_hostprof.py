"""cProfile of the host side of asynchronous agent_step calls (GPU box): where a call's Python / dispatch time goes."""
import sys, cProfile, pstats; sys.path.insert(0, ".")
import torch
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
E = 4096
env = make_env("SawyerPushObstacle-v0", E, seed=5); env.reset()
ro = BatchMoPARollout(env, RolloutConfig(async_planner=True))
gen = torch.Generator(device=env.device); gen.manual_seed(1)
pr = cProfile.Profile()
for t in range(90):
    ac = torch.rand(E, 7, generator=gen, dtype=torch.float64, device=env.device) * 2 - 1
    if t >= 30: pr.enable()
    out = ro.agent_step(ac)
    if t >= 30: pr.disable()
    env.reset(out["done"].bool() & out["stepped"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
