"""Asynchronous rollouts for 4000 calls (uniform random actions; Push, Lift and Pusher): memory, waiting envs, the slowest env's step
count (fairness of the planner queue) and the counters every 1000 calls.  GPU box."""
import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
E = 4096
for env_name in (sys.argv[1:] or ("SawyerPushObstacle-v0", "SawyerLiftObstacle-v0", "PusherObstacle-v0")):
    env = make_env(env_name, E, seed=5, max_episode_steps=250); env.reset()
    ro = BatchMoPARollout(env, RolloutConfig.for_env(env_name, async_planner=True))
    gen = torch.Generator(device=env.device); gen.manual_seed(1)
    t0 = time.perf_counter(); stepped = 0
    for t in range(4000):
        ac = torch.rand(E, env.action_dim, generator=gen, dtype=torch.float64, device=env.device) * 2 - 1
        out = ro.agent_step(ac)
        env.reset(out["done"].bool() & out["stepped"])
        if t % 1000 == 999:
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            print(env_name, "call", t + 1, "elapsed %.1f s" % (time.perf_counter() - t0), "torch alloc %.2f GB reserved %.2f GB, device used %.2f GB" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, (total - free) / 2**30), "jobs", len(ro._jobs), "busy", int(ro.busy.sum()), "t_env min/max", int(ro.t_env.min()), int(ro.t_env.max()), {k: int(v.sum()) for k, v in ro.counters.items()}, flush=True)
    ro.drain()
