"""A/B of the asynchronous Push rollout with and without HIP-graph replay: `python tools/rollout_graphs_ab.py [E] [calls]`."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
dev = torch.device("cuda", 0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for g in ((True,) if os.environ.get("ONLY_GRAPHS") else ((False,) if os.environ.get("ONLY_EAGER") else (False, True))):
    r = bench.rollout_section(torch, bench.ENV, E, dev, calls, async_planner=True, use_graphs=g)
    print("graphs", g, {k: r[k] for k in ("agent_steps_per_s", "env_steps_per_s", "s_per_agent_step_batch", "envs_stepping_per_call")}, flush=True)
