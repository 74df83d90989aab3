"""Asynchronous rollout (bench.py's rollout_section: SAC actor in the loop, exchanges) at several env counts per GPU: does the
agent-step rate grow with the number of resident envs (half of 4096 envs wait for an RRT-Connect query at any time)?  GPU box.
python tools/rollout_envs_sweep.py [env name] [E ...]"""
import sys, os, json; sys.path.insert(0, ".")
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "SawyerPushObstacle-v0"
Es = [int(x) for x in sys.argv[2:]] or [4096, 8192, 16384]
dev = torch.device("cuda:0")
for E in Es:
    r = bench.rollout_section(torch, name, E, dev, 300, async_planner=True)
    print(json.dumps({"env": name, "envs": E, "agent_steps_per_s": round(r["agent_steps_per_s"]), "env_steps_per_s": round(r["env_steps_per_s"]),
                      "ms_per_call": round(r["s_per_agent_step_batch"] * 1e3, 3), "envs_stepping_per_call": r["envs_stepping_per_call"],
                      "counters": r["counters"]}), flush=True)
