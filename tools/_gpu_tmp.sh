set -x
mkdir -p gpurun_out/r04_walk
timeout 900 python -m pytest tests/test_gpu_dyn.py -x -q -k "chunked or waypoint_execution or rollout_runs" 2>&1 | tail -15 > gpurun_out/r04_walk/test.log
cat gpurun_out/r04_walk/test.log
for c in 0 1 2 4; do
MOPA_BENCH_ROLLOUT=walk_chunk=$c timeout 600 python - <<PY 2>&1 | tail -3
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
r = bench.rollout_section(torch, "SawyerPushObstacle-v0", 4096, torch.device("cuda:0"), 12 if $c == 0 else 100, async_planner=True, dynamics=True)
print("chunk $c", json.dumps({k: r[k] for k in ("agent_steps_per_s", "env_steps_per_s", "s_per_agent_step_batch", "envs_stepping_per_call")}))
PY
done
