#!/bin/bash
# GPU box: sweep of the asynchronous planner's pooling knobs on one env (default PusherObstacle-v0) at 4096 / 8192 envs:
#   bash tools/rollout_minjob_sweep.sh [env]      -> "cfg: envs:agent-steps/s(envs stepping per call) ..."
ENVN=${1:-PusherObstacle-v0}
for cfg in "planner_min_job=1024" "planner_min_job=512" "planner_min_job=256" "planner_min_job=128" "planner_min_job=64" "planner_min_job=128,planner_streams=4" "planner_min_job=128,planner_first_iters=100" "planner_min_job=128,planner_first_iters=0" "planner_min_job=256,planner_job_cap=512"; do
  r=$(MOPA_BENCH_ROLLOUT="$cfg" python tools/rollout_envs_sweep.py $ENVN 4096 8192 2>&1 | grep agent_steps | python -c "
import sys, json
print(' '.join('%d:%.0fk(%d)' % (d['envs'], d['agent_steps_per_s'] / 1e3, d['envs_stepping_per_call']) for d in map(json.loads, sys.stdin)))")
  echo "$cfg: $r"
done
