#!/bin/bash
# GPU box: the asynchronous rollout's planner-launch knobs on the current build (bench's section, SAC actor in the loop):
# planner_workgroups x planner_streams x planner_first_iters at 4096 and 8192 envs.   bash tools/rollout_knobs.sh [env]
ENVN=${1:-SawyerPushObstacle-v0}
for wg in 64 128 192 256; do for st in 2 3 4; do for fi in 120 300; do
  r=$(MOPA_BENCH_ROLLOUT="planner_workgroups=$wg,planner_streams=$st,planner_first_iters=$fi" python tools/rollout_envs_sweep.py $ENVN 4096 8192 2>&1 | grep agent_steps | python -c "
import sys, json
print(' '.join('%d:%.0fk(%d)' % (d['envs'], d['agent_steps_per_s'] / 1e3, d['envs_stepping_per_call']) for d in map(json.loads, sys.stdin)))")
  echo "wg $wg streams $st first $fi: $r"
done; done; done
