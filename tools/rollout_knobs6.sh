#!/bin/bash
# sweep on round 6's build: the continuation launches as workgroup-per-query launches (planner_exclusive=1) on few CUs
ENVN=${1:-SawyerPushObstacle-v0}
for cfg in "planner_exclusive=0" "planner_exclusive=1,planner_chain_workgroups=16" "planner_exclusive=1,planner_chain_workgroups=32" "planner_exclusive=1,planner_chain_workgroups=48" "planner_exclusive=1,planner_chain_workgroups=64" "planner_exclusive=1,planner_chain_workgroups=32,planner_streams=4" "planner_exclusive=1,planner_chain_workgroups=32,planner_first_iters=200"; do
  r=$(MOPA_BENCH_ROLLOUT="$cfg" python tools/rollout_envs_sweep.py $ENVN 4096 8192 2>&1 | grep agent_steps | python -c "
import sys, json
print(' '.join('%d:%.0fk(%d)' % (d['envs'], d['agent_steps_per_s'] / 1e3, d['envs_stepping_per_call']) for d in map(json.loads, sys.stdin)))")
  echo "$cfg: $r"
done
