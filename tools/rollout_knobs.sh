#!/bin/bash
# A/B of the asynchronous rollout's planner-launch knobs on the bench's own rollout sections (GPU box):
#   bash tools/rollout_knobs.sh "planner_workgroups=64,planner_streams=2" "planner_workgroups=96,planner_streams=3" ...
for c in "$@"; do
  MOPA_BENCH_ROLLOUT=$c timeout 200 python bench.py --no-cpu --no-plan --no-env --steps 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$c', *[(k, round(d[k]['agent_steps_per_s'] / 1e3), round(d[k]['s_per_agent_step_batch'] * 1e3, 2), round(d[k]['envs_stepping_per_call'])) for k in ('rollout_async', 'rollout_lift')])"
done
