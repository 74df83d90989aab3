# Asynchronous Push rollout with chained planner launches (first phase + continuation on one stream): streams x pool size
for knobs in "planner_streams=3" "planner_streams=4" "planner_streams=5" "planner_streams=3,planner_min_job=512" "planner_streams=3,planner_min_job=256" "planner_streams=4,planner_min_job=512,planner_workgroups=96" "planner_streams=3,planner_workgroups=160" "planner_streams=3,planner_first_iters=200"; do
  out=""
  for r in 1 2; do
    v=$(env ONLY_EAGER=1 MOPA_BENCH_ROLLOUT=$knobs python tools/rollout_graphs_ab.py 4096 300 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*envs_stepping_per_call': \([0-9.]*\).*/\1 \2/" | awk '{printf "%d(%d)", $1, $2}')
    out="$out $v"
  done
  echo "$knobs: agent steps/s (envs stepping per call), two runs:$out"
done
