# Asynchronous Push rollout against the pooling policy (queries a launch waits for / takes at most) -- K3 at two waves per SIMD
for knobs in "planner_min_job=1024" "planner_min_job=512" "planner_min_job=256" "planner_min_job=128" "planner_min_job=64" "planner_min_job=256,planner_streams=4" "planner_min_job=128,planner_streams=4,planner_workgroups=96" "planner_min_job=256,planner_job_cap=512"; do
  out=""
  for r in 1 2; do
    v=$(env ONLY_EAGER=1 MOPA_BENCH_ROLLOUT=$knobs python tools/rollout_graphs_ab.py 4096 200 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*envs_stepping_per_call': \([0-9.]*\).*/\1 \2/" | awk '{printf "%d(%d)", $1, $2}')
    out="$out $v"
  done
  echo "$knobs: agent steps/s (envs stepping per call), two runs:$out"
done
