for knobs in "planner_first_iters=100" "planner_first_iters=150" "planner_first_iters=300" "planner_first_iters=500"; do
  out=""
  for r in 1 2; do
    v=$(env ONLY_EAGER=1 MOPA_BENCH_ROLLOUT=$knobs python tools/rollout_graphs_ab.py 4096 300 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*envs_stepping_per_call': \([0-9.]*\).*/\1 \2/" | awk '{printf "%d(%d)", $1, $2}')
    out="$out $v"
  done
  echo "$knobs:$out"
done
