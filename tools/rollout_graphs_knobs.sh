for knobs in "planner_streams=2,planner_workgroups=128" "planner_streams=3,planner_workgroups=128" "planner_streams=3,planner_workgroups=96"; do
  out=""
  for r in 1 2; do
    v=$(env ONLY_GRAPHS=1 MOPA_BENCH_ROLLOUT=$knobs python tools/rollout_graphs_ab.py 4096 200 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*envs_stepping_per_call': \([0-9.]*\).*/\1 \2/" | awk '{printf "%d(%d)", $1, $2}')
    out="$out $v"
  done
  echo "graphs $knobs:$out"
done
