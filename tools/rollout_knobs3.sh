# repeated A/B of the asynchronous Push rollout over planner streams x workgroups (3 runs of 200 calls each)
for cfg in "3 64" "1 64" "1 128" "2 128" "1 256"; do set -- $cfg
  for g in ONLY_EAGER ONLY_GRAPHS; do
    out=""
    for r in 1 2 3; do
      v=$(env $g=1 MOPA_BENCH_ROLLOUT=planner_streams=$1,planner_workgroups=$2 python tools/rollout_graphs_ab.py 4096 200 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*/\1/" | cut -d. -f1)
      out="$out $v"
    done
    echo "streams=$1 wg=$2 $g:$out"
  done
done
