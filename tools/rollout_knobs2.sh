for st in 1 2 3; do for wg in 32 64 128; do
  for g in ONLY_EAGER ONLY_GRAPHS; do
    r=$(env $g=1 MOPA_BENCH_ROLLOUT=planner_streams=$st,planner_workgroups=$wg python tools/rollout_graphs_ab.py 4096 150 2>&1 | grep "^graphs" | sed -e "s/'s_per_agent.*//")
    echo "streams=$st wg=$wg $r"
  done
done; done
