# asynchronous Push rollout: how many queries a planner launch waits for (planner_min_job) x streams, 2 runs of 200 calls
for mj in 64 128 256 512 1024; do for st in 2 3; do
  for g in ONLY_EAGER ONLY_GRAPHS; do
    out=""
    for r in 1 2; do
      v=$(env $g=1 MOPA_BENCH_ROLLOUT=planner_streams=$st,planner_min_job=$mj python tools/rollout_graphs_ab.py 4096 200 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*envs_stepping_per_call': \([0-9.]*\).*/\1 \2/" | awk '{printf "%d(%d)", $1, $2}')
      out="$out $v"
    done
    echo "min_job=$mj streams=$st $g:$out"
  done
done; done
