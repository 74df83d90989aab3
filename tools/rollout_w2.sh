# A/B: asynchronous Push rollout with the planner family at 1 vs 2 waves per SIMD (MOPA_HIP_LIB = a -DMOPA_PLAN_WAVES=2 build)
for lib in "" "$PWD/mopa_rl_amd/csrc/libmopa_w2.so"; do
  for knobs in "planner_streams=3,planner_workgroups=64" "planner_streams=3,planner_workgroups=128" "planner_streams=2,planner_workgroups=256" "planner_streams=4,planner_workgroups=128"; do
    v=$(env ${lib:+MOPA_HIP_LIB=$lib} ONLY_EAGER=1 MOPA_BENCH_ROLLOUT=$knobs python tools/rollout_graphs_ab.py 4096 200 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*envs_stepping_per_call': \([0-9.]*\).*/\1 \2/" | awk '{printf "%d agent steps/s (%d envs stepping per call)", $1, $2}')
    echo "lib=${lib:-default} $knobs: $v"
  done
done
