#!/bin/bash
# Build container, after `gpurun -- 'bash tools/profile_round.sh rNN'`: copies the judged summaries from gpurun_out/ (scratch) into
# profiles/rNN/ (tracked).  bash tools/collect_round.sh r05
TAG=${1:-r06}
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles/$TAG; mkdir -p $P
python tools/make_traffic_json.py $G/prof_$TAG $P
python tools/summarize_prof.py $G/prof_$TAG > $P/k_is_valid_v5_summary.txt
# the validity kernel on Lift / Assembly (BASELINE configs 4 / 5): summary + sealed traffic record per scene
for sc in lift assembly; do
  [ -d $G/prof_${TAG}_$sc ] || continue
  ENVN=SawyerLiftObstacle-v0; [ $sc = assembly ] && ENVN=SawyerAssemblyObstacle-v0
  MOPA_BENCH_ENV=$ENVN python tools/make_traffic_json.py $G/prof_${TAG}_$sc $P k_is_valid_v5 k_is_valid_traffic_$sc.json > /dev/null
  python tools/summarize_prof.py $G/prof_${TAG}_$sc > $P/k_is_valid_v5_summary_$sc.txt
done
head -12 $G/prof_$TAG/trace_kernel_stats.csv > $P/trace_kernel_stats_top.csv
cp $G/prof_$TAG/trace_domain_stats.csv $P/trace_domain_stats.csv
head -25 $G/prof_$TAG/full_kernel_stats.csv > $P/full_bench_kernel_stats_top.csv
cp $G/k3_$TAG/plan_passes.txt $P/k3_plan_passes.txt
cp $G/k3_$TAG/plan_fail_only.txt $P/k3_fail_only_timing.txt
cp $G/k3_$TAG/trace_kernel_stats_k3.csv $P/k3_fail_only_kernel_stats.csv
cp $G/k3_$TAG/pmc_counter_collection_k3.csv $P/k3_pmc_sq.csv
cp $G/k3_$TAG/pmc2_counter_collection_k3.csv $P/k3_pmc_sq2.csv
cp $G/prof_${TAG}_k7/trace_kernel_stats.csv $P/k7_trace_kernel_stats.csv
cp $G/prof_${TAG}_k7/pmc_sq_counter_collection_dyn.csv $P/k7_pmc_sq_counter_collection_dyn.csv
cp $G/prof_${TAG}_k7/pmc_sq2_counter_collection.csv $P/k7_pmc_sq2_counter_collection.csv
for f in parity_sweep plan_parity_sweep motion_parity_sweep ct_parity_sweep rollout_launches_per_call ct_bench ct_bench_pyramidal dyn_lanes_ab lone_wave k7_icache k3_build_ab ct_tail rollout_knobs bench_line bench_line_full; do
  [ -f $G/round_$TAG/$f.txt ] && cp $G/round_$TAG/$f.txt $P/$f.txt
  [ -f $G/round_$TAG/$f.json ] && cp $G/round_$TAG/$f.json $P/$f.json
done
# compile-time register / scratch table of every kernel (no GPU needed)
python tools/kernel_resources.py > $P/kernel_resources.txt
git status --short $P | head -40
