#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py.
# Small CSV summaries land in gpurun_out/prof_<tag>/ ; copy what you want judged into profiles/.
#   bash tools/profile.sh <tag> [k1only]      (k1only: the validity kernel's passes alone -- the per-scene profiles, with MOPA_BENCH_ENV set)
TAG=${1:-r03}
MODE=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
WORK=/tmp/prof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu --no-plan --no-env --no-rollout --steps 5 --warmup 2"
run() {  # name, rocprof args...
  name=$1; shift
  rocprofv3 "$@" --output-format csv -d $WORK/$name -o $name -- $BENCH > $OUT/${name}_bench.log 2>&1
  for f in $(find $WORK/$name -name "*.csv"); do
    sz=$(stat -c %s $f)
    if [ $sz -lt 400000 ]; then cp $f $OUT/$(basename $f); else
      head -200 $f > $OUT/$(basename $f .csv)_head200.csv
      (head -1 $f; grep -E '"(void )?k_[a-z_0-9]+[<(]' $f | head -3000) > $OUT/$(basename $f .csv)_ours.csv   # rows of this library's kernels
    fi
  done
}
run trace --kernel-trace --stats
if [ "$MODE" = all ]; then
# all kernels of the default bench (planner K3, env step K4, IK K5, rollout) in one kernel-trace pass
BENCH_SAVE=$BENCH; BENCH="python $R/bench.py --no-cpu --steps 5 --warmup 2"
run full --kernel-trace --stats
BENCH=$BENCH_SAVE
fi
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
run pmc_sq2 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
if [ "$MODE" = all ]; then
# the planner kernel's counters (bench with the planner section only)
BENCH_SAVE=$BENCH; BENCH="python $R/bench.py --no-cpu --no-env --no-rollout --steps 2 --warmup 1"
run pmc_plan --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
BENCH=$BENCH_SAVE
fi
run pmc_write --kernel-trace --pmc WRITE_SIZE
ls -la $OUT
echo "== kernel stats"; cat $OUT/trace_kernel_stats.csv 2>/dev/null
tail -1 $OUT/trace_bench.log | cut -c1-300
