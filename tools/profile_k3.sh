#!/bin/bash
# Runs on the GPU box (via gpurun): K3's evidence for profiles/rNN/ -- the per-pass breakdown of a -DMOPA_PLAN_STATS build
# (csrc/libmopa_hip_stats.so: make -C mopa_rl_amd/csrc, then the stats object linked as tools/README says) and one PMC pass
# on the budget-exhausting queries alone.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/k3_$TAG
WORK=/tmp/k3_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd $R
if [ -f mopa_rl_amd/csrc/libmopa_hip_stats.so ]; then
  MOPA_HIP_LIB=$R/mopa_rl_amd/csrc/libmopa_hip_stats.so python tools/plan_passes.py > $OUT/plan_passes.txt 2>&1
fi
python tools/plan_fail_only.py 3 > $OUT/plan_fail_only.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/trace -o trace -- python $R/tools/plan_fail_only.py 3 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $WORK/pmc -o pmc -- python $R/tools/plan_fail_only.py 3 > $OUT/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $WORK/pmc2 -o pmc2 -- python $R/tools/plan_fail_only.py 3 > $OUT/pmc2.log 2>&1
for f in $(find $WORK -name "*.csv"); do
  (head -1 $f; grep -E "k_rrt_connect" $f | tail -40) > $OUT/$(basename $f .csv)_k3.csv
done
cp $(find $WORK/trace -name "*kernel_stats.csv" | head -1) $OUT/trace_kernel_stats.csv 2>/dev/null
ls -la $OUT; cat $OUT/plan_passes.txt | tail -30; cat $OUT/plan_fail_only.txt
