#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + SQ PMC passes for the dynamics kernels inside env.step:
#   bash tools/dyn_prof.sh <tag> push       K6 k_env_dyn4 (servo dynamics, contact-free)
#   bash tools/dyn_prof.sh <tag> contacts   K7 k_env_dyn_ct (contacts behind the constraint solver)
TAG=${1:-r04_dyn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
WORK=/tmp/prof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/dyn_bench.py 4096 ${2:-push}"
run() {
  name=$1; shift
  rocprofv3 "$@" --output-format csv -d $WORK/$name -o $name -- $CMD > $OUT/${name}.log 2>&1
  for f in $(find $WORK/$name -name "*.csv" 2>/dev/null); do
    sz=$(stat -c %s $f)
    if [ $sz -lt 300000 ]; then cp $f $OUT/$(basename $f); else
      (head -1 $f; grep -E 'k_env_dyn' $f | head -400) > $OUT/$(basename $f .csv)_dyn.csv
    fi
  done
}
run trace --kernel-trace --stats
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
run pmc_sq2 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
ls -la $OUT
cat $OUT/trace_kernel_stats.csv 2>/dev/null | head -8
