#!/bin/bash
# Runs on the GPU box: PC sampling (rocprofv3, beta) of the planner on the budget-exhausting queries alone.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pcs
rm -rf /tmp/pcs; mkdir -p $OUT /tmp/pcs
cd /tmp && export TMPDIR=/tmp
METHOD=${1:-host_trap}; UNIT=${2:-time}; INTERVAL=${3:-1}
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INTERVAL \
   --output-format csv -d /tmp/pcs -o pcs -- python $R/tools/plan_fail_only.py 3 > $OUT/run.log 2>&1
echo rc=$?
tail -5 $OUT/run.log
find /tmp/pcs -type f | head; 
for f in $(find /tmp/pcs -name "*.csv"); do echo $f; head -3 $f; wc -l $f; done
f=$(find /tmp/pcs -name "*pc_sampling*.csv" | head -1)
if [ -n "$f" ]; then python $R/tools/pc_hist.py $f > $OUT/pc_hist.txt; head -60 $OUT/pc_hist.txt; fi
