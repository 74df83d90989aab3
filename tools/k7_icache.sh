#!/bin/bash
# GPU box: instruction-cache counters of K7 (k_env_dyn_ct) -- is the kernel's code (87 KB walked once per sub-step) served by the 64 KB
# instruction cache?   bash tools/k7_icache.sh [tag] [lib]
TAG=${1:-k7_icache}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
WORK=/tmp/prof_$TAG
rm -rf $WORK; mkdir -p $OUT $WORK
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export MOPA_HIP_LIB=$2
CMD="python $R/tools/dyn_bench.py 4096 contacts"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $WORK/ic -o ic -- $CMD > $OUT/ic.log 2>&1
f=$(find $WORK/ic -name "*counter_collection.csv" | head -1)
(head -1 $f; grep -E 'k_env_dyn' $f) > $OUT/ic_counter_collection_dyn.csv
python - "$OUT/ic_counter_collection_dyn.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (len(v), round(sum(v) / len(v))) for c, v in d.items()})
PY
