#!/bin/bash
# GPU box: rocprofv3 kernel-trace of bench.py's K1 step on one scene (default Lift): average ns per kernel of this library.
#   bash tools/lift_k1_trace.sh [env-name]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_k1
MOPA_BENCH_ENV=${1:-SawyerLiftObstacle-v0} timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_k1 -o trace -- python $R/bench.py --no-cpu --no-plan --no-env --no-rollout --steps 5 --warmup 2 > /dev/null 2>&1
python - $(find /tmp/pm_k1 -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].lstrip('"').startswith(("void k_", "k_")): print(f'{r["Name"][:44]:46s} calls {r["Calls"]:>3s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
