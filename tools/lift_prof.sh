#!/bin/bash
# GPU box: kernel-trace split of the two validity passes of SawyerLiftObstacle-v0 (main pass + gated mesh pass).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/lift_bench.py <<PY
import sys; sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np, torch
from conftest import sample_states
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
pi = planner_inputs("SawyerLiftObstacle-v0")
bp = BatchPlanner(_lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range))
qa, row = sample_states(pi, 1 << 20, 5, "uniform")
qn, _ = sample_states(pi, 1 << 19, 6, "near"); qa[1 << 19:] = qn
a, r = torch.tensor(qa, device="cuda"), torch.tensor(np.repeat(row, 4096, axis=0), device="cuda")
for _ in range(8): bp.is_valid(a, r, samples_per_env=256)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o lp -- python /tmp/lift_bench.py > /dev/null 2>&1
f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-60,150-260
