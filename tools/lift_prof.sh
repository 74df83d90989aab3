cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/lb.py <<'PY'
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
bench.ENV = "SawyerLiftObstacle-v0"
pi = planner_inputs(bench.ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc)
import numpy as np
from conftest import sample_states
PY
sed -i 's#sys.path.insert(0, "/root/repo/tools")#sys.path.insert(0, "/root/repo/tests")#' /tmp/lb.py
cat >> /tmp/lb.py <<'PY'
qa, row = sample_states(pi, 1 << 20, 5, "uniform")
qn, _ = sample_states(pi, 1 << 19, 6, "near"); qa[1 << 19:] = qn
rows = np.repeat(row, 4096, axis=0)
a, r = torch.tensor(qa, device="cuda"), torch.tensor(rows, device="cuda")
for _ in range(8): bp.is_valid(a, r, samples_per_env=256)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o lp -- python /tmp/lb.py > /dev/null 2>&1
f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-60,150-260
