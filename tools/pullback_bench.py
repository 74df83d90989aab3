"""Times the two forms of mopa_pullback_batch (wave per env / all candidate rows through K1) on 3000 targets (GPU box)."""
import sys, time, os; sys.path.insert(0, ".")
import torch, numpy as np
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import default_qpos, planner_inputs
ENV = "SawyerPushObstacle-v0"
pi = planner_inputs(ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc)
E = 3000
rng = np.random.default_rng(2)
q0 = default_qpos(ENV, pi.model)
cur = np.repeat(q0[None], E, axis=0); cur[:, :7] += rng.normal(0, 0.05, size=(E, 7))
tgt = cur.copy(); tgt[:, :7] += rng.uniform(-1.0, 1.0, size=(E, 7)) * 0.5
tgt[:, :7] = np.clip(tgt[:, :7], pi.jnt_minimum, pi.jnt_maximum)
c, t = torch.tensor(cur, device="cuda"), torch.tensor(tgt, device="cuda")
for form in ("wave", "batch"):
    os.environ["MOPA_PULLBACK"] = form
    for _ in range(3): r = bp.pullback(c, t, 0.02, 100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = bp.pullback(c, t, 0.02, 100)
    torch.cuda.synchronize(); print(form, "%.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), "invalid at start", int((r[1] > 0).sum()), "max trials", int(r[1].max()), "still invalid", int((~r[2].bool()).sum()))
