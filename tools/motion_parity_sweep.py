"""Motion-validation parity sweep on the GPU box: for every scene, N segments of mixed length (planner-range steps and
long jumps) through mopa_check_motion_batch (expanded path) and through the oracle's DiscreteMotionValidator restatement
on all host cores.  Test infrastructure only."""
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from conftest import SUPPORTED_ENVS, sample_states
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
from oracle import oracle as O
O.build()
N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18, 64
tot = bad = 0
for env in SUPPORTED_ENVS:
    pi = planner_inputs(env)
    sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
    bp = BatchPlanner(sc)
    orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    rng = np.random.default_rng(11)
    qa, row = sample_states(pi, N, 300, "near")
    lo, hi = np.asarray(pi.jnt_minimum), np.asarray(pi.jnt_maximum)
    step = rng.normal(0, 1, size=qa.shape); step /= np.abs(step).sum(axis=1, keepdims=True)
    scale = rng.choice([0.5 * pi.spec.range, pi.spec.range, 5 * pi.spec.range, 0.0], size=(N, 1), p=[0.3, 0.4, 0.25, 0.05])
    qb = np.clip(qa + step * scale, lo, hi)
    E = N // S
    rows = np.repeat(row, E, axis=0)
    t_a, t_b, t_r = (torch.tensor(x, device="cuda") for x in (qa, qb, rows))
    v = bp.check_motion(t_a, t_b, t_r, samples_per_env=S).cpu().numpy()
    t0 = time.time(); ov = orc.check_motion_batch(qa, qb, rows, samples_per_env=S, resolution=0.005, nthreads=0); dt = time.time() - t0
    m = int((v != ov).sum()); tot += N; bad += m
    print(f"{env:28s}: {N} segments, valid {ov.mean():.3f}, mismatches {m}; oracle {N/dt/1e6:.2f} M segments/s", flush=True)
print("TOTAL segments", tot, "mismatches", bad)
