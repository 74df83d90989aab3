"""Large randomized parity sweep on the GPU box: for every scene and a few seeds, N states (half uniform in the joint
box, half near the initial pose, per-env passive perturbations) through the production validity path (verdicts and
penetration depths) and through the CPU oracle on all host cores; prints mismatch counts.  Test infrastructure only."""
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from conftest import SUPPORTED_ENVS, sample_states
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
from oracle import oracle as O
O.build()
N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, 64
tot = bad = 0
for env in SUPPORTED_ENVS:
    pi = planner_inputs(env)
    sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
    bp = BatchPlanner(sc)
    orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    for seed in (1, 2, 3):
        qu, row = sample_states(pi, N // 2, 100 + seed, "uniform")
        qn, _ = sample_states(pi, N - N // 2, 200 + seed, "near")
        qa = np.concatenate([qu, qn]); E = N // S
        rng = np.random.default_rng(seed)
        rows = np.repeat(row, E, axis=0)
        rows[:, pi.passive_joint_idx] += rng.normal(0, 0.002, size=(E, len(pi.passive_joint_idx))) * (rng.random((E, 1)) < 0.5)
        t_qa, t_rows = torch.tensor(qa, device="cuda"), torch.tensor(rows, device="cuda")
        v = bp.is_valid(t_qa, t_rows, samples_per_env=S).cpu().numpy()
        v2, md = bp.is_valid(t_qa, t_rows, samples_per_env=S, want_min_dist=True)
        t0 = time.time(); ov, omd = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=0, want_min_dist=True); dt = time.time() - t0
        m1 = int((v != ov).sum()); m2 = int((v2.cpu().numpy() != ov).sum())
        m3 = int((md.cpu().numpy().view(np.uint64) != np.asarray(omd).view(np.uint64)).sum())
        tot += N; bad += m1 + m2 + m3
        print(f"{env:28s} seed {seed}: {N} states, valid {ov.mean():.3f}, verdict mismatches {m1} / {m2} (with depth), depth bit mismatches {m3}; oracle {N/dt/1e6:.1f} M/s", flush=True)
print("TOTAL states", tot, "mismatches", bad)
