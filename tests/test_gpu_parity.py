"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact.

Tolerances: collision verdicts, FK poses, per-pair signed distances, minimum
distances, motion verdicts and planner outputs must be IDENTICAL (== on the
float64 bit patterns) -- both sides follow the same IEEE-754 numerics contract
(DESIGN.md "Numerics"), so any difference is a bug, not round-off.
"""
import numpy as np
import pytest

from conftest import SUPPORTED_ENVS, sample_states

pytestmark = pytest.mark.gpu


# K1 generations: wave-per-state / lane-per-state (the library picks by batch size); "v5-lds" / "v5-slab" pin where the third
# generation keeps a tile's FP32 centres (the library picks by LDS fit: slab for Lift and Assembly, LDS for Push and Pusher)
KERNELS = ["v1", "v2", "v5", "v5-lds", "v5-slab"]


def _scene_with_kernel(kernel, *args, **kw):
    """Create a Scene with MOPA_VALID_KERNEL / MOPA_V5_CENTRES pinned (read once, at scene creation)."""
    import os
    from mopa_rl_amd import _lib
    pins = {}
    if kernel is not None:
        gen, _, cen = kernel.partition("-")
        pins["MOPA_VALID_KERNEL"] = gen
        if cen:
            pins["MOPA_V5_CENTRES"] = cen
    old = {k: os.environ.get(k) for k in pins}
    os.environ.update(pins)
    try:
        return _lib.Scene(*args, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _mk(env, oracle_mod, kernel=None, **scene_kw):
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env)
    sc = _scene_with_kernel(kernel, pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold,
                            range_=pi.spec.range, seed=7, **scene_kw)
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    return pi, sc, orc


def _full(pi, qa, row):
    q = row.copy()
    q[pi.ref_joint_pos_indexes] = qa
    return q


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_fk_bit_exact(env, oracle_mod):
    pi, sc, orc = _mk(env, oracle_mod)
    for mode in ("uniform", "near"):
        qa, rows = sample_states(pi, 64, 11, mode)
        for i in range(len(qa)):
            q = _full(pi, qa[i], rows[0])
            gp, gm = sc.debug_fk(q)
            op, om = orc.fk(q)
            assert np.array_equal(_bits(gp), _bits(op)), f"geom pos differs, state {i}"
            assert np.array_equal(_bits(gm), _bits(om)), f"geom mat differs, state {i}"


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_pair_dist_bit_exact(env, oracle_mod):
    pi, sc, orc = _mk(env, oracle_mod, prune_pairs=False)       # per-pair distances: the full candidate list
    ignored = set(pi.ignored_contacts)
    m = pi.model
    ign_mask = np.array([(min(m.geom_mjid[a], m.geom_mjid[b]), max(m.geom_mjid[a], m.geom_mjid[b])) in ignored
                         for a, b in m.pair_geom])
    nbad = 0
    for mode in ("uniform", "near"):
        qa, rows = sample_states(pi, 96, 5, mode)
        for i in range(len(qa)):
            q = _full(pi, qa[i], rows[0])
            dg = sc.debug_pair_dist(q)
            do = orc.pair_dist(q)
            do[ign_mask] = 1.0e10   # the HIP scene drops ignored pairs altogether
            if not np.array_equal(_bits(dg), _bits(do)):
                bad = np.where(_bits(dg) != _bits(do))[0]
                nbad += len(bad)
                p = bad[0]
                print("mismatch", env, mode, i, "pair", p, m.pair_geom[p], m.geom_type[m.pair_geom[p]], dg[p], do[p])
    assert nbad == 0


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
@pytest.mark.parametrize("mode", ["uniform", "near"])
@pytest.mark.parametrize("kernel", KERNELS)
def test_is_valid_batch_matches_oracle(env, mode, kernel, oracle_mod):
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk(env, oracle_mod, kernel)
    bp = BatchPlanner(sc)
    E, S = 8, 512
    rng = np.random.default_rng(3)
    qa, row = sample_states(pi, E * S, 17, mode)
    rows = np.repeat(row, E, axis=0)
    # perturb passive entries per env where the scene has them (gripper slides / object pose)
    if "Sawyer" in env:
        rows[:, 7:9] = rng.uniform(-0.008, 0.015, size=(E, 2))
    ov, omd = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=0)
    tq = torch.from_numpy(qa).cuda()
    tr = torch.from_numpy(rows).cuda()
    v, md = bp.is_valid(tq, tr, samples_per_env=S, want_min_dist=True)
    v2 = bp.is_valid(tq, tr, samples_per_env=S)
    torch.cuda.synchronize()
    v, md, v2 = v.cpu().numpy(), md.cpu().numpy(), v2.cpu().numpy()
    assert np.array_equal(v, ov), f"{(v != ov).sum()} verdicts differ (min-dist kernel)"
    assert np.array_equal(v2, ov), f"{(v2 != ov).sum()} verdicts differ (early-out kernel)"
    assert np.array_equal(_bits(md), _bits(omd)), f"{(_bits(md) != _bits(omd)).sum()} min distances differ"
    # both classes must be exercised
    assert 0 < ov.sum() < len(ov) or mode == "uniform"


@pytest.mark.parametrize("kernel", KERNELS)
def test_plane_pairs_and_per_env_objects(kernel, oracle_mod):
    """The ground plane only ever touches the free-floating cube in these scenes: park the cube of some envs in the
    floor / in the arm's way so that plane-box and moving-vs-moving pairs decide the verdict."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    env = "SawyerPushObstacle-v0"
    pi, sc, orc = _mk(env, oracle_mod, kernel)
    bp = BatchPlanner(sc)
    E, S = 16, 128
    qa, row = sample_states(pi, E * S, 91, "near")
    rows = np.repeat(row, E, axis=0)
    cube = pi.model.get_joint_qpos_addr("cube")
    rng = np.random.default_rng(5)
    rows[0:4, cube:cube + 3] = [1.6, 0.9, 0.02]          # 1 cm into the floor, away from everything else
    rows[4:8, cube:cube + 3] = [1.6, 0.9, 0.0305]        # 0.5 mm above the -2 mm threshold... still penetrating < 2 mm
    rows[8:12, cube:cube + 3] = [0.6, 0.1, 1.25] + rng.normal(0, 0.05, (4, 3))   # floating inside the arm's workspace
    q = rng.normal(size=(4, 4))
    rows[8:12, cube + 3:cube + 7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    ov, omd = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=0)
    assert ov[:4 * S].sum() == 0 and np.all(omd[:4 * S] <= -0.0099)            # plane-box decides
    assert 0 < ov[8 * S:12 * S].sum() < 4 * S
    tq, tr = torch.from_numpy(qa).cuda(), torch.from_numpy(rows).cuda()
    v, md = bp.is_valid(tq, tr, samples_per_env=S, want_min_dist=True)
    v2 = bp.is_valid(tq, tr, samples_per_env=S)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), ov) and np.array_equal(v2.cpu().numpy(), ov)
    assert np.array_equal(_bits(md.cpu().numpy()), _bits(omd))
    # the single-state (wave-per-state) path must agree as well
    for e in (0, 5, 9):
        qf = rows[e].copy()
        qf[:7] = qa[e * S + 3]
        assert sc.is_valid_state(qf, want_min_dist=True) == (bool(ov[e * S + 3]), omd[e * S + 3])


@pytest.mark.parametrize("cap", [0, 64, 100000])
def test_lift_mesh_row_list_and_its_overflow(cap, oracle_mod, monkeypatch):
    """The gate of k_is_valid_v5 hands the mesh pairs within reach to k_mesh_rows as complete records; rows beyond the list's capacity
    put their state on the state list of the second (MESH) pass instead.  With the can parked at the gripper of most envs (thousands of
    rows) the verdicts and depths must equal the oracle's whether every row fits (cap 100000), almost none does (64), or the row list is
    off (0: every state with a mesh pair within reach goes through the state-list pass)."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    monkeypatch.setenv("MOPA_MESH_ROWS_CAP", str(cap))          # (read per call)
    env = "SawyerLiftObstacle-v0"
    pi, sc, orc = _mk(env, oracle_mod, "v5")
    m = pi.model
    bp = BatchPlanner(sc)
    E, S = 48, 256
    qa, row = sample_states(pi, E * S, 77, "near")
    rows = np.repeat(row, E, axis=0)
    ca = m.get_joint_qpos_addr("cube")
    rng = np.random.default_rng(5)
    names = [m.all_geom_names[i] for i in m.geom_mjid]
    claw = [i for i, n in enumerate(names) if "claw" in n or "finger" in n][0]
    for e in range(40):
        base = qa[e * S].copy()
        qa[e * S:(e + 1) * S] = np.clip(base + rng.normal(0, 0.05, (S, len(base))), pi.jnt_minimum, pi.jnt_maximum)
        gpos, _ = orc.fk(_full(pi, base, row[0]))
        rows[e, ca:ca + 3] = gpos[claw] + rng.normal(0, 0.04, 3)
    quat = rng.normal(size=(40, 4))
    rows[:40, ca + 3:ca + 7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    ov, omd = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=0)
    tq, tr = torch.from_numpy(qa).cuda(), torch.from_numpy(rows).cuda()
    for _ in range(2):          # (twice: the counters of the row list / the state list are reset per call)
        v, md = bp.is_valid(tq, tr, samples_per_env=S, want_min_dist=True)
        v2 = bp.is_valid(tq, tr, samples_per_env=S)
        torch.cuda.synchronize()
        assert np.array_equal(v.cpu().numpy(), ov) and np.array_equal(v2.cpu().numpy(), ov)
        assert np.array_equal(_bits(md.cpu().numpy()), _bits(omd))
    assert 0.02 < ov.mean() < 0.98


@pytest.mark.parametrize("kernel", KERNELS)
def test_lift_can_mesh_pairs_decide(kernel, oracle_mod):
    """SawyerLiftObstacle: the can is a convex mesh.  Park it (random orientation) around the gripper / in the arm's
    workspace / half-way into the table so that primitive-vs-mesh (MPR on the hull) and plane/table-vs-mesh pairs
    decide verdicts and depths; per-pair distances of mesh pairs must be bit-identical as well."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.mjcf import GEOM_MESH
    env = "SawyerLiftObstacle-v0"
    pi, sc, orc = _mk(env, oracle_mod, kernel, prune_pairs=False)      # (per-pair distances below: the full candidate list)
    m = pi.model
    bp = BatchPlanner(sc)
    E, S = 24, 128
    qa, row = sample_states(pi, E * S, 31, "near")
    rows = np.repeat(row, E, axis=0)
    ca = m.get_joint_qpos_addr("cube")
    rng = np.random.default_rng(12)
    names = [m.all_geom_names[i] for i in m.geom_mjid]
    claw = [i for i, n in enumerate(names) if "claw" in n or "finger" in n][0]
    for e in range(16):   # the can floats next to THIS env's gripper; the env's states jitter around one arm pose
        base = qa[e * S].copy()
        qa[e * S:(e + 1) * S] = np.clip(base + rng.normal(0, 0.04, (S, len(base))), pi.jnt_minimum, pi.jnt_maximum)
        gpos, _ = orc.fk(_full(pi, base, row[0]))
        rows[e, ca:ca + 3] = gpos[claw] + rng.normal(0, 0.035, 3)
    rows[16:20, ca + 2] -= rng.uniform(0.0, 0.03, 4)                  # sunk into the table top
    quat = rng.normal(size=(20, 4))
    rows[:20, ca + 3:ca + 7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    ov, omd = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=0)
    tq, tr = torch.from_numpy(qa).cuda(), torch.from_numpy(rows).cuda()
    v, md = bp.is_valid(tq, tr, samples_per_env=S, want_min_dist=True)
    v2 = bp.is_valid(tq, tr, samples_per_env=S)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), ov) and np.array_equal(v2.cpu().numpy(), ov)
    assert np.array_equal(_bits(md.cpu().numpy()), _bits(omd))
    # mesh pairs really were decisive somewhere, and their distances agree bit for bit
    g = int(np.where(m.geom_type == GEOM_MESH)[0][0])
    ignored = set(pi.ignored_contacts)
    ign_mask = np.array([(min(m.geom_mjid[a], m.geom_mjid[b]), max(m.geom_mjid[a], m.geom_mjid[b])) in ignored
                         for a, b in m.pair_geom])
    mesh_pairs = np.where((m.pair_geom == g).any(axis=1) & ~ign_mask)[0]
    n_hit = 0
    for i in list(range(0, 20 * S, 97)):
        qf = _full(pi, qa[i], rows[i // S])
        od, dd = orc.pair_dist(qf), sc.debug_pair_dist(qf)
        od[ign_mask] = 1.0e10   # the HIP scene drops ignored pairs altogether
        assert np.array_equal(_bits(od), _bits(dd))
        n_hit += int((od[mesh_pairs] < 0).any())
    assert n_hit >= 3 and 0 < ov[:20 * S].sum() < 20 * S


@pytest.mark.parametrize("kernel", KERNELS)
def test_offcentre_joint_anchors(kernel, oracle_mod):
    """No reference scene has a joint anchor away from the body origin, and the FK takes a shortcut for that case:
    exercise the general branch on a model with random anchors (FK poses, verdicts and depths, both K1 kernels)."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs
    from test_oracle_fk import _with_offcentre_anchors
    env = "SawyerPushObstacle-v0"
    pi = planner_inputs(env)
    m = _with_offcentre_anchors(pi.model)
    sc = _scene_with_kernel(kernel, m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
    orc = oracle_mod.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    qa, row = sample_states(pi, 4096, 77, "near")
    for i in range(0, 4096, 517):
        q = _full(pi, qa[i], row[0])
        gp, gm = sc.debug_fk(q)
        op, om = orc.fk(q)
        assert np.array_equal(_bits(gp), _bits(op)) and np.array_equal(_bits(gm.reshape(-1, 9)), _bits(om.reshape(-1, 9)))
    ov, omd = orc.is_valid_batch(qa, row, samples_per_env=len(qa), nthreads=0)
    v, md = BatchPlanner(sc).is_valid(torch.from_numpy(qa).cuda(), torch.from_numpy(row).cuda(), samples_per_env=len(qa),
                                      want_min_dist=True)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), ov) and np.array_equal(_bits(md.cpu().numpy()), _bits(omd))
    assert 0 < ov.sum() < len(ov)


def test_kernel_auto_selection_is_result_invariant(oracle_mod):
    """Default scene: small batches take the wave-per-state kernel, large ones the lane-per-state kernel; the verdicts
    and depths of the same states must not depend on which one ran."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk("SawyerPushObstacle-v0", oracle_mod)
    bp = BatchPlanner(sc)
    n_big = 64 * 1024
    qa, row = sample_states(pi, n_big, 23, "near")
    tq, tr = torch.from_numpy(qa).cuda(), torch.from_numpy(row).cuda()
    v_big, md_big = bp.is_valid(tq, tr, samples_per_env=n_big, want_min_dist=True)
    v_small, md_small = bp.is_valid(tq[:1000].contiguous(), tr, samples_per_env=1000, want_min_dist=True)
    torch.cuda.synchronize()
    ov, omd = orc.is_valid_batch(qa[:1000], row, samples_per_env=1000, nthreads=0)
    assert np.array_equal(v_small.cpu().numpy(), ov) and np.array_equal(v_big[:1000].cpu().numpy(), ov)
    assert np.array_equal(_bits(md_small.cpu().numpy()), _bits(omd)) and np.array_equal(_bits(md_big[:1000].cpu().numpy()), _bits(omd))


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_single_state_api(env, oracle_mod):
    pi, sc, orc = _mk(env, oracle_mod)
    qa, rows = sample_states(pi, 32, 23, "near")
    for i in range(len(qa)):
        q = _full(pi, qa[i], rows[0])
        v, md = sc.is_valid_state(q, want_min_dist=True)
        ov, omd = orc.is_valid(q)
        assert v == ov and md == omd
        assert sc.is_valid_state(q) == ov


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
@pytest.mark.parametrize("kernel", KERNELS)   # v1: wave per segment; v2 / v5: segments expanded into states (mopa_motion.inc)
def test_check_motion_matches_oracle(env, kernel, oracle_mod):
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk(env, oracle_mod, kernel)
    bp = BatchPlanner(sc)
    n = 2048
    qa, row = sample_states(pi, n, 31, "near")
    rng = np.random.default_rng(9)
    step = rng.normal(0, 0.05, size=qa.shape)
    step[n // 2:] *= 6.0   # long segments: a dozen interior states
    step[: n // 8] = 0.0   # degenerate zero-length segments
    qb = np.clip(qa + step, pi.jnt_minimum, pi.jnt_maximum)
    ov = orc.check_motion_batch(qa, qb, row, samples_per_env=n, nthreads=0)
    v = bp.check_motion(torch.from_numpy(qa).cuda(), torch.from_numpy(qb).cuda(), torch.from_numpy(row).cuda(),
                        samples_per_env=n)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), ov), f"{(v.cpu().numpy() != ov).sum()} motion verdicts differ"
    assert 0 < ov.sum() < n


def test_check_motion_large_batch_crosses_scan_chunks(oracle_mod):
    """The expanded motion path scans the per-segment state counts in blocks of 256 and then the block sums 1024 at a time
    (mopa_motion.inc): 300 000 segments = 1172 blocks, i.e. more than one pass of the block-sum scan, with ragged counts
    (zero-length, one-state and long segments mixed) and a last block that is not full.  Verdicts against the oracle."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk("PusherObstacle-v0", oracle_mod)
    bp = BatchPlanner(sc)
    n = 300_000
    qa, row = sample_states(pi, n, 77, "near")
    rng = np.random.default_rng(5)
    step = rng.normal(0, 0.03, size=qa.shape) * rng.choice([0.0, 0.2, 1.0, 8.0], size=(n, 1))
    qb = np.clip(qa + step, pi.jnt_minimum, pi.jnt_maximum)
    ov = orc.check_motion_batch(qa, qb, row, samples_per_env=n, nthreads=0)
    v = bp.check_motion(torch.from_numpy(qa).cuda(), torch.from_numpy(qb).cuda(), torch.from_numpy(row).cuda(), samples_per_env=n)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), ov), f"{(v.cpu().numpy() != ov).sum()} motion verdicts differ"
    assert 0.05 * n < ov.sum() < 0.95 * n


def test_laddered_planning_equals_one_full_launch(oracle_mod):
    """`BatchPlanner.plan_laddered`: a short first launch, the unsolved queries again with the full budget on another stream,
    several batches in flight -- every batch's results equal those of one launch with the full budget."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk("SawyerPushObstacle-v0", oracle_mod)
    bp = BatchPlanner(sc)
    E = 256
    qa, row = sample_states(pi, 8000, 47, "near")
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
    good = qa[ov == 1]
    batches = []
    for b in range(3):
        starts = np.repeat(row, E, axis=0)
        goals = starts.copy()
        starts[:, pi.ref_joint_pos_indexes] = good[2 * b * E:(2 * b + 1) * E]
        goals[:, pi.ref_joint_pos_indexes] = good[(2 * b + 1) * E:(2 * b + 2) * E]
        batches.append(dict(start=torch.from_numpy(starts).cuda(), goal=torch.from_numpy(goals).cuda(), seed=11 + b))
    prm = dict(max_iters=400, max_nodes=512, max_path=128)
    lad = bp.plan_laddered(batches, first_iters=40, **prm)
    torch.cuda.synchronize()
    n_retried = 0
    for b, got in zip(batches, lad):
        want = bp.plan(b["start"], b["goal"], seed=b["seed"], **prm)
        first = bp.plan(b["start"], b["goal"], seed=b["seed"], max_iters=40, max_nodes=512, max_path=128)
        n_retried += int((first[2] == -4).sum())
        w, g = [t.cpu().numpy() for t in want], [t.cpu().numpy() for t in got]
        for k in (1, 2, 3):
            assert np.array_equal(w[k], g[k])
        for e in range(E):
            assert np.array_equal(w[0][e, :w[1][e]].view(np.uint64), g[0][e, :g[1][e]].view(np.uint64))
    assert n_retried > 20


def test_plan_results_do_not_depend_on_launch_shape(oracle_mod):
    """A query's outcome is a function of (start, goal, its seed, its stream id) only: capping the launch's persistent
    workgroups, giving per-query seeds instead of one seed, or planning a subset of the queries changes nothing."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk("SawyerPushObstacle-v0", oracle_mod)
    bp = BatchPlanner(sc)
    E = 96
    qa, row = sample_states(pi, 6000, 43, "near")
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
    good = qa[ov == 1]
    starts = np.repeat(row, E, axis=0)
    goals = starts.copy()
    starts[:, pi.ref_joint_pos_indexes] = good[:E]
    goals[:, pi.ref_joint_pos_indexes] = good[E:2 * E]
    s, g = torch.from_numpy(starts).cuda(), torch.from_numpy(goals).cuda()
    prm = dict(max_iters=200, max_nodes=256, max_path=128)
    ids = torch.arange(E, device="cuda", dtype=torch.int64)
    base = [t.cpu().numpy() for t in bp.plan(s, g, seed=5, env_ids=ids, **prm)]
    capped = [t.cpu().numpy() for t in bp.plan(s, g, seed=5, env_ids=ids, max_workgroups=3, **prm)]
    # all resident workgroup slots (two per CU) / a capped launch that keeps other planner workgroups off its CUs
    wide = [t.cpu().numpy() for t in bp.plan(s, g, seed=5, env_ids=ids, max_workgroups=-1, **prm)]
    excl = [t.cpu().numpy() for t in bp.plan(s, g, seed=5, env_ids=ids, max_workgroups=8, exclusive=True, **prm)]
    seeds = torch.full((E,), 5, device="cuda", dtype=torch.int64)
    seeded = [t.cpu().numpy() for t in bp.plan(s, g, seed=999, env_ids=ids, seeds=seeds, **prm)]
    sub = torch.arange(0, E, 3, device="cuda", dtype=torch.int64)
    part = [t.cpu().numpy() for t in bp.plan(s[sub].contiguous(), g[sub].contiguous(), seed=5, env_ids=sub.contiguous(), **prm)]
    torch.cuda.synchronize()
    assert (base[2] == 0).sum() > E // 4

    def same(x, y):      # (path, path_len, status, n_checks): rows past a path's length are never written
        for k in (1, 2, 3):
            assert np.array_equal(x[k], y[k])
        for e in range(len(x[1])):
            assert np.array_equal(x[0][e, :x[1][e]].view(np.uint64), y[0][e, :y[1][e]].view(np.uint64))
    same(base, capped)
    same(base, wide)
    same(base, excl)
    same(base, seeded)
    same([np.ascontiguousarray(a[::3]) for a in base], part)


@pytest.mark.parametrize("env", ["SawyerPushObstacle-v0", "PusherObstacle-v0"])
def test_plan_matches_oracle(env, oracle_mod):
    """Same (seed, env id) sample stream => identical tree growth, path, status and check count."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk(env, oracle_mod)
    bp = BatchPlanner(sc)
    E = 16
    qa, row = sample_states(pi, 4000, 41, "near")
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
    good = qa[ov == 1]
    assert len(good) >= 2 * E
    starts = np.repeat(row, E, axis=0)
    goals = starts.copy()
    starts[:, pi.ref_joint_pos_indexes] = good[:E]
    goals[:, pi.ref_joint_pos_indexes] = good[E:2 * E]
    # one query with an invalid goal
    bad = qa[ov == 0]
    if len(bad):
        goals[0, pi.ref_joint_pos_indexes] = bad[0]
    max_iters, max_nodes, max_path = 300, 512, 256
    path, plen, status, nchk = bp.plan(torch.from_numpy(starts).cuda(), torch.from_numpy(goals).cuda(),
                                       max_iters=max_iters, max_nodes=max_nodes, max_path=max_path, seed=123, env_id_base=5)
    torch.cuda.synchronize()
    path, plen, status, nchk = path.cpu().numpy(), plen.cpu().numpy(), status.cpu().numpy(), nchk.cpu().numpy()
    n_ok = 0
    for e in range(E):
        st, opath, ochk, _ = orc.plan(starts[e], goals[e], pi.spec.range, 0.005, max_iters, max_nodes, seed=123,
                                      env_id=5 + e, max_path=max_path)
        assert status[e] == st, f"env {e}: status {status[e]} vs oracle {st}"
        assert plen[e] == len(opath)
        assert nchk[e] == ochk, f"env {e}: n_checks {nchk[e]} vs oracle {ochk}"
        assert np.array_equal(_bits(path[e, :plen[e]]), _bits(opath)), f"env {e}: path differs"
        n_ok += int(st == 0)
    if len(bad):
        assert status[0] == -5
    assert n_ok >= 1


def test_plan_with_nine_active_coordinates(oracle_mod):
    """The gripper slides made active (9 coordinates, two of them with a 23 mm extent => motions of >100 states):
    the planner's generic nearest-neighbour path (> 8 coordinates) and long bisection queues, against the oracle."""
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs("SawyerPushObstacle-v0")
    passive = [a for a in pi.passive_joint_idx if a not in (7, 8)]
    sc = _lib.Scene(pi.model, passive, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range, seed=7)
    orc = oracle_mod.OracleScene(pi.model, passive, pi.ignored_contacts, pi.spec.contact_threshold)
    bp = BatchPlanner(sc)
    rng = np.random.default_rng(3)
    arm, row = sample_states(pi, 2000, 43, "near")
    qa = np.concatenate([arm, rng.uniform(-0.008, 0.015, size=(len(arm), 2))], axis=1)
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
    good = qa[ov == 1]
    E = 8
    assert len(good) >= 2 * E
    act = list(range(9))
    starts, goals = np.repeat(row, E, axis=0), np.repeat(row, E, axis=0)
    starts[:, act] = good[:E]
    goals[:, act] = good[E:2 * E]
    max_iters, max_nodes, max_path = 150, 512, 256
    path, plen, status, nchk = bp.plan(torch.from_numpy(starts).cuda(), torch.from_numpy(goals).cuda(),
                                       max_iters=max_iters, max_nodes=max_nodes, max_path=max_path, seed=77, env_id_base=0)
    torch.cuda.synchronize()
    path, plen, status, nchk = path.cpu().numpy(), plen.cpu().numpy(), status.cpu().numpy(), nchk.cpu().numpy()
    for e in range(E):
        st, opath, ochk, _ = orc.plan(starts[e], goals[e], pi.spec.range, 0.005, max_iters, max_nodes, seed=77, env_id=e,
                                      max_path=max_path)
        assert status[e] == st and plen[e] == len(opath) and nchk[e] == ochk, (e, status[e], st, nchk[e], ochk)
        assert np.array_equal(_bits(path[e, :plen[e]]), _bits(opath)), f"env {e}: path differs"
    assert (status == 0).sum() >= 1 and nchk.max() > 100


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
@pytest.mark.parametrize("kernel", KERNELS)
def test_hip_matches_committed_golden(env, kernel):
    """HIP path vs the committed fixtures (tests/golden/*.npz, generated by tools/gen_golden.py) -- no oracle
    in the loop, so this also runs where the oracle cannot be built."""
    import os
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", pi.spec.scene + ".npz"))
    sc = _scene_with_kernel(kernel, pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold,
                            range_=pi.spec.range)
    bp = BatchPlanner(sc)
    qa, row = torch.from_numpy(g["q_active"]).cuda(), torch.from_numpy(g["qpos_env"]).cuda()
    v, md = bp.is_valid(qa, row, samples_per_env=len(qa), want_min_dist=True)
    mv = bp.check_motion(qa, torch.from_numpy(g["motion_qb"]).cuda(), row, samples_per_env=len(qa))
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), g["valid"])
    assert np.array_equal(_bits(md.cpu().numpy()), _bits(g["min_dist"]))
    assert np.array_equal(mv.cpu().numpy(), g["motion_valid"])
    for q, gp, gm in zip(g["fk_qpos"], g["fk_geom_pos"], g["fk_geom_mat"]):
        p, m = sc.debug_fk(q)
        assert np.array_equal(_bits(p), _bits(gp)) and np.array_equal(_bits(m.reshape(-1, 9)), _bits(gm.reshape(-1, 9)))
    it, nodes, mp, seed = (int(x) for x in g["plan_params"])
    path, plen, status, nchk = bp.plan(torch.from_numpy(g["plan_start"]).cuda(), torch.from_numpy(g["plan_goal"]).cuda(),
                                       max_iters=it, max_nodes=nodes, max_path=mp, seed=seed, env_id_base=0)
    torch.cuda.synchronize()
    assert np.array_equal(status.cpu().numpy(), g["plan_status"]) and np.array_equal(plen.cpu().numpy(), g["plan_len"])
    assert np.array_equal(nchk.cpu().numpy(), g["plan_checks"])
    assert np.array_equal(_bits(path.cpu().numpy()), _bits(g["plan_path"]))


def test_reference_api_end_to_end():
    """PlannerAgent.plan()/isValidState() exactly as rl/sac_agent.py:84-110 constructs and calls them."""
    import types
    from mopa_rl_amd.planner_agent import PlannerAgent
    from mopa_rl_amd.scene import default_qpos, planner_inputs
    env = "SawyerPushObstacle-v0"
    pi = planner_inputs(env)
    cfg = types.SimpleNamespace(planner_type="rrt_connect", range=pi.spec.range, planner_objective="path_length",
                                threshold=0.0, seed=3, _xml_path="sawyer_push_obstacle.xml",
                                contact_threshold=pi.spec.contact_threshold, timelimit=1.0)
    agent = PlannerAgent(cfg, 7, pi.non_limited_idx, passive_joint_idx=pi.passive_joint_idx,
                         ignored_contacts=pi.ignored_contacts, planner_type="rrt_connect", range_=pi.spec.range)
    q = default_qpos(env, pi.model)
    assert agent.isValidState(q) is True
    assert agent.get_planner_status() == "none"
    goal = q.copy()
    goal[:7] += [0.3, -0.2, 0.1, 0.2, -0.1, 0.2, 0.3]
    assert agent.isValidState(goal)
    traj, success, valid, exact = agent.plan(q, goal, timelimit=1.0)
    assert success and valid and exact
    np.testing.assert_array_equal(traj[-1], goal)
    assert np.all(np.abs(np.diff(np.vstack([q, traj])[:, :7], axis=0)).sum(axis=1) <= pi.spec.range + 1e-12)
    assert agent.get_planner_status() == "Exact solution"
    # invalid goal: arm folded into the pedestal
    bad = q.copy()
    bad[:7] = [0.0, 1.2, 0.0, 3.0, 0.0, 0.0, 0.0]
    if not agent.isValidState(bad):
        traj, success, valid, exact = agent.plan(q, bad)
        assert not success and not valid and traj.shape == (1, pi.model.nq) and np.all(traj == -5)


@pytest.mark.parametrize("env,prod,want_kernel", [
    ("SawyerPushObstacle-v0", "v5-lds", "k_is_valid_v5"),       # what bench.py times (the library's own choice for Push)
    ("SawyerPushObstacle-v0", "v5-slab", "k_is_valid_v5"),
    ("SawyerLiftObstacle-v0", None, "k_is_valid_v5"),           # library's choice: slab centres + mesh gate + gated mesh pass
    ("SawyerAssemblyObstacle-v0", None, "k_is_valid_v5"),       # library's choice: slab centres (34 moving geoms)
])
def test_full_size_batch_properties(env, prod, want_kernel, oracle_mod):
    """BASELINE.json's full single-GPU size (4096 envs x 256 states = 1 048 576 checks, the bench workload) through the
    PRODUCTION kernel and size-independent properties: (i) it agrees with both older K1 generations on every state
    (verdicts and depths), with and without the per-state early-out, (ii) verdicts and depths are equivariant under a
    permutation of the states within each env (no dependence on tile / lane / queue position), (iii) a state checked as a
    zero-length motion gets the same verdict, (iv) a random 16k-state subset equals the CPU oracle bit for bit."""
    import torch
    from bench import make_inputs
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env)
    E, S = 4096, 256
    dev = torch.device("cuda:0")
    qa, rows = make_inputs(torch, pi, E, S, 1234, dev)
    mk = lambda k: _scene_with_kernel(k, pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
    scp, sc2, sc1 = mk(prod), mk("v2"), mk("v1")
    assert scp.valid_kernel(E * S) == want_kernel and sc2.valid_kernel(E * S) == "k_is_valid_v2" and sc1.valid_kernel(E * S) == "k_is_valid"
    bpp, bp2, bp1 = BatchPlanner(scp), BatchPlanner(sc2), BatchPlanner(sc1)
    vp_, mdp_ = bpp.is_valid(qa, rows, samples_per_env=S, want_min_dist=True)
    vpe = bpp.is_valid(qa, rows, samples_per_env=S)                      # early-out instantiation (what the bench times)
    v2, md2 = bp2.is_valid(qa, rows, samples_per_env=S, want_min_dist=True)
    v1, md1 = bp1.is_valid(qa, rows, samples_per_env=S, want_min_dist=True)
    assert torch.equal(v1, v2) and torch.equal(v2, vp_) and torch.equal(vp_, vpe)
    assert torch.equal(md1.view(torch.int64), md2.view(torch.int64)) and torch.equal(md2.view(torch.int64), mdp_.view(torch.int64))
    assert 0.1 < vp_.float().mean().item() < 0.9
    # (ii) permute the states inside every env
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    perm = torch.argsort(torch.rand(E, S, generator=g, device=dev), dim=1)
    flat = (perm + torch.arange(E, device=dev)[:, None] * S).reshape(-1)
    vq, mdq = bpp.is_valid(qa[flat].contiguous(), rows, samples_per_env=S, want_min_dist=True)
    assert torch.equal(vq, vp_[flat]) and torch.equal(mdq.view(torch.int64), mdp_[flat].view(torch.int64))
    # (iii) zero-length motions
    sub = slice(0, 65536)
    mv = bpp.check_motion(qa[sub].contiguous(), qa[sub].contiguous(), rows, samples_per_env=S)
    assert torch.equal(mv, vp_[sub])
    # (iv) oracle on a random subset
    idx = torch.randperm(E * S, generator=g, device=dev)[:16384].sort().values
    qh, rh = qa[idx].cpu().numpy(), rows.cpu().numpy()
    ov = np.zeros(len(idx), dtype=np.uint8)
    omd = np.zeros(len(idx))
    envs = (idx // S).cpu().numpy()
    for e in np.unique(envs):
        m = envs == e
        ov[m], omd[m] = orc_batch(oracle_mod, pi, qh[m], rh[e:e + 1])
    assert np.array_equal(ov, vpe[idx].cpu().numpy()) and np.array_equal(_bits(omd), _bits(mdp_[idx].cpu().numpy()))


def test_two_streams_share_a_scene(oracle_mod):
    """A scene keeps its launch scratch (pose slabs, deferred-pair ring, tile counter, mesh work list, motion / planner
    buffers) per stream: validity launches of different batches on two streams, in flight together, give the results of
    the same launches run one after the other -- for the plain third-generation path (Push) and the gated mesh path (Lift)."""
    import torch
    from bench import make_inputs
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs
    dev = torch.device("cuda:0")
    for env in ("SawyerPushObstacle-v0", "SawyerLiftObstacle-v0"):
        pi, sc, _ = _mk(env, oracle_mod)
        bp = BatchPlanner(sc)
        S = 64
        qa1, rows1 = make_inputs(torch, pi, 1024, S, 1, dev)
        qa2, rows2 = make_inputs(torch, pi, 2048, S, 2, dev)
        ref1, ref2 = bp.is_valid(qa1, rows1, samples_per_env=S).clone(), bp.is_valid(qa2, rows2, samples_per_env=S).clone()
        mref = bp.check_motion(qa1[:8192].contiguous(), qa1[8192:16384].contiguous(), rows1, samples_per_env=S).clone()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for it in range(12):
            o1, o2 = torch.empty_like(ref1), torch.empty_like(ref2)
            with torch.cuda.stream(s1):
                bp.is_valid(qa1, rows1, samples_per_env=S, out=o1, stream=s1)
            with torch.cuda.stream(s2):
                bp.is_valid(qa2, rows2, samples_per_env=S, out=o2, stream=s2)
                m2 = bp.check_motion(qa1[:8192].contiguous(), qa1[8192:16384].contiguous(), rows1, samples_per_env=S, stream=s2) if it % 4 == 0 else None
            outs.append((o1, o2, m2))
        torch.cuda.synchronize()
        for o1, o2, m2 in outs:
            assert torch.equal(o1, ref1) and torch.equal(o2, ref2)
            assert m2 is None or torch.equal(m2, mref)
        assert 0.05 < ref1.float().mean().item() < 0.95


_ORC_CACHE = {}


def orc_batch(oracle_mod, pi, qa, row):
    key = pi.spec.env
    if key not in _ORC_CACHE:
        _ORC_CACHE[key] = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    return _ORC_CACHE[key].is_valid_batch(qa, row, samples_per_env=len(qa), nthreads=0)


@pytest.mark.parametrize("kernel", ["v2", "v5"])
@pytest.mark.parametrize("N,S", [(64, 64), (65, 13), (1000, 7), (4097, 100)])
def test_ragged_batch_sizes(kernel, N, S, oracle_mod):
    """Tile remainders, envs that straddle tiles, samples_per_env that does not divide 64 (lane-per-state kernels)."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    env = "SawyerPushObstacle-v0"
    pi, sc, orc = _mk(env, oracle_mod, kernel)
    E = (N + S - 1) // S
    qa, row = sample_states(pi, N, 5 + N, "near")
    rows = np.repeat(row, E, axis=0)
    rng = np.random.default_rng(N)
    rows[:, 7:9] = rng.uniform(-0.008, 0.015, size=(E, 2))
    cube = pi.model.get_joint_qpos_addr("cube")
    rows[:, cube:cube + 2] += rng.uniform(-0.05, 0.05, size=(E, 2))
    ov, omd = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=0)
    v, md = BatchPlanner(sc).is_valid(torch.from_numpy(qa).cuda(), torch.from_numpy(rows).cuda(), samples_per_env=S, want_min_dist=True)
    v2 = BatchPlanner(sc).is_valid(torch.from_numpy(qa).cuda(), torch.from_numpy(rows).cuda(), samples_per_env=S)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), ov) and np.array_equal(v2.cpu().numpy(), ov)
    assert np.array_equal(_bits(md.cpu().numpy()), _bits(omd))


def test_plan_full_size(oracle_mod):
    """K3 at BASELINE config 3's size -- 4096 Push queries, 2000 iterations, 4096 nodes per tree (the bench's queries): the
    persistent-wave scheduler, full trees and budget exhaustion are what the small planner tests do not reach.
    Oracle (bit-exact status / path / consumed checks) on a random 128-query subset through a thread pool; on ALL queries:
    exact endpoints, every edge <= range in the L1 metric and valid under K2, results invariant under a workgroup cap and
    under a permutation of the queries (sample streams are keyed by the query's id, not its slot)."""
    import os
    import sys
    from concurrent.futures import ThreadPoolExecutor
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk("SawyerPushObstacle-v0", oracle_mod)
    bp = BatchPlanner(sc)
    dev = torch.device("cuda", 0)
    E = 4096
    start, goal = bench.planner_queries(torch, bp, pi, E, dev)
    prm = dict(max_iters=2000, max_nodes=4096, max_path=256, seed=7)
    path, plen, status, nchk = bp.plan(start, goal, **prm)
    torch.cuda.synchronize()
    path_h, plen_h, st_h, chk_h = path.cpu().numpy(), plen.cpu().numpy(), status.cpu().numpy(), nchk.cpu().numpy()
    assert set(np.unique(st_h)) <= {0, -4} and (st_h == 0).mean() > 0.9 and (st_h == -4).sum() > 0
    # --- properties on all queries ---
    ok = np.where(st_h == 0)[0]
    s_h, g_h = start.cpu().numpy(), goal.cpu().numpy()
    arm = list(pi.ref_joint_pos_indexes)
    qa, qb, owner = [], [], []
    for e in ok:
        p = path_h[e, :plen_h[e]]
        assert np.array_equal(_bits(p[0]), _bits(s_h[e])) and np.array_equal(_bits(p[-1][arm]), _bits(g_h[e][arm])), e
        d = np.abs(np.diff(p[:, arm], axis=0)).sum(1)
        assert d.max() <= pi.spec.range * (1 + 1e-12), (e, d.max())
        qa.append(p[:-1, arm]); qb.append(p[1:, arm]); owner.append(np.full(len(p) - 1, e))
    qa, qb, owner = np.ascontiguousarray(np.concatenate(qa)), np.ascontiguousarray(np.concatenate(qb)), np.concatenate(owner)
    # K2 on every edge (env rows addressed through samples_per_env = 1 on the gathered start rows)
    mv = bp.check_motion(torch.from_numpy(qa).to(dev), torch.from_numpy(qb).to(dev), start[torch.from_numpy(owner).to(dev)].contiguous(),
                         samples_per_env=1)
    assert bool(mv.all()), "an edge of a returned path fails motion validation"
    # --- invariances ---
    capped = bp.plan(start, goal, max_workgroups=48, **prm)
    perm = torch.randperm(E, device=dev)
    permd = bp.plan(start[perm].contiguous(), goal[perm].contiguous(), env_ids=perm.to(torch.int64).contiguous(), **prm)
    torch.cuda.synchronize()
    for name, r, idx in (("workgroup cap", capped, None), ("permutation", permd, perm.cpu().numpy())):
        rp, rl, rs, rc = (x.cpu().numpy() for x in r)
        sel = np.arange(E) if idx is None else idx
        assert np.array_equal(rs, st_h[sel]) and np.array_equal(rl, plen_h[sel]) and np.array_equal(rc, chk_h[sel]), name
        assert np.array_equal(_bits(rp), _bits(path_h[sel])), name
    # --- oracle on a random subset, failing queries included ---
    rng = np.random.default_rng(5)
    fails = np.where(st_h == -4)[0]
    sub = np.unique(np.concatenate([rng.choice(E, 120, replace=False), fails[:8]]))

    def one(e):
        return orc.plan(s_h[e], g_h[e], pi.spec.range, 0.005, prm["max_iters"], prm["max_nodes"], seed=prm["seed"], env_id=int(e),
                        max_path=prm["max_path"])
    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(one, sub))
    for e, (st, opath, ochk, _) in zip(sub, res):
        assert st_h[e] == st and plen_h[e] == len(opath) and chk_h[e] == ochk, f"query {e}"
        assert np.array_equal(_bits(path_h[e, :plen_h[e]]), _bits(opath)), f"query {e}: path"


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_pruned_pairs_never_reach_the_threshold(env, oracle_mod):
    """The compile-time proof behind `Scene(prune_pairs=True)` (tools/prove_separated_pairs.py), cross-checked on sampled
    states: the oracle's distance of every pruned pair stays above 0 (> the negative contact threshold), and verdicts /
    depths with and without pruning agree bit for bit; the proof covers joint values inside their ranges."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk(env, oracle_mod)
    _, sc_full, _ = _mk(env, oracle_mod, prune_pairs=False)
    m = pi.model
    never = m.meta.get("never_violating_pairs", [])
    assert len(never) > 0 and sc.npair_pruned > 0 and sc.npair_checked == sc_full.npair_checked - sc.npair_pruned
    idx = {(int(a), int(b)): k for k, (a, b) in enumerate(m.pair_geom)}
    rows_of = [idx[(int(a), int(b))] for a, b in never]
    qa, rows = sample_states(pi, 4000, 17, "uniform")
    lo = np.inf
    for i in range(0, len(qa), 8):
        lo = min(lo, orc.pair_dist(_full(pi, qa[i], rows[0]))[rows_of].min())
    assert lo > 0.0
    tq, tr = torch.tensor(qa, device="cuda"), torch.tensor(rows, device="cuda")
    a = BatchPlanner(sc).is_valid(tq, tr, samples_per_env=len(qa), want_min_dist=True)
    b = BatchPlanner(sc_full).is_valid(tq, tr, samples_per_env=len(qa), want_min_dist=True)
    assert torch.equal(a[0], b[0]) and np.array_equal(_bits(a[1].cpu().numpy()), _bits(b[1].cpu().numpy()))


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
@pytest.mark.parametrize("kernel", ["v1", "v5"])
def test_verdicts_at_the_threshold(env, kernel, oracle_mod):
    """Adversarial states for everything that decides a pair WITHOUT its exact distance (FP32 broad phase, per-pair cull radii,
    enclosing-capsule / separating-axis pre-tests, the deep-overlap shortcut of the verdict-only kernels): segments from a valid
    to an invalid state are bisected (on the oracle's verdict) down to two neighbouring states that straddle the contact
    threshold, and the kernels must agree with the oracle on those and on states a few micrometres to millimetres either side."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk(env, oracle_mod, kernel)
    bp = BatchPlanner(sc)
    qa, row = sample_states(pi, 1200, 23, "uniform")
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa), nthreads=0)
    good, bad = qa[ov.astype(bool)], qa[~ov.astype(bool)]
    n = min(len(good), len(bad), 150)
    assert n >= 40
    states = []
    for a, b in zip(good[:n], bad[:n]):
        lo, hi = 0.0, 1.0
        for _ in range(44):
            mid = 0.5 * (lo + hi)
            ok, _ = orc.is_valid(_full(pi, a + mid * (b - a), row[0]))
            lo, hi = (mid, hi) if ok else (lo, mid)
        for t in (lo, hi):
            for dt in (0.0, 1e-9, -1e-9, 1e-6, -1e-6, 1e-4, -1e-4, 3e-3, -3e-3):
                states.append(a + min(max(t + dt, 0.0), 1.0) * (b - a))
    qs = np.ascontiguousarray(np.array(states))
    ov, omd = orc.is_valid_batch(qs, row, samples_per_env=len(qs), nthreads=0)
    assert 0.2 < ov.mean() < 0.8
    tq, tr = torch.from_numpy(qs).cuda(), torch.from_numpy(row).cuda()
    v, md = bp.is_valid(tq, tr, samples_per_env=len(qs), want_min_dist=True)
    v2 = bp.is_valid(tq, tr, samples_per_env=len(qs))
    torch.cuda.synchronize()
    assert np.array_equal(v2.cpu().numpy(), ov), "verdict-only kernel"
    assert np.array_equal(v.cpu().numpy(), ov), "kernel with depths"
    assert np.array_equal(_bits(md.cpu().numpy()), _bits(omd))
    # the deepest penetration of the states just past the boundary sits at the threshold (that is what was bisected)
    thr = pi.spec.contact_threshold
    assert np.median(np.abs(omd[9::18] - thr)) < 1e-6          # (state 9 of a segment's 18: the first invalid one)


def test_continued_planning_equals_one_full_launch(oracle_mod):
    """`keep_state` / `resume`: a launch with a small budget leaves its unsolved queries' trees and counters behind, a later
    launch continues them (in any order, alone or pooled) -- status, path, and consumed-check count are those of ONE launch
    with the full budget, which in turn equals the oracle (test_plan_matches_oracle).  Two continuations in a row as well."""
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.batch import BatchPlanner, PlanState
    pi, sc, orc = _mk("SawyerPushObstacle-v0", oracle_mod)
    bp = BatchPlanner(sc)
    E = 256
    qa, row = sample_states(pi, 8000, 47, "near")
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
    good = qa[ov == 1]
    starts = np.repeat(row, E, axis=0)
    goals = starts.copy()
    starts[:, pi.ref_joint_pos_indexes] = good[:E]
    goals[:, pi.ref_joint_pos_indexes] = good[E:2 * E]
    goals[:8] = starts[:8]                                      # trivial queries
    goals[8:12, pi.ref_joint_pos_indexes] = qa[ov == 0][:4]     # invalid goals: final whatever the budget
    s, g = torch.from_numpy(starts).cuda(), torch.from_numpy(goals).cuda()
    prm = dict(max_nodes=512, max_path=32, seed=11)             # (a short max_path: some solved queries end as "path too long")
    ids = torch.arange(E, device="cuda", dtype=torch.int64)
    full = [t.cpu().numpy() for t in bp.plan(s, g, max_iters=600, env_ids=ids, **prm)]
    p1 = bp.plan(s, g, max_iters=40, env_ids=ids, keep_state=True, **prm)
    again = torch.nonzero(p1[2] == _lib.PLAN_NO_EXACT).flatten()
    assert 10 < len(again) < E
    perm = again[torch.randperm(len(again), device="cuda")]    # continued in another order ...
    half = len(perm) // 2
    parts = [perm[:half].contiguous(), perm[half:].contiguous()]
    out = [t.clone() for t in p1[:4]]
    # ... the first half in two steps (40 -> 150 -> 600 iterations), the second half pooled from gathered + concatenated states
    a = parts[0]
    q1 = bp.plan(s[a].contiguous(), g[a].contiguous(), max_iters=150, env_ids=a, resume=p1[4].rows(a), keep_state=True, **prm)
    q2 = bp.plan(s[a].contiguous(), g[a].contiguous(), max_iters=600, env_ids=a, resume=q1[4], **prm)
    b = parts[1]
    b1, b2 = b[:len(b) // 2].contiguous(), b[len(b) // 2:].contiguous()
    r2 = bp.plan(s[b].contiguous(), g[b].contiguous(), max_iters=600, env_ids=b, resume=PlanState.cat([p1[4].rows(b1), p1[4].rows(b2)]), **prm)
    torch.cuda.synchronize()
    for k in range(4):
        out[k][a] = q2[k]
        out[k][b] = r2[k]
    out = [t.cpu().numpy() for t in out]
    for k in (1, 2, 3):
        assert np.array_equal(out[k], full[k]), ("path_len", "status", "n_checks")[k - 1]
    for e in range(E):
        assert np.array_equal(out[0][e, :out[1][e]].view(np.uint64), full[0][e, :full[1][e]].view(np.uint64))
    assert (full[2] == 0).sum() > E // 4 and (full[2] == _lib.PLAN_NO_EXACT).sum() > 4 and (full[2] == _lib.PLAN_INVALID_GOAL).sum() == 4


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_states_outside_the_joint_ranges_take_the_full_pair_list(env, oracle_mod):
    """The reference's isValidState accepts ANY state (motion_planners/KinematicPlanner.cpp:253-286), and MuJoCo's soft joint
    limits let a reported qpos sit outside the range.  The compile-time pair pruning is proven inside range + guard band only:
    states up to 0.5 rad (5 cm for slides) beyond every range must come back with the oracle's verdict AND depth -- through the
    guarded entry points (`Scene.is_valid_state`, `BatchPlanner.is_valid(guard=True)`), which route them to the full list."""
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env)
    m = pi.model
    sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
    assert sc.npair_pruned > 0
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    bp = BatchPlanner(sc)
    rng = np.random.default_rng(5)
    N = 600
    lim = [j for j in range(len(m.jnt_names)) if m.jnt_limited[j] and int(m.jnt_type[j]) != 0]
    rows = np.tile(np.array(m.qpos0, dtype=np.float64), (N, 1))
    for i in range(N):
        for j in lim:
            lo, hi = m.jnt_range[j]
            ext = 0.5 if int(m.jnt_type[j]) == 3 else 0.05
            if i % 3 == 0:            # one joint beyond its range, the others inside
                rows[i, int(m.jnt_qposadr[j])] = rng.uniform(lo, hi)
            else:                     # every joint anywhere in the extended box
                rows[i, int(m.jnt_qposadr[j])] = rng.uniform(lo - ext, hi + ext)
        if i % 3 == 0:
            j = lim[rng.integers(len(lim))]
            lo, hi = m.jnt_range[j]
            ext = 0.5 if int(m.jnt_type[j]) == 3 else 0.05
            rows[i, int(m.jnt_qposadr[j])] = (lo - rng.uniform(0, ext)) if rng.random() < 0.5 else (hi + rng.uniform(0, ext))
    outside = sc.outside_guard(rows)
    assert outside.sum() > N // 2 and (~outside).sum() > 5
    qa = torch.tensor(rows[:, sc.active_idx], device="cuda").contiguous()
    qe = torch.tensor(rows, device="cuda").contiguous()
    v, md = bp.is_valid(qa, qe, samples_per_env=1, want_min_dist=True, guard=True)
    v, md = v.cpu().numpy().astype(bool), md.cpu().numpy()
    ov = np.zeros(N, dtype=bool); omd = np.zeros(N)
    for i in range(N):
        ov[i], omd[i] = orc.is_valid(rows[i])
    assert np.array_equal(v, ov) and np.array_equal(md.view(np.uint64), omd.view(np.uint64))
    for i in range(0, N, 37):        # the single-state entry point (the reference's isValidState) guards by itself
        a, d = sc.is_valid_state(rows[i], want_min_dist=True)
        assert a == ov[i] and np.float64(d).view(np.uint64) == omd[i].view(np.uint64)
    assert (~ov).sum() > 20 and ov.sum() > 20
    sc.close()


def test_nearest_neighbour_mirror_sizes_do_not_change_plans(oracle_mod, monkeypatch):
    """K3's one-wave-per-SIMD build sweeps an FP32 mirror of the trees in LDS and the part a tree outgrows in FP64 from HBM; the answer is
    the exact sweep's whatever the mirror holds (unique-candidate test, FP64 fallback): no mirror, 64 slots, all the LDS -- and the
    two-waves-per-SIMD build, which has none -- give the same trees, paths and consumed checks on queries whose trees grow to hundreds of nodes"""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk("SawyerPushObstacle-v0", oracle_mod)
    bp = BatchPlanner(sc)
    E = 48
    qa, row = sample_states(pi, 8000, 77, "uniform")
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
    good = qa[ov == 1]
    starts = np.repeat(row, E, axis=0)
    goals = starts.copy()
    starts[:, pi.ref_joint_pos_indexes] = good[:E]
    goals[:, pi.ref_joint_pos_indexes] = good[E:2 * E]          # far-apart uniform samples: long searches, many run out the budget
    s, g = torch.from_numpy(starts).cuda(), torch.from_numpy(goals).cuda()
    prm = dict(max_iters=1200, max_nodes=2048, max_path=256, seed=11)
    out = {}
    for name, cap, kw in (("all", None, {}), ("none", "0", {}), ("64", "64", {}), ("w2", None, {"max_workgroups": -1})):
        if cap is None:
            monkeypatch.delenv("MOPA_PLAN_NN_CAP", raising=False)
        else:
            monkeypatch.setenv("MOPA_PLAN_NN_CAP", cap)
        out[name] = [t.cpu().numpy() for t in bp.plan(s, g, **prm, **kw)]
    torch.cuda.synchronize()
    st = out["all"][2]
    assert (st != 0).sum() >= 4 and out["all"][3].max() > 1500          # budget-exhausting queries with big trees are in the set
    for name in ("none", "64", "w2"):
        assert np.array_equal(out[name][2], st) and np.array_equal(out[name][1], out["all"][1]) and np.array_equal(out[name][3], out["all"][3]), name
        for e in range(E):
            n = int(out["all"][1][e])
            assert np.array_equal(_bits(out[name][0][e, :n]), _bits(out["all"][0][e, :n])), (name, e)
    # ... and the oracle agrees on a few of them
    for e in range(0, E, 12):
        ost, opath, ochk, _ = orc.plan(starts[e], goals[e], pi.spec.range, 0.005, max_iters=1200, max_nodes=2048, seed=11, env_id=e, max_path=256)
        assert ost == st[e] and ochk == out["all"][3][e]


@pytest.mark.gpu
@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_planner_builds_agree(env, oracle_mod, monkeypatch):
    """K3's three builds -- four queries per workgroup at one / two waves per SIMD (k3w1 / k3w2) and ONE query per workgroup of four
    waves (k3wg: waves 1-3 work out and evaluate what future iterations will ask; mopa_planner_k3.inc) -- give the same status, path
    bits and consumed-check count on every scene, on queries that are trivial, solvable, budget-exhausting and invalid; so does k3wg
    with a tiny tree mirror (its exact FP64 fall-backs) and with fewer workgroups than queries (the workgroup's query hand-over).
    A few queries are checked against the oracle as well."""
    import torch
    from mopa_rl_amd.batch import BatchPlanner
    pi, sc, orc = _mk(env, oracle_mod)
    bp = BatchPlanner(sc)
    E = 40
    qa, row = sample_states(pi, 6000, 91, "uniform")
    ov, _ = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
    good = qa[ov == 1]
    assert len(good) >= 2 * E
    starts = np.repeat(row, E, axis=0)
    goals = starts.copy()
    starts[:, pi.ref_joint_pos_indexes] = good[:E]
    goals[:, pi.ref_joint_pos_indexes] = good[E:2 * E]
    goals[:3] = starts[:3]
    if (ov == 0).sum() >= 2:
        goals[3:5, pi.ref_joint_pos_indexes] = qa[ov == 0][:2]
    s, g = torch.from_numpy(starts).cuda(), torch.from_numpy(goals).cuda()
    prm = dict(max_iters=700, max_nodes=1024, max_path=256, seed=23)
    out = {}
    for name, build, cap, kw in (("w1", "w1", None, {}), ("w2", "w2", None, {"max_workgroups": -1}), ("wg", "wg", None, {}),
                                 ("wg_small_mirror", "wg", "64", {}), ("wg_3_workgroups", "wg", None, {"max_workgroups": 3, "exclusive": True})):
        monkeypatch.setenv("MOPA_PLAN_BUILD", build)
        if cap is None:
            monkeypatch.delenv("MOPA_PLAN_NN_CAP", raising=False)
        else:
            monkeypatch.setenv("MOPA_PLAN_NN_CAP", cap)
        out[name] = [t.cpu().numpy() for t in bp.plan(s, g, **prm, **kw)]
    torch.cuda.synchronize()
    ref = out["w1"]
    assert (ref[2] == 0).sum() >= 3
    for name, o in out.items():
        assert np.array_equal(o[2], ref[2]) and np.array_equal(o[1], ref[1]) and np.array_equal(o[3], ref[3]), name
        for e in range(E):
            n = int(ref[1][e])
            assert np.array_equal(_bits(o[0][e, :n]), _bits(ref[0][e, :n])), (name, e)
    for e in (0, 3, 7, 19, 33):
        ost, opath, ochk, _ = orc.plan(starts[e], goals[e], pi.spec.range, 0.005, max_iters=700, max_nodes=1024, seed=23, env_id=e, max_path=256)
        assert ost == ref[2][e] and ochk == ref[3][e], e
