"""Pin the oracle's damped-least-squares IK (oracle/mopa_oracle.c:orc_ik_solve, the checker of HIP kernel K5) against an
independent numpy restatement of reference env/inverse_kinematics.py:18-135,274-281 that uses a finite-difference
Jacobian of an independent FK and np.linalg.solve, as the reference does."""
import numpy as np
import pytest

from mopa_rl_amd.scene import default_qpos, planner_inputs
from test_oracle_fk import independent_fk

ENVS = ["SawyerPushObstacle-v0", "SawyerAssemblyObstacle-v0"]


def _site(m, name="grip_site"):
    i = m.site_name2id(name)
    return int(m.site_body[i]), np.asarray(m.site_pos[i], dtype=np.float64)


def _site_pos(m, q, sb, so):
    P, Rw = independent_fk(m, q)
    return P[sb] + Rw[sb].apply(so)


def numpy_ik(m, q, target, adrs, sb, so, max_steps=100, tol=1e-2, max_update_norm=2.0, progress_thresh=20.0, reg=3e-2):
    q = q.copy()
    success, err_norm, steps = False, 0.0, 0
    for steps in range(max_steps):
        p = _site_pos(m, q, sb, so)
        err = target - p
        err_norm = np.linalg.norm(err)
        if err_norm < tol:
            success = True
            break
        J = np.zeros((3, len(adrs)))
        for k, a in enumerate(adrs):
            h = 1e-6
            qp, qm = q.copy(), q.copy()
            qp[a] += h
            qm[a] -= h
            J[:, k] = (_site_pos(m, qp, sb, so) - _site_pos(m, qm, sb, so)) / (2 * h)
        dq = np.linalg.solve(J.T @ J + np.eye(len(adrs)) * reg, J.T @ err)       # nullspace_method, regularised branch
        un = np.linalg.norm(dq)
        if err_norm / un > progress_thresh:
            break
        if un > max_update_norm:
            dq *= max_update_norm / un
        q[adrs] += dq
    return q, err_norm, steps, success


@pytest.mark.parametrize("env", ENVS)
def test_ik_matches_independent_numpy(env, oracle_mod):
    pi = planner_inputs(env)
    m = pi.model
    orc = oracle_mod.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    sb, so = _site(m)
    jids = [m.joint_name2id(j) for j in pi.spec.robot_joints]
    adrs = [int(m.jnt_qposadr[j]) for j in jids]
    rng = np.random.default_rng(2)
    q0 = default_qpos(env, m)
    n_ok = n_fail = 0
    for trial in range(9):
        q = q0.copy()
        q[adrs] += rng.normal(0, 0.2, len(adrs))
        reach = 0.06 if trial < 6 else 2.5         # the last targets are out of reach -> no success
        target = _site_pos(m, q, sb, so) + rng.normal(0, 1, 3) * reach
        for tol in (1e-2, 1e-6):
            qo, eo, so_, suo = orc.ik_solve(q, target, jids, sb, so, tol=tol)
            qn, en, sn, sun = numpy_ik(m, q, target, adrs, sb, so, tol=tol)
            assert suo == sun and so_ == sn, (trial, tol, so_, sn)
            np.testing.assert_allclose(qo, qn, rtol=0, atol=2e-7)
            assert abs(eo - en) < 2e-7
            if suo:
                assert np.linalg.norm(_site_pos(m, qo, sb, so) - target) < tol
            n_ok += suo
            n_fail += not suo
    assert n_ok >= 6 and n_fail >= 2


def test_ik_leaves_other_joints_alone_and_handles_zero_steps(oracle_mod):
    env = "SawyerPushObstacle-v0"
    pi = planner_inputs(env)
    m = pi.model
    orc = oracle_mod.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    sb, so = _site(m)
    jids = [m.joint_name2id(j) for j in pi.spec.robot_joints]
    q = default_qpos(env, m)
    p = _site_pos(m, q, sb, so)
    q1, e1, s1, ok1 = orc.ik_solve(q, p + [0.0, 0.0, 0.001], jids, sb, so)          # already within tol
    assert ok1 and s1 == 0 and np.array_equal(q1, q)
    q2, e2, s2, ok2 = orc.ik_solve(q, p + [0.05, -0.04, 0.03], jids[:4], sb, so)     # only 4 movable joints
    assert s2 > 0 and not np.array_equal(q2, q) and np.array_equal(np.delete(q2, [int(m.jnt_qposadr[j]) for j in jids[:4]]),
                                  np.delete(q, [int(m.jnt_qposadr[j]) for j in jids[:4]]))
    q3, e3, s3, ok3 = orc.ik_solve(q, p + [0.05, -0.04, 0.03], jids, sb, so, max_steps=1, tol=1e-9)
    assert not ok3 and s3 == 0


@pytest.mark.parametrize("env,tag", [("SawyerAssemblyObstacle-v0", "assembly"), ("SawyerPushObstacle-v0", "push")])
def test_oracle_ik_equals_reference_qpos_from_site_pose(env, tag, oracle_mod):
    """orc_ik_solve against vectors produced by the REFERENCE'S OWN `qpos_from_site_pose` / `nullspace_method`
    (env/inverse_kinematics.py:18-135,274-281; tests/golden/ref_py_ik.npz, tools/gen_ref_py_golden.py), position-only and
    position + orientation targets: same joint vectors to round-off (np.linalg.solve vs Cholesky, np.arctan2 vs the shared
    atan2), same iteration counts and success flags."""
    import os
    from mopa_rl_amd.scene import ENV_SPECS, load_scene
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_py_ik.npz"))
    m = load_scene(ENV_SPECS[env].scene)
    orc = oracle_mod.OracleScene(m, [], [], 0.0)
    si = m.site_name2id("grip_site")
    jids = [m.joint_name2id(j) for j in ENV_SPECS[env].robot_joints]
    K = len(G[f"{tag}_qpos"])
    for k in range(K):
        uq = bool(G[f"{tag}_use_quat"][k])
        q, en, st, su = orc.ik_solve(G[f"{tag}_qpos"][k], G[f"{tag}_target_pos"][k], jids, int(m.site_body[si]), m.site_pos[si],
                                     max_steps=100, tol=1e-2, target_quat=G[f"{tag}_target_quat"][k] if uq else None,
                                     site_quat=m.site_quat[si])
        np.testing.assert_allclose(q, G[f"{tag}_qpos_out"][k], rtol=0, atol=1e-10, err_msg=str(k))
        assert abs(en - G[f"{tag}_err_norm"][k]) < 1e-10 and st == G[f"{tag}_steps"][k] and su == bool(G[f"{tag}_success"][k]), k


def test_nullspace_method_linear_solve(oracle_mod):
    """the regularised normal-equation solve on its own: (J^T J + 0.03 I) x = J^T d from the reference's `nullspace_method`
    (3 x 7 and 6 x 7 Jacobians) vs a Cholesky solve in numpy with the kernel's operation order"""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_py_ik.npz"))
    for J, d, rows, want in zip(G["ns_J"], G["ns_delta"], G["ns_rows"], G["ns_out"]):
        J, d = J[:rows], d[:rows]
        H = J.T @ J + 0.03 * np.eye(7)
        L = np.linalg.cholesky(H)
        x = np.linalg.solve(L.T, np.linalg.solve(L, J.T @ d))
        np.testing.assert_allclose(x, want, rtol=1e-11, atol=1e-13)


def test_atan2_matches_libm(oracle_mod):
    rng = np.random.default_rng(0)
    xs, ys = rng.normal(size=4000), rng.normal(size=4000)
    got = np.array([oracle_mod.atan2(y, x) for x, y in zip(xs, ys)])
    assert np.abs(got - np.arctan2(ys, xs)).max() < 1e-15
    for y, x in [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0), (1e-300, 1), (1, 1e-300), (0.198912367379658, 1.0), (0.668178637919299, 1.0)]:
        assert abs(oracle_mod.atan2(y, x) - np.arctan2(y, x)) < 1e-15
