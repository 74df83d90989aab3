"""CPU tests of the contact + constraint stage of the dynamics oracle (SURVEY.md 8 f4b stage C; oracle/mopa_oracle_contact.inc):
one contact step against an independent QP solve (tests/dyn_ref.py), the per-pair solver parameters against MuJoCo's mixing
rules on the XML's numbers, and the behaviours the stage exists for -- objects rest, the arm stops at obstacles, the gripper
pushes the cube, the can is pinched and lifted by friction.  PARITY WITH MuJoCo IS UNPINNED (not available here)."""
import numpy as np
import pytest

from mopa_rl_amd.dynamics import contact_facts, dyn_facts
from mopa_rl_amd.kinematic_env import OBJECT_BODY, ENV_KIND, env_facts
from mopa_rl_amd.mjcf import _quat_to_mat
from mopa_rl_amd.scene import ENV_SPECS, load_scene, planner_inputs
from oracle import oracle as O

import dyn_ref

ENVS = ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0", "SawyerAssemblyObstacle-v0"]


def _setup(env, **kw):
    m = load_scene(ENV_SPECS[env].scene)
    f = env_facts(env, m)
    d = dyn_facts(m, f)
    q0 = np.array(m.qpos0, dtype=np.float64)
    q0[f.arm_qpos_idx] = ENV_SPECS[env].init_qpos
    ct = contact_facts(m, d, OBJECT_BODY[ENV_KIND[env]], qpos_ref=q0, **kw)
    return m, f, d, ct, O.OracleDyn(d, ct=ct), q0


def _scene(env, m):
    pi = planner_inputs(env, m)
    return O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)


def _eef(orc, f, q):
    xp, xq = orc.fk_bodies(q)
    b = int(f.frame_body[0])
    return xp[b] + _quat_to_mat(xq[b]) @ f.frame_off[0]


def _ik(env, m, f, orc, q, tg, quat=None):
    jid = [m.joint_name2id(j) for j in ENV_SPECS[env].robot_joints]
    qt, err, steps, ok = orc.ik_solve(q, tg, jid, int(f.frame_body[0]), f.frame_off[0], max_steps=400, tol=1e-4, target_quat=quat)
    assert ok
    return qt


def test_pair_parameters_follow_mujocos_mixing_rules():
    """robot geoms: solref 0.008 1, solimp 0.95 0.95 0.01, margin 0.001 (sawyer_dependencies.xml:36); table / cube: MuJoCo's
    defaults 0.02 1 / 0.9 0.95 0.001, margin 0; cube friction 0.95, everything else 1 (sawyer_push_obstacle.xml)"""
    m, f, d, ct, od, q0 = _setup("SawyerPushObstacle-v0")
    names = [m.body_names[int(m.geom_body[int(g)])] for g in ct.sh_geom]
    seen = set()
    for p in range(len(ct.pr_f)):
        a, b = names[ct.pr_f[p]], names[ct.pr_s[p]]
        mu, margin, K, B, d0, dmax, width = ct.pr_par[p][:7]
        robot = [n.startswith(("right_", "head", "screen", "clawGripper", "rightclaw", "leftclaw", "controller", "pedestal")) for n in (a, b)]
        if "cube" in (a, b) and "table" in (a, b) or ("cube" in (a, b) and "bin1" in (a, b)):
            assert (mu, margin, d0, dmax, width) == (1.0, 0.0, 0.9, 0.95, 0.001)
            assert np.isclose(K, 1 / (0.95 ** 2 * 0.02 ** 2)) and np.isclose(B, 2 / (0.95 * 0.02))
            seen.add("obj-static")
        elif all(robot):
            assert (mu, margin, width) == (1.0, 0.001, 0.01) and d0 == dmax == 0.95
            assert np.isclose(K, 1 / (0.95 ** 2 * 0.008 ** 2)) and np.isclose(B, 2 / (0.95 * 0.008))
            seen.add("robot-robot")
        elif any(robot) and "cube" in (a, b):
            assert (mu, margin) == (1.0, 0.001) and np.isclose(d0, 0.925) and np.isclose(width, 0.0055)
            assert np.isclose(K, 1 / (0.95 ** 2 * 0.014 ** 2))          # timeconst (0.008 + 0.02) / 2
            seen.add("robot-obj")
    assert seen == {"obj-static", "robot-robot", "robot-obj"}
    # the can's solref 0.001 mixes with the table's 0.02 to 0.0105; with a finger's 0.008 to 0.0045 (>= 2 timesteps = 0.004)
    m, f, d, ct, od, q0 = _setup("SawyerLiftObstacle-v0")
    tcs = sorted({round(float(np.sqrt(1.0 / (p[2] * p[5] ** 2))), 6) for p in ct.pr_par})
    assert 0.0105 in tcs and 0.0045 in tcs and min(tcs) >= 0.004


@pytest.mark.parametrize("solver", ["newton", "pgs"])
@pytest.mark.parametrize("env", ENVS)
def test_objects_come_to_rest(env, solver):
    m, f, d, ct, od, q0 = _setup(env, solver=solver)
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    ctrl = q[d.qadr].copy()
    for _ in range(6):
        q, v, lag = od.step(q, v, lag, ctrl, n=75)
    oq = ct.obj_qadr
    z1 = q[oq + 2]
    q, v, lag = od.step(q, v, lag, ctrl, n=75)
    assert np.all(np.isfinite(q)) and abs(q[oq + 2] - z1) < 1e-5 and np.abs(v[d.nd:d.nd + 3]).max() < 1e-3 and np.abs(v[d.nd + 3:]).max() < 2e-2
    assert np.abs(q[oq:oq + 2] - q0[oq:oq + 2]).max() < 2e-3          # it did not wander
    assert od.stats.max_contacts <= ct.maxcon and od.stats.contacts > 0
    # warm-started, a resting contact set needs a handful of sweeps (Newton: one or two iterations), not the cap of 50
    assert od.stats.sweeps / max(od.stats.substeps, 1) < (3 if solver == "newton" else 15)


@pytest.mark.parametrize("solver", ["newton", "newton-pyramidal", "pgs"])
def test_one_contact_step_equals_an_independent_qp_solve(solver):
    """states with arm-table, arm-cube and cube-floor contacts at once: the oracle's sub-step (PGS run to convergence) against
    tests/dyn_ref.contact_step_reference (independent FK / Jacobians / M / bias, exact active-set solve of the dual)"""
    env = "SawyerPushObstacle-v0"
    cone = "elliptic" if solver == "newton" else "pyramidal"      # (the default pairing: the XML's elliptic cones with the Newton solver)
    m, f, d, ct, od, q0 = _setup(env, iterations=3000 if solver == "pgs" else 50, tolerance=0.0, warmstart=False, noslip_iterations=0,
                                 solver=solver.split("-")[0], cone=cone, limit_rows=False)     # (the contact solve alone, run to convergence)
    orc = _scene(env, m)
    oq = ct.obj_qadr
    q0 = q0.copy(); q0[oq:oq + 3] = [0.80, 0.30, 0.853]
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    q, v, lag = od.step(q, v, lag, q0[d.qadr].copy(), n=300)
    cz = q[oq + 2]
    checked = 0
    for i, tg in enumerate([np.array([0.68, 0.30, cz + 0.15]), np.array([0.68, 0.30, cz + 0.03])] + [np.array([0.68 + 0.02 * k, 0.30, cz + 0.03]) for k in range(1, 6)]):
        qt = _ik(env, m, f, orc, q, tg)
        ctrl = qt[d.qadr].copy(); ctrl[7:] = q[d.qadr[7:]]
        for _ in range(6 if i < 2 else 2):
            q, v, lag = od.step(q, v, lag, ctrl, n=74)
            con = od.contacts(q)
            arm = [r for r in con if ct.sh_body[int(r[7])] in range(d.nd) or ct.sh_body[int(r[8])] in range(d.nd)]
            if i >= 2 and arm:
                vr, fr = dyn_ref.contact_step_reference(m, d, ct, con, q, v, lag, ctrl, cone=cone)
                q1, v1, _ = od.step(q, v, lag, ctrl, n=1)
                scale = max(np.abs(v1 - v).max(), 1e-3)
                free = np.ones(od.nv, dtype=bool)          # (a dof the sub-step ends ON its joint stop is not the reference's business)
                free[:d.nd] = ~((d.limited == 1) & ((q1[d.qadr] <= d.lo) | (q1[d.qadr] >= d.hi)))
                assert np.abs(v1 - vr)[free].max() < 2e-4 * scale + 1e-7, (i, np.abs(v1 - vr).max(), scale)
                assert fr.max() > 0
                checked += 1
            q, v, lag = od.step(q, v, lag, ctrl, n=1)
    assert checked >= 6


@pytest.mark.parametrize("env", ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0"])
def test_spinning_sliding_object_step_equals_the_independent_qp(env):
    """round 6, condim 4: the cube / the can on the floor given a spin about the vertical, a tumble and a slide -- the torsional rows are
    loaded (relative angular velocity about the contact normals), the cones leave their sticking zone; one sub-step against
    tests/dyn_ref's projection form of the ellipsoidal cone (rows, friction-row regularisers R_j mu_j^2 = const, scaling derived there)."""
    m, f, d, ct, od, q0 = _setup(env, iterations=50, tolerance=0.0, warmstart=False, noslip_iterations=0, limit_rows=False)
    oq = ct.obj_qadr
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    ctrl = q[d.qadr].copy()
    q, v, lag = od.step(q, v, lag, ctrl, n=150)
    checked = 0
    for vel in ([0.0, 0.0, 0.0, 0.0, 0.0, 6.0], [0.25, -0.1, 0.0, 0.4, -0.3, 3.0], [1.5, 0.0, 0.0, 0.0, 0.0, 40.0], [0.0, 0.0, -0.05, 2.0, 1.0, -15.0]):
        v2 = v.copy(); v2[d.nd:] = vel
        con = od.contacts(q)
        pair_of = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(ct.pr_f, ct.pr_s))}
        assert len(con) >= 3 and all(int(ct.pr_par[pair_of[(int(r[7]), int(r[8]))]][8]) == 4 for r in con)
        vr, fr = dyn_ref.contact_step_reference(m, d, ct, con, q, v2, lag, ctrl, cone="elliptic")
        q1, v1, _ = od.step(q, v2, lag, ctrl, n=1)
        scale = max(np.abs(v1 - v2).max(), 1e-3)
        assert np.abs(v1 - vr).max() < 1e-5 * scale + 1e-9, (vel, np.abs(v1 - vr).max(), scale)
        fr = fr.reshape(-1, 4)
        checked += int(np.abs(fr[:, 3]).max() > 1e-6)          # the torsional row carried force
    assert checked >= 3


@pytest.mark.parametrize("env", ENVS)
def test_resting_objects_do_not_spin_with_elliptic_cones(env):
    """cube, can and furniture at rest on their supports under the default solver form (Newton, elliptic cones, noslip): no residual motion
    at all -- with pyramidal cones the can kept turning at 2e-4 rad/s on its rim contacts (round 4)"""
    m, f, d, ct, od, q0 = _setup(env)
    assert ct.solver == 2
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    ctrl = q[d.qadr].copy()
    for _ in range(10):
        q, v, lag = od.step(q, v, lag, ctrl, n=75)
    assert np.abs(v[d.nd:d.nd + 3]).max() < 1e-6 and np.abs(v[d.nd + 3:]).max() < 1e-5


def test_cube_on_a_tilt_at_0_9_mu_does_not_creep():
    """gravity tilted by atan(0.9) against friction 1 (the cube also stands: it tips at atan(1)): with elliptic cones + noslip the cube
    stays put -- 20 micrometres in two seconds at most --, with pyramidal cones (inscribed in the cone: weaker off the axes' sum) it
    slides; at atan(1.1) it must slide either way"""
    import dataclasses
    env = "SawyerPushObstacle-v0"
    g = 9.81
    drift = {}
    for cone, frac in (("elliptic", 0.9), ("pyramidal", 0.9), ("elliptic", 1.1)):
        m, f, d, ct, od, q0 = _setup(env, cone=cone)
        oq = ct.obj_qadr
        q, v = q0.copy(), np.zeros(od.nv)
        lag = od.forward(q, v[:d.nd], want_M=False)[0]
        ctrl = q[d.qadr].copy()
        q, v, lag = od.step(q, v, lag, ctrl, n=300)
        th = np.arctan(frac)
        tilted = O.OracleDyn(dataclasses.replace(d, gravity=np.array([g * np.sin(th), 0.0, -g * np.cos(th)])), ct=ct)
        lag = tilted.forward(q, v[:d.nd], want_M=False)[0]
        q, v, lag = tilted.step(q, v, lag, ctrl, n=250)          # the transient of the tilt
        p0 = q[oq:oq + 3].copy()
        q, v, lag = tilted.step(q, v, lag, ctrl, n=250 if frac > 1 else 1000)
        drift[(cone, frac)] = q[oq:oq + 3] - p0
    assert np.abs(drift[("elliptic", 0.9)]).max() < 2e-5
    assert drift[("pyramidal", 0.9)][0] > 5e-3
    assert drift[("elliptic", 1.1)][0] > 5e-2


def test_noslip_pass_stops_the_creep_of_a_loaded_friction_contact():
    """Gravity tilted by atan(0.5) against a friction coefficient of 1: the cube must stick.  The soft friction rows alone let
    it creep (4 mm/s -- what MuJoCo's regularised friction does too); the noslip pass (`noslip_iterations="5"`,
    sawyer_dependencies.xml:11: friction dimensions re-solved without the regulariser, normal forces kept) holds it to
    a micrometre per second, and does not cost the main solve its warm start."""
    import dataclasses
    env = "SawyerPushObstacle-v0"
    g, th = 9.81, np.arctan(0.5)
    out = {}
    for ns in (0, 5):
        m, f, d, ct, od, q0 = _setup(env, noslip_iterations=ns)
        assert ct.noslip_iterations == ns
        oq = ct.obj_qadr
        q, v = q0.copy(), np.zeros(od.nv)
        lag = od.forward(q, v[:d.nd], want_M=False)[0]
        ctrl = q[d.qadr].copy()
        q, v, lag = od.step(q, v, lag, ctrl, n=300)                      # settle under the scene's own gravity
        tilted = O.OracleDyn(dataclasses.replace(d, gravity=np.array([g * np.sin(th), 0.0, -g * np.cos(th)])), ct=ct)
        lag = tilted.forward(q, v[:d.nd], want_M=False)[0]
        p0 = q[oq:oq + 3].copy()
        q, v, lag = tilted.step(q, v, lag, ctrl, n=250)
        out[ns] = (q[oq:oq + 3] - p0, v[d.nd:d.nd + 3].copy(), tilted.stats.sweeps / tilted.stats.substeps)
    assert 2e-3 < out[0][1][0] < 6e-3 and out[0][0][0] > 1e-3           # without: creeping downhill
    assert abs(out[5][1][0]) < 2e-5 and abs(out[5][0][0]) < 3e-4        # with: held (the shift is the transient of the tilt)
    assert out[5][2] < out[0][2] + 2.0                                   # the main solve still converges in a few sweeps


def test_joint_limit_is_a_soft_constraint_row_with_the_predicted_rest_violation():
    """A wrist joint servoed 0.5 rad beyond its range: with the limits in the solver (MuJoCo: a constraint row J = -1 on the dof once
    it is beyond the range, solreflimit 0.02 1, solimplimit 0.9 0.95 0.001) it runs into the limit at 5 rad/s, overshoots by ~20 mrad
    and settles a few mrad beyond it where the row's force balances the servo:  f = D aref = K imp |dist| M_ll imp / (1 - imp)
    (regulariser R = (1 - imp) / imp / M_ll).  Without them (stage A's inelastic stop) it sits exactly on the limit."""
    env = "SawyerPushObstacle-v0"
    j = 5
    out = {}
    for lr in (True, False):
        m, f, d, ct, od, q0 = _setup(env, limit_rows=lr)
        assert ct.limit_rows == int(lr)
        q, v = q0.copy(), np.zeros(od.nv)
        lag = od.forward(q, v[:d.nd], want_M=False)[0]
        ctrl = q[d.qadr].copy()
        ctrl[j] = d.hi[j] + 0.5
        peak = 0.0
        for _ in range(10):
            for _ in range(5):
                q, v, lag = od.step(q, v, lag, ctrl, n=15)
                peak = max(peak, q[d.qadr[j]] - d.hi[j])
        out[lr] = (q[d.qadr[j]] - d.hi[j], v[j], peak, q, v, lag)
    assert out[False][0] == 0.0 and out[False][1] == 0.0
    dist, vel, peak, q, v, lag = out[True]
    assert 1e-3 < dist < 1e-2 and abs(vel) < 1e-3 and dist < peak < 0.05
    m, f, d, ct, od, q0 = _setup(env, limit_rows=True)
    bias, M = od.forward(q, v[:d.nd], want_M=True)
    Mll = np.asarray(M).reshape(-1)[j * (j + 1) // 2 + j] if np.asarray(M).ndim == 1 else np.asarray(M)[j, j]
    K, imp = ct.lim_par[2], ct.lim_par[5]                  # |dist| > width: the impedance has reached dmax
    f_row = K * imp * dist * Mll * imp / (1.0 - imp)
    servo = np.clip(d.kp[j] * (ctrl[j] - q[d.qadr[j]]), d.force_lo[j], d.force_hi[j])
    assert abs(f_row - servo) < 0.05 * servo, (f_row, servo)


def test_contact_step_with_a_joint_beyond_its_limit_equals_the_independent_qp():
    """the same independent solve (tests/dyn_ref.py) with a LIMIT row next to the contact rows: the arm resting on the table with a
    wrist joint 20 mrad beyond its range and moving further out -- the Newton solver's sub-step against the exact active-set solve of the
    dual with the limit row J = -e_l, regulariser (1 - imp) / imp / M_ll"""
    env = "SawyerPushObstacle-v0"
    m, f, d, ct, od, q0 = _setup(env, iterations=50, tolerance=0.0, warmstart=False, noslip_iterations=0, limit_rows=True)
    orc = _scene(env, m)
    oq = ct.obj_qadr
    q0 = q0.copy(); q0[oq:oq + 3] = [0.80, 0.30, 0.853]
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    q, v, lag = od.step(q, v, lag, q0[d.qadr].copy(), n=300)
    cz = q[oq + 2]
    qt = _ik(env, m, f, orc, q, np.array([0.70, 0.30, cz + 0.03]))
    ctrl = qt[d.qadr].copy(); ctrl[7:] = q[d.qadr[7:]]
    for _ in range(8):
        q, v, lag = od.step(q, v, lag, ctrl, n=75)
    checked = 0
    for j, sgn in ((6, 1.0), (5, -1.0), (6, -1.0)):
        qq, vv = q.copy(), v.copy()
        qq[d.qadr[j]] = (d.hi[j] + 0.02) if sgn > 0 else (d.lo[j] - 0.02)
        vv[j] = 0.3 * sgn
        lg = od.forward(qq, vv[:d.nd], want_M=False)[0]
        con = od.contacts(qq)
        vr, fr = dyn_ref.contact_step_reference(m, d, ct, con, qq, vv, lg, ctrl, limit_rows=True, cone="elliptic")
        q1, v1, _ = od.step(qq, vv, lg, ctrl, n=1)
        scale = max(np.abs(v1 - vv).max(), 1e-3)
        assert np.abs(v1 - vr).max() < 2e-4 * scale + 1e-7, (j, sgn, np.abs(v1 - vr).max(), scale)
        assert v1[j] * sgn < vv[j] * sgn                     # the limit row decelerates the joint
        checked += 1
    assert checked == 3


def test_arm_stops_at_the_bin_roof():
    """Push: the hand is servoed towards a point below the bin's roof plate (z = 1.22 .. 1.23): with the contact stage it
    comes to rest ON the plate (at rest the deepest pair stays above the planner's -2 mm: the state is valid), the servo
    error stays; without it (stage A) the same command drives the arm through the plate."""
    env = "SawyerPushObstacle-v0"
    m, f, d, ct, od, q0 = _setup(env)
    orc = _scene(env, m)
    free = O.OracleDyn(d)
    tg = np.array([0.80, 0.0, 1.00])
    qt = _ik(env, m, f, orc, q0, tg)
    ctrl = qt[d.qadr].copy(); ctrl[7:] = q0[d.qadr[7:]]
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    qf, vf, lagf = q0.copy(), np.zeros(d.nd), lag.copy()
    through = False
    for k in range(14):
        q, v, lag = od.step(q, v, lag, ctrl, n=75)
        qf, vf, lagf = free.step(qf, vf, lagf, ctrl, n=75)
        ok, depth = orc.is_valid(q)
        # the impact itself (force-saturated servos drive the hand onto the plate) may dent the soft contact for an instant;
        # from then on the state is one the planner calls valid
        assert depth > -0.005 and (ok or k == 0), "the contact stage let the arm sink past the validity threshold"
        through = through or not orc.is_valid(qf)[0]
    assert _eef(orc, f, qf)[2] < 1.05 and through          # stage A: went through the plate, now hangs below it
    e = _eef(orc, f, q)
    assert e[2] > 1.2 and np.abs(q[d.qadr[:7]] - ctrl[:7]).max() > 0.1       # stage C: resting on it, servo error stays
    assert np.abs(v[:7]).max() < 0.05                                          # ... at rest
    con = od.contacts(q)
    assert len(con) > 0 and con[:, 0].min() > -0.002


@pytest.mark.parametrize("condim", ["xml", "3"])
def test_gripper_pushes_the_cube(condim):
    """the cube set on the open table (outside the bin tunnel); the hand comes down behind it (its claw rests on the table) and moves along
    +x.  With elliptic cones friction is fully effective (mu = 1 under the cube, 1 at the claw, which meets the cube's face 4.8 cm up).
    condim "3" (sliding friction only): the cube does not slide out from under the push, it TIPS over its front edge onto its next face
    -- 0.4 N tips it, 0.64 N would slide it -- and is then pushed ahead.  With the XML's condim (round 6) the claw's contact is condim 6 and
    carries the cube's ROLLING friction 0.1 (sawyer_push_obstacle.xml:46 `friction="0.95 0.3 0.1"`; [3P] per pair the larger coefficient): a
    torque of up to 0.1 f_n resists the relative rotation of cube and claw -- more than the 0.02 N m the tipping needs -- so the cube stays
    on its face and slides ahead of the hand.  Either way it stays on the table and stops when the hand stops."""
    env = "SawyerPushObstacle-v0"
    m, f, d, ct, od, q0 = _setup(env, condim=condim)
    assert ct.solver == 2                            # Newton with elliptic cones: the default
    assert (ct.condim_downgraded == 0) == (condim == "xml")
    orc = _scene(env, m)
    oq = ct.obj_qadr
    q0 = q0.copy(); q0[oq:oq + 3] = [0.84, 0.30, 0.853]     # (clear of the descending hand: landing ON the cube, the claw sticks to it -- noslip)
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    q, v, lag = od.step(q, v, lag, q0[d.qadr].copy(), n=300)
    cz, x0 = q[oq + 2], q[oq]
    assert abs(cz - 0.85) < 1e-3                     # table top 0.82 + half the cube
    way = [np.array([0.68, 0.30, cz + 0.15]), np.array([0.68, 0.30, cz + 0.03])] + [np.array([0.68 + 0.01 * k, 0.30, cz + 0.03]) for k in range(1, 22)]
    tipped = False
    for i, tg in enumerate(way):
        qt = _ik(env, m, f, orc, q, tg)
        ctrl = qt[d.qadr].copy(); ctrl[7:] = q[d.qadr[7:]]
        for _ in range(8 if i < 2 else 2):
            q, v, lag = od.step(q, v, lag, ctrl, n=75)
        assert np.all(np.isfinite(q))
        tipped = tipped or abs(q[oq + 3]) < 0.8       # more than ~70 degrees off its first face
    assert tipped == (condim == "3")
    if condim == "xml":
        assert q[oq + 3] > 0.99                       # still on the face it stood on
    assert 0.05 < q[oq] - x0 < 0.25 and abs(q[oq + 2] - cz) < 3e-3 and abs(q[oq + 1] - 0.30) < 0.05 and orc.is_valid(q)[0]
    ctrl = q[d.qadr].copy()                          # the hand holds where it is: friction stops the cube
    for _ in range(6):
        q, v, lag = od.step(q, v, lag, ctrl, n=75)
    assert np.abs(v[d.nd:d.nd + 3]).max() < 1e-4 and abs(q[oq + 2] - cz) < 1e-3


def test_gripper_slides_the_cube_with_pyramidal_cones():
    """the same push with pyramidal cones (selectable; round 4's default): the cube slides ahead of the hand on the face it stands on"""
    env = "SawyerPushObstacle-v0"
    m, f, d, ct, od, q0 = _setup(env, cone="pyramidal")
    orc = _scene(env, m)
    oq = ct.obj_qadr
    q0 = q0.copy(); q0[oq:oq + 3] = [0.84, 0.30, 0.853]
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    q, v, lag = od.step(q, v, lag, q0[d.qadr].copy(), n=300)
    cz, x0 = q[oq + 2], q[oq]
    way = [np.array([0.68, 0.30, cz + 0.15]), np.array([0.68, 0.30, cz + 0.03])] + [np.array([0.68 + 0.01 * k, 0.30, cz + 0.03]) for k in range(1, 13)]
    for i, tg in enumerate(way):
        qt = _ik(env, m, f, orc, q, tg)
        ctrl = qt[d.qadr].copy(); ctrl[7:] = q[d.qadr[7:]]
        for _ in range(8 if i < 2 else 2):
            q, v, lag = od.step(q, v, lag, ctrl, n=75)
        assert np.all(np.isfinite(q)) and orc.is_valid(q)[0]
    assert 0.015 < q[oq] - x0 < 0.15 and abs(q[oq + 2] - cz) < 2e-3 and abs(q[oq + 1] - 0.30) < 0.03 and q[oq + 3] > 0.99


@pytest.mark.parametrize("grasp_dz", [-0.02, 0.0])
def test_can_is_pinched_and_lifted_by_friction(grasp_dz):
    """Lift: the open gripper comes down over the can (pointing down), the finger servos (kp 10000, +-20 N) close on it, the
    arm rises: the can comes along, held by friction alone (mu 0.95, can 15 g) -- and falls when the fingers open.  Gripped
    around its body (2 cm below its centre) or by the finger tips at the height of its centre: since the noslip pass the second
    holds too (without it the can pivoted about the line through the two contacts and crept out)."""
    env = "SawyerLiftObstacle-v0"
    m, f, d, ct, od, q0 = _setup(env)
    orc = _scene(env, m)
    oq = ct.obj_qadr
    down = np.array([0.0, 0.0, 1.0, 0.0])
    q, v = q0.copy(), np.zeros(od.nv)
    lag = od.forward(q, v[:d.nd], want_M=False)[0]
    can = q[oq:oq + 3].copy()
    OPEN, CLOSE = np.full(2, -0.0115), np.full(2, 0.0208)

    def go(z, grip, steps, q, v, lag):
        qt = _ik(env, m, f, orc, q, np.array([can[0], can[1], z]), quat=down)
        ctrl = qt[d.qadr].copy(); ctrl[7:] = grip
        for _ in range(steps):
            q, v, lag = od.step(q, v, lag, ctrl, n=75)
        return q, v, lag

    for z in [1.40, 1.35]:
        q, v, lag = go(z, OPEN, 6, q, v, lag)
    for z in np.arange(1.30, 0.869, -0.05):
        q, v, lag = go(z, OPEN, 3, q, v, lag)
    # the grip site sits at the finger tips: 2 cm below the can's centre the pads span its body
    zg = can[2] + grasp_dz
    q, v, lag = go(zg, OPEN, 5, q, v, lag)
    assert np.abs(q[oq:oq + 2] - can[:2]).max() < 5e-3 and abs(_eef(orc, f, q)[2] - zg) < 5e-3       # straddling the can
    q, v, lag = go(zg, CLOSE, 6, q, v, lag)
    grip = q[d.qadr[7:]]
    # the fingers stopped ON the can (gap 37.5 - (q_l + q_r) mm vs its 50 mm; with the noslip pass the floor's friction holds
    # the can where it stands, so the fingers need not meet it symmetrically)
    # (round 6, the XML's condim: the can's torsional friction on the floor holds it harder still -- one finger may stay at its open stop
    #  -0.0115 while the other brings the can to it)
    assert np.all(grip > -0.0116) and np.all(grip < -0.0015) and -0.0200 < grip.sum() < -0.0080
    off = _eef(orc, f, q) - q[oq:oq + 3]
    for z in (0.90, 0.95, 0.98):
        q, v, lag = go(z, CLOSE, 3, q, v, lag)
    q, v, lag = go(0.98, CLOSE, 6, q, v, lag)
    assert q[oq + 2] > can[2] + 0.10 and np.abs(v[d.nd:d.nd + 3]).max() < 5e-3          # lifted 10+ cm and held ...
    assert np.abs((_eef(orc, f, q) - q[oq:oq + 3]) - off).max() < 1e-3                    # ... without slipping in the grasp
    q, v, lag = go(0.98, OPEN, 6, q, v, lag)
    assert q[oq + 2] < can[2] + 0.01                                                          # released: back on the bin floor
