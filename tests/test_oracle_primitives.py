"""Pin the oracle's narrow phase: closed-form known answers + an independent numeric reference
(tests/geom_ref.py: max over directions of the support-function separation).

Pair functions with a unique geometric answer (everything except the three MPR pairs) must match the
reference to 1e-8.  The MPR pairs ({capsule,cylinder,box}-cylinder) follow libccd's portal refinement as
MuJoCo 2.0 does: they report "no intersection" exactly when the shapes are disjoint, never under-estimate
the minimum translation depth, and are exact in symmetric configurations."""
import math

import numpy as np
import pytest

from geom_ref import BOX, CAPSULE, CYLINDER, PLANE, SPHERE, rand_rot, rand_size, signed_dist

I3 = np.eye(3)
FAR = 1.0e10


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


@pytest.fixture(scope="module")
def gd(oracle_mod):
    return oracle_mod.geom_dist


# ---------------- closed-form known answers ----------------
def test_sphere_sphere(gd):
    assert gd(SPHERE, [0.1, 0, 0], [0, 0, 0], I3, SPHERE, [0.2, 0, 0], [0.5, 0, 0], I3) == pytest.approx(0.2, abs=1e-15)
    assert gd(SPHERE, [0.1, 0, 0], [0, 0, 0], I3, SPHERE, [0.2, 0, 0], [0, 0.25, 0], I3) == pytest.approx(-0.05, abs=1e-15)


def test_plane_family(gd):
    P = ([0, 0, 0], [0, 0, 0], I3)
    assert gd(PLANE, *P, SPHERE, [0.1, 0, 0], [3, 2, 0.5], I3) == pytest.approx(0.4)
    # capsule tilted 90deg: lying on its side, lowest point = centre height - radius
    assert gd(PLANE, *P, CAPSULE, [0.05, 0.2, 0], [0, 0, 0.3], rot_y(math.pi / 2)) == pytest.approx(0.25)
    # capsule upright: lowest = centre - half - radius
    assert gd(PLANE, *P, CAPSULE, [0.05, 0.2, 0], [0, 0, 0.3], I3) == pytest.approx(0.05)
    # cylinder upright / on its side / tilted 45deg
    assert gd(PLANE, *P, CYLINDER, [0.1, 0.3, 0], [0, 0, 0.5], I3) == pytest.approx(0.2)
    assert gd(PLANE, *P, CYLINDER, [0.1, 0.3, 0], [0, 0, 0.5], rot_x(math.pi / 2)) == pytest.approx(0.4)
    assert gd(PLANE, *P, CYLINDER, [0.1, 0.3, 0], [0, 0, 0.5], rot_x(math.pi / 4)) == \
        pytest.approx(0.5 - (0.3 + 0.1) * math.sqrt(0.5))
    # box: deepest vertex
    assert gd(PLANE, *P, BOX, [0.1, 0.2, 0.3], [0, 0, 0.25], I3) == pytest.approx(-0.05)
    assert gd(PLANE, *P, BOX, [0.1, 0.1, 0.1], [0, 0, 1.0], rot_x(math.pi / 4)) == pytest.approx(1.0 - 0.1 * math.sqrt(2))
    # a rotated plane: normal along +x
    assert gd(PLANE, [0, 0, 0], [1, 0, 0], rot_y(math.pi / 2), SPHERE, [0.1, 0, 0], [1.5, 7, 7], I3) == pytest.approx(0.4)


def test_sphere_box_inside_and_outside(gd):
    B = (BOX, [0.1, 0.2, 0.3], [0, 0, 0], I3)
    S = lambda p: (SPHERE, [0.05, 0, 0], p, I3)
    assert gd(*S([0.3, 0, 0]), *B) == pytest.approx(0.15)                       # face
    assert gd(*S([0.2, 0.3, 0]), *B) == pytest.approx(math.hypot(0.1, 0.1) - 0.05)   # edge
    assert gd(*S([0.2, 0.3, 0.4]), *B) == pytest.approx(math.sqrt(0.03) - 0.05)      # vertex
    assert gd(*S([0.08, 0, 0]), *B) == pytest.approx(-0.02 - 0.05)              # centre inside: nearest face
    assert gd(*S([0, 0, 0]), *B) == pytest.approx(-0.1 - 0.05)


def test_sphere_cylinder(gd):
    C = (CYLINDER, [0.1, 0.3, 0], [0, 0, 0], I3)
    S = lambda p: (SPHERE, [0.05, 0, 0], p, I3)
    assert gd(*S([0.4, 0, 0]), *C) == pytest.approx(0.25)            # side
    assert gd(*S([0, 0, 0.5]), *C) == pytest.approx(0.15)            # cap
    assert gd(*S([0.2, 0, 0.4]), *C) == pytest.approx(math.hypot(0.1, 0.1) - 0.05)   # rim
    assert gd(*S([0.05, 0, 0]), *C) == pytest.approx(-0.05 - 0.05)   # inside, nearest = side
    assert gd(*S([0, 0, 0.28]), *C) == pytest.approx(-0.02 - 0.05)   # inside, nearest = cap


def test_capsule_capsule_parallel_and_crossed(gd):
    A = (CAPSULE, [0.05, 0.2, 0], [0, 0, 0], I3)
    assert gd(*A, CAPSULE, [0.03, 0.1, 0], [0.3, 0, 0.05], I3) == pytest.approx(0.3 - 0.08)       # parallel, overlapping spans
    assert gd(*A, CAPSULE, [0.03, 0.1, 0], [0, 0, 0.5], I3) == pytest.approx(0.2 - 0.08)          # collinear end to end
    assert gd(*A, CAPSULE, [0.03, 0.1, 0], [0.2, 0, 0], rot_x(math.pi / 2)) == pytest.approx(0.2 - 0.08)   # crossed
    assert gd(*A, CAPSULE, [0.03, 0.1, 0], [0.06, 0, 0], rot_x(math.pi / 2)) == pytest.approx(0.06 - 0.08)


def test_capsule_box_cases(gd):
    B = (BOX, [0.1, 0.2, 0.3], [0, 0, 0], I3)
    cap = lambda p, m: (CAPSULE, [0.02, 0.15, 0], p, m)
    assert gd(*cap([0.3, 0, 0], I3), *B) == pytest.approx(0.2 - 0.02)                    # parallel to a face
    assert gd(*cap([0.4, 0, 0], rot_y(math.pi / 2)), *B) == pytest.approx(0.4 - 0.15 - 0.1 - 0.02)   # end-on
    assert gd(*cap([0.25, 0.35, 0], I3), *B) == pytest.approx(math.hypot(0.15, 0.15) - 0.02)         # edge-parallel
    # axis pierces the box centre: segment SAT depth = distance to exit through the nearest pair of faces
    assert gd(*cap([0, 0, 0], I3), *B) == pytest.approx(-0.1 - 0.02)


def test_box_box_cases(gd):
    A = (BOX, [0.1, 0.2, 0.3], [0, 0, 0], I3)
    assert gd(*A, BOX, [0.1, 0.1, 0.1], [0.19, 0, 0], I3) == pytest.approx(-0.01)        # face overlap 1 cm
    assert gd(*A, BOX, [0.1, 0.1, 0.1], [0.25, 0, 0], I3) == pytest.approx(0.05)         # face gap
    # edge-edge: cube rotated 45deg about z and 45deg about x pokes an edge... use the numeric reference
    m2 = rot_x(0.7) @ rot_y(0.5)
    ref = signed_dist(BOX, [0.1, 0.2, 0.3], [0, 0, 0], I3, BOX, [0.1, 0.1, 0.1], [0.2, 0.25, 0.3], m2)
    got = gd(*A, BOX, [0.1, 0.1, 0.1], [0.2, 0.25, 0.3], m2)
    assert ref < 0 and got == pytest.approx(ref, abs=1e-8)


def test_mpr_symmetric_known_answers(gd):
    # coaxial cylinders stacked with 3 mm overlap
    d = gd(CYLINDER, [0.1, 0.2, 0], [0, 0, 0], I3, CYLINDER, [0.05, 0.1, 0], [0, 0, 0.297], I3)
    assert d == pytest.approx(-0.003, abs=2e-6)
    # cylinder standing on a box face, 2.5 mm deep
    d = gd(CYLINDER, [0.05, 0.1, 0], [0, 0, 0.3975], I3, BOX, [0.2, 0.2, 0.3], [0, 0, 0], I3)
    assert d == pytest.approx(-0.0025, abs=2e-6)
    # capsule lying across the flat top of a cylinder, 1 mm deep
    d = gd(CAPSULE, [0.03, 0.2, 0], [0, 0, 0.229], rot_y(math.pi / 2), CYLINDER, [0.1, 0.2, 0], [0, 0, 0], I3)
    assert d == pytest.approx(-0.001, abs=2e-6)
    # clearly separated -> sentinel
    assert gd(CYLINDER, [0.1, 0.2, 0], [0, 0, 0], I3, CYLINDER, [0.05, 0.1, 0], [0, 0, 0.31], I3) == FAR
    assert gd(CAPSULE, [0.03, 0.2, 0], [0.5, 0, 0], I3, CYLINDER, [0.1, 0.2, 0], [0, 0, 0], I3) == FAR


# ---------------- randomised comparison with the independent numeric reference ----------------
EXACT_PAIRS = [(SPHERE, SPHERE), (SPHERE, CAPSULE), (SPHERE, CYLINDER), (SPHERE, BOX), (CAPSULE, CAPSULE),
               (CAPSULE, BOX), (BOX, BOX)]


def _rand_pair(rng, t1, t2, dist_hi=0.5):
    s1, s2 = rand_size(rng, t1), rand_size(rng, t2)
    m1, m2 = rand_rot(rng), rand_rot(rng)
    p1 = rng.uniform(-0.1, 0.1, 3)
    dirn = rng.normal(size=3)
    dirn /= np.linalg.norm(dirn)
    return s1, p1, m1, s2, p1 + dirn * rng.uniform(0.0, dist_hi), m2


@pytest.mark.parametrize("t1,t2", EXACT_PAIRS)
def test_exact_pairs_match_numeric_reference(gd, t1, t2):
    rng = np.random.default_rng(100 * t1 + t2)
    n_pen = 0
    for _ in range(14):
        s1, p1, m1, s2, p2, m2 = _rand_pair(rng, t1, t2)
        ref = signed_dist(t1, s1, p1, m1, t2, s2, p2, m2, refine=12)
        got = gd(t1, s1, p1, m1, t2, s2, p2, m2)
        if (t1, t2) == (BOX, BOX) and ref > 0:
            assert 0 < got <= ref + 1e-9      # SAT gap is a lower bound of the true distance when disjoint
        else:
            assert got == pytest.approx(ref, abs=1e-8)
        n_pen += ref < 0
    assert n_pen >= 2


@pytest.mark.parametrize("t1,t2", [(CAPSULE, CYLINDER), (CYLINDER, CYLINDER), (CYLINDER, BOX)])
def test_mpr_pairs_vs_numeric_reference(gd, t1, t2):
    rng = np.random.default_rng(7 * t1 + t2)
    n_pen = 0
    for _ in range(14):
        s1, p1, m1, s2, p2, m2 = _rand_pair(rng, t1, t2, dist_hi=0.45)
        ref = signed_dist(t1, s1, p1, m1, t2, s2, p2, m2, refine=12)
        got = gd(t1, s1, p1, m1, t2, s2, p2, m2)
        if ref > 1e-5:
            assert got == FAR
        elif ref < -1e-5:
            assert got < 0 and got <= ref + 2e-6     # MPR never under-estimates the penetration
            n_pen += 1
    assert n_pen >= 2


def test_rigid_motion_invariance(gd):
    """Moving both shapes by the same rigid transform leaves every distance unchanged (to round-off)."""
    rng = np.random.default_rng(5)
    for t1, t2 in EXACT_PAIRS + [(CYLINDER, BOX)]:
        s1, p1, m1, s2, p2, m2 = _rand_pair(rng, t1, t2, dist_hi=0.3)
        d0 = gd(t1, s1, p1, m1, t2, s2, p2, m2)
        Rg, tg = rand_rot(rng), rng.uniform(-1, 1, 3)
        d1 = gd(t1, s1, Rg @ p1 + tg, Rg @ m1, t2, s2, Rg @ p2 + tg, Rg @ m2)
        if d0 == FAR:
            assert d1 == FAR
        else:
            assert d1 == pytest.approx(d0, abs=5e-6 if (t1, t2) == (CYLINDER, BOX) else 1e-12)


@pytest.mark.parametrize("t1,t2", [(CAPSULE, CYLINDER), (CYLINDER, CYLINDER), (CYLINDER, BOX)])
def test_convex_axes_pretest_never_changes_a_distance(oracle_mod, gd, t1, t2):
    """The separating-axis pre-test of the cylinder classes is an accelerator: with it switched off every such pair goes to
    the portal refinement, and the distance (FAR or the penetration depth) must come out the same -- sampled where the
    pre-test matters: flat discs and long posts a few millimetres to a few centimetres apart, and in contact."""
    rng = np.random.default_rng(100 * t1 + t2)
    n_culled, n_pen = 0, 0
    try:
        for k in range(3000):
            flat = k % 2 == 0
            s1 = np.array([rng.uniform(0.02, 0.06), rng.uniform(0.01, 0.03) if flat else rng.uniform(0.05, 0.4), 0.0])
            s2 = (rng.uniform(0.01, 0.4, 3) if t2 == BOX else
                  np.array([rng.uniform(0.02, 0.06), rng.uniform(0.01, 0.03) if k % 4 < 2 else rng.uniform(0.1, 0.4), 0.0]))
            m1, m2 = rand_rot(rng), (rand_rot(rng) if k % 3 else I3)
            p1 = rng.uniform(-0.2, 0.2, 3)
            # place shape 1 near the surface of shape 2: along a random direction, at a distance around the sum of extents
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            reach = np.linalg.norm(s2) if t2 == BOX else np.hypot(s2[0], s2[1])
            p2 = p1 + u * rng.uniform(0.0, 1.2) * (reach + np.hypot(s1[0], s1[1]))
            oracle_mod.set_convex_axes_pretest(True)
            with_pre = gd(t1, s1, p1, m1, t2, s2, p2, m2)
            oracle_mod.set_convex_axes_pretest(False)
            without = gd(t1, s1, p1, m1, t2, s2, p2, m2)
            assert with_pre == without, (k, with_pre, without)
            n_pen += without != FAR
            # how often the pre-test is what decides: enclosing capsules overlap, shapes disjoint
            enc = gd(CAPSULE, s1, p1, m1, CAPSULE if t2 == CYLINDER else BOX, s2, p2, m2)
            n_culled += (enc <= 1e-9) and without == FAR
    finally:
        oracle_mod.set_convex_axes_pretest(True)
    assert n_pen > 100 and n_culled > 30


def test_convex_axes_pretest_scene_distances_unchanged(oracle_mod):
    """... and on the scenes: every pair distance of sampled states (near-contact and uniform) is the same with and without."""
    from conftest import SUPPORTED_ENVS, sample_states
    from mopa_rl_amd.scene import planner_inputs
    try:
        for env in SUPPORTED_ENVS:
            pi = planner_inputs(env)
            orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
            for mode in ("near", "uniform"):
                qa, rows = sample_states(pi, 300, 3, mode)
                for i in range(len(qa)):
                    q = rows[0].copy()
                    q[pi.ref_joint_pos_indexes] = qa[i]
                    oracle_mod.set_convex_axes_pretest(True)
                    a = orc.pair_dist(q)
                    oracle_mod.set_convex_axes_pretest(False)
                    b = orc.pair_dist(q)
                    assert np.array_equal(a, b), (env, mode, i)
    finally:
        oracle_mod.set_convex_axes_pretest(True)
