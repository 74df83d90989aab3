"""Pin the oracle's convex-mesh narrow phase (the can of SawyerLiftObstacle; MuJoCo collides mesh geoms through their
convex hull with libccd MPR, `mjc_Convex`, and planes through `mjc_PlaneConvex`):
  * a box given as an 8-vertex mesh must behave like the box primitive;
  * against the independent support-function reference (tests/geom_ref.py) MPR reports "disjoint" exactly when the
    shapes are disjoint and never under-estimates the depth;
  * the compiled Lift scene carries the can's hull and every pair type of that scene is served."""
import numpy as np
import pytest

from geom_ref import BOX, CAPSULE, CYLINDER, MESH, PLANE, SPHERE, rand_rot, rand_size, signed_dist

I3 = np.eye(3)
FAR = 1.0e10


def box_verts(h):
    return np.array([[sx * h[0], sy * h[1], sz * h[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=float)


@pytest.fixture(scope="module")
def can():
    from mopa_rl_amd.scene import load_scene
    m = load_scene("sawyer_lift_obstacle")
    return m.mesh_vert[m.mesh_vertadr[0]:m.mesh_vertadr[0] + m.mesh_vertnum[0]].copy()


def test_lift_scene_carries_the_can_hull(can):
    from mopa_rl_amd.mjcf import GEOM_MESH
    from mopa_rl_amd.scene import load_scene
    m = load_scene("sawyer_lift_obstacle")
    g = int(np.where(m.geom_type == GEOM_MESH)[0][0])
    assert m.geom_dataid[g] == 0 and (m.geom_dataid >= 0).sum() == 1 and len(m.pair_geom) == 301
    assert (m.pair_geom == g).any(axis=1).sum() == 28                       # SURVEY section 8 scene table
    # a soda can: radius ~25 mm, height ~80 mm, centred (re-centred at the volume centroid by the compiler)
    r = np.hypot(can[:, 0], can[:, 1])
    assert 100 <= len(can) <= 120 and 0.0245 < r.max() < 0.0255 and 0.079 < np.ptp(can[:, 2]) < 0.081
    assert np.abs(can.mean(0)).max() < 5e-3 and np.allclose(m.geom_size[g], 0.5 * np.ptp(can, axis=0))


def test_box_mesh_behaves_like_the_box_primitive(oracle_mod):
    gd, gdm = oracle_mod.geom_dist, oracle_mod.geom_dist_mesh
    rng = np.random.default_rng(0)
    for _ in range(30):
        h = rng.uniform(0.02, 0.2, 3)
        V, mb, pb = box_verts(h), rand_rot(rng), rng.uniform(-0.1, 0.1, 3)
        # plane: deepest vertex, identical up to round-off
        mp, pp = rand_rot(rng), rng.uniform(-0.3, 0.3, 3)
        assert gdm(PLANE, np.zeros(3), pp, mp, V, pb, mb) == pytest.approx(gd(PLANE, np.zeros(3), pp, mp, BOX, h, pb, mb), abs=1e-14)
        # sphere: closed form vs MPR (tolerance of the portal refinement)
        r = rng.uniform(0.02, 0.1)
        ps = pb + rand_rot(rng) @ np.array([rng.uniform(0, 0.3), 0, 0])
        ref = gd(SPHERE, [r, 0, 0], ps, I3, BOX, h, pb, mb)
        got = gdm(SPHERE, [r, 0, 0], ps, I3, V, pb, mb)
        if ref > 1e-6:
            assert got == FAR
        elif ref < -1e-6:
            assert got <= ref + 2e-6            # MPR never under-estimates the depth (it over-estimates deep overlaps)
            if ref > -0.005:
                assert got > 1.5 * ref - 1e-4   # ... and is tight for the shallow contacts the threshold cares about
    # symmetric known answers: sphere pressed 2 mm into a face / box face-to-face 3 mm
    V = box_verts([0.1, 0.1, 0.05])
    assert gdm(SPHERE, [0.03, 0, 0], [0, 0, 0.078], I3, V, [0, 0, 0], I3) == pytest.approx(-0.002, abs=2e-6)
    assert gdm(BOX, [0.05, 0.05, 0.05], [0, 0, 0.097], I3, V, [0, 0, 0], I3) == pytest.approx(-0.003, abs=2e-6)
    assert gdm(CYLINDER, [0.04, 0.06, 0], [0, 0, 0.109], I3, V, [0, 0, 0], I3) == pytest.approx(-0.001, abs=2e-6)
    assert gdm(CAPSULE, [0.02, 0.05, 0], [0, 0, 0.1195], I3, V, [0, 0, 0], I3) == pytest.approx(-0.0005, abs=2e-6)


@pytest.mark.parametrize("t1", [SPHERE, CAPSULE, CYLINDER, BOX])
def test_can_vs_primitives_against_numeric_reference(oracle_mod, can, t1):
    gdm = oracle_mod.geom_dist_mesh
    rng = np.random.default_rng(10 + t1)
    n_pen = n_far = 0
    for _ in range(16):
        s1 = rand_size(rng, t1) * 0.4
        m1, m2 = rand_rot(rng), rand_rot(rng)
        p2 = rng.uniform(-0.1, 0.1, 3)
        dirn = rng.normal(size=3)
        p1 = p2 + dirn / np.linalg.norm(dirn) * rng.uniform(0.0, 0.15)
        ref = signed_dist(t1, s1, p1, m1, MESH, can, p2, m2, refine=12)
        got = gdm(t1, s1, p1, m1, can, p2, m2)
        if ref > 1e-5:
            assert got == FAR
            n_far += 1
        elif ref < -1e-5:
            assert got < 0 and got <= ref + 5e-6         # MPR never under-estimates the penetration (portal tolerance 1e-6)
            n_pen += 1
    assert n_pen >= 2 and n_far >= 2


def test_plane_can(oracle_mod, can):
    gdm = oracle_mod.geom_dist_mesh
    rng = np.random.default_rng(3)
    for _ in range(20):
        mp, pp, m2, p2 = rand_rot(rng), rng.uniform(-0.1, 0.1, 3), rand_rot(rng), rng.uniform(-0.1, 0.1, 3)
        want = ((can @ m2.T + p2 - pp) @ mp[:, 2]).min()
        assert gdm(PLANE, np.zeros(3), pp, mp, can, p2, m2) == pytest.approx(want, abs=1e-15)
    # standing on the table: bottom of the can 40.3 mm below its centre
    assert gdm(PLANE, np.zeros(3), [0, 0, 0], I3, can, [0, 0, 0.05], I3) == pytest.approx(0.05 + can[:, 2].min(), abs=1e-15)


def test_lift_validity_semantics(oracle_mod):
    """init_qpos of the Lift env is valid (training starts there), the can resting pose does not collide with the table
    beyond the threshold... and pushing the gripper into the can invalidates the state through a mesh pair."""
    from mopa_rl_amd.mjcf import GEOM_MESH
    from mopa_rl_amd.scene import default_qpos, planner_inputs
    pi = planner_inputs("SawyerLiftObstacle-v0")
    m = pi.model
    orc = oracle_mod.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    q = default_qpos("SawyerLiftObstacle-v0", m)
    ok, md = orc.is_valid(q)
    assert ok
    g = int(np.where(m.geom_type == GEOM_MESH)[0][0])
    mesh_pairs = np.where((m.pair_geom == g).any(axis=1))[0]
    # move the can into the gripper: some non-ignored mesh pair must report a penetration and the state turns invalid
    gpos, _ = orc.fk(q)
    names = m.all_geom_names
    finger = [i for i in range(len(m.geom_type)) if "claw" in names[m.geom_mjid[i]] or "finger" in names[m.geom_mjid[i]]]
    assert finger, "no finger geom found"
    ca = m.get_joint_qpos_addr("cube")
    q2 = q.copy()
    q2[ca:ca + 3] = gpos[finger[0]]
    ok2, md2 = orc.is_valid(q2)
    d = orc.pair_dist(q2)
    assert not ok2 and md2 < pi.spec.contact_threshold and (d[mesh_pairs] < 0).any()
