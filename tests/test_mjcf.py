"""Scene compiler: committed compiled scenes have the expected structure, and (when the reference's
asset tree is present) re-compiling the MJCF reproduces them exactly."""
import os

import numpy as np
import pytest

from mopa_rl_amd import mjcf
from mopa_rl_amd.scene import ENV_SPECS, load_scene, planner_inputs, scene_path

XML_DIR = "/root/reference/env/assets/xml"
EXPECT = {
    "sawyer_push_obstacle": dict(nq=36, nbody=50, ngeom=87, ncoll=27, npair=250),
    "sawyer_lift_obstacle": dict(nq=34, nbody=61, ngeom=100, ncoll=29, npair=301),
    "sawyer_assembly_obstacle": dict(nq=34, nbody=53, ngeom=93, ncoll=39, npair=524),
    "pusher_obstacle": dict(nq=16, nbody=26, ngeom=33, ncoll=16, npair=87),
}


@pytest.mark.parametrize("scene", sorted(EXPECT))
def test_compiled_scene_counts(scene):
    m = load_scene(scene)
    e = EXPECT[scene]
    assert (m.nq, len(m.body_names), len(m.all_geom_names), len(m.geom_type), len(m.pair_geom)) == \
        (e["nq"], e["nbody"], e["ngeom"], e["ncoll"], e["npair"])
    # pairs are ordered type1 <= type2 and reference valid geoms on different bodies
    t = m.geom_type[m.pair_geom]
    assert np.all(t[:, 0] <= t[:, 1])
    assert np.all(m.geom_body[m.pair_geom[:, 0]] != m.geom_body[m.pair_geom[:, 1]])
    # quaternions normalised, axes unit
    assert np.allclose(np.linalg.norm(m.body_quat, axis=1), 1.0, atol=1e-12)
    assert np.allclose(np.linalg.norm(m.geom_quat, axis=1), 1.0, atol=1e-12)


def test_push_pair_histogram():
    h = mjcf.pair_type_histogram(load_scene("sawyer_push_obstacle"))
    assert h == {"plane-sphere": 3, "plane-box": 5, "plane-capsule": 4, "plane-cylinder": 2, "sphere-box": 30,
                 "box-box": 38, "capsule-box": 43, "cylinder-box": 46, "sphere-cylinder": 23, "capsule-cylinder": 30,
                 "cylinder-cylinder": 12, "sphere-capsule": 11, "capsule-capsule": 3}


def test_push_qpos_layout_and_planner_inputs():
    pi = planner_inputs("SawyerPushObstacle-v0")
    m = pi.model
    assert pi.ref_joint_pos_indexes == list(range(7))
    assert pi.passive_joint_idx == list(range(7, 36))
    assert m.get_joint_qpos_addr("rc_close") == 7 and m.get_joint_qpos_addr("cube") == 27
    assert m.get_joint_qpos_addr("target_x") == 34
    # ignored = cube x every geom of table/bin1 (rl/trainer.py:62-67): 1 x 18 ordered pairs
    cube = m.geom_name2id("cube")
    assert len(pi.ignored_contacts) == 18 and all(b == cube for _, b in pi.ignored_contacts)
    assert len(pi.non_limited_idx) == 0
    np.testing.assert_allclose(pi.jnt_minimum, [-3.0503, -3.8, -3.0426, -3.0439, -2.9761, -2.9761, -4.7124])
    np.testing.assert_allclose(pi.jnt_maximum, [3.0503, 1.25, 3.0426, 3.0439, 2.9761, 2.9761, 4.7124])


def test_pusher_unlimited_joint():
    pi = planner_inputs("PusherObstacle-v0")
    assert list(pi.non_limited_idx) == [0]
    assert pi.jnt_minimum[0] == -3.14 and pi.jnt_maximum[0] == 3.14     # env/base.py:85-86
    # `ref=` on the target/box slides -> qpos0
    m = pi.model
    assert m.qpos0[m.get_joint_qpos_addr("box_x")] == pytest.approx(0.1)
    assert m.qpos0[m.get_joint_qpos_addr("box_y")] == pytest.approx(-0.1)


def test_fromto_capsule():
    m = load_scene("pusher_obstacle")
    g = m.all_geom_names.index("link0")
    c = list(m.geom_mjid).index(g)
    np.testing.assert_allclose(m.geom_size[c], [0.01, 0.05, 0.0])
    np.testing.assert_allclose(m.geom_pos[c], [0.05, 0, 0], atol=1e-15)
    # local z axis of the geom frame must point along +x
    w, x, y, z = m.geom_quat[c]
    zax = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])
    np.testing.assert_allclose(zax, [1, 0, 0], atol=1e-12)


def test_json_roundtrip_is_bit_exact(tmp_path):
    m = load_scene("sawyer_push_obstacle")
    p = tmp_path / "s.json"
    m.save(str(p))
    m2 = mjcf.CompiledModel.load(str(p))
    for k in mjcf.CompiledModel._ARRAYS:
        a, b = getattr(m, k), getattr(m2, k)
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), k


@pytest.mark.skipif(not os.path.isdir(XML_DIR), reason="reference asset tree not present on this machine")
@pytest.mark.parametrize("scene", sorted(EXPECT))
def test_committed_scene_matches_fresh_compile(scene):
    fresh = mjcf.compile_mjcf(os.path.join(XML_DIR, scene + ".xml"))
    committed = mjcf.CompiledModel.load(scene_path(scene))
    for k in mjcf.CompiledModel._ARRAYS:
        assert getattr(fresh, k).tobytes() == getattr(committed, k).tobytes(), k
    for k in mjcf.CompiledModel._LISTS:
        assert getattr(fresh, k) == getattr(committed, k), k


def test_mesh_scene_is_flagged():
    from mopa_rl_amd.scene import has_mesh_collider
    assert has_mesh_collider(load_scene("sawyer_lift_obstacle"))
    assert not has_mesh_collider(load_scene("sawyer_push_obstacle"))
    assert set(ENV_SPECS) >= {"SawyerPushObstacle-v0", "PusherObstacle-v0"}
