"""Host-side mirror of the reference's planner interface (no GPU needed: the native planner object is
replaced by a stub so only the Python logic restated from the reference is exercised)."""
import types

import numpy as np
import pytest

from mopa_rl_amd import planner_agent, sampling_based_planner as sbp
from mopa_rl_amd.sampling_based_planner import joint_convert


def test_joint_convert_uses_3_14_like_the_reference():
    # util/env.py:15-25 -- wraps with 3.14, not pi
    assert joint_convert(0.5) == pytest.approx(0.5)
    assert joint_convert(3.5) == pytest.approx(3.5 % 3.14 - 3.14)
    assert joint_convert(-3.5) == pytest.approx(-3.5 % -3.14 + 3.14)
    assert joint_convert(6.5) == pytest.approx(6.5 % 3.14)
    assert joint_convert(-0.25) == pytest.approx(-0.25)
    for a in np.linspace(-20, 20, 401):
        assert -3.14 <= joint_convert(a) <= 3.14


class _StubNative:
    """Stands in for PyKinematicPlanner: returns a canned state list."""

    def __init__(self, *a):
        self.args = a
        self.ret = None
        self.calls = []

    def plan(self, start, goal, timelimit):
        self.calls.append((np.array(start), np.array(goal), timelimit))
        return self.ret

    def isValidState(self, s):
        return bool(s[0] < 1.0)

    def getPlannerStatus(self):
        return b"Exact solution"


@pytest.fixture
def cfg():
    return types.SimpleNamespace(planner_type="rrt_connect", range=0.1, planner_objective="path_length", threshold=0.0,
                                 seed=1, _xml_path="sawyer_push_obstacle.xml", contact_threshold=-0.002, timelimit=1.0)


def _mk(monkeypatch, cfg, non_limited_idx=None):
    monkeypatch.setattr(sbp, "PyKinematicPlanner", _StubNative)
    return planner_agent.PlannerAgent(cfg, 7, non_limited_idx, passive_joint_idx=[3], ignored_contacts=[(1, 2)])


def test_constructor_argument_order_matches_pyx(monkeypatch, cfg):
    ag = _mk(monkeypatch, cfg)
    a = ag.planner.planner.args
    # planner.pyx:31-39: xml, algo, num_actions, opt, threshold, range, passive, glue, ignored, contact_threshold,
    # goal_bias, is_simplified, simplified_duration, seed
    assert a[0] == b"sawyer_push_obstacle.xml" and a[1] == b"rrt_connect" and a[2] == 7 and a[3] == b"path_length"
    assert a[4] == 0.0 and a[5] == 0.1 and a[6] == [3] and a[7] == [] and a[8] == [(1, 2)] and a[9] == -0.002
    assert a[10] == 0.05 and a[11] is False and a[12] == 0.1 and a[13] == 1


def test_plan_success_returns_traj_without_start(monkeypatch, cfg):
    ag = _mk(monkeypatch, cfg)
    states = [[0.0, 0.0], [0.1, 0.0], [0.2, 0.1]]
    ag.planner.planner.ret = states
    traj, success, valid, exact = ag.plan(np.array([0.0, 0.0]), np.array([0.2, 0.1]))
    assert success and valid and exact
    np.testing.assert_allclose(traj, states[1:])
    assert ag.planner.planner.calls[0][2] == 1.0         # default timelimit from config
    assert ag.get_planner_status() == "Exact solution"


def test_sentinel_rows(monkeypatch, cfg):
    ag = _mk(monkeypatch, cfg)
    ag.planner.planner.ret = [[-5.0] * 4]
    traj, success, valid, exact = ag.plan(np.zeros(4), np.ones(4), timelimit=0.05)
    assert not success and not valid and exact and traj.shape == (1, 4)
    ag.planner.planner.ret = [[-4.0] * 4]
    traj, success, valid, exact = ag.plan(np.zeros(4), np.ones(4))
    assert not success and valid and not exact


def test_unlimited_joint_unwrapping(monkeypatch, cfg):
    """sampling_based_planner.py:71-99: the native planner works on wrapped angles; the returned trajectory
    is rebuilt from successive differences and crosses the +-3.14 seam continuously."""
    ag = _mk(monkeypatch, cfg, non_limited_idx=[0])
    # wrapped path crossing the seam in the positive direction: 3.0 -> -3.1 (i.e. +0.18 through 3.14)
    ag.planner.planner.ret = [[3.0, 0.0], [-3.1, 0.1], [-3.0, 0.2]]
    start = np.array([3.0 + 2 * 3.14, 0.0])               # an un-wrapped start on another turn
    traj, states, valid, exact = ag.planner.plan(start, np.array([-3.0, 0.2]))
    assert ag.planner.planner.calls[-1][0][0] == pytest.approx(joint_convert(start[0]))
    np.testing.assert_allclose(traj[0], start)
    assert traj[1][0] == pytest.approx(start[0] + (3.14 - 3.0) + (-3.1 + 3.14))
    assert traj[2][0] == pytest.approx(traj[1][0] + 0.1)
    np.testing.assert_allclose(traj[:, 1], [0.0, 0.1, 0.2])
    # negative direction
    ag.planner.planner.ret = [[-3.0, 0.0], [3.1, 0.0]]
    traj, *_ = ag.planner.plan(np.array([-3.0, 0.0]), np.array([3.1, 0.0]))
    assert traj[1][0] == pytest.approx(-3.0 - ((3.14 - 3.1) + (-3.0 + 3.14)))


def test_is_valid_state_passthrough(monkeypatch, cfg):
    ag = _mk(monkeypatch, cfg)
    assert ag.isValidState(np.array([0.5])) and not ag.isValidState(np.array([2.0]))


def test_action_size_helper():
    assert planner_agent.action_size(7) == 7
    box = types.SimpleNamespace(shape=(8,))
    assert planner_agent.action_size(box) == 8
    d = types.SimpleNamespace(spaces={"default": types.SimpleNamespace(shape=(7,)), "g": types.SimpleNamespace(n=3, shape=())})
    assert planner_agent.action_size(d) == 10


def test_native_planner_rejects_unsupported_options():
    from mopa_rl_amd.planner import PyKinematicPlanner
    with pytest.raises(NotImplementedError):
        PyKinematicPlanner(b"sawyer_push_obstacle.xml", b"rrt", 7, b"path_length", 0.0, 0.1, [], [], [], -0.002, 0.05, False, 0.1, 0)
    with pytest.raises(NotImplementedError):
        PyKinematicPlanner(b"sawyer_push_obstacle.xml", b"rrt_connect", 7, b"", 0.0, 0.1, [], [b"a"], [], -0.002, 0.05, False, 0.1, 0)
    with pytest.raises(NotImplementedError):
        PyKinematicPlanner(b"sawyer_push_obstacle.xml", b"rrt_connect", 7, b"", 0.0, 0.1, [], [], [], -0.002, 0.05, True, 0.1, 0)
