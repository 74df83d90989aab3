"""Callers: the restated agent-side planning logic vs the reference's rules, with a scripted
validity oracle (CPU), plus the batched device forms vs the per-env restatement (GPU)."""
import types

import numpy as np
import pytest

from mopa_rl_amd.agent_planning import PlanningMixin, clip_target_to_limits, handle_invalid_target


class _FakePlanner:
    def __init__(self, valid_fn, plan_result=None):
        self.valid_fn = valid_fn
        self.plan_result = plan_result
        self.n_valid_calls = 0
        self.plan_calls = []

    def isValidState(self, q):
        self.n_valid_calls += 1
        return bool(self.valid_fn(np.asarray(q)))

    def plan(self, start, goal, timelimit=None):
        self.plan_calls.append((start.copy(), goal.copy(), timelimit))
        return self.plan_result


def _agent(valid_fn, plan_result=None, simple_result=None, **cfg):
    a = PlanningMixin()
    c = dict(omega=0.7, ac_space_type="piecewise", action_range=0.5, timelimit=1.0, simple_planner_timelimit=0.05,
             interpolation=True, joint_margin=0.001)
    c.update(cfg)
    a._config = types.SimpleNamespace(**c)
    a._planner = _FakePlanner(valid_fn, plan_result)
    a._simple_planner = _FakePlanner(valid_fn, simple_result)
    a._ref_joint_pos_indexes = list(range(7))
    # reference layout: one entry per qpos address (env/base.py:62-88); here 7 arm hinges + 2 gripper slides
    a._jnt_indices = np.arange(9)
    a._jnt_minimum = np.array([-3.0503, -3.8, -3.0426, -3.0439, -2.9761, -2.9761, -4.7124, -0.008, -0.008])
    a._jnt_maximum = np.array([3.0503, 1.25, 3.0426, 3.0439, 2.9761, 2.9761, 4.7124, 0.015, 0.015])
    a._is_jnt_limited = np.ones(9, dtype=bool)
    a._ac_low, a._ac_high = -1.0, 1.0
    return a


def test_is_planner_ac_and_displacement_maps():
    a = _agent(lambda q: True)
    assert not a.is_planner_ac({"default": np.array([0.1, -0.69, 0, 0, 0, 0, 0, 0.9])})     # gripper entry ignored
    assert a.is_planner_ac({"default": np.array([0.1, -0.71, 0, 0, 0, 0, 0, 0.0])})
    ac = np.array([0.0, 0.35, -0.7, 0.85, 1.0, -1.0, 0.69])
    d = a.convert2planner_displacement(ac, 0.05)
    np.testing.assert_allclose(d[:2], [0.0, 0.025])                      # linear inside omega
    assert d[4] == pytest.approx(0.5) and d[5] == pytest.approx(-0.5)    # |ac|=1 -> action_range
    assert d[3] == pytest.approx(0.05 + 0.45 * 0.5)
    small = np.array([0.01, -0.03, 0.049])
    np.testing.assert_allclose(a.invert_displacement(small, 0.05), small * 0.7 / 0.05)
    a2 = _agent(lambda q: True, ac_space_type="normal")
    np.testing.assert_allclose(a2.invert_displacement(a2.convert2planner_displacement(ac, 0.05), 0.05), ac)


def test_clip_qpos_only_when_out_of_limits():
    a = _agent(lambda q: True)
    q = np.array([0.1, -0.2, 0.3, 0, 0, 0, 0.0, 0.0, 0.0])
    assert a.clip_qpos(q) is q
    q2 = q.copy(); q2[1] = 1.3
    c = a.clip_qpos(q2)
    assert c[1] == pytest.approx(1.25 - 0.001) and c[0] == pytest.approx(0.1)


def test_simple_interpolate_step_rule():
    a = _agent(lambda q: True)
    cur = np.zeros(9)
    tgt = np.zeros(9); tgt[0] = 0.5; tgt[3] = -0.2
    traj, success, valid, exact = a.simple_interpolate(cur, tgt, 0.05)
    # scaling = 0.5 / (0.8*0.05) = 12.5 -> int() = 12 equal steps + the exact target
    assert success and valid and exact and traj.shape == (13, 9)
    np.testing.assert_allclose(np.diff(traj[:12, 0]), 0.04, atol=1e-15)
    assert np.all(np.abs(np.diff(np.vstack([cur, traj])[:, :7], axis=0)) <= 0.04 + 1e-12)
    np.testing.assert_array_equal(traj[-1], tgt)
    assert a._planner.n_valid_calls == 12
    # within the bound: one (checked) step onto the target, then the target again -- as the reference does
    tgt2 = np.zeros(9); tgt2[2] = 0.03
    traj, success, *_ = a.simple_interpolate(cur, tgt2, 0.05)
    assert success and traj.shape == (2, 9)


def test_simple_interpolate_blocked_and_fallbacks():
    blocked = lambda q: q[0] < 0.2
    a = _agent(blocked)
    cur = np.zeros(9); tgt = np.zeros(9); tgt[0] = 0.5
    traj, success, valid, exact = a.simple_interpolate(cur, tgt, 0.05)
    assert not success and not valid and not exact
    assert len(traj) == 5 and traj[-1][0] == 0.5 and np.all(traj[:4, 0] < 0.2)      # 4 valid steps + the target
    # use_planner: simple planner first, then the main planner, else [target]
    path = np.array([[0.1] * 9, [0.5] * 9])
    a = _agent(blocked, plan_result=(path, True, True, True), simple_result=(np.array([[-4.0] * 9]), False, True, False))
    traj, success, valid, exact = a.simple_interpolate(cur, tgt, 0.05, use_planner=True)
    assert success and len(a._simple_planner.plan_calls) == 1 and a._simple_planner.plan_calls[0][2] == 0.05
    assert len(a._planner.plan_calls) == 1 and a._planner.plan_calls[0][2] == 1.0
    a = _agent(blocked, plan_result=(np.array([[-4.0] * 9]), False, True, False), simple_result=(np.array([[-4.0] * 9]), False, True, False))
    traj, success, valid, exact = a.simple_interpolate(cur, tgt, 0.05, use_planner=True)
    assert not success and not exact and len(traj) == 1 and traj[0][0] == 0.5


def test_plan_uses_interpolation_first_then_planner_and_densifies():
    ok = _agent(lambda q: True)
    cur = np.zeros(9); tgt = np.zeros(9); tgt[0] = 0.2
    traj, success, interpolation, valid, exact = ok.plan(cur, tgt, ac_scale=0.05)
    assert success and interpolation and len(ok._planner.plan_calls) == 0
    # straight line blocked -> main planner; its long segments are densified through simple_interpolate(use_planner=True)
    way = np.zeros((2, 9)); way[0, 1] = 0.3; way[1, 0] = 0.5; way[1, 1] = 0.3
    blocked_line = lambda q: not (0.1 < q[0] < 0.3 and abs(q[1]) < 1e-9)
    a = _agent(blocked_line, plan_result=(way, True, True, True))
    traj, success, interpolation, valid, exact = a.plan(cur, tgt * 2.5, ac_scale=0.05)
    assert success and not interpolation
    assert len(traj) > 2 and np.all(np.abs(np.diff(np.vstack([cur, traj])[:, :7], axis=0)) <= 0.05 + 1e-12)
    np.testing.assert_allclose(traj[-1], way[1])


def test_clip_target_and_backoff():
    lo, hi = np.array([-1.0, -1.0, 0.0]), np.array([1.0, 1.0, 0.0])
    lim = np.array([True, True, False])
    np.testing.assert_allclose(clip_target_to_limits(np.array([2.0, -3.0, 7.0]), lo, hi, lim), [1.0, -1.0, 7.0])
    pi = _FakePlanner(lambda q: q[0] <= 0.5)
    cur = np.zeros(3); tgt = np.array([1.0, 0.0, 0.0])
    out, n = handle_invalid_target(pi, cur, tgt, 0.2, 100)
    assert n == 3 and out[0] == pytest.approx(0.4)
    out, n = handle_invalid_target(pi, cur, tgt, 0.2, 2)
    assert n == 2 and out[0] == pytest.approx(0.6)          # gave up: still invalid
    out, n = handle_invalid_target(pi, cur, np.array([0.3, 0, 0]), 0.2, 100)
    assert n == 0


# ---------------------------------------------------------------------------
# GPU: batched forms == the per-env restatement driven by the same HIP validity checker
# ---------------------------------------------------------------------------
@pytest.mark.gpu
def test_batched_interpolate_and_backoff_match_per_env():
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.agent_planning import handle_invalid_target_batch, simple_interpolate_batch
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import default_qpos, planner_inputs
    env = "SawyerPushObstacle-v0"
    pi = planner_inputs(env)
    sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
    bp = BatchPlanner(sc)

    class _One:
        def isValidState(self, q):
            return sc.is_valid_state(np.asarray(q, dtype=np.float64))

    a = _agent(lambda q: True)
    a._planner = _One()
    from mopa_rl_amd.scene import qpos_joint_arrays
    a._jnt_indices, a._jnt_minimum, a._jnt_maximum, a._is_jnt_limited = qpos_joint_arrays(pi.model, float32_limits=True)
    rng = np.random.default_rng(3)
    E = 96
    q0 = default_qpos(env, pi.model)
    cur = np.repeat(q0[None], E, axis=0)
    cur[:, :7] += rng.normal(0, 0.02, (E, 7))
    tgt = cur.copy()
    tgt[:, :7] += rng.uniform(-0.5, 0.5, (E, 7)) * (rng.random((E, 7)) < 0.6)
    tgt[:, :7] = np.clip(tgt[:, :7], pi.jnt_minimum, pi.jnt_maximum)
    tc, tt = torch.from_numpy(cur).cuda(), torch.from_numpy(tgt).cuda()
    traj, tlen, success, nsteps = simple_interpolate_batch(bp, tc, tt, pi.spec.ac_scale, pi.ref_joint_pos_indexes)
    torch.cuda.synchronize()
    traj, tlen, success = traj.cpu().numpy(), tlen.cpu().numpy(), success.cpu().numpy()
    n_fail = 0
    for e in range(E):
        rt, rs, rv, rx = a.simple_interpolate(cur[e], tgt[e], pi.spec.ac_scale)
        assert bool(success[e]) == rs and tlen[e] == len(rt), e
        assert np.array_equal(traj[e, :tlen[e]], rt), e        # bit-identical waypoints
        n_fail += not rs
    assert 0 < n_fail < E
    # back-off of invalid targets
    bad_t = tgt.copy()
    bad_t[:, :7] = np.clip(cur[:, :7] + rng.uniform(-1.5, 1.5, (E, 7)), pi.jnt_minimum, pi.jnt_maximum)
    out, trials, valid = handle_invalid_target_batch(bp, tc, torch.from_numpy(bad_t).cuda(), pi.spec.step_size, 25)
    torch.cuda.synchronize()
    out, trials, valid = out.cpu().numpy(), trials.cpu().numpy(), valid.cpu().numpy()
    moved = 0
    for e in range(E):
        ro, rn = handle_invalid_target(a._planner, cur[e], bad_t[e], pi.spec.step_size, 25)
        # both forms sum the squares of the Euclidean norm left to right (agent_planning.norm_seq): bit-identical
        assert rn == trials[e] and np.array_equal(ro, out[e]), e
        assert bool(valid[e]) == a._planner.isValidState(out[e])
        moved += rn > 0
    assert moved > 0
