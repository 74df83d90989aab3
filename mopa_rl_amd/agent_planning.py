"""Callers of the validity checker / planner.

`PlanningMixin` restates, with the same names, arguments and return tuples, the planner-facing half of the
reference's SAC/TD3 agents (rl/sac_agent.py:145-318, duplicated in rl/td3_agent.py): `is_planner_ac`,
`convert2planner_displacement`, `invert_displacement`, `clip_qpos`, `simple_interpolate`, `plan`, `isValidState`.
`handle_invalid_target` restates the invalid-target back-off of the rollout runner (rl/mopa_rollouts.py:119-143).
Everything here is host-side numpy, exactly where the reference keeps it; validity comes from whatever object
exposes `isValidState` / `plan` (the `PlannerAgent` mirror -> libmopa_hip.so).

`simple_interpolate_batch` / `handle_invalid_target_batch` are the device-resident forms for E environments at once:
the same rules evaluated with one batched validity launch per step instead of a Python loop per environment.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


# ---------------------------------------------------------------------------
# single-environment restatement (reference semantics, quirks included)
# ---------------------------------------------------------------------------
class PlanningMixin:
    """Needs: self._config (omega, ac_space_type, action_range, timelimit, simple_planner_timelimit, interpolation,
    joint_margin), self._planner, self._simple_planner (PlannerAgent-like), self._ref_joint_pos_indexes,
    self._jnt_indices, self._jnt_minimum, self._jnt_maximum, self._is_jnt_limited, self._ac_low, self._ac_high
    (bounds of ac_space['default'], reference: self._ac_space['default'].low[0] / .high[0])."""

    # rl/sac_agent.py:148-153
    def is_planner_ac(self, ac) -> bool:
        a = np.asarray(ac["default"] if isinstance(ac, dict) else ac)[: len(self._ref_joint_pos_indexes)]
        return bool(np.any(a < -self._config.omega) or np.any(a > self._config.omega))

    # rl/sac_agent.py:155-156
    def isValidState(self, state) -> bool:
        return self._planner.isValidState(state)

    # rl/sac_agent.py:158-175
    def convert2planner_displacement(self, ac, ac_scale):
        cfg = self._config
        if cfg.ac_space_type == "normal":
            return ac * cfg.action_range
        if cfg.ac_space_type == "piecewise":
            om = cfg.omega
            return np.where(np.abs(ac) < om, ac / (om / ac_scale),
                            np.sign(ac) * (ac_scale + (cfg.action_range - ac_scale) * ((np.abs(ac) - om) / (1 - om))))
        raise NotImplementedError

    # rl/sac_agent.py:177-196
    def invert_displacement(self, displacement, ac_scale):
        cfg = self._config
        if cfg.ac_space_type == "normal":
            return displacement / cfg.action_range
        if cfg.ac_space_type == "piecewise":
            om = cfg.omega
            return np.where(np.abs(displacement) < ac_scale, displacement * (om / ac_scale),
                            np.sign(displacement) * ((np.abs(displacement) - ac_scale)
                                                     / ((cfg.action_range - ac_scale) / (1.0 - ac_scale))
                                                     / ((1.0 - ac_scale) / (1.0 - om)) + om))
        raise NotImplementedError

    # rl/sac_agent.py:237-260
    def clip_qpos(self, curr_qpos):
        tmp_pos = curr_qpos.copy()
        lim = self._is_jnt_limited[self._jnt_indices]
        lo, hi = self._jnt_minimum[self._jnt_indices], self._jnt_maximum[self._jnt_indices]
        if np.any(curr_qpos[lim] < lo[lim]) or np.any(curr_qpos[lim] > hi[lim]):
            new = np.clip(curr_qpos.copy(), lo + self._config.joint_margin, hi - self._config.joint_margin)
            new[np.invert(lim)] = tmp_pos[np.invert(lim)]
            curr_qpos = new
        return curr_qpos

    # rl/sac_agent.py:262-318
    def simple_interpolate(self, curr_qpos, target_qpos, ac_scale, use_planner=False):
        success, exact = True, True
        curr_qpos = self.clip_qpos(curr_qpos)
        traj = []
        n = len(self._ref_joint_pos_indexes)
        min_action = self._ac_low * ac_scale * 0.8
        max_action = self._ac_high * ac_scale * 0.8
        assert max_action > min_action, "action space box is ill defined"
        assert max_action > 0 and min_action < 0, "action space MAY be ill defined. Check this assertion"
        diff = target_qpos[:n] - curr_qpos[:n]
        out = np.where((diff > max_action) | (diff < min_action))[0]
        out_diff = diff[out]
        scales = np.where(out_diff > max_action, out_diff / max_action, out_diff / min_action)
        scaling_factor = 1.0 if len(scales) == 0 else max(max(scales), 1.0)
        scaled_ac = diff[:n] / scaling_factor
        valid = True
        interp_qpos = curr_qpos.copy()
        for _ in range(int(scaling_factor)):
            interp_qpos[:n] += scaled_ac
            if not self._planner.isValidState(interp_qpos):
                valid = False
                break
            traj.append(interp_qpos.copy())
        if not valid and use_planner:
            traj, success, valid, exact = self._simple_planner.plan(curr_qpos, target_qpos,
                                                                    self._config.simple_planner_timelimit)
            if not success:
                traj, success, valid, exact = self._planner.plan(curr_qpos, target_qpos, self._config.timelimit)
                if not success:
                    traj = [target_qpos]
                    success, exact = False, False
        else:
            if not valid:
                success, exact = False, False
            traj.append(target_qpos)
        return np.array(traj), success, valid, exact

    # rl/sac_agent.py:198-235
    def plan(self, curr_qpos, target_qpos, ac_scale=None):
        curr_qpos = self.clip_qpos(curr_qpos)
        interpolation = True
        traj, success, valid, exact = self.simple_interpolate(curr_qpos, target_qpos, ac_scale)
        if not success:
            if not exact:
                traj, success, valid, exact = self._planner.plan(curr_qpos, target_qpos, self._config.timelimit)
                interpolation = False
                if self._config.interpolation and success:
                    n = len(self._ref_joint_pos_indexes)
                    new_traj = []
                    start = curr_qpos
                    for i in range(len(traj)):
                        diff = traj[i] - start
                        if np.any(diff[:n] < -ac_scale) or np.any(diff[:n] > ac_scale):
                            inner, _, _, _ = self.simple_interpolate(start, traj[i], ac_scale, use_planner=True)
                            new_traj.extend(inner)
                        else:
                            new_traj.append(traj[i])
                        start = traj[i]
                    traj = np.array(new_traj)
        return traj, success, interpolation, valid, exact


def clip_target_to_limits(target_qpos, jnt_minimum, jnt_maximum, is_jnt_limited):
    """rl/mopa_rollouts.py:119-130: clip the planner target to the joint limits, unlimited joints untouched."""
    tmp = target_qpos.copy()
    out = np.clip(target_qpos, jnt_minimum, jnt_maximum)
    out[np.invert(is_jnt_limited)] = tmp[np.invert(is_jnt_limited)]
    return out


def norm_seq(d):
    """Euclidean norm with the squares summed left to right.  The reference calls np.linalg.norm (BLAS dot: the
    summation order, hence the last bit, depends on the BLAS build); this fixed order is what the scalar and the
    batched back-off share so that they agree bit for bit."""
    acc = 0.0
    for x in np.asarray(d, dtype=np.float64).ravel():
        acc = acc + x * x
    return float(np.sqrt(acc))


def handle_invalid_target(pi, curr_qpos, target_qpos, step_size: float, num_trials: int):
    """rl/mopa_rollouts.py:133-143: walk an invalid target back toward the current state in steps of
    `step_size` (Euclidean, over the full qpos vector) until it is valid or `num_trials` is exhausted.
    Returns (target_qpos, n_trials)."""
    target_qpos = target_qpos.copy()
    trial = 0
    if not pi.isValidState(target_qpos):
        while not pi.isValidState(target_qpos) and trial < num_trials:
            d = curr_qpos - target_qpos
            target_qpos += step_size * d / norm_seq(d)
            trial += 1
    return target_qpos, trial


# ---------------------------------------------------------------------------
# batched, device-resident forms (torch tensors on the GPU; validity through BatchPlanner)
# ---------------------------------------------------------------------------
def simple_interpolate_batch(bp, curr_qpos, target_qpos, ac_scale: float, ref_idx: Sequence[int], ac_low: float = -1.0,
                             ac_high: float = 1.0, max_steps: int = 64):
    """`simple_interpolate` (use_planner=False) for E envs at once.

    curr_qpos / target_qpos: [E, nq] float64 CUDA tensors (curr already clipped).  Returns
    (traj [E, max_steps+1, nq], traj_len [E], success [E] bool, n_steps [E]) where row e holds the
    int(scaling_factor_e) interpolated states followed by the exact target -- the reference's `traj` -- and
    success[e] is False when one of its interpolated states is invalid (then traj_len[e] counts the valid prefix
    + the target, exactly as the reference returns it).  All E * max(n_steps) states are checked in ONE launch."""
    import torch
    E, nq = curr_qpos.shape
    idx = torch.as_tensor(list(ref_idx), device=curr_qpos.device)
    n = len(idx)
    assert list(ref_idx) == list(range(n)), "the reference slices qpos[:n] (rl/sac_agent.py:275-278)"
    min_action, max_action = ac_low * ac_scale * 0.8, ac_high * ac_scale * 0.8
    diff = target_qpos[:, :n] - curr_qpos[:, :n]
    # NB: `tensor / python_float` is evaluated as a multiplication by the reciprocal on the GPU (1 ulp off);
    # dividing by a tensor keeps IEEE division, i.e. bit-identical waypoints to the numpy reference
    t_max = torch.full_like(diff, max_action)
    t_min = torch.full_like(diff, min_action)
    scale_pos = torch.where(diff > max_action, diff / t_max, torch.zeros_like(diff))
    scale_neg = torch.where(diff < min_action, diff / t_min, torch.zeros_like(diff))
    scaling = torch.clamp(torch.maximum(scale_pos, scale_neg).amax(dim=1), min=1.0)
    n_steps = scaling.to(torch.int64)                     # int() truncation, as the reference
    if int(n_steps.max().item()) > max_steps:
        raise ValueError(f"interpolation needs {int(n_steps.max().item())} steps > max_steps={max_steps}")
    K = max(1, int(n_steps.max().item()))
    scaled = diff / scaling[:, None]
    # the reference accumulates `interp_qpos += scaled_ac` step by step: same order of additions => same rounding
    acc = curr_qpos[:, :n].clone()
    rows = []
    for _ in range(K):
        acc = acc + scaled
        rows.append(acc)
    steps = torch.stack(rows, dim=1)
    q_active = steps.reshape(E * K, n).contiguous()
    if n < bp.na:
        raise ValueError("planner has more active joints than the interpolated ones")
    valid = bp.is_valid(q_active, curr_qpos.contiguous(), samples_per_env=K).reshape(E, K).bool()
    k_idx = torch.arange(K, device=valid.device)[None, :]
    in_range = k_idx < n_steps[:, None]
    bad = (~valid) & in_range
    first_bad = torch.where(bad.any(dim=1), torch.argmax(bad.int(), dim=1), n_steps)
    success = ~bad.any(dim=1)
    n_keep = torch.minimum(first_bad, n_steps)
    traj = curr_qpos[:, None, :].repeat(1, K + 1, 1)
    traj[:, :K, :n] = steps
    # the exact target goes right after the kept prefix
    traj[torch.arange(E, device=traj.device), n_keep] = target_qpos
    return traj, n_keep + 1, success, n_steps


def handle_invalid_target_batch(bp, curr_qpos, target_qpos, step_size: float, num_trials: int):
    """`handle_invalid_target` for E envs at once: every iteration moves the still-invalid targets one step and
    re-checks all of them with one launch.  Returns (target_qpos [E, nq], n_trials [E], valid [E] bool)."""
    import torch
    target = target_qpos.clone()
    E, nq = target.shape
    na = bp.na
    act = torch.as_tensor(bp.scene.active_idx.astype("int64"), device=target.device)

    def check(t):
        # the target's own passive entries are part of the state being validated
        return bp.is_valid(t[:, act].contiguous(), t.contiguous(), samples_per_env=1).bool()

    valid = check(target)
    trials = torch.zeros(E, dtype=torch.int64, device=target.device)
    # Columns in which no env's target differs from its current state stay that way (their step is 0 / norm = 0), and
    # their squares are +0.0, which leaves a non-negative running sum bit-for-bit unchanged: the sequential sum below
    # visits only the other columns (the arm joints), in the same left-to-right order as norm_seq.
    cols = torch.nonzero(((curr_qpos - target) != 0).any(dim=0)).flatten().tolist()
    for _ in range(num_trials):
        todo = ~valid
        if not bool(todo.any().item()):
            break
        d = curr_qpos - target
        sq = d * d
        acc = torch.zeros_like(sq[:, 0])
        for c in cols:
            acc = acc + sq[:, c]
        step = step_size * d / torch.sqrt(acc)[:, None]
        target = torch.where(todo[:, None], target + step, target)
        trials += todo.to(torch.int64)
        valid = torch.where(todo, check(target), valid)
    return target, trials, valid
