"""Batched forms of what the reference's agents and rollout runner do AROUND the planner (SURVEY.md section 8 rows
A10-A11), for E environments at once on torch tensors:

    is_planner_action          rl/sac_agent.py:148-153      some |a_j| beyond omega -> the step is a planner step
    action_to_displacement     rl/sac_agent.py:158-175      policy output -> joint displacement (piecewise / normal)
    displacement_to_action     rl/sac_agent.py:177-196      its inverse (used by the reuse_data relabelling)
    JointLimits.clip_state     rl/sac_agent.py:237-260      a state beyond a joint limit is pulled inside by joint_margin
    JointLimits.clip_target    rl/mopa_rollouts.py:119-131  a planner target is clipped onto the limits
    simple_interpolate_batch   rl/sac_agent.py:262-318      straight line in steps <= 0.8 ac_scale, every step validated
    handle_invalid_target_batch rl/mopa_rollouts.py:133-143 invalid target walked back toward the current state

The reference keeps these as per-environment numpy methods on its SAC/TD3 agents; its agents keep using their own.  The
forms here exist for the batched rollout (rollout.py): one validity launch covers all environments, every arithmetic
step is ordered as in the reference so results agree with it element for element.  They are checked against vectors
produced by the reference's own code (tests/golden/ref_py_*.npz, tools/gen_ref_py_golden.py).

Tensors may live on the GPU (product) or on the CPU (the host tests drive the same code with a CPU validity checker);
validity always comes from the `bp` object passed in (`BatchPlanner`: libmopa_hip.so).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


def _torch():
    import torch
    return torch


def is_planner_action(ac, omega: float):
    """[E, n] -> [E] bool: any joint entry outside [-omega, omega]."""
    return ((ac < -omega) | (ac > omega)).any(dim=-1)


def action_to_displacement(ac, ac_scale: float, omega: float, action_range: float, ac_space_type: str = "piecewise"):
    """Policy output in [-1, 1] -> joint displacement.  Inside omega the map is linear onto [-ac_scale, ac_scale]; beyond
    it the remaining interval is stretched onto (ac_scale, action_range].  Divisions are by tensors: `tensor /
    python_float` runs as a multiplication by the reciprocal on the GPU (1 ulp off IEEE division)."""
    torch = _torch()
    if ac_space_type == "normal":
        return ac * action_range
    if ac_space_type != "piecewise":
        raise NotImplementedError(ac_space_type)
    mag = ac.abs()
    near = ac / torch.full_like(ac, omega / ac_scale)
    beyond = (mag - omega) / torch.full_like(ac, 1 - omega)
    far = torch.sign(ac) * (ac_scale + (action_range - ac_scale) * beyond)
    return torch.where(mag < omega, near, far)


def displacement_to_action(disp, ac_scale: float, omega: float, action_range: float, ac_space_type: str = "piecewise"):
    """Inverse map on numpy arrays (host side: the relabelling works on recorded waypoints).  The far branch divides by
    (action_range - ac_scale) / (1 - ac_scale) and then by (1 - ac_scale) / (1 - omega), in that order, as the
    reference does (the two do not cancel to the algebraic inverse of `action_to_displacement`; kept as is)."""
    disp = np.asarray(disp, dtype=np.float64)
    if ac_space_type == "normal":
        return disp / action_range
    if ac_space_type != "piecewise":
        raise NotImplementedError(ac_space_type)
    mag = np.abs(disp)
    stretch = (mag - ac_scale) / ((action_range - ac_scale) / (1.0 - ac_scale)) / ((1.0 - ac_scale) / (1.0 - omega))
    return np.where(mag < ac_scale, disp * (omega / ac_scale), np.sign(disp) * (stretch + omega))


class JointLimits:
    """Per-qpos-address joint limits of a model, in the two precisions the reference uses them in.

    The env clips planner *targets* against float64 limits (`env._jnt_minimum`, env/base.py:77-89).  The agents clip
    *states* against `joint_space['default'].low/high`, a float32 gym Box (rl/sac_agent.py:56-57), and form the
    margin-shrunk bounds in float32 too (`low + joint_margin` on a float32 array stays float32).  Both are kept."""

    def __init__(self, lo, hi, limited, joint_margin: float, device=None):
        torch = _torch()
        lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
        lim = np.asarray(limited, dtype=bool)
        lo32, hi32 = lo.astype(np.float32), hi.astype(np.float32)
        shrunk_lo = (lo32 + np.float32(joint_margin)).astype(np.float64)
        shrunk_hi = (hi32 - np.float32(joint_margin)).astype(np.float64)
        t = lambda a: torch.tensor(a, dtype=torch.float64, device=device)
        inf = np.inf
        self.limited = torch.tensor(lim, device=device)
        self.lo, self.hi = t(np.where(lim, lo, -inf)), t(np.where(lim, hi, inf))
        self.lo_state, self.hi_state = t(np.where(lim, lo32.astype(np.float64), -inf)), t(np.where(lim, hi32.astype(np.float64), inf))
        self.lo_shrunk, self.hi_shrunk = t(np.where(lim, shrunk_lo, -inf)), t(np.where(lim, shrunk_hi, inf))
        self._np = tuple(x.cpu().numpy() for x in (self.lo_state, self.hi_state, self.lo_shrunk, self.hi_shrunk))

    def clip_target(self, q):
        torch = _torch()
        return torch.minimum(torch.maximum(q, self.lo), self.hi)          # unlimited entries have infinite bounds

    def clip_state(self, q):
        """Rows with some limited joint beyond its limit are clipped -- all their limited entries -- to the shrunk
        bounds; the other rows pass through untouched."""
        torch = _torch()
        beyond = ((q < self.lo_state) | (q > self.hi_state)).any(dim=-1, keepdim=True)
        return torch.where(beyond, torch.minimum(torch.maximum(q, self.lo_shrunk), self.hi_shrunk), q)


    def clip_state_np(self, q):
        """`clip_state` on host rows (numpy, [nq] or [S, nq])."""
        lo, hi, lo_s, hi_s = self._np
        beyond = ((q < lo) | (q > hi)).any(axis=-1, keepdims=True)
        return np.where(beyond, np.minimum(np.maximum(q, lo_s), hi_s), q)


def interpolation_steps(diff, ac_scale: float, ac_low: float = -1.0, ac_high: float = 1.0):
    """Step count rule of the straight-line pre-check: with the per-step bound b = 0.8 ac_scale, the line is cut into
    int(s) equal steps, s = max(1, max_j diff_j / (+-b)) over the joints that exceed the bound.  Returns (s [E], n [E])."""
    torch = _torch()
    lo_b, hi_b = ac_low * ac_scale * 0.8, ac_high * ac_scale * 0.8
    zero = torch.zeros_like(diff)
    over = torch.where(diff > hi_b, diff / torch.full_like(diff, hi_b), zero)
    under = torch.where(diff < lo_b, diff / torch.full_like(diff, lo_b), zero)
    s = torch.clamp(torch.maximum(over, under).amax(dim=1), min=1.0)
    return s, s.to(torch.int64)        # int(): truncation


def max_interpolation_steps(action_range: float, ac_scale: float) -> int:
    """Upper bound of the step count for targets produced by `action_to_displacement` (|diff_j| <= action_range)."""
    return int(np.floor(action_range / (0.8 * ac_scale) * (1 + 1e-12))) + 1


def simple_interpolate_batch(bp, curr_qpos, target_qpos, ac_scale: float, ref_idx: Sequence[int], ac_low: float = -1.0,
                             ac_high: float = 1.0, max_steps: int = 64, fixed_steps: int = 0):
    """Straight-line pre-check for E envs at once.

    curr_qpos / target_qpos: [E, nq] float64 tensors (curr already clipped by `JointLimits.clip_state`).  Returns
    (traj [E, K+1, nq], traj_len [E], success [E] bool, n_steps [E]): row e holds its int(s_e) interpolated states
    followed by the exact target, success[e] is False when one of them is invalid -- then traj_len[e] counts the valid
    prefix + the target.  All E x K states are validated in ONE launch.  K = the largest step count of the batch (one
    host read-back) or, with `fixed_steps` > 0, that constant (no read-back; rows that would need more raise later
    through `n_steps`, callers pass `max_interpolation_steps`)."""
    torch = _torch()
    E, nq = curr_qpos.shape
    n = len(ref_idx)
    if list(ref_idx) != list(range(n)):
        raise ValueError("the interpolated joints must be qpos[:n] (the reference slices qpos[:len(ref_joint_pos_indexes)])")
    if n < bp.na:
        raise ValueError("planner has more active joints than the interpolated ones")
    if (fixed_steps > 0 and ac_low == -1.0 and ac_high == 1.0 and n == bp.na and curr_qpos.is_cuda
            and getattr(getattr(bp, "scene", None), "_h", None) is not None):
        # fixed width: the whole pre-check as three launches of the library (mopa_interpolate_batch), no read-back
        from . import _lib
        from .batch import _ptr, _stream_handle
        K = int(fixed_steps)
        cur_c, tgt_c = curr_qpos.contiguous(), target_qpos.contiguous()
        traj = torch.zeros(E, K + 1, nq, dtype=torch.float64, device=cur_c.device)
        tlen = torch.empty(E, dtype=torch.int32, device=cur_c.device)
        ok = torch.empty(E, dtype=torch.uint8, device=cur_c.device)
        nst = torch.empty(E, dtype=torch.int32, device=cur_c.device)
        _lib.check(_lib.lib().mopa_interpolate_batch(bp.scene._h, E, n, K, _ptr(cur_c), _ptr(tgt_c), float(ac_scale), _ptr(traj), _ptr(tlen),
                                                     _ptr(ok), _ptr(nst), _stream_handle(None)))
        return traj, tlen.to(torch.int64), ok.bool(), nst.to(torch.int64)
    diff = target_qpos[:, :n] - curr_qpos[:, :n]
    scaling, n_steps = interpolation_steps(diff, ac_scale, ac_low, ac_high)
    if fixed_steps > 0:
        K = int(fixed_steps)
    else:
        K = max(1, int(n_steps.max().item()))
        if K > max_steps:
            raise ValueError(f"interpolation needs {K} steps > max_steps={max_steps}")
    per_step = diff / scaling[:, None]
    # running sum cur + d + d + ... as K dependent additions (k*d, or a parallel prefix sum, would round differently)
    acc, rows = curr_qpos[:, :n], []
    for _ in range(K):
        acc = acc + per_step
        rows.append(acc)
    walk = torch.stack(rows, dim=1)
    valid = bp.is_valid(walk.reshape(E * K, n).contiguous(), curr_qpos.contiguous(), samples_per_env=K).reshape(E, K).bool()
    k_idx = torch.arange(K, device=valid.device)[None, :]
    bad = (~valid) & (k_idx < n_steps[:, None])
    blocked = bad.any(dim=1)
    n_keep = torch.where(blocked, torch.argmax(bad.int(), dim=1), n_steps).clamp(max=K)
    traj = curr_qpos[:, None, :].repeat(1, K + 1, 1)
    traj[:, :K, :n] = walk
    traj[torch.arange(E, device=traj.device), n_keep] = target_qpos          # the exact target right after the kept prefix
    return traj, n_keep + 1, ~blocked, n_steps


def handle_invalid_target_batch(bp, curr_qpos, target_qpos, step_size: float, num_trials: int):
    """Back-off of invalid planner targets, host-orchestrated form (the product path is the one-launch
    `BatchPlanner.pullback`; this form is its cross-check): every iteration moves the still-invalid targets by
    step_size * d / |d|, d = current - target over the whole qpos vector, and re-validates them with one launch.
    Returns (target [E, nq], n_trials [E], valid [E] bool).  |d| sums the squares left to right over the columns in
    which some env differs from its current state (all-zero columns contribute +0.0 and cannot change a non-negative
    running sum), the order `BatchPlanner.pullback` uses."""
    torch = _torch()
    target = target_qpos.clone()
    E = target.shape[0]
    act = torch.as_tensor(np.asarray(bp.scene.active_idx).astype("int64"), device=target.device)

    def check(t):       # the target's own passive entries are part of the state being validated
        return bp.is_valid(t[:, act].contiguous(), t.contiguous(), samples_per_env=1).bool()

    valid = check(target)
    trials = torch.zeros(E, dtype=torch.int64, device=target.device)
    cols = torch.nonzero(((curr_qpos - target) != 0).any(dim=0)).flatten().tolist()
    for _ in range(num_trials):
        todo = ~valid
        if not bool(todo.any()):
            break
        d = curr_qpos - target
        sq = d * d
        acc = torch.zeros_like(sq[:, 0])
        for c in cols:
            acc = acc + sq[:, c]
        target = torch.where(todo[:, None], target + step_size * d / torch.sqrt(acc)[:, None], target)
        trials += todo.to(torch.int64)
        valid = torch.where(todo, check(target), valid)
    return target, trials, valid
