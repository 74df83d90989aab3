"""Batched rollout step of MoPA-RL: the body of `MoPARolloutRunner.run`'s inner loop
(reference rl/mopa_rollouts.py:70-375) for E envs at once, on the batched pieces of this package.

One call of :meth:`BatchMoPARollout.agent_step` is one iteration of the reference's `while not done` loop for every
env: the policy action either is executed directly (`env.step(ac / omega)`, :336-356) or -- some |ac_j| > omega
(`is_planner_ac`, rl/sac_agent.py:148-153) -- is turned into a joint-space target (:116-131), pulled back while
invalid (:133-143), planned to (`SACAgent.plan`, rl/sac_agent.py:198-235: straight-line pre-check, then RRT-Connect,
then densification of the planner path) and executed waypoint by waypoint with `env.step(.., is_planner=True)` while
the SMDP reward `sum_i gamma^i r_i` and `intra_steps` are accumulated (:152-199); a failed plan costs one env step with
the current reward (:303-334).  Counters `mp / rl / interpolation / mp_fail / approximate / invalid` are kept per env.

Action spaces: joint-space MoPA-SAC, and MoPA + IK (`use_ik_target`: Cartesian displacement + rotation quaternion of the
ik_target site, turned into a joint displacement by the batched damped-LS IK -- BASELINE config 5); `discrete_action` is not
offered.  Envs: the three Sawyer
obstacle envs (no unlimited joints; 7 arm entries per action, Lift adds the gripper entry, which a planner step applies at
the last waypoint of its path, :163-167).  The `reuse_data` relabelling (:204-300) -- extra transitions between random
pairs of waypoints of an executed path -- is `reuse_transitions()` below, fed by `agent_step(..., record=True)`.
The env is the KINEMATIC one (kinematic_env.py) -- not dynamics parity.

Where the work runs: every validity check (targets, pull-back, interpolated states, densification) and every
RRT-Connect query is one batched GPU launch over all envs that need it; env steps are one K4 launch per waypoint index.
The ragged bookkeeping of planner paths (a minority of envs per step) is done on the host in numpy with the same
arithmetic as the scalar code, so that results equal the per-env reference loop bit for bit (tests/test_gpu_rollout.py).
RNG streams: the RRT-Connect query of env e at agent step t uses (seed + t, stream e); the fallback planners inside the
densification use streams E + e (simple planner) and 2E + e (main planner).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from . import _lib
from .agent_planning import (JointLimits, action_to_displacement, displacement_to_action, interpolation_steps,
                             is_planner_action, simple_interpolate_batch)
from .batch import BatchPlanner, _torch
from .planner import ITERS_PER_SECOND
from .scene import ENV_SPECS, planner_inputs

COUNTERS = ("mp", "rl", "interpolation", "mp_fail", "approximate", "invalid")


@dataclass
class RolloutConfig:
    """Defaults restated from the reference: config/__init__.py:24-110 (mopa section), config/sawyer.py:76-112,
    config/motion_planner.py:4-70."""
    omega: float = 0.7
    ac_space_type: str = "piecewise"
    action_range: float = 0.5
    ac_scale: float = 0.05
    invalid_target_handling: bool = True
    num_trials: int = 100
    step_size: float = 0.02
    interpolation: bool = True
    discount_factor: float = 0.99
    timelimit: float = 1.0
    simple_planner_timelimit: float = 0.05
    range: float = 0.1
    simple_planner_range: float = 0.05
    joint_margin: float = 0.001
    contact_threshold: float = -0.002
    max_nodes: int = 1024
    max_path: int = 256
    seed: int = 1234
    # MoPA + IK action space (config/__init__.py --use_ik_target / --ik_target; rl/trainer.py:93-125): the policy outputs a
    # Cartesian displacement of the ik_target site (3) + a rotation quaternion (4) [+ the gripper entry]
    use_ik_target: bool = False
    ik_target: str = "grip_site"
    min_world_size: tuple = (-1.2, -1.2, 0.0)        # env/sawyer/sawyer.py:52-53
    max_world_size: tuple = (1.2, 1.2, 2.0)


def reuse_transitions(out, cfg, n_arm: int, rng, max_reuse_data: int = 30):
    """The `reuse_data` relabelling of rl/mopa_rollouts.py:204-300 on the record of one `agent_step(..., record=True)`:
    for every env that executed a planner path with more than 3 waypoints, up to min(len, max_reuse_data) random
    (start, goal) waypoint pairs become extra transitions  ob_list[start] --inverse-displacement action--> ob_list[goal]
    with reward (meta_rew[goal] - meta_rew[start]) * gamma^-(start+1), done = done_list[goal],
    intra_steps = goal - start - 1, kept only if the relabelled action is a planner action inside [-1, 1].
    `rng`: a numpy RandomState-like object (`randint(low, high)`) shared by all envs, or a callable env -> such an object
    (the reference draws from the global np.random, one env per process).
    Returns a list of dicts (env, start, goal, ob, ac, rew, done, intra_steps, ob_next) of numpy values."""
    rec = out["record"]
    ob, mr, dn, wp = (rec[k].cpu().numpy() for k in ("ob", "meta_rew", "done", "waypoint"))
    nexec = rec["n_exec"].cpu().numpy()
    extra = []
    for e in np.where(nexec > 3)[0]:
        draw = rng(int(e)) if callable(rng) else rng
        L = int(nexec[e])
        seen = set()
        for _ in range(min(L, max_reuse_data)):
            start = draw.randint(low=0, high=L - 1)
            if start + 1 > L - 1:
                continue
            goal = draw.randint(low=start + 1, high=L)
            if (start, goal) in seen:
                continue
            seen.add((start, goal))
            # env.form_action(traj[goal], traj[start]) -> pi.invert_displacement
            ac = displacement_to_action(wp[e, goal, :n_arm] - wp[e, start, :n_arm], cfg.ac_scale, cfg.omega, cfg.action_range,
                                        cfg.ac_space_type)
            is_planner = bool(np.any(ac < -cfg.omega) or np.any(ac > cfg.omega))
            in_box = bool(np.all(ac >= -1.0) and np.all(ac <= 1.0))
            if not (is_planner and in_box):
                continue
            rew = (mr[e, goal] - mr[e, start]) * cfg.discount_factor ** (-(start + 1))
            extra.append({"env": int(e), "start": start, "goal": goal, "ob": ob[e, start], "ac": ac, "rew": float(rew),
                          "done": int(dn[e, goal]), "intra_steps": goal - start - 1, "ob_next": ob[e, goal]})
    return extra


class BatchMoPARollout:
    def __init__(self, env, cfg: Optional[RolloutConfig] = None):
        torch = _torch()
        self.env = env
        self.cfg = cfg if cfg is not None else RolloutConfig()
        if abs(env.ac_scale - self.cfg.ac_scale) > 0:
            raise _lib.MopaError("env.ac_scale and RolloutConfig.ac_scale differ")
        spec = ENV_SPECS[env.env_name]
        pi = planner_inputs(env.env_name, env.model)
        if len(pi.non_limited_idx):
            raise NotImplementedError("unlimited joints (3.14 wrap of SamplingBasedPlanner) are not handled by the batched rollout")
        self.pi = pi
        dev_index = env.device.index if env.device.index is not None else -1
        mk = lambda r: _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, self.cfg.contact_threshold, range_=r,
                                  seed=self.cfg.seed, device=dev_index)
        self.scene, self.simple_scene = mk(self.cfg.range), mk(self.cfg.simple_planner_range)
        self.bp = BatchPlanner(self.scene)
        self.E, self.nq, self.n = env.E, env.nq, env.n_arm
        self.arm = list(int(i) for i in env.facts.arm_qpos_idx)
        assert self.arm == list(range(self.n)), "the reference slices qpos[:n] (rl/sac_agent.py:275-278)"
        f = env.facts
        dev, f64 = env.device, torch.float64
        self.limits = JointLimits(f.qpos_min, f.qpos_max, f.qpos_limited, self.cfg.joint_margin, device=dev)
        self.counters: Dict[str, "object"] = {k: torch.zeros(self.E, dtype=torch.int64, device=dev) for k in COUNTERS}
        self.t = 0     # agent steps taken (part of the RNG stream of the planner queries)
        self.main_iters = max(1, int(round(self.cfg.timelimit * ITERS_PER_SECOND)))
        self.simple_iters = max(1, int(round(self.cfg.simple_planner_timelimit * ITERS_PER_SECOND)))
        self.ik = None
        if self.cfg.use_ik_target:
            from .ik import BatchIK
            self.ik = BatchIK(env.model, self.cfg.ik_target, spec.robot_joints, device=dev_index)
            self._world_lo = torch.tensor(self.cfg.min_world_size, dtype=f64, device=dev)
            self._world_hi = torch.tensor(self.cfg.max_world_size, dtype=f64, device=dev)
            # `_cart2dispalcement` clips the IK result against env._jnt_minimum/_jnt_maximum[jnt_indices] (float64; unlimited
            # joints carry +-3.14 there, env/base.py:85-86) -- only the arm entries are read back
            self._ik_lo = torch.tensor(f.qpos_min[self.arm], dtype=f64, device=dev)
            self._ik_hi = torch.tensor(f.qpos_max[self.arm], dtype=f64, device=dev)
        self.ac_dim = (7 if self.cfg.use_ik_target else self.n) + (env.action_dim - self.n)

    # ------------------------------------------------------------------
    def close(self):
        self.scene.close()
        self.simple_scene.close()

    def clip_qpos(self, q):
        """`SACAgent.clip_qpos` (rl/sac_agent.py:237-260) per row, with the float32 limits the agents hold."""
        return self.limits.clip_state(q)

    def ik_displacement(self, ac, cur):
        """`MoPARolloutRunner._cart2dispalcement` (rl/mopa_rollouts.py:87-99,681-728) for E envs: the policy's Cartesian action
        -> joint displacement of the arm through the damped-LS IK (K5).
            target_cart = clip(site_xpos + action_range * ac[:3], world box)
            target_quat = mulQuat(q_site[(w, x, y, y)], ac[3:7] / |ac[3:7]|)      # the reference indexes [3, 0, 1, 1] (sic)
            qpos_from_site_pose(ik_env, ik_target, target_cart, target_quat, robot_joints, max_steps=100, tol=1e-2)
            displacement = clip(result, joint limits)[arm] - curr[arm]
        q_site is the site's orientation as `util.env.mat2quat` returns it: from the float32-rounded rotation matrix, sign
        w >= 0.  (The reference takes the dominant eigenvector of the 4 x 4 K-matrix; for a rotation matrix that is the
        closed-form quaternion used here, to the float32 rounding of the matrix -- a 1e-7 effect on the target.)"""
        torch = _torch()
        cfg = self.cfg
        site_pos, site_mat = self.ik.site_pose(cur.contiguous())
        target_cart = torch.minimum(torch.maximum(site_pos + cfg.action_range * ac[:, :3], self._world_lo), self._world_hi).contiguous()
        m = site_mat.to(torch.float32).to(torch.float64)                      # np.array(rmat, dtype=np.float32)
        m00, m01, m02, m10, m11, m12, m20, m21, m22 = (m[:, i, j] for i in range(3) for j in range(3))
        tr = m00 + m11 + m22
        # closed-form quaternion, branch on the largest of (trace, m00, m11, m22) for conditioning
        qw = torch.stack([1.0 + tr, m21 - m12, m02 - m20, m10 - m01], dim=1)
        qx = torch.stack([m21 - m12, 1.0 + m00 - m11 - m22, m01 + m10, m02 + m20], dim=1)
        qy = torch.stack([m02 - m20, m01 + m10, 1.0 - m00 + m11 - m22, m12 + m21], dim=1)
        qz = torch.stack([m10 - m01, m02 + m20, m12 + m21, 1.0 - m00 - m11 + m22], dim=1)
        pick = torch.stack([tr, m00, m11, m22], dim=1).argmax(dim=1)
        q = torch.stack([qw, qx, qy, qz], dim=1)[torch.arange(len(m), device=m.device), pick]      # (w, x, y, z), unnormalised
        q = q / q.norm(dim=1, keepdim=True)
        q = torch.where(q[:, :1] < 0, -q, q)
        tq = torch.stack([q[:, 0], q[:, 1], q[:, 2], q[:, 2]], dim=1)                               # [[3, 0, 1, 1]] of (x, y, z, w)
        aq = ac[:, 3:7] / ac[:, 3:7].norm(dim=1, keepdim=True)
        aw, ax, ay, az = tq.unbind(1)
        bw, bx, by, bz = aq.unbind(1)
        target_quat = torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                                   aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=1).contiguous()
        q_ik = cur.clone()
        self.ik.solve(q_ik, target_cart, target_quat, max_steps=100, tol=1e-2)
        arm_t = torch.minimum(torch.maximum(q_ik[:, :self.n], self._ik_lo), self._ik_hi)
        return arm_t - cur[:, :self.n]

    def _valid(self, q):
        return self.bp.is_valid(q[:, self.arm].contiguous(), q.contiguous(), samples_per_env=1).bool()

    # ------------------------------------------------------------------
    def plan(self, cur, target, env_ids):
        """`SACAgent.plan` (rl/sac_agent.py:198-235) for M envs.  Returns (traj [M, L, nq] device tensor, length [M] device
        int64 -- 0 where the plan failed --, and the boolean numpy arrays success, interpolation, valid, exact).  Straight
        lines that validate never leave the GPU; only the envs that needed RRT-Connect are post-processed on the host
        (ragged paths: successive differences, densification) and uploaded into their rows."""
        torch = _torch()
        cfg, n = self.cfg, self.n
        M = cur.shape[0]
        cur = self.clip_qpos(cur)
        traj_t, tlen, succ, _ = simple_interpolate_batch(self.bp, cur, target, cfg.ac_scale, self.arm)
        succ_h = succ.cpu().numpy()
        success, interpolation = succ_h.copy(), np.ones(M, dtype=bool)
        valid, exact = succ_h.copy(), succ_h.copy()
        lens = torch.where(succ, tlen.to(torch.int64), torch.zeros_like(tlen, dtype=torch.int64))
        fail = np.where(~succ_h)[0]
        if len(fail) == 0:
            return traj_t, lens, success, interpolation, valid, exact
        # ---- main planner for the envs whose straight line is blocked (:205-209)
        fi = torch.as_tensor(fail, device=cur.device)
        ids = env_ids[fi].contiguous()
        cur_f = cur[fi].contiguous()
        path, plen, status, _ = self.bp.plan(cur_f, target[fi].contiguous(), max_iters=self.main_iters,
                                             max_nodes=cfg.max_nodes, max_path=cfg.max_path, seed=cfg.seed + self.t, env_ids=ids)
        plen_h, st_h = plen.cpu().numpy(), status.cpu().numpy()
        path_h = path[:, :max(1, int(plen_h.max()))].cpu().numpy()      # the [max_path] tail of every row is unused
        cur_h, ids_h = cur_f.cpu().numpy(), ids.cpu().numpy()
        interpolation[fail] = False
        trajs = {}             # j (index into `fail`) -> [L_j, nq] numpy
        seg_jobs = []          # (j, i, start, end) of planner-path segments that need densification
        bad = st_h != 0        # sentinel rows (sampling_based_planner.py:64-69): -5 goal invalid, -4 no exact solution
        valid[fail[bad]] = st_h[bad] != _lib.PLAN_INVALID_GOAL
        exact[fail[bad]] = st_h[bad] != _lib.PLAN_NO_EXACT
        success[fail[bad]] = False
        good = np.where(~bad)[0]
        if len(good):
            success[fail[good]] = valid[fail[good]] = exact[fail[good]] = True
            # SamplingBasedPlanner.plan rebuilds the trajectory from successive differences (:71-99) and PlannerAgent
            # drops row 0: tr[k] = tr[k-1] + (states[k] - states[k-1]), tr[0] = cur.  np.add.accumulate is that same
            # strictly sequential sum, for all paths at once (rows past a path's length hold garbage and are cut off).
            P = path_h[good]
            A = np.concatenate([cur_h[good][:, None, :], P[:, 1:] - P[:, :-1]], axis=1)
            T = np.add.accumulate(A, axis=1)
            nrow = plen_h[good] - 1
            if cfg.interpolation:
                step = T[:, 1:, :n] - T[:, :-1, :n]                      # waypoint i minus its predecessor (cur for i = 0)
                far = ((step < -cfg.ac_scale) | (step > cfg.ac_scale)).any(axis=2)
                far &= np.arange(far.shape[1])[None, :] < nrow[:, None]
            for r, j in enumerate(good):
                trajs[j] = T[r, 1:nrow[r] + 1].copy()
                if cfg.interpolation:
                    for i in np.nonzero(far[r])[0]:
                        seg_jobs.append((j, int(i), T[r, i].copy(), trajs[j][i]))
        if seg_jobs:
            self._densify(trajs, seg_jobs, cur_h, ids_h)
        if trajs:
            L = max(traj_t.shape[1], max(len(tr) for tr in trajs.values()))
            if L > traj_t.shape[1]:
                traj_t = torch.cat([traj_t, torch.zeros(M, L - traj_t.shape[1], self.nq, dtype=traj_t.dtype, device=traj_t.device)], dim=1)
            js = sorted(trajs)
            pad = np.zeros((len(js), L, self.nq))
            ln = np.zeros(len(js), dtype=np.int64)
            for r, j in enumerate(js):
                pad[r, :len(trajs[j])] = trajs[j]
                ln[r] = len(trajs[j])
            rows = torch.as_tensor(fail[js], device=cur.device)
            traj_t[rows] = torch.as_tensor(pad, device=cur.device)
            lens[rows] = torch.as_tensor(ln, device=cur.device)
        return traj_t, lens, success, interpolation, valid, exact

    def _densify(self, trajs, seg_jobs, cur_h, ids_h):
        """rl/sac_agent.py:216-233 -- a planner-path segment longer than ac_scale in some joint is replaced by the
        straight-line rule (steps <= 0.8 ac_scale from the segment's clipped start, then the segment's end).  All segments
        of all envs are cut at once and all their interior states validated in ONE launch; the (rare) segments with an
        invalid interior state fall back to the single-query planners: simple planner, main planner, else [end]
        (:300-313)."""
        torch = _torch()
        cfg, n, nq = self.cfg, self.n, self.nq
        S = len(seg_jobs)
        starts = np.stack([self.limits.clip_state_np(j[2]) for j in seg_jobs])      # :266 clip_qpos on every segment start
        ends = np.stack([j[3] for j in seg_jobs])
        bound = cfg.ac_scale * 0.8
        diff = ends[:, :n] - starts[:, :n]
        ratio = np.maximum(np.where(diff > bound, diff / bound, 0.0), np.where(diff < -bound, diff / -bound, 0.0))
        scale = np.maximum(ratio.max(axis=1), 1.0)
        count = scale.astype(np.int64)                                              # int(): truncation
        per_step = diff / scale[:, None]
        K = int(count.max())
        walk = np.repeat(starts[:, None, :], K, axis=1)
        acc = starts[:, :n].copy()
        for k in range(K):                                                          # K dependent additions, as the scalar rule
            acc = acc + per_step
            walk[:, k, :n] = acc
        live = np.arange(K)[None, :] < count[:, None]
        verdict = np.ones((S, K), dtype=bool)
        verdict[live] = self._valid(torch.tensor(walk[live], device=self.env.device)).cpu().numpy()
        clear = verdict.all(axis=1)
        replacement = {}
        for k, (m, i, _, end) in enumerate(seg_jobs):
            if clear[k]:
                replacement[(m, i)] = list(walk[k, :count[k]]) + [end]
                continue
            e = int(ids_h[m])
            found = None
            for scene, iters, base in ((self.simple_scene, self.simple_iters, 1), (self.scene, self.main_iters, 2)):
                st, p, _ = scene.plan(starts[k], end, max_iters=iters, max_nodes=cfg.max_nodes, max_path=cfg.max_path,
                                      seed=cfg.seed + self.t, env_id=self.E * base + e)
                if st == 0:
                    # SamplingBasedPlanner.plan: start + running sum of successive differences; PlannerAgent drops row 0
                    found = list(np.add.accumulate(np.vstack([starts[k][None], p[1:] - p[:-1]]), axis=0)[1:])
                    break
            replacement[(m, i)] = found if found is not None else [end]
        for m in sorted({j[0] for j in seg_jobs}):
            trajs[m] = np.array([row for i in range(len(trajs[m])) for row in replacement.get((m, i), [trajs[m][i]])])

    # ------------------------------------------------------------------
    def agent_step(self, ac, record: bool = False):
        """One agent step for all E envs.  ac: float64 [E, >= env.action_dim] GPU tensor (policy output in [-1, 1]).
        Returns a dict of GPU tensors: ob [E,obs_dim] (before), ob_next [E,obs_dim], rew [E] (SMDP return of the step), done [E]
        uint8, intra_steps [E] int64, is_planner [E] bool, success [E] bool (env success flag), plus `path_len`.
        record=True adds `record`: per executed waypoint k the obs after it, the running SMDP return, the done flag and
        the waypoint itself ([E, L, ...]; `n_exec` [E] = waypoints actually executed) -- the `ob_list / meta_rew_list /
        done_list / traj` of the reference, input of `reuse_transitions`."""
        torch = _torch()
        env, cfg, E, n = self.env, self.cfg, self.E, self.n
        dev = env.device
        tm = getattr(self, "timing", None)     # optional dict: phase -> seconds (each mark synchronises; profiling only)
        if tm is not None:
            import time as _time
            torch.cuda.synchronize()
            _t = [_time.perf_counter()]

            def mark(name):
                torch.cuda.synchronize()
                now = _time.perf_counter()
                tm[name] = tm.get(name, 0.0) + now - _t[0]
                _t[0] = now
        else:
            def mark(name):
                pass
        prev_ob = env.obs.clone()
        cur = env.qpos.clone()
        ar = torch.arange(E, device=dev)
        if cfg.use_ik_target:
            # MoPA + IK: the action is Cartesian; its joint displacement decides planner / direct and IS the direct action
            a = self.ik_displacement(ac, cur)
            extra_ac = ac[:, 7:7 + (env.action_dim - n)]
            mark("ik")
        else:
            a = ac[:, :n].contiguous()
            extra_ac = ac[:, n:env.action_dim]
        is_pl = is_planner_action(a, cfg.omega)
        plan_ok = torch.zeros(E, dtype=torch.bool, device=dev)
        path_len = torch.zeros(E, dtype=torch.int64, device=dev)
        traj_pad = None
        pl_idx = torch.nonzero(is_pl).flatten()
        if len(pl_idx):
            target = cur[pl_idx].clone()
            if not cfg.use_ik_target:
                disp = action_to_displacement(a[pl_idx], cfg.ac_scale, cfg.omega, cfg.action_range, cfg.ac_space_type)
                target[:, :n] += disp
                # np.clip to the joint limits, unlimited entries restored (:121-131)
                target = self.limits.clip_target(target)
            # (with use_ik_target the reference never moves target_qpos off curr_qpos -- :113-131 is skipped and
            # `_cart2dispalcement` keeps its result local --, so a planner step plans from the current state to itself: two
            # zero-motion env steps.  Reproduced as is.)
            if cfg.invalid_target_handling:
                target, _, tv = self.bp.pullback(cur[pl_idx].contiguous(), target.contiguous(), cfg.step_size, cfg.num_trials)
                tv = tv.bool()
            else:
                tv = self._valid(target)
            mark("target")
            v_idx = torch.nonzero(tv).flatten()
            self.counters["mp_fail"][pl_idx[~tv]] += 1          # invalid target: success, valid, exact = False, False, True
            self.counters["invalid"][pl_idx[~tv]] += 1
            if len(v_idx):
                ids = pl_idx[v_idx].contiguous()
                traj_dev, lens_dev, success, interpolation, valid, exact = self.plan(cur[ids].contiguous(), target[v_idx].contiguous(), ids)
                mark("plan")
                t = lambda x: torch.as_tensor(x, device=dev)
                s_t = t(success)
                plan_ok[ids] = s_t
                self.counters["interpolation"][ids[s_t & t(interpolation)]] += 1
                self.counters["mp"][ids[s_t & ~t(interpolation)]] += 1
                self.counters["mp_fail"][ids[~s_t]] += 1
                self.counters["approximate"][ids[~s_t & ~t(exact)]] += 1
                self.counters["invalid"][ids[~s_t & ~t(valid)]] += 1
                L = int(lens_dev.max().item())
                if L:
                    traj_pad = torch.zeros(E, L, self.nq, dtype=torch.float64, device=dev)
                    traj_pad[ids] = traj_dev[:, :L]
                    path_len[ids] = lens_dev
        mark("pad")
        direct = ~is_pl
        self.counters["rl"][direct] += 1
        # ---- direct execution (:336-356) and failed plans (:303-334: reward of the current state, one env step) in one launch
        act0 = torch.where(direct[:, None], a / torch.full_like(a, cfg.omega), torch.zeros_like(a))
        if env.action_dim > n:          # Lift: the gripper entry is passed through unscaled (:340-343)
            act0 = torch.cat([act0, torch.where(direct[:, None], extra_ac, torch.zeros_like(extra_ac))], dim=1)
        act0 = act0.contiguous()
        flags = torch.where(direct, 1, torch.where(plan_ok, 2, 0)).to(torch.uint8).contiguous()
        env._launch(act0, False, flags)
        rew = torch.where(plan_ok, torch.zeros_like(env.reward), env.reward)
        done = torch.where(plan_ok, torch.zeros_like(env.done), env.done)
        intra = torch.zeros(E, dtype=torch.int64, device=dev)
        # ---- waypoint execution (:152-199)
        rec = None
        if record:
            Lr = traj_pad.shape[1] if traj_pad is not None else 0
            rec = {"ob": torch.zeros(E, Lr, env.obs.shape[1], dtype=torch.float64, device=dev),
                   "meta_rew": torch.zeros(E, Lr, dtype=torch.float64, device=dev),
                   "done": torch.zeros(E, Lr, dtype=torch.uint8, device=dev),
                   "waypoint": traj_pad if traj_pad is not None else torch.zeros(E, 0, self.nq, dtype=torch.float64, device=dev),
                   "n_exec": torch.zeros(E, dtype=torch.int64, device=dev)}
        if traj_pad is not None:
            # one launch: every env walks its own waypoints until its path ends or a step reports done
            L = traj_pad.shape[1]
            disc = torch.tensor([cfg.discount_factor ** k for k in range(L)], dtype=torch.float64, device=dev)
            rew, done = rew.contiguous(), done.to(torch.uint8).contiguous()
            env.exec_trajectories(traj_pad, torch.where(plan_ok, path_len, torch.zeros_like(path_len)).contiguous(), disc,
                                  rew, done, intra, rec={k: rec[k] for k in ("ob", "meta_rew", "done", "n_exec")} if rec else None,
                                  last_extra=extra_ac[:, 0].contiguous() if env.action_dim > n else None)
        mark("execute")
        env.has_prev.zero_()                                     # env._reset_prev_state()
        self.t += 1
        del ar
        res = {"ob": prev_ob, "ob_next": env.obs.clone(), "rew": rew, "done": done, "intra_steps": intra, "is_planner": is_pl,
               "success": env.success.clone(), "path_len": path_len, "plan_ok": plan_ok}
        if rec is not None:
            res["record"] = rec
        return res
