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
ik_target site, turned into a joint displacement by the batched damped-LS IK -- BASELINE config 5), and `discrete_action`
(the policy's `ac_type` head routes a step, :86-88,106-111,349).  Envs: the three Sawyer
obstacle envs (7 arm entries per action, Lift adds the gripper entry, which a planner step applies at
the last waypoint of its path, :163-167) and PusherObstacle-v0 (4 joints, joint0 UNLIMITED: query endpoints are wrapped into
(-3.14, 3.14) before planning and the returned steps un-wrapped across the seam, as `SamplingBasedPlanner.plan` does --
`wrap_unlimited`, `seam_steps_np`, `mopa_paths_unwrap_seam_batch`; `RolloutConfig.for_env` carries config/pusher.py).  The `reuse_data` relabelling (:204-300) -- extra transitions between random
pairs of waypoints of an executed path -- is `reuse_transitions()` below, fed by `agent_step(..., record=True)`.
The env is the KINEMATIC one (kinematic_env.py) -- not dynamics parity.

Where the work runs: every validity check (targets, pull-back, interpolated states, densification) and every
RRT-Connect query is one batched GPU launch over all envs that need it; env steps are one K4 launch per waypoint index.
The ragged bookkeeping of planner paths (a minority of envs per step) is done on the host in numpy with the same
arithmetic as the scalar code, so that results equal the per-env reference loop bit for bit (tests/test_gpu_rollout.py).
RNG streams: the RRT-Connect query of env e at agent step t uses (seed + t, stream e); the fallback planners inside the
densification use streams E + e (simple planner) and 2E + e (main planner); e is the GLOBAL env id (`env_id_base` + row) and
E the global env count (`env_id_total`), so a sharded rollout draws what the unsharded one draws.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from . import _lib
from .agent_planning import (JointLimits, action_to_displacement, displacement_to_action, interpolation_steps, max_interpolation_steps,
                             is_planner_action, simple_interpolate_batch)
from .batch import BatchPlanner, _torch
from .planner import ITERS_PER_SECOND
from .scene import ENV_SPECS, planner_inputs

COUNTERS = ("mp", "rl", "interpolation", "mp_fail", "approximate", "invalid")


@dataclass
class RolloutConfig:
    """Defaults restated from the reference: config/__init__.py:24-110 (mopa section), config/sawyer.py:76-112,
    config/motion_planner.py:4-70.  `RolloutConfig.for_env(name)` fills the env-specific ones (config/pusher.py differs)."""
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
    # data-parallel runs (sample streams keyed by (seed, GLOBAL env id, iteration), so results do not depend on how
    # the envs are sharded): this rank's envs are rows env_id_base .. env_id_base + E - 1 of env_id_total envs in all
    # (rank * E and world * E; 0 total = this rank alone)
    env_id_base: int = 0
    env_id_total: int = 0
    # MoPA + IK action space (config/__init__.py --use_ik_target / --ik_target; rl/trainer.py:93-125): the policy outputs a
    # Cartesian displacement of the ik_target site (3) + a rotation quaternion (4) [+ the gripper entry]
    async_planner: bool = False       # RRT-Connect on side streams; envs waiting for a query sit out (see agent_step)
    planner_streams: int = 3          # RRT-Connect launches in flight at most (async_planner)
    planner_job_cap: int = 2048       # queries per asynchronous launch at most (the rest waits for the next free stream)
    planner_min_job: int = 1024       # queries an asynchronous launch waits for while other launches are in flight
    planner_first_iters: int = 300    # asynchronous launches first run with this iteration budget; the queries it does not solve
                                      # (a few %) are launched again with the full budget, among their kind.  A query's outcome
                                      # is a function of its endpoints and sample stream only, and a budget only ends the loop:
                                      # the second run retraces the first and goes on, so results are those of one full run --
                                      # but the many quick queries no longer wait for (or hold wave slots next to) the few that
                                      # take 2000 iterations.  0: one launch with the full budget
    device_paths: bool = True         # planner rows -> trajectories (un-wrap, densification) on the device (batch.postprocess_paths);
                                      # False: the array-operation form on the host (also serves the rare queries whose
                                      # densification needs the fallback planners)
    planner_chain: int = 1            # a first-phase launch is followed, on the same stream and with no host in between, by the
                                      # launch that continues its unsolved queries with the full budget (the kernel skips the
                                      # settled ones): the quick queries are picked up at an event between the two, the
                                      # budget-exhausting ones never wait in a retry pool.  0: pooled retry launches (round 2's form)
    planner_workgroups: int = 128     # persistent workgroups of an asynchronous launch.  A planner wave holds 256 registers (two
                                      # per SIMD): launches that took every slot would stall the main stream's kernels for their
                                      # whole bulk phase.  3 streams x 128 measured best on Push (tools/rollout_ab.sh: 1.03 M agent
                                      # steps/s; 3 x 64: 0.79 M, 2 x 256: 0.93 M)
    planner_chain_workgroups: int = 0 # workgroups of the continuation launch behind a first-phase launch (0: planner_workgroups).  With planner_exclusive
                                      # >= 1 that launch runs the workgroup-per-query build of K3 (four waves on a budget-exhausting query, the CU to itself):
                                      # few workgroups keep the CUs it takes away from the main stream's kernels few
    planner_exclusive: int = 0        # 1: the full-budget launches (continuation of a first-phase launch / pooled retries), 2: every asynchronous
                                      # planner launch keeps its CUs to itself (C ABI `exclusive_cu`): the one-wave-per-SIMD build of K3 with the
                                      # FP32 tree mirror in the CU's whole LDS (a budget-exhausting query: 41 instead of 46+ ms)
    discrete_action: bool = False     # --discrete_action (config/__init__.py:110; rl/mopa_rollouts.py:86-88,106-111,349): the policy's
                                      # `ac_type` head (1 = planner), not the action's magnitude, routes a step; direct actions
                                      # are then NOT divided by omega
    use_graphs: bool = False          # async_planner, joint-space actions, record=False: the fixed-shape halves of a call (policy
                                      # action -> target -> pull-back -> straight-line pre-check; execution + bookkeeping when no
                                      # planner launch finished in the call) are captured once as HIP graphs and replayed -- ~90
                                      # launches of host dispatch per call become two.  The returned tensors are then the graphs'
                                      # static buffers: valid until the next call
    walk_chunk: int = 0               # dynamics envs (a waypoint = one 75-sub-step physics launch whose time does not depend on how
                                      # many envs take part): a call executes at most this many waypoints per env; envs still on
                                      # their path are busy -- they sit out the policy's action of the following calls like envs
                                      # waiting for a query -- and complete their agent step in the call their walk ends in.  Every
                                      # env goes through the same transitions; launches stay full instead of draining towards the
                                      # longest path of the call.  0: a call walks every path to its end
    fused: bool = True                # the elementwise bookkeeping of a call as six library kernels (mopa_rollout_stage) instead of
                                      # ~170 torch launches; False: the torch form (the two agree bit for bit, tests/test_gpu_rollout.py)
    use_ik_target: bool = False
    ik_target: str = "grip_site"
    min_world_size: tuple = (-1.2, -1.2, 0.0)        # env/sawyer/sawyer.py:52-53
    max_world_size: tuple = (1.2, 1.2, 2.0)

    @classmethod
    def for_env(cls, env_name: str, **over):
        """the config the reference builds for `env_name`: config/sawyer.py / config/pusher.py / config/motion_planner.py defaults as
        restated in scene.ENV_SPECS (Pusher: ac_scale 0.1, action_range 1.0, range 0.2, simple_planner_range 0.1,
        simple_planner_timelimit 0.02, step_size 0.04, joint_margin 0, contact_threshold -0.0015), then `over`."""
        from .scene import ENV_SPECS
        sp = ENV_SPECS[env_name]
        kw = dict(omega=sp.omega, action_range=sp.action_range, ac_scale=sp.ac_scale, num_trials=sp.num_trials, step_size=sp.step_size,
                  timelimit=sp.timelimit, simple_planner_timelimit=sp.simple_planner_timelimit, range=sp.range,
                  simple_planner_range=sp.simple_planner_range, joint_margin=sp.joint_margin, contact_threshold=sp.contact_threshold)
        kw.update(over)
        return cls(**kw)


def reuse_transitions(out, cfg, n_arm: int, rng, max_reuse_data: int = 30, grip_qpos_idx=None):
    """The `reuse_data` relabelling of rl/mopa_rollouts.py:204-300 on the record of one `agent_step(..., record=True)`:
    for every env that executed a planner path with more than 3 waypoints, up to min(len, max_reuse_data) random
    (start, goal) waypoint pairs become extra transitions  ob_list[start] --inverse-displacement action--> ob_list[goal]
    with reward (meta_rew[goal] - meta_rew[start]) * gamma^-(start+1), done = done_list[goal],
    intra_steps = goal - start - 1, kept only if the relabelled action is a planner action inside [-1, 1].
    `rng`: a numpy RandomState-like object (`randint(low, high)`) shared by all envs, or a callable env -> such an object
    (the reference draws from the global np.random, one env per process).
    `grip_qpos_idx`: qpos address of the first gripper joint for envs whose action has a gripper entry (Lift, dof 8):
    `env.form_action` appends that joint's difference (env/sawyer/sawyer.py:283-299), `valid_action` checks the whole vector,
    `is_planner_ac` the arm entries.
    Returns a list of dicts (env, start, goal, ob, ac, rew, done, intra_steps, ob_next) of numpy values."""
    rec = out["record"]
    if cfg.use_ik_target:
        # With use_ik_target the reference never moves target_qpos off curr_qpos (BatchMoPARollout._seg_pre), so a planner step
        # executes the two waypoints of a zero-length line and `len(ob_list) > 3` (:222) never holds: the IK branch of the
        # relabelling (:247-262, cart_list / quat_list) cannot be reached.  Anything longer here is not the reference's rollout.
        if bool((rec["n_exec"] > 3).any()):
            raise NotImplementedError("reuse_data relabelling for the IK action space (cart_list / quat_list, rl/mopa_rollouts.py:247-262)")
        return []
    if grip_qpos_idx is None and int(out["ac"].shape[1]) > n_arm:
        raise ValueError("the env's action has a gripper entry: pass grip_qpos_idx (BatchMoPARollout.reuse_transitions does)")
    ac_type = out["ac_type"].cpu().numpy() if (cfg.discrete_action and "ac_type" in out) else None
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
            if grip_qpos_idx is not None:
                ac = np.concatenate([ac, [wp[e, goal, grip_qpos_idx] - wp[e, start, grip_qpos_idx]]])
            in_box = bool(np.all(ac >= -1.0) and np.all(ac <= 1.0))
            if not (is_planner and in_box):
                continue
            rew = (mr[e, goal] - mr[e, start]) * cfg.discount_factor ** (-(start + 1))
            extra.append({"env": int(e), "start": start, "goal": goal, "ob": ob[e, start], "ac": ac, "rew": float(rew),
                          "done": int(dn[e, goal]), "intra_steps": goal - start - 1, "ob_next": ob[e, goal]})
            if ac_type is not None:       # `inter_subgoal_ac["ac_type"] = ac["ac_type"]` (rl/mopa_rollouts.py:266-267)
                extra[-1]["ac_type"] = int(ac_type[e])
    return extra


_SIDE_STREAMS = {}


def _side_streams(dev, n):
    """The planner's side streams, shared by every rollout object of the process: the GPU runs only a few hardware queues
    side by side (4 by default), streams beyond that share a queue with another stream and a 48 ms planner launch then sits
    in front of the main stream's kernels -- so streams are created once per device and handed out again."""
    torch = _torch()
    pool = _SIDE_STREAMS.setdefault(str(dev), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def wrap_unlimited(q, idx):
    """`SamplingBasedPlanner.convert_nonlimited` (motion_planners/sampling_based_planner.py:51-55) for rows of states [M, nq]
    (torch): `util.env.joint_convert` (util/env.py:15-25) on the columns `idx` of the unlimited joints -- Python's float // and %
    spelled out (fmod; the quotient rounded to the nearest integer).  Returns a copy (or `q` itself when `idx` is empty); the
    caller's un-wrapped state, from which the trajectory is rebuilt, is left alone."""
    if not len(idx):
        return q
    torch = _torch()
    q = q.clone()
    for j in idx:
        a = q[:, j]
        m = torch.fmod(a, 3.14)                                           # a % 3.14 for a > 0, a % -3.14 for a <= 0 (sign of a either way)
        k = torch.round((a - m) / torch.where(a > 0, a.new_tensor(3.14), a.new_tensor(-3.14)))       # a // +-3.14 (the divisor in a's float64)
        odd = torch.remainder(k, 2.0) != 0
        r = torch.where(odd, torch.where(a > 0, m - 3.14, m + 3.14), m)
        q[:, j] = torch.where(a == 0, torch.full_like(a, -0.0), r)        # (0.0 % -3.14 is -0.0 in Python)
    return q


def seam_steps_np(P, idx):
    """successive differences P[..., k+1, :] - P[..., k, :] of planner rows (numpy) with the seam rule of the reference's un-wrap
    loop on the unlimited joints' columns `idx` (sampling_based_planner.py:79-97; same sums, same order)"""
    prev, cur = P[..., :-1, :], P[..., 1:, :]
    step = cur - prev
    for j in idx:
        p, c = prev[..., j], cur[..., j]
        jump = np.abs(c - p) > 3.14
        up = jump & (p > 0) & (c <= 0)
        down = jump & ~up & (p < 0) & (c > 0)
        sj = step[..., j]
        sj[up] = ((3.14 - p[up]) + c[up]) + 3.14
        sj[down] = -(((3.14 - c[down]) + p[down]) + 3.14)
    return step



def ik_targets_torch(site_pos, site_mat, ac, action_range, world_lo, world_hi):
    """The IK problem of `ik_displacement` as array operations (what `BatchIK.targets` computes in one launch): returns (target_cart, target_quat)."""
    torch = _torch()
    lo = torch.tensor(world_lo, dtype=torch.float64, device=ac.device)
    hi = torch.tensor(world_hi, dtype=torch.float64, device=ac.device)
    target_cart = torch.minimum(torch.maximum(site_pos + action_range * ac[:, :3], lo), hi).contiguous()
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
    return target_cart, target_quat


class BatchMoPARollout:
    def reuse_transitions(self, out, rng, max_reuse_data: int = 30):
        """`reuse_transitions` on a recorded step of this rollout, with the env's gripper joint supplied where its action has
        a gripper entry (Lift)"""
        f = self.env.facts
        grip = int(f.grip_qpos_idx[0]) if self.ac_dim > self.n and len(f.grip_qpos_idx) else None
        return reuse_transitions(out, self.cfg, self.n, rng, max_reuse_data=max_reuse_data, grip_qpos_idx=grip)

    def __init__(self, env, cfg: Optional[RolloutConfig] = None):
        torch = _torch()
        self.env = env
        self.cfg = cfg if cfg is not None else RolloutConfig()
        if abs(env.ac_scale - self.cfg.ac_scale) > 0:
            raise _lib.MopaError("env.ac_scale and RolloutConfig.ac_scale differ")
        spec = ENV_SPECS[env.env_name]
        pi = planner_inputs(env.env_name, env.model)
        # unlimited joints (`non_limited_idx`, rl/trainer.py:80-82; Pusher: joint0): SamplingBasedPlanner wraps a query's start and
        # goal into (-3.14, 3.14) there before planning and takes the returned steps across the seam the short way round
        # (motion_planners/sampling_based_planner.py:51-55,60-99) -- `_wrap_q` / the seam mask of the path post-processing
        self._seam_idx = [int(i) for i in pi.non_limited_idx]
        self._seam_mask = sum(1 << i for i in self._seam_idx)
        self.pi = pi
        dev_index = env.device.index if env.device.index is not None else -1
        mk = lambda r: _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, self.cfg.contact_threshold, range_=r,
                                  seed=self.cfg.seed, device=dev_index)
        self.scene, self.simple_scene = mk(self.cfg.range), mk(self.cfg.simple_planner_range)
        self.bp, self._bp_simple = BatchPlanner(self.scene), BatchPlanner(self.simple_scene)
        self.E, self.nq, self.n = env.E, env.nq, env.n_arm
        self.arm = list(int(i) for i in env.facts.arm_qpos_idx)
        assert self.arm == list(range(self.n)), "the reference slices qpos[:n] (rl/sac_agent.py:275-278)"
        f = env.facts
        dev, f64 = env.device, torch.float64
        self.limits = JointLimits(f.qpos_min, f.qpos_max, f.qpos_limited, self.cfg.joint_margin, device=dev)
        self.counters: Dict[str, "object"] = {k: torch.zeros(self.E, dtype=torch.int64, device=dev) for k in COUNTERS}
        self.t_env = torch.zeros(self.E, dtype=torch.int64, device=dev)     # agent steps every env has completed
        self._t = 0
        self._t_dev = torch.zeros((), dtype=torch.int64, device=dev)         # the same counter on the device (captured launches read it)
        self.busy = torch.zeros(self.E, dtype=torch.bool, device=dev)        # env waits for an RRT-Connect query (async_planner)
        self._jobs = []
        # blocked envs waiting for the next RRT-Connect launch: mask + their (clipped) current state and target
        self._pool_mask = torch.zeros(self.E, dtype=torch.bool, device=dev)
        self._res = None      # per-env parking buffers of planner state between a first-phase launch and its retry (_park_state)
        self._retry_mask = torch.zeros(self.E, dtype=torch.bool, device=dev)     # ... and those whose first, short launch ran out
        self._wait_since = torch.zeros(self.E, dtype=torch.int64, device=dev)    # call in which an env started to wait
        self.n_retried = torch.zeros((), dtype=torch.int64, device=dev)
        self._q_cur = torch.zeros(self.E, self.nq, dtype=torch.float64, device=dev)
        self._q_tgt = torch.zeros(self.E, self.nq, dtype=torch.float64, device=dev)
        self._k_interp = max_interpolation_steps(self.cfg.action_range, self.cfg.ac_scale)
        self._interp_overflow = torch.zeros((), dtype=torch.bool, device=dev)
        self._disc = {}
        # a state that is certainly valid (the env's initial pose): stands in for the rows a launch has no business with
        self._safe_q = torch.tensor(np.asarray(env.init_qpos_row, dtype=np.float64)[None], device=dev)
        if not bool(self._valid(self._safe_q)[0]):
            raise _lib.MopaError("the env's initial pose is not a valid state")
        self._streams = _side_streams(dev, max(1, self.cfg.planner_streams)) if self.cfg.async_planner else []
        self._next_stream = 0
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
        self._pend_ob = torch.zeros(self.E, env.obs_dim, dtype=f64, device=dev)      # ob / ac of the step a busy env is in
        self._pend_ac = torch.zeros(self.E, self.ac_dim, dtype=f64, device=dev)
        self._ac_type_in = torch.zeros(self.E, dtype=torch.int64, device=dev)      # discrete_action: this call's ac_type / that of a pending step
        self._pend_type = torch.zeros(self.E, dtype=torch.int64, device=dev)

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
        # (one launch; the module's `ik_targets_torch` is the array-operation form it replaced -- ~70 elementwise launches per call -- kept as
        #  the checker of tests/test_gpu_ik.py)
        target_cart, target_quat = self.ik.targets(site_pos, site_mat, ac, cfg.action_range, cfg.min_world_size, cfg.max_world_size)
        q_ik = cur.clone()
        self.ik.solve(q_ik, target_cart, target_quat, max_steps=100, tol=1e-2)
        arm_t = torch.minimum(torch.maximum(q_ik[:, :self.n], self._ik_lo), self._ik_hi)
        return arm_t - cur[:, :self.n]

    def _valid(self, q):
        return self.bp.is_valid(q[:, self.arm].contiguous(), q.contiguous(), samples_per_env=1).bool()

    # ------------------------------------------------------------------
    @property
    def t(self) -> int:
        """agent steps taken (lock-step count); part of the planner's sample-stream key.  Assigning it puts every env at that
        step (tests replay recorded steps)."""
        return self._t

    @t.setter
    def t(self, value: int):
        self._t = int(value)
        self._t_dev.fill_(int(value))
        self.t_env.fill_(int(value))

    def _wrap_q(self, q):
        return wrap_unlimited(q, self._seam_idx)

    def _seam_steps_np(self, P):
        return seam_steps_np(P, self._seam_idx)

    def _rrt_launch(self, cur_f, target_f, ids, stream=None, iters=None, keep=False, resume=None, chain=False, steps=None, seeds=None):
        """RRT-Connect (K3) for the envs `ids` whose straight line is blocked (:205-209): asynchronous, optionally on a side
        stream.  The sample stream of a query is keyed by (cfg.seed + the env's own step count, env id), so an env's plans do
        not depend on which other envs are planned with it or when."""
        torch = _torch()
        cfg = self.cfg
        if seeds is None:
            seeds = (self.t_env[ids] + cfg.seed).contiguous()
        gids = (ids + int(cfg.env_id_base)).contiguous() if cfg.env_id_base else ids       # the planner's stream id: the GLOBAL env id
        iters = self.main_iters if iters is None else int(iters)
        job = {"ids": ids, "cur": cur_f, "target": target_f, "steps": self.t_env[ids].clone() if steps is None else steps, "event": None, "stage": "rrt", "stream": stream,
               "iters": iters}
        if self._seam_idx:       # the planner sees the wrapped endpoints; job["cur"] stays the caller's state
            cur_f, target_f = self._wrap_q(cur_f).contiguous(), self._wrap_q(target_f).contiguous()
        if stream is None:
            job["path"], job["plen"], job["status"], _ = self.bp.plan(cur_f, target_f, max_iters=iters, max_nodes=cfg.max_nodes,
                                                                      max_path=cfg.max_path, seed=cfg.seed, env_ids=gids, seeds=seeds)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                res = self.bp.plan(cur_f, target_f, max_iters=iters, max_nodes=cfg.max_nodes, max_path=cfg.max_path, seed=cfg.seed,
                                   env_ids=gids, seeds=seeds, stream=stream, max_workgroups=cfg.planner_workgroups,
                                   keep_state=keep, resume=resume,
                                   exclusive=cfg.planner_exclusive >= 2 or (cfg.planner_exclusive == 1 and (resume is not None or not keep) and iters == self.main_iters))
                job["path"], job["plen"], job["status"] = res[0], res[1], res[2]
                if keep:
                    job["pstate"] = res[4]      # trees + counters of the queries this budget leaves unsolved (see _seg_plan)
                if chain:
                    # the quick queries' results are complete HERE; what follows on this stream is the continuation of the rest
                    job["event"] = torch.cuda.Event()
                    job["event"].record(stream)
                    rb = self.bp.plan(cur_f, target_f, max_iters=self.main_iters, max_nodes=cfg.max_nodes, max_path=cfg.max_path,
                                      seed=cfg.seed, env_ids=gids, seeds=seeds, stream=stream,
                                      max_workgroups=cfg.planner_chain_workgroups or cfg.planner_workgroups,
                                      resume=res[4], exclusive=cfg.planner_exclusive >= 1)
                    ev_b = torch.cuda.Event()
                    ev_b.record(stream)
                    job["chain"] = {"path": rb[0], "plen": rb[1], "status": rb[2], "event": ev_b}
                    job["post_on_main"] = True      # this stream is busy with the continuation: post-process on the caller's
                    for t in (cur_f, target_f, ids, gids, seeds):
                        t.record_stream(stream)
                    return job
                for t in (cur_f, target_f, ids, gids, seeds):
                    t.record_stream(stream)
                job["event"] = torch.cuda.Event()
                job["event"].record(stream)
        return job

    def _rrt_advance(self, job, wait: bool):
        """Drive a planner job through its stages; returns True when it is finished (job["result"] = (trajs: row -> [L, nq]
        numpy for the successful rows; success, valid, exact as numpy bool arrays over the job's rows)).
          stage "rrt"     results of the main RRT-Connect launch: sentinel decoding, the un-wrapped trajectory of
                          `SamplingBasedPlanner.plan` / `PlannerAgent.plan`, densification (rl/sac_agent.py:216-233): a path
                          segment longer than ac_scale in some joint is cut by the straight-line rule from its clipped start; all
                          interior states of all segments are validated in ONE launch
          stage "simple"  the (rare) segments with an invalid interior state: simple planner, one batched launch (:300-303)
          stage "main"    those it could not connect: main planner, one batched launch (:304-306); else the segment stays [end]
        With `wait` every stage is waited for (lock-step); otherwise a stage whose launch has not finished returns False."""
        torch = _torch()
        cfg, n = self.cfg, self.n
        if job["stage"] == "split":
            return self._rrt_join(job, wait)
        if job["stage"] in ("rrt", "device") and cfg.device_paths and self.nq <= 64 and "bucketed" not in job and "unwrapped" not in job:
            return self._rrt_device(job, wait)
        while True:
            if job["event"] is not None:
                if wait:
                    job["event"].synchronize()
                elif not job["event"].query():
                    return False
            plen_h, st_h = job["plen"].cpu().numpy(), job["status"].cpu().numpy()
            if job["stage"] == "rrt" and len(plen_h) > 64 and "bucketed" not in job:
                return self._rrt_split(job, plen_h, wait)
            path_h = job["path"][:, :max(1, int(plen_h.max()))].cpu().numpy()      # the [max_path] tail of every row is unused
            path_h[np.arange(path_h.shape[1])[None, :] >= plen_h[:, None]] = 0.0   # ... and holds whatever the allocator left
            if job["stage"] == "rrt":
                cur_h = job["cur"].cpu().numpy()
                job["ids_h"], job["steps_h"] = job["ids"].cpu().numpy(), job["steps"].cpu().numpy()
                M = len(job["ids_h"])
                bad = st_h != 0        # sentinel rows (sampling_based_planner.py:64-69): -5 goal invalid, -4 no exact solution
                valid, exact = np.ones(M, dtype=bool), np.ones(M, dtype=bool)
                valid[bad] = st_h[bad] != _lib.PLAN_INVALID_GOAL
                exact[bad] = st_h[bad] != _lib.PLAN_NO_EXACT
                good = np.where(~bad)[0]
                job.update(flags=(~bad, valid, exact), good=good, replacement={}, fb=[], seg_r=np.zeros(0, dtype=np.int64),
                           seg_i=np.zeros(0, dtype=np.int64), T=None, nrow=np.zeros(0, dtype=np.int64))
                if len(good):
                    # SamplingBasedPlanner.plan rebuilds the trajectory from successive differences (:71-99) and PlannerAgent
                    # drops row 0: tr[k] = tr[k-1] + (states[k] - states[k-1]), tr[0] = cur.  np.add.accumulate is that same
                    # strictly sequential sum, for all paths at once (rows past a path's length hold garbage and are cut off).
                    P = path_h[good]
                    if job.get("unwrapped"):       # rows that went through mopa_paths_unwrap_batch already (row 0 = cur)
                        T = P
                    else:
                        A = np.concatenate([cur_h[good][:, None, :], self._seam_steps_np(P)], axis=1)
                        T = np.add.accumulate(A, axis=1)
                    nrow = plen_h[good] - 1
                    job["T"], job["nrow"] = T, nrow
                    if cfg.interpolation and T.shape[1] > 1:
                        step = T[:, 1:, :n] - T[:, :-1, :n]                      # waypoint i minus its predecessor (cur for i = 0)
                        far = ((step < -cfg.ac_scale) | (step > cfg.ac_scale)).any(axis=2)
                        far &= np.arange(far.shape[1])[None, :] < nrow[:, None]
                        job["seg_r"], job["seg_i"] = np.nonzero(far)             # segment k: waypoint seg_i[k] of path seg_r[k]
                if len(job["seg_r"]):
                    self._densify_cut(job)
                if not job["fb"]:
                    return self._rrt_done(job)
                self._fallback_launch(job, "simple")
                continue
            # a fallback stage came back: rows = job["fb"] (indices into seg_jobs)
            left = []
            for r, k in enumerate(job["fb"]):
                if st_h[r] == 0:
                    p = path_h[r, :plen_h[r]]
                    # SamplingBasedPlanner.plan: start + running sum of successive differences; PlannerAgent drops row 0
                    job["replacement"][k] = np.add.accumulate(np.vstack([job["starts"][k][None], self._seam_steps_np(p)]), axis=0)[1:]
                elif job["stage"] == "simple":
                    left.append(k)
                else:
                    job["replacement"][k] = job["ends"][k][None]
            job["fb"] = left
            if not left:
                return self._rrt_done(job)
            self._fallback_launch(job, "main")

    def _rrt_device(self, job, wait: bool):
        """Stage "rrt" with the path post-processing on the device (batch.postprocess_paths: three small launches around one
        validity launch on the job's stream; one read-back of two totals).  Queries whose densification met an invalid
        interior state -- they need the fallback planners -- go through the host form as a sub-job, from their un-wrapped
        rows.  job["result"] then holds device tensors."""
        import contextlib
        torch = _torch()
        cfg = self.cfg
        just_enqueued = False
        if job["stage"] == "rrt":
            if job["event"] is not None:
                if wait:
                    job["event"].synchronize()
                elif not job["event"].query():
                    return False
            just_enqueued = True
            if "lazy" in job:        # continuation half of a chained launch: its rows of the continuation's outputs
                src, rows = job.pop("lazy")
                if job["stream"] is not None and job.get("built") is not None:
                    job["stream"].wait_event(job.pop("built"))
                with (torch.cuda.stream(job["stream"]) if job["stream"] is not None else contextlib.nullcontext()):
                    job["path"], job["plen"], job["status"] = src["path"][rows], src["plen"][rows], src["status"][rows]
            pstream = None if job.get("post_on_main") else job["stream"]
            if job.get("post_on_main"):
                for x in (job["path"], job["plen"], job["status"], job["cur"]):
                    x.record_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(pstream) if pstream is not None else contextlib.nullcontext()
            with ctx:
                from .batch import postprocess_paths
                out, ln, need = postprocess_paths(job["path"], job["plen"], job["status"], job["cur"], self.n, cfg.ac_scale,
                                                  cfg.interpolation, self.limits, self._valid, stream=pstream, seam_mask=self._seam_mask)
                st = job["status"]
                ok = st == 0           # sentinel rows (sampling_based_planner.py:64-69): -5 goal invalid, -4 no exact solution
                job["dev"] = [out, ln, ok, st != _lib.PLAN_INVALID_GOAL, st != _lib.PLAN_NO_EXACT]
                rows = torch.nonzero(need).flatten()
                if len(rows):
                    sub = {k: job[k][rows].contiguous() for k in ("ids", "cur", "target", "steps", "plen", "status")}
                    sub.update(path=job["path"][rows].contiguous(), event=None, stage="rrt", stream=pstream, unwrapped=True)
                    job["sub"], job["sub_rows"] = sub, rows
                if pstream is not None:
                    job["event"] = torch.cuda.Event()
                    job["event"].record(pstream)
                else:
                    job["event"] = None
            job["stage"] = "device"
        # stage "device": the launches above (and the sub-job, if any) have to be finished.  Right after the post-processing
        # was enqueued what is left of it is one small assemble launch (its read-backs already waited for the rest): waiting
        # those microseconds out here saves the job -- and the envs that wait for it -- a whole call
        if job["event"] is not None:
            if wait or (just_enqueued and "sub" not in job):
                job["event"].synchronize()
            elif not job["event"].query():
                return False
        out, ln, ok, v, e = job["dev"]
        if "sub" in job:
            if not self._rrt_advance(job["sub"], wait):
                return False
            tr_s, ln_s, *_ = job["sub"]["result"]
            rows = job["sub_rows"]
            tr_s, ln_s = torch.as_tensor(tr_s, device=out.device), torch.as_tensor(ln_s, device=out.device)
            if tr_s.shape[1] > out.shape[1]:
                out = torch.cat([out, torch.zeros(out.shape[0], tr_s.shape[1] - out.shape[1], self.nq, dtype=out.dtype, device=out.device)], dim=1)
            out[rows, :tr_s.shape[1]] = tr_s
            ln[rows] = ln_s
        job["result"] = (out, ln, ok, v, e)
        return True

    _BUCKETS = (6, 12, 24, 48)

    def _rrt_split(self, job, plen_h, wait):
        """A finished launch's paths are post-processed as padded [rows, longest path, nq] arrays; a few long paths among many
        short ones would make that mostly padding.  Split the job's rows by path length into sub-jobs (each padded to its own
        longest path) and advance those; job["result"] is put together from theirs."""
        torch = _torch()
        edges = np.searchsorted(np.array(self._BUCKETS), plen_h, side="left")
        subs, result_rows = [], []
        for b in np.unique(edges):
            rows = np.nonzero(edges == b)[0]
            rt = torch.as_tensor(rows, device=job["path"].device)
            L = max(1, int(plen_h[rows].max()))
            sub = {k: job[k][rt] for k in ("ids", "cur", "target", "steps", "plen", "status")}
            sub.update(path=job["path"][rt, :L], event=None, stage="rrt", stream=job["stream"], bucketed=True)
            if job.get("unwrapped"):
                sub["unwrapped"] = True
            subs.append(sub)
            result_rows.append(rows)
        job["subs"], job["sub_rows"], job["stage"] = subs, result_rows, "split"
        return self._rrt_join(job, wait)

    def _rrt_join(self, job, wait):
        done = True
        for sub in job["subs"]:
            if "result" not in sub:
                done &= bool(self._rrt_advance(sub, wait))
        if not done:
            return False
        M, nq = len(job["plen"]), self.nq
        L = max(s["result"][0].shape[1] for s in job["subs"])
        traj, ln = np.zeros((M, L, nq)), np.zeros(M, dtype=np.int64)
        flags = [np.zeros(M, dtype=bool) for _ in range(3)]
        for sub, rows in zip(job["subs"], job["sub_rows"]):
            tr, l, *fl = sub["result"]
            traj[rows, :tr.shape[1]], ln[rows] = tr, l
            for f, g in zip(flags, fl):
                f[rows] = g
        job["result"] = (traj, ln) + tuple(flags)
        return True

    def _fallback_launch(self, job, stage):
        torch = _torch()
        cfg = self.cfg
        dev = self.env.device
        ks = job["fb"]
        scene_bp, iters, base = ((self._bp_simple, self.simple_iters, 1) if stage == "simple" else (self.bp, self.main_iters, 2))
        starts = self._wrap_q(torch.tensor(job["starts"][ks], device=dev))
        ends = self._wrap_q(torch.tensor(job["ends"][ks], device=dev))
        rows = job["good"][job["seg_r"][ks]]
        total = int(cfg.env_id_total) if cfg.env_id_total else self.E
        ids = torch.tensor(total * base + int(cfg.env_id_base) + job["ids_h"][rows], dtype=torch.int64, device=dev)
        seeds = torch.tensor(cfg.seed + job["steps_h"][rows], dtype=torch.int64, device=dev)
        stream = job["stream"]
        job["stage"] = stage
        if stream is None:
            job["path"], job["plen"], job["status"], _ = scene_bp.plan(starts, ends, max_iters=iters, max_nodes=cfg.max_nodes,
                                                                       max_path=cfg.max_path, seed=cfg.seed, env_ids=ids, seeds=seeds)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                job["path"], job["plen"], job["status"], _ = scene_bp.plan(starts, ends, max_iters=iters, max_nodes=cfg.max_nodes,
                                                                           max_path=cfg.max_path, seed=cfg.seed, env_ids=ids, seeds=seeds,
                                                                           stream=stream, max_workgroups=cfg.planner_workgroups)
                job["keep"] = (starts, ends, ids, seeds)
                job["event"] = torch.cuda.Event()
                job["event"].record(stream)

    def _rrt_done(self, job):
        """assemble the final trajectories: [M, L, nq] padded + lengths (0 for the failed rows).  Waypoint i of path r becomes
        `pieces[r, i]` rows: itself, or -- densified -- its cut (count interior states + the waypoint), or a fallback path."""
        M, nq = len(job["ids_h"]), self.nq
        good, T, nrow = job["good"], job["T"], job["nrow"]
        if not len(good):
            job["result"] = (np.zeros((M, 1, nq)), np.zeros(M, dtype=np.int64)) + job["flags"]
            return True
        R, W = T.shape[0], T.shape[1] - 1
        seg_r, seg_i = job["seg_r"], job["seg_i"]
        pieces = (np.arange(W)[None, :] < nrow[:, None]).astype(np.int64)
        seg_len = np.zeros(len(seg_r), dtype=np.int64)
        if len(seg_r):
            seg_len[:] = job["count"] + 1
            for k, rep in job["replacement"].items():
                seg_len[k] = len(rep)
            pieces[seg_r, seg_i] = seg_len
        off = np.cumsum(pieces, axis=1) - pieces                       # first output row of waypoint i
        length = pieces.sum(axis=1)
        out = np.zeros((R, max(1, int(length.max())), nq))
        r_idx, i_idx = np.nonzero(np.arange(W)[None, :] < nrow[:, None])
        out[r_idx, off[r_idx, i_idx] + pieces[r_idx, i_idx] - 1] = T[r_idx, i_idx + 1]      # every piece ends in its waypoint
        if len(seg_r):
            plain = np.array([k not in job["replacement"] for k in range(len(seg_r))])
            kk = np.nonzero(plain)[0]
            if len(kk):
                cnt = job["count"][kk]
                kr = np.repeat(kk, cnt)                                                      # (segment, interior state c) pairs
                c = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
                out[seg_r[kr], off[seg_r[kr], seg_i[kr]] + c] = job["walk"][kr, c]
            for k, rep in job["replacement"].items():
                o = off[seg_r[k], seg_i[k]]
                out[seg_r[k], o:o + len(rep)] = rep
        traj = np.zeros((M, out.shape[1], nq))
        ln = np.zeros(M, dtype=np.int64)
        traj[good], ln[good] = out, length
        job["result"] = (traj, ln) + job["flags"]
        return True

    def _rrt_finish(self, job):
        self._rrt_advance(job, wait=True)
        return job["result"]

    def plan(self, cur, target, env_ids):
        """`SACAgent.plan` (rl/sac_agent.py:198-235) for M envs, synchronously.  Returns (traj [M, L, nq] device tensor, length
        [M] device int64 -- 0 where the plan failed --, and the boolean numpy arrays success, interpolation, valid, exact).
        Straight lines that validate never leave the GPU; only the envs that needed RRT-Connect are post-processed on the
        host (ragged paths: successive differences, densification) and uploaded into their rows."""
        torch = _torch()
        cfg = self.cfg
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
        fi = torch.as_tensor(fail, device=cur.device)
        job = self._rrt_launch(cur[fi].contiguous(), target[fi].contiguous(), env_ids[fi].contiguous())
        tr_j, ln_j, s_j, v_j, e_j = self._rrt_finish(job)
        s_j, v_j, e_j = (x.cpu().numpy() if hasattr(x, "cpu") else x for x in (s_j, v_j, e_j))
        interpolation[fail] = False
        success[fail], valid[fail], exact[fail] = s_j, v_j, e_j
        traj_t, lens = self._merge_paths(traj_t, lens, tr_j, ln_j, fi)
        return traj_t, lens, success, interpolation, valid, exact

    def _merge_paths(self, traj_t, lens, tr_j, ln_j, rows):
        """write a job's padded trajectories (tr_j [M, L, nq], ln_j [M]; 0 = no path; device tensors or host arrays) into rows
        `rows` (device index tensor) of traj_t / lens"""
        torch = _torch()
        L = int(tr_j.shape[1])
        if L > traj_t.shape[1]:
            traj_t = torch.cat([traj_t, torch.zeros(traj_t.shape[0], L - traj_t.shape[1], self.nq, dtype=traj_t.dtype, device=traj_t.device)], dim=1)
        traj_t[rows, :L] = torch.as_tensor(tr_j, device=traj_t.device)
        lens[rows] = torch.as_tensor(ln_j, device=traj_t.device)
        return traj_t, lens

    def _densify_cut(self, job):
        """cut every long segment of a job by the straight-line rule (steps <= 0.8 ac_scale from the segment's clipped start,
        then the segment's end), validate all interior states in one launch; segments whose interior states are all valid are
        done, the others are listed in job["fb"] for the fallback planners.  All segments at once (array operations)."""
        torch = _torch()
        cfg, n = self.cfg, self.n
        T, seg_r, seg_i = job["T"], job["seg_r"], job["seg_i"]
        starts = self.limits.clip_state_np(T[seg_r, seg_i])                         # :266 clip_qpos on every segment start
        ends = T[seg_r, seg_i + 1]
        S = len(seg_r)
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
        job.update(starts=starts, ends=ends, count=count, walk=walk, fb=list(np.nonzero(~verdict.all(axis=1))[0]))

    # ------------------------------------------------------------------
    def agent_step(self, ac, record: bool = False, ac_type=None):
        """One agent step for all E envs.  ac: float64 [E, >= self.ac_dim] GPU tensor (policy output in [-1, 1]).
        Returns a dict of GPU tensors: ob [E,obs_dim] (before), ac (the action each transition belongs to), ob_next
        [E,obs_dim], rew [E] (SMDP return of the step), done [E] uint8, intra_steps [E] int64, is_planner [E] bool, success [E]
        (env success flag), `path_len`, `plan_ok`, and `stepped` [E] bool: the envs that completed an agent step in this call.
        record=True adds `record`: per executed waypoint k the obs after it, the running SMDP return, the done flag and
        the waypoint itself ([E, L, ...]; `n_exec` [E] = waypoints actually executed) -- the `ob_list / meta_rew_list /
        done_list / traj` of the reference, input of `reuse_transitions`.

        Lock-step (default): every env steps in every call; the call returns when the slowest RRT-Connect query of the step is
        done.  `cfg.async_planner`: the RRT-Connect queries of a call run on side streams while the call returns; their envs
        are `busy` -- they sit out the following calls (their rows of `ac` are ignored) -- until their query has finished, and
        complete their step (path execution or the failed-plan step) in the first call after that.  Envs are independent and a
        query's sample stream is keyed by the env's own step count, so each env goes through the same sequence of transitions
        either way; only their interleaving differs.  Rows of the outputs are meaningful where `stepped`."""
        if self.cfg.discrete_action:
            if ac_type is None:
                raise _lib.MopaError("discrete_action: agent_step needs the policy's ac_type [E] (1 = motion planner, 0 = direct)")
            self._ac_type_in.copy_(ac_type.reshape(-1))
        if self.cfg.use_graphs and not record and getattr(self, "timing", None) is None:
            return self._agent_step_graphs(ac)
        bag = self._seg_pre(ac)
        self._seg_plan(bag)
        res = self._seg_exec(bag, record)
        self._t += 1
        return res

    def _agent_step_graphs(self, ac, warmup: int = 2):
        """agent_step with the two fixed-shape parts replayed from HIP graphs (cfg.use_graphs).  The first `warmup` calls run
        eagerly on the stream the graphs are then captured on -- so that every per-stream scratch buffer of the library
        exists before capture --, the capture itself is the execution of its call (capture, then replay)."""
        torch = _torch()
        if self.cfg.use_ik_target or not self.cfg.async_planner:
            raise _lib.MopaError("use_graphs serves the asynchronous joint-space rollout")
        if getattr(self.env, "dynamics", False):
            # waypoint execution through the physics loops over a data-dependent path length on the host: not capturable
            raise _lib.MopaError("use_graphs serves the kinematic env (a dynamics env executes waypoints step by step on the host)")
        G = getattr(self, "_graphs", None)
        main = torch.cuda.current_stream(self.env.device)
        if G is None:
            G = self._graphs = {"calls": 0, "stream": torch.cuda.Stream(device=self.env.device), "pool": torch.cuda.graph_pool_handle(),
                                "ac": torch.zeros(self.E, ac.shape[1], dtype=torch.float64, device=self.env.device),
                                "pre": None, "exec": None}
        st = G["stream"]
        if G["calls"] < warmup:
            G["calls"] += 1
            st.wait_stream(main)
            with torch.cuda.stream(st):
                bag = self._seg_pre(ac)
                self._seg_plan(bag)
                res = self._seg_exec(bag, False)
            for v in res.values():
                v.record_stream(main)
            main.wait_stream(st)
            self._t += 1
            return res
        G["ac"].copy_(ac)
        import os as _os
        which = _os.environ.get("MOPA_ROLLOUT_GRAPHS", "both")       # pre | exec | both: bisecting knob
        if which == "exec":
            bag = self._seg_pre(G["ac"])
            self._seg_plan(bag)
            res = self._seg_exec(bag, False)
            self._t += 1
            return res
        if G["pre"] is None:
            st.wait_stream(main)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=G["pool"], stream=st, capture_error_mode="relaxed"):
                G["bag"] = self._seg_pre(G["ac"])
            G["pre"] = g
        G["pre"].replay()
        bag = dict(G["bag"])
        self._seg_plan(bag)
        if which == "both" and bag["n_finished"] == 0 and bag["traj_pad"] is G["bag"]["traj_pad"]:
            bag["finished"] = None       # (the all-false mask is then created inside the graph: a static buffer)
            if G["exec"] is None:
                st.wait_stream(main)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=G["pool"], stream=st, capture_error_mode="relaxed"):
                    G["res"] = self._seg_exec(bag, False)
                G["exec"] = g
            G["exec"].replay()
            res = G["res"]
        else:
            res = self._seg_exec(bag, False)
        self._t += 1
        return res

    def _mark(self, name):
        tm = getattr(self, "timing", None)     # optional dict: phase -> seconds (each mark synchronises; profiling only)
        if tm is None:
            return
        import time as _time
        torch = _torch()
        torch.cuda.current_stream().synchronize()    # the main stream only: planner launches on side streams go on
        now = _time.perf_counter()
        last = getattr(self, "_mark_t", None)
        if last is not None and name is not None:
            tm[name] = tm.get(name, 0.0) + now - last
        self._mark_t = now

    # The three parts of a call.  _seg_pre and _seg_exec have fixed shapes, touch persistent state only IN PLACE and read no
    # Python-side counters, so they can be captured as graphs; _seg_plan is host logic (planner launches, pick-ups).
    def _seg_pre(self, ac):
        """policy action -> planner / direct, target, pull-back, straight-line pre-check; blocked envs join the pool"""
        torch = _torch()
        env, cfg, E, n = self.env, self.cfg, self.E, self.n
        mark = self._mark
        mark(None)
        if cfg.fused:
            return self._seg_pre_fused(ac)
        busy0 = self.busy.clone()
        active = ~busy0
        prev_ob = torch.where(busy0[:, None], self._pend_ob, env.obs)
        ac_tr = torch.where(busy0[:, None], self._pend_ac, ac[:, :self.ac_dim])
        cur = env.qpos.clone()
        if cfg.use_ik_target:
            # MoPA + IK: the action is Cartesian; its joint displacement decides planner / direct and IS the direct action
            a = self.ik_displacement(ac, cur)
            extra_ac = ac_tr[:, 7:7 + (env.action_dim - n)]
            mark("ik")
        else:
            a = ac[:, :n].contiguous()
            extra_ac = ac_tr[:, n:env.action_dim]
        if cfg.discrete_action:
            # the discrete head decides (rl/mopa_rollouts.py:86-88,106-111); a busy env's pending step keeps its own type
            ac_type = torch.where(busy0, self._pend_type, self._ac_type_in)
            is_pl = (ac_type != 0) & active
        else:
            ac_type = None
            is_pl = is_planner_action(a, cfg.omega) & active
        # Everything below works on all E rows with masks -- no index lists, so no host read-back of how many envs take which
        # branch.  Rows that are not planner actions carry a known-valid dummy state through the validity launches; nothing of
        # theirs is used.
        safe = self._safe_q
        if cfg.use_ik_target:
            # (with use_ik_target the reference never moves target_qpos off curr_qpos -- :113-131 is skipped and
            # `_cart2dispalcement` keeps its result local --, so a planner step plans from the current state to itself: two
            # zero-motion env steps.  Reproduced as is.)
            target = cur
        else:
            target = cur.clone()
            target[:, :n] += action_to_displacement(a, cfg.ac_scale, cfg.omega, cfg.action_range, cfg.ac_space_type)
            target = self.limits.clip_target(target)          # np.clip to the joint limits, unlimited entries restored (:121-131)
        target = torch.where(is_pl[:, None], target, safe).contiguous()
        if cfg.invalid_target_handling:
            target, _, tv = self.bp.pullback(torch.where(is_pl[:, None], cur, safe).contiguous(), target, cfg.step_size, cfg.num_trials)
            tv = tv.bool()
        else:
            tv = self._valid(target)
        mark("target")
        bad_target = (is_pl & ~tv).to(torch.int64)       # invalid target: success, valid, exact = False, False, True
        self.counters["mp_fail"] += bad_target
        self.counters["invalid"] += bad_target
        pv = is_pl & tv
        # ---- SACAgent.plan: straight-line pre-check for all of them in one launch (:198-204) ----
        cur_v = torch.where(pv[:, None], self.clip_qpos(cur), safe).contiguous()
        tgt_v = torch.where(pv[:, None], target, safe).contiguous()
        traj_i, tlen, succ, nst = simple_interpolate_batch(self.bp, cur_v, tgt_v, cfg.ac_scale, self.arm, fixed_steps=self._k_interp)
        # a line that needs more than the fixed width (cur outside its limits by more than action_range) is not executed as a
        # truncated walk: it goes the planner's way; drain() still reports it
        over = nst > self._k_interp
        self._interp_overflow |= over.any()
        succ = succ & ~over
        plan_ok = pv & succ
        self.counters["interpolation"] += plan_ok.to(torch.int64)
        traj_pad = torch.where(plan_ok[:, None, None], traj_i, torch.zeros_like(traj_i))
        path_len = torch.where(plan_ok, tlen.to(torch.int64), torch.zeros_like(tlen, dtype=torch.int64))
        mark("interpolate")
        # ---- the blocked ones go to RRT-Connect; lock-step waits for it below, async_planner does not ----
        blocked = pv & ~succ
        self._q_cur.copy_(torch.where(blocked[:, None], cur_v, self._q_cur))
        self._q_tgt.copy_(torch.where(blocked[:, None], tgt_v, self._q_tgt))
        self._pool_mask |= blocked
        self._wait_since.copy_(torch.where(blocked, self._t_dev.expand_as(self._wait_since), self._wait_since))
        self.busy |= blocked
        self._pend_ob.copy_(torch.where(blocked[:, None], prev_ob, self._pend_ob))
        self._pend_ac.copy_(torch.where(blocked[:, None], ac_tr, self._pend_ac))
        if cfg.discrete_action:
            self._pend_type.copy_(torch.where(blocked, ac_type, self._pend_type))
        return {"ac_type": ac_type, "active": active, "prev_ob": prev_ob, "ac_tr": ac_tr, "a": a, "extra_ac": extra_ac, "is_pl": is_pl, "plan_ok": plan_ok,
                "traj_pad": traj_pad, "path_len": path_len, "n_finished": 0}

    def _park_state(self, ps, jid, again):
        """the planner state (trees, counters) of a first-phase launch's unsolved queries -> the per-env parking buffers"""
        torch = _torch()
        from .batch import PlanState
        E, w = self.E, 2 * ps.max_nodes * ps.na
        if self._res is None:
            dev = ps.state.device
            self._res = PlanState(torch.empty(E * w + 8, dtype=torch.float64, device=dev), torch.empty(E * 2 * ps.max_nodes, dtype=torch.int32, device=dev),
                                  torch.zeros(E, 4, dtype=torch.int64, device=dev), ps.max_nodes, ps.na)
        rows = torch.nonzero(again).flatten()
        if len(rows) == 0:
            return
        env_rows = jid[rows]
        n = ps.state.shape[0]
        self._res.tree_q[:E * w].view(E, w)[env_rows] = ps.tree_q[:n * w].view(n, w)[rows]
        self._res.tree_p.view(E, 2 * ps.max_nodes)[env_rows] = ps.tree_p.view(n, 2 * ps.max_nodes)[rows]
        self._res.state[env_rows] = ps.state[rows]
        for x in (ps.tree_q, ps.tree_p, ps.state):
            x.record_stream(torch.cuda.current_stream())

    def _seg_plan(self, bag):
        """RRT-Connect launches for the pooled envs, pick-up of the launches that are done (host logic, dynamic shapes)"""
        torch = _torch()
        env, cfg, E, n = self.env, self.cfg, self.E, self.n
        dev = env.device
        mark = self._mark
        plan_ok, traj_pad, path_len = bag["plan_ok"], bag["traj_pad"], bag["path_len"]
        # One K3 launch takes about as long for 40 queries as for 4000 (its time is the latency of the slowest query), so the
        # waiting envs of several calls share a launch: a job starts only when a side stream has no job in flight (and only
        # then the waiting envs are listed: the one read-back of this part).
        two_phase = cfg.async_planner and 0 < cfg.planner_first_iters < self.main_iters
        for retry in ((False, True) if two_phase else (False,)):
            side = None
            if cfg.async_planner:
                used = {j["stream"] for j in self._jobs}
                free = [st for st in self._streams if st not in used]
                side = free[0] if free else None
            if side is None and cfg.async_planner:
                break
            mask = self._retry_mask if retry else self._pool_mask
            # (a launch costs the host about as much as a call: with several streams free, small pools wait for company
            # unless nothing of their kind is in flight)
            kind_in_flight = any(j.get("retry", False) == retry for j in self._jobs)
            min_job = cfg.planner_min_job // 4 if retry else cfg.planner_min_job
            known = self._pool_counts_known() if cfg.async_planner else None
            if known is not None:
                # listing the waiting envs reads their number back (the host waits for the stream): only in calls in which
                # the count an earlier call left behind says a launch is due
                if not (known[int(retry)] >= min_job or (known[int(retry)] > 0 and not kind_in_flight)):
                    continue
            cap = cfg.planner_job_cap // 4 if retry else cfg.planner_job_cap
            picked = None
            if cfg.fused and cfg.async_planner and mask.is_cuda:
                # the pick-up in one library launch (ids ascending as torch.nonzero lists them, mask cleared, start / target rows, step counts
                # and seeds gathered) when the count lies within [what a launch waits for, the launch's cap]; one read-back as before
                picked = self._pool_pick(mask, 1 if not kind_in_flight else min_job, cap)
                if picked[0] == 0 or (picked[1] is None and picked[0] <= cap):
                    continue          # nothing waits / too few: the pool waits for company
            if picked is not None and picked[1] is not None:
                n_p, bi, cur_p, tgt_p, steps_p, seeds_p = picked
                if bool(self._interp_overflow):
                    raise _lib.MopaError(f"a straight-line pre-check needed more than {self._k_interp} steps (targets further than "
                                         "action_range from the current state?)")
                iters = cfg.planner_first_iters if (two_phase and not retry) else self.main_iters
                res_in = None
                if retry and self._res is not None:
                    from .batch import PlanState
                    res_in = PlanState(self._res.tree_q, self._res.tree_p, self._res.state, cfg.max_nodes, self._res.na).rows(bi)
                chain = bool(cfg.planner_chain) and two_phase and not retry and side is not None and cfg.device_paths and self.nq <= 64
                job = self._rrt_launch(cur_p, tgt_p, bi, side, iters=iters, keep=two_phase and not retry and side is not None, resume=res_in,
                                       chain=chain, steps=steps_p, seeds=seeds_p)
                job["retry"] = retry
                self._jobs.append(job)
                continue
            bi = torch.nonzero(mask).flatten()
            if len(bi) and (not cfg.async_planner or len(bi) >= min_job or not kind_in_flight):
                if bool(self._interp_overflow):
                    raise _lib.MopaError(f"a straight-line pre-check needed more than {self._k_interp} steps (targets further than "
                                         "action_range from the current state?)")
                if cfg.async_planner and len(bi) > cap:
                    # the envs that have waited longest go first (a plain prefix would starve the high env indices whenever
                    # more envs wait than a launch takes)
                    bi = bi[torch.argsort(self._wait_since[bi], stable=True)[:cap]]
                bi = bi.contiguous()
                mask[bi] = False
                iters = cfg.planner_first_iters if (two_phase and not retry) else self.main_iters
                # a first-phase launch keeps its trees; the retry launch of its unsolved queries continues from them (parked per
                # env in between) instead of retracing the first iterations
                res_in = None
                if retry and self._res is not None:
                    from .batch import PlanState
                    res_in = PlanState(self._res.tree_q, self._res.tree_p, self._res.state, cfg.max_nodes, self._res.na).rows(bi)
                chain = bool(cfg.planner_chain) and two_phase and not retry and side is not None and cfg.device_paths and self.nq <= 64
                job = self._rrt_launch(self._q_cur[bi].contiguous(), self._q_tgt[bi].contiguous(), bi, side, iters=iters,
                                       keep=two_phase and not retry and side is not None, resume=res_in, chain=chain)
                job["retry"] = retry
                self._jobs.append(job)
        # ---- RRT-Connect jobs that are done (lock-step: all of them) ----
        finished = torch.zeros(E, dtype=torch.bool, device=dev) if self._jobs else None
        n_fin = 0
        still = []
        for job in self._jobs:
            if not cfg.async_planner:
                self._rrt_advance(job, wait=True)
            elif not self._rrt_advance(job, wait=False):
                still.append(job)
                continue
            tr_j, ln_j, s_j, v_j, e_j = job["result"]
            if job["stream"] is not None:
                # results allocated on the job's side stream are read by this stream's launches below: tell the allocator,
                # or the next job could be handed their memory while those launches are still queued
                for x in (tr_j, ln_j, s_j, v_j, e_j):
                    if hasattr(x, "record_stream"):
                        x.record_stream(torch.cuda.current_stream())
            jid = job["ids"]
            t = lambda x: torch.as_tensor(x, device=dev)
            s_t, v_t, e_t = t(s_j), t(v_j), t(e_j)
            # (everything below is index arithmetic over the job's rows with a `fin` mask -- no boolean-mask indexing: each `x[mask]` is a
            #  torch.nonzero underneath, three launches, a memset and a read-back the host waits for)
            fin = None
            if job["iters"] < self.main_iters:
                # first-phase launch: "no exact solution" may only mean that the short budget ran out -- those queries run
                # again with the full budget (their envs stay busy); everything else is final
                again = ~s_t & ~e_t
                if "chain" in job:
                    # their continuation is already running behind this launch on its stream: a job of its own, finished at
                    # the second event
                    rows = torch.nonzero(again).flatten()
                    if len(rows):
                        lazy = {"ids": jid[rows], "cur": job["cur"][rows], "target": job["target"][rows], "steps": job["steps"][rows],
                                "event": job["chain"]["event"], "stage": "rrt", "stream": job["stream"], "iters": self.main_iters,
                                "retry": True, "lazy": (job["chain"], rows)}
                        # its tensors come from THIS stream's ops (nonzero, gathers) and are read on the side stream later:
                        # an event the side stream waits for, and the allocator told about the second reader
                        lazy["built"] = torch.cuda.Event()
                        lazy["built"].record(torch.cuda.current_stream())
                        if job["stream"] is not None:
                            for x in (lazy["ids"], lazy["cur"], lazy["target"], lazy["steps"], rows):
                                x.record_stream(job["stream"])
                        still.append(lazy)
                else:
                    self._retry_mask[jid] = self._retry_mask[jid] | again
                    if "pstate" in job:
                        self._park_state(job["pstate"], jid, again)
                self.n_retried = self.n_retried + again.sum()
                fin = ~again
            ok_f = s_t if fin is None else (s_t & fin)
            bad_f = ~s_t if fin is None else (~s_t & fin)
            finished[jid] = True if fin is None else fin
            n_fin += 1
            plan_ok[jid] = s_t if fin is None else torch.where(fin, s_t, plan_ok[jid])
            self.counters["mp"].index_add_(0, jid, ok_f.to(self.counters["mp"].dtype))
            self.counters["mp_fail"].index_add_(0, jid, bad_f.to(self.counters["mp_fail"].dtype))
            self.counters["approximate"].index_add_(0, jid, (bad_f & ~e_t).to(self.counters["approximate"].dtype))
            self.counters["invalid"].index_add_(0, jid, (bad_f & ~v_t).to(self.counters["invalid"].dtype))
            ln_t = t(ln_j)
            if fin is not None:      # (the rows of the queries that run again keep their length 0: their trajectory rows are never read)
                ln_t = torch.where(fin, ln_t.to(path_len.dtype), path_len[jid])
            traj_pad, path_len = self._merge_paths(traj_pad, path_len, t(tr_j), ln_t, jid)
            self.busy[jid] = False if fin is None else (self.busy[jid] & ~fin)
        self._jobs = still
        mark("plan")
        bag.update(plan_ok=plan_ok, traj_pad=traj_pad, path_len=path_len, finished=finished, n_finished=n_fin)

    def _seg_exec(self, bag, record: bool = False):
        """direct steps, failed-plan steps and waypoint execution; counters and the returned transition"""
        torch = _torch()
        env, cfg, E, n = self.env, self.cfg, self.E, self.n
        dev = env.device
        mark = self._mark
        active, is_pl, a, extra_ac = bag["active"], bag["is_pl"], bag["a"], bag["extra_ac"]
        plan_ok, traj_pad, path_len, prev_ob, ac_tr = bag["plan_ok"], bag["traj_pad"], bag["path_len"], bag["prev_ob"], bag["ac_tr"]
        finished = bag.get("finished")
        if finished is None:
            finished = torch.zeros(E, dtype=torch.bool, device=dev)
        chunked = cfg.walk_chunk > 0 and getattr(env, "dynamics", False)
        if cfg.fused and not chunked:
            return self._seg_exec_fused(bag, record)
        direct = active & ~is_pl
        self.counters["rl"] += direct.to(torch.int64)
        sitting = self.busy & ~finished            # still waiting for their query (or on a walk): nothing of theirs is touched
        stepped = ~sitting
        # ---- direct execution (:336-356) and failed plans (:303-334: reward of the current state, one env step) in one launch
        # (:349-352: with the discrete head the direct action goes to the env as it is, otherwise rescaled by 1 / omega)
        act0 = torch.where(direct[:, None], a if cfg.discrete_action else a / torch.full_like(a, cfg.omega), torch.zeros_like(a))
        if env.action_dim > n:          # Lift: the gripper entry is passed through unscaled (:340-343)
            act0 = torch.cat([act0, torch.where(direct[:, None], extra_ac, torch.zeros_like(extra_ac))], dim=1)
        act0 = act0.contiguous()
        flags = torch.where(direct, 1, torch.where(plan_ok | sitting, 2, 0)).to(torch.uint8).contiguous()
        is_pl_out = is_pl | finished
        last_extra = extra_ac[:, 0].contiguous() if env.action_dim > n else None
        rec = None
        if chunked:
            # (the direct / failed-plan steps share a launch with the first waypoint of every env on a walk)
            rew, done, intra, rec, path_len, plan_ok, is_pl_out, stepped = self._walk_chunk(bag, act0, flags, record, sitting, is_pl_out, last_extra)
        else:
            env._launch(act0, False, flags)
            rew = torch.where(plan_ok, torch.zeros_like(env.reward), env.reward)
            done = torch.where(plan_ok, torch.zeros_like(env.done), env.done)
            intra = torch.zeros(E, dtype=torch.int64, device=dev)
            # ---- waypoint execution (:152-199)
            L = traj_pad.shape[1]
            if record:
                rec = {"ob": torch.zeros(E, L, env.obs.shape[1], dtype=torch.float64, device=dev),
                       "meta_rew": torch.zeros(E, L, dtype=torch.float64, device=dev),
                       "done": torch.zeros(E, L, dtype=torch.uint8, device=dev),
                       "waypoint": traj_pad, "n_exec": torch.zeros(E, dtype=torch.int64, device=dev)}
            # one launch: every env walks its own waypoints until its path ends or a step reports done (envs without a path: length 0)
            disc = self._disc_pow(L)
            rew, done = rew.contiguous(), done.to(torch.uint8).contiguous()
            env.exec_trajectories(traj_pad.contiguous(), torch.where(plan_ok, path_len, torch.zeros_like(path_len)).contiguous(), disc,
                                  rew, done, intra, rec={k: rec[k] for k in ("ob", "meta_rew", "done", "n_exec")} if rec else None,
                                  last_extra=last_extra)
        mark("execute")
        if chunked:
            env.has_prev.copy_(torch.where(self._walk["on"], env.has_prev, torch.zeros_like(env.has_prev)))
        else:
            env.has_prev.zero_()                                 # env._reset_prev_state()
        self.t_env += stepped.to(torch.int64)
        self._t_dev += 1
        res = {"ob": prev_ob, "ac": ac_tr, "ob_next": env.obs.clone(), "rew": rew, "done": done, "intra_steps": intra,
               "is_planner": is_pl_out, "success": env.success.clone(), "path_len": path_len, "plan_ok": plan_ok, "stepped": stepped}
        if cfg.discrete_action:
            res["ac_type"] = bag["ac_type"]
        if rec is not None:
            res["record"] = rec
        return res

    # ---- the same two parts with the elementwise work in the library's bookkeeping kernels (cfg.fused) ----
    def _fused_struct(self):
        torch = _torch()
        S = getattr(self, "_fs", None)
        if S is not None:
            return S
        import ctypes as C
        env, cfg = self.env, self.cfg
        if _lib.lib().mopa_rollout_step_size() != C.sizeof(_lib.MopaRolloutStep):
            raise _lib.MopaError("MopaRolloutStep: the binding's layout differs from the library's")
        S = self._fs = _lib.MopaRolloutStep()
        S.E, S.nq, S.n_arm, S.ac_dim, S.adim, S.obs_dim, S.K = self.E, self.nq, self.n, self.ac_dim, env.action_dim, env.obs.shape[1], self._k_interp
        S.discrete, S.normal_space = int(cfg.discrete_action), int(cfg.ac_space_type == "normal")
        if cfg.ac_space_type not in ("normal", "piecewise"):
            raise NotImplementedError(cfg.ac_space_type)
        S.omega, S.ac_scale, S.action_range = cfg.omega, cfg.ac_scale, cfg.action_range
        S.omega_over_scale, S.one_minus_omega, S.range_minus_scale = cfg.omega / cfg.ac_scale, 1 - cfg.omega, cfg.action_range - cfg.ac_scale
        lim = self.limits
        self._fs_keep = [t.contiguous() for t in (lim.lo, lim.hi, lim.lo_state, lim.hi_state, lim.lo_shrunk, lim.hi_shrunk, self._safe_q.reshape(-1))]
        for k, t in zip(("lim_lo", "lim_hi", "lo_state", "hi_state", "lo_shrunk", "hi_shrunk", "safe_q"), self._fs_keep):
            setattr(S, k, t.data_ptr())
        if self._interp_overflow.dtype != torch.bool:
            raise _lib.MopaError("interp_overflow must be a bool scalar")
        return S

    def _fused_bind(self, **tensors):
        """device addresses of this call's tensors -> the step struct (None -> NULL); every tensor must be contiguous"""
        S = self._fs
        for k, t in tensors.items():
            if t is None:
                setattr(S, k, None)
                continue
            if not t.is_contiguous():
                raise _lib.MopaError(f"rollout step buffer {k} is not contiguous")
            setattr(S, k, t.data_ptr())

    def _pool_pick(self, mask, min_n: int, cap: int):
        """mopa_rollout_pool_pick on the current stream: (count, ids, start rows, target rows, step counts, seeds) of the envs whose `mask` byte is
        set -- the last five None when the count is outside [min_n, cap] (nothing was touched then)."""
        torch = _torch()
        from .batch import _stream_handle
        dev = mask.device
        ids = torch.empty(cap, dtype=torch.int64, device=dev)
        cur = torch.empty(cap, self.nq, dtype=torch.float64, device=dev)
        tgt = torch.empty(cap, self.nq, dtype=torch.float64, device=dev)
        steps = torch.empty(cap, dtype=torch.int64, device=dev)
        seeds = torch.empty(cap, dtype=torch.int64, device=dev)
        cnt = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().mopa_rollout_pool_pick(self.E, self.nq, int(min_n), int(cap), mask.data_ptr(), self._q_cur.data_ptr(), self._q_tgt.data_ptr(),
                                                     self.t_env.data_ptr(), int(self.cfg.seed), ids.data_ptr(), cur.data_ptr(), tgt.data_ptr(),
                                                     steps.data_ptr(), seeds.data_ptr(), cnt.data_ptr(), _stream_handle(None)))
        n = int(cnt.item())
        if n < min_n or n > cap:
            return n, None, None, None, None, None
        return n, ids[:n], cur[:n], tgt[:n], steps[:n], seeds[:n]

    def _pool_counts_known(self):
        """(envs waiting for a planner launch, for a retry launch) as of the latest earlier call whose bookkeeping has finished
        on the device, or None (torch form of the calls / no such call yet): copied to pinned memory behind each call"""
        ring = getattr(self, "_count_ring", None)
        if not ring:
            return None
        for k in range(len(ring) - 1, -1, -1):
            if ring[k][0].query():
                self._count_free.extend(ring[:k])             # older entries (finished before this one, same stream) are recycled
                del ring[:k]
                return [int(x) for x in ring[0][1]]
        return None

    def _pool_counts_push(self):
        torch = _torch()
        ring = getattr(self, "_count_ring", None)
        if ring is None:
            ring = self._count_ring = []
            self._count_free = []
        # at most two calls enqueued ahead of the device: the planner's pick-ups are decided on the host's clock, and a host
        # far ahead would enqueue calls in which envs sit out although their query has long finished
        pending = [r for r in ring if not r[0].query()]
        if len(pending) >= 2:
            pending[0][0].synchronize()
        while len(ring) > 4:
            self._count_free.append(ring.pop(0))
        ev, host = self._count_free.pop() if self._count_free else (torch.cuda.Event(), torch.zeros(2, dtype=torch.int64).pin_memory())
        host.copy_(self._pool_counts, non_blocking=True)
        ev.record(torch.cuda.current_stream())
        ring.append((ev, host))

    def _fused_stage(self, stage):
        import ctypes as C
        from .batch import _stream_handle
        _lib.check(_lib.lib().mopa_rollout_stage(C.byref(self._fs), int(stage), _stream_handle(None)))

    def _seg_pre_fused(self, ac):
        torch = _torch()
        from .batch import _ptr, _stream_handle
        env, cfg, E, n, nq = self.env, self.cfg, self.E, self.n, self.nq
        dev, f64, u8, i64 = env.device, torch.float64, torch.bool, torch.int64
        mark = self._mark
        S = self._fused_struct()
        ac = ac if ac.is_contiguous() else ac.contiguous()
        if ac.dtype != f64 or ac.shape[0] != E or ac.shape[1] < self.ac_dim:
            raise _lib.MopaError("agent_step: ac must be float64 [E, >= ac_dim]")
        a_in = None
        if cfg.use_ik_target:
            a_in = self.ik_displacement(ac, env.qpos.clone()).contiguous()
            mark("ik")
        em = lambda *sh, dt=f64: torch.empty(*sh, dtype=dt, device=dev)
        n_extra = env.action_dim - n
        B = {"prev_ob": em(E, S.obs_dim), "ac_tr": em(E, self.ac_dim), "a": em(E, n), "extra_ac": em(E, max(n_extra, 1)),
             "target": em(E, nq), "cur_m": em(E, nq), "cur_v": em(E, nq), "tgt_v": em(E, nq), "traj": em(E, self._k_interp + 1, nq),
             "active": em(E, dt=u8), "is_pl": em(E, dt=u8), "pv": em(E, dt=u8), "plan_ok": em(E, dt=u8),
             "ac_type": em(E, dt=i64) if cfg.discrete_action else None, "path_len": em(E, dt=i64)}
        S.ac_stride = int(ac.shape[1])
        c = self.counters
        self._fused_bind(qpos=env.qpos, obs=env.obs, reward=env.reward, done=env.done, success=env.success, has_prev=env.has_prev,
                         busy=self.busy, pool_mask=self._pool_mask, interp_overflow=self._interp_overflow, wait_since=self._wait_since,
                         t_dev=self._t_dev, t_env=self.t_env, pend_type=self._pend_type, q_cur=self._q_cur, q_tgt=self._q_tgt,
                         pend_ob=self._pend_ob, pend_ac=self._pend_ac, c_rl=c["rl"], c_interp=c["interpolation"], c_mp_fail=c["mp_fail"],
                         c_invalid=c["invalid"], ac=ac, ac_type_in=self._ac_type_in, a_in=a_in, **B)
        self._fused_stage(0)
        if cfg.invalid_target_handling:
            trials = em(E, dt=torch.int32)
            tv = em(E, dt=torch.uint8)
            _lib.check(_lib.lib().mopa_pullback_batch(self.scene.handle, _ptr(B["cur_m"]), _ptr(B["target"]), E, float(cfg.step_size), int(cfg.num_trials),
                                                      _ptr(trials), _ptr(tv), _stream_handle(None)))
        else:
            tv = self._valid(B["target"]).contiguous()
        mark("target")
        self._fused_bind(tv=tv)
        self._fused_stage(1)
        tlen, ok, nst = em(E, dt=torch.int32), em(E, dt=torch.uint8), em(E, dt=torch.int32)
        _lib.check(_lib.lib().mopa_interpolate_batch(self.scene._h, E, n, self._k_interp, _ptr(B["cur_v"]), _ptr(B["tgt_v"]), float(cfg.ac_scale),
                                                     _ptr(B["traj"]), _ptr(tlen), _ptr(ok), _ptr(nst), _stream_handle(None)))
        self._fused_bind(tlen=tlen, ok=ok, nst=nst)
        self._fused_stage(2)
        mark("interpolate")
        self._fs_call = (ac, a_in, tv, tlen, ok, nst)          # alive until the call's last stage has been enqueued
        return {"ac_type": B["ac_type"], "active": B["active"], "prev_ob": B["prev_ob"], "ac_tr": B["ac_tr"], "a": B["a"],
                "extra_ac": B["extra_ac"][:, :n_extra] if n_extra else B["extra_ac"][:, :0], "is_pl": B["is_pl"], "plan_ok": B["plan_ok"],
                "traj_pad": B["traj"], "path_len": B["path_len"], "n_finished": 0, "_extra_buf": B["extra_ac"]}

    def _seg_exec_fused(self, bag, record: bool = False):
        torch = _torch()
        env, cfg, E, n = self.env, self.cfg, self.E, self.n
        dev, f64, u8, i64 = env.device, torch.float64, torch.bool, torch.int64
        mark = self._mark
        S = self._fs
        em = lambda *sh, dt=f64: torch.empty(*sh, dtype=dt, device=dev)
        traj_pad, plan_ok, path_len = bag["traj_pad"], bag["plan_ok"], bag["path_len"]
        lift = env.action_dim > n
        X = {"act0": em(E, env.action_dim), "flags": em(E, dt=torch.uint8), "sitting": em(E, dt=u8), "stepped": em(E, dt=u8),
             "is_pl_out": em(E, dt=u8), "plen_m": em(E, dt=i64), "last_extra": em(E) if lift else None, "rew": em(E),
             "done_out": em(E, dt=torch.uint8), "intra": em(E, dt=i64), "ob_next": em(E, S.obs_dim), "success_out": em(E, dt=env.success.dtype)}
        # (the planner's pick-ups may have replaced plan_ok / path_len rows in place and widened traj_pad)
        counting = cfg.async_planner and not cfg.use_graphs
        if counting and getattr(self, "_pool_counts", None) is None:
            self._pool_counts = torch.zeros(2, dtype=i64, device=dev)
        self._fused_bind(finished=bag.get("finished"), plan_ok=plan_ok, path_len=path_len, active=bag["active"], is_pl=bag["is_pl"], a=bag["a"],
                         extra_ac=bag["_extra_buf"], retry_mask=self._retry_mask if counting else None,
                         pool_counts=self._pool_counts if counting else None, **X)
        self._fused_stage(3)
        env._launch(X["act0"], False, X["flags"])
        self._fused_stage(4)
        rec = None
        L = traj_pad.shape[1]
        if record:
            rec = {"ob": torch.zeros(E, L, env.obs.shape[1], dtype=f64, device=dev), "meta_rew": torch.zeros(E, L, dtype=f64, device=dev),
                   "done": torch.zeros(E, L, dtype=torch.uint8, device=dev), "waypoint": traj_pad, "n_exec": torch.zeros(E, dtype=i64, device=dev)}
        env.exec_trajectories(traj_pad if traj_pad.is_contiguous() else traj_pad.contiguous(), X["plen_m"], self._disc_pow(L), X["rew"], X["done_out"],
                              X["intra"], rec={k: rec[k] for k in ("ob", "meta_rew", "done", "n_exec")} if rec else None, last_extra=X["last_extra"])
        mark("execute")
        self._fused_stage(5)
        if counting:
            self._pool_counts_push()
        self._fs_call = None
        res = {"ob": bag["prev_ob"], "ac": bag["ac_tr"], "ob_next": X["ob_next"], "rew": X["rew"], "done": X["done_out"], "intra_steps": X["intra"],
               "is_planner": X["is_pl_out"], "success": X["success_out"], "path_len": path_len, "plan_ok": plan_ok, "stepped": X["stepped"]}
        if cfg.discrete_action:
            res["ac_type"] = bag["ac_type"]
        if rec is not None:
            res["record"] = rec
        return res

    def _disc_pow(self, L):
        disc = self._disc.get(L)
        if disc is None:
            disc = self._disc[L] = _torch().tensor([self.cfg.discount_factor ** k for k in range(L)], dtype=_torch().float64, device=self.env.device)
        return disc

    def _walk_chunk(self, bag, act0, flags0, record, sitting, is_pl_out, last_extra):
        """cfg.walk_chunk (dynamics env): the paths handed out in this call join the persistent walk state; every env on a walk
        advances by at most walk_chunk waypoints, each a physics launch over all envs on a walk -- the first of them also
        carries this call's direct and failed-plan steps (act0 / flags0).  The envs whose walk ended complete their agent
        step, the others stay busy.  Per env the arithmetic is that of one uninterrupted walk."""
        torch = _torch()
        env, cfg, E, n = self.env, self.cfg, self.E, self.n
        dev = env.device
        plan_ok, traj_pad, path_len = bag["plan_ok"], bag["traj_pad"], bag["path_len"]
        L = int(traj_pad.shape[1])
        W = getattr(self, "_walk", None)
        if W is None:
            z = lambda *sh, dt=torch.float64: torch.zeros(*sh, dtype=dt, device=dev)
            W = self._walk = {"on": z(E, dt=torch.bool), "traj": z(E, L, self.nq), "len": z(E, dt=torch.int64), "pos": z(E, dt=torch.int64),
                              "rew": z(E), "done": z(E, dt=torch.uint8), "intra": z(E, dt=torch.int64), "is_pl": z(E, dt=torch.bool),
                              "extra": z(E), "rec": None}
        if L > W["traj"].shape[1]:                                   # a longer path than any before: widen (rare)
            grow = lambda x: torch.cat([x, torch.zeros(E, L - x.shape[1], *x.shape[2:], dtype=x.dtype, device=dev)], dim=1)
            W["traj"] = grow(W["traj"])
            if W["rec"] is not None:
                for k in ("ob", "meta_rew", "done"):
                    W["rec"][k] = grow(W["rec"][k])
        Lw = int(W["traj"].shape[1])
        if L < Lw:
            traj_pad = torch.cat([traj_pad, torch.zeros(E, Lw - L, self.nq, dtype=traj_pad.dtype, device=dev)], dim=1)
        on0 = W["on"]
        new = plan_ok & ~sitting                                     # paths handed out in this call (straight lines, finished queries)
        zi = torch.zeros_like(W["len"])
        W["traj"].copy_(torch.where(new[:, None, None], traj_pad, W["traj"]))
        W["len"].copy_(torch.where(new, torch.clamp(path_len, max=Lw), torch.where(on0, W["len"], zi)))
        W["pos"].copy_(torch.where(on0, W["pos"], zi))
        W["rew"].copy_(torch.where(on0, W["rew"], torch.zeros_like(W["rew"])))
        W["done"].copy_(torch.where(on0, W["done"], torch.zeros_like(W["done"])))
        W["intra"].copy_(torch.where(on0, W["intra"], zi))
        W["is_pl"].copy_(torch.where(on0, W["is_pl"], is_pl_out))
        if last_extra is not None:
            W["extra"].copy_(torch.where(new, last_extra, W["extra"]))
        rec = None
        if record:
            if W["rec"] is None:
                W["rec"] = {"ob": torch.zeros(E, Lw, env.obs.shape[1], dtype=torch.float64, device=dev),
                            "meta_rew": torch.zeros(E, Lw, dtype=torch.float64, device=dev),
                            "done": torch.zeros(E, Lw, dtype=torch.uint8, device=dev), "n_exec": torch.zeros(E, dtype=torch.int64, device=dev)}
            rec = W["rec"]
            fresh = ~on0
            rec["ob"].copy_(torch.where(fresh[:, None, None], torch.zeros_like(rec["ob"]), rec["ob"]))
            rec["meta_rew"].copy_(torch.where(fresh[:, None], torch.zeros_like(rec["meta_rew"]), rec["meta_rew"]))
            rec["done"].copy_(torch.where(fresh[:, None], torch.zeros_like(rec["done"]), rec["done"]))
            rec["n_exec"].copy_(torch.where(fresh, torch.zeros_like(rec["n_exec"]), rec["n_exec"]))
        disc = self._disc_pow(Lw)
        ext = W["extra"] if last_extra is not None else None
        # the launch runs with planner-step semantics (the action IS the displacement): a direct action's arm entries are
        # multiplied by ac_scale here instead of in the kernel -- the same one multiplication
        other = act0.clone()
        other[:, :n] = act0[:, :n] * torch.full_like(act0[:, :n], cfg.ac_scale)
        walked = env.walk_round(W["traj"], W["len"], W["pos"], disc, W["rew"], W["done"], W["intra"], rec, ext, other_action=other, other_flags=flags0)
        rew1, done1 = env.reward.clone(), env.done.clone()           # of the envs that took their direct / failed-plan step
        if cfg.walk_chunk > 1:
            env.exec_trajectories(W["traj"], W["len"], disc, W["rew"], W["done"], W["intra"], rec=rec, last_extra=ext, pos=W["pos"],
                                  chunk=cfg.walk_chunk - 1)
        still = W["pos"] < W["len"]
        # envs on a walk are busy: the following calls ignore the policy's rows for them; the transition's first half is parked
        # exactly as for envs waiting for a query
        self._pend_ob.copy_(torch.where(still[:, None], bag["prev_ob"], self._pend_ob))
        self._pend_ac.copy_(torch.where(still[:, None], bag["ac_tr"], self._pend_ac))
        if cfg.discrete_action:
            self._pend_type.copy_(torch.where(still, bag["ac_type"], self._pend_type))
        self.busy.copy_((self.busy & ~on0) | still)
        W["on"] = still
        stepped = (~sitting | on0) & ~still
        walker = on0 | new
        out_rec = None
        if record:
            out_rec = {k: v.clone() for k, v in rec.items()}
            out_rec["waypoint"] = W["traj"].clone()
        return (torch.where(walker, W["rew"], rew1), torch.where(walker, W["done"], done1.to(torch.uint8)), W["intra"].clone(), out_rec,
                torch.where(walker, W["len"], path_len), plan_ok | on0, W["is_pl"].clone(), stepped)

    def drain(self):
        """wait for every RRT-Connect launch in flight (async_planner); their envs complete their step in a following agent_step"""
        for job in self._jobs:
            if job["event"] is not None:
                job["event"].synchronize()


    def run_episode(self, policy, max_step: int = 10000, is_train: bool = True, random_exploration: bool = False, reset: bool = True):
        """`MoPARolloutRunner.run_episode` (reference rl/mopa_rollouts.py:401-678: the evaluation loop -- ONE episode, `while not done
        and ep_len < max_step`, every agent step the same routing as `run`) for all E envs at once: each env runs one episode from
        `env.reset()` (`reset=False`: from the state the env is in) and sits out once its episode is over (it gets the zero action
        and nothing of it is recorded any more).  Lock-step only: a call is one agent step of every env still in its episode.

        policy(ob [E, obs_dim], is_train=..., random_exploration=...) -> ac [E, >= ac_dim] (float64, in [-1, 1]; with
        `discrete_action` a pair (ac, ac_type [E])) -- the batched `pi.act` (:432-437).

        What differs from `run`, as in the reference: the per-step reward of a planner step is the PLAIN sum of its waypoints'
        rewards (`meta_rew += reward`, :540 -- not the discounted SMDP return of :171), a failed plan is one step with the current
        reward (:594-611), no `reuse_data` transitions.  Returns (rollout, ep_info):
          rollout  ob [T + 1, E, obs_dim] (row t = the obs an env's t-th agent step started from; the row after its last step = the
                   final obs: `rollout.add({"ob": ll_ob})`, :661), ac [T, E, ac_dim], rew [T, E], done [T, E] uint8, valid [T, E] bool
                   (env e took an agent step in call t), n_steps [E], qpos_final [E, nq] (the state each episode ended in) -- the reference's
                   evaluation rollout holds ob / ac of planner steps only (:577-585 against :648-653), this one of every step;
          ep_info  len, rew (:663-670), success (the env's `episode_success`), and the six counters (:671), one entry per env.
        `contact_force` (:538, MuJoCo's contact solver forces) has no kinematic counterpart and is not reported.  `max_step` below the
        env's own episode cap would cut a path between two waypoints (:575): build the env with that cap instead."""
        torch = _torch()
        env, cfg, E = self.env, self.cfg, self.E
        if cfg.async_planner:
            raise _lib.MopaError("run_episode is the lock-step evaluation loop (async_planner=False)")
        if max_step < env.max_episode_steps:
            raise _lib.MopaError(f"max_step {max_step} < the env's max_episode_steps {env.max_episode_steps}: make the env with that episode cap")
        dev = env.device
        if reset:
            env.reset()
        gamma, disc_cache = cfg.discount_factor, self._disc
        cfg.discount_factor, self._disc = 1.0, {}
        try:
            alive = torch.ones(E, dtype=torch.bool, device=dev)
            ep_len = torch.zeros(E, dtype=torch.int64, device=dev)
            ep_rew = torch.zeros(E, dtype=torch.float64, device=dev)
            cnt = {k: torch.zeros(E, dtype=torch.int64, device=dev) for k in COUNTERS}
            success = torch.zeros(E, dtype=torch.bool, device=dev)
            obs, acs, rews, dones, valids = [], [], [], [], []
            last_ob, last_q = env.obs.clone(), env.qpos.clone()
            while True:
                got = policy(env.obs.clone(), is_train=is_train, random_exploration=random_exploration)
                ac, ac_type = got if cfg.discrete_action else (got, None)
                ac = torch.where(alive[:, None], ac.to(torch.float64), torch.zeros_like(ac, dtype=torch.float64)).contiguous()
                if ac_type is not None:
                    ac_type = torch.where(alive, ac_type.reshape(-1).to(torch.int64), torch.zeros(E, dtype=torch.int64, device=dev))
                before = {k: self.counters[k].clone() for k in COUNTERS}
                out = self.agent_step(ac, ac_type=ac_type)
                ep_len += torch.where(alive, out["intra_steps"] + 1, torch.zeros_like(ep_len))
                ep_rew += torch.where(alive, out["rew"], torch.zeros_like(ep_rew))
                for k in COUNTERS:
                    cnt[k] += torch.where(alive, self.counters[k] - before[k], torch.zeros_like(cnt[k]))
                success |= alive & out["success"].bool()
                obs.append(torch.where(alive[:, None], out["ob"], last_ob))
                acs.append(out["ac"].clone())
                rews.append(torch.where(alive, out["rew"], torch.zeros_like(out["rew"])))
                dones.append(torch.where(alive, out["done"].to(torch.uint8), torch.zeros_like(out["done"], dtype=torch.uint8)))
                valids.append(alive.clone())
                last_ob = torch.where(alive[:, None], out["ob_next"], last_ob)
                last_q = torch.where(alive[:, None], env.qpos, last_q)
                alive = alive & ~(out["done"].bool() | (ep_len >= max_step))
                if not bool(alive.any()):         # (one host read per agent step: this is the evaluation loop, not the training one)
                    break
            obs.append(last_ob)
        finally:
            cfg.discount_factor, self._disc = gamma, disc_cache
        valid = torch.stack(valids)
        # row n_steps[e] of `ob` = env e's final obs; rows beyond it repeat it
        ob_t = torch.stack(obs)
        rollout = {"ob": ob_t, "ac": torch.stack(acs), "rew": torch.stack(rews), "done": torch.stack(dones), "valid": valid,
                   "n_steps": valid.sum(0), "qpos_final": last_q}
        ep_info = {"len": ep_len, "rew": ep_rew, "success": success, **cnt}
        return rollout, ep_info
