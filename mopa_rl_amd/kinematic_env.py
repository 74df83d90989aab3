"""Batched KINEMATIC SawyerPushObstacle env on libmopa_hip.so.

Restates the reference env's arithmetic around the physics -- action scaling and `desired_state`
(env/sawyer/sawyer_push_obstacle.py:162-208), joint-limit clamp + episode bookkeeping (env/base.py:269-314),
reward/success (sawyer_push_obstacle.py:71-104), the 40-number observation in dict order (env/sawyer/sawyer.py:317-338,
sawyer_push_obstacle.py:106-119) and `_reset` (sawyer_push_obstacle.py:36-52) -- and replaces `_do_simulation`
(75 MuJoCo sub-steps of a position servo) by its kinematic limit: the arm reaches `desired_state`, velocities are
zero, nothing else moves.  **Not dynamics parity**: no contact forces, the cube never moves, so the push reward can
only be collected by a policy in the real env.  What it is for: the "env-steps/sec" half of the metric, and the
planner-side rollouts of MoPA-RL (a planned, collision-checked joint path is executed kinematically by construction).

`block_invalid=True` adds the one piece of contact behaviour a kinematic arm can have: a step whose desired state is
in collision (K1 validity kernel, same rule as the planner) is not executed -- the arm stays where it is.

State lives in torch tensors on the GPU; every step is one launch of `k_env_step` (plus one validity launch with
`block_invalid`).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .batch import BatchPlanner, _ptr, _stream_handle, _torch
from .scene import ENV_SPECS, load_scene, planner_inputs, qpos_joint_arrays

OBS_DIM = 40
# observation layout == the reference's OrderedDict order (sawyer.py:317-338 then sawyer_push_obstacle.py:106-119)
OBS_LAYOUT = OrderedDict([
    ("joint_pos", 7), ("joint_vel", 7), ("gripper_qpos", 2), ("gripper_qvel", 2), ("eef_pos", 3), ("eef_quat", 4),
    ("target_pos", 3), ("cube_pos", 3), ("cube_quat", 4), ("gripper_to_cube", 3), ("cube_to_target", 2)])


@dataclass
class PushEnvFacts:
    """Name -> id resolution of what SawyerPushObstacleEnv._get_reference looks up (sawyer.py:164-199,
    sawyer_push_obstacle.py:17-30), on a CompiledModel."""
    arm_qpos_idx: np.ndarray
    grip_qpos_idx: np.ndarray
    target_qpos_idx: np.ndarray
    eef_body: int
    eef_off: np.ndarray
    rfinger_body: int
    rfinger_off: np.ndarray
    lfinger_body: int
    lfinger_off: np.ndarray
    ee_quat_body: int
    cube_body: int
    target_body: int
    qpos_min: np.ndarray
    qpos_max: np.ndarray
    qpos_limited: np.ndarray


def push_env_facts(model) -> PushEnvFacts:
    m = model
    spec = ENV_SPECS["SawyerPushObstacle-v0"]

    def site(name):
        i = m.site_name2id(name)
        return int(m.site_body[i]), np.asarray(m.site_pos[i], dtype=np.float64).copy()

    eb, eo = site("grip_site")
    rb, ro = site("right_eef")
    lb, lo = site("left_eef")
    jidx, jlo, jhi, jlim = qpos_joint_arrays(m)
    return PushEnvFacts(
        arm_qpos_idx=np.array([m.get_joint_qpos_addr(j) for j in spec.robot_joints], dtype=np.int32),
        grip_qpos_idx=np.array([m.get_joint_qpos_addr(j) for j in ("rc_close", "lc_close")], dtype=np.int32),
        target_qpos_idx=np.array([m.get_joint_qpos_addr(j) for j in ("target_x", "target_y")], dtype=np.int32),
        eef_body=eb, eef_off=eo, rfinger_body=rb, rfinger_off=ro, lfinger_body=lb, lfinger_off=lo,
        ee_quat_body=m.body_names.index("right_ee_attchment"), cube_body=m.body_names.index("cube"),
        target_body=m.body_names.index("target"),
        qpos_min=jlo[jidx].astype(np.float64), qpos_max=jhi[jidx].astype(np.float64),
        qpos_limited=jlim[jidx].astype(np.int32))


class BatchKinematicPushEnv:
    """E SawyerPushObstacle envs stepped kinematically on one GPU."""

    env_name = "SawyerPushObstacle-v0"

    def __init__(self, num_envs: int, device=None, seed: int = 0, max_episode_steps: int = 250,
                 distance_threshold: float = 0.06, success_reward: float = 150.0, ac_scale: Optional[float] = None,
                 block_invalid: bool = False, model=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.MopaError("BatchKinematicPushEnv needs a HIP device (there is no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.spec = ENV_SPECS[self.env_name]
        self.model = model if model is not None else load_scene(self.spec.scene)
        self.facts = push_env_facts(self.model)
        self.E = int(num_envs)
        self.nq = self.model.nq
        self.n_arm = len(self.facts.arm_qpos_idx)
        self.max_episode_steps = int(max_episode_steps)
        self.ac_scale = float(self.spec.ac_scale if ac_scale is None else ac_scale)
        L = _lib.lib()
        f = self.facts
        keep = []
        desc = _lib.MopaEnvDesc()
        desc.model = _lib.model_struct(self.model, keep)

        def ip(a):
            a, p = _lib._i(a); keep.append(a); return p

        def dp(a):
            a, p = _lib._d(a); keep.append(a); return p

        desc.n_arm, desc.arm_qpos_idx = self.n_arm, ip(f.arm_qpos_idx)
        desc.n_grip, desc.grip_qpos_idx = len(f.grip_qpos_idx), ip(f.grip_qpos_idx)
        desc.eef_body, desc.eef_off = f.eef_body, (C.c_double * 3)(*f.eef_off)
        desc.rfinger_body, desc.rfinger_off = f.rfinger_body, (C.c_double * 3)(*f.rfinger_off)
        desc.lfinger_body, desc.lfinger_off = f.lfinger_body, (C.c_double * 3)(*f.lfinger_off)
        desc.ee_quat_body, desc.cube_body, desc.target_body = f.ee_quat_body, f.cube_body, f.target_body
        desc.qpos_min, desc.qpos_max, desc.qpos_limited = dp(f.qpos_min), dp(f.qpos_max), ip(f.qpos_limited)
        desc.ac_scale = self.ac_scale
        desc.distance_threshold = float(distance_threshold)
        desc.success_reward = float(success_reward)
        desc.max_episode_steps = self.max_episode_steps
        desc.device = self.device.index if self.device.index is not None else -1
        h = C.c_void_p()
        _lib.check(L.mopa_env_create(C.byref(desc), C.byref(h)))
        self._h = h

        dev, f64 = self.device, torch.float64
        self.qpos = torch.zeros(self.E, self.nq, dtype=f64, device=dev)
        self.prev_state = torch.zeros(self.E, self.n_arm, dtype=f64, device=dev)
        self.has_prev = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        self.ep_len = torch.zeros(self.E, dtype=torch.int32, device=dev)
        self.obs = torch.zeros(self.E, OBS_DIM, dtype=f64, device=dev)
        self.reward = torch.zeros(self.E, dtype=f64, device=dev)
        self.done = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        self.success = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(int(seed))
        self._qpos0 = torch.tensor(self.model.qpos0, dtype=f64, device=dev)
        self._init_arm = torch.tensor(self.spec.init_qpos, dtype=f64, device=dev)
        self._arm_idx = torch.tensor(f.arm_qpos_idx, dtype=torch.long, device=dev)
        self._target_idx = torch.tensor(f.target_qpos_idx, dtype=torch.long, device=dev)
        self._planner = None
        self._scene = None
        self._desired = torch.zeros(self.E, self.n_arm, dtype=f64, device=dev)
        self._move = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        if block_invalid:
            pi = planner_inputs(self.env_name, self.model)
            self._scene = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, self.spec.contact_threshold,
                                     range_=self.spec.range, device=desc.device)
            self._planner = BatchPlanner(self._scene)

    # ------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().mopa_env_destroy(self._h)
            self._h = C.c_void_p()
        if self._scene is not None:
            self._scene.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _launch(self, action, is_planner: bool, move_mask, stream=None):
        _lib.check(_lib.lib().mopa_env_step_batch(
            self._h, self.E, _ptr(self.qpos), _ptr(self.prev_state), _ptr(self.has_prev), _ptr(self.ep_len),
            _ptr(action) if action is not None else None, int(bool(is_planner)),
            _ptr(move_mask) if move_mask is not None else None, _ptr(self.obs), _ptr(self.reward), _ptr(self.done),
            _ptr(self.success), _stream_handle(stream)))

    def exec_trajectories(self, traj, path_len, disc_pow, smdp_rew, smdp_done, intra, rec=None, stream=None):
        """Waypoint execution of the rollout (rl/mopa_rollouts.py:152-199) in one launch: env e steps through
        traj[e, :path_len[e]] ([E, L, nq] f64, [E] int64) until its path ends or a step reports done; smdp_rew [E] f64,
        smdp_done [E] uint8 and intra [E] int64 are updated in place (disc_pow [L] f64 = discount^k).  rec: optional dict
        with 'ob' [E,L,40] f64, 'meta_rew' [E,L] f64, 'done' [E,L] uint8, 'n_exec' [E] int64 filled per executed waypoint."""
        L = int(traj.shape[1])
        r = rec or {}
        p = lambda k: _ptr(r[k]) if k in r else None
        _lib.check(_lib.lib().mopa_env_exec_batch(
            self._h, self.E, _ptr(self.qpos), _ptr(self.prev_state), _ptr(self.has_prev), _ptr(self.ep_len), _ptr(traj),
            _ptr(path_len), L, _ptr(disc_pow), _ptr(self.obs), _ptr(self.reward), _ptr(self.done), _ptr(self.success),
            _ptr(smdp_rew), _ptr(smdp_done), _ptr(intra), p("ob"), p("meta_rew"), p("done"), p("n_exec"), _stream_handle(stream)))

    # ------------------------------------------------------------------
    def reset(self, mask=None):
        """`_reset` of the reference (sawyer_push_obstacle.py:36-52): arm = init_qpos + N(0, 0.02^2), target sliders
        += U(-0.01, 0.01); everything else qpos0.  `mask` (bool/uint8 [E]) resets only those envs."""
        torch = _torch()
        E, dev = self.E, self.device
        q = self._qpos0.expand(E, self.nq).clone()
        q[:, self._arm_idx] = self._init_arm + 0.02 * torch.randn(E, self.n_arm, dtype=torch.float64, device=dev, generator=self._gen)
        q[:, self._target_idx] += (torch.rand(E, 2, dtype=torch.float64, device=dev, generator=self._gen) * 0.02 - 0.01)
        if mask is None:
            self.qpos.copy_(q)
            self.has_prev.zero_()
            self.ep_len.zero_()
        else:
            mk = mask.to(torch.bool)
            self.qpos[mk] = q[mk]
            self.has_prev[mk] = 0
            self.ep_len[mk] = 0
        self._launch(None, False, None)
        return self.obs

    def set_state(self, qpos):
        """Load explicit qpos rows [E, nq] (tests, replaying recorded states) and refresh the obs."""
        self.qpos.copy_(qpos)
        self.has_prev.zero_()
        self.ep_len.zero_()
        self._launch(None, False, None)
        return self.obs

    def step(self, action, is_planner: bool = False, stream=None):
        """`env.step(action, is_planner)` for all E envs.  action: float64 [E, 7] on the GPU.
        Returns (obs [E,40], reward [E], done [E] uint8, info) -- tensors are the env's own buffers (overwritten by the
        next step).  info: success [E] uint8, episode_length [E], and `blocked` [E] with block_invalid."""
        torch = _torch()
        if action.dtype != torch.float64 or not action.is_cuda or not action.is_contiguous() or tuple(action.shape) != (self.E, self.n_arm):
            raise _lib.MopaError(f"action must be a contiguous float64 GPU tensor of shape [{self.E}, {self.n_arm}]")
        info = {}
        move = None
        if self._planner is not None:
            # desired (limit-clamped) state exactly as the step kernel will form it, then the planner's validity rule
            desired = self._desired
            _lib.check(_lib.lib().mopa_env_desired_batch(self._h, self.E, _ptr(self.qpos), _ptr(self.prev_state),
                                                         _ptr(self.has_prev), _ptr(action), int(bool(is_planner)),
                                                         _ptr(desired), _stream_handle(stream)))
            move = self._planner.is_valid(desired, self.qpos, samples_per_env=1, out=self._move, stream=stream)
            info["blocked"] = move ^ 1
        self._launch(action, is_planner, move, stream)
        info["success"] = self.success
        info["episode_length"] = self.ep_len
        return self.obs, self.reward, self.done, info

    def obs_dict(self, obs=None):
        """The reference's OrderedDict view of an obs tensor (`_get_obs`)."""
        o = self.obs if obs is None else obs
        out, k = OrderedDict(), 0
        for name, n in OBS_LAYOUT.items():
            out[name] = o[..., k:k + n]
            k += n
        return out


def make_env(env_name: str, num_envs: int, **kwargs):
    """Batched kinematic env by the reference's gym id (`gym.make(config.env, ...)`, rl/trainer.py:49)."""
    if env_name == BatchKinematicPushEnv.env_name:
        return BatchKinematicPushEnv(num_envs, **kwargs)
    raise _lib.MopaError(f"no batched kinematic env for {env_name!r}")


class SawyerPushObstacleKinematicEnv:
    """Single-env facade with the reference call shapes: `reset() -> ob`, `step(action, is_planner=False) ->
    (ob, reward, done, info)` with `ob` an OrderedDict of numpy arrays (env/base.py:229-246)."""

    def __init__(self, **kwargs):
        self._b = BatchKinematicPushEnv(1, **kwargs)

    @property
    def dof(self):
        return 7

    @property
    def sim_qpos(self):
        return self._b.qpos[0].cpu().numpy()

    def _ob(self):
        return OrderedDict((k, v[0].cpu().numpy()) for k, v in self._b.obs_dict().items())

    def reset(self):
        self._b.reset()
        return self._ob()

    def step(self, action, is_planner: bool = False):
        torch = _torch()
        if isinstance(action, dict):   # env/base.py:233-243: OrderedDict actions are concatenated ("default" key)
            action = np.concatenate([np.asarray(v, dtype=np.float64).ravel() for k, v in action.items() if k != "ac_type"])
        a = torch.tensor(np.asarray(action, dtype=np.float64)[: self.dof].reshape(1, -1), device=self._b.device)
        _, r, d, info = self._b.step(a.contiguous(), is_planner)
        return self._ob(), float(r[0]), bool(d[0]), {"episode_success": int(info["success"][0]),
                                                     "episode_length": int(info["episode_length"][0])}
