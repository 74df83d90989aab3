"""Batched KINEMATIC Sawyer envs (SURVEY.md 8f row 1 + the Lift / Assembly envs of BASELINE configs 4 and 5) on
libmopa_hip.so: SawyerPushObstacle-v0, SawyerLiftObstacle-v0, SawyerAssemblyObstacle-v0.

Restates the reference envs' arithmetic around the physics -- action scaling and `desired_state` (`_step` of
env/sawyer/sawyer_{push,lift,assembly}_obstacle.py; Lift adds the gripper command `gripper qpos + action[-1]`,
sawyer.py:340-342), joint-limit clamp + episode bookkeeping (env/base.py:269-314), reward / success (`compute_reward`),
the observation in dict order (env/sawyer/sawyer.py:317-338 + the env's `_get_obs`) and `_reset` -- and replaces
`_do_simulation` (75 MuJoCo sub-steps of position servos) by its kinematic limit: every actuated joint reaches its
target, clamped to the actuator's ctrlrange; velocities are zero, nothing else moves.  **Not dynamics parity**: no
contact forces, the manipulated object never moves, so push / lift rewards that need the object to move can only be
collected by a policy in the real env (the assembly reward -- peg head to hole -- is purely kinematic).  What it is for:
the "env-steps/sec" half of the metric, and the planner-side rollouts of MoPA-RL (a planned, collision-checked joint
path is executed kinematically by construction).  Checked against the reference's own env classes run over a
sim-shaped adapter (tests/golden/ref_py_env_*.npz, tools/gen_ref_py_golden.py) and bit for bit against the CPU oracle.

`dynamics=True` (stage A) puts the physics back for a robot that touches nothing: `_do_simulation` becomes
the reference's 75 sub-steps of force-limited position servos with gravity compensation, on the arm's own tree (K6,
`csrc/mopa_dyn.inc`, `dynamics.py`): the arm lags `desired_state` as the real one does, the obs carries joint velocities,
`qvel` / `bias_lag` are carried per env.  Without `contacts` nothing but the robot moves.

`contacts=True` (stage C, all three envs; K7 `csrc/mopa_contact.inc`) adds the contacts: the arm against the scene (it stops
at the bin walls and the table), the manipulated object (Push: cube, Lift: can, Assembly: furniture) as a free rigid body,
and the two against each other, two-way coupled behind ONE soft-constraint solve per sub-step -- MuJoCo's constraint model
restated from its published formulation (solref / solimp impedance, pyramidal friction cones, projected Gauss-Seidel with
the XML's iteration cap), PARITY UNPINNED; the collision geometry is sampled feature points in exact signed-distance
functions.  The cube can be pushed, the can pinched and lifted by friction.  `contact_options` go to
`dynamics.contact_facts` (maxcon, maxpair, iterations, tolerance, warmstart ...).
`contacts="penalty"` (stage B of round 3, Push only) keeps the earlier stand-in: the cube under penalty springs, one-way
coupled (spring-damper normal force, capped regularised Coulomb friction) -- NOT a constraint solver, labelled as such.

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

KIND_PUSH, KIND_LIFT, KIND_ASSEMBLY, KIND_PUSHER = 0, 1, 2, 3
OBJECT_BODY = {KIND_PUSH: "cube", KIND_LIFT: "cube", KIND_ASSEMBLY: "furniture"}     # the manipulated object's free body (stage C)
ENV_KIND = {"SawyerPushObstacle-v0": KIND_PUSH, "SawyerLiftObstacle-v0": KIND_LIFT, "SawyerAssemblyObstacle-v0": KIND_ASSEMBLY,
            "PusherObstacle-v0": KIND_PUSHER}

# observation layouts == the reference's OrderedDict order (sawyer.py:317-338, then the env's own `_get_obs`)
_COMMON = [("joint_pos", 7), ("joint_vel", 7), ("gripper_qpos", 2), ("gripper_qvel", 2), ("eef_pos", 3), ("eef_quat", 4)]
OBS_LAYOUTS = {
    KIND_PUSH: OrderedDict(_COMMON + [("target_pos", 3), ("cube_pos", 3), ("cube_quat", 4), ("gripper_to_cube", 3), ("cube_to_target", 2)]),
    KIND_LIFT: OrderedDict(_COMMON + [("cube_pos", 3), ("cube_quat", 4), ("gripper_to_cube", 3)]),
    KIND_ASSEMBLY: OrderedDict(_COMMON + [("hole", 3), ("pegHead", 3), ("pegEnd", 3), ("peg_quat", 4)]),
    # env/pusher/pusher_obstacle.py:183-204: default = [cos theta 4, sin theta 4, box qpos 2, joint vel 4, box vel 2]
    KIND_PUSHER: OrderedDict([("default", 16), ("fingertip", 2), ("goal", 2)]),
}
OBS_LAYOUT = OBS_LAYOUTS[KIND_PUSH]
OBS_DIM = 40


@dataclass
class EnvFacts:
    """Name -> id resolution of what the reference env's `_get_reference` / `compute_reward` / `_get_obs` look up, on a
    CompiledModel.  Frames (world position of a body-fixed point) and quats (world orientation of a body) are listed in
    the slot order the kernel expects for `kind`:
      all       frame 0 = site grip_site,  quat 0 = body right_ee_attchment
      push      frames 1 right_eef, 2 left_eef (sites), 3 cube, 4 target (bodies);  quat 1 = cube
      lift      frames 1 cube, 2 bin1 (bodies);  quat 1 = cube;  touch geoms = [can, left-finger geoms, right-finger geoms]
      assembly  frames 1 hole, 2 hole_bottom, 3 pegHead, 4 pegEnd (sites);  quat 1 = peg
      pusher    frames 0 site fingertip, 1 body fingertip, 2 box, 3 target (bodies);  quats = fingertip, box (unused)"""
    kind: int
    arm_qpos_idx: np.ndarray
    grip_qpos_idx: np.ndarray
    act_qpos_idx: np.ndarray
    act_lo: np.ndarray
    act_hi: np.ndarray
    frame_body: np.ndarray
    frame_off: np.ndarray
    quat_body: np.ndarray
    touch_geom: np.ndarray
    n_touch_left: int
    qpos_min: np.ndarray
    qpos_max: np.ndarray
    qpos_limited: np.ndarray
    reset_jitter_idx: np.ndarray      # qpos addresses that `_reset` moves by U(-0.01, 0.01) (push: the target sliders)

    @property
    def action_dim(self) -> int:
        return len(self.arm_qpos_idx) + (1 if self.kind == KIND_LIFT else 0)

    # named views of the slots (all kinds: eef / ee_quat; push: fingers, cube, target)
    eef_body = property(lambda self: int(self.frame_body[0]))
    eef_off = property(lambda self: self.frame_off[0])
    ee_quat_body = property(lambda self: int(self.quat_body[0]))
    rfinger_body = property(lambda self: int(self.frame_body[1]))
    rfinger_off = property(lambda self: self.frame_off[1])
    lfinger_body = property(lambda self: int(self.frame_body[2]))
    lfinger_off = property(lambda self: self.frame_off[2])
    cube_body = property(lambda self: int(self.frame_body[3] if self.kind == KIND_PUSH else self.frame_body[1]))
    target_body = property(lambda self: int(self.frame_body[4]))
    target_qpos_idx = property(lambda self: self.reset_jitter_idx)

    @property
    def obs_dim(self) -> int:
        return sum(OBS_LAYOUTS[self.kind].values())


def env_facts(env_name: str, model) -> EnvFacts:
    m = model
    spec = ENV_SPECS[env_name]
    kind = ENV_KIND[env_name]

    def site(name):
        i = m.site_name2id(name)
        return int(m.site_body[i]), np.asarray(m.site_pos[i], dtype=np.float64).copy()

    def body(name):
        return m.body_names.index(name), np.zeros(3)

    def cgeom(name):          # index among the collidable geoms
        return int(np.where(m.geom_mjid == m.geom_name2id(name))[0][0])

    if kind == KIND_PUSHER:
        # env/pusher/pusher_obstacle.py: four hinges (joint0 unlimited) under torque motors driven by the env's PID loop
        # (env/base.py:200-209); kinematic limit: the joints reach `desired_state` (no ctrl range: the ctrl is a torque)
        arm = np.array([m.get_joint_qpos_addr(j) for j in spec.robot_joints], dtype=np.int32)
        frames = [site("fingertip"), body("fingertip"), body("box"), body("target")]
        jidx, jlo, jhi, jlim = qpos_joint_arrays(m)
        return EnvFacts(
            kind=kind, arm_qpos_idx=arm, grip_qpos_idx=np.zeros(0, dtype=np.int32), act_qpos_idx=arm.copy(),
            act_lo=np.full(len(arm), -np.inf), act_hi=np.full(len(arm), np.inf),
            frame_body=np.array([f[0] for f in frames], dtype=np.int32), frame_off=np.array([f[1] for f in frames], dtype=np.float64),
            quat_body=np.array([m.body_names.index("fingertip"), m.body_names.index("box")], dtype=np.int32),
            touch_geom=np.zeros(0, dtype=np.int32), n_touch_left=0,
            qpos_min=jlo[jidx].astype(np.float64), qpos_max=jhi[jidx].astype(np.float64), qpos_limited=jlim[jidx].astype(np.int32),
            reset_jitter_idx=np.zeros(0, dtype=np.int64))
    frames = [site("grip_site")]
    quats = [m.body_names.index("right_ee_attchment")]
    touch, n_left, jitter = [], 0, []
    if kind == KIND_PUSH:
        frames += [site("right_eef"), site("left_eef"), body("cube"), body("target")]
        quats.append(m.body_names.index("cube"))
        jitter = [m.get_joint_qpos_addr(j) for j in ("target_x", "target_y")]
    elif kind == KIND_LIFT:
        frames += [body("cube"), body("bin1")]
        quats.append(m.body_names.index("cube"))
        left = ["l_finger_g0", "l_finger_g1", "l_fingertip_g0"]          # sawyer_lift_obstacle.py:42-48
        right = ["r_finger_g0", "r_finger_g1", "r_fingertip_g0"]
        touch = [cgeom("cube")] + [cgeom(g) for g in left] + [cgeom(g) for g in right]
        n_left = len(left)
    else:
        frames += [site("hole"), site("hole_bottom"), site("pegHead"), site("pegEnd")]
        quats.append(m.body_names.index("peg"))
    arm = np.array([m.get_joint_qpos_addr(j) for j in spec.robot_joints], dtype=np.int32)
    # position actuators in ctrl order; the reference writes `desired_state` (+ Lift: the two gripper targets) into ctrl
    act_adr = np.array([m.jnt_qposadr[j] for j in m.act_joint], dtype=np.int32)
    n_act = len(arm) + (2 if kind == KIND_LIFT else 0)
    if len(act_adr) < n_act or list(act_adr[:len(arm)]) != list(arm):
        raise _lib.MopaError(f"{env_name}: the model's actuators do not start with the arm's position servos")
    lim = m.act_ctrllimited[:n_act] == 1
    jidx, jlo, jhi, jlim = qpos_joint_arrays(m)
    return EnvFacts(
        kind=kind, arm_qpos_idx=arm,
        grip_qpos_idx=np.array([m.get_joint_qpos_addr(j) for j in ("rc_close", "lc_close")], dtype=np.int32),
        act_qpos_idx=act_adr[:n_act], act_lo=np.where(lim, m.act_ctrlrange[:n_act, 0], -np.inf),
        act_hi=np.where(lim, m.act_ctrlrange[:n_act, 1], np.inf),
        frame_body=np.array([f[0] for f in frames], dtype=np.int32), frame_off=np.array([f[1] for f in frames], dtype=np.float64),
        quat_body=np.array(quats, dtype=np.int32), touch_geom=np.array(touch, dtype=np.int32), n_touch_left=n_left,
        qpos_min=jlo[jidx].astype(np.float64), qpos_max=jhi[jidx].astype(np.float64), qpos_limited=jlim[jidx].astype(np.int32),
        reset_jitter_idx=np.array(jitter, dtype=np.int64))


def push_env_facts(model) -> EnvFacts:
    return env_facts("SawyerPushObstacle-v0", model)


class BatchKinematicEnv:
    """E envs of one of the three Sawyer obstacle tasks -- or of PusherObstacle-v0 (KIND_PUSHER: four hinges, joint0 unlimited) --
    stepped kinematically on one GPU."""

    def __init__(self, env_name: str, num_envs: int, device=None, seed: int = 0, max_episode_steps: Optional[int] = None,
                 distance_threshold: Optional[float] = None, success_reward: float = 150.0, ac_scale: Optional[float] = None,
                 block_invalid: bool = False, model=None, dynamics: bool = False, frame_dt: float = 0.15, contacts=False,
                 contact_options: dict = None, dyn_lanes: int = 1):
        torch = _torch()
        if env_name not in ENV_KIND:
            raise _lib.MopaError(f"no batched kinematic env for {env_name!r}")
        if distance_threshold is None:        # config/sawyer.py: 0.06, config/pusher.py:16-20: 0.05
            distance_threshold = 0.05 if ENV_KIND[env_name] == KIND_PUSHER else 0.06
        if dynamics and ENV_KIND[env_name] == KIND_PUSHER:
            raise _lib.MopaError("PusherObstacle-v0: kinematic env only (the dynamics / contact kernels restate the Sawyer envs' servo model)")
        if not torch.cuda.is_available():
            raise _lib.MopaError("BatchKinematicEnv needs a HIP device (there is no CPU fallback)")
        self.env_name = env_name
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.spec = ENV_SPECS[env_name]
        self.model = model if model is not None else load_scene(self.spec.scene)
        self.facts = f = env_facts(env_name, self.model)
        self.kind = f.kind
        self.E = int(num_envs)
        self.nq = self.model.nq
        self.n_arm = len(f.arm_qpos_idx)
        self.action_dim, self.obs_dim = f.action_dim, f.obs_dim
        self.obs_layout = OBS_LAYOUTS[f.kind]
        if max_episode_steps is None:         # config/sawyer.py: 250, config/pusher.py: 150 (the reference's per-env defaults)
            max_episode_steps = 150 if env_name == "PusherObstacle-v0" else 250
        self.max_episode_steps = int(max_episode_steps)
        self.ac_scale = float(self.spec.ac_scale if ac_scale is None else ac_scale)
        L = _lib.lib()
        keep = []
        desc = _lib.MopaEnvDesc()
        desc.model = _lib.model_struct(self.model, keep)

        def ip(a):
            a, p = _lib._i(a); keep.append(a); return p

        def dp(a):
            a, p = _lib._d(a); keep.append(a); return p

        desc.kind = f.kind
        desc.n_arm, desc.arm_qpos_idx = self.n_arm, ip(f.arm_qpos_idx)
        desc.n_grip, desc.grip_qpos_idx = len(f.grip_qpos_idx), ip(f.grip_qpos_idx)
        desc.n_act, desc.act_qpos_idx, desc.act_ctrl_lo, desc.act_ctrl_hi = len(f.act_qpos_idx), ip(f.act_qpos_idx), dp(f.act_lo), dp(f.act_hi)
        desc.n_frames, desc.frame_body, desc.frame_off = len(f.frame_body), ip(f.frame_body), dp(f.frame_off)
        desc.n_quats, desc.quat_body = len(f.quat_body), ip(f.quat_body)
        desc.n_touch, desc.touch_geom, desc.n_touch_left = len(f.touch_geom), ip(f.touch_geom), int(f.n_touch_left)
        desc.qpos_min, desc.qpos_max, desc.qpos_limited = dp(f.qpos_min), dp(f.qpos_max), ip(f.qpos_limited)
        desc.ac_scale = self.ac_scale
        desc.distance_threshold = float(distance_threshold)
        desc.success_reward = float(success_reward)
        desc.max_episode_steps = self.max_episode_steps
        desc.device = self.device.index if self.device.index is not None else -1
        h = C.c_void_p()
        _lib.check(L.mopa_env_create(C.byref(desc), C.byref(h)))
        self._h = h
        assert L.mopa_env_obs_dim(h) == self.obs_dim and L.mopa_env_action_dim(h) == self.action_dim

        dev, f64 = self.device, torch.float64
        self.qpos = torch.zeros(self.E, self.nq, dtype=f64, device=dev)
        self.prev_state = torch.zeros(self.E, self.n_arm, dtype=f64, device=dev)
        self.has_prev = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        self.ep_len = torch.zeros(self.E, dtype=torch.int32, device=dev)
        self.obs = torch.zeros(self.E, self.obs_dim, dtype=f64, device=dev)
        self.reward = torch.zeros(self.E, dtype=f64, device=dev)
        self.done = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        self.success = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(int(seed))
        self._qpos0 = torch.tensor(self.model.qpos0, dtype=f64, device=dev)
        init_arm = self.spec.init_qpos if len(self.spec.init_qpos) else np.asarray(self.model.qpos0)[np.asarray(f.arm_qpos_idx, dtype=np.int64)]
        self._init_arm = torch.tensor(np.asarray(init_arm, dtype=np.float64), dtype=f64, device=dev)
        row = np.array(self.model.qpos0, dtype=np.float64)
        row[np.asarray(self.facts.arm_qpos_idx, dtype=np.int64)] = np.asarray(init_arm, dtype=np.float64)
        self.init_qpos_row = row            # qpos0 with the arm at the env's init_qpos: the reset pose without its noise
        self._arm_idx = torch.tensor(f.arm_qpos_idx, dtype=torch.long, device=dev)
        self._jitter_idx = torch.tensor(f.reset_jitter_idx, dtype=torch.long, device=dev)
        self._planner = None
        self._scene = None
        self.dynamics = bool(dynamics)
        self.dyn = None
        self.obj = None
        if self.dynamics:
            from .dynamics import dyn_facts
            self.dyn = df = dyn_facts(self.model, f, frame_dt=frame_dt)
            dd = _lib.MopaDynDesc()
            dd.nd = df.nd
            dd.parent, dd.jtype, dd.qadr = ip(df.parent), ip(df.jtype), ip(df.qadr)
            dd.rel_pos, dd.rel_quat, dd.axis, dd.jpos, dd.qref = dp(df.rel_pos), dp(df.rel_quat), dp(df.axis), dp(df.jpos), dp(df.qref)
            dd.mass, dd.ipos, dd.inertia = dp(df.mass), dp(df.ipos), dp(df.inertia)
            dd.damping, dd.armature = dp(df.damping), dp(df.armature)
            dd.limited, dd.lo, dd.hi = ip(df.limited), dp(df.lo), dp(df.hi)
            dd.actuated, dd.kp, dd.force_lo, dd.force_hi, dd.gravcomp = ip(df.actuated), dp(df.kp), dp(df.force_lo), dp(df.force_hi), ip(df.gravcomp)
            dd.gravity = (C.c_double * 3)(*[float(x) for x in df.gravity])
            dd.timestep, dd.nsub = float(df.timestep), int(df.nsub)
            self.obj = None
            self.ct = None
            if contacts == "penalty":
                # stage B (round 3): the Push cube under penalty springs, one-way coupled -- kept as a labelled stand-in
                if f.kind != KIND_PUSH:
                    raise _lib.MopaError("contacts='penalty': the penalty-contact object model is built for the Push cube only")
                from .dynamics import obj_facts
                self.obj = of = obj_facts(self.model, df)
                od = _lib.MopaObjDesc()
                od.qadr, od.mass, od.damping, od.rbound = int(of.qadr), float(of.mass), float(of.damping), float(of.rbound)
                od.inertia = (C.c_double * 3)(*[float(x) for x in of.inertia])
                od.half = (C.c_double * 3)(*[float(x) for x in of.half])
                od.nfeat, od.feat = len(of.feat), dp(of.feat)
                od.ncol, od.co_body, od.co_type = len(of.co_body), ip(of.co_body), ip(of.co_type)
                od.co_size, od.co_pos, od.co_mat, od.co_mu, od.co_rbound = dp(of.co_size), dp(of.co_pos), dp(of.co_mat), dp(of.co_mu), dp(of.co_rbound)
                od.kn, od.dn, od.eps_v, od.ct_max = float(of.kn), float(of.dn), float(of.eps_v), float(of.ct_max)
                od.inv_mass = float(of.inv_mass)
                od.inv_inertia = (C.c_double * 3)(*[float(x) for x in of.inv_inertia])
                od.precull_every, od.precull_margin = int(of.precull_every), float(of.precull_margin)
                keep.append(od)
                dd.obj = C.pointer(od)
            elif contacts:
                # stage C: contacts of arm and object behind the constraint solver, all three envs
                from .dynamics import contact_facts
                self.ct = contact_facts(self.model, df, OBJECT_BODY[f.kind], qpos_ref=self.init_qpos_row, **(contact_options or {}))
            _lib.check(L.mopa_env_attach_dynamics(self._h, C.byref(dd)))
            assert L.mopa_env_dyn_dofs(self._h) == df.nd
            if dyn_lanes not in (1, 16):
                raise _lib.MopaError("dyn_lanes: 16 (lanes per env: the contact kernel's mapping) or 1 (K6: a lane per env)")
            self.dyn_lanes = 16 if self.ct is not None else (1 if self.obj is not None else int(dyn_lanes))
            if self.ct is None and self.obj is None and dyn_lanes == 16:
                # contact-free servo dynamics in the 16-lanes-per-env mapping of the contact kernel: a contact stage with no object, no
                # pairs, limits as inelastic stops.  Same results; measured (tools/dyn_lanes_ab.py) it is NOT faster than K6 at 4096 envs
                # (1.58 vs 1.54 ms per env.step: either way a sub-step is one env's serial chain, ~20 us) and a quarter of K6's rate
                # once K6 has 16 384 envs to fill the chip with (2.7 vs 10.6 M env-steps/s) -- hence not the default
                cd = _lib.MopaCtDesc()
                cd.ns = cd.nf = cd.np = 0
                cd.obj_qadr = -1
                cd.maxcon, cd.maxpair, cd.iterations = 0, 1, 1
                cd.tolerance, cd.inv_scale = 0.0, 1.0
                cd.precull_every, cd.precull_margin, cd.warmstart = 15, 0.0, 0
                cd.near_every, cd.near_margin = 3, 0.0
                cd.noslip_iterations, cd.noslip_tolerance = 0, 0.0
                cd.solver, cd.limit_rows = 0, 0
                _lib.check(L.mopa_env_attach_contacts(self._h, C.byref(cd)))
            if self.ct is not None:
                ct = self.ct
                cd = _lib.MopaCtDesc()
                cd.ns, cd.sh_body, cd.sh_type = len(ct.sh_body), ip(ct.sh_body), ip(ct.sh_type)
                cd.sh_size, cd.sh_pos, cd.sh_mat, cd.sh_rbound, cd.sh_feat0 = dp(ct.sh_size), dp(ct.sh_pos), dp(ct.sh_mat), dp(ct.sh_rbound), ip(ct.sh_feat0)
                cd.nf, cd.ft_pos, cd.ft_rad = len(ct.ft_rad), dp(ct.ft_pos), dp(ct.ft_rad)
                cd.np, cd.pr_f, cd.pr_s, cd.pr_par = len(ct.pr_f), ip(ct.pr_f), ip(ct.pr_s), dp(ct.pr_par)
                cd.obj_qadr, cd.obj_mass, cd.obj_damping = int(ct.obj_qadr), float(ct.obj_mass), float(ct.obj_damping)
                cd.obj_inertia = (C.c_double * 3)(*[float(x) for x in ct.obj_inertia])
                cd.obj_ipos = (C.c_double * 3)(*[float(x) for x in ct.obj_ipos])
                cd.obj_iquat = (C.c_double * 4)(*[float(x) for x in ct.obj_iquat])
                cd.obj_inv_mass, cd.obj_inv_mass_d = float(ct.obj_inv_mass), float(ct.obj_inv_mass_d)
                cd.obj_inv_inertia = (C.c_double * 3)(*[float(x) for x in ct.obj_inv_inertia])
                cd.obj_inv_inertia_d = (C.c_double * 3)(*[float(x) for x in ct.obj_inv_inertia_d])
                cd.maxcon, cd.maxpair, cd.iterations = int(ct.maxcon), int(ct.maxpair), int(ct.iterations)
                cd.tolerance, cd.inv_scale = float(ct.tolerance), float(ct.inv_scale)
                cd.precull_every, cd.precull_margin, cd.warmstart = int(ct.precull_every), float(ct.precull_margin), int(ct.warmstart)
                cd.near_every, cd.near_margin = int(ct.near_every), float(ct.near_margin)
                cd.noslip_iterations, cd.noslip_tolerance = int(ct.noslip_iterations), float(ct.noslip_tolerance)
                cd.solver = int(ct.solver)
                cd.limit_rows = int(ct.limit_rows)
                cd.lim_par = (C.c_double * 8)(*[float(x) for x in ct.lim_par])
                cd.arena = int(ct.arena)
                assert np.asarray(ct.pr_par).shape[1] == 12
                _lib.check(L.mopa_env_attach_contacts(self._h, C.byref(cd)))
                # the arena the library sized (0 asked for its default): a checker of this env has to drop the same contacts
                import dataclasses
                self.ct = dataclasses.replace(ct, arena=int(L.mopa_env_contact_arena(self._h)) if ct.solver == 2 else 0)
            self.nv = int(L.mopa_env_dyn_qvel_width(self._h))
            assert self.nv == df.nd + (6 if contacts else 0)
            # velocities of the dynamic dofs (arm, gripper) [+ the object's linear / angular velocity, world frame]
            self.qvel = torch.zeros(self.E, self.nv, dtype=f64, device=dev)
            self.bias_lag = torch.zeros(self.E, df.nd, dtype=f64, device=dev)    # qfrc_bias of the last mj_forward
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
        if self.dynamics:
            _lib.check(_lib.lib().mopa_env_step_dyn_batch(
                self._h, self.E, _ptr(self.qpos), _ptr(self.qvel), _ptr(self.bias_lag), _ptr(self.prev_state), _ptr(self.has_prev),
                _ptr(self.ep_len), _ptr(action) if action is not None else None, int(bool(is_planner)),
                _ptr(move_mask) if move_mask is not None else None, _ptr(self.obs), _ptr(self.reward), _ptr(self.done),
                _ptr(self.success), _stream_handle(stream)))
            return
        _lib.check(_lib.lib().mopa_env_step_batch(
            self._h, self.E, _ptr(self.qpos), _ptr(self.prev_state), _ptr(self.has_prev), _ptr(self.ep_len),
            _ptr(action) if action is not None else None, int(bool(is_planner)),
            _ptr(move_mask) if move_mask is not None else None, _ptr(self.obs), _ptr(self.reward), _ptr(self.done),
            _ptr(self.success), _stream_handle(stream)))

    def exec_trajectories(self, traj, path_len, disc_pow, smdp_rew, smdp_done, intra, rec=None, last_extra=None, stream=None,
                          pos=None, chunk=0):
        """Waypoint execution of the rollout (rl/mopa_rollouts.py:152-199) in one launch: env e steps through
        traj[e, :path_len[e]] ([E, L, nq] f64, [E] int64) until its path ends or a step reports done; smdp_rew [E] f64,
        smdp_done [E] uint8 and intra [E] int64 are updated in place (disc_pow [L] f64 = discount^k).  last_extra [E] f64
        (envs with more action entries than arm joints, i.e. Lift): the policy's gripper action, applied at the LAST
        waypoint of a path (:163-167); the other waypoints carry `form_action`'s gripper difference.  rec: optional dict
        with 'ob' [E,L,obs_dim] f64, 'meta_rew' [E,L] f64, 'done' [E,L] uint8, 'n_exec' [E] int64 filled per executed waypoint.
        Dynamics envs only: pos [E] int64 = every env's next waypoint index (advanced in place; set to its path_len when a
        step reports done), chunk > 0 = at most that many step launches -- the resumable form a rollout uses to spread long
        walks over several calls."""
        if self.dynamics:
            return self._exec_trajectories_dyn(traj, path_len, disc_pow, smdp_rew, smdp_done, intra, rec, last_extra, stream, pos, chunk)
        if pos is not None or chunk:
            raise _lib.MopaError("pos / chunk: the kinematic env walks a path in one launch")
        L = int(traj.shape[1])
        r = rec or {}
        p = lambda k: _ptr(r[k]) if k in r else None
        if self.action_dim > self.n_arm and last_extra is None:
            raise _lib.MopaError("this env needs last_extra (the gripper action of the last waypoint)")
        _lib.check(_lib.lib().mopa_env_exec_batch(
            self._h, self.E, _ptr(self.qpos), _ptr(self.prev_state), _ptr(self.has_prev), _ptr(self.ep_len), _ptr(traj),
            _ptr(path_len), L, _ptr(disc_pow), _ptr(last_extra) if last_extra is not None else None, _ptr(self.obs), _ptr(self.reward),
            _ptr(self.done), _ptr(self.success), _ptr(smdp_rew), _ptr(smdp_done), _ptr(intra), p("ob"), p("meta_rew"), p("done"),
            p("n_exec"), _stream_handle(stream)))

    def _exec_trajectories_dyn(self, traj, path_len, disc_pow, smdp_rew, smdp_done, intra, rec, last_extra, stream, pos=None, chunk=0):
        """`exec_trajectories` with the servo dynamics as the physics: a waypoint is a full env.step (75 dependent sub-steps),
        so the walk is one step launch per round over the envs still on their paths (the others sit the launch out through
        the step's move mask), every env at its OWN waypoint index `pos`, with the SMDP return / done / intra_steps folded
        between launches by the same arithmetic as the kinematic kernel (`rew = rew + gamma^k r_k`; stop at the first
        `done`).  One host read-back (the longest remaining walk)."""
        torch = _torch()
        if stream is not None:
            raise _lib.MopaError("the dynamics form of exec_trajectories runs on the current stream")
        E, L = self.E, int(traj.shape[1])
        if self.action_dim > self.n_arm and last_extra is None:
            raise _lib.MopaError("this env needs last_extra (the gripper action of the last waypoint)")
        plen = torch.clamp(path_len, max=L)
        if pos is None:
            pos = torch.zeros(E, dtype=torch.int64, device=self.device)
        n_walk = int((plen - pos).max().item()) if E else 0
        if chunk > 0:
            n_walk = min(n_walk, int(chunk))
        for _ in range(max(n_walk, 0)):
            self.walk_round(traj, plen, pos, disc_pow, smdp_rew, smdp_done, intra, rec, last_extra)

    def walk_round(self, traj, plen, pos, disc_pow, smdp_rew, smdp_done, intra, rec=None, last_extra=None, other_action=None, other_flags=None):
        """One round of the dynamics walk: ONE step launch in which every env with pos < plen executes waypoint traj[e, pos[e]]
        (`is_planner` step towards it) and folds the step into its SMDP return / done / intra_steps / records; pos advances
        (to plen when the step reports done).  The other envs sit the launch out -- or, with other_action [E, action_dim] /
        other_flags [E] uint8 (move-mask values), take that step in the same launch (a rollout's direct and failed-plan steps;
        their arm actions already multiplied by ac_scale, since the launch runs with planner-step semantics; has_prev of
        those envs must be 0)."""
        torch = _torch()
        E, L = self.E, int(traj.shape[1])
        rows = getattr(self, "_rows", None)
        if rows is None:
            rows = self._rows = torch.arange(E, device=self.device)
        g0 = int(self.facts.grip_qpos_idx[0]) if self.action_dim > self.n_arm else None
        act = pos < plen
        k = torch.clamp(pos, max=L - 1)
        wp = traj[rows, k]
        a = wp[:, self._arm_idx] - self.qpos[:, self._arm_idx]          # env.form_action(waypoint): waypoint - current arm state
        if g0 is not None:
            extra = torch.where(plen - 1 == k, last_extra, wp[:, g0] - self.qpos[:, g0])
            a = torch.cat([a, extra[:, None]], dim=1)
        if other_action is not None:
            a = torch.where(act[:, None], a, other_action)
            flags = torch.where(act, torch.ones_like(other_flags), other_flags).contiguous()
        else:
            flags = torch.where(act, 1, 2).to(torch.uint8).contiguous()
        self._launch(a.contiguous(), True, flags)
        smdp_rew.copy_(torch.where(act, smdp_rew + disc_pow[k] * self.reward, smdp_rew))
        smdp_done.copy_(torch.where(act, self.done, smdp_done))
        intra.copy_(torch.where(act, k, intra))
        if rec:
            rec["ob"][rows, k] = torch.where(act[:, None], self.obs, rec["ob"][rows, k])
            rec["meta_rew"][rows, k] = torch.where(act, smdp_rew, rec["meta_rew"][rows, k])
            rec["done"][rows, k] = torch.where(act, self.done, rec["done"][rows, k])
            rec["n_exec"].copy_(torch.where(act, k + 1, rec["n_exec"]))
        # `if done or ep_len >= max_step: break`: the walk of such an env is over
        pos.copy_(torch.where(act, torch.where(self.done.bool(), plen, pos + 1), pos))
        return act

    # ------------------------------------------------------------------
    def reset(self, mask=None):
        """`_reset` of the reference envs: arm = init_qpos + N(0, 0.02^2) (+ push: target sliders += U(-0.01, 0.01),
        sawyer_push_obstacle.py:36-52); everything else qpos0.  `mask` (bool/uint8 [E]) resets only those envs."""
        torch = _torch()
        E, dev = self.E, self.device
        if self.kind == KIND_PUSHER:
            q = self._reset_pusher(mask)
        else:
            q = self._qpos0.expand(E, self.nq).clone()
            q[:, self._arm_idx] = self._init_arm + 0.02 * torch.randn(E, self.n_arm, dtype=torch.float64, device=dev, generator=self._gen)
        if len(self._jitter_idx):
            q[:, self._jitter_idx] += (torch.rand(E, len(self._jitter_idx), dtype=torch.float64, device=dev, generator=self._gen) * 0.02 - 0.01)
        if mask is None:
            self.qpos.copy_(q)
            self.has_prev.zero_()
            self.ep_len.zero_()
        else:
            mk = mask.to(torch.bool)          # (torch.where, not mask indexing: no host read-back of the count)
            self.qpos.copy_(torch.where(mk[:, None], q, self.qpos))
            self.has_prev.copy_(torch.where(mk, torch.zeros_like(self.has_prev), self.has_prev))
            self.ep_len.copy_(torch.where(mk, torch.zeros_like(self.ep_len), self.ep_len))
        self._rest(mask)
        self._launch(None, False, None)
        return self.obs

    def _reset_pusher(self, mask=None, draws: int = 64, pool_factor: int = 16):
        """`PusherObstacleEnv._reset` (env/pusher/pusher_obstacle.py:40-68) for the envs of `mask` (None: all): goal and box ~
        U([-0.35, 0.13], [-0.24, 0.2]) written into their sliders, every qpos entry += U(-0.02, 0.02); a draw is kept when nothing
        touches (`ncon == 0`: K1 with no ignored pair at threshold 0), box and target are more than 0.1 apart and goal_x <= box_x.
        The reference loops until a draw passes (about 1.5 % do: both points come from one 0.11 x 0.07 patch).  Here accepted draws
        are produced in bulk (E x `draws` candidates per validity launch, the accepted ones kept in draw order) into a pool of
        `pool_factor` x E states that resets consume front to back -- every reset state is an independent accepted draw, used once.
        The pool's cursor lives on the device; the host only keeps an upper bound of it (E per masked call) and reads the true value
        when that bound says the pool might run out: one synchronisation every ~pool_factor calls."""
        torch = _torch()
        E, K, dev, f64 = self.E, int(draws), self.device, torch.float64
        if getattr(self, "_reset_scene", None) is None:
            pi = planner_inputs(self.env_name, self.model)
            self._reset_scene = _lib.Scene(pi.model, pi.passive_joint_idx, [], 0.0, device=dev.index if dev.index is not None else -1,
                                           prune_pairs=False)
            self._reset_bp = BatchPlanner(self._reset_scene)
            m = self.model
            self._box_off = torch.tensor(np.asarray(m.body_pos[m.body_names.index("box")]) - np.asarray(m.body_pos[m.body_names.index("target")]),
                                         dtype=f64, device=dev)
            self._pool = torch.zeros(0, self.nq, dtype=f64, device=dev)
            self._pool_cursor = torch.zeros((), dtype=torch.int64, device=dev)
            self._pool_upper = 0
        P = int(pool_factor) * E
        if self._pool_upper + E > self._pool.shape[0]:
            used = min(int(self._pool_cursor.item()), self._pool.shape[0])
            if used + E <= self._pool.shape[0]:
                self._pool_upper = used          # (masked resets consumed far less than the bound assumed: no refill yet)
        if self._pool_upper + E > self._pool.shape[0]:
            used = min(int(self._pool_cursor.item()), self._pool.shape[0])
            parts, have = [self._pool[used:]], self._pool.shape[0] - used
            lo = torch.tensor([-0.35, 0.13], dtype=f64, device=dev)
            hi = torch.tensor([-0.24, 0.2], dtype=f64, device=dev)
            while have < P:
                N = E * K
                c = self._qpos0 + (torch.rand(N, self.nq, dtype=f64, device=dev, generator=self._gen) * 0.04 - 0.02)
                goal = lo + (hi - lo) * torch.rand(N, 2, dtype=f64, device=dev, generator=self._gen)
                box = lo + (hi - lo) * torch.rand(N, 2, dtype=f64, device=dev, generator=self._gen)
                c[:, -4:-2], c[:, -2:] = goal, box
                free = self._reset_bp.is_valid(c[:, self._arm_idx].contiguous(), c, samples_per_env=1).bool()
                d = torch.cat([box - goal, torch.zeros(N, 1, dtype=f64, device=dev)], dim=1) + self._box_off
                ok = free & (d.norm(dim=1) > 0.1) & (goal[:, 0] <= box[:, 0])
                acc = c[ok]
                parts.append(acc)
                have += acc.shape[0]
            self._pool = torch.cat(parts)[:max(P, E)].contiguous()
            self._pool_cursor.zero_()
            self._pool_upper = 0
        if mask is None:
            idx = self._pool_cursor + torch.arange(E, device=dev)
            self._pool_cursor += E
        else:
            mk = mask.to(torch.bool)
            rank = torch.cumsum(mk.to(torch.int64), 0) - 1
            idx = self._pool_cursor + torch.where(mk, rank, torch.zeros_like(rank))
            self._pool_cursor += mk.sum()
        self._pool_upper += E
        return self._pool[idx.clamp(max=self._pool.shape[0] - 1)]

    def set_state(self, qpos):
        """Load explicit qpos rows [E, nq] (tests, replaying recorded states) and refresh the obs."""
        self.qpos.copy_(qpos)
        self.has_prev.zero_()
        self.ep_len.zero_()
        self._rest(None)
        self._launch(None, False, None)
        return self.obs

    def _rest(self, mask):
        """dynamics: the (re)set envs are at rest; bias_lag <- qfrc_bias of the `sim.forward()` that follows a reset."""
        if not self.dynamics:
            return
        torch = _torch()
        skip = None
        if mask is None:
            self.qvel.zero_()
        else:
            mk = mask.to(torch.bool)
            self.qvel.copy_(torch.where(mk[:, None], torch.zeros_like(self.qvel), self.qvel))
            skip = ((~mk).to(torch.uint8) * 2).contiguous()       # bit 1: env sits this call out
        _lib.check(_lib.lib().mopa_env_dyn_forward_batch(self._h, self.E, _ptr(self.qpos), _ptr(self.qvel), _ptr(self.bias_lag), None,
                                                         _ptr(skip) if skip is not None else None, None))

    def dyn_substeps(self, ctrl, n: int = 1, stream=None):
        """n raw sub-steps towards servo targets ctrl [E, nd] (tests / parity per sub-step)."""
        _lib.check(_lib.lib().mopa_env_dyn_substeps_batch(self._h, self.E, _ptr(self.qpos), _ptr(self.qvel), _ptr(self.bias_lag),
                                                          _ptr(ctrl), int(n), _stream_handle(stream)))

    def dyn_forward(self, want_M: bool = False):
        """(qfrc_bias [E, nd], packed lower triangle of M [E, nd (nd + 1) / 2] or None) at the current (qpos, qvel)."""
        torch = _torch()
        nd = self.dyn.nd
        bias = torch.zeros(self.E, nd, dtype=torch.float64, device=self.device)
        M = torch.zeros(self.E, nd * (nd + 1) // 2, dtype=torch.float64, device=self.device) if want_M else None
        _lib.check(_lib.lib().mopa_env_dyn_forward_batch(self._h, self.E, _ptr(self.qpos), _ptr(self.qvel), _ptr(bias),
                                                         _ptr(M) if want_M else None, None, None))
        return bias, M

    def step(self, action, is_planner: bool = False, stream=None):
        """`env.step(action, is_planner)` for all E envs.  action: float64 [E, action_dim] on the GPU (7 arm entries; Lift:
        + the gripper entry).  Returns (obs [E, obs_dim], reward [E], done [E] uint8, info) -- tensors are the env's own
        buffers (overwritten by the next step).  info: success [E] uint8, episode_length [E], and `blocked` [E] with
        block_invalid."""
        torch = _torch()
        if action.dtype != torch.float64 or not action.is_cuda or not action.is_contiguous() or tuple(action.shape) != (self.E, self.action_dim):
            raise _lib.MopaError(f"action must be a contiguous float64 GPU tensor of shape [{self.E}, {self.action_dim}]")
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
        for name, n in self.obs_layout.items():
            out[name] = o[..., k:k + n]
            k += n
        return out


class BatchKinematicPushEnv(BatchKinematicEnv):
    """E SawyerPushObstacle envs (the first of the three; kept under its own name)."""
    env_name = "SawyerPushObstacle-v0"

    def __init__(self, num_envs: int, **kwargs):
        super().__init__(self.env_name, num_envs, **kwargs)


def make_env(env_name: str, num_envs: int, **kwargs) -> BatchKinematicEnv:
    """Batched kinematic env by the reference's gym id (`gym.make(config.env, ...)`, rl/trainer.py:49)."""
    return BatchKinematicEnv(env_name, num_envs, **kwargs)


class SawyerKinematicEnv:
    """Single-env facade with the reference call shapes: `reset() -> ob`, `step(action, is_planner=False) ->
    (ob, reward, done, info)` with `ob` an OrderedDict of numpy arrays (env/base.py:229-246)."""

    def __init__(self, env_name: str = "SawyerPushObstacle-v0", **kwargs):
        self._b = BatchKinematicEnv(env_name, 1, **kwargs)

    @property
    def dof(self):
        return self._b.action_dim

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


class SawyerPushObstacleKinematicEnv(SawyerKinematicEnv):
    def __init__(self, **kwargs):
        super().__init__("SawyerPushObstacle-v0", **kwargs)
