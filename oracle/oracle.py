"""ctypes binding of the CPU oracle (oracle/mopa_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under mopa_rl_amd/.
PARITY UNPINNED vs MuJoCo/OMPL (see mopa_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmopa_oracle.so")
ORC_FAR = 1.0e10


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("mopa_oracle.c", "mopa_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmopa_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [
            C.c_int, C.c_int, _ip, _dp, _dp, _ip, _ip,
            C.c_int, _ip, _ip, _dp, _dp, _dp, _ip, _dp,
            C.c_int, _ip, _ip, _ip, _dp, _dp, _dp,
            C.c_int, _ip, C.c_int, _ip, C.c_int, _ip, C.c_double]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_num_active.argtypes = [C.c_void_p]
        L.orc_num_active.restype = C.c_int
        L.orc_active_idx.argtypes = [C.c_void_p, _ip]
        L.orc_sincos.argtypes = [C.c_double, _dp, _dp]
        L.orc_set_convex_axes_pretest.argtypes = [C.c_int]
        L.orc_set_convex_axes_pretest.restype = None
        L.orc_fk.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_fk_bodies.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_geom_dist.restype = C.c_double
        L.orc_geom_dist.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.orc_pair_dist.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_is_valid.restype = C.c_int
        L.orc_is_valid.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_is_valid_batch.argtypes = [C.c_void_p, _dp, _dp, C.c_int64, C.c_int64, _u8p, _dp, C.c_int]
        L.orc_check_motion.restype = C.c_int
        L.orc_check_motion.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, C.POINTER(C.c_int64)]
        L.orc_check_motion_batch.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int64, C.c_int64, C.c_double, _u8p, C.c_int]
        L.orc_rng_u64.restype = C.c_uint64
        L.orc_rng_u64.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_rng_uniform.restype = C.c_double
        L.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_plan_batch.restype = None
        L.orc_plan_batch.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _dp, C.c_int,
                                     _ip, _ip, C.c_void_p, C.c_int]
        L.orc_plan.restype = C.c_int
        L.orc_plan.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                               _dp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        L.orc_scene_set_meshes.restype = None
        L.orc_scene_set_meshes.argtypes = [C.c_void_p, C.c_int, _ip, _ip, C.c_int, _dp, _ip]
        L.orc_geom_dist_mesh.restype = C.c_double
        L.orc_geom_dist_mesh.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_exp.restype = C.c_double
        L.orc_exp.argtypes = [C.c_double]
        L.orc_tanh_pos.restype = C.c_double
        L.orc_tanh_pos.argtypes = [C.c_double]
        L.orc_env_step.restype = None
        L.orc_env_step.argtypes = [C.c_void_p, C.POINTER(OrcEnvDesc), _dp, _dp, _u8p, _ip, _dp, C.c_int, C.c_int, _dp, _dp,
                                   _u8p, _u8p]
        L.orc_env_obs_dim.restype = C.c_int
        L.orc_env_obs_dim.argtypes = [C.POINTER(OrcEnvDesc)]
        L.orc_env_action_dim.restype = C.c_int
        L.orc_env_action_dim.argtypes = [C.POINTER(OrcEnvDesc)]
        L.orc_ik_solve.restype = None
        L.orc_ik_solve.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_double,
                                   C.c_double, C.c_double, C.c_double, _dp, _ip, _u8p]
        L.orc_atan2.restype = C.c_double
        L.orc_atan2.argtypes = [C.c_double, C.c_double]
        L.orc_env_step_batch.restype = None
        L.orc_env_step_batch.argtypes = [C.c_void_p, C.POINTER(OrcEnvDesc), C.c_int64, _dp, _dp, _u8p, _ip, _dp, C.c_int, _u8p,
                                         _dp, _dp, _u8p, _u8p, C.c_int]
        L.orc_dyn_forward.restype = None
        L.orc_dyn_forward.argtypes = [C.POINTER(OrcDynDesc), _dp, _dp, _dp, _dp]
        L.orc_dyn_step.restype = None
        L.orc_dyn_step.argtypes = [C.POINTER(OrcDynDesc), _dp, _dp, _dp, _dp, C.c_int]
        L.orc_dyn_step_obj.restype = None
        L.orc_dyn_step_obj.argtypes = [C.POINTER(OrcDynDesc), _dp, _dp, _dp, _dp, C.c_int, _dp]
        L.orc_ct_step.restype = None
        L.orc_ct_step.argtypes = [C.POINTER(OrcDynDesc), _dp, _dp, _dp, _dp, C.c_int, C.c_void_p]
        L.orc_ct_desc_size.restype = C.c_int
        L.orc_ct_desc_size.argtypes = []
        L.orc_ct_contacts.restype = C.c_int
        L.orc_ct_contacts.argtypes = [C.POINTER(OrcDynDesc), _dp, _dp]
        L.orc_env_step_dyn_batch.restype = None
        L.orc_env_step_dyn_batch.argtypes = [C.c_void_p, C.POINTER(OrcEnvDesc), C.POINTER(OrcDynDesc), C.c_int64, _dp, _dp, _dp,
                                             _dp, _u8p, _ip, _dp, C.c_int, _u8p, _dp, _dp, _u8p, _u8p, C.c_int]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class OrcEnvDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_arm", C.c_int32), ("arm_qpos_idx", _ip), ("n_grip", C.c_int32), ("grip_qpos_idx", _ip),
        ("n_act", C.c_int32), ("act_qpos_idx", _ip), ("act_lo", _dp), ("act_hi", _dp),
        ("n_frames", C.c_int32), ("frame_body", _ip), ("frame_off", _dp), ("n_quats", C.c_int32), ("quat_body", _ip),
        ("n_touch", C.c_int32), ("touch_geom", _ip), ("n_touch_left", C.c_int32),
        ("qpos_min", _dp), ("qpos_max", _dp), ("qpos_limited", _ip),
        ("ac_scale", C.c_double), ("distance_threshold", C.c_double), ("success_reward", C.c_double),
        ("max_episode_steps", C.c_int32),
    ]


class OrcObjDesc(C.Structure):
    _fields_ = [
        ("qadr", C.c_int32), ("mass", C.c_double), ("inertia", C.c_double * 3), ("damping", C.c_double),
        ("half", C.c_double * 3), ("rbound", C.c_double), ("nfeat", C.c_int32), ("feat", _dp),
        ("ncol", C.c_int32), ("co_body", _ip), ("co_type", _ip), ("co_size", _dp), ("co_pos", _dp), ("co_mat", _dp),
        ("co_mu", _dp), ("co_rbound", _dp), ("inv_mass", C.c_double), ("inv_inertia", C.c_double * 3),
        ("precull_every", C.c_int32), ("precull_margin", C.c_double),
        ("kn", C.c_double), ("dn", C.c_double), ("eps_v", C.c_double), ("ct_max", C.c_double),
    ]


class OrcDynDesc(C.Structure):
    _fields_ = [
        ("nd", C.c_int32), ("parent", _ip), ("jtype", _ip), ("qadr", _ip), ("rel_pos", _dp), ("rel_quat", _dp),
        ("axis", _dp), ("jpos", _dp), ("qref", _dp), ("mass", _dp), ("ipos", _dp), ("inertia", _dp),
        ("damping", _dp), ("armature", _dp), ("limited", _ip), ("lo", _dp), ("hi", _dp),
        ("actuated", _ip), ("kp", _dp), ("force_lo", _dp), ("force_hi", _dp), ("gravcomp", _ip),
        ("gravity", C.c_double * 3), ("timestep", C.c_double), ("nsub", C.c_int32), ("obj", C.POINTER(OrcObjDesc)),
        ("ct", C.c_void_p),
    ]


class OrcCtDesc(C.Structure):
    _fields_ = [
        ("ns", C.c_int32), ("sh_body", _ip), ("sh_type", _ip), ("sh_size", _dp), ("sh_pos", _dp), ("sh_mat", _dp), ("sh_rbound", _dp),
        ("sh_feat0", _ip), ("nf", C.c_int32), ("ft_pos", _dp), ("ft_rad", _dp),
        ("np", C.c_int32), ("pr_f", _ip), ("pr_s", _ip), ("pr_par", _dp),
        ("obj_qadr", C.c_int32), ("obj_mass", C.c_double), ("obj_inertia", C.c_double * 3), ("obj_ipos", C.c_double * 3),
        ("obj_iquat", C.c_double * 4), ("obj_damping", C.c_double), ("obj_inv_mass", C.c_double), ("obj_inv_inertia", C.c_double * 3),
        ("obj_inv_mass_d", C.c_double), ("obj_inv_inertia_d", C.c_double * 3),
        ("maxcon", C.c_int32), ("maxpair", C.c_int32), ("iterations", C.c_int32), ("tolerance", C.c_double), ("inv_scale", C.c_double),
        ("precull_every", C.c_int32), ("precull_margin", C.c_double), ("near_every", C.c_int32), ("near_margin", C.c_double), ("warmstart", C.c_int32),
        ("solver", C.c_int32), ("limit_rows", C.c_int32), ("lim_par", C.c_double * 8), ("noslip_iterations", C.c_int32), ("noslip_tolerance", C.c_double),
        ("arena", C.c_int32),
    ]


class OrcCtStats(C.Structure):
    _fields_ = [("substeps", C.c_int64), ("contacts", C.c_int64), ("sweeps", C.c_int64), ("dropped", C.c_int64), ("max_contacts", C.c_int32),
                ("hot_pairs", C.c_int64), ("active_pairs", C.c_int64)]


class OracleDyn:
    """The servo dynamics of mopa_oracle_dyn.inc over a `mopa_rl_amd.dynamics.DynFacts` (plain arrays)."""

    def __init__(self, f, obj=None, ct=None):
        """f: DynFacts; obj: ObjFacts (stage B: the manipulated object moves under penalty contacts) or None;
        ct: CtFacts (stage C: contacts of arm and object behind the constraint solver) or None."""
        assert obj is None or ct is None
        self.f, self.nd = f, int(f.nd)
        self.nv = self.nd + (6 if (obj is not None or ct is not None) else 0)       # width of a qvel row: the dofs, then the object's (v, w)
        self._keep = []
        d = OrcDynDesc()

        def ip(a):
            a, p = _i(a); self._keep.append(a); return p

        def dp(a):
            a, p = _d(a); self._keep.append(a); return p

        d.nd = self.nd
        d.parent, d.jtype, d.qadr = ip(f.parent), ip(f.jtype), ip(f.qadr)
        d.rel_pos, d.rel_quat, d.axis, d.jpos, d.qref = dp(f.rel_pos), dp(f.rel_quat), dp(f.axis), dp(f.jpos), dp(f.qref)
        d.mass, d.ipos, d.inertia = dp(f.mass), dp(f.ipos), dp(f.inertia)
        d.damping, d.armature = dp(f.damping), dp(f.armature)
        d.limited, d.lo, d.hi = ip(f.limited), dp(f.lo), dp(f.hi)
        d.actuated, d.kp, d.force_lo, d.force_hi, d.gravcomp = ip(f.actuated), dp(f.kp), dp(f.force_lo), dp(f.force_hi), ip(f.gravcomp)
        d.gravity = (C.c_double * 3)(*[float(x) for x in f.gravity])
        d.timestep, d.nsub = float(f.timestep), int(f.nsub)
        self.obj = None
        if obj is not None:
            o = OrcObjDesc()
            o.qadr, o.mass, o.damping, o.rbound = int(obj.qadr), float(obj.mass), float(obj.damping), float(obj.rbound)
            o.inertia = (C.c_double * 3)(*[float(x) for x in obj.inertia])
            o.half = (C.c_double * 3)(*[float(x) for x in obj.half])
            o.nfeat, o.feat = len(obj.feat), dp(obj.feat)
            o.ncol, o.co_body, o.co_type = len(obj.co_body), ip(obj.co_body), ip(obj.co_type)
            o.co_size, o.co_pos, o.co_mat, o.co_mu, o.co_rbound = dp(obj.co_size), dp(obj.co_pos), dp(obj.co_mat), dp(obj.co_mu), dp(obj.co_rbound)
            o.kn, o.dn, o.eps_v, o.ct_max = float(obj.kn), float(obj.dn), float(obj.eps_v), float(obj.ct_max)
            o.inv_mass = float(obj.inv_mass)
            o.inv_inertia = (C.c_double * 3)(*[float(x) for x in obj.inv_inertia])
            o.precull_every, o.precull_margin = int(obj.precull_every), float(obj.precull_margin)
            self.obj = o
            d.obj = C.pointer(o)
        self.ct = None
        if ct is not None:
            c = OrcCtDesc()
            c.ns, c.sh_body, c.sh_type = len(ct.sh_body), ip(ct.sh_body), ip(ct.sh_type)
            c.sh_size, c.sh_pos, c.sh_mat, c.sh_rbound, c.sh_feat0 = dp(ct.sh_size), dp(ct.sh_pos), dp(ct.sh_mat), dp(ct.sh_rbound), ip(ct.sh_feat0)
            c.nf, c.ft_pos, c.ft_rad = len(ct.ft_rad), dp(ct.ft_pos), dp(ct.ft_rad)
            c.np, c.pr_f, c.pr_s, c.pr_par = len(ct.pr_f), ip(ct.pr_f), ip(ct.pr_s), dp(ct.pr_par)
            c.obj_qadr, c.obj_mass, c.obj_damping = int(ct.obj_qadr), float(ct.obj_mass), float(ct.obj_damping)
            c.obj_inertia = (C.c_double * 3)(*[float(x) for x in ct.obj_inertia])
            c.obj_ipos = (C.c_double * 3)(*[float(x) for x in ct.obj_ipos])
            c.obj_iquat = (C.c_double * 4)(*[float(x) for x in ct.obj_iquat])
            c.obj_inv_mass, c.obj_inv_mass_d = float(ct.obj_inv_mass), float(ct.obj_inv_mass_d)
            c.obj_inv_inertia = (C.c_double * 3)(*[float(x) for x in ct.obj_inv_inertia])
            c.obj_inv_inertia_d = (C.c_double * 3)(*[float(x) for x in ct.obj_inv_inertia_d])
            c.maxcon, c.maxpair, c.iterations = int(ct.maxcon), int(ct.maxpair), int(ct.iterations)
            c.tolerance, c.inv_scale = float(ct.tolerance), float(ct.inv_scale)
            c.precull_every, c.precull_margin, c.warmstart = int(ct.precull_every), float(ct.precull_margin), int(ct.warmstart)
            c.near_every, c.near_margin = int(ct.near_every), float(ct.near_margin)
            c.noslip_iterations, c.noslip_tolerance = int(ct.noslip_iterations), float(ct.noslip_tolerance)
            c.solver = int(ct.solver)
            c.limit_rows = int(ct.limit_rows)
            c.lim_par = (C.c_double * 8)(*[float(x) for x in ct.lim_par])
            c.arena = int(getattr(ct, "arena", 0))
            assert np.asarray(ct.pr_par).shape[1] == 12
            self.ct = c
            d.ct = C.cast(C.pointer(c), C.c_void_p)
        self.stats = OrcCtStats()
        self.desc = d

    def forward(self, qpos, qvel, want_M: bool = True):
        q, qp = _d(qpos)
        v, vp = _d(qvel)
        bias = np.zeros(self.nd)
        M = np.zeros((self.nd, self.nd)) if want_M else None
        lib().orc_dyn_forward(C.byref(self.desc), qp, vp, bias.ctypes.data_as(_dp), M.ctypes.data_as(_dp) if want_M else None)
        return bias, M

    def contacts(self, qpos):
        """stage C: the contacts of one configuration, rows [dist, pos 3, normal 3, shape F, shape S, feature]"""
        q, qp = _d(qpos)
        out = np.zeros((self.ct.maxcon, 10))
        n = lib().orc_ct_contacts(C.byref(self.desc), qp, out.ctypes.data_as(_dp))
        return out[:n]

    def step(self, qpos, qvel, bias_lag, ctrl, n: int = 1):
        """n sub-steps in place on copies; returns (qpos, qvel, bias_lag).  qvel: [nd], or [nd + 6] with an object."""
        q, v, lag = (np.array(x, dtype=np.float64, copy=True) for x in (qpos, qvel, bias_lag))
        c, cp = _d(ctrl)
        if self.ct is not None:
            assert len(v) == self.nd + 6
            lib().orc_ct_step(C.byref(self.desc), q.ctypes.data_as(_dp), v.ctypes.data_as(_dp), lag.ctypes.data_as(_dp), cp, int(n), C.byref(self.stats))
            return q, v, lag
        ov = v[self.nd:].ctypes.data_as(_dp) if (self.obj is not None and len(v) == self.nd + 6) else None
        lib().orc_dyn_step_obj(C.byref(self.desc), q.ctypes.data_as(_dp), v.ctypes.data_as(_dp), lag.ctypes.data_as(_dp), cp, int(n), ov)
        return q, v, lag


def atan2(y: float, x: float) -> float:
    return lib().orc_atan2(float(y), float(x))


def exp_(x: float) -> float:
    return lib().orc_exp(float(x))


def tanh_pos(x: float) -> float:
    return lib().orc_tanh_pos(float(x))


class OracleEnv:
    """E kinematic Sawyer envs (push / lift / assembly) stepped through orc_env_step (the checker of K4).
    `facts` is mopa_rl_amd.kinematic_env.EnvFacts (plain name->id data, no product code runs here)."""

    def __init__(self, scene: "OracleScene", facts, E: int, ac_scale=0.05, distance_threshold=0.06, success_reward=150.0,
                 max_episode_steps=250, dyn=None, obj=None, ct=None):
        self.scene, self.E, self.nq = scene, int(E), scene.nq
        self.dyn = OracleDyn(dyn, obj, ct) if dyn is not None else None     # DynFacts (+ ObjFacts): env.step runs the servo dynamics
        self._keep = []
        d = OrcEnvDesc()

        def ip(a):
            a, p = _i(a); self._keep.append(a); return p

        def dp(a):
            a, p = _d(a); self._keep.append(a); return p

        d.kind = int(facts.kind)
        d.n_arm, d.arm_qpos_idx = len(facts.arm_qpos_idx), ip(facts.arm_qpos_idx)
        d.n_grip, d.grip_qpos_idx = len(facts.grip_qpos_idx), ip(facts.grip_qpos_idx)
        d.n_act, d.act_qpos_idx, d.act_lo, d.act_hi = len(facts.act_qpos_idx), ip(facts.act_qpos_idx), dp(facts.act_lo), dp(facts.act_hi)
        d.n_frames, d.frame_body, d.frame_off = len(facts.frame_body), ip(facts.frame_body), dp(facts.frame_off)
        d.n_quats, d.quat_body = len(facts.quat_body), ip(facts.quat_body)
        d.n_touch, d.touch_geom, d.n_touch_left = len(facts.touch_geom), ip(facts.touch_geom), int(facts.n_touch_left)
        d.qpos_min, d.qpos_max, d.qpos_limited = dp(facts.qpos_min), dp(facts.qpos_max), ip(facts.qpos_limited)
        d.ac_scale, d.distance_threshold, d.success_reward = ac_scale, distance_threshold, success_reward
        d.max_episode_steps = max_episode_steps
        self.desc = d
        self.n_arm = d.n_arm
        self.obs_dim, self.action_dim = lib().orc_env_obs_dim(C.byref(d)), lib().orc_env_action_dim(C.byref(d))
        self.qpos = np.zeros((self.E, self.nq))
        self.prev_state = np.zeros((self.E, self.n_arm))
        self.has_prev = np.zeros(self.E, dtype=np.uint8)
        self.ep_len = np.zeros(self.E, dtype=np.int32)
        self.obs = np.zeros((self.E, self.obs_dim))
        self.reward = np.zeros(self.E)
        self.done = np.zeros(self.E, dtype=np.uint8)
        self.success = np.zeros(self.E, dtype=np.uint8)
        if self.dyn is not None:
            self.qvel = np.zeros((self.E, self.dyn.nv))
            self.bias_lag = np.zeros((self.E, self.dyn.nd))

    def _call(self, e, action, is_planner, move):
        L = lib()
        ap = None
        if action is not None:
            a, ap = _d(action[e])
        L.orc_env_step(self.scene._h, C.byref(self.desc), self.qpos[e].ctypes.data_as(_dp),
                       self.prev_state[e].ctypes.data_as(_dp), self.has_prev[e:e + 1].ctypes.data_as(_u8p),
                       self.ep_len[e:e + 1].ctypes.data_as(_ip), ap, int(is_planner), int(move),
                       self.obs[e].ctypes.data_as(_dp), self.reward[e:e + 1].ctypes.data_as(_dp),
                       self.done[e:e + 1].ctypes.data_as(_u8p), self.success[e:e + 1].ctypes.data_as(_u8p))

    def set_state(self, qpos):
        self.qpos[:] = qpos
        self.has_prev[:] = 0
        self.ep_len[:] = 0
        for e in range(self.E):
            self._call(e, None, 0, 1)
        if self.dyn is not None:      # reset: at rest, qfrc_bias of the `sim.forward()` that follows set_state
            self.qvel[:] = 0.0
            for e in range(self.E):
                self.bias_lag[e] = self.dyn.forward(self.qpos[e], self.qvel[e, :self.dyn.nd], want_M=False)[0]
        return self.obs

    def step(self, action, is_planner=False, move_mask=None, nthreads: int = 1):
        action = np.ascontiguousarray(action, dtype=np.float64)
        assert action.shape == (self.E, self.action_dim)
        mm = None if move_mask is None else np.ascontiguousarray(move_mask, dtype=np.uint8)
        if self.dyn is not None:
            lib().orc_env_step_dyn_batch(
                self.scene._h, C.byref(self.desc), C.byref(self.dyn.desc), self.E, self.qpos.ctypes.data_as(_dp),
                self.qvel.ctypes.data_as(_dp), self.bias_lag.ctypes.data_as(_dp), self.prev_state.ctypes.data_as(_dp),
                self.has_prev.ctypes.data_as(_u8p), self.ep_len.ctypes.data_as(_ip), action.ctypes.data_as(_dp), int(is_planner),
                mm.ctypes.data_as(_u8p) if mm is not None else None, self.obs.ctypes.data_as(_dp),
                self.reward.ctypes.data_as(_dp), self.done.ctypes.data_as(_u8p), self.success.ctypes.data_as(_u8p), int(nthreads))
            return self.obs, self.reward, self.done, self.success
        lib().orc_env_step_batch(
            self.scene._h, C.byref(self.desc), self.E, self.qpos.ctypes.data_as(_dp), self.prev_state.ctypes.data_as(_dp),
            self.has_prev.ctypes.data_as(_u8p), self.ep_len.ctypes.data_as(_ip), action.ctypes.data_as(_dp), int(is_planner),
            mm.ctypes.data_as(_u8p) if mm is not None else None, self.obs.ctypes.data_as(_dp),
            self.reward.ctypes.data_as(_dp), self.done.ctypes.data_as(_u8p), self.success.ctypes.data_as(_u8p), int(nthreads))
        return self.obs, self.reward, self.done, self.success


OraclePushEnv = OracleEnv      # earlier name


def sincos(x: float) -> Tuple[float, float]:
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def set_convex_axes_pretest(on: bool) -> None:
    """The cylinder classes' separating-axis pre-test (on by default; an accelerator that must never change a distance)."""
    lib().orc_set_convex_axes_pretest(1 if on else 0)


def geom_dist(t1, size1, pos1, mat1, t2, size2, pos2, mat2) -> float:
    k1 = [_d(np.asarray(x, dtype=np.float64).ravel()) for x in (size1, pos1, mat1)]
    k2 = [_d(np.asarray(x, dtype=np.float64).ravel()) for x in (size2, pos2, mat2)]
    return lib().orc_geom_dist(int(t1), k1[0][1], k1[1][1], k1[2][1], int(t2), k2[0][1], k2[1][1], k2[2][1])


def geom_dist_mesh(t1, size1, pos1, mat1, verts, pos2, mat2) -> float:
    """primitive (type t1) vs the convex hull of `verts` ([n,3], mesh frame) posed at (pos2, mat2)"""
    k1 = [_d(np.asarray(x, dtype=np.float64).ravel()) for x in (size1, pos1, mat1)]
    v = _d(np.asarray(verts, dtype=np.float64).reshape(-1, 3))
    k2 = [_d(np.asarray(x, dtype=np.float64).ravel()) for x in (pos2, mat2)]
    return lib().orc_geom_dist_mesh(int(t1), k1[0][1], k1[1][1], k1[2][1], v[1], len(v[0]), k2[0][1], k2[1][1])


def rng_uniform(seed: int, stream: int, counter: int) -> float:
    return lib().orc_rng_uniform(seed, stream, counter)


class OracleScene:
    """Oracle-side planner object: compiled model + passive/ignored lists + threshold."""

    def __init__(self, model, passive_joint_idx: Sequence[int], ignored_contacts: Sequence[Tuple[int, int]],
                 contact_threshold: float):
        m = model
        self.model = m
        self._keep = []

        def d(a):
            a, p = _d(a); self._keep.append(a); return p

        def i(a):
            a, p = _i(a); self._keep.append(a); return p

        ign = np.asarray(ignored_contacts, dtype=np.int32).reshape(-1, 2)
        pas = np.asarray(passive_joint_idx, dtype=np.int32)
        self._h = lib().orc_scene_create(
            m.nq, len(m.body_names), i(m.body_parent), d(m.body_pos), d(m.body_quat), i(m.body_jntadr), i(m.body_jntnum),
            len(m.jnt_names), i(m.jnt_type), i(m.jnt_qposadr), d(m.jnt_axis), d(m.jnt_pos), d(m.jnt_ref),
            i(m.jnt_limited), d(m.jnt_range),
            len(m.geom_type), i(m.geom_type), i(m.geom_body), i(m.geom_mjid), d(m.geom_size), d(m.geom_pos), d(m.geom_quat),
            len(m.pair_geom), i(m.pair_geom), len(pas), i(pas), len(ign), i(ign), float(contact_threshold))
        if len(getattr(m, "mesh_vertnum", ())):
            lib().orc_scene_set_meshes(self._h, len(m.mesh_vertnum), i(m.mesh_vertadr), i(m.mesh_vertnum), len(m.mesh_vert),
                                       d(m.mesh_vert), i(m.geom_dataid))
        self.nq = m.nq
        self.na = lib().orc_num_active(self._h)
        ai = np.zeros(self.na, dtype=np.int32)
        lib().orc_active_idx(self._h, ai.ctypes.data_as(_ip))
        self.active_idx = ai
        self.ngeom = len(m.geom_type)
        self.npair = len(m.pair_geom)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def fk(self, qpos):
        q, qp = _d(qpos)
        gpos = np.zeros((self.ngeom, 3)); gmat = np.zeros((self.ngeom, 9))
        lib().orc_fk(self._h, qp, gpos.ctypes.data_as(_dp), gmat.ctypes.data_as(_dp))
        return gpos, gmat.reshape(-1, 3, 3)

    def fk_bodies(self, qpos):
        q, qp = _d(qpos)
        nb = len(self.model.body_names)
        xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4))
        lib().orc_fk_bodies(self._h, qp, xpos.ctypes.data_as(_dp), xquat.ctypes.data_as(_dp))
        return xpos, xquat

    def pair_dist(self, qpos):
        q, qp = _d(qpos)
        out = np.zeros(self.npair)
        lib().orc_pair_dist(self._h, qp, out.ctypes.data_as(_dp))
        return out

    def is_valid(self, qpos) -> Tuple[bool, float]:
        q, qp = _d(qpos)
        md = C.c_double()
        v = lib().orc_is_valid(self._h, qp, C.byref(md))
        return bool(v), md.value

    def is_valid_batch(self, q_active, qpos_env, samples_per_env: Optional[int] = None, nthreads: int = 1,
                       want_min_dist: bool = True):
        qa, qap = _d(q_active)
        qe, qep = _d(np.atleast_2d(qpos_env))
        N = qa.shape[0]
        assert qa.shape[1] == self.na and qe.shape[1] == self.nq
        spe = samples_per_env if samples_per_env is not None else max(1, N // qe.shape[0])
        assert (N + spe - 1) // spe <= qe.shape[0]
        valid = np.zeros(N, dtype=np.uint8)
        md = np.zeros(N) if want_min_dist else None
        lib().orc_is_valid_batch(self._h, qap, qep, N, spe, valid.ctypes.data_as(_u8p),
                                 md.ctypes.data_as(_dp) if md is not None else None, nthreads)
        return valid, md

    def check_motion(self, qpos_env, qa, qb, resolution: float = 0.005):
        e, ep = _d(qpos_env); a, ap = _d(qa); b, bp = _d(qb)
        n = C.c_int64(0)
        r = lib().orc_check_motion(self._h, ep, ap, bp, resolution, C.byref(n))
        return bool(r), n.value

    def check_motion_batch(self, qa, qb, qpos_env, samples_per_env: Optional[int] = None,
                           resolution: float = 0.005, nthreads: int = 1):
        a, ap = _d(qa); b, bp = _d(qb)
        qe, qep = _d(np.atleast_2d(qpos_env))
        N = a.shape[0]
        spe = samples_per_env if samples_per_env is not None else max(1, N // qe.shape[0])
        valid = np.zeros(N, dtype=np.uint8)
        lib().orc_check_motion_batch(self._h, ap, bp, qep, N, spe, resolution, valid.ctypes.data_as(_u8p), nthreads)
        return valid

    def ik_solve(self, qpos, target_pos, joint_ids, site_body, site_off, max_steps=100, tol=1e-2, max_update_norm=2.0,
                 progress_thresh=20.0, reg_strength=3e-2, target_quat=None, site_quat=None, rot_weight=1.0):
        """damped-LS IK (env/inverse_kinematics.py:18-135 restated); returns (qpos, err_norm, steps, success)"""
        q = np.ascontiguousarray(qpos, dtype=np.float64).copy()
        t, tp = _d(target_pos); ji, jp = _i(joint_ids); so, sp = _d(site_off)
        tq, tqp = _d(target_quat) if target_quat is not None else (None, None)
        sq, sqp = _d(site_quat) if site_quat is not None else (None, None)
        en, st, su = C.c_double(0), C.c_int32(0), C.c_uint8(0)
        lib().orc_ik_solve(self._h, len(ji), jp, int(site_body), sp, sqp, q.ctypes.data_as(_dp), tp, tqp, float(rot_weight),
                           int(max_steps), float(tol), float(max_update_norm), float(progress_thresh), float(reg_strength),
                           C.byref(en), C.byref(st), C.byref(su))
        return q, en.value, st.value, bool(su.value)

    def plan_batch(self, start, goal, range_: float, resolution: float = 0.005, max_iters: int = 2000, max_nodes: int = 4096, seed: int = 0,
                   env_id_base: int = 0, max_path: int = 512, nthreads: int = 1):
        """E queries with OpenMP over queries inside the oracle; returns (status [E], path_len [E], n_checks [E])"""
        s = np.ascontiguousarray(start, dtype=np.float64); g = np.ascontiguousarray(goal, dtype=np.float64)
        E = len(s)
        st = np.zeros(E, dtype=np.int32); pl = np.zeros(E, dtype=np.int32); nc = np.zeros(E, dtype=np.int64)
        lib().orc_plan_batch(self._h, E, s.ctypes.data_as(_dp), g.ctypes.data_as(_dp), range_, resolution, max_iters, max_nodes, seed, env_id_base,
                             None, max_path, st.ctypes.data_as(_ip), pl.ctypes.data_as(_ip), nc.ctypes.data_as(C.c_void_p), int(nthreads))
        return st, pl, nc

    def plan(self, start, goal, range_: float, resolution: float = 0.005, max_iters: int = 2000,
             max_nodes: int = 4096, seed: int = 0, env_id: int = 0, max_path: int = 512):
        s, sp = _d(start); g, gp = _d(goal)
        path = np.zeros((max_path, self.nq))
        plen = C.c_int(0); nchk = C.c_int64(0); nit = C.c_int(0)
        st = lib().orc_plan(self._h, sp, gp, range_, resolution, max_iters, max_nodes, seed, env_id,
                            path.ctypes.data_as(_dp), max_path, C.byref(plen), C.byref(nchk), C.byref(nit))
        return st, path[:plen.value].copy(), nchk.value, nit.value
