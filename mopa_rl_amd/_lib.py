"""ctypes binding of libmopa_hip.so (the C ABI in include/mopa_hip.h).

There is no CPU fallback: if the shared library is missing, or no HIP device
is visible when a scene is created, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOPA_HIP_LIB", os.path.join(_HERE, "csrc", "libmopa_hip.so"))   # env override: A/B builds

MOPA_OK = 0
MOPA_FAR = 1.0e10
PLAN_OK, PLAN_NO_EXACT, PLAN_INVALID_GOAL = 0, -4, -5

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# every symbol include/mopa_hip.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "mopa_last_error", "mopa_version", "mopa_device_count", "mopa_scene_create", "mopa_scene_destroy",
    "mopa_scene_num_active", "mopa_scene_active_idx", "mopa_scene_num_pairs", "mopa_scene_lds_bytes", "mopa_scene_valid_kernel",
    "mopa_is_valid_batch", "mopa_check_motion_batch", "mopa_plan_batch", "mopa_pullback_batch", "mopa_is_valid_state", "mopa_plan",
    "mopa_planner_status", "mopa_debug_fk", "mopa_debug_pair_dist",
    "mopa_env_create", "mopa_env_destroy", "mopa_env_obs_dim", "mopa_env_action_dim", "mopa_env_step_batch", "mopa_env_exec_batch", "mopa_env_desired_batch",
    "mopa_env_attach_dynamics", "mopa_env_attach_contacts", "mopa_env_set_contact_stats", "mopa_rollout_stage", "mopa_rollout_pool_pick", "mopa_rollout_step_size", "mopa_ct_desc_size", "mopa_env_contact_arena", "mopa_env_dyn_dofs", "mopa_env_dyn_qvel_width", "mopa_env_dyn_forward_batch", "mopa_env_dyn_substeps_batch", "mopa_env_step_dyn_batch",
    "mopa_ik_create", "mopa_ik_destroy", "mopa_ik_solve_batch", "mopa_ik_site_pose_batch", "mopa_ik_targets_batch",
    "mopa_paths_unwrap_batch", "mopa_paths_unwrap_seam_batch", "mopa_paths_walk_batch", "mopa_paths_assemble_batch", "mopa_interpolate_batch",
]


class MopaError(RuntimeError):
    pass


class MopaModel(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nbody", C.c_int32), ("njnt", C.c_int32), ("ngeom", C.c_int32), ("npair", C.c_int32),
        ("body_parent", _ip), ("body_pos", _dp), ("body_quat", _dp), ("body_jntadr", _ip), ("body_jntnum", _ip),
        ("jnt_type", _ip), ("jnt_qposadr", _ip), ("jnt_axis", _dp), ("jnt_pos", _dp), ("jnt_ref", _dp),
        ("jnt_limited", _ip), ("jnt_range", _dp),
        ("geom_type", _ip), ("geom_body", _ip), ("geom_mjid", _ip), ("geom_size", _dp), ("geom_pos", _dp),
        ("geom_quat", _dp), ("pair_geom", _ip),
        ("nmesh", C.c_int32), ("nmeshvert", C.c_int32), ("mesh_vertadr", _ip), ("mesh_vertnum", _ip), ("mesh_vert", _dp),
        ("geom_dataid", _ip),
    ]


class MopaSceneDesc(C.Structure):
    _fields_ = [
        ("model", MopaModel), ("n_passive", C.c_int32), ("passive_qpos_idx", _ip), ("n_ignored", C.c_int32),
        ("ignored_pairs", _ip), ("contact_threshold", C.c_double), ("range", C.c_double), ("resolution", C.c_double),
        ("seed", C.c_uint64), ("device", C.c_int32), ("pair_cull_radius", _dp),
    ]


class MopaEnvDesc(C.Structure):
    _fields_ = [
        ("model", MopaModel), ("kind", C.c_int32), ("n_arm", C.c_int32), ("arm_qpos_idx", _ip), ("n_grip", C.c_int32), ("grip_qpos_idx", _ip),
        ("n_act", C.c_int32), ("act_qpos_idx", _ip), ("act_ctrl_lo", _dp), ("act_ctrl_hi", _dp),
        ("n_frames", C.c_int32), ("frame_body", _ip), ("frame_off", _dp), ("n_quats", C.c_int32), ("quat_body", _ip),
        ("n_touch", C.c_int32), ("touch_geom", _ip), ("n_touch_left", C.c_int32),
        ("qpos_min", _dp), ("qpos_max", _dp), ("qpos_limited", _ip),
        ("ac_scale", C.c_double), ("distance_threshold", C.c_double), ("success_reward", C.c_double),
        ("max_episode_steps", C.c_int32), ("device", C.c_int32),
    ]


class MopaObjDesc(C.Structure):
    _fields_ = [
        ("qadr", C.c_int32), ("mass", C.c_double), ("inertia", C.c_double * 3), ("damping", C.c_double),
        ("half", C.c_double * 3), ("rbound", C.c_double), ("nfeat", C.c_int32), ("feat", _dp),
        ("ncol", C.c_int32), ("co_body", _ip), ("co_type", _ip), ("co_size", _dp), ("co_pos", _dp), ("co_mat", _dp),
        ("co_mu", _dp), ("co_rbound", _dp), ("inv_mass", C.c_double), ("inv_inertia", C.c_double * 3),
        ("precull_every", C.c_int32), ("precull_margin", C.c_double),
        ("kn", C.c_double), ("dn", C.c_double), ("eps_v", C.c_double), ("ct_max", C.c_double),
    ]


class MopaDynDesc(C.Structure):
    _fields_ = [
        ("nd", C.c_int32), ("parent", _ip), ("jtype", _ip), ("qadr", _ip), ("rel_pos", _dp), ("rel_quat", _dp),
        ("axis", _dp), ("jpos", _dp), ("qref", _dp), ("mass", _dp), ("ipos", _dp), ("inertia", _dp),
        ("damping", _dp), ("armature", _dp), ("limited", _ip), ("lo", _dp), ("hi", _dp),
        ("actuated", _ip), ("kp", _dp), ("force_lo", _dp), ("force_hi", _dp), ("gravcomp", _ip),
        ("gravity", C.c_double * 3), ("timestep", C.c_double), ("nsub", C.c_int32), ("obj", C.POINTER(MopaObjDesc)),
    ]


class MopaRolloutStep(C.Structure):
    """include/mopa_hip.h MopaRolloutStep: every pointer as an address (c_void_p), filled from tensors' data_ptr()"""
    _I32 = ("nq", "n_arm", "ac_dim", "ac_stride", "adim", "obs_dim", "K", "discrete", "normal_space")
    _F64 = ("omega", "ac_scale", "action_range", "omega_over_scale", "one_minus_omega", "range_minus_scale")
    _PTR = ("lim_lo", "lim_hi", "lo_state", "hi_state", "lo_shrunk", "hi_shrunk", "safe_q",
            "qpos", "obs", "reward", "done", "success", "has_prev",
            "busy", "pool_mask", "interp_overflow", "wait_since", "t_dev", "t_env", "pend_type", "q_cur", "q_tgt", "pend_ob", "pend_ac",
            "c_rl", "c_interp", "c_mp_fail", "c_invalid",
            "ac", "ac_type_in", "a_in",
            "prev_ob", "ac_tr", "a", "extra_ac", "target", "cur_m", "cur_v", "tgt_v", "traj",
            "active", "is_pl", "pv", "plan_ok", "ac_type", "path_len", "tv", "ok", "nst", "tlen", "finished",
            "act0", "flags", "sitting", "stepped", "is_pl_out", "plen_m", "last_extra", "rew", "done_out", "intra", "ob_next", "success_out", "retry_mask", "pool_counts")
    _fields_ = ([("E", C.c_int64)] + [(k, C.c_int32) for k in _I32] + [(k, C.c_double) for k in _F64] + [(k, C.c_void_p) for k in _PTR])


class MopaCtDesc(C.Structure):
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


class MopaIkDesc(C.Structure):
    _fields_ = [("model", MopaModel), ("n_joints", C.c_int32), ("joint_ids", _ip), ("site_body", C.c_int32),
                ("site_off", C.c_double * 3), ("site_quat", C.c_double * 4), ("device", C.c_int32)]


class MopaPlanParams(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("max_nodes", C.c_int32), ("max_path", C.c_int32), ("seed", C.c_uint64),
                ("env_id_base", C.c_uint64), ("env_ids_dev", C.c_void_p), ("seeds_dev", C.c_void_p), ("max_workgroups", C.c_int32), ("exclusive_cu", C.c_int32),
                ("tree_q_dev", C.c_void_p), ("tree_p_dev", C.c_void_p), ("state_dev", C.c_void_p),
                ("resume_tree_q", C.c_void_p), ("resume_tree_p", C.c_void_p), ("resume_state", C.c_void_p)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libmopa_hip.so (built in-tree by __graft_entry__.build / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MopaError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mopa_rl_amd/csrc`). There is no CPU fallback.")
    # One HIP runtime per process: torch ships its own libamdhip64.so.7 and must be loaded first so
    # that libmopa_hip.so binds to that same copy (two runtimes in one process cannot both own the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.mopa_last_error.restype = C.c_char_p
    L.mopa_version.restype = C.c_char_p
    L.mopa_device_count.restype = C.c_int
    L.mopa_scene_create.argtypes = [C.POINTER(MopaSceneDesc), C.POINTER(vp)]
    L.mopa_scene_destroy.argtypes = [vp]
    L.mopa_scene_destroy.restype = None
    L.mopa_scene_num_active.argtypes = [vp]
    L.mopa_scene_active_idx.argtypes = [vp, _ip]
    L.mopa_scene_num_pairs.argtypes = [vp]
    L.mopa_scene_lds_bytes.argtypes = [vp]
    L.mopa_is_valid_batch.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, vp, vp, vp]
    L.mopa_check_motion_batch.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, vp, vp]
    L.mopa_plan_batch.argtypes = [vp, vp, vp, C.c_int64, C.POINTER(MopaPlanParams), vp, vp, vp, vp, vp]
    L.mopa_pullback_batch.argtypes = [vp, vp, vp, C.c_int64, C.c_double, C.c_int32, vp, vp, vp]
    L.mopa_is_valid_state.argtypes = [vp, _dp, C.POINTER(C.c_int32), _dp]
    L.mopa_plan.argtypes = [vp, _dp, _dp, C.POINTER(MopaPlanParams), _dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                            C.POINTER(C.c_int64)]
    L.mopa_planner_status.argtypes = [vp]
    L.mopa_planner_status.restype = C.c_char_p
    L.mopa_debug_fk.argtypes = [vp, _dp, _dp, _dp]
    L.mopa_debug_pair_dist.argtypes = [vp, _dp, _dp]
    L.mopa_env_create.argtypes = [C.POINTER(MopaEnvDesc), C.POINTER(vp)]
    L.mopa_env_destroy.argtypes = [vp]
    L.mopa_env_destroy.restype = None
    L.mopa_env_step_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.mopa_env_obs_dim.argtypes = [vp]
    L.mopa_env_action_dim.argtypes = [vp]
    L.mopa_env_exec_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mopa_env_desired_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_int32, vp, vp]
    L.mopa_env_attach_dynamics.argtypes = [vp, C.POINTER(MopaDynDesc)]
    L.mopa_env_attach_contacts.argtypes = [vp, C.POINTER(MopaCtDesc)]
    L.mopa_env_set_contact_stats.argtypes = [vp, vp]
    L.mopa_rollout_stage.argtypes = [C.POINTER(MopaRolloutStep), C.c_int32, vp]
    L.mopa_rollout_pool_pick.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.c_int64] + [vp] * 4 + [C.c_int64] + [vp] * 7
    L.mopa_rollout_step_size.argtypes = []
    L.mopa_ct_desc_size.argtypes = []
    L.mopa_env_contact_arena.argtypes = [vp]
    L.mopa_env_contact_arena.restype = C.c_int
    L.mopa_env_dyn_dofs.argtypes = [vp]
    L.mopa_env_dyn_qvel_width.argtypes = [vp]
    L.mopa_env_dyn_forward_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.mopa_env_dyn_substeps_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_int32, vp]
    L.mopa_env_step_dyn_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.mopa_ik_create.argtypes = [C.POINTER(MopaIkDesc), C.POINTER(vp)]
    L.mopa_ik_destroy.argtypes = [vp]
    L.mopa_ik_destroy.restype = None
    L.mopa_ik_solve_batch.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                      vp, vp, vp, vp]
    L.mopa_ik_site_pose_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    L.mopa_ik_targets_batch.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), vp, vp, vp]
    L.mopa_scene_valid_kernel.argtypes = [vp, C.c_int64, C.c_char_p, C.c_int32]
    i32, i64, f64 = C.c_int32, C.c_int64, C.c_double
    L.mopa_paths_unwrap_batch.argtypes = [C.c_int, i64, i32, i32, vp, i32, vp, vp, vp, f64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mopa_paths_unwrap_seam_batch.argtypes = [C.c_int, i64, i32, i32, vp, i32, vp, vp, vp, f64, i32, vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp]
    L.mopa_paths_walk_batch.argtypes = [C.c_int, i64, i32, i32, vp, i32, vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mopa_paths_assemble_batch.argtypes = [C.c_int, i64, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.mopa_interpolate_batch.argtypes = [vp, i64, i32, i32, vp, vp, f64, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != MOPA_OK:
        raise MopaError(f"libmopa_hip error {rc}: {lib().mopa_last_error().decode()}")


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def model_struct(m, keep: list, pair_geom=None) -> MopaModel:
    """MopaModel view of a CompiledModel; the numpy buffers it points at are appended to `keep`.
    pair_geom: a (reduced) candidate-pair list to hand over instead of the model's."""
    def d(a):
        a, p = _d(a); keep.append(a); return p

    def i(a):
        a, p = _i(a); keep.append(a); return p

    pairs = m.pair_geom if pair_geom is None else pair_geom
    return MopaModel(
        m.nq, len(m.body_names), len(m.jnt_names), len(m.geom_type), len(pairs),
        i(m.body_parent), d(m.body_pos), d(m.body_quat), i(m.body_jntadr), i(m.body_jntnum),
        i(m.jnt_type), i(m.jnt_qposadr), d(m.jnt_axis), d(m.jnt_pos), d(m.jnt_ref), i(m.jnt_limited), d(m.jnt_range),
        i(m.geom_type), i(m.geom_body), i(m.geom_mjid), d(m.geom_size), d(m.geom_pos), d(m.geom_quat), i(pairs),
        len(m.mesh_vertnum), len(m.mesh_vert), i(m.mesh_vertadr), i(m.mesh_vertnum), d(m.mesh_vert), i(m.geom_dataid))


class Scene:
    """Owns one MopaScene* (== one KinematicPlanner instance of the reference)."""

    def __init__(self, model, passive_joint_idx, ignored_contacts, contact_threshold: float, range_: float = 0.1,
                 resolution: float = 0.005, seed: int = 0, device: int = -1, prune_pairs: Optional[bool] = None):
        """prune_pairs (default: on, MOPA_PRUNE_PAIRS=0 turns it off): candidate pairs that the scene's compile-time proof
        (tools/prove_separated_pairs.py, `meta["never_violating_pairs"]`: a Lipschitz branch-and-bound over the joint ranges)
        shows can never reach the contact threshold are not handed to the kernels at all.  Verdicts and depths are
        unchanged for joint values inside their ranges inflated by the proof's guard band (`meta["prune_guard_band"]`: 0.05 rad /
        2 mm) -- the states OMPL samples and the rollouts clip to, and what MuJoCo's soft joint limits let through.  The reference's
        isValidState takes ANY state (motion_planners/KinematicPlanner.cpp:253-286): `is_valid_state` here, and `BatchPlanner.
        is_valid(guard=True)`, send a state with a joint beyond range + band through a sibling scene with the full pair list."""
        L = lib()
        m = model
        keep = []
        if prune_pairs is None:
            prune_pairs = os.environ.get("MOPA_PRUNE_PAIRS", "1") != "0"
        pairs = np.asarray(m.pair_geom, dtype=np.int32).reshape(-1, 2)
        meta = getattr(m, "meta", {})
        never = list(meta.get("never_violating_pairs") or [])
        at = meta.get("never_violating_pairs_thr") or {}
        if at and float(contact_threshold) <= float(at.get("threshold", -np.inf)):
            never += list(at.get("pairs") or [])      # may touch, never reach a threshold this negative
        self.npair_pruned = 0
        self._full = None
        self._ctor = (model, list(passive_joint_idx), list(ignored_contacts), float(contact_threshold), float(range_), float(resolution), int(seed),
                      int(device))
        # the box inside which the pruning is proven: range + guard band of every limited joint (scenes proven without a band: the range)
        band = meta.get("prune_guard_band") or {}
        lim = np.asarray(m.jnt_limited).astype(bool) & (np.asarray(m.jnt_type) != 0)
        bw = np.where(np.asarray(m.jnt_type) == 2, float(band.get("slide", 0.0)), float(band.get("hinge", 0.0)))
        self.guard_adr = np.asarray(m.jnt_qposadr, dtype=np.int64)[lim]
        self.guard_lo = (np.asarray(m.jnt_range, dtype=np.float64)[:, 0] - bw)[lim]
        self.guard_hi = (np.asarray(m.jnt_range, dtype=np.float64)[:, 1] + bw)[lim]
        if prune_pairs and len(never) and float(contact_threshold) <= 0.0:
            drop = {(int(a), int(b)) for a, b in never} | {(int(b), int(a)) for a, b in never}
            keep_row = np.array([(int(a), int(b)) not in drop for a, b in pairs], dtype=bool)
            self.npair_pruned = int((~keep_row).sum())
            pairs = np.ascontiguousarray(pairs[keep_row])

        def d(a):
            a, p = _d(a); keep.append(a); return p

        def i(a):
            a, p = _i(a); keep.append(a); return p

        ign = np.asarray(list(ignored_contacts), dtype=np.int32).reshape(-1, 2)
        pas = np.asarray(list(passive_joint_idx), dtype=np.int32)
        desc = MopaSceneDesc()
        desc.model = model_struct(m, keep, pair_geom=pairs)
        # per-pair bound on the centre distance at which the pair can reach the threshold (same proof machinery): the FP32
        # broad phase culls with it instead of the bounding-sphere sum (closed gripper fingers: 9 mm instead of 10 cm)
        cr = meta.get("pair_cull_radius") or {}
        self.npair_tightened = 0
        if prune_pairs and cr and float(contact_threshold) <= float(cr.get("threshold", -np.inf)):
            rad = {(int(a), int(b)): float(r) for a, b, r in cr.get("pairs") or []}
            arr = np.array([rad.get((int(a), int(b)), rad.get((int(b), int(a)), 0.0)) for a, b in pairs], dtype=np.float64)
            if (arr > 0).any():
                self.npair_tightened = int((arr > 0).sum())
                desc.pair_cull_radius = d(arr)
        desc.n_passive = len(pas)
        desc.passive_qpos_idx = i(pas)
        desc.n_ignored = len(ign)
        desc.ignored_pairs = i(ign)
        desc.contact_threshold = float(contact_threshold)
        desc.range = float(range_)
        desc.resolution = float(resolution)
        desc.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        desc.device = int(device)
        h = C.c_void_p()
        check(L.mopa_scene_create(C.byref(desc), C.byref(h)))
        self._h = h
        self.model = m
        self.nq = m.nq
        self.na = L.mopa_scene_num_active(h)
        ai = np.zeros(self.na, dtype=np.int32)
        check(L.mopa_scene_active_idx(h, ai.ctypes.data_as(_ip)))
        self.active_idx = ai
        self.npair_checked = L.mopa_scene_num_pairs(h)
        self.lds_bytes = L.mopa_scene_lds_bytes(h)
        self.ngeom = len(m.geom_type)
        self.npair = len(m.pair_geom)
        self.seed = int(seed)

    def close(self):
        if getattr(self, "_full", None) is not None:
            self._full.close()
            self._full = None
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mopa_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def full(self) -> "Scene":
        """the sibling scene with the FULL candidate-pair list (created on first use): where states outside the pruning proof's
        box -- a joint beyond its range + guard band -- are evaluated"""
        if not self.npair_pruned:
            return self
        if self._full is None:
            mdl, pas, ign, thr, rng, res, seed, dev = self._ctor
            self._full = Scene(mdl, pas, ign, thr, range_=rng, resolution=res, seed=seed, device=dev, prune_pairs=False)
        return self._full

    def outside_guard(self, qpos_rows) -> np.ndarray:
        """bool per qpos row: some limited joint lies beyond its range + guard band (the pruned pair list is not proven there)"""
        q = np.atleast_2d(np.asarray(qpos_rows, dtype=np.float64))[:, self.guard_adr]
        return ((q < self.guard_lo) | (q > self.guard_hi)).any(axis=1)

    # ---- single-state forms (host pointers) ----
    def is_valid_state(self, qpos, want_min_dist: bool = False):
        q, qp = _d(qpos)
        if q.shape != (self.nq,):
            raise MopaError(f"state has dimension {q.shape}, expected nq={self.nq}")
        if self.npair_pruned and bool(self.outside_guard(q)[0]):
            return self.full().is_valid_state(q, want_min_dist)
        v = C.c_int32(0)
        md = C.c_double(0.0)
        check(lib().mopa_is_valid_state(self._h, qp, C.byref(v), C.byref(md) if want_min_dist else None))
        return (bool(v.value), md.value) if want_min_dist else bool(v.value)

    def plan(self, start, goal, max_iters: int, max_nodes: int = 4096, max_path: int = 512, seed: Optional[int] = None,
             env_id: int = 0):
        s, sp = _d(start)
        g, gp = _d(goal)
        prm = MopaPlanParams(int(max_iters), int(max_nodes), int(max_path),
                             int(self.seed if seed is None else seed) & 0xFFFFFFFFFFFFFFFF, int(env_id))
        path = np.zeros((max_path, self.nq))
        plen, st, chk = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        check(lib().mopa_plan(self._h, sp, gp, C.byref(prm), path.ctypes.data_as(_dp), C.byref(plen), C.byref(st),
                              C.byref(chk)))
        return st.value, path[:plen.value].copy(), chk.value

    def valid_kernel(self, n_states: int) -> str:
        """name of the validity kernel `mopa_is_valid_batch` dispatches for a batch of n_states"""
        buf = C.create_string_buffer(32)
        check(lib().mopa_scene_valid_kernel(self._h, int(n_states), buf, 32))
        return buf.value.decode()

    def planner_status(self) -> bytes:
        return lib().mopa_planner_status(self._h)

    def debug_fk(self, qpos):
        q, qp = _d(qpos)
        gpos = np.zeros((self.ngeom, 3)); gmat = np.zeros((self.ngeom, 9))
        check(lib().mopa_debug_fk(self._h, qp, gpos.ctypes.data_as(_dp), gmat.ctypes.data_as(_dp)))
        return gpos, gmat.reshape(-1, 3, 3)

    def debug_pair_dist(self, qpos):
        q, qp = _d(qpos)
        out = np.zeros(self.npair)
        check(lib().mopa_debug_pair_dist(self._h, qp, out.ctypes.data_as(_dp)))
        return out
