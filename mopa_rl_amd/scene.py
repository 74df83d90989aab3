"""Scene registry: compiled models + the per-environment planner inputs.

The reference derives the planner's inputs from a live mujoco-py env in
rl/trainer.py:55-83:

    ignored_contacts  = manipulation geoms x geoms of the static bodies
                        (make_ordered_pair, util/misc.py:18-19)
    passive_joint_idx = every qpos address except env.ref_joint_pos_indexes
    non_limited_idx   = arm joints with jnt_limited == 0

Here the same three lists are derived from a :class:`CompiledModel` plus the
environment facts (names of static bodies / manipulation geoms / robot joints
and the planner defaults) restated from the reference's env and config files.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .mjcf import CompiledModel, compile_mjcf, GEOM_MESH

SCENE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")
_SAWYER_JOINTS = [f"right_j{i}" for i in range(7)]


@dataclass(frozen=True)
class EnvSpec:
    """Planner-relevant facts of one reference environment."""
    env: str
    scene: str                       # compiled scene / MJCF stem
    robot_joints: Tuple[str, ...]    # env.ref_joint_pos_indexes, by name
    static_bodies: Tuple[str, ...] = ()
    static_geoms: Tuple[str, ...] = ()
    manipulation_geoms: Tuple[str, ...] = ()
    manipulation_bodies: Tuple[str, ...] = ()
    init_qpos: Tuple[float, ...] = ()
    # config defaults (reference config/sawyer.py:76-112, config/pusher.py:57-93,
    # config/motion_planner.py:4-70)
    contact_threshold: float = -0.002
    range: float = 0.1
    simple_planner_range: float = 0.05
    timelimit: float = 1.0
    simple_planner_timelimit: float = 0.05
    step_size: float = 0.02
    joint_margin: float = 0.001
    ac_scale: float = 0.05
    action_range: float = 0.5
    omega: float = 0.7
    threshold: float = 0.0
    num_trials: int = 100


ENV_SPECS: Dict[str, EnvSpec] = {
    # env/sawyer/sawyer_push_obstacle.py:33-34,119-160
    "SawyerPushObstacle-v0": EnvSpec(
        env="SawyerPushObstacle-v0", scene="sawyer_push_obstacle", robot_joints=tuple(_SAWYER_JOINTS),
        static_bodies=("table", "bin1"), manipulation_geoms=("cube",),
        init_qpos=(4.57e-4, -0.114, 3.21e-2, -7.12e-3, 3.03e-2, -3.02e-2, -9.94e-3)),
    # env/sawyer/sawyer_lift_obstacle.py:13-14,163-189 (the can is a collidable mesh: collides through its convex hull)
    "SawyerLiftObstacle-v0": EnvSpec(
        env="SawyerLiftObstacle-v0", scene="sawyer_lift_obstacle", robot_joints=tuple(_SAWYER_JOINTS),
        static_bodies=("table", "bin1"), manipulation_geoms=("cube",),
        init_qpos=(-0.0305, -0.7325, 0.03043, 1.16124, 1.87488, 0.0, 0.0)),
    # env/sawyer/sawyer_assembly_obstacle.py:19-20,62-95
    "SawyerAssemblyObstacle-v0": EnvSpec(
        env="SawyerAssemblyObstacle-v0", scene="sawyer_assembly_obstacle", robot_joints=tuple(_SAWYER_JOINTS),
        static_bodies=("table",), manipulation_bodies=("furniture", "0_part0", "1_part1", "4_part4", "2_part2"),
        init_qpos=(0.427, 0.13, 0.0557, 0.114, -0.0622, 0.0276, 0.00356)),
    # env/pusher/pusher_obstacle.py:74-79,138-151 ; config/pusher.py
    "PusherObstacle-v0": EnvSpec(
        env="PusherObstacle-v0", scene="pusher_obstacle", robot_joints=("joint0", "joint1", "joint2", "joint3"),
        static_geoms=tuple(f"obstacle{i}_geom" for i in range(1, 8)), manipulation_geoms=("box",),
        contact_threshold=-0.0015, range=0.2, simple_planner_range=0.1, simple_planner_timelimit=0.02,
        step_size=0.04, joint_margin=0.0, ac_scale=0.1, action_range=1.0),
}


def scene_path(scene: str) -> str:
    return os.path.join(SCENE_DIR, scene + ".json")


def load_scene(scene_or_path: str) -> CompiledModel:
    """Load a compiled scene by stem name, compiled-JSON path, or MJCF path."""
    if scene_or_path.endswith(".xml"):
        stem = os.path.splitext(os.path.basename(scene_or_path))[0]
        if os.path.exists(scene_or_path):
            return compile_mjcf(scene_or_path)
        # the reference passes an XML path (planner.pyx:33); on machines without
        # the asset tree fall back to the compiled copy of the same scene.
        if os.path.exists(scene_path(stem)):
            return CompiledModel.load(scene_path(stem))
        raise FileNotFoundError(scene_or_path)
    if scene_or_path.endswith(".json"):
        return CompiledModel.load(scene_or_path)
    return CompiledModel.load(scene_path(scene_or_path))


def make_ordered_pair(a: int, b: int) -> Tuple[int, int]:
    """reference util/misc.py:18-19"""
    return (min(a, b), max(a, b))


@dataclass
class PlannerInputs:
    model: CompiledModel
    spec: EnvSpec
    ref_joint_pos_indexes: List[int]
    passive_joint_idx: List[int]
    ignored_contacts: List[Tuple[int, int]]
    non_limited_idx: np.ndarray
    jnt_minimum: np.ndarray
    jnt_maximum: np.ndarray
    is_jnt_limited: np.ndarray


def planner_inputs(env: str, model: Optional[CompiledModel] = None) -> PlannerInputs:
    """Restates rl/trainer.py:55-83 on a compiled model."""
    spec = ENV_SPECS[env]
    m = model if model is not None else load_scene(spec.scene)
    ref_idx = [m.get_joint_qpos_addr(j) for j in spec.robot_joints]
    manip = [m.geom_name2id(g) for g in spec.manipulation_geoms]
    manip += m.geoms_of_bodies(list(spec.manipulation_bodies)) if spec.manipulation_bodies else []
    static = [m.geom_name2id(g) for g in spec.static_geoms]
    static += m.geoms_of_bodies(list(spec.static_bodies)) if spec.static_bodies else []
    ignored = [make_ordered_pair(a, b) for a in manip for b in static]
    passive = [i for i in range(m.nq) if i not in ref_idx]
    jids = [m.joint_name2id(j) for j in spec.robot_joints]
    limited = m.jnt_limited[jids].astype(bool)
    lo = m.jnt_range[jids, 0].copy()
    hi = m.jnt_range[jids, 1].copy()
    # env/base.py:85-86: unlimited joints get +-3.14 in Python
    lo[~limited] = -3.14
    hi[~limited] = 3.14
    return PlannerInputs(model=m, spec=spec, ref_joint_pos_indexes=ref_idx, passive_joint_idx=passive,
                         ignored_contacts=ignored, non_limited_idx=np.where(~limited)[0],
                         jnt_minimum=lo, jnt_maximum=hi, is_jnt_limited=limited)


def has_mesh_collider(m: CompiledModel) -> bool:
    return bool(np.any(m.geom_type == GEOM_MESH))


def default_qpos(env: str, model: Optional[CompiledModel] = None) -> np.ndarray:
    """qpos0 with the arm at the env's init_qpos (env/sawyer/*.py `_reset`)."""
    pi = planner_inputs(env, model)
    q = pi.model.qpos0.copy()
    if pi.spec.init_qpos:
        q[pi.ref_joint_pos_indexes] = pi.spec.init_qpos
    return q


def qpos_joint_arrays(model: CompiledModel, float32_limits: bool = False):
    """The per-qpos joint bookkeeping of reference env/base.py:62-88: `jnt_indices` (joint id of every qpos
    address; free joints repeat 7x, ball joints 4x), `jnt_minimum` / `jnt_maximum` per joint (unlimited joints get
    +-3.14) and `is_jnt_limited`.  `float32_limits=True` reproduces what the SAC/TD3 agents actually hold: they take
    the limits from `joint_space['default'].low/high`, a float32 gym Box (rl/sac_agent.py:56-57)."""
    from .mjcf import JNT_BALL, JNT_FREE
    idx = []
    for j, t in enumerate(model.jnt_type):
        idx.extend([j] * (7 if t == JNT_FREE else 4 if t == JNT_BALL else 1))
    lim = model.jnt_limited.astype(bool)
    lo = np.where(lim, model.jnt_range[:, 0], -3.14)
    hi = np.where(lim, model.jnt_range[:, 1], 3.14)
    if float32_limits:
        lo, hi = lo.astype(np.float32), hi.astype(np.float32)
    return np.array(idx, dtype=np.int64), lo, hi, lim
