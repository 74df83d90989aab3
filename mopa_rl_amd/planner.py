"""PyKinematicPlanner -- the drop-in boundary object.

Mirrors the Cython class of the reference (motion_planners/planner.pyx:31-52):
same constructor arguments (14, positional), same three methods, same return
conventions (``plan`` returns a list of rows, or ``[[-5]*nq]`` when the goal is
invalid / ``[[-4]*nq]`` when no exact solution was found --
motion_planners/KinematicPlanner.cpp:181-184,249-250).  The work happens in
libmopa_hip.so on the GPU; there is no CPU path.

Differences that cannot be hidden (DESIGN.md "Semantics"):
  * ``timelimit`` seconds are converted to an RRT-Connect iteration budget
    (``ITERS_PER_SECOND`` per second) -- the reference stops on wall-clock and
    is therefore not reproducible;
  * only ``algo == b"rrt_connect"`` is implemented (``b"rrt"`` means RRT* in
    the reference, KinematicPlanner.cpp:89,99-101);
  * ``opt``, ``num_actions``, ``goal_bias``, ``simplified_duration`` are
    accepted and ignored exactly as the reference ignores them
    (KinematicPlanner.cpp:42-61); ``is_simplified=True`` is rejected
    (PathSimplifier is randomised and time-bounded);
  * ``glue_bodies`` must be empty (the reference never passes any).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .scene import load_scene

#: RRT-Connect iterations granted per second of the reference's ``timelimit``
ITERS_PER_SECOND = int(os.environ.get("MOPA_ITERS_PER_SECOND", "2000"))
MAX_NODES = int(os.environ.get("MOPA_MAX_NODES", "4096"))
MAX_PATH = int(os.environ.get("MOPA_MAX_PATH", "1024"))


def _to_str(b) -> str:
    return b.decode("utf-8") if isinstance(b, (bytes, bytearray)) else str(b)


class PyKinematicPlanner:
    def __init__(self, xml_filename, algo, num_actions, opt, threshold, _range, passive_joint_idx, glue_bodies,
                 ignored_contacts, contact_threshold, goal_bias, is_simplified, simplified_duration, seed):
        self.xml_filename = _to_str(xml_filename)
        self.algo = _to_str(algo)
        self.num_actions = int(num_actions)
        self.opt = _to_str(opt)
        self.threshold = float(threshold)
        self._range = float(_range)
        self.passive_joint_idx = [int(i) for i in passive_joint_idx]
        self.glue_bodies = list(glue_bodies)
        self.ignored_contacts = [(int(a), int(b)) for a, b in ignored_contacts]
        self.contact_threshold = float(contact_threshold)
        self.isSimplified = bool(is_simplified)
        self.simplifiedDuration = float(simplified_duration)
        self.seed = int(seed)
        if self.glue_bodies:
            raise NotImplementedError("glue_bodies: never used by the reference callers, not implemented")
        if self.algo != "rrt_connect":
            raise NotImplementedError(f"algo={self.algo!r}: only 'rrt_connect' is implemented")
        if self.isSimplified:
            raise NotImplementedError("is_simplified=True (OMPL PathSimplifier) is not implemented")
        self._model = load_scene(self.xml_filename)
        self._scene = _lib.Scene(self._model, self.passive_joint_idx, self.ignored_contacts, self.contact_threshold,
                                 range_=self._range, resolution=0.005, seed=self.seed)
        self._plan_count = 0

    # -- reference API -----------------------------------------------------
    def isValidState(self, state_vec) -> bool:
        return self._scene.is_valid_state(np.asarray(state_vec, dtype=np.float64))

    def plan(self, start_vec, goal_vec, timelimit) -> List[List[float]]:
        start = np.asarray(start_vec, dtype=np.float64)
        goal = np.asarray(goal_vec, dtype=np.float64)
        max_iters = max(1, int(round(float(timelimit) * ITERS_PER_SECOND)))
        # every plan() call of one planner object draws a fresh sample stream
        status, path, _ = self._scene.plan(start, goal, max_iters=max_iters, max_nodes=MAX_NODES, max_path=MAX_PATH,
                                           seed=self.seed, env_id=self._plan_count)
        self._plan_count += 1
        nq = self._scene.nq
        if status == _lib.PLAN_INVALID_GOAL:
            return [[-5.0] * nq]
        if status != _lib.PLAN_OK:
            return [[-4.0] * nq]
        return path.tolist()

    def getPlannerStatus(self) -> bytes:
        return self._scene.planner_status()

    # -- extras used by the batched host code --------------------------------
    @property
    def scene(self) -> "_lib.Scene":
        return self._scene

    @property
    def model(self):
        return self._model
