"""`SamplingBasedPlanner`: the thin host wrapper the reference puts between its agents and the native planner
(reference motion_planners/sampling_based_planner.py:11-107, util/env.py:15-25 for the angle wrap).  Same constructor
arguments, methods and return tuples; the body is this repo's own formulation (array operations instead of per-waypoint
loops) of the same arithmetic:

* unlimited joints are wrapped into (-3.14, 3.14) before planning -- the reference wraps at 3.14, not at pi, and so
  does this;
* a planner result whose entries are all equal is a sentinel row (-5: invalid goal, -4: no exact solution);
* otherwise the trajectory handed to the agent is rebuilt from the successive differences of the planner's states,
  starting at the caller's un-wrapped `start`, with steps across the +-3.14 seam of an unlimited joint un-wrapped.
"""
from __future__ import annotations

import numpy as np

from .planner import PyKinematicPlanner

_SEAM = 3.14            # sic: the reference's wrap constant (util/env.py:15-25)
_INVALID_GOAL, _NO_EXACT = -5, -4


def joint_convert(angle):
    """Wrap one unlimited joint angle into (-3.14, 3.14) with the reference's floor-division rule."""
    half = _SEAM if angle > 0 else -_SEAM
    rem = angle % half
    return rem if (angle // half) % 2 == 0 else rem - half


class SamplingBasedPlanner:
    def __init__(self, config, xml_path, num_actions, non_limited_idx, planner_type=None, passive_joint_idx=[],
                 glue_bodies=[], ignored_contacts=[], contact_threshold=0.0, goal_bias=0.05, is_simplified=False,
                 simplified_duration=0.1, range_=None):
        self.config = config
        self.non_limited_idx = non_limited_idx
        algo = config.planner_type if planner_type is None else planner_type
        step = config.range if range_ is None else range_
        enc = lambda text: text.encode("utf-8")
        self.planner = PyKinematicPlanner(enc(xml_path), enc(algo), num_actions, enc(config.planner_objective), config.threshold,
                                          step, passive_joint_idx, glue_bodies, ignored_contacts, contact_threshold, goal_bias,
                                          is_simplified, simplified_duration, config.seed)

    # ------------------------------------------------------------------
    def convert_nonlimited(self, state):
        """In place: every unlimited joint of `state` wrapped by `joint_convert`."""
        for j in (self.non_limited_idx if self.non_limited_idx is not None else ()):
            state[j] = joint_convert(state[j])
        return state

    def isValidState(self, state):
        return self.planner.isValidState(state)

    def get_planner_status(self):
        return self.planner.getPlannerStatus().decode("utf-8")

    # ------------------------------------------------------------------
    def _unwrapped_steps(self, states):
        """Per-waypoint displacement of the planner's states, with the steps of unlimited joints that cross the seam
        replaced by the short way round (same expressions, evaluated in the same order, as the reference's loop)."""
        prev, cur = states[:-1], states[1:]
        step = cur - prev
        for j in (self.non_limited_idx if self.non_limited_idx is not None else ()):
            p, c = prev[:, j], cur[:, j]
            jump = np.abs(c - p) > _SEAM
            up = jump & (p > 0) & (c <= 0)                    # left through +3.14, came back in at -3.14
            down = jump & ~up & (p < 0) & (c > 0)
            step[up, j] = ((_SEAM - p[up]) + c[up]) + _SEAM
            step[down, j] = -((( _SEAM - c[down]) + p[down]) + _SEAM)
        return step

    def plan(self, start, goal, timelimit=1.0):
        """-> (traj, states, valid_state, exact).  `states` are the planner's (wrapped) waypoints; `traj` the same path as
        seen from the caller's `start`.  For a sentinel result both are the sentinel row."""
        q0 = self.convert_nonlimited(np.array(start, dtype=float, copy=True))
        q1 = self.convert_nonlimited(np.array(goal, dtype=float, copy=True))
        states = np.array(self.planner.plan(q0, q1, timelimit))
        if np.unique(states).size == 1:                       # one repeated value: a sentinel row
            tag = states[0][0]
            return states, states, tag != _INVALID_GOAL, tag != _NO_EXACT
        # running sum start + d1 + d2 + ... in that order (np.add.accumulate is strictly sequential)
        rows = np.vstack([np.asarray(start, dtype=float)[None, :], self._unwrapped_steps(states)])
        return np.add.accumulate(rows, axis=0), states, True, True
