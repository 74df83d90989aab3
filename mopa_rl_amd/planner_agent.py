"""`PlannerAgent`: what the reference's SAC/TD3 agents hold as `self._planner` / `self._simple_planner`
(reference rl/planner_agent.py:9-58): owns one `SamplingBasedPlanner` and turns its 4-tuple into the agent-side
`(trajectory, success, valid, exact)` -- on success the first row (the start state itself) is dropped."""
from __future__ import annotations

from .sampling_based_planner import SamplingBasedPlanner


def action_size(ac_space) -> int:
    """Number of action dimensions (reference util/gym.py `action_size`).  Accepts a gym Dict / Box-like object (anything
    with `.spaces` or `.shape`) or simply an int, so that gym itself is not needed."""
    if isinstance(ac_space, int):
        return ac_space
    spaces = getattr(ac_space, "spaces", None)
    if spaces is None:
        return int(ac_space.shape[0])
    return sum(int(sp.n) if hasattr(sp, "n") else int(sp.shape[0]) for sp in spaces.values())


class PlannerAgent:
    def __init__(self, config, ac_space, non_limited_idx=None, passive_joint_idx=[], ignored_contacts=[],
                 planner_type=None, goal_bias=0.05, is_simplified=False, simplified_duration=0.1, range_=None):
        self._config = config
        self._is_simplified, self._simplified_duration = is_simplified, simplified_duration
        self.planner = SamplingBasedPlanner(config, config._xml_path, action_size(ac_space), non_limited_idx,
                                            planner_type=planner_type, passive_joint_idx=passive_joint_idx,
                                            ignored_contacts=ignored_contacts, contact_threshold=config.contact_threshold,
                                            goal_bias=goal_bias, is_simplified=is_simplified,
                                            simplified_duration=simplified_duration, range_=range_)

    def isValidState(self, state):
        return self.planner.isValidState(state)

    def get_planner_status(self):
        return self.planner.get_planner_status()

    def plan(self, start, goal, timelimit=None, attempts=15):
        """-> (traj, success, valid, exact); `attempts` is accepted and unused, as in the reference."""
        budget = self._config.timelimit if timelimit is None else timelimit
        traj, _states, valid, exact = self.planner.plan(start, goal, budget)
        solved = bool(valid and exact)
        return (traj[1:] if solved else traj), solved, valid, exact
