"""mopa_rl_amd -- MI355X-native state-validity / motion-planning hot path for
MoPA-RL-style training (drop-in for the reference's PlannerAgent.plan() /
isValidState() surface)."""
__version__ = "0.1.0"
