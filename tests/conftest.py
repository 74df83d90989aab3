import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


SUPPORTED_ENVS = ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0", "SawyerAssemblyObstacle-v0", "PusherObstacle-v0"]


def sample_states(pi, n, seed, mode="uniform"):
    """Synthetic planner states for an env: (q_active[n,na], qpos_env[1,nq])."""
    from mopa_rl_amd.scene import default_qpos
    rng = np.random.default_rng(seed)
    lo, hi = pi.jnt_minimum, pi.jnt_maximum
    if mode == "uniform":
        qa = rng.uniform(lo, hi, size=(n, len(lo)))
    else:  # near the env's initial pose
        q0 = default_qpos(pi.spec.env, pi.model)[pi.ref_joint_pos_indexes]
        qa = np.clip(q0 + rng.normal(0.0, 0.3, size=(n, len(lo))), lo, hi)
    return qa, default_qpos(pi.spec.env, pi.model)[None, :]


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O
