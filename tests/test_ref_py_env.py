"""Kinematic env.step (K4's checker `orc_env_step`, and on the GPU K4 itself) against the REFERENCE'S OWN env classes:
`SawyerPushObstacleEnv / SawyerLiftObstacleEnv / SawyerAssemblyObstacleEnv.step` (env/base.py:232-247 -> `_step`,
`compute_reward`, `_get_obs`, `_after_step`) were run in the build container over a sim-shaped adapter whose kinematics are
the CPU oracle's and whose `_do_simulation` is the kinematic limit (tools/gen_ref_py_golden.py, tools/refshim.py) on scripted
direct / planner actions; tests/golden/ref_py_env_*.npz holds inputs and outputs.

Joint states and flags must be identical; rewards and observations agree to round-off (numpy's tanh / norm / matmul vs the
kernel's own tanh and fma ordering)."""
import os

import numpy as np
import pytest

ENVS = [("SawyerPushObstacle-v0", "push"), ("SawyerLiftObstacle-v0", "lift"), ("SawyerAssemblyObstacle-v0", "assembly"),
        ("PusherObstacle-v0", "pusher")]       # pusher: env/pusher/pusher_obstacle.py, its PID loop at the kinematic limit
DIST_THRESHOLD = {"pusher": 0.05}              # config/pusher.py:16-20 (the Sawyer envs: 0.06)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _check(G, e, t, qpos, obs, reward, done, success):
    assert np.array_equal(_bits(qpos), _bits(G["qpos_after"][e, t])), (e, t, np.abs(qpos - G["qpos_after"][e, t]).max())
    assert int(done) == G["done"][e, t] and int(success) == G["success"][e, t], (e, t)
    np.testing.assert_allclose(reward, G["reward"][e, t], rtol=1e-12, atol=1e-13, err_msg=str((e, t)))
    np.testing.assert_allclose(obs, G["obs"][e, t], rtol=0, atol=1e-12, err_msg=str((e, t)))


@pytest.mark.parametrize("env,tag", ENVS)
def test_oracle_env_equals_reference_env(env, tag, oracle_mod):
    from mopa_rl_amd.kinematic_env import env_facts
    from mopa_rl_amd.scene import planner_inputs
    G = np.load(os.path.join(GOLD, f"ref_py_env_{tag}.npz"))
    pi = planner_inputs(env)
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    facts = env_facts(env, pi.model)
    E, T = G["action"].shape[:2]
    ref = oracle_mod.OracleEnv(orc, facts, E, ac_scale=pi.spec.ac_scale, max_episode_steps=int(G["max_episode_steps"]),
                               distance_threshold=DIST_THRESHOLD.get(tag, 0.06))
    assert ref.obs_dim == G["obs"].shape[2] and ref.action_dim == G["action"].shape[2]
    ref.set_state(G["qpos0"])
    np.testing.assert_allclose(ref.obs, G["obs0"], rtol=0, atol=1e-12)
    for e in range(E):
        for t in range(int(G["n_steps"][e])):
            if G["fresh_prev"][e, t]:
                ref.has_prev[e] = 0
            ref._call(e, G["action"][e:e + 1].repeat(E, axis=0)[:, t], int(G["is_planner"][e, t]), 1)
            _check(G, e, t, ref.qpos[e], ref.obs[e], ref.reward[e], ref.done[e], ref.success[e])
    assert (G["reward"] != 0).sum() > 10


@pytest.mark.gpu
@pytest.mark.parametrize("env,tag", ENVS)
def test_hip_env_equals_reference_env(env, tag):
    import torch
    from mopa_rl_amd.kinematic_env import make_env
    G = np.load(os.path.join(GOLD, f"ref_py_env_{tag}.npz"))
    E, T = G["action"].shape[:2]
    b = make_env(env, E, max_episode_steps=int(G["max_episode_steps"]))
    b.set_state(torch.tensor(G["qpos0"], device="cuda"))
    np.testing.assert_allclose(b.obs.cpu().numpy(), G["obs0"], rtol=0, atol=1e-12)
    live = np.ones(E, dtype=bool)
    for t in range(T):
        live &= G["n_steps"] > t
        if not live.any():
            break
        fresh = torch.tensor(G["fresh_prev"][:, t].astype(bool), device="cuda")
        b.has_prev[fresh] = 0
        # is_planner differs per env: two masked launches (bit 1 of the flag = sit this call out)
        for pl in (0, 1):
            sel = live & (G["is_planner"][:, t] == pl)
            if sel.any():
                flags = torch.tensor(np.where(sel, 1, 2).astype(np.uint8), device="cuda")
                b._launch(torch.tensor(G["action"][:, t], device="cuda").contiguous(), bool(pl), flags)
        q, o, r, d, s = (x.cpu().numpy() for x in (b.qpos, b.obs, b.reward, b.done, b.success))
        for e in np.where(live)[0]:
            _check(G, e, t, q[e], o[e], r[e], d[e], s[e])
