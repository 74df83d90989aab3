"""K7 against the oracle over many envs: `env.step` rollouts (obs, qpos incl. the object's pose, qvel, flags) bit for bit, all three envs,
the three solver forms (Newton + elliptic cones: the default; Newton + pyramidal; Gauss-Seidel), several seeds; initial states as tests/test_gpu_dyn.py::_ct_states (rest, impact, sliding, arm-object and arm-scene
contacts, joints near their limits).  Needs the oracle: test infrastructure, run on the GPU box with the repo.
   python tools/ct_parity_sweep.py [E] [steps] > profiles/rNN/ct_parity_sweep.txt"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import oracle as O
import test_gpu_dyn as T

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
total = bad = 0
t00 = time.perf_counter()
for env_name in T.ENVS:
    for solver in ("newton", "newton-pyramidal", "pgs"):
        for seed in (31, 32):
            pi, orc, env, ref = T._setup_ct(O, env_name, E, contact_options={"newton": {"solver": "newton"}, "newton-pyramidal": {"solver": "newton", "cone": "pyramidal"}, "pgs": {"solver": "pgs"}}[solver],
                                             max_episode_steps=1 << 20)
            q, v = T._ct_states(env, orc, E, seed=seed)
            env.set_state(torch.tensor(q, device=env.device)); env.qvel.copy_(torch.tensor(v, device=env.device))
            ref.set_state(q); ref.qvel[:] = v
            lag = env.dyn_forward()[0]; env.bias_lag.copy_(lag); ref.bias_lag[:] = lag.cpu().numpy()
            rng = np.random.default_rng(seed)
            mism = its = con = 0
            stats = torch.zeros(E, 4, dtype=torch.int32, device=env.device)
            from mopa_rl_amd import _lib
            _lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, stats.data_ptr()))
            for t in range(steps):
                a = rng.uniform(-1.5, 1.5, size=(E, env.action_dim))
                env.step(torch.tensor(a, device=env.device))
                ref.step(a, nthreads=16)
                s = stats.cpu().numpy()
                con += int(s[:, 0].sum()); its += int(s[:, 1].sum())
                for name, g, r in (("qpos", env.qpos, ref.qpos), ("qvel", env.qvel, ref.qvel), ("obs", env.obs, ref.obs), ("reward", env.reward, ref.reward)):
                    mism += int((bits(g.cpu().numpy()) != bits(r)).sum())
                mism += int((env.done.cpu().numpy() != ref.done).sum())
            total += E * steps; bad += mism
            print(f"{env_name:28s} solver {solver:16s} seed {seed}: {E} envs x {steps} env.steps ({E * steps * env.dyn.nsub} sub-steps, "
                  f"{con / (E * steps * env.dyn.nsub):.2f} contacts and {its / (E * steps * env.dyn.nsub):.2f} solver iterations per sub-step): "
                  f"{mism} mismatching words", flush=True)
            env.close()
print(f"TOTAL {total} env.steps, {bad} mismatching words, {time.perf_counter() - t00:.0f} s")
