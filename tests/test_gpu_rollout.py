"""Batched rollout step (mopa_rl_amd/rollout.py) against fixtures produced by running the reference's own
rollout runner in the build container (tools/gen_ref_py_golden.py): same actions, same RNG streams => every env must end
every agent step in the same state with the same SMDP reward / done / intra_steps / counters."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENV = "SawyerPushObstacle-v0"


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load_state(env, qpos, ep_len):
    """put every env into a recorded state (qpos row + episode length), fresh `_prev_state`, refreshed obs"""
    import torch
    env.set_state(torch.tensor(qpos, device=env.device))
    env.ep_len.copy_(torch.tensor(ep_len, dtype=torch.int32, device=env.device))


def _check_qpos(got, want, pulled_back, msg):
    """Joint states are identical bit for bit -- except in steps whose target went through the invalid-target back-off: the
    reference divides by np.linalg.norm (a BLAS dot, summation order build-dependent), the kernel sums the squares left to
    right, so those targets agree to the last bit or two only."""
    exact = pulled_back == 0
    assert np.array_equal(_bits(got[exact]), _bits(want[exact])), msg
    np.testing.assert_allclose(got[~exact], want[~exact], rtol=0, atol=1e-14, err_msg=msg)


def _make(G, E, env_name=ENV, **kw):
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    timelimit, max_nodes, max_path, seed, max_episode_steps, num_trials = G["params"]
    env = make_env(env_name, E, seed=0, max_episode_steps=int(max_episode_steps))
    env.reset()
    cfg = RolloutConfig.for_env(env_name, timelimit=float(timelimit), max_nodes=int(max_nodes), max_path=int(max_path), seed=int(seed),
                                num_trials=int(num_trials), **kw)
    return env, BatchMoPARollout(env, cfg)


@pytest.mark.parametrize("env_name,tag", [("SawyerPushObstacle-v0", "push"), ("SawyerLiftObstacle-v0", "lift"),
                                          ("SawyerAssemblyObstacle-v0", "assembly"), ("PusherObstacle-v0", "pusher")])
def test_batched_rollout_equals_reference_rollout_runner(env_name, tag):
    """Every agent step of every env against the REFERENCE'S OWN `MoPARolloutRunner.run` (rl/mopa_rollouts.py:70-375), run in
    the build container env by env on the same scripted actions with the reference's SACAgent / PlannerAgent /
    SamplingBasedPlanner / env classes (tests/golden/ref_py_rollout_push.npz, tools/gen_ref_py_golden.py; validity, RRT-Connect
    and FK underneath came from the CPU oracle, the physics from its kinematic limit).  Joint states, done flags,
    intra_steps and the six counters must be identical; rewards and observations agree to round-off (the reference uses
    np.tanh / np.linalg.norm, the kernel its own correctly-rounded-to-1-ulp tanh and a fixed summation order)."""
    import torch
    from mopa_rl_amd.rollout import COUNTERS
    G = np.load(os.path.join(GOLD, f"ref_py_rollout_{tag}.npz"))
    E, T = G["ac"].shape[:2]
    env, ro = _make(G, E, env_name)
    for t in range(T):
        _load_state(env, G["qpos_start"][:, t], G["ep_len_start"][:, t])
        before = {k: ro.counters[k].clone() for k in COUNTERS}
        ro.t = t
        out = ro.agent_step(torch.tensor(G["ac"][:, t], device=env.device))
        _check_qpos(env.qpos.cpu().numpy(), G["qpos_end"][:, t], G["pulled_back"][:, t], f"step {t}: qpos")
        assert np.array_equal(out["done"].cpu().numpy().astype(np.int64), G["done"][:, t]), f"step {t}: done"
        assert np.array_equal(out["intra_steps"].cpu().numpy(), G["intra"][:, t]), f"step {t}: intra_steps"
        got_c = np.stack([(ro.counters[k] - before[k]).cpu().numpy() for k in COUNTERS], axis=1)
        assert np.array_equal(got_c, G["counters"][:, t]), f"step {t}: counters"
        np.testing.assert_allclose(out["rew"].cpu().numpy(), G["rew"][:, t], rtol=1e-12, atol=1e-13, err_msg=f"step {t}: reward")
        ob, ob_want = out["ob"].cpu().numpy(), G["ob"][:, t].copy()
        ob_n, ob_n_want = out["ob_next"].cpu().numpy(), G["ob_next"][:, t].copy()
        if tag == "pusher":
            # `PusherObstacleEnv._reset` draws joint / box velocities (pusher_obstacle.py:51-56) that the obs of an episode reports
            # until the first env.step (a failed plan with an invalid target takes none); the kinematic env has no velocity state
            # (every step ends at rest): those six entries are not compared
            for a in (ob, ob_want, ob_n, ob_n_want):
                a[:, 10:16] = 0.0
        np.testing.assert_allclose(ob, ob_want, rtol=0, atol=1e-12, err_msg=f"step {t}: ob")
        np.testing.assert_allclose(ob_n, ob_n_want, rtol=0, atol=1e-12, err_msg=f"step {t}: ob_next")
    tot = dict(zip(COUNTERS, G["counters"].sum(axis=(0, 1))))
    if tag == "push":
        assert all(v > 0 for v in tot.values()), tot            # every branch of the loop is in the fixture
    assert tot["rl"] > 0 and tot["interpolation"] > 0 and tot["mp_fail"] > 0 and tot["invalid"] > 0, tot
    assert G["done"].sum() > 0 and G["intra"].max() >= 8
    if tag == "pusher":      # BASELINE config 1's env: the unlimited joint0 leaves (-3.14, 3.14) and queries start from beyond the seam
        assert np.abs(G["qpos_start"][:, :, 0]).max() > 3.2 and tot["mp"] > 0


def test_seam_crossing_paths_unwrap_like_the_single_env_planner():
    """Planner rows of a model with an unlimited joint (Pusher: joint0 is SO(2)) through `postprocess_paths` with the seam mask,
    against `SamplingBasedPlanner.plan`'s own un-wrap (the mirror tests/test_ref_py_host.py pins to the reference): queries whose
    wrapped endpoints lie on either side of +-3.14, so that RRT-Connect connects them ACROSS the seam; every un-wrapped row must
    be the mirror's, bit for bit."""
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.agent_planning import JointLimits
    from mopa_rl_amd.batch import BatchPlanner, postprocess_paths
    from mopa_rl_amd.kinematic_env import env_facts
    from mopa_rl_amd.rollout import seam_steps_np, wrap_unlimited
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs("PusherObstacle-v0")
    f = env_facts("PusherObstacle-v0", pi.model)
    scene = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range, seed=5)
    bp = BatchPlanner(scene)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2)
    q0 = np.asarray(pi.model.qpos0, dtype=np.float64)
    cand = np.tile(q0, (4096, 1))
    cand[:, 0] = rng.choice([-1.0, 1.0], 4096) * rng.uniform(2.6, 3.6, 4096)          # un-wrapped: some beyond +-3.14
    cand[:, 1:4] = rng.uniform(-2.5, 2.5, (4096, 3))
    ok = bp.is_valid(torch.tensor(cand[:, :4], device=dev).contiguous(), torch.tensor(cand, device=dev), samples_per_env=1).bool().cpu().numpy()
    cand = cand[ok]
    M = (len(cand) // 2) & ~1
    assert M >= 64
    start, goal = cand[:M], cand[M:2 * M].copy()
    goal[:, 1:4] = start[:, 1:4] + rng.normal(0, 0.2, (M, 3))                          # near-by arm shapes: most queries solve
    okg = bp.is_valid(torch.tensor(goal[:, :4], device=dev).contiguous(), torch.tensor(goal, device=dev), samples_per_env=1).bool().cpu().numpy()
    start, goal = start[okg], goal[okg]
    M = len(start)
    st_t, go_t = torch.tensor(start, device=dev), torch.tensor(goal, device=dev)
    sw, gw = wrap_unlimited(st_t, [0]).contiguous(), wrap_unlimited(go_t, [0]).contiguous()
    assert float(sw[:, 0].abs().max()) <= 3.14
    path, plen, status, _ = bp.plan(sw, gw, max_iters=600, max_nodes=512, max_path=128, seed=5)
    raw = path.cpu().numpy().copy()
    lim = JointLimits(f.qpos_min, f.qpos_max, f.qpos_limited, 0.0, device=dev)
    valid = lambda rows: bp.is_valid(rows[:, :4].contiguous(), rows.contiguous(), samples_per_env=1)
    out, ln, need = postprocess_paths(path, plen, status, st_t.contiguous(), 4, pi.spec.ac_scale, False, lim, valid, seam_mask=1)
    out, ln, plen_h, st_h = out.cpu().numpy(), ln.cpu().numpy(), plen.cpu().numpy(), status.cpu().numpy()
    crossed = 0
    for q in np.where(st_h == 0)[0]:
        P = raw[q, :plen_h[q]]
        want = np.add.accumulate(np.vstack([start[q][None], seam_steps_np(P, [0])]), axis=0)
        crossed += int((np.abs(P[1:, 0] - P[:-1, 0]) > 3.14).any())
        assert ln[q] == plen_h[q] - 1
        assert np.array_equal(_bits(out[q, :ln[q]]), _bits(want[1:])), q
    assert (st_h == 0).sum() >= 32 and crossed >= 8, ((st_h == 0).sum(), crossed)


def test_discrete_action_rollout_equals_reference_runner():
    """--discrete_action (rl/mopa_rollouts.py:86-88,106-111,349): the policy's `ac_type` head routes a step to the planner, and
    direct actions are not rescaled by 1 / omega -- against the reference runner run with config.discrete_action
    (tests/golden/ref_py_rollout_push_discrete.npz)."""
    import torch
    from mopa_rl_amd.rollout import COUNTERS
    G = np.load(os.path.join(GOLD, "ref_py_rollout_push_discrete.npz"))
    E, T = G["ac"].shape[:2]
    env, ro = _make(G, E, ENV, discrete_action=True)
    for t in range(T):
        _load_state(env, G["qpos_start"][:, t], G["ep_len_start"][:, t])
        before = {k: ro.counters[k].clone() for k in COUNTERS}
        ro.t = t
        out = ro.agent_step(torch.tensor(G["ac"][:, t], device=env.device), ac_type=torch.tensor(G["ac_type"][:, t], device=env.device))
        _check_qpos(env.qpos.cpu().numpy(), G["qpos_end"][:, t], G["pulled_back"][:, t], f"step {t}: qpos")
        assert np.array_equal(out["done"].cpu().numpy().astype(np.int64), G["done"][:, t]), f"step {t}: done"
        assert np.array_equal(out["intra_steps"].cpu().numpy(), G["intra"][:, t]), f"step {t}: intra_steps"
        assert np.array_equal(out["ac_type"].cpu().numpy(), G["ac_type"][:, t])
        got_c = np.stack([(ro.counters[k] - before[k]).cpu().numpy() for k in COUNTERS], axis=1)
        assert np.array_equal(got_c, G["counters"][:, t]), f"step {t}: counters"
        np.testing.assert_allclose(out["rew"].cpu().numpy(), G["rew"][:, t], rtol=1e-12, atol=1e-13, err_msg=f"step {t}: reward")
        np.testing.assert_allclose(out["ob_next"].cpu().numpy(), G["ob_next"][:, t], rtol=0, atol=1e-12, err_msg=f"step {t}: ob_next")
    tot = dict(zip(COUNTERS, G["counters"].sum(axis=(0, 1))))
    assert tot["rl"] > 0 and tot["interpolation"] > 0 and tot["mp_fail"] > 0, tot
    # the head, not the magnitude, decided: small actions that went to the planner and large ones executed directly
    n = 7
    big = (np.abs(G["ac"][:, :, :n]) > 0.7).any(axis=2)
    assert (G["ac_type"].astype(bool) & ~big).any() and (~G["ac_type"].astype(bool) & big).any()


def test_ik_action_space_rollout_equals_reference_runner():
    """BASELINE config 5's action space (MoPA + IK, `use_ik_target`): Cartesian displacement + rotation quaternion of the grip site
    -> joint displacement through the batched damped-LS IK (K5, position + orientation target) -> planner / direct decision ->
    env steps, against the reference runner on SawyerAssemblyObstacle (tests/golden/ref_py_rollout_assembly_ik.npz).
    The reference forms the IK's orientation target from a float32 rotation matrix through an eigen-decomposition
    (util/env.py:232-288); the batched form uses the closed-form quaternion of the same float32 matrix, so joint states agree
    to ~1e-7 rather than bit for bit; flags and counters are identical."""
    import torch
    from mopa_rl_amd.rollout import COUNTERS
    env_name = "SawyerAssemblyObstacle-v0"
    G = np.load(os.path.join(GOLD, "ref_py_rollout_assembly_ik.npz"))
    E, T = G["ac"].shape[:2]
    env, ro = _make(G, E, env_name, use_ik_target=True)
    assert ro.ac_dim == 7
    for t in range(T):
        _load_state(env, G["qpos_start"][:, t], G["ep_len_start"][:, t])
        before = {k: ro.counters[k].clone() for k in COUNTERS}
        ro.t = t
        out = ro.agent_step(torch.tensor(G["ac"][:, t], device=env.device), record=True)
        # (a planner step of this action space executes the two waypoints of a zero-length line: nothing for `reuse_data` to
        #  relabel -- its `len(ob_list) > 3` never holds, rl/mopa_rollouts.py:222)
        assert int(out["record"]["n_exec"].max()) <= 2 and ro.reuse_transitions(out, np.random.RandomState(0)) == []
        np.testing.assert_allclose(env.qpos.cpu().numpy(), G["qpos_end"][:, t], rtol=0, atol=2e-6, err_msg=f"step {t}: qpos")
        assert np.array_equal(out["done"].cpu().numpy().astype(np.int64), G["done"][:, t]), f"step {t}: done"
        assert np.array_equal(out["intra_steps"].cpu().numpy(), G["intra"][:, t]), f"step {t}: intra_steps"
        got_c = np.stack([(ro.counters[k] - before[k]).cpu().numpy() for k in COUNTERS], axis=1)
        assert np.array_equal(got_c, G["counters"][:, t]), f"step {t}: counters"
        np.testing.assert_allclose(out["rew"].cpu().numpy(), G["rew"][:, t], rtol=1e-5, atol=1e-6, err_msg=f"step {t}: reward")
        np.testing.assert_allclose(out["ob_next"].cpu().numpy(), G["ob_next"][:, t], rtol=0, atol=1e-5, err_msg=f"step {t}: ob_next")
    tot = dict(zip(COUNTERS, G["counters"].sum(axis=(0, 1))))
    assert tot["rl"] > 0 and tot["interpolation"] > 0, tot


def test_reuse_data_relabelling_equals_reference():
    """`reuse_transitions` against the relabelled sub-trajectory transitions the reference's runner emitted (reuse_data=True,
    rl/mopa_rollouts.py:204-300) on the same steps with the same random draws."""
    import torch
    from mopa_rl_amd.rollout import reuse_transitions
    G = np.load(os.path.join(GOLD, "ref_py_rollout_push_reuse.npz"))
    E, T = G["ac"].shape[:2]
    env, ro = _make(G, E)
    n_checked = 0
    for t in range(T):
        _load_state(env, G["qpos_start"][:, t], G["ep_len_start"][:, t])
        ro.t = t
        out = ro.agent_step(torch.tensor(G["ac"][:, t], device=env.device), record=True)
        _check_qpos(env.qpos.cpu().numpy(), G["qpos_end"][:, t], G["pulled_back"][:, t], f"step {t}: qpos")
        rec = out["record"]
        nexec = rec["n_exec"].cpu().numpy()
        assert np.array_equal(nexec, np.where(out["plan_ok"].cpu().numpy(), out["intra_steps"].cpu().numpy() + 1, 0))
        got = reuse_transitions(out, ro.cfg, 7, lambda e: np.random.RandomState(1000 * e + t))
        sel = G["x_t"] == t
        want_env = G["x_env"][sel]
        assert [g["env"] for g in got] == list(want_env), f"step {t}"
        assert [g["intra_steps"] for g in got] == list(G["x_intra"][sel]) and [g["done"] for g in got] == list(G["x_done"][sel])
        if len(got):
            np.testing.assert_allclose(np.array([g["ac"] for g in got]), G["x_ac"][sel], rtol=0, atol=1e-12, err_msg=f"step {t}: relabelled actions")
            np.testing.assert_allclose([g["rew"] for g in got], G["x_rew"][sel], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(np.array([g["ob"] for g in got]), G["x_ob"][sel], rtol=0, atol=1e-12)
            np.testing.assert_allclose(np.array([g["ob_next"] for g in got]), G["x_ob_next"][sel], rtol=0, atol=1e-12)
        n_checked += len(got)
    assert n_checked == len(G["x_env"]) and n_checked > 30


@pytest.mark.parametrize("env_name", ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0", "SawyerAssemblyObstacle-v0"])
def test_async_planner_gives_every_env_the_same_transitions(env_name):
    """`async_planner`: RRT-Connect queries run on side streams while the other envs go on stepping; the envs waiting for a
    query sit out.  Every env must still go through exactly the transitions of the lock-step run (its plans are keyed by its own
    step count), only later in wall-clock order."""
    import torch
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    E, T = 192, 5
    rng = np.random.default_rng(4)
    n_ac = 8 if env_name == "SawyerLiftObstacle-v0" else 7
    AC = rng.uniform(-1, 1, size=(E, T, n_ac)) * rng.choice([0.6, 0.9, 1.0], size=(E, T, 1))
    AC[: E // 2, 1, 1], AC[: E // 2, 1, 3] = 1.0, -1.0           # blocked straight lines: RRT-Connect queries
    AC[E // 4: 3 * E // 4, 3, 1], AC[E // 4: 3 * E // 4, 3, 3] = 1.0, -1.0
    ACt = torch.tensor(AC, device="cuda")
    runs = {}
    for mode in ("lockstep", "async", "graphs", "pooled"):
        env = make_env(env_name, E, seed=12, max_episode_steps=1000)
        env.reset()
        # (async: a first launch with 60 of the 300 iterations, the continuation of its unsolved queries chained right behind
        #  it on the same stream; graphs: the same with the fixed-shape halves of a call replayed from HIP graphs; pooled: the
        #  unsolved queries wait in a retry pool, their trees parked per env, and continue in a launch of their own)
        ro = BatchMoPARollout(env, RolloutConfig(timelimit=0.15, max_nodes=512, max_path=128, num_trials=10, async_planner=(mode != "lockstep"),
                                                 planner_first_iters=60, planner_min_job=1, use_graphs=(mode == "graphs"),
                                                 planner_chain=0 if mode == "pooled" else 1))
        seq = [[] for _ in range(E)]
        calls = n_sitting = 0
        while min(len(q) for q in seq) < T:
            te = ro.t_env.clamp(max=T - 1)
            ac = ACt[torch.arange(E, device="cuda"), te].contiguous()
            out = ro.agent_step(ac)
            st = out["stepped"].cpu().numpy()
            rows = np.concatenate([out["rew"].cpu().numpy()[:, None], out["done"].cpu().numpy()[:, None].astype(np.float64),
                                   out["intra_steps"].cpu().numpy()[:, None].astype(np.float64), env.qpos.cpu().numpy()[:, :9],
                                   out["ac"].cpu().numpy()], axis=1)
            for e in np.where(st)[0]:
                seq[e].append(rows[e])
            n_sitting += int((~st).sum())
            calls += 1
            assert calls < 200
        runs[mode] = (np.array([np.array(q[:T]) for q in seq]), calls, n_sitting, {k: v.clone() for k, v in ro.counters.items()},
                      getattr(ro, "n_retried", 0))
    a, b, c, d = runs["lockstep"], runs["async"], runs["graphs"], runs["pooled"]
    assert a[2] == 0 and a[1] == T
    assert np.array_equal(_bits(a[0]), _bits(b[0]))
    assert np.array_equal(_bits(a[0]), _bits(c[0])), "graph replay changes an env's transitions"
    assert np.array_equal(_bits(a[0]), _bits(d[0])), "pooled retry launches change an env's transitions"
    assert int(a[3]["mp"].sum()) > 0                                          # RRT-Connect was exercised ...
    if env_name == ENV:
        assert int(a[3]["mp_fail"].sum()) > 0 and b[4] > 0 and d[4] > 0       # ... with both outcomes, and second launches


def test_pusher_rollout_forms_agree_across_the_seam():
    """PusherObstacle-v0 (joint0 unlimited) started next to +-3.14 and driven across it: the lock-step run, the asynchronous
    planner run, and the lock-step run with the torch bookkeeping + host-side path post-processing must give every env the same
    transitions (wrap of the query endpoints, seam rule of the un-wrap: device kernel vs numpy form)."""
    import torch
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    env_name = "PusherObstacle-v0"
    E, T = 192, 5
    rng = np.random.default_rng(6)
    AC = rng.uniform(-1, 1, size=(E, T, 4)) * rng.choice([0.5, 0.9, 1.0], size=(E, T, 1))
    sign = np.where(np.arange(E) % 2 == 0, 1.0, -1.0)
    AC[: E // 2] *= 0.6
    AC[: E // 2, :, 0] = sign[: E // 2, None] * rng.uniform(0.71, 0.85, size=(E // 2, T))
    ACt = torch.tensor(AC, device="cuda")
    # valid start states with joint0 next to the seam on the side its env is driven towards
    probe = make_env(env_name, 8192, seed=1)
    probe.reset()
    q = probe.qpos.clone()
    g = torch.Generator(device=probe.device); g.manual_seed(3)
    q[:, 0] = 3.02 + 0.11 * torch.rand(8192, generator=g, dtype=torch.float64, device=probe.device)
    q[:, 1:4] = (torch.rand(8192, 3, generator=g, dtype=torch.float64, device=probe.device) * 2 - 1) * 2.5
    q[1::2, 0] *= -1.0
    ro0 = BatchMoPARollout(probe, RolloutConfig.for_env(env_name))
    ok = ro0._valid(q).cpu().numpy()
    qn = q.cpu().numpy()
    pos, neg = qn[ok & (qn[:, 0] > 0)], qn[ok & (qn[:, 0] < 0)]
    assert len(pos) >= E // 2 and len(neg) >= E // 2
    start = np.empty((E, qn.shape[1]))
    start[0::2], start[1::2] = pos[: E // 2], neg[: E // 2]
    probe.close()
    runs = {}
    for mode in ("lockstep", "async", "host"):
        env = make_env(env_name, E, seed=12, max_episode_steps=1000)
        env.set_state(torch.tensor(start, device=env.device))
        ro = BatchMoPARollout(env, RolloutConfig.for_env(env_name, timelimit=0.3, max_nodes=512, max_path=128, num_trials=10,
                                                         async_planner=(mode == "async"), planner_first_iters=60, planner_min_job=1,
                                                         device_paths=(mode != "host"), fused=(mode != "host")))
        seq = [[] for _ in range(E)]
        calls = 0
        while min(len(x) for x in seq) < T:
            te = ro.t_env.clamp(max=T - 1)
            out = ro.agent_step(ACt[torch.arange(E, device="cuda"), te].contiguous())
            st = out["stepped"].cpu().numpy()
            rows = np.concatenate([out["rew"].cpu().numpy()[:, None], out["done"].cpu().numpy()[:, None].astype(np.float64),
                                   out["intra_steps"].cpu().numpy()[:, None].astype(np.float64), env.qpos.cpu().numpy()[:, :4],
                                   out["ac"].cpu().numpy()], axis=1)
            for e in np.where(st)[0]:
                seq[e].append(rows[e])
            calls += 1
            assert calls < 200
        runs[mode] = (np.array([np.array(x[:T]) for x in seq]), {k: int(v.sum()) for k, v in ro.counters.items()})
        ro.drain() if hasattr(ro, "drain") else None
        env.close()
    a, b, c = runs["lockstep"], runs["async"], runs["host"]
    assert np.array_equal(_bits(a[0]), _bits(b[0])), "the asynchronous planner changes an env's transitions"
    assert np.array_equal(_bits(a[0]), _bits(c[0])), "host-side post-processing / torch bookkeeping change an env's transitions"
    assert a[1]["mp"] > 0 and a[1]["interpolation"] > 0 and a[1]["mp_fail"] > 0, a[1]
    assert np.abs(a[0][:, :, 3]).max() > 3.3          # joint0 ended up beyond the seam


@pytest.mark.parametrize("env_name,kw", [("SawyerPushObstacle-v0", {}), ("SawyerLiftObstacle-v0", {}),
                                         ("SawyerPushObstacle-v0", {"discrete_action": True}),
                                         ("SawyerPushObstacle-v0", {"invalid_target_handling": False, "ac_space_type": "normal"}),
                                         ("SawyerAssemblyObstacle-v0", {"use_ik_target": True})])
def test_fused_bookkeeping_equals_the_torch_form(env_name, kw):
    """`RolloutConfig.fused` (the call's elementwise bookkeeping as six library kernels, mopa_rollstep.inc) against the torch
    form of the same call: every returned tensor, the env's state and the rollout's persistent state bit for bit, call by
    call, in lock-step runs (the asynchronous form of the fused calls -- envs parked while their query runs -- is what
    test_async_planner_gives_every_env_the_same_transitions compares with these)."""
    import torch
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    E, T = 160, 6
    rng = np.random.default_rng(9)
    for asyn in (False,):
        runs = []
        for fused in (False, True):
            env = make_env(env_name, E, seed=12, max_episode_steps=9)
            env.reset()
            ro = BatchMoPARollout(env, RolloutConfig(timelimit=0.15, max_nodes=512, max_path=128, num_trials=10, async_planner=asyn,
                                                     planner_first_iters=60, planner_min_job=1, fused=fused, **kw))
            r2 = np.random.default_rng(4)
            log = []
            for t in range(T):
                AC = r2.uniform(-1, 1, size=(E, ro.ac_dim + 1)) * r2.choice([0.6, 0.9, 1.0], size=(E, 1))      # (a spare column: ac_stride > ac_dim)
                if t in (1, 3) and not kw.get("use_ik_target"):
                    AC[: E // 2, 1], AC[: E // 2, 3] = 1.0, -1.0                # blocked straight lines: RRT-Connect queries
                ac_type = torch.tensor(r2.integers(0, 2, size=E), device="cuda") if kw.get("discrete_action") else None
                if asyn:
                    ro.drain()            # (pick-ups then happen in the same calls in both runs)
                out = ro.agent_step(torch.tensor(AC, device="cuda"), record=(t % 2 == 1), ac_type=ac_type)
                row = {k: v.cpu().numpy().copy() for k, v in out.items() if k != "record"}
                if "record" in out:
                    row.update({"rec_" + k: v.cpu().numpy().copy() for k, v in out["record"].items()})
                row.update(qpos=env.qpos.cpu().numpy().copy(), has_prev=env.has_prev.cpu().numpy().copy(), ep_len=env.ep_len.cpu().numpy().copy(),
                           busy=ro.busy.cpu().numpy().copy(), pool=ro._pool_mask.cpu().numpy().copy(), wait=ro._wait_since.cpu().numpy().copy(),
                           q_cur=ro._q_cur.cpu().numpy().copy(), q_tgt=ro._q_tgt.cpu().numpy().copy(), pend_ob=ro._pend_ob.cpu().numpy().copy(),
                           pend_ac=ro._pend_ac.cpu().numpy().copy(), pend_type=ro._pend_type.cpu().numpy().copy(), t_env=ro.t_env.cpu().numpy().copy(),
                           t_dev=ro._t_dev.cpu().numpy().copy(), overflow=ro._interp_overflow.cpu().numpy().copy(),
                           **{"c_" + k: v.cpu().numpy().copy() for k, v in ro.counters.items()})
                log.append(row)
            runs.append(log)
            ro.drain()
            env.close()
        for t, (ra, rb) in enumerate(zip(*runs)):
            assert ra.keys() == rb.keys()
            st = ra["stepped"].astype(bool)
            for k in ra:
                a, b = ra[k], rb[k]
                assert a.dtype == b.dtype and a.shape == b.shape, (asyn, t, k, a.dtype, b.dtype)
                if k in ("rew", "done", "intra_steps", "ob_next", "success", "is_planner", "path_len", "plan_ok", "ac_type") or k.startswith("rec_"):
                    a, b = a[st], b[st]                         # rows of envs that sat the call out are not meaningful
                same = np.array_equal(_bits(a), _bits(b)) if a.dtype == np.float64 else np.array_equal(a, b)
                assert same, (asyn, t, k)
        tot = {k[2:]: int(v.sum()) for k, v in runs[1][-1].items() if k.startswith("c_")}
        assert tot["rl"] > 0 and tot["interpolation"] > 0, tot
        if not kw.get("use_ik_target"):
            assert tot["mp"] + tot["mp_fail"] > 0, tot
        if asyn and not kw.get("use_ik_target"):
            assert any(r["busy"].any() for r in runs[1])


def test_pullback_kernel_equals_host_form(oracle_mod):
    """`mopa_pullback_batch` (one launch) against `handle_invalid_target_batch` (torch ops + one validity launch per
    trial, itself the batched form of rl/mopa_rollouts.py:133-143): same targets, trial counts and verdicts, bit for bit."""
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.agent_planning import handle_invalid_target_batch
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import default_qpos, planner_inputs
    pi = planner_inputs(ENV)
    sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
    bp = BatchPlanner(sc)
    E = 300
    rng = np.random.default_rng(2)
    q0 = default_qpos(ENV, pi.model)
    cur = np.repeat(q0[None], E, axis=0)
    cur[:, :7] += rng.normal(0, 0.05, size=(E, 7))
    tgt = cur.copy()
    tgt[:, :7] += rng.uniform(-1.0, 1.0, size=(E, 7)) * rng.choice([0.3, 1.0, 2.0], size=(E, 1))
    tgt[:, :7] = np.clip(tgt[:, :7], pi.jnt_minimum, pi.jnt_maximum)
    tgt[0] = cur[0]                                   # degenerate: target == current state
    c, t = torch.tensor(cur, device="cuda"), torch.tensor(tgt, device="cuda")
    import os
    for num_trials in (100, 3, 0):
        want_t, want_n, want_v = handle_invalid_target_batch(bp, c, t, 0.02, num_trials)
        # both device forms: one wave per env walking its own trials / all candidate rows of all envs through K1 at once
        for form in ("wave", "batch"):
            os.environ["MOPA_PULLBACK"] = form
            try:
                got_t, got_n, got_v = bp.pullback(c, t, 0.02, num_trials)
            finally:
                del os.environ["MOPA_PULLBACK"]
            assert np.array_equal(got_v.cpu().numpy().astype(bool), want_v.cpu().numpy()), form
            assert np.array_equal(got_n.cpu().numpy().astype(np.int64), want_n.cpu().numpy()), form
            assert np.array_equal(_bits(got_t.cpu().numpy()), _bits(want_t.cpu().numpy())), form
    want_t, want_n, want_v = handle_invalid_target_batch(bp, c, t, 0.02, 3)
    n = want_n.cpu().numpy()
    assert (n > 0).sum() > 30 and (~want_v.cpu().numpy()).sum() >= 0 and n.max() == 3
    # A batch whose worst case is large (E x 100 candidate rows) but whose count, known on the device only, is small: the candidate rows
    # then go through the wave-per-state kernel (both validity launches are enqueued, each reads the count, one leaves at once)
    inv = np.where(n > 0)[0][:12]
    tgt2 = cur.copy()
    tgt2[inv] = tgt[inv]
    t2 = torch.tensor(tgt2, device="cuda")
    want_t, want_n, want_v = handle_invalid_target_batch(bp, c, t2, 0.02, 100)
    os.environ["MOPA_PULLBACK"] = "batch"
    try:
        got_t, got_n, got_v = bp.pullback(c, t2, 0.02, 100)
    finally:
        del os.environ["MOPA_PULLBACK"]
    assert np.array_equal(got_v.cpu().numpy().astype(bool), want_v.cpu().numpy())
    assert np.array_equal(got_n.cpu().numpy().astype(np.int64), want_n.cpu().numpy())
    assert np.array_equal(_bits(got_t.cpu().numpy()), _bits(want_t.cpu().numpy()))
    assert len(inv) <= int((want_n.cpu().numpy() > 0).sum()) < 90        # (x 100 rows: below the 9216-state switch to the lane-per-state kernel)


@pytest.mark.parametrize("env_name", ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0"])
def test_device_path_postprocessing_equals_host_form(env_name):
    """Planner rows -> trajectories (un-wrap by successive differences, densification of long steps with validated
    interior states) as the device launches `mopa_paths_*` against the array-operation form on the host, on the results of
    real RRT-Connect queries (successes, sentinel rows, paths that need densification): same rows, lengths and flags, bit
    for bit; queries the device form hands to the fallback planners come back through the host form."""
    import torch
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    E = 192
    res = {}
    for dev_paths in (False, True):
        env = make_env(env_name, E, seed=3)
        env.reset()
        # (a long planner range: most path steps exceed ac_scale in some joint and are densified)
        ro = BatchMoPARollout(env, RolloutConfig(device_paths=dev_paths, timelimit=0.1, range=0.5))
        g = torch.Generator(device=env.device)
        g.manual_seed(4)
        cur = ro.clip_qpos(env.qpos.clone())
        tgt = cur.clone()
        tgt[:, :ro.n] += (torch.rand(E, ro.n, generator=g, dtype=torch.float64, device=env.device) * 2 - 1) * 0.45
        tgt = ro.limits.clip_target(tgt)
        tgt, _, tv = ro.bp.pullback(cur.contiguous(), tgt.contiguous(), 0.02, 100)
        ids = torch.arange(E, device=env.device)
        job = ro._rrt_launch(cur.contiguous(), tgt.contiguous(), ids)
        tr, ln, s, v, e = ro._rrt_finish(job)
        conv = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
        res[dev_paths] = tuple(conv(x) for x in (tr, ln, s, v, e)) + (conv(job["plen"]),)
    (tr_h, ln_h, s_h, v_h, e_h, pl), (tr_d, ln_d, s_d, v_d, e_d, _) = res[False], res[True]
    assert np.array_equal(ln_h, ln_d) and np.array_equal(s_h, s_d) and np.array_equal(v_h, v_d) and np.array_equal(e_h, e_d)
    assert s_h.sum() > E // 3 and ((~s_h).sum() > 0 or env_name != "SawyerPushObstacle-v0")     # Push: sentinel rows too
    assert (ln_h[s_h] > pl[s_h] - 1).sum() > 20         # densification happened: more rows than planner waypoints
    for q in range(E):
        assert np.array_equal(_bits(tr_h[q, :ln_h[q]]), _bits(tr_d[q, :ln_d[q]])), q


def test_path_postprocessing_edge_cases():
    """`postprocess_paths`: a batch of sentinel rows only (no path at all), one-row paths (start == goal: nothing to execute but
    the waypoint count is 0) and the argument checks of the C entry points."""
    import ctypes as C
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.batch import BatchPlanner, postprocess_paths
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    env = make_env(ENV, 8, seed=1)
    env.reset()
    ro = BatchMoPARollout(env, RolloutConfig())
    M, P, nq = 5, 16, ro.nq
    dev = env.device
    cur = env.qpos[:M].clone().contiguous()
    path = torch.zeros(M, P, nq, dtype=torch.float64, device=dev)
    path[:, 0] = cur
    plen = torch.tensor([1, 1, 0, 0, 1], dtype=torch.int32, device=dev)
    status = torch.tensor([0, _lib.PLAN_NO_EXACT, _lib.PLAN_INVALID_GOAL, _lib.PLAN_NO_EXACT, 0], dtype=torch.int32, device=dev)
    out, ln, need = postprocess_paths(path, plen, status, cur, ro.n, 0.05, True, ro.limits, ro._valid)
    assert ln.tolist() == [0, 0, 0, 0, 0] and not bool(need.any()) and out.shape[0] == M
    # a two-row path with one long step: start -> start + 0.13 in joint 0  =>  int(0.13 / 0.04) = 3 interior states + the waypoint
    path[0, 1] = cur[0]
    path[0, 1, 0] += 0.13
    plen[0] = 2
    out, ln, need = postprocess_paths(path.clone(), plen, status, cur, ro.n, 0.05, True, ro.limits, lambda q: torch.ones(len(q), dtype=torch.uint8, device=dev))
    assert ln.tolist() == [4, 0, 0, 0, 0]
    got = out[0, :4, 0].cpu().numpy() - float(cur[0, 0])
    assert np.allclose(got, [0.13 / 3.25 * k for k in (1, 2, 3)] + [0.13], atol=1e-12)
    # ... and flagged for the fallback planners when an interior state is invalid
    out, ln, need = postprocess_paths(path.clone(), plen, status, cur, ro.n, 0.05, True, ro.limits, lambda q: torch.zeros(len(q), dtype=torch.uint8, device=dev))
    assert need.tolist() == [True, False, False, False, False]
    L = _lib.lib()
    assert L.mopa_paths_unwrap_batch(0, 1, 65, 7, None, 4, None, None, None, 0.05, 1, None, None, None, None, None, None, None, None) == 1   # MOPA_ERR_INVALID_ARG
    assert L.mopa_paths_unwrap_batch(0, 0, 36, 7, None, 4, None, None, None, 0.05, 1, None, None, None, None, None, None, None, None) == 0
    assert L.mopa_interpolate_batch(ro.scene._h, 4, 7, 0, None, None, 0.05, None, None, None, None, None) == 1   # MOPA_ERR_INVALID_ARG


def test_sharded_rollout_equals_the_unsharded_one():
    """"results are independent of G".  A rank's rollout draws its planner sample streams by the GLOBAL env id
    (`RolloutConfig.env_id_base` = rank * E_g, `env_id_total` = world * E_g; reference rank set-up: rl/main.py:24-35), so the
    rows a shard produces are bit for bit the rows the unsharded rollout produces for the same envs."""
    import torch
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    E, T, G = 128, 4, 2
    rng = np.random.default_rng(7)
    AC = rng.uniform(-1, 1, size=(E, T, 7)) * rng.choice([0.6, 0.9, 1.0], size=(E, T, 1))
    AC[:, 1, 1], AC[:, 1, 3] = 1.0, -1.0           # blocked straight lines: every env plans with RRT-Connect at step 1
    base = make_env(ENV, E, seed=12, max_episode_steps=1000)
    base.reset()
    q0 = base.qpos.clone()

    def run(rows, id_base, total):
        env = make_env(ENV, len(rows), seed=99, max_episode_steps=1000)
        env.set_state(q0[rows].clone())
        ro = BatchMoPARollout(env, RolloutConfig(timelimit=0.15, max_nodes=512, max_path=128, num_trials=10, env_id_base=id_base, env_id_total=total))
        out_rows = []
        for t in range(T):
            out = ro.agent_step(torch.tensor(AC[rows, t], device="cuda").contiguous())
            out_rows.append(np.concatenate([out["rew"].cpu().numpy()[:, None], out["done"].cpu().numpy()[:, None].astype(np.float64),
                                            out["intra_steps"].cpu().numpy()[:, None].astype(np.float64), env.qpos.cpu().numpy(),
                                            out["ob_next"].cpu().numpy()], axis=1))
        return np.stack(out_rows, axis=1), int(ro.counters["mp"].sum()) + int(ro.counters["approximate"].sum())

    full, n_mp = run(np.arange(E), 0, E)
    assert n_mp > 8           # RRT-Connect queries (solved or budget-exhausted: the sample streams under test) were drawn
    for g in range(G):
        rows = np.arange(g * E // G, (g + 1) * E // G)
        part, _ = run(rows, g * E // G, E)
        assert np.array_equal(_bits(part), _bits(full[rows])), f"shard {g} of {G} differs from its rows of the unsharded rollout"
    # ... and the ids matter: the second shard with rank-local ids (base 0) draws other streams
    wrong, _ = run(np.arange(E // G, E), 0, E // G)
    assert not np.array_equal(_bits(wrong), _bits(full[E // G:]))


def test_pool_pick_kernel_equals_nonzero_and_gathers():
    """round 6: mopa_rollout_pool_pick -- the planner's pick-up of the waiting envs in one launch -- against the torch operations it replaces
    (nonzero, gathers, index_put), incl. the two ways it leaves everything untouched (fewer than min_n set, more than cap)"""
    import torch
    from mopa_rl_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    for E, nq, p, min_n, cap in [(4096, 36, 0.05, 1, 2048), (4099, 34, 0.3, 64, 2048), (70, 16, 0.5, 1, 64), (4096, 36, 0.9, 1, 2048), (4096, 36, 0.01, 512, 2048), (64, 7, 0.0, 1, 16)]:
        mask = torch.rand(E, generator=g, device=dev) < p
        q_cur = torch.rand(E, nq, generator=g, dtype=torch.float64, device=dev)
        q_tgt = torch.rand(E, nq, generator=g, dtype=torch.float64, device=dev)
        t_env = torch.randint(0, 1000, (E,), generator=g, device=dev)
        want = torch.nonzero(mask).flatten()
        m2 = mask.clone()
        ids = torch.full((cap,), -1, dtype=torch.int64, device=dev)
        cur = torch.zeros(cap, nq, dtype=torch.float64, device=dev); tgt = torch.zeros_like(cur)
        steps = torch.zeros(cap, dtype=torch.int64, device=dev); seeds = torch.zeros_like(steps)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().mopa_rollout_pool_pick(E, nq, min_n, cap, m2.data_ptr(), q_cur.data_ptr(), q_tgt.data_ptr(), t_env.data_ptr(), 1234,
                                                     ids.data_ptr(), cur.data_ptr(), tgt.data_ptr(), steps.data_ptr(), seeds.data_ptr(), cnt.data_ptr(), None))
        torch.cuda.synchronize()
        n = int(cnt.item())
        assert n == len(want)
        if min_n <= n <= cap:
            assert torch.equal(ids[:n], want) and not bool(m2.any())
            assert torch.equal(cur[:n], q_cur[want]) and torch.equal(tgt[:n], q_tgt[want])
            assert torch.equal(steps[:n], t_env[want]) and torch.equal(seeds[:n], t_env[want] + 1234)
        else:
            assert torch.equal(m2, mask) and bool((ids == -1).all())


@pytest.mark.parametrize("env_name,tag", [("SawyerPushObstacle-v0", "push"), ("SawyerLiftObstacle-v0", "lift")])
def test_run_episode_equals_reference_evaluation_loop(env_name, tag):
    """`BatchMoPARollout.run_episode` against the REFERENCE'S OWN `MoPARolloutRunner.run_episode` (rl/mopa_rollouts.py:401-678), run
    in the build container env by env on the same scripted actions from the same start states (tests/golden/ref_py_episode_*.npz,
    tools/gen_ref_py_golden.py `episode`): whole episodes, free-running -- nothing is re-loaded between agent steps.  Agent steps per
    episode, episode length, the six counters, done flags and success must be identical; per-step rewards (plain sums over a path's
    waypoints, not the SMDP return) and the episode reward agree to round-off; the final joint state is identical bit for bit in
    episodes without an invalid-target back-off and to 1e-12 in the others (see _check_qpos)."""
    import torch
    from mopa_rl_amd.rollout import COUNTERS
    G = np.load(os.path.join(GOLD, f"ref_py_episode_{tag}.npz"))
    E, T = G["ac"].shape[:2]
    env, ro = _make(G, E, env_name)
    _load_state(env, G["qpos_start"][:, 0], np.zeros(E, dtype=np.int64))
    AC = torch.tensor(G["ac"], device=env.device)
    calls = {"t": 0}

    def policy(ob, is_train=True, random_exploration=False):
        t = calls["t"]
        calls["t"] += 1
        assert is_train and not random_exploration
        return AC[:, t] if t < T else torch.zeros_like(AC[:, 0])

    gamma = ro.cfg.discount_factor
    rollout, info = ro.run_episode(policy, reset=False)
    assert ro.cfg.discount_factor == gamma                      # (the loop sums rewards undiscounted and restores the SMDP discount)
    n = G["n_steps"]
    assert np.array_equal(rollout["n_steps"].cpu().numpy(), n), "agent steps per episode"
    assert calls["t"] == int(n.max()) and rollout["valid"].shape[0] == int(n.max())
    assert np.array_equal(info["len"].cpu().numpy(), G["ep_len"]), "episode length"
    assert np.array_equal(np.stack([info[k].cpu().numpy() for k in COUNTERS], axis=1), G["counters"]), "counters"
    assert np.array_equal(info["success"].cpu().numpy().astype(np.int64), G["ep_success"])
    valid = rollout["valid"].cpu().numpy().T                     # [E, Tn]
    Tn = valid.shape[1]
    assert np.array_equal(valid, np.arange(Tn)[None, :] < n[:, None])
    assert np.array_equal(rollout["done"].cpu().numpy().T.astype(np.int64)[valid], G["done"][:, :Tn][valid]), "done flags"
    np.testing.assert_allclose(rollout["rew"].cpu().numpy().T[valid], G["rew"][:, :Tn][valid], rtol=1e-12, atol=1e-13, err_msg="per-step rewards")
    np.testing.assert_allclose(info["rew"].cpu().numpy(), G["ep_rew"], rtol=1e-12, atol=1e-12, err_msg="episode reward")
    pulled = G["pulled_back"].sum(1)
    _check_qpos(rollout["qpos_final"].cpu().numpy(), G["qpos_final"], pulled, "final joint state")
    # the obs each agent step started from, and the final obs (`rollout.add({"ob": ll_ob})`, :661)
    ob = rollout["ob"].cpu().numpy()                             # [Tn + 1, E, obs_dim]
    np.testing.assert_allclose(ob[:Tn].transpose(1, 0, 2)[valid], G["ob"][:, :Tn][valid], rtol=1e-12, atol=1e-12, err_msg="obs before each step")
    np.testing.assert_allclose(ob[n, np.arange(E)], G["ob_final"], rtol=1e-12, atol=1e-12, err_msg="final obs")
    env.close(); ro.close()
