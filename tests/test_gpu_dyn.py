"""GPU parity of K6 (`k_env_dyn`, the servo dynamics inside env.step; SURVEY.md 8 f4b stage A) against
oracle/mopa_oracle_dyn.inc: qfrc_bias, the joint-space inertia, every sub-step's (qpos, qvel, lagged bias) and whole
env.step rollouts (obs with joint velocities, reward, flags, carried state) must be equal BIT FOR BIT."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENVS = ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0", "SawyerAssemblyObstacle-v0"]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "the gpu-marked tests need a HIP device"
    return torch


def _setup(oracle_mod, env_name, E, **kw):
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env_name)
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    env = make_env(env_name, E, dynamics=True, **kw)
    ref = oracle_mod.OracleEnv(orc, env.facts, E, ac_scale=env.ac_scale, dyn=env.dyn, **kw)
    return pi, env, ref


def _states(env, E, seed, spread=1.0):
    """qpos rows with the dynamic dofs anywhere in their range (some ON a limit), velocities of either sign."""
    d = env.dyn
    rng = np.random.default_rng(seed)
    q = np.tile(env.init_qpos_row, (E, 1))
    lo = np.where(d.limited == 1, d.lo, -3.0)
    hi = np.where(d.limited == 1, d.hi, 3.0)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    q[:, d.qadr] = mid + spread * half * rng.uniform(-1, 1, size=(E, d.nd))
    edge = rng.random((E, d.nd)) < 0.05
    q[:, d.qadr] = np.where(edge, np.where(rng.random((E, d.nd)) < 0.5, lo, hi), q[:, d.qadr])
    v = rng.normal(0, 1.0, size=(E, d.nd)) * np.where(d.jtype == 3, 1.0, 0.02)
    return q, v


@pytest.mark.parametrize("env_name", ENVS)
def test_forward_bias_and_inertia_bit_exact(oracle_mod, torch_mod, env_name):
    torch = torch_mod
    E = 200
    pi, env, ref = _setup(oracle_mod, env_name, E)
    q, v = _states(env, E, seed=1)
    env.set_state(torch.tensor(q, device=env.device))
    env.qvel.copy_(torch.tensor(v, device=env.device))
    bias, M = env.dyn_forward(want_M=True)
    bias, M = bias.cpu().numpy(), M.cpu().numpy()
    nd = env.dyn.nd
    il = np.tril_indices(nd)
    for e in range(E):
        ob, oM = ref.dyn.forward(q[e], v[e])
        assert np.array_equal(_bits(bias[e]), _bits(ob)), f"env {e}: qfrc_bias"
        assert np.array_equal(_bits(M[e]), _bits(oM[il])), f"env {e}: M"


@pytest.mark.parametrize("env_name", ENVS)
@pytest.mark.parametrize("n", [1, 7, 75])
def test_substeps_bit_exact(oracle_mod, torch_mod, env_name, n):
    torch = torch_mod
    E = 130
    pi, env, ref = _setup(oracle_mod, env_name, E)
    d = env.dyn
    q, v = _states(env, E, seed=10 + n)
    rng = np.random.default_rng(n)
    ctrl = q[:, d.qadr] + rng.uniform(-0.3, 0.3, size=(E, d.nd)) * np.where(d.jtype == 3, 1.0, 0.05)
    env.set_state(torch.tensor(q, device=env.device))
    env.qvel.copy_(torch.tensor(v, device=env.device))
    lag0 = env.dyn_forward()[0]
    env.bias_lag.copy_(lag0)
    env.dyn_substeps(torch.tensor(ctrl, device=env.device), n)
    gq, gv, gl = env.qpos.cpu().numpy(), env.qvel.cpu().numpy(), env.bias_lag.cpu().numpy()
    lag0 = lag0.cpu().numpy()
    hit = 0
    for e in range(E):
        oq, ov, ol = ref.dyn.step(q[e], v[e], lag0[e], ctrl[e], n)
        assert np.array_equal(_bits(gq[e]), _bits(oq)), f"env {e}: qpos after {n} sub-steps"
        assert np.array_equal(_bits(gv[e]), _bits(ov)), f"env {e}: qvel"
        assert np.array_equal(_bits(gl[e]), _bits(ol)), f"env {e}: lagged bias"
        hit += int(np.any((oq[d.qadr] == d.lo) | (oq[d.qadr] == d.hi)))
    assert hit > 0      # the joint stops were exercised


@pytest.mark.parametrize("env_name", ENVS)
@pytest.mark.parametrize("E", [5, 130])
def test_env_step_rollout_bit_identical_to_oracle(oracle_mod, torch_mod, env_name, E):
    torch = torch_mod
    pi, env, ref = _setup(oracle_mod, env_name, E, max_episode_steps=5)
    q, _ = _states(env, E, seed=E, spread=0.5)
    env.set_state(torch.tensor(q, device=env.device))
    ref.set_state(q)

    def compare(what):
        assert np.array_equal(_bits(env.qpos.cpu().numpy()), _bits(ref.qpos)), f"{what}: qpos"
        assert np.array_equal(_bits(env.qvel.cpu().numpy()), _bits(ref.qvel)), f"{what}: qvel"
        assert np.array_equal(_bits(env.bias_lag.cpu().numpy()), _bits(ref.bias_lag)), f"{what}: bias_lag"
        assert np.array_equal(_bits(env.obs.cpu().numpy()), _bits(ref.obs)), f"{what}: obs"
        assert np.array_equal(_bits(env.prev_state.cpu().numpy()), _bits(ref.prev_state)), f"{what}: prev_state"
        assert np.array_equal(env.ep_len.cpu().numpy(), ref.ep_len) and np.array_equal(env.has_prev.cpu().numpy(), ref.has_prev), what

    compare("after set_state")
    rng = np.random.default_rng(7 + E)
    n_done = 0
    for t in range(6):
        is_planner = bool(t % 3 == 1)
        a = rng.uniform(-0.08, 0.08, size=(E, env.action_dim)) if is_planner else rng.uniform(-1.5, 1.5, size=(E, env.action_dim))
        if env.action_dim > 7:
            a[:, 7] = rng.uniform(-0.01, 0.01, size=E)
        obs, rew, done, info = env.step(torch.tensor(a, device=env.device), is_planner=is_planner)
        ref.step(a, is_planner=is_planner)
        compare(f"step {t}")
        assert np.array_equal(_bits(rew.cpu().numpy()), _bits(ref.reward)), f"step {t}: reward"
        assert np.array_equal(done.cpu().numpy(), ref.done) and np.array_equal(info["success"].cpu().numpy(), ref.success)
        assert np.abs(obs[:, 7:14].cpu().numpy()).max() > 1e-3          # the obs carries joint velocities now
        n_done += int(ref.done.sum())
    assert n_done == E          # every env hit max_episode_steps once


def test_move_mask_and_partial_reset(oracle_mod, torch_mod):
    """bit 1: env sits out; bit 0 clear: command recorded, no physics; reset(mask) puts only those envs at rest."""
    torch = torch_mod
    E = 70
    pi, env, ref = _setup(oracle_mod, "SawyerPushObstacle-v0", E)
    q, _ = _states(env, E, seed=3, spread=0.4)
    env.set_state(torch.tensor(q, device=env.device))
    ref.set_state(q)
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, size=(E, 7))
    env.step(torch.tensor(a, device=env.device))
    ref.step(a)
    mm = rng.integers(0, 4, size=E).astype(np.uint8)
    a = rng.uniform(-1, 1, size=(E, 7))
    env._launch(torch.tensor(a, device=env.device), False, torch.tensor(mm, device=env.device))
    ref.step(a, move_mask=mm)
    assert np.array_equal(_bits(env.qpos.cpu().numpy()), _bits(ref.qpos))
    assert np.array_equal(_bits(env.qvel.cpu().numpy()), _bits(ref.qvel))
    assert np.array_equal(_bits(env.obs.cpu().numpy()), _bits(ref.obs))
    assert np.array_equal(env.ep_len.cpu().numpy(), ref.ep_len)
    # partial reset
    mask = torch.tensor(rng.random(E) < 0.5, device=env.device)
    v_before = env.qvel.clone()
    env.reset(mask)
    mk = mask.cpu().numpy()
    assert np.all(env.qvel.cpu().numpy()[mk] == 0.0)
    assert np.array_equal(_bits(env.qvel.cpu().numpy()[~mk]), _bits(v_before.cpu().numpy()[~mk]))
    lag = env.bias_lag.cpu().numpy()
    qn = env.qpos.cpu().numpy()
    for e in np.where(mk)[0][:10]:
        assert np.array_equal(_bits(lag[e]), _bits(ref.dyn.forward(qn[e], np.zeros(env.dyn.nd), want_M=False)[0]))


# ---- stage B: the Push cube as a free body with penalty contacts ----------------------------------------------------
def _setup_obj(oracle_mod, E, **kw):
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.scene import planner_inputs
    env_name = "SawyerPushObstacle-v0"
    pi = planner_inputs(env_name)
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    env = make_env(env_name, E, dynamics=True, contacts="penalty", **kw)
    ref = oracle_mod.OracleEnv(orc, env.facts, E, ac_scale=env.ac_scale, dyn=env.dyn, obj=env.obj, **kw)
    return pi, orc, env, ref


def _obj_states(env, orc, E, seed):
    """arm near its initial pose, the cube scattered around the hand (inside, touching, clear of it), on / in / above the
    table, with random orientation and velocity -- every branch of the contact code sees states"""
    from mopa_rl_amd.mjcf import _quat_to_mat
    d, o, f = env.dyn, env.obj, env.facts
    rng = np.random.default_rng(seed)
    q = np.tile(env.init_qpos_row, (E, 1))
    q[:, d.qadr[:7]] += rng.normal(0, 0.25, size=(E, 7))
    q[:, d.qadr[:7]] = np.clip(q[:, d.qadr[:7]], d.lo[:7], d.hi[:7])
    v = np.zeros((E, d.nd + 6))
    v[:, :7] = rng.normal(0, 0.5, size=(E, 7))
    for e in range(E):
        xp, xq = orc.fk_bodies(q[e])
        b = int(f.frame_body[0])
        hand = xp[b] + _quat_to_mat(xq[b]) @ f.frame_off[0]
        mode = e % 4
        if mode == 0:
            c = hand + rng.normal(0, 0.03, 3)                     # in / at the hand
        elif mode == 1:
            c = np.array([0.92, 0.0, 0.8597]) + rng.normal(0, [0.05, 0.05, 0.002])     # resting height on the table
        elif mode == 2:
            c = hand + rng.normal(0, 0.08, 3)
        else:
            c = np.array([rng.uniform(0.6, 1.3), rng.uniform(-0.5, 0.5), rng.uniform(0.0, 1.0)])      # anywhere: legs, ground, air
        qq = rng.normal(size=4)
        q[e, o.qadr:o.qadr + 3] = c
        q[e, o.qadr + 3:o.qadr + 7] = qq / np.linalg.norm(qq)
        v[e, d.nd:d.nd + 3] = rng.normal(0, 0.2, 3)
        v[e, d.nd + 3:] = rng.normal(0, 2.0, 3)
    return q, v


@pytest.mark.parametrize("n", [1, 5, 75])
def test_object_substeps_bit_exact(oracle_mod, torch_mod, n):
    torch = torch_mod
    E = 256
    pi, orc, env, ref = _setup_obj(oracle_mod, E)
    d = env.dyn
    q, v = _obj_states(env, orc, E, seed=n)
    rng = np.random.default_rng(100 + n)
    ctrl = q[:, d.qadr] + rng.uniform(-0.2, 0.2, size=(E, d.nd)) * np.where(d.jtype == 3, 1.0, 0.0)
    env.set_state(torch.tensor(q, device=env.device))
    env.qvel.copy_(torch.tensor(v, device=env.device))
    lag0 = env.dyn_forward()[0]
    env.bias_lag.copy_(lag0)
    env.dyn_substeps(torch.tensor(ctrl, device=env.device), n)
    gq, gv = env.qpos.cpu().numpy(), env.qvel.cpu().numpy()
    lag0 = lag0.cpu().numpy()
    moved = 0
    for e in range(E):
        oq, ov, _ = ref.dyn.step(q[e], v[e], lag0[e], ctrl[e], n)
        assert np.array_equal(_bits(gq[e]), _bits(oq)), f"env {e}: qpos (object pose included) after {n} sub-steps"
        assert np.array_equal(_bits(gv[e]), _bits(ov)), f"env {e}: qvel (object velocity included)"
        free_fall = v[e, d.nd:d.nd + 3] + n * d.timestep * d.gravity
        moved += int(np.abs(ov[d.nd:d.nd + 3] - free_fall).max() > 1e-6)       # a contact force acted
    assert moved > E // 8


def test_object_env_rollout_bit_identical_to_oracle(oracle_mod, torch_mod):
    torch = torch_mod
    E = 64
    pi, orc, env, ref = _setup_obj(oracle_mod, E, max_episode_steps=6)
    q, v = _obj_states(env, orc, E, seed=11)
    env.set_state(torch.tensor(q, device=env.device))
    ref.set_state(q)
    rng = np.random.default_rng(3)
    for t in range(6):
        a = rng.uniform(-1.5, 1.5, size=(E, 7))
        obs, rew, done, info = env.step(torch.tensor(a, device=env.device))
        ref.step(a)
        assert np.array_equal(_bits(env.qpos.cpu().numpy()), _bits(ref.qpos)), f"step {t}: qpos"
        assert np.array_equal(_bits(env.qvel.cpu().numpy()), _bits(ref.qvel)), f"step {t}: qvel"
        assert np.array_equal(_bits(obs.cpu().numpy()), _bits(ref.obs)), f"step {t}: obs"
        assert np.array_equal(_bits(rew.cpu().numpy()), _bits(ref.reward)), f"step {t}: reward"
        assert np.array_equal(done.cpu().numpy(), ref.done)
    # the cube moved in the obs (cube_pos slice of the Push layout: 21 + 3 .. + 6)
    assert np.abs(ref.obs[:, 24:27] - q[:, env.obj.qadr:env.obj.qadr + 3]).max() > 1e-3


@pytest.mark.parametrize("contacts", [False, "penalty"])
def test_workgroup_and_single_wave_forms_agree(oracle_mod, torch_mod, contacts, monkeypatch):
    """K6 has two mappings -- four waves sharing the 64 envs of a workgroup (default) and one wave per 64 envs
    (MOPA_DYN_KERNEL=wave): same arithmetic per quantity, so the same bits, for ragged batch sizes too."""
    torch = torch_mod
    from mopa_rl_amd.kinematic_env import make_env
    E = 200                # not a multiple of 64: the last workgroup walks with idle lanes
    outs = {}
    for form in ("block", "wave"):
        monkeypatch.setenv("MOPA_DYN_KERNEL", form)
        env = make_env("SawyerPushObstacle-v0", E, dynamics=True, contacts=contacts, seed=5)
        env.reset()
        rng = np.random.default_rng(9)
        mm = torch.tensor(rng.integers(0, 4, size=E).astype(np.uint8), device=env.device)
        for t in range(3):
            a = torch.tensor(rng.uniform(-1.5, 1.5, size=(E, 7)), device=env.device)
            if t == 1:
                env._launch(a, True, mm)
            else:
                env.step(a, is_planner=bool(t))
        bias, M = env.dyn_forward(want_M=True)
        outs[form] = [x.cpu().numpy().copy() for x in (env.qpos, env.qvel, env.bias_lag, env.obs, env.prev_state, bias, M)]
        env.close()
    for a, b in zip(outs["block"], outs["wave"]):
        assert np.array_equal(_bits(a), _bits(b))


def test_waypoint_execution_with_dynamics_equals_stepwise_oracle(oracle_mod, torch_mod):
    """`exec_trajectories` on a dynamics env: every waypoint is a full env.step through the physics; the SMDP return, done,
    intra_steps and the final state must equal the oracle env stepped waypoint by waypoint with the runner's rules
    (rl/mopa_rollouts.py:152-199: action = form_action(waypoint), is_planner steps, stop at done)."""
    torch = torch_mod
    E, L = 48, 5
    pi, env, ref = _setup(oracle_mod, "SawyerLiftObstacle-v0", E, max_episode_steps=4)
    q, _ = _states(env, E, seed=2, spread=0.3)
    env.set_state(torch.tensor(q, device=env.device))
    ref.set_state(q)
    rng = np.random.default_rng(4)
    arm = list(env.facts.arm_qpos_idx)
    g0 = int(env.facts.grip_qpos_idx[0])
    traj = np.repeat(q[:, None, :], L, axis=1)
    traj[:, :, arm] += np.cumsum(rng.uniform(-0.04, 0.04, size=(E, L, 7)), axis=1)
    traj[:, :, g0] += rng.uniform(-0.002, 0.002, size=(E, L))
    plen = rng.integers(0, L + 1, size=E)
    last_extra = rng.uniform(-0.01, 0.01, size=E)
    gamma = 0.99
    disc = np.array([gamma ** k for k in range(L)])
    dev = env.device
    smdp_rew = torch.zeros(E, dtype=torch.float64, device=dev)
    smdp_done = torch.zeros(E, dtype=torch.uint8, device=dev)
    intra = torch.zeros(E, dtype=torch.int64, device=dev)
    env.exec_trajectories(torch.tensor(traj, device=dev), torch.tensor(plen, device=dev), torch.tensor(disc, device=dev), smdp_rew, smdp_done, intra,
                          last_extra=torch.tensor(last_extra, device=dev))
    # the oracle, waypoint by waypoint
    o_rew, o_done, o_intra = np.zeros(E), np.zeros(E, dtype=np.uint8), np.zeros(E, dtype=np.int64)
    alive = plen > 0
    for k in range(L):
        act = alive & (plen > k)
        if not act.any():
            break
        a = traj[:, k][:, arm] - ref.qpos[:, arm]
        extra = np.where(plen - 1 == k, last_extra, traj[:, k, g0] - ref.qpos[:, g0])
        a = np.concatenate([a, extra[:, None]], axis=1)
        ref.step(a, is_planner=True, move_mask=np.where(act, 1, 2).astype(np.uint8))
        o_rew = np.where(act, o_rew + disc[k] * ref.reward, o_rew)
        o_done = np.where(act, ref.done, o_done)
        o_intra = np.where(act, k, o_intra)
        alive = alive & ~(act & (ref.done != 0))
    assert np.array_equal(_bits(smdp_rew.cpu().numpy()), _bits(o_rew))
    assert np.array_equal(smdp_done.cpu().numpy(), o_done) and np.array_equal(intra.cpu().numpy(), o_intra)
    assert np.array_equal(_bits(env.qpos.cpu().numpy()), _bits(ref.qpos)) and np.array_equal(_bits(env.qvel.cpu().numpy()), _bits(ref.qvel))
    assert o_done.any() and (o_intra > 1).any()


def test_rollout_runs_on_the_dynamics_env(torch_mod):
    """BatchMoPARollout over an env whose physics is the servo dynamics + contacts (stage C): the planner / direct routing is
    unchanged (it reads qpos), paths are executed through the physics, counters move, nothing goes non-finite."""
    torch = torch_mod
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    E = 128
    env = make_env("SawyerPushObstacle-v0", E, dynamics=True, contacts=True, seed=3)
    env.reset()
    ro = BatchMoPARollout(env, RolloutConfig(timelimit=0.1, max_nodes=512, max_path=128, num_trials=10))
    g = torch.Generator(device=env.device)
    g.manual_seed(1)
    n_pl = 0
    for t in range(4):
        ac = (torch.rand(E, 7, generator=g, dtype=torch.float64, device=env.device) * 2 - 1).contiguous()
        out = ro.agent_step(ac)
        assert bool(torch.isfinite(out["ob_next"]).all()) and bool(torch.isfinite(out["rew"]).all())
        n_pl += int(out["is_planner"].sum())
        env.reset(out["done"].bool())
    assert n_pl > 0 and int(ro.counters["rl"].sum()) > 0 and int(ro.counters["interpolation"].sum()) > 0
    assert float(env.qvel.abs().max()) > 1e-3


@pytest.mark.parametrize("env_name", ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0"])
def test_chunked_walks_give_every_env_the_same_transitions(torch_mod, env_name):
    """`walk_chunk` on a dynamics env: a call executes at most that many waypoints per env (the first in the same launch as
    the call's direct and failed-plan steps), envs still on their path sit out the following calls.  Every env must go through
    exactly the transitions -- and waypoint records -- of the run that walks every path to its end within its call."""
    torch = torch_mod
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    E, T = 96, 4
    rng = np.random.default_rng(11)
    n_ac = 8 if env_name == "SawyerLiftObstacle-v0" else 7
    AC = rng.uniform(-1, 1, size=(E, T, n_ac)) * rng.choice([0.5, 0.69, 1.0], size=(E, T, 1))
    AC[: E // 3, 1, 1], AC[: E // 3, 1, 3] = 1.0, -1.0           # blocked straight lines: RRT-Connect queries
    ACt = torch.tensor(AC, device="cuda")
    runs = {}
    for mode, chunk, asyn in (("whole", 0, False), ("chunk1", 1, True), ("chunk3", 3, True), ("chunk2_lockstep", 2, False)):
        env = make_env(env_name, E, dynamics=True, contacts=True, seed=5, max_episode_steps=14)
        env.reset()
        ro = BatchMoPARollout(env, RolloutConfig(timelimit=0.1, max_nodes=512, max_path=128, num_trials=10, async_planner=asyn,
                                                 planner_first_iters=60, planner_min_job=1, walk_chunk=chunk))
        seq = [[] for _ in range(E)]
        recs = [[] for _ in range(E)]
        calls = n_sitting = 0
        while min(len(q) for q in seq) < T:
            te = ro.t_env.clamp(max=T - 1)
            ac = ACt[torch.arange(E, device="cuda"), te].contiguous()
            out = ro.agent_step(ac, record=True)
            st = out["stepped"].cpu().numpy()
            rows = np.concatenate([out["rew"].cpu().numpy()[:, None], out["done"].cpu().numpy()[:, None].astype(np.float64),
                                   out["intra_steps"].cpu().numpy()[:, None].astype(np.float64),
                                   out["is_planner"].cpu().numpy()[:, None].astype(np.float64), out["plan_ok"].cpu().numpy()[:, None].astype(np.float64),
                                   out["path_len"].cpu().numpy()[:, None].astype(np.float64),
                                   env.qpos.cpu().numpy(), env.qvel.cpu().numpy(), out["ac"].cpu().numpy(), out["ob"].cpu().numpy(),
                                   out["ob_next"].cpu().numpy()], axis=1)
            r = out["record"]
            nx = r["n_exec"].cpu().numpy()
            r_ob, r_rew, r_done, r_wp = (r[k].cpu().numpy() for k in ("ob", "meta_rew", "done", "waypoint"))
            for e in np.where(st)[0]:
                if len(seq[e]) < T:
                    seq[e].append(rows[e])
                    k = int(nx[e])
                    recs[e].append(np.concatenate([[k], r_ob[e, :k].ravel(), r_rew[e, :k], r_done[e, :k].astype(np.float64), r_wp[e, :k].ravel()]))
            n_sitting += int((~st).sum())
            calls += 1
            assert calls < 400
        runs[mode] = (np.array([np.array(q) for q in seq]), recs, calls, n_sitting, {k: int(v.sum()) for k, v in ro.counters.items()})
        env.close()
    a = runs["whole"]
    assert a[3] == 0 and a[2] == T
    assert a[4]["interpolation"] > 0 and a[4]["rl"] > 0 and a[4]["mp"] + a[4]["mp_fail"] > 0
    assert a[0][:, :, 2].max() >= 4 and a[0][:, :, 1].sum() > 0          # walks longer than every chunk, and walks cut by `done`
    for mode in ("chunk1", "chunk3", "chunk2_lockstep"):
        b = runs[mode]
        assert np.array_equal(_bits(a[0]), _bits(b[0])), mode
        for e in range(E):
            for t in range(T):
                assert np.array_equal(_bits(a[1][e][t]), _bits(b[1][e][t])), (mode, e, t)
        assert b[3] > 0                                                  # envs did sit calls out


# ---- stage C: contacts of arm and object behind the constraint solver (K7, mopa_contact.inc) -------------------------------
def _setup_ct(oracle_mod, env_name, E, contact_options=None, **kw):
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env_name)
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    env = make_env(env_name, E, dynamics=True, contacts=True, contact_options=contact_options, **kw)
    ref = oracle_mod.OracleEnv(orc, env.facts, E, ac_scale=env.ac_scale, dyn=env.dyn, ct=env.ct, **kw)
    return pi, orc, env, ref


def _ct_states(env, orc, E, seed):
    """arm around its initial pose (some rows pushed into the table / the bin), the object resting, scattered around the
    hand, dropped from above, tilted, with random velocities -- so that rest, impact, sliding, arm-object and arm-scene
    contacts all occur in one batch"""
    from mopa_rl_amd.mjcf import _quat_to_mat
    d, ct, f = env.dyn, env.ct, env.facts
    rng = np.random.default_rng(seed)
    q = np.tile(env.init_qpos_row, (E, 1))
    q[:, d.qadr[:7]] += rng.normal(0, 0.2, size=(E, 7))
    q[:, d.qadr[:7]] = np.clip(q[:, d.qadr[:7]], d.lo[:7], d.hi[:7])
    v = np.zeros((E, d.nd + 6))
    v[:, :7] = rng.normal(0, 0.5, size=(E, 7))
    oq = ct.obj_qadr
    rest = env.init_qpos_row[oq:oq + 7].copy()
    for e in range(E):
        xp, xq = orc.fk_bodies(q[e])
        b = int(f.frame_body[0])
        hand = xp[b] + _quat_to_mat(xq[b]) @ f.frame_off[0]
        mode = e % 4
        if mode == 0:          # where the scene puts it (resting / settling)
            q[e, oq:oq + 7] = rest
            q[e, oq:oq + 2] += rng.normal(0, 0.01, 2)
        elif mode == 1:        # at the hand
            q[e, oq:oq + 3] = hand + rng.normal(0, 0.04, 3)
            qq = rng.normal(size=4)
            q[e, oq + 3:oq + 7] = qq / np.linalg.norm(qq)
        elif mode == 2:        # dropped from a little above its rest pose, tilted
            q[e, oq:oq + 7] = rest
            q[e, oq + 2] += rng.uniform(0.0, 0.05)
            qq = rest[3:] + rng.normal(0, 0.1, 4)
            q[e, oq + 3:oq + 7] = qq / np.linalg.norm(qq)
        else:                  # anywhere
            q[e, oq:oq + 3] = rest[:3] + rng.normal(0, [0.15, 0.15, 0.1])
            qq = rng.normal(size=4)
            q[e, oq + 3:oq + 7] = qq / np.linalg.norm(qq)
        v[e, d.nd:d.nd + 3] = rng.normal(0, 0.2, 3)
        v[e, d.nd + 3:] = rng.normal(0, 1.0, 3)
    return q, v


@pytest.mark.parametrize("env_name", ENVS)
@pytest.mark.parametrize("n,solver", [(1, "newton"), (4, "newton"), (75, "newton"), (4, "newton-pyramidal"), (75, "newton-pyramidal"), (4, "pgs"), (75, "pgs")])
def test_contact_substeps_bit_exact(oracle_mod, torch_mod, env_name, n, solver):
    """sub-steps with contacts, all three solver forms (Newton with the XML's ELLIPTIC cones: MuJoCo's default solver, what the reference
    runs; Newton with pyramidal cones: round 4's form; projected Gauss-Seidel: its first form), each followed by the noslip pass: state
    after n sub-steps bit for bit against the oracle"""
    torch = torch_mod
    E = 130                 # not a multiple of 4: the last workgroup carries idle groups
    opts = {"newton": {"solver": "newton"}, "newton-pyramidal": {"solver": "newton", "cone": "pyramidal"}, "pgs": {"solver": "pgs"}}[solver]
    pi, orc, env, ref = _setup_ct(oracle_mod, env_name, E, contact_options=opts)
    assert env.ct.solver == {"pgs": 0, "newton-pyramidal": 1, "newton": 2}[solver]
    d = env.dyn
    q, v = _ct_states(env, orc, E, seed=10 + n)
    rng = np.random.default_rng(200 + n)
    ctrl = q[:, d.qadr] + rng.uniform(-0.3, 0.3, size=(E, d.nd)) * np.where(d.jtype == 3, 1.0, 0.02)
    env.set_state(torch.tensor(q, device=env.device))
    env.qvel.copy_(torch.tensor(v, device=env.device))
    lag0 = env.dyn_forward()[0]
    env.bias_lag.copy_(lag0)
    env.dyn_substeps(torch.tensor(ctrl, device=env.device), n)
    gq, gv, gl = env.qpos.cpu().numpy(), env.qvel.cpu().numpy(), env.bias_lag.cpu().numpy()
    lag0 = lag0.cpu().numpy()
    touched = 0
    for e in range(E):
        st0 = ref.dyn.stats.contacts
        oq, ov, ol = ref.dyn.step(q[e], v[e], lag0[e], ctrl[e], n)
        touched += int(ref.dyn.stats.contacts > st0)
        assert np.array_equal(_bits(gq[e]), _bits(oq)), f"env {e}: qpos (object pose included) after {n} sub-steps"
        assert np.array_equal(_bits(gv[e]), _bits(ov)), f"env {e}: qvel (object velocity included)"
        assert np.array_equal(_bits(gl[e]), _bits(ol)), f"env {e}: lagged bias"
    assert touched > E // 4
    if solver == "newton":
        # the XML's condim is what was solved: the batch holds contacts with a torsional row (the object: condim 4) and, where the gripper
        # meets the object, rolling rows (the pads: condim 6) -- and nothing was downgraded
        pair_of = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(env.ct.pr_f, env.ct.pr_s))}
        dims = set()
        for e in range(E):
            dims |= {int(env.ct.pr_par[pair_of[(int(r[7]), int(r[8]))]][8]) for r in ref.dyn.contacts(q[e])}
        assert env.ct.condim_downgraded == 0 and 6 in dims and (4 in dims or env_name == "SawyerAssemblyObstacle-v0"), dims
    else:
        assert env.ct.condim_downgraded > 0 and set(env.ct.pr_par[:, 8]) == {3.0}


@pytest.mark.parametrize("env_name", ENVS)
@pytest.mark.parametrize("opts", [{"arena": 150}, {"arena": 330, "maxpair": 3}, {"condim": "3"}, {"maxcon": 5, "maxpair": 2}])
def test_contact_record_arena_and_caps_bit_exact(oracle_mod, torch_mod, env_name, opts):
    """round 6: the elliptic form's contact records are of variable size (8 + 3 dim + dim (dim + 1) / 2 + 6 dim [object] + nd dim [arm] doubles)
    and share an LDS arena; a contact whose record does not fit is dropped, like one beyond maxcon / maxpair.  Small arenas, small caps and
    condim forced to 3: 30 sub-steps bit for bit against the oracle, and the same number of dropped contacts on both sides."""
    torch = torch_mod
    E = 66
    pi, orc, env, ref = _setup_ct(oracle_mod, env_name, E, contact_options=opts)
    assert env.ct.solver == 2 and env.ct.arena == opts.get("arena", env.ct.arena) and env.ct.arena >= 135
    d = env.dyn
    q, v = _ct_states(env, orc, E, seed=77)
    rng = np.random.default_rng(78)
    ctrl = q[:, d.qadr] + rng.uniform(-0.3, 0.3, size=(E, d.nd)) * np.where(d.jtype == 3, 1.0, 0.02)
    env.set_state(torch.tensor(q, device=env.device))
    env.qvel.copy_(torch.tensor(v, device=env.device))
    lag0 = env.dyn_forward()[0]
    env.bias_lag.copy_(lag0)
    stats = torch.zeros(E, 4, dtype=torch.int32, device=env.device)
    from mopa_rl_amd import _lib
    _lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, stats.data_ptr()))
    env.dyn_substeps(torch.tensor(ctrl, device=env.device), 30)
    gq, gv = env.qpos.cpu().numpy(), env.qvel.cpu().numpy()
    st = stats.cpu().numpy()
    lag0 = lag0.cpu().numpy()
    dropped = 0
    for e in range(E):
        d0, c0 = ref.dyn.stats.dropped, ref.dyn.stats.contacts
        oq, ov, ol = ref.dyn.step(q[e], v[e], lag0[e], ctrl[e], 30)
        assert np.array_equal(_bits(gq[e]), _bits(oq)) and np.array_equal(_bits(gv[e]), _bits(ov)), (opts, e)
        assert st[e, 0] == ref.dyn.stats.contacts - c0 and st[e, 2] == ref.dyn.stats.dropped - d0, (opts, e, st[e])
        dropped += int(st[e, 2])
    assert dropped > 0 or "condim" in opts
    env.close()


def test_joint_limit_rows_bit_identical_to_oracle(oracle_mod, torch_mod):
    """joints servoed beyond their ranges (every arm joint, either side, over the envs): with the limits as rows of the Newton solver the
    joints end a few mrad BEYOND the range, resting on the soft row -- sub-steps bit for bit against the oracle; with
    limit_rows=False they sit exactly on it (stage A's inelastic stop)"""
    torch = torch_mod
    E = 56
    for lr in (True, False):
        pi, orc, env, ref = _setup_ct(oracle_mod, "SawyerLiftObstacle-v0", E, contact_options={"limit_rows": lr})
        d = env.dyn
        q = np.tile(env.init_qpos_row, (E, 1))
        v = np.zeros((E, env.qvel.shape[1]))
        ctrl = q[:, d.qadr].copy()
        for e in range(E):
            j = e % 7
            ctrl[e, j] = (d.hi[j] + 0.4) if (e // 7) % 2 == 0 else (d.lo[j] - 0.4)
            q[e, d.qadr[j]] = (d.hi[j] - 0.05) if (e // 7) % 2 == 0 else (d.lo[j] + 0.05)
        env.set_state(torch.tensor(q, device=env.device))
        env.qvel.copy_(torch.tensor(v, device=env.device))
        lag0 = env.dyn_forward()[0]
        env.bias_lag.copy_(lag0)
        env.dyn_substeps(torch.tensor(ctrl, device=env.device), 150)
        gq, gv = env.qpos.cpu().numpy(), env.qvel.cpu().numpy()
        lag0 = lag0.cpu().numpy()
        beyond = 0
        for e in range(E):
            oq, ov, ol = ref.dyn.step(q[e], v[e], lag0[e], ctrl[e], 150)
            assert np.array_equal(_bits(gq[e]), _bits(oq)) and np.array_equal(_bits(gv[e]), _bits(ov)), (lr, e)
            j = e % 7
            over = max(oq[d.qadr[j]] - d.hi[j], d.lo[j] - oq[d.qadr[j]])
            beyond += int(over > 1e-4)
            assert over < 0.03 and (lr or over <= 0.0)
        assert beyond > (E // 2 if lr else -1) and (lr or beyond == 0)
        env.close()


def test_contact_caps_per_solver_form(torch_mod):
    """Newton with pyramidal cones maps a contact's four pyramid rows onto the 16 lanes of an env, two rows per lane: maxcon <= 8; with
    elliptic cones a contact's cone lives on one lane: maxcon <= 16, as for the Gauss-Seidel form"""
    from mopa_rl_amd.kinematic_env import make_env
    with pytest.raises(ValueError, match="maxcon <= 8"):
        make_env("SawyerPushObstacle-v0", 8, dynamics=True, contacts=True, contact_options={"maxcon": 12, "solver": "newton", "cone": "pyramidal"})
    with pytest.raises(ValueError, match="maxcon <= 16"):
        make_env("SawyerPushObstacle-v0", 8, dynamics=True, contacts=True, contact_options={"maxcon": 20})
    env = make_env("SawyerPushObstacle-v0", 8, dynamics=True, contacts=True, contact_options={"maxcon": 12})
    env.reset()
    env.step(torch_mod.zeros(8, env.action_dim, dtype=torch_mod.float64, device=env.device))
    assert bool(torch_mod.isfinite(env.qpos).all())
    env.close()


@pytest.mark.parametrize("env_name", ENVS)
def test_contact_env_rollout_bit_identical_to_oracle(oracle_mod, torch_mod, env_name):
    torch = torch_mod
    E = 40
    pi, orc, env, ref = _setup_ct(oracle_mod, env_name, E, max_episode_steps=5)
    q, v = _ct_states(env, orc, E, seed=21)
    env.set_state(torch.tensor(q, device=env.device))
    ref.set_state(q)
    rng = np.random.default_rng(5)
    for t in range(5):
        a = rng.uniform(-1.5, 1.5, size=(E, env.action_dim))
        obs, rew, done, info = env.step(torch.tensor(a, device=env.device))
        ref.step(a, nthreads=8)
        assert np.array_equal(_bits(env.qpos.cpu().numpy()), _bits(ref.qpos)), f"step {t}: qpos"
        assert np.array_equal(_bits(env.qvel.cpu().numpy()), _bits(ref.qvel)), f"step {t}: qvel"
        assert np.array_equal(_bits(obs.cpu().numpy()), _bits(ref.obs)), f"step {t}: obs"
        assert np.array_equal(_bits(rew.cpu().numpy()), _bits(ref.reward)), f"step {t}: reward"
        assert np.array_equal(done.cpu().numpy(), ref.done)


def test_contact_free_envs_equal_the_servo_dynamics(oracle_mod, torch_mod):
    """an env in which nothing touches anything and the object is in free fall far away: K7's arm state after a step equals
    K6's (stage A) bit for bit -- the contact stage adds exact zeros"""
    torch = torch_mod
    from mopa_rl_amd.kinematic_env import make_env
    E = 64
    a = make_env("SawyerPushObstacle-v0", E, dynamics=True, seed=2)
    b = make_env("SawyerPushObstacle-v0", E, dynamics=True, contacts=True, seed=2)
    q = np.tile(a.init_qpos_row, (E, 1))
    rng = np.random.default_rng(0)
    q[:, a.dyn.qadr[:7]] += rng.normal(0, 0.02, size=(E, 7))
    oq = b.ct.obj_qadr
    q[:, oq:oq + 3] = [3.0, 3.0, 50.0]
    for env in (a, b):
        env.set_state(torch.tensor(q, device=env.device))
    act = torch.tensor(rng.uniform(-0.3, 0.3, size=(E, 7)), device=a.device)
    a.step(act)
    b.step(act)
    qa, qb = a.qpos.cpu().numpy(), b.qpos.cpu().numpy()
    assert np.array_equal(_bits(qa[:, a.dyn.qadr]), _bits(qb[:, a.dyn.qadr]))
    assert np.array_equal(_bits(a.qvel.cpu().numpy()), _bits(b.qvel.cpu().numpy()[:, :a.dyn.nd]))
    assert np.all(qb[:, oq + 2] < 50.0)            # the object fell


def test_arm_stops_at_the_bin_roof_on_the_gpu(oracle_mod, torch_mod):
    """the scenario of tests/test_oracle_contact.py::test_arm_stops_at_the_bin_roof through the env's own stepping entry point
    (K7): the hand servoed to a point below the roof plate of the Push bin comes to rest ON the plate -- and every env's
    state equals the oracle's, sub-step by sub-step"""
    torch = torch_mod
    from mopa_rl_amd.mjcf import _quat_to_mat
    from mopa_rl_amd.scene import ENV_SPECS
    env_name = "SawyerPushObstacle-v0"
    E = 6
    pi, orc, env, ref = _setup_ct(oracle_mod, env_name, E)
    d, f, m = env.dyn, env.facts, pi.model
    q0 = np.tile(env.init_qpos_row, (E, 1))
    jid = [m.joint_name2id(j) for j in ENV_SPECS[env_name].robot_joints]
    ctrl = np.zeros((E, d.nd))
    for e in range(E):
        qt, err, steps, ok = orc.ik_solve(q0[e], np.array([0.78 + 0.01 * e, 0.0, 1.00]), jid, int(f.frame_body[0]), f.frame_off[0], max_steps=400, tol=1e-4)
        assert ok
        ctrl[e] = qt[d.qadr]
        ctrl[e, 7:] = q0[e, d.qadr[7:]]
    env.set_state(torch.tensor(q0, device=env.device))
    lag0 = env.dyn_forward()[0]
    env.bias_lag.copy_(lag0)
    oq, ov, ol = q0.copy(), np.zeros((E, d.nd + 6)), lag0.cpu().numpy().copy()
    for k in range(10):
        env.dyn_substeps(torch.tensor(ctrl, device=env.device), 75)
        for e in range(E):
            oq[e], ov[e], ol[e] = ref.dyn.step(oq[e], ov[e], ol[e], ctrl[e], 75)
        assert np.array_equal(_bits(env.qpos.cpu().numpy()), _bits(oq)) and np.array_equal(_bits(env.qvel.cpu().numpy()), _bits(ov))
    gq = env.qpos.cpu().numpy()
    for e in range(E):
        xp, xq = orc.fk_bodies(gq[e])
        b = int(f.frame_body[0])
        eef = xp[b] + _quat_to_mat(xq[b]) @ f.frame_off[0]
        assert eef[2] > 1.2 and orc.is_valid(gq[e])[0] and np.abs(gq[e, d.qadr[:7]] - ctrl[e, :7]).max() > 0.1


def test_contact_free_dynamics_in_the_16_lane_mapping_equals_k6(torch_mod):
    """dyn_lanes=16: stage A's contact-free servo dynamics through the contact kernel's mapping (16 lanes per env, no object, no pairs,
    joint limits as inelastic stops) -- env.step rollouts bit-identical to K6's lane-per-env form"""
    torch = torch_mod
    from mopa_rl_amd.kinematic_env import make_env
    E = 70
    for env_name in ENVS:
        a = make_env(env_name, E, dynamics=True, seed=3, max_episode_steps=1 << 20)
        b = make_env(env_name, E, dynamics=True, seed=3, max_episode_steps=1 << 20, dyn_lanes=16)
        assert a.dyn_lanes == 1 and b.dyn_lanes == 16
        a.reset(); b.reset()
        g = torch.Generator(device=a.device); g.manual_seed(5)
        for t in range(4):
            act = (torch.rand(E, a.action_dim, generator=g, dtype=torch.float64, device=a.device) * 3 - 1.5)
            a.step(act); b.step(act)
            for x, y in ((a.qpos, b.qpos), (a.qvel, b.qvel), (a.bias_lag, b.bias_lag), (a.obs, b.obs), (a.reward, b.reward)):
                assert torch.equal(x.view(torch.int64), y.view(torch.int64)), (env_name, t)
        a.close(); b.close()
