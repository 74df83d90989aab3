"""world_size-2 `gloo` test of the env-sharded path (runs on CPU).

Each rank owns a block of envs, computes its validity masks (the oracle stands in for the GPU kernel:
this is a test of the sharding/gather plumbing in mopa_rl_amd/dist.py, not of the kernel), and the
all-gathered result must equal the unsharded computation, independent of the number of ranks."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E, S, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import sample_states
    from mopa_rl_amd.dist import all_gather_concat, all_reduce_mean_, shard_range
    from mopa_rl_amd.scene import planner_inputs
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pi = planner_inputs("SawyerPushObstacle-v0")
    orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    qa, row = sample_states(pi, E * S, 5, "near")       # every rank can regenerate the global batch
    rows = np.repeat(row, E, axis=0)
    rows[:, 7] = np.linspace(-0.008, 0.015, E)           # per-env passive state
    lo, hi = shard_range(E, world, rank)
    v, _ = orc.is_valid_batch(qa[lo * S:hi * S], rows[lo:hi], samples_per_env=S)
    full = all_gather_concat(torch.from_numpy(v))
    g = torch.full((10,), float(rank + 1))
    all_reduce_mean_(g)
    if rank == 0:
        q.put((full.numpy(), g.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,E", [(2, 8), (2, 7)])
def test_sharded_validity_matches_unsharded(world, E, oracle_mod):
    import torch.multiprocessing as mp
    from conftest import sample_states
    from mopa_rl_amd.scene import planner_inputs
    S = 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, g = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pi = planner_inputs("SawyerPushObstacle-v0")
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    qa, row = sample_states(pi, E * S, 5, "near")
    rows = np.repeat(row, E, axis=0)
    rows[:, 7] = np.linspace(-0.008, 0.015, E)
    want, _ = orc.is_valid_batch(qa, rows, samples_per_env=S)
    assert np.array_equal(full, want)
    np.testing.assert_allclose(g, np.mean(np.arange(1, world + 1)))


def test_shard_range_partition():
    from mopa_rl_amd.dist import shard_range
    for n in (0, 1, 7, 8, 4096, 4099):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _og_worker(rank, world, port, n, steps, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mopa_rl_amd.dist import OverlappedGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    og = OverlappedGather(n, torch.uint8, torch.device("cpu"))
    seen = []
    for k in range(steps):
        buf = og.buffer(k)                       # safe to overwrite: the gather of step k-2 has completed
        buf.copy_(torch.full((n,), (17 * k + 3 * rank) % 251, dtype=torch.uint8))
        og.launch(k)
        if k >= 1:
            seen.append(og.result(k - 1).clone())   # consume step k-1 while step k's collective is in flight
    seen.append(og.result(steps - 1).clone())
    og.drain()
    if rank == 0:
        q.put(torch.stack(seen).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_double_buffering():
    """bench.py's per-step exchange (mopa_rl_amd/dist.py::OverlappedGather) on 2 gloo ranks: every step's gathered masks
    are that step's values from both ranks, although buffers are reused every third step."""
    import torch.multiprocessing as mp
    world, n, steps = 2, 1000, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_og_worker, args=(r, world, port, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape == (steps, world * n)
    for k in range(steps):
        for r in range(world):
            assert (got[k, r * n:(r + 1) * n] == (17 * k + 3 * r) % 251).all(), (k, r)


def _tx_rollout(oracle_mod, env_name, lo, hi, E, T):
    """T direct kinematic env steps of envs [lo, hi) of an E-env job on the CPU oracle; states and actions are keyed by the
    GLOBAL env id, so a shard computes exactly the rows it owns of the unsharded job"""
    from mopa_rl_amd.kinematic_env import env_facts
    from mopa_rl_amd.scene import default_qpos, planner_inputs
    pi = planner_inputs(env_name)
    orc = oracle_mod.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    f = env_facts(env_name, pi.model)
    n = hi - lo
    env = oracle_mod.OracleEnv(orc, f, n, ac_scale=pi.spec.ac_scale, max_episode_steps=3)
    q = np.tile(default_qpos(env_name, pi.model), (n, 1))
    for i, e in enumerate(range(lo, hi)):
        q[i, :7] += np.random.default_rng(1000 + e).normal(0, 0.02, 7)
    env.set_state(q)
    steps = []
    for t in range(T):
        ob = env.obs.copy()
        ac = np.stack([np.random.default_rng(7919 * t + e).uniform(-1, 1, env.action_dim) for e in range(lo, hi)])
        env.step(ac)
        steps.append(dict(ob=ob, ac=ac, rew=env.reward.copy(), done=env.done.copy(), intra_steps=np.zeros(n), ob_next=env.obs.copy()))
    return steps, env.obs_dim, env.action_dim


def _tx_worker(rank, world, port, E, T, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mopa_rl_amd.dist import TransitionExchange, shard_range
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(E, world, rank)
    steps, od, ad = _tx_rollout(O, "SawyerLiftObstacle-v0", lo, hi, E, T)
    tx = TransitionExchange(hi - lo, od, ad, torch.device("cpu"))
    got = []
    for k, s in enumerate(steps):
        stepped = torch.from_numpy(((np.arange(lo, hi) + k) % 3 != 0).astype(np.float64))     # some envs "sit out" a call
        tx.pack(k, *(torch.from_numpy(np.asarray(s[f], dtype=np.float64)) for f in ("ob", "ac", "rew", "done", "intra_steps", "ob_next")),
                stepped=stepped)
        tx.launch(k)
        if k >= 1:
            got.append({f: v.clone().numpy() for f, v in tx.result(k - 1).items()})   # consume step k-1 while step k is in flight
    got.append({f: v.clone().numpy() for f, v in tx.result(T - 1).items()})
    tx.drain()
    if rank == 0:
        q.put((got, tx.bytes_per_step))
    dist.barrier()
    dist.destroy_process_group()


def test_transition_all_gather_equals_unsharded(oracle_mod):
    """BASELINE config 4's exchange (env shard + all-gather of rollout transitions) on 2 gloo ranks: the gathered records of
    every step equal the unsharded job's, field by field (float32 records), with double-buffered asynchronous collectives"""
    import torch.multiprocessing as mp
    world, E, T = 2, 8, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tx_worker, args=(r, world, port, E, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, nbytes = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want, od, ad = _tx_rollout(oracle_mod, "SawyerLiftObstacle-v0", 0, E, E, T)
    assert nbytes == (E // world) * (2 * od + ad + 4) * 4 and (od, ad) == (35, 8)
    for k in range(T):
        for f in ("ob", "ac", "rew", "done", "intra_steps", "ob_next"):
            assert np.array_equal(got[k][f], np.asarray(want[k][f], dtype=np.float32)), (k, f)
        # the `stepped` column travels with the record: a receiving rank can drop the rows of envs that sat the call out
        assert np.array_equal(got[k]["stepped"], ((np.arange(E) + k) % 3 != 0).astype(np.float32)), k
    assert sum(int(w["done"].sum()) for w in want) > 0
