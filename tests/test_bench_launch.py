"""`bench.py --gpus N` must really run N ranks (round-2 review: the flag was parsed and ignored).  CPU: the launch /
rendezvous path with the gloo backend and no GPU work (`--rendezvous-only`); a mismatching external launcher is an error."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(MOPA_BENCH_BACKEND="gloo", **kw)
    return e


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen_by_all_reduce"] == 2
    assert out["config"]["parallelism"] == "env-shard x2" and len(out["exchange"]["collectives"]) == 3


def test_gpus_1_stays_single_process():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rendezvous-only"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--rendezvous-only"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_gpus_2_full_bench_on_one_device():
    """Two ranks pinned to the same GPU over gloo (test knobs): the whole multi-rank bench path incl. the collectives."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs", "256", "--samples", "64",
                        "--no-rollout"], env=_env(MOPA_BENCH_DEVICE="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["exchange"]["ranks"] == 2 and out["value"] > 0
    assert out["config"]["parallelism"].startswith("env-shard x2") and "gloo all_gather" in out["config"]["parallelism"]      # the backend that ran, by name
    assert out["exchange"]["ranks_seen"] == [0, 1] and len(out["exchange"]["kernel_ms_per_rank"]) == 2


def _preflight_worker(rank, world, port, q, lie):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = bench.exchange_preflight(torch, dist, world, 0 if lie else rank, torch.device("cpu"))      # lie: both ranks stamp themselves 0
        q.put((rank, "ok", seen))
    except SystemExit as e:
        q.put((rank, "refused", str(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lie", [False, True])
def test_exchange_preflight_under_gloo(lie):
    """bench.py --gpus N: before any timing every rank all-gathers a rank-stamped tensor over the real backend and must see N distinct
    stamps; a collective that does not reach every rank (here: two ranks claiming the same stamp) refuses to run, on every rank"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_preflight_worker, args=(r, 2, port, q, lie)) for r in range(2)]
    [p.start() for p in ps]
    got = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    if lie:
        assert [g[1] for g in got] == ["refused", "refused"] and "preflight failed" in got[0][2]
    else:
        assert [g[1:] for g in got] == [("ok", [0, 1]), ("ok", [0, 1])]
