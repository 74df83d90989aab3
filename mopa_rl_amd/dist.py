"""Multi-GPU plumbing: env sharding + the collectives of the hot path.

One process per GPU (torchrun); `torch.distributed` backend "nccl" is RCCL on ROCm.  The validity /
planning work itself needs no communication -- every (env, state) is independent -- so envs are block
partitioned over ranks and the only exchange per step is an all-gather of the uint8 validity masks (and,
for rollouts, of the transition records).  The reference's only distributed code is host-side mpi4py
(util/mpi.py:1-33, util/pytorch.py:107-159); `all_reduce_mean_` is the device-side stand-in for its
`sync_grads` Allreduce(SUM)/size.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Block partition [lo, hi) of n_items over `world` ranks (first n % world ranks get one extra)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist


def world_info() -> Tuple[int, int]:
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def all_gather_concat(local, sizes=None):
    """Concatenate a per-rank tensor along dim 0 on every rank.  Equal sizes use one
    all_gather_into_tensor; ragged shards (uneven env partition) fall back to padded gather."""
    import torch
    dist = _dist()
    world, _ = world_info()
    if world == 1:
        return local
    n = local.shape[0]
    if sizes is None:
        t = torch.tensor([n], dtype=torch.int64, device=local.device)
        all_n = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_n, t)
        sizes = [int(x.item()) for x in all_n]
    if len(set(sizes)) == 1:
        out = torch.empty((world * n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        try:
            dist.all_gather_into_tensor(out, local.contiguous())
            return out
        except (RuntimeError, NotImplementedError):
            pass
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:n] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def all_reduce_mean_(flat):
    """In-place mean over ranks of a flat gradient buffer (reference util/pytorch.py:153-159)."""
    dist = _dist()
    world, _ = world_info()
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
    return flat


class OverlappedGather:
    """Multi-buffered all-gather of a fixed-size per-rank tensor (the uint8 verdict masks of a step), overlapped with
    the producers of the NEXT steps: `buffer(k)` hands out the local buffer of step k after waiting for the collective
    that may still be reading it (the one launched `depth` steps earlier), `launch(k)` starts the asynchronous all-gather
    of step k on the backend's own stream, `result(k)` waits for it and returns the gathered tensor [world * n],
    `drain()` waits for everything in flight.  depth = 3: the validity kernel is persistent and fills every CU, so the
    collective of step k only gets CUs in the tail of kernel k+1 -- with two buffers kernel k+2 would then have to wait
    for it; with three it has a whole kernel of slack.  With one rank nothing is communicated and `result` is the local
    buffer."""

    def __init__(self, n: int, dtype, device, depth: int = 3):
        import torch
        self.world, self.rank = world_info()
        self.depth = depth
        self.local = [torch.empty(n, dtype=dtype, device=device) for _ in range(depth)]
        self.gathered = [torch.empty(self.world * n, dtype=dtype, device=device) for _ in range(depth)] if self.world > 1 else None
        self.pending = [None] * depth

    def _wait(self, b: int):
        if self.pending[b] is not None:
            self.pending[b].wait()          # the current stream waits for the collective that still uses buffer b
            self.pending[b] = None

    def buffer(self, k: int):
        b = k % self.depth
        self._wait(b)
        return self.local[b]

    def launch(self, k: int):
        if self.world > 1:
            b = k % self.depth
            self.pending[b] = _dist().all_gather_into_tensor(self.gathered[b], self.local[b], async_op=True)

    def result(self, k: int):
        b = k % self.depth
        self._wait(b)
        return self.gathered[b] if self.world > 1 else self.local[b]

    def drain(self):
        for b in range(self.depth):
            self._wait(b)


class TransitionExchange:
    """All-gather of the transition records of one agent step (every rank fills the same replay
    buffer; the reference keeps one buffer per MPI rank and never exchanges transitions, rl/dataset.py).  A record is one
    float32 row per env:  ob | ac | rew | done | intra_steps | stepped | ob_next  (Lift: 35 + 8 + 4 + 35 = 82 floats = 328 B;
    `stepped` = 1 for envs that completed a transition this call -- with the asynchronous planner the rows of envs still
    waiting for a query are meaningless and a receiving rank drops them by this column).
    `pack(k, ...)` fills the local buffer of step k on the current stream, `launch(k)` starts the asynchronous all-gather on
    the backend's stream (so it overlaps whatever the current stream does next -- the next agent step's validity / planner
    launches), `result(k)` waits for it and returns the fields as views of the gathered [world * E, W] tensor.  Buffers are
    reused round-robin (`depth`); a buffer is handed out again only after the collective reading it has completed.
    With one rank nothing is communicated."""

    def __init__(self, n_envs: int, obs_dim: int, ac_dim: int, device, depth: int = 2):
        import torch
        self.world, self.rank = world_info()
        self.E, self.obs_dim, self.ac_dim, self.depth = int(n_envs), int(obs_dim), int(ac_dim), int(depth)
        self.width = 2 * self.obs_dim + self.ac_dim + 4
        self.local = [torch.zeros(self.E, self.width, dtype=torch.float32, device=device) for _ in range(depth)]
        self.gathered = ([torch.zeros(self.world * self.E, self.width, dtype=torch.float32, device=device) for _ in range(depth)]
                         if self.world > 1 else None)
        self.pending = [None] * depth

    @property
    def bytes_per_step(self) -> int:
        return self.E * self.width * 4

    def _wait(self, b: int):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def pack(self, k: int, ob, ac, rew, done, intra_steps, ob_next, stepped=None):
        b = k % self.depth
        self._wait(b)
        buf, o, a = self.local[b], self.obs_dim, self.ac_dim
        buf[:, :o] = ob
        buf[:, o:o + a] = ac[:, :a]
        buf[:, o + a] = rew
        buf[:, o + a + 1] = done
        buf[:, o + a + 2] = intra_steps
        if stepped is None:
            buf[:, o + a + 3] = 1.0
        else:
            buf[:, o + a + 3] = stepped
        buf[:, o + a + 4:] = ob_next
        return buf

    def launch(self, k: int):
        if self.world > 1:
            b = k % self.depth
            self.pending[b] = _dist().all_gather_into_tensor(self.gathered[b], self.local[b], async_op=True)

    def result(self, k: int):
        b = k % self.depth
        self._wait(b)
        g, o, a = (self.gathered[b] if self.world > 1 else self.local[b]), self.obs_dim, self.ac_dim
        return {"ob": g[:, :o], "ac": g[:, o:o + a], "rew": g[:, o + a], "done": g[:, o + a + 1], "intra_steps": g[:, o + a + 2],
                "stepped": g[:, o + a + 3], "ob_next": g[:, o + a + 4:]}

    def drain(self):
        for b in range(self.depth):
            self._wait(b)
