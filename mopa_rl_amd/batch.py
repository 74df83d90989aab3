"""Batched (device-resident) entry points over torch tensors.

torch is used only for device memory and streams: every call hands raw device
pointers to libmopa_hip.so through the C ABI (include/mopa_hip.h).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
from typing import Optional, Tuple

from . import _lib


def _torch():
    import torch
    return torch


def _ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _check_f64(t, name, cols=None):
    torch = _torch()
    if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
        raise _lib.MopaError(f"{name} must be a contiguous float64 tensor on the GPU")
    if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
        raise _lib.MopaError(f"{name} must have shape [*, {cols}], got {tuple(t.shape)}")


def _stream_handle(stream) -> C.c_void_p:
    if stream is not None:
        return C.c_void_p(stream.cuda_stream)
    # the current stream's raw handle without building a torch.cuda.Stream object (a dozen lookups per agent_step call:
    # ~9 us each through torch.cuda.current_stream(), well under 1 us this way)
    return C.c_void_p(_raw_stream_fn()())


_RAW_STREAM = None


def _raw_stream_fn():
    """the cheapest way this torch build offers to read the current stream's raw handle (chosen once): the private
    torch._C._cuda_getCurrentRawStream when it exists, else the documented torch.cuda.current_stream().cuda_stream"""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        torch = _torch()
        tc = torch._C
        if os.environ.get("MOPA_STREAM_PUBLIC_API") != "1" and hasattr(tc, "_cuda_getCurrentRawStream") and hasattr(tc, "_cuda_getDevice"):
            _RAW_STREAM = lambda: tc._cuda_getCurrentRawStream(tc._cuda_getDevice())
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream
    return _RAW_STREAM


class PlanState:
    """Trees, iteration and check counts a planner launch left behind for its queries (`BatchPlanner.plan(keep_state=True)`)."""

    def __init__(self, tree_q, tree_p, state, max_nodes: int, na: int):
        self.tree_q, self.tree_p, self.state, self.max_nodes, self.na = tree_q, tree_p, state, int(max_nodes), int(na)

    def rows(self, idx):
        """the state of the queries `idx` (int64 tensor), in that order -- what a continuing launch of those queries takes"""
        torch = _torch()
        E = self.state.shape[0]
        w = 2 * self.max_nodes * self.na
        tq = torch.empty(len(idx) * w + 8, dtype=torch.float64, device=self.tree_q.device)
        tq[:len(idx) * w].view(len(idx), w).copy_(self.tree_q[:E * w].view(E, w)[idx])
        return PlanState(tq, self.tree_p.view(E, 2 * self.max_nodes)[idx].contiguous().view(-1), self.state[idx].contiguous(),
                         self.max_nodes, self.na)

    @staticmethod
    def cat(parts):
        torch = _torch()
        p0 = parts[0]
        w = 2 * p0.max_nodes * p0.na
        n = sum(p.state.shape[0] for p in parts)
        tq = torch.empty(n * w + 8, dtype=torch.float64, device=p0.tree_q.device)
        o = 0
        for p in parts:
            k = p.state.shape[0] * w
            tq[o:o + k].copy_(p.tree_q[:k])
            o += k
        return PlanState(tq, torch.cat([p.tree_p for p in parts]), torch.cat([p.state for p in parts]), p0.max_nodes, p0.na)


class BatchPlanner:
    """N-state validity / motion checks and E-env RRT-Connect on one GPU."""

    def __init__(self, scene: "_lib.Scene"):
        self.scene = scene
        self.na = scene.na
        self.nq = scene.nq

    # valid[i] for state i = qpos_env[i // samples_per_env] with active entries <- q_active[i]
    def is_valid(self, q_active, qpos_env, samples_per_env: Optional[int] = None, want_min_dist: bool = False,
                 out=None, stream=None, guard: bool = False):
        """guard: states with a joint beyond its range + the pruning proof's guard band are re-evaluated with the full pair list
        (a device-side range test, one host read of the count; off on the hot paths, whose states are sampled / clipped inside
        the ranges -- `Scene.is_valid_state`, the reference's isValidState, always guards)."""
        torch = _torch()
        _check_f64(q_active, "q_active", self.na)
        _check_f64(qpos_env, "qpos_env", self.nq)
        N = q_active.shape[0]
        spe = int(samples_per_env) if samples_per_env is not None else max(1, N // max(1, qpos_env.shape[0]))
        if N and (N + spe - 1) // spe > qpos_env.shape[0]:
            raise _lib.MopaError("qpos_env has fewer rows than ceil(N / samples_per_env)")
        valid = out if out is not None else torch.empty(N, dtype=torch.uint8, device=q_active.device)
        md = torch.empty(N, dtype=torch.float64, device=q_active.device) if want_min_dist else None
        _lib.check(_lib.lib().mopa_is_valid_batch(self.scene.handle, _ptr(q_active), _ptr(qpos_env), N, spe, _ptr(valid),
                                                  _ptr(md) if md is not None else None, _stream_handle(stream)))
        if guard and self.scene.npair_pruned and N:
            sc = self.scene
            dev = q_active.device
            if getattr(self, "_guard_t", None) is None or self._guard_t[0].device != dev:
                pos = {int(a): k for k, a in enumerate(sc.active_idx)}
                act = [k for k, a in enumerate(sc.guard_adr) if int(a) in pos]
                pas = [k for k, a in enumerate(sc.guard_adr) if int(a) not in pos]
                mk = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
                self._guard_t = (mk([pos[int(sc.guard_adr[k])] for k in act], torch.long), mk(sc.guard_lo[act], torch.float64), mk(sc.guard_hi[act], torch.float64),
                                 mk(sc.guard_adr[pas], torch.long), mk(sc.guard_lo[pas], torch.float64), mk(sc.guard_hi[pas], torch.float64))
            ia, la, ha, ip_, lp, hp = self._guard_t
            qa = q_active[:, ia]
            oor = ((qa < la) | (qa > ha)).any(dim=1)
            if len(ip_):
                qp = qpos_env[:, ip_]
                oor_env = ((qp < lp) | (qp > hp)).any(dim=1)
                oor = oor | oor_env[torch.arange(N, device=dev) // spe]
            rows = torch.nonzero(oor).flatten()
            if len(rows):
                if getattr(self, "_full_bp", None) is None:
                    self._full_bp = BatchPlanner(sc.full())
                r = self._full_bp.is_valid(q_active[rows].contiguous(), qpos_env[rows // spe].contiguous(), samples_per_env=1,
                                           want_min_dist=want_min_dist, stream=stream)
                if want_min_dist:
                    valid[rows], md[rows] = r[0], r[1]
                else:
                    valid[rows] = r
        return (valid, md) if want_min_dist else valid

    def check_motion(self, qa, qb, qpos_env, samples_per_env: Optional[int] = None, stream=None):
        torch = _torch()
        _check_f64(qa, "qa", self.na)
        _check_f64(qb, "qb", self.na)
        _check_f64(qpos_env, "qpos_env", self.nq)
        N = qa.shape[0]
        spe = int(samples_per_env) if samples_per_env is not None else max(1, N // max(1, qpos_env.shape[0]))
        valid = torch.empty(N, dtype=torch.uint8, device=qa.device)
        _lib.check(_lib.lib().mopa_check_motion_batch(self.scene.handle, _ptr(qa), _ptr(qb), _ptr(qpos_env), N, spe,
                                                      _ptr(valid), _stream_handle(stream)))
        return valid

    def plan(self, start, goal, max_iters: int = 2000, max_nodes: int = 1024, max_path: int = 256, seed: int = 0,
             env_id_base: int = 0, stream=None, env_ids=None, seeds=None, max_workgroups: int = 0, exclusive: bool = False,
             keep_state: bool = False, resume=None) -> Tuple["object", "object", "object", "object"]:
        """E independent RRT-Connect queries.  Returns (path[E,max_path,nq], path_len[E], status[E], n_checks[E]).
        env_ids (int64 [E] GPU tensor, optional): the sample-stream id of every query (default env_id_base + index).
        seeds (int64 [E] GPU tensor, optional): a seed per query instead of `seed`.
        max_workgroups: 0 = one persistent workgroup per CU (shortest lone launch), < 0 = as many as the chip holds (throughput:
        launches that overlap others), > 0 = explicit cap; exclusive: no other planner workgroup shares this launch's CUs
        (include/mopa_hip.h).
        keep_state: the launch keeps its trees in tensors of its own and returns a fifth element, a `PlanState` (tree_q, tree_p,
        state, max_nodes), from which a later launch with a larger `max_iters` continues the unsolved queries: `resume=` a
        PlanState whose rows are in this launch's query order (`PlanState.rows(idx)` gathers; same max_nodes, same seeds / ids).
        The continued run gives what one launch with the larger budget gives, without retracing the first iterations."""
        torch = _torch()
        _check_f64(start, "start", self.nq)
        _check_f64(goal, "goal", self.nq)
        E = start.shape[0]
        dev = start.device
        ps = None
        if keep_state:
            na = self.scene.na
            ps = PlanState(torch.empty(E * 2 * max_nodes * na + 8, dtype=torch.float64, device=dev),
                           torch.empty(E * 2 * max_nodes, dtype=torch.int32, device=dev),
                           torch.zeros(E, 4, dtype=torch.int64, device=dev), int(max_nodes), na)
        if resume is not None and (resume.max_nodes != int(max_nodes) or resume.state.shape[0] != E):
            raise _lib.MopaError("resume: a PlanState of this launch's queries (same order, same max_nodes) is needed")
        path = torch.zeros(E, max_path, self.nq, dtype=torch.float64, device=dev)
        plen = torch.zeros(E, dtype=torch.int32, device=dev)
        status = torch.zeros(E, dtype=torch.int32, device=dev)
        nchk = torch.zeros(E, dtype=torch.int64, device=dev)
        if env_ids is not None and (env_ids.dtype != torch.int64 or not env_ids.is_cuda or not env_ids.is_contiguous()
                                    or tuple(env_ids.shape) != (E,)):
            raise _lib.MopaError("env_ids must be a contiguous int64 GPU tensor of shape [E]")
        if seeds is not None and (seeds.dtype != torch.int64 or not seeds.is_cuda or not seeds.is_contiguous() or tuple(seeds.shape) != (E,)):
            raise _lib.MopaError("seeds must be a contiguous int64 GPU tensor of shape [E]")
        prm = _lib.MopaPlanParams(int(max_iters), int(max_nodes), int(max_path), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                  int(env_id_base), _ptr(env_ids) if env_ids is not None else None,
                                  _ptr(seeds) if seeds is not None else None, int(max_workgroups), 1 if exclusive else 0,
                                  _ptr(ps.tree_q) if ps else None, _ptr(ps.tree_p) if ps else None, _ptr(ps.state) if ps else None,
                                  _ptr(resume.tree_q) if resume is not None else None, _ptr(resume.tree_p) if resume is not None else None,
                                  _ptr(resume.state) if resume is not None else None)
        _lib.check(_lib.lib().mopa_plan_batch(self.scene.handle, _ptr(start), _ptr(goal), E, C.byref(prm), _ptr(path),
                                              _ptr(plen), _ptr(status), _ptr(nchk), _stream_handle(stream)))
        if resume is not None and stream is not None:
            for t in (resume.tree_q, resume.tree_p, resume.state):
                t.record_stream(stream)
        return (path, plen, status, nchk, ps) if keep_state else (path, plen, status, nchk)

    def plan_laddered(self, batches, max_iters: int = 2000, first_iters: int = 100, max_nodes: int = 1024, max_path: int = 256,
                      retry_streams=None, first_stream=None, max_workgroups_first: int = -1, retry_min: int = 1024, resume: bool = True,
                      retry_exclusive: bool = False):
        """A stream of query batches through RRT-Connect with an iteration ladder.  `batches`: list of dicts with `start`,
        `goal` ([E, nq] tensors), `seed` and optionally `env_ids` / `seeds` as for `plan`.  Every batch first runs with
        `first_iters`; the queries that come back "no exact solution" (a few %: the ones that would have kept the whole
        launch waiting for their 2000 iterations) run again with `max_iters` -- pooled over batches until `retry_min` of them
        wait -- on other streams, next to the following batches' first launches.  A query's outcome depends on its endpoints and sample stream only and the budget merely
        ends the loop, so the second run continues where the first stopped (`resume`: from its trees and counters; False: it retraces
        the first iterations): each batch's (path, path_len, status, n_checks)
        are those of `plan(..., max_iters=max_iters)`, bit for bit.  Returns the list of those tuples (after all launches
        have finished).  One host read-back per batch (which queries go again).  The defaults of `first_iters` / `retry_min` are
        the ones that measured best over long streams of 4096-query batches on Push (tools/ladder_grid.py: 100 / 1024; they only
        schedule the work)."""
        torch = _torch()
        if first_iters <= 0 or first_iters >= max_iters:
            return [self.plan(b["start"], b["goal"], max_iters=max_iters, max_nodes=max_nodes, max_path=max_path, seed=b.get("seed", 0),
                              env_ids=b.get("env_ids"), seeds=b.get("seeds")) for b in batches]
        dev = batches[0]["start"].device
        main = torch.cuda.current_stream(dev)
        sa = first_stream if first_stream is not None else torch.cuda.Stream(device=dev)
        sbs = list(retry_streams) if retry_streams else [torch.cuda.Stream(device=dev) for _ in range(2)]
        sa.wait_stream(main)
        for st in sbs:
            st.wait_stream(main)
        out, pend, wait = [], [], []         # wait: unsolved queries of finished first launches, pooled into retry launches
        n_wait, n_retry = 0, 0

        def retry():
            nonlocal n_wait, n_retry, wait
            sb = sbs[n_retry % len(sbs)]
            n_retry += 1
            sb.wait_stream(sa)
            with torch.cuda.stream(sb):
                cat = lambda k: torch.cat([w[k] for w in wait]).contiguous()
                r2 = self.plan(cat("start"), cat("goal"), max_iters=max_iters, max_nodes=max_nodes, max_path=max_path, seed=0,
                               env_ids=cat("ids"), seeds=cat("seeds"), stream=sb, max_workgroups=0 if retry_exclusive else -1,
                               exclusive=retry_exclusive, resume=PlanState.cat([w["state"] for w in wait]) if resume else None)
            # the pooled slices were allocated on `sa` and are read by the cat on `sb`: tell the caching allocator, or the
            # next first launch on `sa` may be handed their blocks while `sb` still waits behind an earlier retry
            for w in wait:
                for k in ("start", "goal", "ids", "seeds", "rows"):
                    w[k].record_stream(sb)
                if resume:
                    for t in (w["state"].tree_q, w["state"].tree_p, w["state"].state):
                        t.record_stream(sb)
            pend.append(([(w["batch"], w["rows"]) for w in wait], r2, sb))
            wait, n_wait = [], 0

        for i, b in enumerate(batches):
            E = b["start"].shape[0]
            ids = b.get("env_ids")
            if ids is None:
                ids = torch.arange(E, device=dev, dtype=torch.int64)
            with torch.cuda.stream(sa):
                res = self.plan(b["start"], b["goal"], max_iters=first_iters, max_nodes=max_nodes, max_path=max_path, seed=b.get("seed", 0),
                                env_ids=ids, seeds=b.get("seeds"), stream=sa, max_workgroups=max_workgroups_first, keep_state=resume)
                kept = res[4] if resume else None
                res = res[:4]
                again = torch.nonzero(res[2] == _lib.PLAN_NO_EXACT).flatten()       # (waits for this launch: the one read-back)
                if len(again):
                    seeds = (b["seeds"][again] if b.get("seeds") is not None
                             else torch.full((len(again),), int(b.get("seed", 0)), dtype=torch.int64, device=dev))
                    wait.append(dict(start=b["start"][again], goal=b["goal"][again], ids=ids[again], seeds=seeds, batch=i, rows=again,
                                     state=kept.rows(again) if resume else None))
                    n_wait += len(again)
            out.append(list(res))
            if n_wait >= retry_min:
                retry()
        if n_wait:
            retry()
        for parts, r2, sb in pend:
            with torch.cuda.stream(sb):
                o = 0
                for bi, rows in parts:
                    for k in range(4):
                        out[bi][k][rows] = r2[k][o:o + len(rows)]
                    o += len(rows)
            main.wait_stream(sb)
        main.wait_stream(sa)
        for o in out:              # results live in blocks of `sa`'s pool (patched on the retry streams), consumed on `main`
            for t in o:
                t.record_stream(main)
                for sb in sbs:
                    t.record_stream(sb)
        return [tuple(o) for o in out]

    def pullback(self, cur, target, step_size: float, num_trials: int, stream=None):
        """The rollout's invalid-target back-off (rl/mopa_rollouts.py:133-143) for E envs in one launch.
        cur / target: [E, nq] float64 GPU tensors.  Returns (target' [E, nq], n_trials [E] int32, valid [E] uint8)."""
        torch = _torch()
        _check_f64(cur, "cur", self.nq)
        _check_f64(target, "target", self.nq)
        E = cur.shape[0]
        out = target.clone()
        trials = torch.zeros(E, dtype=torch.int32, device=cur.device)
        valid = torch.zeros(E, dtype=torch.uint8, device=cur.device)
        _lib.check(_lib.lib().mopa_pullback_batch(self.scene.handle, _ptr(cur), _ptr(out), E, float(step_size), int(num_trials),
                                                  _ptr(trials), _ptr(valid), _stream_handle(stream)))
        return out, trials, valid


def postprocess_paths(path, path_len, status, cur, n_arm: int, ac_scale: float, interpolate: bool, limits, is_valid, stream=None,
                      seam_mask: int = 0):
    """Planner rows -> executable trajectories on the device (C ABI `mopa_paths_*`): un-wrap by successive differences
    (reference motion_planners/sampling_based_planner.py:71-99; `path` [M, max_path, nq] is overwritten with the un-wrapped
    rows), then -- `interpolate` -- the reference's densification of steps longer than ac_scale (rl/sac_agent.py:205-233)
    with every interior state validated by `is_valid(rows [S, nq]) -> uint8/bool [S]`.

    limits: agent_planning.JointLimits (float32 state limits + margin).  Returns (traj [M, L, nq], length [M] int64 -- 0 for
    queries with status != 0 --, needs_fallback [M] bool: a long step of that query has an invalid interior state, the
    reference plans such a step with its fallback planners and the caller has to).  One small read-back (totals).
    seam_mask: bit c set = qpos coordinate c belongs to an unlimited joint (`non_limited_idx`): its steps across +-3.14 are taken
    the short way round (sampling_based_planner.py:79-97)."""
    torch = _torch()
    L = _lib.lib()
    M, max_path, nq = path.shape
    dev = path.device
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _stream_handle(stream)
    seg = torch.empty(M, max_path, dtype=torch.int32, device=dev)
    n_walk = torch.empty(M, dtype=torch.int32, device=dev)
    out_len = torch.empty(M, dtype=torch.int32, device=dev)
    lim = [t.contiguous() for t in (limits.lo_state, limits.hi_state, limits.lo_shrunk, limits.hi_shrunk)]
    _lib.check(L.mopa_paths_unwrap_seam_batch(di, M, nq, int(n_arm), _ptr(path), max_path, _ptr(path_len), _ptr(status), _ptr(cur),
                                              float(ac_scale), int(bool(interpolate)), *[_ptr(t) for t in lim], _ptr(seg), _ptr(n_walk),
                                              _ptr(out_len), int(seam_mask), st))
    walk_off = torch.cumsum(n_walk.to(torch.int64), 0) - n_walk.to(torch.int64)
    tot_walk, rows = (int(x) for x in torch.stack([n_walk.sum(), out_len.max()]).cpu())
    rows = max(rows, 1)
    walk = walk_valid = None
    if tot_walk > 0:
        walk = torch.empty(tot_walk, nq, dtype=torch.float64, device=dev)
        _lib.check(L.mopa_paths_walk_batch(di, M, nq, int(n_arm), _ptr(path), max_path, _ptr(path_len), _ptr(out_len), float(ac_scale),
                                           *[_ptr(t) for t in lim], _ptr(seg), _ptr(walk_off), _ptr(walk), st))
        walk_valid = is_valid(walk).to(torch.uint8).contiguous()
    out = torch.zeros(M, rows, nq, dtype=torch.float64, device=dev)
    need = torch.zeros(M, dtype=torch.uint8, device=dev)
    _lib.check(L.mopa_paths_assemble_batch(di, M, nq, _ptr(path), max_path, _ptr(path_len), _ptr(out_len), _ptr(seg), _ptr(walk_off),
                                           _ptr(walk) if walk is not None else None, _ptr(walk_valid) if walk_valid is not None else None,
                                           _ptr(out), rows, _ptr(need), st))
    return out, out_len.to(torch.int64), need.bool()
