#!/usr/bin/env python3
"""Planner ladder (BatchPlanner.plan_laddered) over a stream of batches: how the rate depends on the stream's length (the
last retry launch drains alone for one straggler's latency), the first rung's budget and the retry pool.
    python tools/ladder_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from mopa_rl_amd import _lib  # noqa: E402
from mopa_rl_amd.batch import BatchPlanner  # noqa: E402
from mopa_rl_amd.scene import planner_inputs  # noqa: E402

E = 4096
dev = torch.device("cuda", 0)
pi = planner_inputs(bench.ENV)
bp = BatchPlanner(_lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range))
start, goal = bench.planner_queries(torch, bp, pi, E, dev)
prm = dict(max_iters=2000, max_nodes=4096, max_path=256)
streams = [torch.cuda.Stream(device=dev) for _ in range(6)]


def run(nb, first, rmin, nret):
    batches = [dict(start=start, goal=goal, seed=7 + 13 * i) for i in range(nb)]
    kw = dict(first_iters=first, first_stream=streams[0], retry_streams=streams[1:1 + nret], retry_min=rmin, **prm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bp.plan_laddered(batches, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run(2, 200, 512, 2)
QUICK = os.environ.get("MOPA_LADDER_QUICK", "0") != "0"
if QUICK:       # one full launch, then the default ladder over a long stream
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bp.plan(start, goal, seed=7, **prm)
    torch.cuda.synchronize()
    print(f"one full launch: {(time.perf_counter()-t0)*1e3:.2f} ms per 4096 queries")
    for wg in (256, 512):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bp.plan(start, goal, seed=7, max_workgroups=wg, **prm)
        torch.cuda.synchronize()
        print(f"one full launch, at most {wg} workgroups: {(time.perf_counter()-t0)*1e3:.2f} ms per 4096 queries")
for nb in ((32,) if QUICK else (8, 16, 32)):
    for first in ((100, 200, 400) if os.environ.get("MOPA_LADDER_FIRST") else (200,) if QUICK else (100, 200, 400)):
        for rmin, nret in (((512, 2), (512, 4), (1024, 4)) if QUICK else ((512, 2), (256, 3), (1024, 2))):
            dt = run(nb, first, rmin, nret)
            print(f"batches {nb:3d} first_iters {first:4d} retry_min {rmin:5d} retry_streams {nret}: {dt*1e3:8.1f} ms  {dt*1e3/nb:6.2f} ms/batch  {nb*E/dt/1e3:7.1f} k plans/s", flush=True)
# the first rung alone
for first in (100, 200, 400):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(4):
        bp.plan(start, goal, max_iters=first, max_nodes=4096, max_path=256, seed=7 + 13 * i)
    torch.cuda.synchronize()
    print(f"first rung alone, {first} iterations: {(time.perf_counter()-t0)/4*1e3:.2f} ms per 4096 queries")
