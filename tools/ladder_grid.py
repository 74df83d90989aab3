#!/usr/bin/env python3
"""Finer grid of the planner ladder's scheduling knobs (first rung's budget, retry pool size, retry streams) on the bench's
queries, 24 batches as in bench.py; three repeats each.  Results are those of full-budget launches whatever the knobs.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
E = 4096
dev = torch.device("cuda", 0)
pi = planner_inputs(bench.ENV)
bp = BatchPlanner(_lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range))
start, goal = bench.planner_queries(torch, bp, pi, E, dev)
prm = dict(max_iters=2000, max_nodes=4096, max_path=256)
streams = [torch.cuda.Stream(device=dev) for _ in range(7)]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 24
batches = [dict(start=start, goal=goal, seed=7 + 13 * i) for i in range(nb)]
def run(first, rmin, nret):
    kw = dict(first_iters=first, first_stream=streams[0], retry_streams=streams[1:1 + nret], retry_min=rmin, **prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bp.plan_laddered(batches, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(200, 512, 2)
for first in (60, 80, 100, 120, 150, 200):
    for rmin in (512, 1024, 2048):
        for nret in (2, 3, 4, 6):
            run(first, rmin, nret)
            ts = [run(first, rmin, nret) for _ in range(3)]
            print(f"first {first:4d} retry_min {rmin:5d} streams {nret}: " + " ".join(f"{nb * E / t / 1e3:6.1f}" for t in ts) + " k plans/s", flush=True)
