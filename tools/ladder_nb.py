#!/usr/bin/env python3
"""A few ladder settings at several stream lengths (is a setting's rate a property of the setting or of how the last retry launch
happens to land?).  GPU box."""
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
def run(nb, first, rmin, nret):
    batches = [dict(start=start, goal=goal, seed=7 + 13 * i) for i in range(nb)]
    kw = dict(first_iters=first, first_stream=streams[0], retry_streams=streams[1:1 + nret], retry_min=rmin, **prm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bp.plan_laddered(batches, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(8, 200, 512, 2)
for first, rmin, nret in ((200, 512, 2), (120, 1024, 3), (120, 1024, 4), (100, 1024, 2), (150, 1024, 4), (120, 768, 4), (130, 1024, 4), (110, 1024, 4)):
    row = []
    for nb in (12, 16, 24, 32, 48, 64):
        run(nb, first, rmin, nret)
        row.append(min(run(nb, first, rmin, nret) for _ in range(2)))
    print(f"first {first:4d} retry_min {rmin:5d} streams {nret}: " + " ".join(f"nb{nb}:{nb * E / t / 1e3:6.1f}" for nb, t in zip((12, 16, 24, 32, 48, 64), row)) + " k plans/s", flush=True)
