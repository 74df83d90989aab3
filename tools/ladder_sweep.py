"""Throughput of BatchPlanner.plan_laddered on the bench's planner queries for several first-launch budgets (GPU box)."""
import sys, time; sys.path.insert(0, ".")
import torch
import bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
pi = planner_inputs(bench.ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc); dev = torch.device("cuda:0")
E, nb = 4096, 8
start, goal = bench.planner_queries(torch, bp, pi, E, dev)
batches = [dict(start=start, goal=goal, seed=7 + 13 * i) for i in range(nb)]
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
for first, rmin, nretry in ((300, 512, 2), (150, 512, 2), (200, 512, 2), (500, 512, 2), (300, 256, 3), (300, 1024, 2), (200, 768, 3)):
    kw = dict(max_iters=2000, first_iters=first, max_nodes=4096, max_path=256, first_stream=streams[0], retry_streams=streams[1:1 + nretry], retry_min=rmin)
    bp.plan_laddered(batches[:2], **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); bp.plan_laddered(batches, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"first {first:4d} retry_min {rmin:5d} retry streams {nretry}: {dt * 1e3 / nb:6.2f} ms per batch, {nb * E / dt / 1e3:6.1f} k plans/s", flush=True)
