"""K3 on the budget-exhausting queries ALONE (the queries that set a launch's latency): one full launch of the bench's 4096
queries to find them, then `reps` launches of just those -- the LAST `reps` k_rrt_connect dispatches of a kernel trace / PMC
pass of this script are the ones to read (tools/profile_k3.sh).   python tools/plan_fail_only.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pi = planner_inputs(bench.ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc)
dev = torch.device("cuda:0")
start, goal = bench.planner_queries(torch, bp, pi, 4096, dev)
path, plen, st, nchk = bp.plan(start, goal, max_iters=2000, max_nodes=4096, max_path=256, seed=7)
fi = torch.nonzero(st != 0).flatten().contiguous()
s2, g2 = start[fi].contiguous(), goal[fi].contiguous()
torch.cuda.synchronize()
for r in range(reps):
    t0 = time.perf_counter()
    bp.plan(s2, g2, max_iters=2000, max_nodes=4096, max_path=256, seed=7, env_ids=fi)
    torch.cuda.synchronize()
    print(f"failing queries alone ({len(fi)} of 4096): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
# how much of a budget-exhausting query's latency is contention with the other waves (instruction cache, LDS, ...)?
for k in (1, 2, 8, 32):
    fk = fi[:k].contiguous()
    s3, g3 = start[fk].contiguous(), goal[fk].contiguous()
    bp.plan(s3, g3, max_iters=2000, max_nodes=4096, max_path=256, seed=7, env_ids=fk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bp.plan(s3, g3, max_iters=2000, max_nodes=4096, max_path=256, seed=7, env_ids=fk)
    torch.cuda.synchronize()
    print(f"{k} failing quer{'y' if k == 1 else 'ies'} alone: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
