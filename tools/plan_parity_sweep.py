"""Full-scale planner parity on the GPU box: the bench's 4096 RRT-Connect queries (2000 iterations, 4096-node trees)
through K3 and, one by one, through the CPU oracle with the same sample streams: status, path length, every waypoint's
bit pattern and the count of consumed validity checks must agree.  Test infrastructure only."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
import bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
from oracle import oracle as O
O.build()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pi = planner_inputs(bench.ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc)
orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
start, goal = bench.planner_queries(torch, bp, pi, E, torch.device("cuda:0"))
prm = dict(max_iters=2000, max_nodes=4096, max_path=256, seed=7)
path, plen, st, nchk = (x.cpu().numpy() for x in bp.plan(start, goal, **prm))
s_h, g_h = start.cpu().numpy(), goal.cpu().numpy()
bad = 0
t0 = time.time()
for e in range(E):
    ost, opath, ochk, _ = orc.plan(s_h[e], g_h[e], pi.spec.range, 0.005, prm["max_iters"], prm["max_nodes"], seed=prm["seed"], env_id=e,
                                   max_path=prm["max_path"])
    ok = st[e] == ost and plen[e] == len(opath) and nchk[e] == ochk and \
        np.array_equal(np.ascontiguousarray(path[e, :plen[e]]).view(np.uint64), np.ascontiguousarray(opath).view(np.uint64))
    if not ok:
        bad += 1
        print("MISMATCH env", e, st[e], ost, plen[e], len(opath), nchk[e], ochk, flush=True)
print(f"{E} queries: solved {(st == 0).sum()}, no exact solution {(st == -4).sum()}, invalid goal {(st == -5).sum()}; "
      f"consumed checks {int(nchk.sum())}; mismatches vs the oracle: {bad}  (oracle: {time.time() - t0:.1f} s on one core)")
