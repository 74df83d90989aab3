import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np, bench, time
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
dev = torch.device("cuda", 0); E = 4096
pi = planner_inputs(bench.ENV)
bp = BatchPlanner(_lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range))
start, goal = bench.planner_queries(torch, bp, pi, E, dev)
prm = dict(max_nodes=4096, max_path=256, seed=7)
full = bp.plan(start, goal, max_iters=2000, **prm)
p1 = bp.plan(start, goal, max_iters=200, keep_state=True, **prm)
again = torch.nonzero(p1[2] == _lib.PLAN_NO_EXACT).flatten()
ids = again.contiguous()
t0 = time.perf_counter()
p2 = bp.plan(start[again].contiguous(), goal[again].contiguous(), max_iters=2000, env_ids=ids, resume=p1[4].rows(again), **prm)
torch.cuda.synchronize(); print("resume launch of", len(again), "queries: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
out = [t.clone() for t in p1[:4]]
for k in range(4): out[k][again] = p2[k]
ok = True
for k, nm in ((1, "path_len"), (2, "status"), (3, "n_checks")):
    same = bool((out[k] == full[k]).all()); ok &= same; print(nm, "identical:", same)
pl = full[1].cpu().numpy(); a = out[0].cpu().numpy(); b = full[0].cpu().numpy()
same = all(np.array_equal(a[e, :pl[e]].view(np.uint64), b[e, :pl[e]].view(np.uint64)) for e in range(E)); print("paths identical:", same)
