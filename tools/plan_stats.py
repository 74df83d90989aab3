import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
import bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs, default_qpos
pi = planner_inputs(bench.ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc); dev = torch.device("cuda:0")
E = 4096
start, goal = bench.planner_queries(torch, bp, pi, E, dev)
for iters in (2000, 500):
    bp.plan(start, goal, max_iters=iters, max_nodes=4096, max_path=256, seed=7); torch.cuda.synchronize()
    t0 = time.perf_counter(); path, plen, st, nchk = bp.plan(start, goal, max_iters=iters, max_nodes=4096, max_path=256, seed=7); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    f = st != 0
    print(f"iters {iters}: {dt*1e3:.1f} ms; fail {int(f.sum())}; checks fail mean {nchk[f].float().mean().item():.0f} max {nchk.max().item()}; ok mean {nchk[~f].float().mean().item():.0f} max {nchk[~f].max().item()}; us per check (slowest env) {dt*1e6/nchk.max().item():.1f}")
# only the failing envs
fi = torch.nonzero(st != 0).flatten()
s2, g2 = start[fi].contiguous(), goal[fi].contiguous()
bp.plan(s2, g2, max_iters=2000, max_nodes=4096, max_path=256, seed=7, env_ids=fi.contiguous()); torch.cuda.synchronize()
t0 = time.perf_counter(); r = bp.plan(s2, g2, max_iters=2000, max_nodes=4096, max_path=256, seed=7, env_ids=fi.contiguous()); torch.cuda.synchronize()
print(f"failing envs alone ({len(fi)}): {(time.perf_counter()-t0)*1e3:.1f} ms")
ok_i = torch.nonzero(st == 0).flatten()
s3, g3 = start[ok_i].contiguous(), goal[ok_i].contiguous()
t0 = time.perf_counter(); r = bp.plan(s3, g3, max_iters=2000, max_nodes=4096, max_path=256, seed=7, env_ids=ok_i.contiguous()); torch.cuda.synchronize()
print(f"succeeding envs alone ({len(ok_i)}): {(time.perf_counter()-t0)*1e3:.1f} ms")
