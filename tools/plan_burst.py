import sys, time, os; sys.path.insert(0, "/root/repo")
import torch, bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
from mopa_rl_amd.rollout import _side_streams
dev = torch.device("cuda", 0); E = 4096
pi = planner_inputs(bench.ENV)
bp = BatchPlanner(_lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range))
start, goal = bench.planner_queries(torch, bp, pi, E, dev)
prm = dict(max_iters=2000, max_nodes=4096, max_path=256, seed=7)
bp.plan(start, goal, **prm); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); bp.plan(start, goal, **prm); torch.cuda.synchronize(); print("single default %.2f ms" % ((time.perf_counter() - t0) * 1e3))
streams = _side_streams(dev, 2) + [torch.cuda.Stream(device=dev), torch.cuda.current_stream()]
def burst(cap):
    for i, st in enumerate(streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            bp.plan(start, goal, **dict(prm, seed=7 + 13 * i), stream=st, max_workgroups=cap)
    for st in streams: torch.cuda.current_stream().wait_stream(st)
for cap in (64, 128):
    burst(cap); torch.cuda.synchronize()
    t0 = time.perf_counter(); burst(cap); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("burst of 4, cap %d: %.1f ms  %.1f k plans/s" % (cap, dt * 1e3, 4 * E / dt / 1e3))
