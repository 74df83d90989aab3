"""Throughput of mopa_check_motion_batch (K2) on planner-like segments (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ENV, make_inputs
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs

pi = planner_inputs(ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc)
dev = torch.device("cuda:0")
E, S = 4096, 64
qa, rows = make_inputs(torch, pi, E, S, 5, dev, mode="near")
g = torch.Generator(device=dev); g.manual_seed(2)
lo = torch.tensor(pi.jnt_minimum, dtype=torch.float64, device=dev); hi = torch.tensor(pi.jnt_maximum, dtype=torch.float64, device=dev)
for step in (0.1, 0.5):
    d = torch.randn(E * S, 7, generator=g, dtype=torch.float64, device=dev)
    d = d / d.abs().sum(dim=1, keepdim=True) * step          # L1 length = step (range 0.1 = one RRT extension)
    qb = torch.minimum(torch.maximum(qa + d, lo), hi).contiguous()
    for _ in range(2): v = bp.check_motion(qa, qb, rows, samples_per_env=S)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 5
    for _ in range(R): v = bp.check_motion(qa, qb, rows, samples_per_env=S)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
    print(f"L1 step {step}: {E*S/dt/1e6:.2f} M motions/s  ({dt*1e3:.2f} ms per {E*S} segments), valid {v.float().mean().item():.3f}")
