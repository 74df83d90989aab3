"""Time mopa_is_valid_batch for small batches under both K1 generations (run on the GPU box):
    for k in v1 v2; do MOPA_VALID_KERNEL=$k python tools/crossover.py; done"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import ENV, make_inputs
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs

pi = planner_inputs(ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
bp = BatchPlanner(sc)
dev = torch.device("cuda:0")
print("kernel", os.environ.get("MOPA_VALID_KERNEL", "auto"))
for E in (1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536, 262144):
    qa, rows = make_inputs(torch, pi, E, 1, 3, dev, mode="near")
    out = torch.empty(E, dtype=torch.uint8, device=dev)
    for _ in range(3):
        bp.is_valid(qa, rows, samples_per_env=1, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        bp.is_valid(qa, rows, samples_per_env=1, out=out)
    torch.cuda.synchronize()
    print(f"N={E:7d}  {1e6 * (time.perf_counter() - t0) / R:9.1f} us   valid {out.float().mean().item():.3f}")
