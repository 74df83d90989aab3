import sys, time; sys.path.insert(0, ".")
import torch
from bench import ENV, make_inputs
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
pi = planner_inputs(ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
bp = BatchPlanner(sc); dev = torch.device("cuda:0")
for E in (256, 1024):
    qa, rows = make_inputs(torch, pi, E, 1, 3, dev, mode="near")
    out = torch.empty(E, dtype=torch.uint8, device=dev)
    for _ in range(5): bp.is_valid(qa, rows, samples_per_env=1, out=out)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): bp.is_valid(qa, rows, samples_per_env=1, out=out)
    e.record(); torch.cuda.synchronize()
    print(f"N={E}: {s.elapsed_time(e)/20*1e3:.1f} us per launch")
