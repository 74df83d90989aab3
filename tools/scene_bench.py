import sys, time; sys.path.insert(0, ".")
import torch, numpy as np
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs, default_qpos
dev = torch.device("cuda:0")
for env in ["SawyerPushObstacle-v0","SawyerAssemblyObstacle-v0","PusherObstacle-v0","SawyerLiftObstacle-v0"]:
    pi = planner_inputs(env)
    sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    bp = BatchPlanner(sc)
    E, S = 4096, 256
    g = torch.Generator(device=dev); g.manual_seed(1)
    na = sc.na
    lo = torch.tensor(pi.jnt_minimum, dtype=torch.float64, device=dev); hi = torch.tensor(pi.jnt_maximum, dtype=torch.float64, device=dev)
    q0 = torch.tensor(default_qpos(env, pi.model), dtype=torch.float64, device=dev)
    qa = lo + (hi - lo) * torch.rand(E, S, na, generator=g, dtype=torch.float64, device=dev)
    near = torch.minimum(torch.maximum(q0[pi.ref_joint_pos_indexes] + 0.3 * torch.randn(E, S, na, generator=g, dtype=torch.float64, device=dev), lo), hi)
    qa[:, S // 2:] = near[:, S // 2:]
    qa = qa.reshape(E * S, na).contiguous(); rows = q0.repeat(E, 1).contiguous()
    out = torch.empty(E * S, dtype=torch.uint8, device=dev)
    for _ in range(3): bp.is_valid(qa, rows, samples_per_env=S, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): bp.is_valid(qa, rows, samples_per_env=S, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{env:28s} {E*S/dt/1e6:8.1f} M checks/s   valid {out.float().mean().item():.3f}")
