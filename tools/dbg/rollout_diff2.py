import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_rollout as T
from mopa_rl_amd.agent_planning import action_to_displacement, interpolation_steps
G = np.load(os.path.join(T.GOLD, "ref_py_rollout_push.npz"))
ac = torch.tensor(G["ac"][:, 0, :7]); cur = torch.tensor(G["qpos_start"][:, 0])
for dev in ("cpu", "cuda"):
    a, c = ac.to(dev), cur.to(dev)
    disp = action_to_displacement(a, 0.05, 0.7, 0.5)
    tgt = c.clone(); tgt[:, :7] += disp
    diff = tgt[:, :7] - c[:, :7]
    s, n = interpolation_steps(diff, 0.05)
    per = diff / s[:, None]
    globals()[dev] = dict(disp=disp.cpu().numpy(), tgt=tgt.cpu().numpy(), s=s.cpu().numpy(), per=per.cpu().numpy())
for k in cpu:
    d = np.abs(cpu[k] - cuda[k])
    print(k, "max diff cpu vs cuda", d.max(), np.argwhere(d > 0)[:10].tolist())
# numpy reference formula
acn = G["ac"][:, 0, :7]
ref = np.where(np.abs(acn) < 0.7, acn / (0.7 / 0.05), np.sign(acn) * (0.05 + (0.5 - 0.05) * ((np.abs(acn) - 0.7) / (1 - 0.7))))
print("numpy vs cpu torch", np.abs(ref - cpu["disp"]).max(), "numpy vs cuda", np.abs(ref - cuda["disp"]).max())
x = torch.rand(1000000, dtype=torch.float64) * 2 - 1; y = torch.rand(1000000, dtype=torch.float64) + 0.5
print("div mismatches", int(((x / y) != (x.cuda() / y.cuda()).cpu()).sum()), "mul", int(((x * y) != (x.cuda() * y.cuda()).cpu()).sum()))
print("div by scalar tensor", int(((x / torch.full_like(x, 14.0)) != (x.cuda() / torch.full_like(x.cuda(), 14.0)).cpu()).sum()))
