import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_rollout as T
from mopa_rl_amd.agent_planning import action_to_displacement
G = np.load(os.path.join(T.GOLD, "ref_py_rollout_push.npz"))
E = G["ac"].shape[0]
env, ro = T._make(G, E)
T._load_state(env, G["qpos_start"][:, 0], G["ep_len_start"][:, 0])
ac = torch.tensor(G["ac"][:, 0], device="cuda")
cur = env.qpos.clone()
disp = action_to_displacement(ac[:, :7].contiguous(), 0.05, 0.7, 0.5)
tgt = cur.clone(); tgt[:, :7] += disp
tgt = ro.limits.clip_target(tgt)
ids = torch.arange(E, device="cuda")
ro.t = 0
traj, lens, success, interp, valid, exact = ro.plan(cur, tgt, ids)
traj, lens = traj.cpu().numpy(), lens.cpu().numpy()
out = ro.agent_step(ac)
q_gpu = env.qpos.cpu().numpy()
for e in (5, 8, 0):
    q = G["qpos_start"][e, 0].copy(); prev = None
    for k in range(lens[e]):
        a = traj[e, k, :7] - q[:7]
        if prev is None: prev = q[:7].copy()
        des = prev + np.clip(a, -0.05, 0.05)
        q[:7] = des; prev = des.copy()
    print("env", e, "len", lens[e], "numpy-exec vs fixture", np.abs(q - G["qpos_end"][e, 0]).max(), "numpy-exec vs gpu", np.abs(q - q_gpu[e]).max(),
          "last wp == tgt", np.abs(traj[e, lens[e]-1] - tgt[e].cpu().numpy()).max())
