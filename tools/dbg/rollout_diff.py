import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_rollout as T
G = np.load(os.path.join(T.GOLD, "ref_py_rollout_push.npz"))
E, Tn = G["ac"].shape[:2]
env, ro = T._make(G, E)
from mopa_rl_amd.rollout import COUNTERS
for t in range(Tn):
    T._load_state(env, G["qpos_start"][:, t], G["ep_len_start"][:, t])
    before = {k: ro.counters[k].clone() for k in COUNTERS}
    ro.t = t
    out = ro.agent_step(torch.tensor(G["ac"][:, t], device=env.device))
    q = env.qpos.cpu().numpy()
    d = np.abs(q - G["qpos_end"][:, t])
    bad = np.where(d.max(axis=1) > 0)[0]
    got_c = np.stack([(ro.counters[k] - before[k]).cpu().numpy() for k in COUNTERS], axis=1)
    for e in bad:
        print("t", t, "env", e, "maxdiff", d[e].max(), "cols", np.where(d[e] > 0)[0], "counters got", got_c[e], "want", G["counters"][e, t],
              "intra", int(out["intra_steps"][e]), G["intra"][e, t], "done", int(out["done"][e]), G["done"][e, t])
    print("t", t, "nbad", len(bad), "rew maxdiff", np.abs(out["rew"].cpu().numpy() - G["rew"][:, t]).max())
