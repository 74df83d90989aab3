"""Census of the asynchronous rollout (bench policy, 4096 envs): envs busy / pooled / in planner launches, envs stepping per call, how long the busy
envs have been waiting.   python tools/rollout_census.py [env-name]"""
import sys, os; sys.path.insert(0, ".")
import numpy as np, torch
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
name = sys.argv[1] if len(sys.argv) > 1 else "PusherObstacle-v0"
E = 4096
env = make_env(name, E, seed=21); env.reset()
ro = BatchMoPARollout(env, RolloutConfig.for_env(name, async_planner=True))
print("main_iters", ro.main_iters, "simple_iters", ro.simple_iters, "first_iters", ro.cfg.planner_first_iters)
torch.manual_seed(8)
nn = torch.nn
actor = nn.Sequential(nn.Linear(env.obs.shape[1], 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 2 * ro.ac_dim)).to(env.device)
g = torch.Generator(device=env.device); g.manual_seed(8)
def act():
    with torch.no_grad():
        mu, ls = actor(env.obs.float()).chunk(2, dim=1)
        eps = torch.randn(E, ro.ac_dim, generator=g, dtype=torch.float32, device=env.device)
        return torch.tanh(mu + torch.exp(ls.clamp(-10, 2)) * eps).double()
rows = []
for t in range(260):
    out = ro.agent_step(act()); env.reset(out["done"].bool() & out["stepped"])
    if t >= 60:
        jobs = ro._jobs
        rows.append((int(ro.busy.sum()), int(ro._pool_mask.sum()), int(ro._retry_mask.sum()), len(jobs), int(out["stepped"].sum()), int((out["stepped"] & out["is_planner"]).sum())))
a = np.array(rows, dtype=float)
print("mean over 200 calls: busy %.0f, in pool %.0f, in retry pool %.0f, launches in flight %.1f, stepped %.0f (planner steps %.0f)" % tuple(a.mean(0)))
wait = (ro._t - ro._wait_since[ro.busy]).float()
print("waiting time of the busy envs now (calls): mean %.1f median %.1f p90 %.1f max %.0f" % (wait.mean().item(), wait.median().item(), wait.quantile(0.9).item(), wait.max().item()))
for j in ro._jobs[:6]: print({k: (v if isinstance(v, (int, float, str, bool)) else type(v).__name__) for k, v in j.items() if k in ("n", "phase", "iters", "stage", "keep", "chain")}, "ids" , int(j["ids"].numel()) if "ids" in j else None)
