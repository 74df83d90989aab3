"""Where the envs of an asynchronous rollout wait (diagnostic; its read-backs slow the calls a little): per call, how many envs
sit in the pool, in first-phase launches, in the retry pool, in retry launches; per finished job its size and its life in calls / ms."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig

E, calls = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
env = make_env(bench.ENV, E, device=dev, seed=21, max_episode_steps=250); env.reset()
over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("MOPA_BENCH_ROLLOUT", "").split(",") if kv)}
ro = BatchMoPARollout(env, RolloutConfig(async_planner=True, **over))
torch.manual_seed(8)
nn = torch.nn
ad = ro.ac_dim      # the bench's policy: a random-init SAC actor, tanh-Gaussian samples
actor = nn.Sequential(nn.Linear(env.obs.shape[1], 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 2 * ad)).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(8)
born = {}
rows = []
t_start = time.perf_counter()
for t in range(calls):
    with torch.no_grad():
        mu, log_std = actor(env.obs.float()).chunk(2, dim=1)
        eps = torch.randn(E, ad, generator=gen, dtype=torch.float32, device=dev)
        ac = torch.tanh(mu + torch.exp(log_std.clamp(-10.0, 2.0)) * eps).double()
    out = ro.agent_step(ac)
    d = out["done"].bool() & out["stepped"]
    if bool(d.any()): env.reset(d)
    now = time.perf_counter()
    ids = set(id(j) for j in ro._jobs)
    for j in ro._jobs:
        if id(j) not in born: born[id(j)] = (t, now, len(j["ids"]), bool(j.get("retry")), )
    for k in [k for k in born if k not in ids]:
        t0, w0, n, r = born.pop(k); rows.append((n, r, t - t0, (now - w0) * 1e3))
    if t >= 50 and t % 25 == 0:
        in_first = sum(len(j["ids"]) for j in ro._jobs if not j.get("retry")); in_retry = sum(len(j["ids"]) for j in ro._jobs if j.get("retry"))
        print(f"call {t}: busy {int(ro.busy.sum())} = pool {int(ro._pool_mask.sum())} + first-phase jobs {in_first} ({sum(1 for j in ro._jobs if not j.get('retry'))}) + retry pool {int(ro._retry_mask.sum())} + retry jobs {in_retry} ({sum(1 for j in ro._jobs if j.get('retry'))}); stepped {int(out['stepped'].sum())}; stages {[j['stage'] for j in ro._jobs]}", flush=True)
print("ms per call %.2f" % ((time.perf_counter() - t_start) / calls * 1e3))
r = np.array([(a, b, c, d) for a, b, c, d in rows if True], dtype=float)
for kind in (0, 1):
    m = r[:, 1] == kind
    if m.any(): print(("retry" if kind else "first-phase"), "jobs", int(m.sum()), "mean size %.0f" % r[m, 0].mean(), "life: calls %.1f  ms %.1f (p90 %.1f)" % (r[m, 2].mean(), r[m, 3].mean(), np.percentile(r[m, 3], 90)))
