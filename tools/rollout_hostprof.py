"""cProfile of the host side of the bench's asynchronous rollout call (SAC actor + agent_step + exchange pack + reset): where a
call's Python / dispatch time goes.  GPU box.  python tools/rollout_hostprof.py [E]"""
import sys, cProfile, pstats, time; sys.path.insert(0, ".")
import torch
from mopa_rl_amd.kinematic_env import make_env
from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
from mopa_rl_amd.dist import TransitionExchange
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
env = make_env("SawyerPushObstacle-v0", E, seed=5, max_episode_steps=250); env.reset()
ro = BatchMoPARollout(env, RolloutConfig(async_planner=True))
nn = torch.nn
ad = ro.ac_dim
torch.manual_seed(8)
actor = nn.Sequential(nn.Linear(env.obs.shape[1], 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 2 * ad)).to(dev)
g = torch.Generator(device=dev); g.manual_seed(8)
tx = TransitionExchange(E, env.obs_dim, ad, dev)
def one(k):
    with torch.no_grad():
        mu, log_std = actor(env.obs.float()).chunk(2, dim=1)
        eps = torch.randn(E, ad, generator=g, dtype=torch.float32, device=dev)
        a = torch.tanh(mu + torch.exp(log_std.clamp(-10.0, 2.0)) * eps).double()
    out = ro.agent_step(a)
    tx.pack(k, out["ob"], out["ac"], out["rew"], out["done"], out["intra_steps"], out["ob_next"], stepped=out["stepped"])
    tx.launch(k)
    env.reset(out["done"].bool() & out["stepped"])
for k in range(40): one(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(40, 240): one(k)
torch.cuda.synchronize()
print(f"unprofiled: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per call")
# host-only time: the same calls with the device idle in between would need a sync per call; instead time the submission alone
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for k in range(240, 340): one(k)
pr.disable()
print(f"profiled: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per call (100 calls)")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
st.sort_stats("cumulative").print_stats(22)
