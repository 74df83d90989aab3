"""K7 (env.step with contacts, stage C) timing probe: ms per env.step of E envs, contacts / solver sweeps per sub-step.
   python tools/ct_bench.py [E] [steps] [maxcon]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from mopa_rl_amd import _lib
from mopa_rl_amd.kinematic_env import make_env

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
maxcon = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
for env_name in [n for n in ("SawyerPushObstacle-v0", "SawyerLiftObstacle-v0", "SawyerAssemblyObstacle-v0") if os.environ.get("CT_ENVS", "") in n]:
    for scale in (1.0, 0.0):
        env = make_env(env_name, E, device=dev, seed=11, dynamics=True, contacts=True, max_episode_steps=1 << 30,
                       contact_options=dict({"maxcon": maxcon, "maxpair": min(8, maxcon)}, **json.loads(os.environ.get("CT_OPTS", "{}"))))
        g = torch.Generator(device=dev); g.manual_seed(3)
        acts = ((torch.rand(steps + 2, E, env.action_dim, generator=g, dtype=torch.float64, device=dev) * 2 - 1) * scale).contiguous()
        stats = torch.zeros(E, 4, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, stats.data_ptr()))
        env.reset()
        env.step(acts[0]); env.step(acts[1])
        torch.cuda.synchronize()
        tot = np.zeros(4)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if os.environ.get("CT_GC"):
            import gc
            gc.collect(); gc.disable()
        ev0.record()
        for t in range(steps):
            if os.environ.get("CT_TRACE"):
                torch.cuda.synchronize(); _t = time.perf_counter()
            env.step(acts[2 + t])
            s = stats.cpu().numpy().astype(np.float64)
            if os.environ.get("CT_TRACE"):
                sw = s[:, 1] / env.dyn.nsub
                print(f"    step {t}: {(time.perf_counter() - _t) * 1e3:7.2f} ms, iterations / sub-step max {sw.max():.2f} (env {int(sw.argmax())}), finite {bool(torch.isfinite(env.qpos).all())}", flush=True)
            tot[:3] += s[:, :3].sum(0); tot[3] = max(tot[3], s[:, 3].max())
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / steps
        nsub = env.dyn.nsub * E * steps
        print(f"{env_name:28s} action scale {scale}: {ms:7.3f} ms per env.step of {E} envs = {E / ms / 1e3:6.3f} M env-steps/s | "
              f"contacts/sub-step {tot[0] / nsub:5.2f} sweeps/sub-step {tot[1] / nsub:5.2f} dropped/sub-step {tot[2] / nsub:5.3f} max {int(tot[3])} "
              f"| lds {env.ct.maxcon} contacts", flush=True)
        L = _lib.lib()
        if hasattr(L, "mopa_debug_ct_prof"):
            import ctypes
            buf = (ctypes.c_ulonglong * 32)()
            L.mopa_debug_ct_prof(buf, 1)
            t = np.array(list(buf), dtype=np.float64)
            nw = (E + 3) // 4 * env.dyn.nsub * (steps + 2)
            names = ["m-rows+frames", "precull", "cull", "rows", "solver", "forces", "integrate", "narrow", "walk", "owner", "bias", "crb-acc", "nt:rows+H", "nt:ldl+solve", "nt:linesearch", "nt:forces"]
            print("    us per sub-step (lane 0 of each wave, cycles / 2400): " + "  ".join(f"{n} {t[i] / nw / 2400:.2f}" for i, n in enumerate(names)), flush=True)
            if t[20]:
                print(f"    cull trips per sub-step {t[20] / nw:.2f}; us per trip: list walk {t[16] / t[20] / 2400:.2f}  record load {t[17] / t[20] / 2400:.2f}  bound {t[18] / t[20] / 2400:.2f}  reverse bound {t[19] / t[20] / 2400:.2f}  between trips {t[25] / t[20] / 2400:.2f}", flush=True)
        env.close()
