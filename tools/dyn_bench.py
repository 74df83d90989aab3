"""Time env.step with the servo dynamics (K6) next to the kinematic step: `python tools/dyn_bench.py [E]`."""
import sys
import time
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mopa_rl_amd.kinematic_env import make_env

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
only = sys.argv[2] if len(sys.argv) > 2 else ""      # "push": the dynamics step of Push only (profiling)
for name in (["SawyerPushObstacle-v0"] if only in ("push", "contacts") else ["SawyerPushObstacle-v0", "SawyerLiftObstacle-v0", "SawyerAssemblyObstacle-v0"]):
    for dyn in ((("contacts",) if only == "contacts" else (True,)) if only else (False, True, "contacts")):
        env = make_env(name, E, dynamics=bool(dyn), contacts=(dyn == "contacts"))
        env.reset()
        n = 20
        # fresh uniform actions every step (a random walk about the reset pose, as bench.py's env section does);
        # MOPA_DYN_BENCH_ACT=zero: the arm holds still; =const: one action repeated (the arm runs into its joint stops)
        kind = os.environ.get("MOPA_DYN_BENCH_ACT", "walk")
        acts = (torch.rand(n + 3, E, env.action_dim, dtype=torch.float64, device=env.device) * 2 - 1).contiguous()
        if kind == "zero":
            acts.zero_()
        elif kind == "const":
            acts[:] = acts[0]
        for k in range(3):
            env.step(acts[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            env.step(acts[3 + k])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:28s} E={E} dynamics={dyn!s:5s} {dt * 1e3:8.3f} ms/step  {E / dt / 1e6:8.3f} M env-steps/s"
              + (f"  ({E * 75 / dt / 1e6:.1f} M sub-steps/s)" if dyn else ""), flush=True)
        env.close()
