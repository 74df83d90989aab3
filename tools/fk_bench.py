"""Debug tool (-DMOPA_PLAN_STATS build: make -C mopa_rl_amd/csrc libmopa_hip_stats.so): the pose phases of one planner validity
pass -- state fill, joint table, chain walk -- alone on one lone wave, per pass of 4 states.
    MOPA_HIP_LIB=mopa_rl_amd/csrc/libmopa_hip_stats.so python tools/fk_bench.py [env]"""
import ctypes, sys
sys.path.insert(0, ".")
import numpy as np, torch
from mopa_rl_amd import _lib
from mopa_rl_amd.scene import planner_inputs, default_qpos
env = sys.argv[1] if len(sys.argv) > 1 else "SawyerPushObstacle-v0"
pi = planner_inputs(env)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
lib = _lib.lib()
dev = torch.device("cuda:0")
row = torch.tensor(default_qpos(env, pi.model), dtype=torch.float64, device=dev)
rng = np.random.default_rng(0)
qs = torch.tensor(rng.uniform(pi.jnt_minimum, pi.jnt_maximum, size=(4, len(pi.jnt_minimum))), dtype=torch.float64, device=dev).contiguous()
sink = torch.zeros(4 * 64 * 16, dtype=torch.float64, device=dev)
t = (ctypes.c_ulonglong * 3)()
iters = 2000
lib.mopa_debug_fk_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for ns in (1, 4):
    for _ in range(2):
        _lib.check(lib.mopa_debug_fk_bench(sc._h, row.data_ptr(), qs.data_ptr(), ns, iters, t, sink.data_ptr()))
    print(f"{env} ns={ns}: per pass  state fill {t[0] / iters * 10:.0f} ns  joint table {t[1] / iters * 10:.0f} ns  chain walk {t[2] / iters * 10:.0f} ns")
