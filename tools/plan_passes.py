"""Debug tool: build the library with -DMOPA_PLAN_STATS first (make -C mopa_rl_amd/csrc EXTRA=-DMOPA_PLAN_STATS).
Prints, for the bench's planner queries, validity passes / evaluated states / tree sizes per failing and succeeding env."""
import sys; sys.path.insert(0, ".")
import torch, numpy as np
import bench
from mopa_rl_amd import _lib
from mopa_rl_amd.batch import BatchPlanner
from mopa_rl_amd.scene import planner_inputs
pi = planner_inputs(bench.ENV)
sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
bp = BatchPlanner(sc)
start, goal = bench.planner_queries(torch, bp, pi, 4096, torch.device("cuda:0"))
path, plen, st, nchk = bp.plan(start, goal, max_iters=2000, max_nodes=4096, max_path=256, seed=7)
n = nchk.cpu().numpy(); st = st.cpu().numpy()
chk, npass, nst, n0, n1 = n & 0xffff, (n >> 16) & 0x3fff, (n >> 30) & 0x3fff, (n >> 44) & 0x3ff, (n >> 54) & 0x3ff
for name, m in (("fail", st != 0), ("ok", st == 0)):
    print(name, int(m.sum()), "checks", chk[m].mean(), "passes", npass[m].mean(), "states", nst[m].mean(), "tree0", n0[m].mean(), "tree1", n1[m].mean())
f = np.where(st != 0)[0][:12]
for e in f: print(e, chk[e], npass[e], nst[e], n0[e], n1[e])

import ctypes
lib = _lib.lib()
if hasattr(lib, "mopa_debug_plan_times"):
    buf = (ctypes.c_ulonglong * 64)()
    lib.mopa_debug_plan_times(buf, 1)
    fi = torch.nonzero(torch.tensor(st != 0)).flatten().to(start.device)
    bp.plan(start[fi].contiguous(), goal[fi].contiguous(), max_iters=2000, max_nodes=4096, max_path=256, seed=7, env_ids=fi.contiguous())
    lib.mopa_debug_plan_times(buf, 1)
    t = list(buf); n = len(fi)
    print("failing envs only: per env, 100 MHz ticks -> us: pose+fk %.0f broad %.0f narrow %.0f | survivors/pass %.1f states/pass %.2f passes %d" % (
        t[0] / n / 100, t[1] / n / 100, t[2] / n / 100, t[3] / max(t[4], 1), t[5] / max(t[4], 1), t[4] / n))
    print("  pose+fk split (cumulative): after the state fill %.0f, after sin/cos %.0f, after the chain walk %.0f us per env" % (t[42] / n / 100, t[43] / n / 100, t[0] / n / 100))
    print("  whole query %.0f us per env; inside growTree %.0f us (%.0f calls per env)" % (t[44] / n / 100, t[45] / n / 100, t[46] / n))
    print("  growTree: node fetch + distance + steer %.0f us, checkMotion (passes included) %.0f us, tree append %.0f us per env" % (t[48] / n / 100, t[49] / n / 100, t[50] / n / 100))
    print("  nearest-neighbour sweeps that fell back to the exact FP64 sweep: %.1f per env" % (t[47] / n))
    if t[40]:
        print("  narrow phase split (cumulative from the start of the collision sweep): after broad %.0f, after closed forms %.0f, after refinement %.0f us per env" % (t[1] / n / 100, t[40] / n / 100, t[2] / n / 100))
    print("  own nearest-neighbour %.0f us, speculation (memo lookup + fused NN + steering) %.0f us per env" % (t[6] / n / 100, t[7] / n / 100))
    print("  speculation split: fused NN %.0f us, two steer+push %.0f us, connect-chain %.0f us per env" % (t[22] / n / 100, t[23] / n / 100, t[38] / n / 100))
    names = "PLANE_SPHERE PLANE_CAPSULE PLANE_CYLINDER PLANE_BOX SPHERE_SPHERE SPHERE_CAPSULE SPHERE_CYLINDER SPHERE_BOX CAPSULE_CAPSULE CAPSULE_BOX BOX_BOX CONVEX PLANE_MESH CONVEX_MESH".split()
    for k, nm in enumerate(names):
        if t[24 + k]: print("  class %-16s rounds/pass %.2f  us/round %.2f  us/pass %.2f" % (nm, t[24 + k] / t[4], t[8 + k] / t[24 + k] / 100, t[8 + k] / t[4] / 100))

if hasattr(lib, "mopa_debug_plan_mpr_pairs"):
    pb = (ctypes.c_uint * 8192)()
    lib.mopa_debug_plan_mpr_pairs(pb, 1)
    c = np.array(list(pb)).reshape(2, 64, 64)
    m = pi.model
    print("  portal refinement entries per pass %.2f; by pair:" % (c.sum() / max(t[4], 1)))
    nm = lambda g: "g%d(%s)@%s" % (g, ["plane", "", "sphere", "capsule", "", "cylinder", "box", "mesh"][int(m.geom_type[g])], m.body_names[int(m.geom_body[g])])
    both = c[0] + c[1]
    for k in np.argsort(-both.ravel())[:16]:
        g1, g2 = divmod(int(k), 64)
        if both[g1, g2] == 0: break
        print("   %-34s %-34s disjoint %.3f  penetrating %.3f per pass" % (nm(g1), nm(g2), c[0, g1, g2] / t[4], c[1, g1, g2] / t[4]))
    if t[51] or t[53]:
        print("  workgroup-per-query build: table hits per env: nearest neighbour %.0f, verdict %.0f; states posed per env by wave 0 / 1 / 2 / 3: %.0f / %.0f / %.0f / %.0f" % (
            t[51] / n, t[52] / n, t[53] / n, t[54] / n, t[55] / n, t[56] / n))
        print("  growTree calls answered from the table alone: %.0f per env (%.0f us per env in that path); passes by cause: first call %.0f, connect first %.0f, connect later %.0f, interior states %.0f per env" % (
            t[57] / n, t[39] / n / 100, t[58] / n, t[59] / n, t[60] / n, t[61] / n))
