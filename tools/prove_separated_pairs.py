#!/usr/bin/env python3
"""Compile-time pruning of candidate pairs that can NEVER violate the contact threshold (K1).

Some candidate pairs of two robot geoms survive every bounding-sphere test -- the links are neighbours but one -- and
never collide anywhere in the joint box (l0's sphere / l2's capsule: present in 100 % of the states, minimum distance
6.7 cm).  This tool PROVES such pairs separated over the whole box of the joints their relative pose depends on, by
branch and bound with a Lipschitz bound:

    true_dist(q) >= LB(q_c) - sum_j rho_j |q_j - q_c,j|        for every q of a box with centre q_c

  * LB(q_c): a lower bound of the true distance at the centre -- the oracle's closed forms (exact for sphere / capsule /
    box combinations, a lower bound for separated boxes); a cylinder is replaced by its enclosing capsule (contains it);
  * rho_j: the largest distance any point of the far-side geom can have from joint j's axis, bounded independently of the
    configuration by the link offsets along the chain + the geom's offset + its bounding radius (1 for a slide): the true
    distance is 1-Lipschitz in the relative displacement of the two point sets.

A box is proven when LB(q_c) - sum_j rho_j w_j > margin (> 0 > contact_threshold); otherwise it is split along its widest
(rho-weighted) side.  Pairs whose relative pose depends on more than 3 joints, on a free joint, or that cannot be proven
within the evaluation budget stay in the list.  The proof covers joint values inside the joint ranges INFLATED BY A GUARD BAND
(0.05 rad / 2 mm; OMPL samples inside the ranges, the rollouts clip to them, MuJoCo's soft limits let a reported qpos sit a little
outside); `_lib.Scene` routes states beyond range + band through the unpruned pair list.  Output: `never_violating_pairs` (pairs of collidable-geom indices) into the scene JSON's meta,
which `_lib.Scene` drops from the kernel's pair list (the oracle keeps checking them: the parity sweeps are the cross-check).

    python tools/prove_separated_pairs.py            # all four scenes, rewrites mopa_rl_amd/scenes/*.json meta
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mopa_rl_amd.mjcf import GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_SPHERE, JNT_FREE, JNT_HINGE, JNT_SLIDE  # noqa: E402
from mopa_rl_amd.scene import ENV_SPECS, planner_inputs, scene_path  # noqa: E402
from oracle import oracle as O  # noqa: E402

MARGIN = 1e-4
MAX_JOINTS = 3
# guard band: the box of joint values is inflated by this much beyond every limited joint's range before it is proven, so that
# the pruning stays sound for states a little OUTSIDE the ranges (MuJoCo's joint limits are soft: a reported qpos can sit a few
# milliradians beyond them).  The runtime routes states beyond range + band through the unpruned pair list (mopa_rl_amd/_lib.py).
BAND_HINGE, BAND_SLIDE = 0.05, 0.002


def joint_box(m, j):
    """(lo, hi) of the proof's box for joint j, or None: an unlimited SLIDE has no box (an unlimited hinge has: one turn)"""
    jt = int(m.jnt_type[j])
    if m.jnt_limited[j]:
        band = BAND_SLIDE if jt == JNT_SLIDE else BAND_HINGE
        return float(m.jnt_range[j][0]) - band, float(m.jnt_range[j][1]) + band
    if jt == JNT_SLIDE:
        return None
    return -np.pi, np.pi


def chain_joints(m, body, stop):
    """joints of the bodies on the path body -> ... -> (exclusive) stop, each with the bodies between it and `body`"""
    out, path = [], []
    b = body
    while b != stop and b > 0:
        path.append(b)
        for j in range(int(m.body_jntadr[b]), int(m.body_jntadr[b]) + int(m.body_jntnum[b])):
            out.append((j, list(path)))
        b = int(m.body_parent[b])
    return out


def lca(m, a, b):
    anc = set()
    x = a
    while x > 0:
        anc.add(x)
        x = int(m.body_parent[x])
    anc.add(0)
    x = b
    while x not in anc:
        x = int(m.body_parent[x])
    return x


def rbound(t, s):
    return {GEOM_SPHERE: s[0], GEOM_CAPSULE: s[0] + s[1], GEOM_CYLINDER: float(np.hypot(s[0], s[1])), GEOM_BOX: float(np.linalg.norm(s))}[t]


def cull_radius(m, orc, q0, a, b, max_evals, floor, leaf=5e-4):
    """The largest distance the two geoms' CENTRES can have in any configuration in which the pair (possibly) violates
    `floor`: a pair whose centres are further apart than that is separated, whatever the bounding spheres say (the closed
    gripper fingers: bounding spheres of 5 cm each, touching only when their centres are 8 mm apart).  Same branch and
    bound: boxes proven violation-free are dropped, the others are split down to `leaf` metres of reach and contribute
    |c_a - c_b|(q_c) + sum_j rho_j w_j (the centres move no further than that inside the box).  Returns (radius, evals) or
    None."""
    setup = pair_setup(m, a, b)
    if setup is None:
        return None
    joints, (t1, s1), (t2, s2), ia, ib = setup
    adr = [j[0] for j in joints]
    rho = np.array([j[3] for j in joints])
    stack = [(np.array([(j[1] + j[2]) / 2 for j in joints]), np.array([(j[2] - j[1]) / 2 for j in joints]))]
    evals, radius = 0, 0.0
    q = q0.copy()
    while stack:
        c, w = stack.pop()
        q[adr] = c
        gp, gm = orc.fk(q)
        d = O.geom_dist(t1, s1, gp[ia], gm[ia], t2, s2, gp[ib], gm[ib])
        evals += 1
        slack = float(rho @ w)
        if d - slack > floor:
            continue
        if float((rho * w).max()) <= leaf or evals > max_evals:
            radius = max(radius, float(np.linalg.norm(gp[ia] - gp[ib])) + slack)
            continue
        k = int(np.argmax(rho * w))
        w2 = w.copy()
        w2[k] /= 2
        for sgn in (-1, 1):
            c2 = c.copy()
            c2[k] += sgn * w2[k]
            stack.append((c2, w2))
    return radius, evals


def pair_setup(m, a, b):
    """(joints [(qadr, lo, hi, rho)], (type1, size1), (type2, size2), geom of shape 1, geom of shape 2) or None"""
    ta, tb = int(m.geom_type[a]), int(m.geom_type[b])
    ok_types = (GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX)
    if ta not in ok_types or tb not in ok_types:
        return None
    ba, bb = int(m.geom_body[a]), int(m.geom_body[b])
    top = lca(m, ba, bb)
    joints = []
    for g, body in ((a, ba), (b, bb)):
        for j, path in chain_joints(m, body, top):
            jt = int(m.jnt_type[j])
            if jt == JNT_FREE:
                return None
            reach = float(np.linalg.norm(m.geom_pos[g])) + rbound(int(m.geom_type[g]), m.geom_size[g]) + float(np.linalg.norm(m.jnt_pos[j]))
            for pb in path[:-1]:
                reach += float(np.linalg.norm(m.body_pos[pb]))
                for jj in range(int(m.body_jntadr[pb]), int(m.body_jntadr[pb]) + int(m.body_jntnum[pb])):
                    if int(m.jnt_type[jj]) == JNT_SLIDE:
                        if not m.jnt_limited[jj]:
                            return None
                        reach += float(np.abs(m.jnt_range[jj] - m.jnt_ref[jj]).max()) + BAND_SLIDE
            rho = 1.0 if jt == JNT_SLIDE else reach
            box = joint_box(m, j)
            if box is None:
                return None
            joints.append((int(m.jnt_qposadr[j]), box[0], box[1], rho))
    if not joints or len(joints) > MAX_JOINTS:
        return None

    def cap(t, s):
        return (GEOM_CAPSULE, s) if t == GEOM_CYLINDER else (t, s)
    (t1, s1), (t2, s2) = cap(ta, m.geom_size[a]), cap(tb, m.geom_size[b])
    ia, ib = (a, b)
    if t1 > t2:
        (t1, s1), (t2, s2) = (t2, s2), (t1, s1)
        ia, ib = b, a
    return joints, (t1, s1), (t2, s2), ia, ib


CT_FLOOR = 2.0e-3      # proven lower bound required before the contact stage drops a pair (its margins are 1 mm)


def prove_pair(m, orc, q0, a, b, max_evals, floor=MARGIN):
    """floor: the proof shows dist > floor everywhere.  MARGIN (> 0: never even touching) is valid for every pair type;
    a negative floor (contact_threshold + MARGIN: touching allowed, the threshold never reached) only where the oracle's
    distance is the exact signed distance -- no cylinder (portal refinement may over-estimate a depth) in the pair."""
    ta, tb = int(m.geom_type[a]), int(m.geom_type[b])
    ok_types = (GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX)
    if ta not in ok_types or tb not in ok_types:
        return None, "type"
    ba, bb = int(m.geom_body[a]), int(m.geom_body[b])
    top = lca(m, ba, bb)
    joints = []
    for g, body in ((a, ba), (b, bb)):
        for j, path in chain_joints(m, body, top):
            jt = int(m.jnt_type[j])
            if jt == JNT_FREE:
                return None, "free joint"
            # reach of geom g about joint j: offsets of the bodies between the joint's body and the geom's body, the joint
            # anchor, the geom's offset and its bounding radius
            reach = float(np.linalg.norm(m.geom_pos[g])) + rbound(int(m.geom_type[g]), m.geom_size[g]) + float(np.linalg.norm(m.jnt_pos[j]))
            for pb in path[:-1]:
                reach += float(np.linalg.norm(m.body_pos[pb]))
                for jj in range(int(m.body_jntadr[pb]), int(m.body_jntadr[pb]) + int(m.body_jntnum[pb])):
                    if int(m.jnt_type[jj]) == JNT_SLIDE:      # a slide below the joint lengthens the arm by its travel
                        if not m.jnt_limited[jj]:
                            return None, "unlimited slide"
                        reach += float(np.abs(m.jnt_range[jj] - m.jnt_ref[jj]).max()) + BAND_SLIDE
            rho = 1.0 if jt == JNT_SLIDE else reach
            box = joint_box(m, j)
            if box is None:
                return None, "unlimited slide"
            joints.append((int(m.jnt_qposadr[j]), box[0], box[1], rho))
    if not joints:
        return None, "rigid"
    if len(joints) > MAX_JOINTS:
        return None, f"{len(joints)} joints"

    def cap(t, s):      # a cylinder inside the capsule of the same axis, radius and half length
        return (GEOM_CAPSULE, s) if t == GEOM_CYLINDER else (t, s)
    (t1, s1), (t2, s2) = cap(ta, m.geom_size[a]), cap(tb, m.geom_size[b])
    ia, ib = (a, b) if t1 <= t2 else (b, a)
    if t1 > t2:
        (t1, s1), (t2, s2) = (t2, s2), (t1, s1)
    adr = [j[0] for j in joints]
    rho = np.array([j[3] for j in joints])
    stack = [(np.array([(j[1] + j[2]) / 2 for j in joints]), np.array([(j[2] - j[1]) / 2 for j in joints]))]
    evals, worst = 0, np.inf
    q = q0.copy()
    while stack:
        c, w = stack.pop()
        q[adr] = c
        gp, gm = orc.fk(q)
        d = O.geom_dist(t1, s1, gp[ia], gm[ia], t2, s2, gp[ib], gm[ib])
        evals += 1
        worst = min(worst, d)
        if d <= floor:
            return False, f"LB {d:.4f} at {np.round(c, 3)}"
        if d - float(rho @ w) > floor:
            continue
        if evals > max_evals:
            return None, f"budget ({evals} evals, min LB {worst:.4f})"
        k = int(np.argmax(rho * w))
        w2 = w.copy()
        w2[k] /= 2
        for sgn in (-1, 1):
            c2 = c.copy()
            c2[k] += sgn * w2[k]
            stack.append((c2, w2))
    return True, f"{evals} evals, min LB {worst:.4f}, {len(joints)} joints"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-evals", type=int, default=400000)
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--max-joints", type=int, default=3)
    args = ap.parse_args()
    global MAX_JOINTS
    MAX_JOINTS = args.max_joints
    from mopa_rl_amd.mjcf import CompiledModel
    for env, spec in ENV_SPECS.items():
        pi = planner_inputs(env)
        m = pi.model
        orc = O.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, spec.contact_threshold)
        ign = set(tuple(p) for p in pi.ignored_contacts)
        q0 = np.array(m.qpos0, dtype=np.float64)
        proven, proven_thr, radii, proven_ct = [], [], [], []
        t0 = time.time()
        for a, b in m.pair_geom:
            a, b = int(a), int(b)
            ia, ib = int(m.geom_mjid[a]), int(m.geom_mjid[b])
            if (min(ia, ib), max(ia, ib)) in ign:
                continue
            res, why = prove_pair(m, orc, q0, a, b, args.max_evals)
            if res is False and GEOM_CYLINDER not in (int(m.geom_type[a]), int(m.geom_type[b])):
                # they can touch; can they reach the (negative) threshold?
                res, why = prove_pair(m, orc, q0, a, b, args.max_evals, floor=spec.contact_threshold + MARGIN)
                if res:
                    proven_thr.append([a, b])
                    print(f"  {env}: PROVEN above the threshold  {(m.all_geom_names[int(m.geom_mjid[a])] or a)} / {(m.all_geom_names[int(m.geom_mjid[b])] or b)}: {why}", flush=True)
                    continue
            name = lambda g: (m.all_geom_names[int(m.geom_mjid[g])] or f"g{int(m.geom_mjid[g])}") + "@" + m.body_names[int(m.geom_body[g])]
            if not res:
                # not prunable: is the bounding-sphere cull at least loose for it?
                has_cyl = GEOM_CYLINDER in (int(m.geom_type[a]), int(m.geom_type[b]))
                # (floor = touching, not the threshold: K1 also reports the deepest penetration, and an overlap shallower than
                #  the threshold must not vanish from it)
                cr = cull_radius(m, orc, q0, a, b, args.max_evals // 4, MARGIN)
                if cr is not None:
                    rsum = rbound(int(m.geom_type[a]), m.geom_size[a]) + rbound(int(m.geom_type[b]), m.geom_size[b])
                    if cr[0] + 1e-3 < 0.9 * rsum:
                        radii.append([a, b, cr[0] + 1e-3])
                        print(f"  {env}: cull radius  {name(a)} / {name(b)}: centres within {cr[0] + 1e-3:.4f} m whenever the threshold is reached "
                              f"(bounding spheres: {rsum:.4f}); {cr[1]} evals", flush=True)
            if res:
                proven.append([a, b])
                print(f"  {env}: PROVEN separated  {name(a)} / {name(b)}: {why}", flush=True)
                # the contact stage of env.step (mopa_rl_amd/dynamics.py: contact_facts) makes contacts at dist < the pair's margin
                # (1 mm in the reference's scenes): a pair leaves ITS table only when proven to stay beyond CT_FLOOR
                res2, why2 = prove_pair(m, orc, q0, a, b, args.max_evals, floor=CT_FLOOR)
                if res2:
                    proven_ct.append([a, b])
            elif res is None and why not in ("type", "free joint", "rigid", "unlimited slide") and "joints" not in why:
                print(f"  {env}: undecided        {name(a)} / {name(b)}: {why}", flush=True)
        print(f"{env}: {len(radii)} tightened cull radii; {len(proven)} (+ {len(proven_thr)} that may touch) of {len(m.pair_geom)} candidate pairs proven never to violate the threshold, {len(proven_ct)} to stay beyond {CT_FLOOR} m ({time.time() - t0:.0f} s)", flush=True)
        if not args.dry:
            path = scene_path(spec.scene)
            cm = CompiledModel.load(path)
            cm.meta["never_violating_pairs"] = proven
            # pairs that may touch but provably stay above contact_threshold: pruned only by scenes whose threshold is <= this one
            cm.meta["never_violating_pairs_thr"] = {"threshold": spec.contact_threshold, "pairs": proven_thr}
            # [geom a, geom b, radius]: the pair can only reach `threshold` while its geom centres are within `radius`
            cm.meta["pair_cull_radius"] = {"threshold": spec.contact_threshold, "pairs": radii}
            cm.meta["prune_guard_band"] = {"hinge": BAND_HINGE, "slide": BAND_SLIDE}
            cm.meta["never_within_margin_pairs"] = {"floor": CT_FLOOR, "pairs": proven_ct}
            cm.meta["proof_geometry_sha256"] = cm.geometry_sha256()
            cm.meta["never_violating_pairs_note"] = ("tools/prove_separated_pairs.py: branch-and-bound Lipschitz proof over the joint ranges inflated by "
                                                     f"the guard band ({BAND_HINGE} rad / {BAND_SLIDE} m), margin {MARGIN} m; valid for joint values "
                                                     "inside range + band -- the runtime sends states beyond that through the unpruned pair list")
            cm.save(path)


if __name__ == "__main__":
    main()
