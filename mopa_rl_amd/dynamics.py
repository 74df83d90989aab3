"""Servo-dynamics facts of a Sawyer env (stage A): the actuated kinematic tree, LUMPED for simulation.

The reference's `env.step` runs `int(frame_dt / dt)` = 75 MuJoCo sub-steps per call with the arm's position servos
tracking `desired_state` and `qfrc_applied = qfrc_bias` as gravity compensation
(env/sawyer/sawyer_push_obstacle.py:186-203, env/base.py:388-392; servo gains env/assets/xml/common/
sawyer_joint_pos_act.xml, joint damping / armature sawyer_dependencies.xml:38,66-78, timestep :11).  The dynamic state of
that loop -- when nothing is in contact -- is the arm's own tree: 7 hinges + the 2 gripper slides.  This module derives,
from a :class:`CompiledModel`, what the HIP kernel (`csrc/mopa_dyn.inc`) and the test oracle both take as input:

  * one body per dof, parents before children; bodies welded to a jointed body (head, screen, gripper base, finger
    tips, the peg ...) are folded into that body's mass / centre of mass / inertia tensor (exact: a weld is rigid);
  * each body's frame relative to its parent dynamic body (welded intermediates composed), the base frame in the world;
  * per dof: damping, armature, joint range, servo gain / force range, whether the env gravity-compensates it.

Everything here is host-side set-up in plain numpy; nothing is on the hot path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .mjcf import JNT_HINGE, JNT_SLIDE, _quat_mul, _quat_to_mat

DYN_MAX = 9   # dofs the kernel keeps per env (LDS budget, csrc/mopa_dyn.inc)


@dataclass
class DynFacts:
    nd: int
    body: np.ndarray          # [nd] model body id of each dynamic body
    parent: np.ndarray        # [nd] i32, -1 = fixed base
    jtype: np.ndarray         # [nd] i32
    qadr: np.ndarray          # [nd] i32
    rel_pos: np.ndarray       # [nd,3]
    rel_quat: np.ndarray      # [nd,4]
    axis: np.ndarray          # [nd,3]
    jpos: np.ndarray          # [nd,3]
    qref: np.ndarray          # [nd]
    mass: np.ndarray          # [nd]
    ipos: np.ndarray          # [nd,3]
    inertia: np.ndarray       # [nd,6] xx yy zz xy xz yz about the COM, body axes
    damping: np.ndarray
    armature: np.ndarray
    limited: np.ndarray       # i32
    lo: np.ndarray
    hi: np.ndarray
    actuated: np.ndarray      # i32
    kp: np.ndarray
    force_lo: np.ndarray
    force_hi: np.ndarray
    gravcomp: np.ndarray      # i32
    gravity: np.ndarray       # [3]
    timestep: float
    nsub: int


def _compose(p1, q1, p2, q2):
    """frame 2 given in frame 1, frame 1 given in frame 0  ->  frame 2 in frame 0"""
    return p1 + _quat_to_mat(q1) @ p2, _quat_mul(q1, q2)


def dyn_facts(model, facts, frame_dt: float = 0.15) -> DynFacts:
    """`facts`: the env's :class:`EnvFacts` (arm / actuator qpos addresses, joint limits as the env clamps them)."""
    m = model
    if len(getattr(m, "body_mass", ())) == 0:
        raise ValueError("compiled scene carries no inertials (recompile with tools/compile_scenes.py)")
    jnt_of_qadr = {int(a): j for j, a in enumerate(m.jnt_qposadr)}
    root = int(m.jnt_body[jnt_of_qadr[int(facts.arm_qpos_idx[0])]])
    nb = len(m.body_names)

    def in_subtree(b):
        while b > 0:
            if b == root:
                return True
            b = int(m.body_parent[b])
        return False

    sub = [b for b in range(1, nb) if in_subtree(b)]
    dyn_bodies = [b for b in sub if m.body_jntnum[b] > 0]
    for b in dyn_bodies:
        j = int(m.body_jntadr[b])
        if m.body_jntnum[b] != 1 or int(m.jnt_type[j]) not in (JNT_HINGE, JNT_SLIDE):
            raise ValueError(f"body {m.body_names[b]!r}: the dynamic tree takes one hinge / slide joint per body")
        if m.jnt_stiffness[j] != 0.0:
            raise ValueError("joint stiffness is not modelled")
    nd = len(dyn_bodies)
    if nd > DYN_MAX:
        raise ValueError(f"{nd} dofs in the actuated tree, the kernel keeps at most {DYN_MAX}")
    idx = {b: i for i, b in enumerate(dyn_bodies)}

    def owner(b):      # the dynamic body a (possibly welded) body moves with; -1: fixed to the world
        while b > 0 and b not in idx:
            b = int(m.body_parent[b])
        return idx.get(b, -1)

    def frame_in(b, anc):   # pose of body b's frame in the frame of its ancestor `anc` (0 = world)
        p, q = np.zeros(3), np.array([1.0, 0.0, 0.0, 0.0])
        chain = []
        while b != anc:
            chain.append(b)
            b = int(m.body_parent[b])
        for c in reversed(chain):
            p, q = _compose(p, q, np.asarray(m.body_pos[c], dtype=np.float64), np.asarray(m.body_quat[c], dtype=np.float64))
        return p, q

    parent = np.full(nd, -1, dtype=np.int32)
    rel_pos, rel_quat = np.zeros((nd, 3)), np.zeros((nd, 4))
    mass, ipos, inertia = np.zeros(nd), np.zeros((nd, 3)), np.zeros((nd, 6))
    for i, b in enumerate(dyn_bodies):
        pb = int(m.body_parent[b])
        parent[i] = owner(pb)
        anc = dyn_bodies[parent[i]] if parent[i] >= 0 else 0
        rel_pos[i], rel_quat[i] = frame_in(b, anc)
        # lump: the body itself + everything welded to it
        parts = []
        for w in sub:
            if owner(w) != i:
                continue
            if w != b and m.body_jntnum[w] > 0:
                continue
            if m.body_mass[w] <= 0.0:
                continue
            p, q = frame_in(w, b)
            R = _quat_to_mat(q)
            xx, yy, zz, xy, xz, yz = m.body_inertia[w]
            T = R @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ R.T
            parts.append((float(m.body_mass[w]), p + R @ np.asarray(m.body_ipos[w], dtype=np.float64), T))
        mt = sum(p[0] for p in parts)
        if mt <= 0.0:
            raise ValueError(f"dynamic body {m.body_names[b]!r} has no mass")
        c = sum(p[0] * p[1] for p in parts) / mt
        T = np.zeros((3, 3))
        for pm, pc, pT in parts:
            d = pc - c
            T += pT + pm * (float(d @ d) * np.eye(3) - np.outer(d, d))
        mass[i], ipos[i] = mt, c
        inertia[i] = [T[0, 0], T[1, 1], T[2, 2], T[0, 1], T[0, 2], T[1, 2]]

    jid = np.array([int(m.body_jntadr[b]) for b in dyn_bodies])
    qadr = m.jnt_qposadr[jid].astype(np.int32)
    arm = set(int(a) for a in facts.arm_qpos_idx)
    actuated, kp = np.zeros(nd, dtype=np.int32), np.zeros(nd)
    flo, fhi = np.full(nd, -np.inf), np.full(nd, np.inf)
    n_act = len(facts.act_qpos_idx)
    for k in range(n_act):
        i = int(np.where(qadr == int(facts.act_qpos_idx[k]))[0][0])
        if int(m.act_kind[k]) != 1:
            raise ValueError("only position servos are modelled")
        if len(getattr(m, "act_gear", ())) and float(m.act_gear[k]) != 1.0:
            raise ValueError("actuator gear != 1 is not modelled (the servo force is kp (ctrl - q) on the joint itself)")
        if not int(m.act_ctrllimited[k]):
            raise ValueError("the kernels clamp ctrl to the actuator's ctrlrange: an actuator without ctrllimited is not modelled")
        actuated[i], kp[i] = 1, float(m.act_gain[k])
        if m.act_forcelimited[k]:
            flo[i], fhi[i] = m.act_forcerange[k]
    nsub = int(frame_dt / float(m.opt[3]))
    return DynFacts(
        nd=nd, body=np.array(dyn_bodies, dtype=np.int32), parent=parent, jtype=m.jnt_type[jid].astype(np.int32), qadr=qadr,
        rel_pos=rel_pos, rel_quat=rel_quat, axis=np.asarray(m.jnt_axis[jid], dtype=np.float64).copy(),
        jpos=np.asarray(m.jnt_pos[jid], dtype=np.float64).copy(), qref=np.asarray(m.jnt_ref[jid], dtype=np.float64).copy(),
        mass=mass, ipos=ipos, inertia=inertia,
        damping=np.asarray(m.jnt_damping[jid], dtype=np.float64).copy(), armature=np.asarray(m.jnt_armature[jid], dtype=np.float64).copy(),
        limited=np.asarray(facts.qpos_limited, dtype=np.int32)[qadr].copy(), lo=np.asarray(facts.qpos_min, dtype=np.float64)[qadr].copy(),
        hi=np.asarray(facts.qpos_max, dtype=np.float64)[qadr].copy(),
        actuated=actuated, kp=kp, force_lo=flo, force_hi=fhi,
        gravcomp=np.array([1 if int(a) in arm else 0 for a in qadr], dtype=np.int32),
        gravity=np.asarray(m.opt[:3], dtype=np.float64).copy(), timestep=float(m.opt[3]), nsub=nsub)


# ---- stage B (Push): the manipulated object as a free body with penalty contacts -------------------------------------
@dataclass
class ObjFacts:
    """What the contact model of `csrc/mopa_dyn.inc` / the test oracle takes for the manipulated object (Push: the cube,
    env/assets/xml/sawyer_push_obstacle.xml): its free joint, inertial, box shape and feature points, and the colliders --
    every geom MuJoCo would pair with it (the compiled candidate-pair list), static ones posed in the world, robot geoms in
    the frame of their (lumped) dynamic body.  The force law is NOT MuJoCo's (penalty spring-damper + capped Coulomb
    friction, one-way coupling); kn / dn / eps_v / ct_max are its constants."""
    qadr: int
    mass: float
    inertia: np.ndarray        # [3] principal (body frame)
    damping: float
    half: np.ndarray           # [3]
    rbound: float
    feat: np.ndarray           # [nfeat,3] vertices first
    co_body: np.ndarray        # [ncol] i32, -1 static
    co_type: np.ndarray        # [ncol] i32 (mjtGeom)
    co_size: np.ndarray        # [ncol,3]
    co_pos: np.ndarray         # [ncol,3]
    co_mat: np.ndarray         # [ncol,9]
    co_mu: np.ndarray          # [ncol]
    co_rbound: np.ndarray      # [ncol]
    kn: float
    dn: float
    eps_v: float
    ct_max: float
    inv_mass: float = 0.0
    inv_inertia: np.ndarray = None
    precull_every: int = 15       # the full collider scan runs on every 15th sub-step (30 ms) of a call, with the bounding
    precull_margin: float = 0.15  # spheres inflated by 15 cm; the other sub-steps visit only the colliders that passed it


def obj_facts(model, dyn: DynFacts, geom_name: str = "cube", kn: float = 500.0, dn: float = 3.0, eps_v: float = 1e-3) -> ObjFacts:
    from .mjcf import GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_PLANE, GEOM_SPHERE, JNT_FREE
    m = model
    cg = int(np.where(m.geom_mjid == m.geom_name2id(geom_name))[0][0])
    if int(m.geom_type[cg]) != GEOM_BOX:
        raise ValueError("the contact model takes a box object")
    ob = int(m.geom_body[cg])
    j = int(m.body_jntadr[ob])
    if m.body_jntnum[ob] != 1 or int(m.jnt_type[j]) != JNT_FREE:
        raise ValueError("the object must hang on a free joint")
    if np.abs(m.geom_pos[cg]).max() > 0 or np.abs(m.body_ipos[ob]).max() > 1e-12 or np.abs(m.body_inertia[ob][3:]).max() > 1e-15:
        raise ValueError("object geom / inertial frame must coincide with the body frame")
    half = np.asarray(m.geom_size[cg], dtype=np.float64).copy()
    sg = [-1.0, 1.0]
    verts = [[(x if k & 1 else -x) for k, x in ((i, half[0]), (i >> 1, half[1]), (i >> 2, half[2]))] for i in range(8)]
    edges = [[a * half[0], b * half[1], 0.0] for a in sg for b in sg] + [[a * half[0], 0.0, b * half[2]] for a in sg for b in sg] + \
            [[0.0, a * half[1], b * half[2]] for a in sg for b in sg]
    faces = [[a * half[0], 0.0, 0.0] for a in sg] + [[0.0, a * half[1], 0.0] for a in sg] + [[0.0, 0.0, a * half[2]] for a in sg]
    feat = np.array(verts + edges + faces, dtype=np.float64)
    idx = {int(b): i for i, b in enumerate(dyn.body)}
    nb = len(m.body_names)

    def owner(b):
        while b > 0 and b not in idx:
            b = int(m.body_parent[b])
        return idx.get(b, -1)

    def moving(b):
        while b > 0:
            if m.body_jntnum[b] > 0:
                return True
            b = int(m.body_parent[b])
        return False

    def frame_in(b, anc):
        p, q = np.zeros(3), np.array([1.0, 0.0, 0.0, 0.0])
        chain = []
        while b != anc:
            chain.append(b)
            b = int(m.body_parent[b])
        for c in reversed(chain):
            p, q = _compose(p, q, np.asarray(m.body_pos[c], dtype=np.float64), np.asarray(m.body_quat[c], dtype=np.float64))
        return p, q

    def rbound(t, s):
        return {GEOM_SPHERE: s[0], GEOM_CAPSULE: s[0] + s[1], GEOM_CYLINDER: float(np.hypot(s[0], s[1])), GEOM_BOX: float(np.linalg.norm(s)),
                GEOM_PLANE: 0.0}[t]

    cols = []
    for a, b in m.pair_geom:
        if cg not in (int(a), int(b)):
            continue
        g = int(b if int(a) == cg else a)
        t = int(m.geom_type[g])
        if t not in (GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX):
            raise ValueError("unsupported collider type")
        gb = int(m.geom_body[g])
        own = owner(gb)
        if own < 0 and moving(gb):
            continue          # a moving body outside the dynamic tree (nothing of the kind in the Sawyer scenes)
        anc = int(dyn.body[own]) if own >= 0 else 0
        bp, bq = frame_in(gb, anc)
        gp, gq = _compose(bp, bq, np.asarray(m.geom_pos[g], dtype=np.float64), np.asarray(m.geom_quat[g], dtype=np.float64))
        cols.append((own, t, np.asarray(m.geom_size[g], dtype=np.float64), gp, _quat_to_mat(gq).ravel(),
                     max(float(m.geom_friction[g]), float(m.geom_friction[cg])), rbound(t, m.geom_size[g])))
    cols.sort(key=lambda c: (c[0] >= 0, c[0]))        # static ones first, then by dynamic body (stable)
    mass = float(m.body_mass[ob])
    return ObjFacts(
        qadr=int(m.jnt_qposadr[j]), mass=mass, inertia=np.asarray(m.body_inertia[ob][:3], dtype=np.float64).copy(),
        damping=float(m.jnt_damping[j]), half=half, rbound=float(np.linalg.norm(half)), feat=feat,
        co_body=np.array([c[0] for c in cols], dtype=np.int32), co_type=np.array([c[1] for c in cols], dtype=np.int32),
        co_size=np.array([c[2] for c in cols]), co_pos=np.array([c[3] for c in cols]), co_mat=np.array([c[4] for c in cols]),
        co_mu=np.array([c[5] for c in cols]), co_rbound=np.array([c[6] for c in cols]),
        kn=float(kn), dn=float(dn), eps_v=float(eps_v), ct_max=mass / (float(dyn.timestep) * 8.0),
        inv_mass=1.0 / mass, inv_inertia=1.0 / np.asarray(m.body_inertia[ob][:3], dtype=np.float64))


# ---- stage C: contacts of the arm and of the manipulated object, behind one constraint solve -----------------------------
CT_MAXCON = 16        # contacts the C ABI / the test oracle keep per env and sub-step at most (one per lane of an env's 16)


@dataclass
class CtFacts:
    """What the contact + constraint stage of `csrc/mopa_contact.inc` / the test oracle takes (stage C).
    Bodies: 0 .. nd-1 the lumped dynamic bodies of the arm (`DynFacts`), nd the manipulated object (free body: Push the
    cube, Lift the can, Assembly the furniture with its welded parts lumped), -1 the world.
    Shapes = the collidable geoms, posed in the frame of their body; a mesh geom (the can) enters as the bounding cylinder of
    its hull.  Features = sample points (sphere-swept when `ft_rad` > 0) on a shape, same frame.  A directed pair (F, S) tests
    the features of F against the signed-distance function of S; its solver parameters are MuJoCo's per-pair mix of the two
    geoms' friction / margin / solref / solimp (env/assets/xml/common/sawyer_dependencies.xml:36, the scene XMLs)."""
    sh_geom: np.ndarray        # [ns] collidable-geom index of the shape
    sh_body: np.ndarray        # [ns] i32
    sh_type: np.ndarray        # [ns] i32 (mjtGeom; mesh -> cylinder)
    sh_size: np.ndarray        # [ns,3]
    sh_pos: np.ndarray         # [ns,3]
    sh_mat: np.ndarray         # [ns,9]
    sh_rbound: np.ndarray      # [ns]
    sh_feat0: np.ndarray       # [ns+1] i32
    ft_pos: np.ndarray         # [nf,3]
    ft_rad: np.ndarray         # [nf]
    pr_f: np.ndarray           # [np] i32
    pr_s: np.ndarray           # [np] i32
    pr_par: np.ndarray         # [np,12]: mu, margin, K, B, d0, dmax, width, 0, condim (3 / 4 / 6), torsional friction, rolling friction, 0
    obj_qadr: int
    obj_mass: float
    obj_inertia: np.ndarray    # [3] principal moments at the COM
    obj_ipos: np.ndarray       # [3] COM in the object's body frame
    obj_iquat: np.ndarray      # [4] principal frame in the body frame
    obj_damping: float
    obj_inv_mass: float
    obj_inv_inertia: np.ndarray
    obj_inv_mass_d: float      # 1 / (m + h damping)
    obj_inv_inertia_d: np.ndarray
    maxcon: int
    maxpair: int
    iterations: int
    tolerance: float
    inv_scale: float
    precull_every: int
    precull_margin: float
    warmstart: int
    near_every: int = 3           # third culling level: the active pairs within near_margin of contact, re-listed every near_every sub-steps
    near_margin: float = 0.03     # (3 sub-steps at 5 m/s of relative motion -- the precull's own assumption: 15 cm per 15 sub-steps)
    noslip_iterations: int = 5    # sawyer_dependencies.xml:11 noslip_iterations="5"
    noslip_tolerance: float = 1e-6   # MuJoCo's default
    solver: int = 2                  # 2: Newton with ELLIPTIC cones (the XML: cone="elliptic", solver unnamed = MuJoCo's default Newton),
                                     # 1: Newton with pyramidal cones, 0: projected Gauss-Seidel (pyramidal)
    limit_rows: int = 1              # joint limits as rows of the Newton solver (MuJoCo) instead of stage A's inelastic stop
    lim_par: tuple = (0.0,) * 8      # their solver parameters in a pair record's layout: -, margin, K, B, d0, dmax, width, -
    condim_downgraded: int = 0       # directed pairs whose geoms ask for condim 4 / 6 (torsional / rolling friction) and are solved as condim 3
                                     # (the pyramidal solver forms only: solver 2 solves every pair with the XML's condim)
    arena: int = 0                   # solver 2: doubles of an env's LDS share the contact records have (a contact's record: record_size); 0 = as
                                     # many as keep four waves on a CU (the library's default)


def _joint_space_inertia_diag(dyn: DynFacts, qpos_row: np.ndarray) -> np.ndarray:
    """diag(M) of the lumped arm at `qpos_row` (for MuJoCo's `meaninertia`): sum over the bodies at or below each dof of the
    body's inertia about the dof's axis (hinge) or its mass (slide), + armature."""
    nd = dyn.nd
    pos, mat = np.zeros((nd, 3)), np.zeros((nd, 3, 3))
    for i in range(nd):
        R = _quat_to_mat(dyn.rel_quat[i])
        p = np.asarray(dyn.rel_pos[i], dtype=np.float64)
        if dyn.parent[i] >= 0:
            p, R = pos[dyn.parent[i]] + mat[dyn.parent[i]] @ p, mat[dyn.parent[i]] @ R
        dq = float(qpos_row[dyn.qadr[i]] - dyn.qref[i])
        if dyn.jtype[i] == JNT_SLIDE:
            p = p + R @ dyn.axis[i] * dq
        else:
            a = dyn.axis[i] / np.linalg.norm(dyn.axis[i])
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            Rj = np.eye(3) + np.sin(dq) * K + (1 - np.cos(dq)) * (K @ K)
            anchor = p + R @ dyn.jpos[i]
            R = R @ Rj
            p = anchor - R @ dyn.jpos[i]
        pos[i], mat[i] = p, R
    diag = np.zeros(nd)
    for k in range(nd):
        a = mat[k] @ dyn.axis[k]
        anchor = pos[k] + mat[k] @ dyn.jpos[k]
        for b in range(nd):
            c = b
            while c >= 0 and c != k:
                c = int(dyn.parent[c])
            if c != k:
                continue
            if dyn.jtype[k] == JNT_SLIDE:
                diag[k] += dyn.mass[b]
            else:
                xx, yy, zz, xy, xz, yz = dyn.inertia[b]
                Iw = mat[b] @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ mat[b].T
                r = pos[b] + mat[b] @ dyn.ipos[b] - anchor
                diag[k] += float(a @ Iw @ a) + dyn.mass[b] * float(np.dot(np.cross(a, r), np.cross(a, r)))
        diag[k] += dyn.armature[k]
    return diag


def _spread_order(n: int):
    """indices 0 .. n-1 of points on a circle, ordered so that every prefix is well spread (bit reversal): a per-pair contact
    cap that keeps the first few hits then keeps a support polygon, not one side of the rim"""
    bits = max(1, int(np.ceil(np.log2(n))))
    order = sorted(range(1 << bits), key=lambda i: int(format(i, f"0{bits}b")[::-1], 2))
    return [i for i in order if i < n]


def contact_facts(model, dyn: DynFacts, object_body: str, maxcon: int = None, maxpair: int = None, iterations: int = 50,
                  tolerance: float = 1e-10, precull_every: int = 15, precull_margin: float = 0.15, near_every: int = 3, near_margin: float = 0.03, warmstart: bool = True,
                  noslip_iterations: int = 5, noslip_tolerance: float = 1e-6, solver: str = "newton", cone: str = None, limit_rows=None,
                  limit_solref=(0.02, 1.0), limit_solimp=(0.9, 0.95, 0.001),
                  qpos_ref: np.ndarray = None, condim: str = "xml", arena: int = None) -> CtFacts:
    from .mjcf import GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_MESH, GEOM_PLANE, GEOM_SPHERE, JNT_FREE
    m = model
    if len(getattr(m, "geom_solref", ())) == 0:
        raise ValueError("compiled scene carries no solver parameters (recompile with tools/compile_scenes.py)")
    # friction cone: the XML's `cone="elliptic"` (sawyer_dependencies.xml:11) with the Newton solver; the projected Gauss-Seidel form knows
    # pyramids only
    if cone is None:
        cone = "elliptic" if solver == "newton" else "pyramidal"
    if cone not in ("elliptic", "pyramidal") or (cone == "elliptic" and solver != "newton"):
        raise ValueError("cone: 'elliptic' (Newton solver only) or 'pyramidal'")
    # the solver keeps per-contact state one contact per lane of an env's 16 (elliptic) / two pyramid rows per lane (Newton, pyramidal)
    cap = {("newton", "elliptic"): 16, ("newton", "pyramidal"): 8, ("pgs", "pyramidal"): 16}[(solver, cone)]
    # defaults: the elliptic form keeps up to 16 contacts (one per lane) and 8 of a directed pair out of an LDS arena of variable-size
    # records; the pyramidal forms keep round 5's fixed records (8 contacts, 4 of a pair)
    if maxcon is None:
        maxcon = 16 if cone == "elliptic" else 8
    if maxpair is None:
        maxpair = 8 if cone == "elliptic" else 4
    if condim not in ("xml", "3"):
        raise ValueError("condim: 'xml' (the pairs' own: max of the two geoms') or '3' (sliding friction only)")
    if maxcon > cap:
        raise ValueError(f"maxcon <= {cap} for solver {solver!r} with {cone} cones")
    names = list(m.body_names)
    ob = names.index(object_body)
    jo = int(m.body_jntadr[ob])
    if m.body_jntnum[ob] != 1 or int(m.jnt_type[jo]) != JNT_FREE:
        raise ValueError("the object must hang on a free joint")
    idx = {int(b): i for i, b in enumerate(dyn.body)}
    nd = dyn.nd
    h = float(dyn.timestep)

    def owner(b):       # -> (body index, frame body id): dyn body i, nd = object, -1 = world, None = moves but is not simulated
        c = b
        while c > 0:
            if c in idx:
                return idx[c], c
            if c == ob:
                return nd, ob
            if m.body_jntnum[c] > 0:
                return None, c
            c = int(m.body_parent[c])
        return -1, 0

    def frame_in(b, anc):
        p, q = np.zeros(3), np.array([1.0, 0.0, 0.0, 0.0])
        chain = []
        while b != anc:
            chain.append(b)
            b = int(m.body_parent[b])
        for c in reversed(chain):
            p, q = _compose(p, q, np.asarray(m.body_pos[c], dtype=np.float64), np.asarray(m.body_quat[c], dtype=np.float64))
        return p, q

    # ---- the object's inertial: the free body + everything welded below it
    parts = []
    for w in range(1, len(names)):
        ow, _ = owner(w)
        if ow != nd or float(m.body_mass[w]) <= 0.0:
            continue
        p, q = frame_in(w, ob)
        R = _quat_to_mat(q)
        xx, yy, zz, xy, xz, yz = m.body_inertia[w]
        T = R @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ R.T
        parts.append((float(m.body_mass[w]), p + R @ np.asarray(m.body_ipos[w], dtype=np.float64), T))
    mt = sum(p[0] for p in parts)
    if mt <= 0.0:
        raise ValueError("the object has no mass")
    com = sum(p[0] * p[1] for p in parts) / mt
    T = np.zeros((3, 3))
    for pm, pc, pT in parts:
        dd = pc - com
        T += pT + pm * (float(dd @ dd) * np.eye(3) - np.outer(dd, dd))
    if np.abs(T - np.diag(np.diag(T))).max() <= 1e-12 * np.abs(T).max():
        prin, iquat = np.diag(T).copy(), np.array([1.0, 0.0, 0.0, 0.0])
    else:
        from scipy.spatial.transform import Rotation
        w_, V = np.linalg.eigh(T)
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        x, y, z, w4 = Rotation.from_matrix(V).as_quat()
        prin, iquat = w_.copy(), np.array([w4, x, y, z])
    damp = float(m.jnt_damping[jo])

    # ---- shapes and features
    def rbound(t, s):
        return {GEOM_SPHERE: s[0], GEOM_CAPSULE: s[0] + s[1], GEOM_CYLINDER: float(np.hypot(s[0], s[1])), GEOM_BOX: float(np.linalg.norm(s)),
                GEOM_PLANE: 0.0}[t]

    ng = len(m.geom_type)
    shape_of = {}
    sh = []
    feats_of = []
    for g in range(ng):
        gb = int(m.geom_body[g])
        own, fb = owner(gb)
        if own is None:
            continue
        bp, bq = frame_in(gb, fb)
        gp, gq = _compose(bp, bq, np.asarray(m.geom_pos[g], dtype=np.float64), np.asarray(m.geom_quat[g], dtype=np.float64))
        R = _quat_to_mat(gq)
        t = int(m.geom_type[g])
        size = np.asarray(m.geom_size[g], dtype=np.float64).copy()
        pts, rad = np.zeros((0, 3)), 0.0
        if t == GEOM_MESH:
            did = int(m.geom_dataid[g])
            v = np.asarray(m.mesh_vert[int(m.mesh_vertadr[did]):int(m.mesh_vertadr[did]) + int(m.mesh_vertnum[did])], dtype=np.float64)
            zlo, zhi = float(v[:, 2].min()), float(v[:, 2].max())
            rho = np.hypot(v[:, 0], v[:, 1])
            # features: per distinct height ring of the hull, the vertex nearest each of 8 equally spaced directions
            rings = []
            for z in np.unique(np.round(v[:, 2], 4)):
                ring = v[np.abs(v[:, 2] - z) < 1e-4]
                ang = np.arctan2(ring[:, 1], ring[:, 0])
                pick = []
                for a in np.arange(8)[_spread_order(8)] * (np.pi / 4):
                    i = int(np.argmin(np.abs(np.angle(np.exp(1j * (ang - a))))))
                    if i not in pick:
                        pick.append(i)
                rings.append(ring[pick])
            pts = np.concatenate(rings)
            t, size = GEOM_CYLINDER, np.array([float(rho.max()), 0.5 * (zhi - zlo), 0.0])
            gp = gp + R @ np.array([0.0, 0.0, 0.5 * (zhi + zlo)])
            pts = pts - np.array([0.0, 0.0, 0.5 * (zhi + zlo)])
        elif t == GEOM_SPHERE:
            pts, rad = np.zeros((1, 3)), float(size[0])
        elif t == GEOM_CAPSULE:
            n = int(np.ceil(2.0 * size[1] / size[0])) + 1
            pts = np.stack([np.zeros(n), np.zeros(n), np.linspace(-size[1], size[1], n)], axis=1)
            rad = float(size[0])
        elif t == GEOM_CYLINDER:
            nr = 8 if size[0] < 0.015 else 16       # rim sampling: a flat face sinks at most r (1 - cos(pi / nr)) between two samples
            ang = np.arange(nr)[_spread_order(nr)] * (2.0 * np.pi / nr)
            ring = np.stack([size[0] * np.cos(ang), size[0] * np.sin(ang), np.zeros(nr)], axis=1)
            pts = np.concatenate([ring + [0, 0, size[1]], ring - [0, 0, size[1]], [[0, 0, size[1]], [0, 0, -size[1]]]])
        elif t == GEOM_BOX:
            pts = np.array([[(size[0] if i & 1 else -size[0]), (size[1] if i & 2 else -size[1]), (size[2] if i & 4 else -size[2])] for i in range(8)])
        elif t != GEOM_PLANE:
            raise ValueError("unsupported collider type")
        shape_of[g] = len(sh)
        sh.append((g, own, t, size, gp, R.ravel(), rbound(t, size)))
        feats_of.append((gp + pts @ R.T, rad))

    # Pairs left out of the contact tables: only those PROVEN to stay beyond this stage's own reach -- tools/prove_separated_pairs.py shows
    # dist > floor (2 mm) over the joint ranges + guard band for them; a pair goes only if its margin (+ 0.5 mm slack) is inside that floor.
    # (The planner's `never_violating_pairs` proof only shows dist > 0.1 mm: enough for a threshold <= 0, not for contacts made at 1 mm.)
    nwm = m.meta.get("never_within_margin_pairs") or {}
    never_floor = float(nwm.get("floor", 0.0))
    never = {tuple(sorted((int(a), int(b)))) for a, b in (nwm.get("pairs") or [])}
    round_t = (GEOM_SPHERE, GEOM_CAPSULE)
    pairs = []
    n_downgraded = 0
    used_as_f = set()
    for a, b in m.pair_geom:
        a, b = int(a), int(b)
        if a not in shape_of or b not in shape_of:
            continue
        if tuple(sorted((a, b))) in never and max(float(m.geom_margin[a]), float(m.geom_margin[b])) + 5e-4 <= never_floor:
            continue
        sa, sb = shape_of[a], shape_of[b]
        if sh[sa][1] < 0 and sh[sb][1] < 0:
            continue
        ta, tb = sh[sa][2], sh[sb][2]
        if ta in round_t:
            dirs = [(sa, sb)]
        elif tb in round_t:
            dirs = [(sb, sa)]
        else:
            dirs = [(sa, sb), (sb, sa)]
        mu = max(float(m.geom_friction[a]), float(m.geom_friction[b]))
        # [3P] contact condim = max of the geoms'; friction coefficients = max, each of the three kinds (equal priorities)
        cdim = 3
        f_tors, f_roll = 0.005, 0.0001
        if len(getattr(m, "geom_condim", ())):
            cdim = max(int(m.geom_condim[a]), int(m.geom_condim[b]))
            if cdim not in (3, 4, 6):
                raise ValueError("condim 1 / other values are not supported")
        if len(getattr(m, "geom_friction3", ())):
            f_tors = max(float(m.geom_friction3[a][1]), float(m.geom_friction3[b][1]))
            f_roll = max(float(m.geom_friction3[a][2]), float(m.geom_friction3[b][2]))
        elif cdim > 3:
            raise ValueError("compiled scene carries no torsional / rolling friction (recompile with tools/compile_scenes.py)")
        margin = max(float(m.geom_margin[a]), float(m.geom_margin[b]))
        tc = max(0.5 * (float(m.geom_solref[a][0]) + float(m.geom_solref[b][0])), 2.0 * h)
        dr = 0.5 * (float(m.geom_solref[a][1]) + float(m.geom_solref[b][1]))
        d0, dmax, width = (0.5 * (float(m.geom_solimp[a][k]) + float(m.geom_solimp[b][k])) for k in range(3))
        if not (m.geom_solref[a][0] > 0 and m.geom_solref[b][0] > 0 and dr > 0 and 0 < d0 <= dmax < 1 and width > 0):
            raise ValueError("unsupported solref / solimp")
        K, B = 1.0 / (dmax * dmax * tc * tc * dr * dr), 2.0 / (dmax * tc)
        for f_, s_ in dirs:
            if sh[f_][2] == GEOM_PLANE:
                continue
            use_dim = cdim if (cone == "elliptic" and condim == "xml") else 3
            if cdim > use_dim:
                n_downgraded += 1
            pairs.append((f_, s_, [mu, margin, K, B, d0, dmax, width, 0.0, float(use_dim), f_tors, f_roll, 0.0]))
            used_as_f.add(f_)
    # features only of shapes that appear as F
    feat0, ft_pos, ft_rad = [0], [], []
    for s_ in range(len(sh)):
        if s_ in used_as_f:
            p, r = feats_of[s_]
            ft_pos.append(p)
            ft_rad += [r] * len(p)
        feat0.append(len(ft_rad))
    # meaninertia: mean of diag(M) over the arm dofs and the object's six
    row = np.asarray(m.qpos0 if qpos_ref is None else qpos_ref, dtype=np.float64)
    dg = np.concatenate([_joint_space_inertia_diag(dyn, row), [mt] * 3, prin])
    nv = nd + 6
    if near_every < 1 or precull_every % near_every != 0 or not (0.0 <= near_margin <= precull_margin):
        raise ValueError("precull_every must be a multiple of near_every, and near_margin <= precull_margin")
    # joint limits as solver rows: MuJoCo's joint defaults (solreflimit 0.02 1, solimplimit 0.9 0.95 0.001, margin 0; the XML sets
    # none), the time constant kept >= 2 timesteps as for contacts
    if limit_rows is None:
        limit_rows = solver == "newton"
    if limit_rows and solver != "newton":
        raise ValueError("joint limits as solver rows need the Newton solver")
    l_tc, l_dr = max(float(limit_solref[0]), 2.0 * float(dyn.timestep)), float(limit_solref[1])
    l_d0, l_dmax, l_w = (float(x) for x in limit_solimp)
    lim_par = (0.0, 0.0, 1.0 / (l_dmax ** 2 * l_tc ** 2 * l_dr ** 2), 2.0 / (l_dmax * l_tc), l_d0, l_dmax, l_w, 0.0)
    return CtFacts(
        sh_geom=np.array([s[0] for s in sh], dtype=np.int32), sh_body=np.array([s[1] for s in sh], dtype=np.int32),
        sh_type=np.array([s[2] for s in sh], dtype=np.int32), sh_size=np.array([s[3] for s in sh]), sh_pos=np.array([s[4] for s in sh]),
        sh_mat=np.array([s[5] for s in sh]), sh_rbound=np.array([s[6] for s in sh]), sh_feat0=np.array(feat0, dtype=np.int32),
        ft_pos=np.concatenate(ft_pos) if ft_pos else np.zeros((0, 3)), ft_rad=np.array(ft_rad, dtype=np.float64),
        pr_f=np.array([p[0] for p in pairs], dtype=np.int32), pr_s=np.array([p[1] for p in pairs], dtype=np.int32),
        pr_par=np.array([p[2] for p in pairs], dtype=np.float64).reshape(-1, 12),
        obj_qadr=int(m.jnt_qposadr[jo]), obj_mass=mt, obj_inertia=prin, obj_ipos=com, obj_iquat=iquat, obj_damping=damp,
        obj_inv_mass=1.0 / mt, obj_inv_inertia=1.0 / prin, obj_inv_mass_d=1.0 / (mt + h * damp), obj_inv_inertia_d=1.0 / (prin + h * damp),
        maxcon=int(maxcon), maxpair=int(maxpair), iterations=int(iterations), tolerance=float(tolerance),
        inv_scale=1.0 / (float(dg.mean()) * max(1, nv)), precull_every=int(precull_every), precull_margin=float(precull_margin),
        warmstart=int(bool(warmstart)), near_every=int(near_every), near_margin=float(near_margin), noslip_iterations=int(noslip_iterations), noslip_tolerance=float(noslip_tolerance),
        solver={("pgs", "pyramidal"): 0, ("newton", "pyramidal"): 1, ("newton", "elliptic"): 2}[(solver, cone)], limit_rows=int(limit_rows), lim_par=lim_par,
        condim_downgraded=n_downgraded, arena=int(arena or 0))
