"""Independent reference for the servo dynamics (tests only): joint-space inertia and bias force of the actuated tree from
the UN-lumped compiled model, by a different route than oracle/mopa_oracle_dyn.inc takes.

  M(q)    = sum over every body b of the arm's subtree (welded ones included, each with its own inertial):
                m_b Jv_b^T Jv_b + Jw_b^T (R_b Ic_b R_b^T) Jw_b        (+ armature on the diagonal)
            with Jv_b / Jw_b the geometric Jacobians of the body's centre of mass / orientation, built from a scipy
            composition of the kinematic tree (no shared code with the oracle's FK, no spatial algebra, no lumping);
  bias    = C(q, qd) qd + G(q):  G = -sum_b m_b Jv_b^T g;   C qd = Mdot qd - 1/2 d(qd^T M qd)/dq   by central differences of M.

Agreement is to finite-difference accuracy (1e-6 relative), not bit level: this pins the EQUATIONS the oracle integrates.
"""
import numpy as np
from scipy.spatial.transform import Rotation


def _rot(q):     # wxyz -> matrix
    return Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()


def fk(model, qpos):
    m = model
    nb = len(m.body_names)
    P, R = np.zeros((nb, 3)), np.tile(np.eye(3), (nb, 1, 1))
    for b in range(1, nb):
        pb = int(m.body_parent[b])
        ja, jn = int(m.body_jntadr[b]), int(m.body_jntnum[b])
        if jn == 1 and m.jnt_type[ja] == 0:
            a = int(m.jnt_qposadr[ja])
            P[b], R[b] = qpos[a:a + 3], _rot(qpos[a + 3:a + 7] / np.linalg.norm(qpos[a + 3:a + 7]))
            continue
        p = P[pb] + R[pb] @ m.body_pos[b]
        r = R[pb] @ _rot(m.body_quat[b])
        for j in range(ja, ja + jn):
            dq = qpos[int(m.jnt_qposadr[j])] - m.jnt_ref[j]
            ax = np.asarray(m.jnt_axis[j], dtype=np.float64)
            if m.jnt_type[j] == 2:
                p = p + r @ ax * dq
            else:
                anchor = p + r @ m.jnt_pos[j]
                r = r @ Rotation.from_rotvec(ax * dq).as_matrix()
                p = anchor - r @ m.jnt_pos[j]
        P[b], R[b] = p, r
    return P, R


def subtree_bodies(model, root):
    out = []
    for b in range(1, len(model.body_names)):
        c = b
        while c > 0 and c != root:
            c = int(model.body_parent[c])
        if c == root:
            out.append(b)
    return out


def mass_matrix(model, qpos, dyn_bodies, armature):
    """dyn_bodies: model body ids of the jointed bodies of the tree, in dof order."""
    m = model
    P, R = fk(m, qpos)
    nd = len(dyn_bodies)
    axes, anchors, types = [], [], []
    for b in dyn_bodies:
        j = int(m.body_jntadr[b])
        axes.append(R[b] @ m.jnt_axis[j])
        anchors.append(P[b] + R[b] @ m.jnt_pos[j])
        types.append(int(m.jnt_type[j]))
    M = np.diag(np.asarray(armature, dtype=np.float64))
    G = np.zeros(nd)
    grav = np.asarray(m.opt[:3], dtype=np.float64)
    for b in subtree_bodies(m, dyn_bodies[0]):
        mb = float(m.body_mass[b])
        if mb <= 0.0:
            continue
        c = P[b] + R[b] @ m.body_ipos[b]
        xx, yy, zz, xy, xz, yz = m.body_inertia[b]
        Ic = R[b] @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ R[b].T
        Jv, Jw = np.zeros((3, nd)), np.zeros((3, nd))
        for i, db in enumerate(dyn_bodies):
            a = b                               # dof i moves body b iff db is b or one of its ancestors
            while a > 0 and a != db:
                a = int(m.body_parent[a])
            if a != db:
                continue
            if types[i] == 2:
                Jv[:, i] = axes[i]
            else:
                Jv[:, i] = np.cross(axes[i], c - anchors[i])
                Jw[:, i] = axes[i]
        M += mb * Jv.T @ Jv + Jw.T @ Ic @ Jw
        G -= mb * Jv.T @ grav
    return M, G


def bias_force(model, qpos, qvel, dyn_bodies, qadr, armature, eps=1e-6):
    nd = len(dyn_bodies)
    _, G = mass_matrix(model, qpos, dyn_bodies, armature)
    dM = []
    for k in range(nd):
        qp, qm = np.array(qpos, dtype=np.float64), np.array(qpos, dtype=np.float64)
        qp[qadr[k]] += eps
        qm[qadr[k]] -= eps
        dM.append((mass_matrix(model, qp, dyn_bodies, armature)[0] - mass_matrix(model, qm, dyn_bodies, armature)[0]) / (2 * eps))
    Mdot = sum(dM[k] * qvel[k] for k in range(nd))
    c = Mdot @ qvel - 0.5 * np.array([qvel @ dM[k] @ qvel for k in range(nd)])
    return c + G
