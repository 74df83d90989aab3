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


# ---- stage C: one sub-step with contacts, by an independent route (tests/test_oracle_contact.py) ---------------------------
def _proj_cone(y, mu):
    """Euclidean projection of y = (y_n, y_t...) onto the friction cone |y_t| <= mu y_n (textbook second-order-cone projection; any
    number of tangential components)"""
    t = float(np.linalg.norm(y[1:]))
    if t <= mu * y[0]:
        return y.copy()
    if mu * t <= -y[0]:
        return np.zeros(len(y))
    a = (y[0] + mu * t) / (1.0 + mu * mu)
    return np.concatenate([[a], mu * a * y[1:] / t])


def contact_step_reference(model, dyn, ct, contacts, qpos, qvel, bias_lag, ctrl, limit_rows=False, cone="pyramidal"):
    """New velocities [nd + 6] (arm dofs, then the object's COM velocity and world angular velocity) after ONE sub-step from
    (qpos, qvel), given the oracle's contact list (rows: dist, pos 3, normal 3, shape F, shape S, feature) -- everything else
    independently: geometric Jacobians over scipy FK of the un-lumped model, M by `mass_matrix`, the bias by finite differences,
    the constraint forces as the exact solution (active-set NNLS) of MuJoCo's dual problem
        min_{f >= 0}  1/2 f^T (A + R) f + f^T (J a_smooth - aref),      A = J M^-1 J^T,  R = (1 - imp) / imp diag(A)
    with pyramidal rows J_n +- mu J_t and aref = -B J v - K imp (dist - margin).
    cone="elliptic": rows n, t1, t2 per contact (condim 4: + the relative angular velocity about n, condim 6: + about t1, t2; the pair's
    condim and torsional / rolling friction in ct.pr_par[:, 8:11]), the normal row's regulariser r = (1 - imp) / imp A_nn, the friction
    rows' r_j = r mu_1^2 / mu_j^2 (MuJoCo's rule r_j mu_j^2 = const), their aref without the position term, forces in the ellipsoidal cone
    sum_j (f_j / mu_j)^2 <= f_n^2.  Under the scaling S = diag(1, mu_j / mu_1) that cone is circular with mu_1 and the regulariser
    isotropic, so the dual's per-contact minimiser is a Euclidean cone projection: f = S proj_K(-S (J q - aref)) / r -- which gives the
    PRIMAL form, written from the projection alone:
        min_q  1/2 (q - a_smooth)^T M (q - a_smooth) + sum_c |proj_K(-S_c (J_c q - aref_c))|^2 / (2 r_c)  (+ the limit rows' half-quadratics)
    solved by damped Newton with a finite-difference Hessian (15 unknowns), to a gradient of 1e-12 of its start."""
    from scipy.optimize import nnls
    m = model
    nd, h = dyn.nd, float(dyn.timestep)
    bodies = [int(b) for b in dyn.body]
    P, R = fk(m, qpos)
    M, _ = mass_matrix(m, qpos, bodies, dyn.armature)
    bias = bias_force(m, qpos, qvel[:nd], bodies, [int(a) for a in dyn.qadr], dyn.armature)
    q = qpos[dyn.qadr]
    act = np.where(dyn.actuated == 1, np.clip(dyn.kp * (ctrl - q), dyn.force_lo, dyn.force_hi), 0.0)
    tau = -dyn.damping * qvel[:nd] - bias + np.where(dyn.gravcomp == 1, bias_lag, 0.0) + act
    # the object: dofs = COM velocity (world), angular velocity in the principal frame
    oq = ct.obj_qadr
    Rb = _rot(qpos[oq + 3:oq + 7] / np.linalg.norm(qpos[oq + 3:oq + 7]))
    c = qpos[oq:oq + 3] + Rb @ ct.obj_ipos
    RD = Rb @ _rot(ct.obj_iquat)
    wb = RD.T @ qvel[nd + 3:nd + 6]
    Mo = np.concatenate([[ct.obj_mass] * 3, ct.obj_inertia])
    Fo = ct.obj_mass * np.asarray(m.opt[:3]) - ct.obj_damping * qvel[nd:nd + 3]
    Tb = -ct.obj_damping * wb - np.cross(wb, ct.obj_inertia * wb)
    nv = nd + 6
    Mfull = np.zeros((nv, nv)); Mfull[:nd, :nd] = M; Mfull[nd:, nd:] = np.diag(Mo)
    tfull = np.concatenate([tau, Fo, Tb])
    vfull = np.concatenate([qvel[:nd], qvel[nd:nd + 3], wb])
    axes, anchors, types = [], [], []
    for b in bodies:
        j = int(m.body_jntadr[b])
        axes.append(R[b] @ m.jnt_axis[j]); anchors.append(P[b] + R[b] @ m.jnt_pos[j]); types.append(int(m.jnt_type[j]))

    def point_jac(model_body, pos, d):      # d . velocity of the material point `pos` of a model body, per unit dof velocity
        row = np.zeros(nv)
        on_obj = False
        a = model_body
        while a > 0:
            if a in bodies:
                i = bodies.index(a)
                row[i] = float(d @ (axes[i] if types[i] == 2 else np.cross(axes[i], pos - anchors[i])))
            if int(m.body_jntnum[a]) == 1 and int(m.jnt_type[int(m.body_jntadr[a])]) == 0:
                on_obj = True
            a = int(m.body_parent[a])
        if on_obj:
            row[nd:nd + 3] = d
            row[nd + 3:] = RD.T @ np.cross(pos - c, d)
        return row

    def ang_jac(model_body, d):      # d . angular velocity of a model body, per unit dof velocity
        row = np.zeros(nv)
        on_obj = False
        a = model_body
        while a > 0:
            if a in bodies:
                i = bodies.index(a)
                row[i] = 0.0 if types[i] == 2 else float(d @ axes[i])
            if int(m.body_jntnum[a]) == 1 and int(m.jnt_type[int(m.body_jntadr[a])]) == 0:
                on_obj = True
            a = int(m.body_parent[a])
        if on_obj:
            row[nd + 3:] = RD.T @ d
        return row

    if cone == "elliptic":
        return _contact_step_elliptic(m, dyn, ct, contacts, point_jac, Mfull, tfull, vfull, RD, q, M, limit_rows, ang_jac)
    rows, pars, dists, mus = [], [], [], []
    pair_of = {(int(f), int(s)): k for k, (f, s) in enumerate(zip(ct.pr_f, ct.pr_s))}
    for r in contacts:
        dist, pos, n = r[0], r[1:4], r[4:7]
        sf, ss = int(r[7]), int(r[8])
        par = ct.pr_par[pair_of[(sf, ss)]]
        e = np.array([1.0, 0.0, 0.0]) if abs(n[0]) < 0.5 else np.array([0.0, 1.0, 0.0])
        t1 = e - n * (n @ e); t1 /= np.linalg.norm(t1)
        t2 = np.cross(n, t1)
        bF, bS = int(m.geom_body[int(ct.sh_geom[sf])]), int(m.geom_body[int(ct.sh_geom[ss])])
        Jd = [point_jac(bF, pos, d) - point_jac(bS, pos, d) for d in (n, t1, t2)]
        mu = par[0]
        for t in (1, 2):
            for sg in (1.0, -1.0):
                rows.append(Jd[0] + sg * mu * Jd[t]); pars.append(par); dists.append(dist)
    # joint limits as rows (limit_rows): a limited arm joint beyond its range, J = +-e_l, the joint defaults' solver parameters
    # (ct.lim_par in a pair record's layout), regulariser from 1 / M_ll instead of the row's own A_ii
    n_contact_rows = len(rows)
    lim_diag = []
    if limit_rows:
        for i in range(nd):
            if not dyn.limited[i]:
                continue
            dlo, dhi = q[i] - dyn.lo[i], dyn.hi[i] - q[i]
            side, dist = (1.0, dlo) if dlo < 0 else ((-1.0, dhi) if dhi < 0 else (0.0, 0.0))
            if side == 0.0:
                continue
            row = np.zeros(nv); row[i] = side
            rows.append(row); pars.append(np.asarray(ct.lim_par)); dists.append(dist); lim_diag.append(1.0 / M[i, i])
    if not rows:
        f = np.zeros(0); J = np.zeros((0, nv))
    else:
        J = np.array(rows)
        Minv = np.linalg.inv(Mfull)
        A = J @ Minv @ J.T
        dA = np.diag(A).copy()
        dA[n_contact_rows:] = lim_diag
        imp, aref = np.zeros(len(rows)), np.zeros(len(rows))
        for i, (par, dist) in enumerate(zip(pars, dists)):
            x = abs(dist - par[1]) / par[6]
            y = 1.0 if x >= 1 else (2 * x * x if x <= 0.5 else 1 - 2 * (1 - x) ** 2)
            imp[i] = par[4] + y * (par[5] - par[4])
            aref[i] = -par[3] * (J[i] @ vfull) - par[2] * imp[i] * (dist - par[1])
        Rr = (1 - imp) / imp * dA
        Q = A + np.diag(Rr)
        b = J @ (Minv @ tfull) - aref
        L = np.linalg.cholesky(Q)
        f, _ = nnls(L.T, -np.linalg.solve(L, b), maxiter=100 * len(b))
    Dfull = np.concatenate([dyn.damping, [ct.obj_damping] * 6])
    qacc = np.linalg.solve(Mfull + h * np.diag(Dfull), tfull + J.T @ f)
    vn = vfull + h * qacc
    return np.concatenate([vn[:nd + 3], RD @ vn[nd + 3:]]), f


def _contact_step_elliptic(m, dyn, ct, contacts, point_jac, Mfull, tfull, vfull, RD, q, M, limit_rows, ang_jac):
    nd, h = dyn.nd, float(dyn.timestep)
    nv = nd + 6
    Minv = np.linalg.inv(Mfull)
    a0 = Minv @ tfull
    pair_of = {(int(f), int(s)): k for k, (f, s) in enumerate(zip(ct.pr_f, ct.pr_s))}

    def impedance(par, dist):
        x = abs(dist - par[1]) / par[6]
        y = 1.0 if x >= 1 else (2 * x * x if x <= 0.5 else 1 - 2 * (1 - x) ** 2)
        return par[4] + y * (par[5] - par[4])
    cons = []
    for r in contacts:
        dist, pos, n = r[0], r[1:4], r[4:7]
        sf, ss = int(r[7]), int(r[8])
        par = ct.pr_par[pair_of[(sf, ss)]]
        e = np.array([1.0, 0.0, 0.0]) if abs(n[0]) < 0.5 else np.array([0.0, 1.0, 0.0])
        t1 = e - n * (n @ e); t1 /= np.linalg.norm(t1)
        t2 = np.cross(n, t1)
        bF, bS = int(m.geom_body[int(ct.sh_geom[sf])]), int(m.geom_body[int(ct.sh_geom[ss])])
        dim = int(par[8]) if len(par) > 8 else 3
        rows = [point_jac(bF, pos, d) - point_jac(bS, pos, d) for d in (n, t1, t2)]
        rows += [ang_jac(bF, d) - ang_jac(bS, d) for d in (n, t1, t2)[:dim - 3]]
        J = np.array(rows)
        mus = np.array([par[0], par[0], max(par[9], 1e-5), max(par[10], 1e-5), max(par[10], 1e-5)][:dim - 1]) if dim > 3 else np.array([par[0], par[0]])
        S = np.concatenate([[1.0], mus / par[0]])
        imp = impedance(par, dist)
        rr = (1 - imp) / imp * float(J[0] @ Minv @ J[0])
        jv = J @ vfull
        aref = -par[3] * jv
        aref[0] -= par[2] * imp * (dist - par[1])
        cons.append((J, aref, rr, float(par[0]), S))
    lims = []
    if limit_rows:
        for i in range(nd):
            if not dyn.limited[i]:
                continue
            dlo, dhi = q[i] - dyn.lo[i], dyn.hi[i] - q[i]
            side, dist = (1.0, dlo) if dlo < 0 else ((-1.0, dhi) if dhi < 0 else (0.0, 0.0))
            if side == 0.0:
                continue
            par = np.asarray(ct.lim_par)
            imp = impedance(par, dist)
            rr = (1 - imp) / imp / M[i, i]
            lims.append((i, side, -par[3] * side * vfull[i] - par[2] * imp * dist, 1.0 / rr))

    def forces(qa):
        return [S * _proj_cone(-(S * (J @ qa - aref)), mu) / rr for J, aref, rr, mu, S in cons]

    def grad(qa):
        g = Mfull @ (qa - a0)
        for (J, aref, rr, mu, S), f in zip(cons, forces(qa)):
            g = g - J.T @ f
        for i, side, aref, D in lims:
            x = side * qa[i] - aref
            if x < 0:
                g[i] += side * D * x
        return g

    def cost(qa):
        c = 0.5 * (qa - a0) @ Mfull @ (qa - a0)
        for (J, aref, rr, mu, S), f in zip(cons, forces(qa)):
            c += 0.5 * rr * float((f / S) @ (f / S))
        for i, side, aref, D in lims:
            x = side * qa[i] - aref
            if x < 0:
                c += 0.5 * D * x * x
        return c
    qa = a0.copy()
    g0 = max(np.linalg.norm(grad(qa)), 1e-30)
    for _ in range(200):
        g = grad(qa)
        if np.linalg.norm(g) <= 1e-12 * g0:
            break
        eps = 1e-7 * max(1.0, np.abs(qa).max())
        H = np.array([(grad(qa + eps * np.eye(nv)[k]) - grad(qa - eps * np.eye(nv)[k])) / (2 * eps) for k in range(nv)])
        H = 0.5 * (H + H.T)
        step = -np.linalg.solve(H + 1e-12 * np.trace(H) / nv * np.eye(nv), g)
        t, c0 = 1.0, cost(qa)
        while t > 1e-8 and cost(qa + t * step) > c0 + 1e-4 * t * float(g @ step):
            t *= 0.5
        qa = qa + t * step
    f = np.concatenate(forces(qa)) if cons else np.zeros(0)
    Jt = sum((J.T @ fc for (J, _, _, _, _), fc in zip(cons, forces(qa))), np.zeros(nv))
    for i, side, aref, D in lims:
        x = side * qa[i] - aref
        if x < 0:
            Jt[i] += side * (-D * x)
    Dfull = np.concatenate([dyn.damping, [ct.obj_damping] * 6])
    qacc = np.linalg.solve(Mfull + h * np.diag(Dfull), tfull + Jt)
    vn = vfull + h * qacc
    return np.concatenate([vn[:nd + 3], RD @ vn[nd + 3:]]), f
