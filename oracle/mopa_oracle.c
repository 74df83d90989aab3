/*
 * mopa_oracle.c -- plain-C, double-precision, scalar CPU restatement of the
 * MoPA-RL state-validity / motion-validation / RRT-Connect path.
 *
 * TEST INFRASTRUCTURE ONLY (see mopa_oracle.h).  PARITY UNPINNED against
 * MuJoCo 2.0 / OMPL: neither library nor any golden vector exists in
 * /root/reference or in this image.
 *
 * What it follows (reference = /root/reference, [3P] = third-party library
 * whose source is not in the tree; restated from its published algorithm):
 *
 *   validity rule ......... motion_planners/src/mujoco_ompl_interface.cpp:909-978
 *   active/passive split .. motion_planners/KinematicPlanner.cpp:253-286,
 *                           mujoco_ompl_interface.cpp:369-488,778-785
 *   forward kinematics .... [3P] MuJoCo mj_kinematics (called through
 *                           mj_fwdPosition at mujoco_ompl_interface.cpp:932)
 *   collision ............. [3P] MuJoCo mj_collision: bounding-sphere broad
 *                           phase + per-pair narrow phase; closed forms for
 *                           plane-x, sphere-x, capsule-capsule, capsule-box;
 *                           15-axis SAT for box-box; libccd MPR
 *                           (ccdMPRPenetration, tolerance 1e-6, 50 iterations
 *                           = MuJoCo's mpr_tolerance / mpr_iterations) for
 *                           {capsule,cylinder,box}-cylinder, as MuJoCo 2.0
 *                           routes every such pair through mjc_Convex.
 *   motion validation ..... [3P] OMPL DiscreteMotionValidator, resolution set
 *                           at motion_planners/KinematicPlanner.cpp:87
 *   RRT-Connect ........... [3P] OMPL geometric::RRTConnect, set up at
 *                           KinematicPlanner.cpp:90,102-104, run at :188;
 *                           sentinels -5 / -4 as KinematicPlanner.cpp:181-184,249-250
 *   state space ........... mujoco_ompl_interface.cpp:149-281 (limited hinge
 *                           -> R^1 with the joint range, weight 1 => L1 metric)
 *
 * Numerics contract shared with the HIP kernels (so verdicts are
 * bit-identical): IEEE double, no implicit contraction (-ffp-contract=off),
 * every fused multiply-add is an explicit fma(), sqrt and / are IEEE
 * correctly rounded, sin/cos are orc_sincos() below (never libm).
 *
 * Deliberate deviations from MuJoCo (documented in DESIGN.md):
 *   - geom `margin` is ignored: contact_threshold < 0 so only penetration of
 *     at least |threshold| can invalidate a state;
 *   - the broad phase culls with zero margin for the same reason;
 *   - sphere-cylinder is analytic (MuJoCo 2.0 used MPR, later releases made
 *     it analytic);
 *   - capsule-box is the exact segment/box distance (MuJoCo uses a
 *     closest-feature heuristic that agrees for shallow contact);
 *   - box-box returns the SAT minimum-translation depth (= MuJoCo's deepest
 *     contact for face and edge contacts);
 *   - the planner stops on an iteration budget, not wall-clock, and draws
 *     samples from a counter-based RNG instead of std::mt19937.
 */
#include "mopa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { G_PLANE = 0, G_HFIELD = 1, G_SPHERE = 2, G_CAPSULE = 3, G_ELLIPSOID = 4, G_CYLINDER = 5, G_BOX = 6, G_MESH = 7 };
enum { J_FREE = 0, J_BALL = 1, J_SLIDE = 2, J_HINGE = 3 };

#define MINVAL 1e-15 /* mjMINVAL */
#define ORC_PI 3.14159265358979323846

struct OrcScene {
    int nq, nbody, njnt, ngeom, npair, na;
    int32_t *body_parent, *body_jntadr, *body_jntnum;
    double *body_pos, *body_quat;
    int32_t *jnt_type, *jnt_qposadr, *jnt_limited;
    double *jnt_axis, *jnt_pos, *jnt_ref, *jnt_range;
    int32_t *geom_type, *geom_body, *geom_mjid;
    double *geom_size, *geom_pos, *geom_quat, *geom_rbound;
    int32_t *pair_geom;
    uint8_t *pair_ignored;
    int32_t *active_idx;   /* qpos addresses planned over */
    double *act_lo, *act_hi, *act_ext;
    uint8_t *act_so2;  /* unlimited hinge -> OMPL SO2StateSpace (mujoco_ompl_interface.cpp:234-239) */
    uint8_t *body_needed;  /* body has a collidable geom below it */
    uint8_t *geom_static;  /* no joint between the geom's body and the world */
    double *geom_aabb;     /* [ngeom,3] world-AABB half extents of static geoms (second-stage cull) */
    int nmesh, nmeshvert;
    int32_t *mesh_vertadr, *mesh_vertnum, *geom_dataid;   /* convex hulls of mesh geoms; geom_dataid NULL = no meshes */
    double *mesh_vert;
    double thr;
};

/* ------------------------------------------------------------------ */
/* small vector helpers -- the expression order here IS the spec       */
/* ------------------------------------------------------------------ */
static inline double dot3(const double *a, const double *b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
static inline void cross3(double *r, const double *a, const double *b) {
    double r0 = fma(a[1], b[2], -(a[2] * b[1]));
    double r1 = fma(a[2], b[0], -(a[0] * b[2]));
    double r2 = fma(a[0], b[1], -(a[1] * b[0]));
    r[0] = r0; r[1] = r1; r[2] = r2;
}
static inline void sub3(double *r, const double *a, const double *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void add3(double *r, const double *a, const double *b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
/* r = a + b*s */
static inline void addscl3(double *r, const double *a, const double *b, double s) {
    r[0] = fma(b[0], s, a[0]); r[1] = fma(b[1], s, a[1]); r[2] = fma(b[2], s, a[2]);
}
static inline double norm3(const double *a) { return sqrt(dot3(a, a)); }
/* r = M v  (row-major 3x3) */
static inline void mat_vec(double *r, const double *M, const double *v) {
    double r0 = dot3(M, v), r1 = dot3(M + 3, v), r2 = dot3(M + 6, v);
    r[0] = r0; r[1] = r1; r[2] = r2;
}
/* r = M^T v */
static inline void matT_vec(double *r, const double *M, const double *v) {
    double r0 = fma(M[6], v[2], fma(M[3], v[1], M[0] * v[0]));
    double r1 = fma(M[7], v[2], fma(M[4], v[1], M[1] * v[0]));
    double r2 = fma(M[8], v[2], fma(M[5], v[1], M[2] * v[0]));
    r[0] = r0; r[1] = r1; r[2] = r2;
}
static inline void col3(double *r, const double *M, int j) { r[0] = M[j]; r[1] = M[3 + j]; r[2] = M[6 + j]; }
static inline double dmin(double a, double b) { return (a < b) ? a : b; }
static inline double dmax(double a, double b) { return (a > b) ? a : b; }
static inline double clampd(double x, double lo, double hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); }
static inline double signd(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); }

static inline void quat_mul(double *r, const double *a, const double *b) {
    double r0 = fma(-a[3], b[3], fma(-a[2], b[2], fma(-a[1], b[1], a[0] * b[0])));
    double r1 = fma(-a[3], b[2], fma(a[2], b[3], fma(a[1], b[0], a[0] * b[1])));
    double r2 = fma(a[3], b[1], fma(a[2], b[0], fma(-a[1], b[3], a[0] * b[2])));
    double r3 = fma(a[3], b[0], fma(-a[2], b[1], fma(a[1], b[2], a[0] * b[3])));
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
}
/* [3P] mju_normalize4 */
static inline void quat_normalize(double *q) {
    double n = sqrt(fma(q[3], q[3], fma(q[2], q[2], fma(q[1], q[1], q[0] * q[0]))));
    if (n < MINVAL) { q[0] = 1.0; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0; }
    else if (fabs(n - 1.0) > MINVAL) {
        double inv = 1.0 / n;
        q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
    }
}
/* [3P] mju_quat2Mat */
static inline void quat2mat(double *M, const double *q) {
    double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
    double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
    M[0] = ((q00 + q11) - q22) - q33;
    M[4] = ((q00 - q11) + q22) - q33;
    M[8] = ((q00 - q11) - q22) + q33;
    M[1] = 2.0 * (q12 - q03);
    M[2] = 2.0 * (q13 + q02);
    M[3] = 2.0 * (q12 + q03);
    M[5] = 2.0 * (q23 - q01);
    M[6] = 2.0 * (q13 - q02);
    M[7] = 2.0 * (q23 + q01);
}
/* [3P] mju_rotVecQuat: rotate through the matrix of the quaternion */
static inline void rot_vec_quat(double *r, const double *v, const double *q) {
    double M[9];
    quat2mat(M, q);
    mat_vec(r, M, v);
}

/* ------------------------------------------------------------------ */
/* deterministic sin/cos: Cody-Waite reduction by pi/2 + fdlibm-style   */
/* minimax kernels, every operation an explicit IEEE op.                */
/* ------------------------------------------------------------------ */
void orc_sincos(double x, double *sout, double *cout) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double P2 = 6.07710050630396597660e-11; /* next 33 bits */
    const double P3 = 2.02226624879595063154e-21; /* tail */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double k = rint(x * TWO_OVER_PI);
    double r = fma(-k, P1, x);
    r = fma(-k, P2, r);
    r = fma(-k, P3, r);
    double z = r * r;
    double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
    double sn = fma(r * z, ps, r);
    double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    long long q = (long long)k;
    switch ((int)(q & 3)) {
        case 0: *sout = sn; *cout = cs; break;
        case 1: *sout = cs; *cout = -sn; break;
        case 2: *sout = -sn; *cout = -cs; break;
        default: *sout = -cs; *cout = sn; break;
    }
}

/* deterministic exp / tanh shared (as a specification) with the HIP env kernel: see mopa_device.hpp */
double orc_exp(double x) {
    const double LOG2E = 1.44269504088896338700e+00, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double k = rint(x * LOG2E);
    double r = fma(-k, LN2_HI, x);
    r = fma(-k, LN2_LO, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    union { uint64_t u; double d; } sc;
    sc.u = (uint64_t)((long long)k + 1023) << 52;
    return p * sc.d;
}
double orc_tanh_pos(double x) {
    const double t = orc_exp(-2.0 * x);
    return (1.0 - t) / (1.0 + t);
}

/* deterministic atan2 shared (as a specification) with the HIP IK kernel: a = min/max in [0,1]; reduction
 * atan(a) = atan(c) + atan((a - c) / (1 + a c)) with c in {0, tan(pi/8), 1} so that |t| <= tan(pi/16); odd Taylor series of
 * atan(t) to t^25 (|t|^27/27 < 5e-21), Horner in t^2; quadrant fix-ups.  atan2(0, 0) = 0. */
double orc_atan2(double y, double x) {
    const double PI = 3.14159265358979311600e+00, PI_2 = 1.57079632679489655800e+00;
    const double T1 = 1.98912367379658006912e-01 /* tan(pi/16) */, T3 = 6.68178637919298919998e-01 /* tan(3pi/16) */;
    const double C1 = 4.14213562373095048802e-01 /* tan(pi/8) */, A1 = 3.92699081698724139500e-01 /* pi/8 */,
                 A2 = 7.85398163397448279000e-01 /* pi/4 */;
    const double ax = fabs(x), ay = fabs(y);
    const double hi = (ax > ay) ? ax : ay, lo = (ax > ay) ? ay : ax;
    double r = 0.0;
    if (hi > 0.0) {
        const double a = lo / hi;
        double c = 0.0, base = 0.0;
        if (a >= T3) { c = 1.0; base = A2; }
        else if (a >= T1) { c = C1; base = A1; }
        const double t = (a - c) / fma(a, c, 1.0);
        const double z = t * t;
        double p = 1.0 / 25.0;
        p = fma(p, z, -1.0 / 23.0);
        p = fma(p, z, 1.0 / 21.0);
        p = fma(p, z, -1.0 / 19.0);
        p = fma(p, z, 1.0 / 17.0);
        p = fma(p, z, -1.0 / 15.0);
        p = fma(p, z, 1.0 / 13.0);
        p = fma(p, z, -1.0 / 11.0);
        p = fma(p, z, 1.0 / 9.0);
        p = fma(p, z, -1.0 / 7.0);
        p = fma(p, z, 1.0 / 5.0);
        p = fma(p, z, -1.0 / 3.0);
        r = base + fma(t * z, p, t);
        if (ay > ax) r = PI_2 - r;
    }
    if (x < 0.0) r = PI - r;
    return (y < 0.0) ? -r : r;
}

/* [3P] MuJoCo's quaternion helpers the IK calls through dm_control (env/inverse_kinematics.py:88-92), restated from the
 * published engine_util_spatial.c: mju_mat2Quat (largest-component branch, then normalisation), mju_negQuat (conjugate),
 * mju_mulQuat, mju_quat2Vel (axis * angle, angle in (-pi, pi], divided by dt = 1). */
static void mj_mat2quat(double *q, const double *m) {
    if (m[0] + m[4] + m[8] > 0.0) {
        q[0] = 0.5 * sqrt(1.0 + m[0] + m[4] + m[8]);
        q[1] = 0.25 * (m[7] - m[5]) / q[0];
        q[2] = 0.25 * (m[2] - m[6]) / q[0];
        q[3] = 0.25 * (m[3] - m[1]) / q[0];
    } else if (m[0] > m[4] && m[0] > m[8]) {
        q[1] = 0.5 * sqrt(1.0 + m[0] - m[4] - m[8]);
        q[0] = 0.25 * (m[7] - m[5]) / q[1];
        q[2] = 0.25 * (m[1] + m[3]) / q[1];
        q[3] = 0.25 * (m[2] + m[6]) / q[1];
    } else if (m[4] > m[8]) {
        q[2] = 0.5 * sqrt(1.0 - m[0] + m[4] - m[8]);
        q[0] = 0.25 * (m[2] - m[6]) / q[2];
        q[1] = 0.25 * (m[1] + m[3]) / q[2];
        q[3] = 0.25 * (m[5] + m[7]) / q[2];
    } else {
        q[3] = 0.5 * sqrt(1.0 - m[0] - m[4] + m[8]);
        q[0] = 0.25 * (m[3] - m[1]) / q[3];
        q[1] = 0.25 * (m[2] + m[6]) / q[3];
        q[2] = 0.25 * (m[5] + m[7]) / q[3];
    }
    quat_normalize(q);     /* [3P] mju_normalize4 */
}
static void mj_quat2vel(double *res, const double *q) {
    double ax[3] = {q[1], q[2], q[3]};
    const double s = sqrt(fma(ax[2], ax[2], fma(ax[1], ax[1], ax[0] * ax[0])));
    if (s < 1e-15) { ax[0] = 1.0; ax[1] = 0.0; ax[2] = 0.0; }        /* [3P] mju_normalize3 */
    else if (fabs(s - 1.0) > 1e-15) { const double inv = 1.0 / s; ax[0] *= inv; ax[1] *= inv; ax[2] *= inv; }
    double speed = 2.0 * orc_atan2(s, q[0]);
    if (speed > 3.14159265358979311600e+00) speed = speed - 2.0 * 3.14159265358979311600e+00;
    res[0] = ax[0] * speed; res[1] = ax[1] * speed; res[2] = ax[2] * speed;
}

/* ------------------------------------------------------------------ */
/* scene                                                               */
/* ------------------------------------------------------------------ */
static void *dupmem(const void *p, size_t n) {
    void *r = malloc(n ? n : 1);
    if (n) memcpy(r, p, n);
    return r;
}

static void scene_static_aabbs(OrcScene *s);

OrcScene *orc_scene_create(
    int nq, int nbody, const int32_t *body_parent, const double *body_pos, const double *body_quat,
    const int32_t *body_jntadr, const int32_t *body_jntnum,
    int njnt, const int32_t *jnt_type, const int32_t *jnt_qposadr, const double *jnt_axis,
    const double *jnt_pos, const double *jnt_ref, const int32_t *jnt_limited, const double *jnt_range,
    int ngeom, const int32_t *geom_type, const int32_t *geom_body, const int32_t *geom_mjid,
    const double *geom_size, const double *geom_pos, const double *geom_quat,
    int npair, const int32_t *pair_geom,
    int n_passive, const int32_t *passive_qpos_idx,
    int n_ignored, const int32_t *ignored_pairs, double contact_threshold) {
    OrcScene *s = (OrcScene *)calloc(1, sizeof(OrcScene));
    s->nq = nq; s->nbody = nbody; s->njnt = njnt; s->ngeom = ngeom; s->npair = npair;
    s->thr = contact_threshold;
    s->body_parent = dupmem(body_parent, sizeof(int32_t) * nbody);
    s->body_pos = dupmem(body_pos, sizeof(double) * 3 * nbody);
    s->body_quat = dupmem(body_quat, sizeof(double) * 4 * nbody);
    s->body_jntadr = dupmem(body_jntadr, sizeof(int32_t) * nbody);
    s->body_jntnum = dupmem(body_jntnum, sizeof(int32_t) * nbody);
    s->jnt_type = dupmem(jnt_type, sizeof(int32_t) * njnt);
    s->jnt_qposadr = dupmem(jnt_qposadr, sizeof(int32_t) * njnt);
    s->jnt_limited = dupmem(jnt_limited, sizeof(int32_t) * njnt);
    s->jnt_axis = dupmem(jnt_axis, sizeof(double) * 3 * njnt);
    s->jnt_pos = dupmem(jnt_pos, sizeof(double) * 3 * njnt);
    s->jnt_ref = dupmem(jnt_ref, sizeof(double) * njnt);
    s->jnt_range = dupmem(jnt_range, sizeof(double) * 2 * njnt);
    s->geom_type = dupmem(geom_type, sizeof(int32_t) * ngeom);
    s->geom_body = dupmem(geom_body, sizeof(int32_t) * ngeom);
    s->geom_mjid = dupmem(geom_mjid, sizeof(int32_t) * ngeom);
    s->geom_size = dupmem(geom_size, sizeof(double) * 3 * ngeom);
    s->geom_pos = dupmem(geom_pos, sizeof(double) * 3 * ngeom);
    s->geom_quat = dupmem(geom_quat, sizeof(double) * 4 * ngeom);
    s->pair_geom = dupmem(pair_geom, sizeof(int32_t) * 2 * npair);
    /* bounding radii ([3P] geom_rbound) */
    s->geom_rbound = (double *)calloc(ngeom ? ngeom : 1, sizeof(double));
    for (int g = 0; g < ngeom; g++) {
        const double *sz = geom_size + 3 * g;
        switch (geom_type[g]) {
            case G_SPHERE: s->geom_rbound[g] = sz[0]; break;
            case G_CAPSULE: s->geom_rbound[g] = sz[0] + sz[1]; break;
            case G_CYLINDER: s->geom_rbound[g] = sqrt(fma(sz[1], sz[1], sz[0] * sz[0])); break;
            case G_BOX: s->geom_rbound[g] = sqrt(fma(sz[2], sz[2], fma(sz[1], sz[1], sz[0] * sz[0]))); break;
            default: s->geom_rbound[g] = 0.0; break;
        }
    }
    /* ignored pairs: ordered MuJoCo geom ids (mujoco_ompl_interface.cpp:952) */
    s->pair_ignored = (uint8_t *)calloc(npair ? npair : 1, 1);
    for (int p = 0; p < npair; p++) {
        int a = geom_mjid[pair_geom[2 * p]], b = geom_mjid[pair_geom[2 * p + 1]];
        int lo = a < b ? a : b, hi = a < b ? b : a;
        for (int i = 0; i < n_ignored; i++)
            if (ignored_pairs[2 * i] == lo && ignored_pairs[2 * i + 1] == hi) s->pair_ignored[p] = 1;
    }
    /* active qpos addresses = all minus passive (KinematicPlanner.cpp:263-269) */
    s->active_idx = (int32_t *)calloc(nq ? nq : 1, sizeof(int32_t));
    s->act_lo = (double *)calloc(nq ? nq : 1, sizeof(double));
    s->act_hi = (double *)calloc(nq ? nq : 1, sizeof(double));
    s->act_ext = (double *)calloc(nq ? nq : 1, sizeof(double));
    s->act_so2 = (uint8_t *)calloc(nq ? nq : 1, 1);
    s->na = 0;
    for (int i = 0; i < nq; i++) {
        int passive = 0;
        for (int k = 0; k < n_passive; k++) if (passive_qpos_idx[k] == i) passive = 1;
        if (!passive) {
            int a = s->na++;
            s->active_idx[a] = i;
            for (int j = 0; j < njnt; j++)
                if (jnt_qposadr[j] == i) {
                    if (jnt_type[j] == J_HINGE && !jnt_limited[j]) {
                        /* SO2: samples in [-pi, pi], maximum extent pi */
                        s->act_so2[a] = 1; s->act_lo[a] = -ORC_PI; s->act_hi[a] = ORC_PI; s->act_ext[a] = ORC_PI;
                    } else {
                        s->act_lo[a] = jnt_range[2 * j]; s->act_hi[a] = jnt_range[2 * j + 1];
                        s->act_ext[a] = s->act_hi[a] - s->act_lo[a];
                    }
                }
        }
    }
    s->body_needed = (uint8_t *)calloc(nbody, 1);
    for (int g = 0; g < ngeom; g++) {
        int b = geom_body[g];
        while (b > 0 && !s->body_needed[b]) { s->body_needed[b] = 1; b = body_parent[b]; }
    }
    s->geom_static = (uint8_t *)calloc(ngeom ? ngeom : 1, 1);
    s->geom_aabb = (double *)calloc(ngeom ? 3 * ngeom : 1, sizeof(double));
    scene_static_aabbs(s);
    return s;
}

/* convex hulls of the mesh geoms (mopa_rl_amd.mjcf: geom_dataid / mesh_vertadr / mesh_vertnum / mesh_vert) */
void orc_scene_set_meshes(OrcScene *s, int nmesh, const int32_t *mesh_vertadr, const int32_t *mesh_vertnum, int nmeshvert,
                          const double *mesh_vert, const int32_t *geom_dataid) {
    s->nmesh = nmesh; s->nmeshvert = nmeshvert;
    s->mesh_vertadr = dupmem(mesh_vertadr, sizeof(int32_t) * nmesh);
    s->mesh_vertnum = dupmem(mesh_vertnum, sizeof(int32_t) * nmesh);
    s->mesh_vert = dupmem(mesh_vert, sizeof(double) * 3 * nmeshvert);
    s->geom_dataid = dupmem(geom_dataid, sizeof(int32_t) * s->ngeom);
    for (int g = 0; g < s->ngeom; g++) {
        if (s->geom_type[g] != G_MESH || geom_dataid[g] < 0) continue;
        /* bounding radius about the geom origin: max |v| over the hull */
        double r2 = 0.0;
        const double *V = s->mesh_vert + 3 * mesh_vertadr[geom_dataid[g]];
        for (int i = 0; i < mesh_vertnum[geom_dataid[g]]; i++) r2 = dmax(r2, dot3(V + 3 * i, V + 3 * i));
        s->geom_rbound[g] = sqrt(r2);
    }
    scene_static_aabbs(s);
}

void orc_scene_destroy(OrcScene *s) {
    if (!s) return;
    free(s->body_parent); free(s->body_pos); free(s->body_quat); free(s->body_jntadr); free(s->body_jntnum);
    free(s->jnt_type); free(s->jnt_qposadr); free(s->jnt_limited); free(s->jnt_axis); free(s->jnt_pos);
    free(s->jnt_ref); free(s->jnt_range); free(s->geom_type); free(s->geom_body); free(s->geom_mjid);
    free(s->geom_size); free(s->geom_pos); free(s->geom_quat); free(s->geom_rbound); free(s->pair_geom);
    free(s->mesh_vertadr); free(s->mesh_vertnum); free(s->mesh_vert); free(s->geom_dataid);
    free(s->geom_static); free(s->geom_aabb);
    free(s->pair_ignored); free(s->active_idx); free(s->act_lo); free(s->act_hi); free(s->act_ext); free(s->act_so2);
    free(s->body_needed);
    free(s);
}
int orc_num_active(const OrcScene *s) { return s->na; }
void orc_active_idx(const OrcScene *s, int32_t *out) { memcpy(out, s->active_idx, sizeof(int32_t) * s->na); }

/* ------------------------------------------------------------------ */
/* forward kinematics -- [3P] MuJoCo mj_kinematics restated            */
/* ------------------------------------------------------------------ */
static void fk_bodies(const OrcScene *s, const double *qpos, double *xpos, double *xquat, double *xmat, int only_needed) {
    xpos[0] = xpos[1] = xpos[2] = 0.0;
    xquat[0] = 1.0; xquat[1] = xquat[2] = xquat[3] = 0.0;
    quat2mat(xmat, xquat);
    for (int b = 1; b < s->nbody; b++) {
        if (only_needed && !s->body_needed[b]) continue;
        double *p = xpos + 3 * b, *q = xquat + 4 * b;
        int ja = s->body_jntadr[b], jn = s->body_jntnum[b];
        if (jn == 1 && s->jnt_type[ja] == J_FREE) {
            const double *qp = qpos + s->jnt_qposadr[ja];
            p[0] = qp[0]; p[1] = qp[1]; p[2] = qp[2];
            q[0] = qp[3]; q[1] = qp[4]; q[2] = qp[5]; q[3] = qp[6];
            quat_normalize(q);
        } else {
            int pid = s->body_parent[b];
            double v[3];
            mat_vec(v, xmat + 9 * pid, s->body_pos + 3 * b);
            add3(p, xpos + 3 * pid, v);
            quat_mul(q, xquat + 4 * pid, s->body_quat + 4 * b);
            for (int j = ja; j < ja + jn; j++) {
                const double *ax = s->jnt_axis + 3 * j, *jp = s->jnt_pos + 3 * j;
                double dq = qpos[s->jnt_qposadr[j]] - s->jnt_ref[j];
                if (s->jnt_type[j] == J_SLIDE) {
                    double xaxis[3];
                    rot_vec_quat(xaxis, ax, q);
                    addscl3(p, p, xaxis, dq);
                } else if (s->jnt_type[j] == J_HINGE) {
                    double sn, cs, ql[4], qt[4];
                    orc_sincos(0.5 * dq, &sn, &cs);
                    ql[0] = cs; ql[1] = ax[0] * sn; ql[2] = ax[1] * sn; ql[3] = ax[2] * sn;
                    if (jp[0] == 0.0 && jp[1] == 0.0 && jp[2] == 0.0) {
                        /* anchor at the body origin (every joint of the reference's robots): the off-centre
                         * correction below is the identity up to the sign of a zero -- skipped, as in the kernels */
                        quat_mul(qt, q, ql);
                        q[0] = qt[0]; q[1] = qt[1]; q[2] = qt[2]; q[3] = qt[3];
                    } else {
                        double xanchor[3], vec[3];
                        rot_vec_quat(xanchor, jp, q);
                        add3(xanchor, xanchor, p);
                        quat_mul(qt, q, ql);
                        q[0] = qt[0]; q[1] = qt[1]; q[2] = qt[2]; q[3] = qt[3];
                        /* correct for off-center rotation */
                        rot_vec_quat(vec, jp, q);
                        sub3(p, xanchor, vec);
                    }
                }
                /* ball / extra free joints: not present in the supported scenes */
            }
            quat_normalize(q);
        }
        quat2mat(xmat + 9 * b, q);
    }
}

static void fk_geoms(const OrcScene *s, const double *xpos, const double *xquat, const double *xmat,
                     double *gpos, double *gmat) {
    for (int g = 0; g < s->ngeom; g++) {
        int b = s->geom_body[g];
        double v[3], q[4];
        mat_vec(v, xmat + 9 * b, s->geom_pos + 3 * g);
        add3(gpos + 3 * g, xpos + 3 * b, v);
        quat_mul(q, xquat + 4 * b, s->geom_quat + 4 * g);
        quat2mat(gmat + 9 * g, q);
    }
}

void orc_fk_bodies(const OrcScene *s, const double *qpos, double *xpos, double *xquat) {
    double *xmat = (double *)malloc(sizeof(double) * 9 * s->nbody);
    fk_bodies(s, qpos, xpos, xquat, xmat, 0);
    free(xmat);
}

void orc_fk(const OrcScene *s, const double *qpos, double *gpos, double *gmat) {
    double *buf = (double *)malloc(sizeof(double) * 16 * s->nbody);
    double *xpos = buf, *xquat = buf + 3 * s->nbody, *xmat = buf + 7 * s->nbody;
    fk_bodies(s, qpos, xpos, xquat, xmat, 1);
    fk_geoms(s, xpos, xquat, xmat, gpos, gmat);
    free(buf);
}

/* world-AABB half extents of every static geom (no joint on the way to the world), for the second broad-phase
 * stage; conservative, + 1e-9 so that rounding can never cut into the shape.  Same expressions as
 * static_aabb_half() in the kernels' header. */
static void scene_static_aabbs(OrcScene *s) {
    double *q0 = (double *)calloc(s->nq ? s->nq : 1, sizeof(double));
    double *buf = (double *)malloc(sizeof(double) * (16 * (size_t)s->nbody + 12 * (size_t)s->ngeom));
    double *xpos = buf, *xquat = buf + 3 * s->nbody, *xmat = buf + 7 * s->nbody;
    double *gpos = buf + 16 * s->nbody, *gmat = gpos + 3 * s->ngeom;
    fk_bodies(s, q0, xpos, xquat, xmat, 1);      /* static bodies do not read qpos */
    fk_geoms(s, xpos, xquat, xmat, gpos, gmat);
    for (int g = 0; g < s->ngeom; g++) {
        int b = s->geom_body[g], st = 1;
        for (; b > 0; b = s->body_parent[b]) if (s->body_jntnum[b] > 0) st = 0;
        s->geom_static[g] = (uint8_t)st;
        double *H = s->geom_aabb + 3 * g;
        H[0] = H[1] = H[2] = 0.0;
        if (!st || s->geom_type[g] == G_PLANE) continue;
        const double *M = gmat + 9 * g, *sz = s->geom_size + 3 * g;
        for (int i = 0; i < 3; i++) {
            double a0 = fabs(M[3 * i]), a1 = fabs(M[3 * i + 1]), a2 = fabs(M[3 * i + 2]), hv;
            switch (s->geom_type[g]) {
                case G_BOX: hv = fma(a2, sz[2], fma(a1, sz[1], a0 * sz[0])); break;
                case G_SPHERE: hv = sz[0]; break;
                case G_CAPSULE: hv = fma(a2, sz[1], sz[0]); break;
                case G_CYLINDER: { double s2 = fma(-a2, a2, 1.0); hv = fma(a2, sz[1], sz[0] * ((s2 > 0.0) ? sqrt(s2) : 0.0)); } break;
                default: hv = s->geom_rbound[g]; break;
            }
            H[i] = hv + 1e-9;
        }
    }
    free(buf); free(q0);
}

/* ------------------------------------------------------------------ */
/* narrow phase                                                        */
/* ------------------------------------------------------------------ */
typedef struct { int type; const double *size, *pos, *mat; const double *verts; int nvert; } Geom;   /* verts: convex-hull vertices of a mesh geom (geom frame) */

/* [3P] mjc_PlaneSphere */
static double d_plane_sphere(const Geom *P, const Geom *S) {
    double n[3], diff[3];
    col3(n, P->mat, 2);
    sub3(diff, S->pos, P->pos);
    return dot3(diff, n) - S->size[0];
}
/* [3P] mjc_PlaneCapsule: the two end spheres */
static double d_plane_capsule(const Geom *P, const Geom *C) {
    double n[3], a[3], e[3], diff[3];
    col3(n, P->mat, 2);
    col3(a, C->mat, 2);
    addscl3(e, C->pos, a, C->size[1]);
    sub3(diff, e, P->pos);
    double d1 = dot3(diff, n) - C->size[0];
    addscl3(e, C->pos, a, -C->size[1]);
    sub3(diff, e, P->pos);
    double d2 = dot3(diff, n) - C->size[0];
    return dmin(d1, d2);
}
/* [3P] mjc_PlaneCylinder: deepest rim point = support function along -n */
static double d_plane_cylinder(const Geom *P, const Geom *C) {
    double n[3], a[3], diff[3];
    col3(n, P->mat, 2);
    col3(a, C->mat, 2);
    sub3(diff, C->pos, P->pos);
    double d0 = dot3(diff, n);
    double na = dot3(n, a);
    double s2 = fma(-na, na, 1.0);
    double sr = (s2 > 0.0) ? sqrt(s2) : 0.0;
    return (d0 - C->size[1] * fabs(na)) - C->size[0] * sr;
}
/* [3P] mjc_PlaneConvex for a mesh: the hull vertex that is deepest along -n (first one on ties) */
static double d_plane_mesh(const Geom *P, const Geom *M) {
    double n[3], ln[3], w[3], diff[3];
    col3(n, P->mat, 2);
    matT_vec(ln, M->mat, n);
    int best = 0;
    double bd = dot3(ln, M->verts);
    for (int i = 1; i < M->nvert; i++) {
        double d = dot3(ln, M->verts + 3 * i);
        if (d < bd) { bd = d; best = i; }
    }
    mat_vec(w, M->mat, M->verts + 3 * best);
    add3(w, w, M->pos);
    sub3(diff, w, P->pos);
    return dot3(diff, n);
}
/* [3P] mjc_PlaneBox: deepest vertex */
static double d_plane_box(const Geom *P, const Geom *B) {
    double n[3], u[3], diff[3];
    col3(n, P->mat, 2);
    sub3(diff, B->pos, P->pos);
    double d0 = dot3(diff, n);
    double ext = 0.0;
    for (int i = 0; i < 3; i++) {
        col3(u, B->mat, i);
        ext = fma(fabs(dot3(n, u)), B->size[i], ext);
    }
    return d0 - ext;
}
/* [3P] mjc_SphereSphere */
static double d_sphere_sphere(const Geom *A, const Geom *B) {
    double diff[3];
    sub3(diff, B->pos, A->pos);
    return norm3(diff) - (A->size[0] + B->size[0]);
}
/* [3P] mjc_SphereCapsule: nearest point on the capsule axis */
static double d_sphere_capsule(const Geom *S, const Geom *C) {
    double a[3], vec[3], pt[3], diff[3];
    col3(a, C->mat, 2);
    sub3(vec, S->pos, C->pos);
    double x = clampd(dot3(a, vec), -C->size[1], C->size[1]);
    addscl3(pt, C->pos, a, x);
    sub3(diff, S->pos, pt);
    return norm3(diff) - (S->size[0] + C->size[0]);
}
/* [3P] mjc_CapsuleCapsule: closest points of the two axis segments */
static double d_capsule_capsule(const Geom *A, const Geom *B) {
    double a1[3], a2[3], r[3], w[3];
    col3(a1, A->mat, 2);
    col3(a2, B->mat, 2);
    sub3(r, A->pos, B->pos);
    double h1 = A->size[1], h2 = B->size[1];
    double b = dot3(a1, a2), c = dot3(a1, r), f = dot3(a2, r);
    double denom = fma(-b, b, 1.0);
    double sp = 0.0;
    if (denom > 1e-12) sp = clampd(fma(b, f, -c) / denom, -h1, h1);
    double tp = fma(b, sp, f);
    if (tp < -h2) { tp = -h2; sp = clampd(fma(b, tp, -c), -h1, h1); }
    else if (tp > h2) { tp = h2; sp = clampd(fma(b, tp, -c), -h1, h1); }
    addscl3(w, r, a1, sp);
    addscl3(w, w, a2, -tp);
    return norm3(w) - (A->size[0] + B->size[0]);
}
/* [3P] mjc_SphereBox */
static double d_sphere_box(const Geom *S, const Geom *B) {
    double v[3], l[3], e[3];
    sub3(v, S->pos, B->pos);
    matT_vec(l, B->mat, v);
    int inside = 1;
    for (int i = 0; i < 3; i++) {
        double cl = clampd(l[i], -B->size[i], B->size[i]);
        e[i] = l[i] - cl;
        if (e[i] != 0.0) inside = 0;
    }
    if (inside) {
        double m = dmin(dmin(B->size[0] - fabs(l[0]), B->size[1] - fabs(l[1])), B->size[2] - fabs(l[2]));
        return -m - S->size[0];
    }
    return norm3(e) - S->size[0];
}
/* analytic point/cylinder signed distance (MuJoCo >= 2.1.2 mjc_SphereCylinder;
 * 2.0 used MPR -- documented deviation) */
static double d_sphere_cylinder(const Geom *S, const Geom *C) {
    double a[3], v[3], w[3];
    col3(a, C->mat, 2);
    sub3(v, S->pos, C->pos);
    double z = dot3(v, a);
    addscl3(w, v, a, -z);
    double rho = norm3(w);
    double dr = rho - C->size[0];
    double dz = fabs(z) - C->size[1];
    double dp;
    if (dr <= 0.0 && dz <= 0.0) dp = dmax(dr, dz);
    else {
        double er = dmax(dr, 0.0), ez = dmax(dz, 0.0);
        dp = sqrt(fma(ez, ez, er * er));
    }
    return dp - S->size[0];
}

/* exact segment / box distance.  f(t) = |p(t) - clamp(p(t))|^2 is convex and
 * piecewise quadratic; f' is piecewise linear with knots where p(t) crosses a
 * slab plane, so the minimiser is found by bracketing the sign change of f'
 * among the knots and interpolating linearly (exact up to rounding). */
static inline double segbox_half_fprime(const double *p0, const double *d, const double *h, double t, double *e) {
    double g = 0.0;
    for (int i = 0; i < 3; i++) {
        double pi = fma(d[i], t, p0[i]);
        double cl = clampd(pi, -h[i], h[i]);
        e[i] = pi - cl;
    }
    g = dot3(e, d);
    return g;
}
static double d_capsule_box(const Geom *C, const Geom *B) {
    double a[3], v[3], cl[3], al[3], p0[3], d[3], e[3];
    const double *h = B->size;
    col3(a, C->mat, 2);
    sub3(v, C->pos, B->pos);
    matT_vec(cl, B->mat, v);   /* capsule centre in box frame */
    matT_vec(al, B->mat, a);   /* capsule axis in box frame */
    double hh = C->size[1];
    addscl3(p0, cl, al, -hh);
    d[0] = al[0] * (2.0 * hh); d[1] = al[1] * (2.0 * hh); d[2] = al[2] * (2.0 * hh);
    double tstar;
    double g0 = segbox_half_fprime(p0, d, h, 0.0, e);
    if (g0 >= 0.0) tstar = 0.0;
    else {
        double g1 = segbox_half_fprime(p0, d, h, 1.0, e);
        if (g1 <= 0.0) tstar = 1.0;
        else {
            double tL = 0.0, gL = g0, tR = 1.0, gR = g1;
            for (int i = 0; i < 3; i++) {
                if (d[i] == 0.0) continue;
                for (int sg = 0; sg < 2; sg++) {
                    double tk = ((sg ? h[i] : -h[i]) - p0[i]) / d[i];
                    if (!(tk > 0.0 && tk < 1.0)) continue;
                    double gk = segbox_half_fprime(p0, d, h, tk, e);
                    if (gk < 0.0) { if (tk > tL) { tL = tk; gL = gk; } }
                    else { if (tk < tR) { tR = tk; gR = gk; } }
                }
            }
            tstar = fma(tR - tL, (-gL) / (gR - gL), tL);
        }
    }
    segbox_half_fprime(p0, d, h, tstar, e);
    double dseg = norm3(e);
    if (dseg > 0.0) return dseg - C->size[0];
    /* axis segment pierces the box: SAT depth of segment vs box (+ radius) */
    double m[3], hd[3];
    addscl3(m, p0, d, 0.5);
    hd[0] = 0.5 * d[0]; hd[1] = 0.5 * d[1]; hd[2] = 0.5 * d[2];
    double best = -ORC_FAR;
    for (int i = 0; i < 3; i++) {
        double sep = fabs(m[i]) - (h[i] + fabs(hd[i]));
        best = dmax(best, sep);
    }
    for (int i = 0; i < 3; i++) {
        int j = (i + 1) % 3, k = (i + 2) % 3;
        /* L = e_i x al  -> components (j: -al[k], k: al[j]) */
        double l2 = fma(al[k], al[k], al[j] * al[j]);
        if (l2 < 1e-12) continue;
        double tl = fma(m[k], al[j], -(m[j] * al[k]));
        double ra = fma(h[k], fabs(al[j]), h[j] * fabs(al[k]));
        double sep = (fabs(tl) - ra) / sqrt(l2);
        best = dmax(best, sep);
    }
    return best - C->size[0];
}

/* 15-axis SAT; returns max normalised separation (<=0: -MTD, exact) */
static double d_box_box(const Geom *A, const Geom *B) {
    double R[3][3], AR[3][3], t[3], v[3];
    const double *ha = A->size, *hb = B->size;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            R[i][j] = fma(A->mat[6 + i], B->mat[6 + j], fma(A->mat[3 + i], B->mat[3 + j], A->mat[i] * B->mat[j]));
            AR[i][j] = fabs(R[i][j]);
        }
    sub3(v, B->pos, A->pos);
    matT_vec(t, A->mat, v);
    double best = -ORC_FAR;
    for (int i = 0; i < 3; i++) {
        double rb = fma(hb[2], AR[i][2], fma(hb[1], AR[i][1], hb[0] * AR[i][0]));
        double sep = fabs(t[i]) - (ha[i] + rb);
        best = dmax(best, sep);
    }
    for (int j = 0; j < 3; j++) {
        double tb = fma(t[2], R[2][j], fma(t[1], R[1][j], t[0] * R[0][j]));
        double ra = fma(ha[2], AR[2][j], fma(ha[1], AR[1][j], ha[0] * AR[0][j]));
        double sep = fabs(tb) - (ra + hb[j]);
        best = dmax(best, sep);
    }
    for (int i = 0; i < 3; i++) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        for (int j = 0; j < 3; j++) {
            int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            double l2 = fma(R[i2][j], R[i2][j], R[i1][j] * R[i1][j]);
            if (l2 < 1e-12) continue;
            double tl = fma(t[i2], R[i1][j], -(t[i1] * R[i2][j]));
            double ra = fma(ha[i2], AR[i1][j], ha[i1] * AR[i2][j]);
            double rb = fma(hb[j2], AR[i][j1], hb[j1] * AR[i][j2]);
            double sep = (fabs(tl) - (ra + rb)) / sqrt(l2);
            best = dmax(best, sep);
        }
    }
    return best;
}

/* ------------------------------------------------------------------ */
/* [3P] libccd MPR (ccdMPRPenetration) restated, with MuJoCo's support  */
/* functions (mjccd_support) for capsule / cylinder / box.              */
/* ------------------------------------------------------------------ */
#define CCD_EPS 2.2204460492503131e-16
#define MPR_TOL 1e-6
#define MPR_MAXIT 50
#define MPR_PORTAL_MAXIT 100 /* guard for the two loops libccd leaves unbounded */

static inline int is_zero(double x) { return fabs(x) < CCD_EPS; }
static inline int ccd_eq(double a, double b) {
    double ab = fabs(a - b);
    if (ab < CCD_EPS) return 1;
    a = fabs(a); b = fabs(b);
    if (b > a) return ab < CCD_EPS * b;
    return ab < CCD_EPS * a;
}
static inline int vec_eq(const double *a, const double *b) { return ccd_eq(a[0], b[0]) && ccd_eq(a[1], b[1]) && ccd_eq(a[2], b[2]); }
static inline void normalize3(double *v) {
    double inv = 1.0 / norm3(v);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

static void support_geom(const Geom *g, const double *dir, double *out) {
    double ld[3], lr[3], w[3];
    matT_vec(ld, g->mat, dir);
    switch (g->type) {
        case G_CAPSULE:
            lr[0] = ld[0] * g->size[0]; lr[1] = ld[1] * g->size[0];
            lr[2] = fma(ld[2], g->size[0], signd(ld[2]) * g->size[1]);
            break;
        case G_CYLINDER: {
            double tmp = sqrt(fma(ld[1], ld[1], ld[0] * ld[0]));
            if (tmp > MINVAL) {
                double sc = g->size[0] / tmp;
                lr[0] = ld[0] * sc; lr[1] = ld[1] * sc;
            } else { lr[0] = 0.0; lr[1] = 0.0; }
            lr[2] = signd(ld[2]) * g->size[1];
        } break;
        case G_BOX:
            lr[0] = signd(ld[0]) * g->size[0]; lr[1] = signd(ld[1]) * g->size[1]; lr[2] = signd(ld[2]) * g->size[2];
            break;
        case G_MESH: { /* [3P] mjccd_support, mesh: exhaustive search over the hull vertices, first maximum wins */
            int best = 0;
            double bd = dot3(ld, g->verts);
            for (int i = 1; i < g->nvert; i++) {
                double d = dot3(ld, g->verts + 3 * i);
                if (d > bd) { bd = d; best = i; }
            }
            lr[0] = g->verts[3 * best]; lr[1] = g->verts[3 * best + 1]; lr[2] = g->verts[3 * best + 2];
        } break;
        default: /* sphere */
            lr[0] = ld[0] * g->size[0]; lr[1] = ld[1] * g->size[0]; lr[2] = ld[2] * g->size[0];
            break;
    }
    mat_vec(w, g->mat, lr);
    add3(out, w, g->pos);
}
/* Minkowski-difference support: s1(dir) - s2(-dir)  (libccd __ccdSupport) */
static void support_md(const Geom *g1, const Geom *g2, const double *dir, double *v) {
    double s1[3], s2[3], nd[3] = { -dir[0], -dir[1], -dir[2] };
    support_geom(g1, dir, s1);
    support_geom(g2, nd, s2);
    sub3(v, s1, s2);
}
static inline void portal_dir(const double *v1, const double *v2, const double *v3, double *dir) {
    double a[3], b[3];
    sub3(a, v2, v1);
    sub3(b, v3, v1);
    cross3(dir, a, b);
    normalize3(dir);
}
static inline int portal_reach_tol(const double *v1, const double *v2, const double *v3, const double *v4, const double *dir) {
    double dv1 = dot3(v1, dir), dv2 = dot3(v2, dir), dv3 = dot3(v3, dir), dv4 = dot3(v4, dir);
    double d1 = dv4 - dv1, d2 = dv4 - dv2, d3 = dv4 - dv3;
    double m = dmin(dmin(d1, d2), d3);
    return ccd_eq(m, MPR_TOL) || m < MPR_TOL;
}
static inline void expand_portal(const double *v0, double *v1, double *v2, double *v3, const double *v4) {
    double v4v0[3];
    cross3(v4v0, v4, v0);
    double dot = dot3(v1, v4v0);
    if (dot > 0.0) {
        dot = dot3(v2, v4v0);
        if (dot > 0.0) memcpy(v1, v4, 24); else memcpy(v3, v4, 24);
    } else {
        dot = dot3(v3, v4v0);
        if (dot > 0.0) memcpy(v2, v4, 24); else memcpy(v1, v4, 24);
    }
}
static double point_seg_dist2(const double *P, const double *x0, const double *b) {
    /* libccd ccdVec3PointSegmentDist2 */
    double d[3], a[3], w[3];
    sub3(d, b, x0);
    sub3(a, x0, P);
    double t = -1.0 * dot3(a, d);
    t = t / dot3(d, d);
    if (t < 0.0 || is_zero(t)) { sub3(w, x0, P); return dot3(w, w); }
    if (t > 1.0 || ccd_eq(t, 1.0)) { sub3(w, b, P); return dot3(w, w); }
    addscl3(w, a, d, t);
    return dot3(w, w);
}
static double point_tri_dist2(const double *P, const double *x0, const double *B, const double *C) {
    /* libccd ccdVec3PointTriDist2 */
    double d1[3], d2[3], a[3];
    sub3(d1, B, x0);
    sub3(d2, C, x0);
    sub3(a, x0, P);
    double u = dot3(a, a), v = dot3(d1, d1), w = dot3(d2, d2);
    double p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
    double s = fma(q, r, -(w * p)) / fma(w, v, -(r * r));
    double t = (fma(-s, r, -q)) / w;
    if ((is_zero(s) || s > 0.0) && (ccd_eq(s, 1.0) || s < 1.0) && (is_zero(t) || t > 0.0) &&
        (ccd_eq(t, 1.0) || t < 1.0) && (ccd_eq(t + s, 1.0) || t + s < 1.0)) {
        /* witness form (libccd is called with a witness by findPenetr): |x0 + s d1 + t d2 - P|^2 */
        double wv[3];
        (void)u;
        addscl3(wv, a, d1, s);
        addscl3(wv, wv, d2, t);
        return dot3(wv, wv);
    }
    double dist = point_seg_dist2(P, x0, B);
    double dd = point_seg_dist2(P, x0, C);
    if (dd < dist) dist = dd;
    dd = point_seg_dist2(P, B, C);
    if (dd < dist) dist = dd;
    return dist;
}

/* returns 0 and *depth on intersection, -1 otherwise */
static int mpr_penetration(const Geom *g1, const Geom *g2, double *depth) {
    const double origin[3] = { 0.0, 0.0, 0.0 };
    double v0[3], v1[3], v2[3], v3[3], v4[3], dir[3], va[3], vb[3];
    double dot;
    /* --- discoverPortal --- */
    sub3(v0, g1->pos, g2->pos);
    if (vec_eq(v0, origin)) v0[0] += CCD_EPS * 10.0;
    dir[0] = -v0[0]; dir[1] = -v0[1]; dir[2] = -v0[2];
    normalize3(dir);
    support_md(g1, g2, dir, v1);
    dot = dot3(v1, dir);
    if (is_zero(dot) || dot < 0.0) return -1;
    cross3(dir, v0, v1);
    if (is_zero(dot3(dir, dir))) {
        if (vec_eq(v1, origin)) { *depth = 0.0; return 0; }   /* touching contact */
        *depth = norm3(v1);                                   /* origin on v0-v1 segment */
        return 0;
    }
    normalize3(dir);
    support_md(g1, g2, dir, v2);
    dot = dot3(v2, dir);
    if (is_zero(dot) || dot < 0.0) return -1;
    sub3(va, v1, v0);
    sub3(vb, v2, v0);
    cross3(dir, va, vb);
    normalize3(dir);
    dot = dot3(dir, v0);
    if (dot > 0.0) {
        double tmp[3];
        memcpy(tmp, v1, 24); memcpy(v1, v2, 24); memcpy(v2, tmp, 24);
        dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2];
    }
    int it = 0;
    for (;;) {
        if (++it > MPR_PORTAL_MAXIT) return -1;
        support_md(g1, g2, dir, v3);
        dot = dot3(v3, dir);
        if (is_zero(dot) || dot < 0.0) return -1;
        int cont = 0;
        cross3(va, v1, v3);
        dot = dot3(va, v0);
        if (dot < 0.0 && !is_zero(dot)) { memcpy(v2, v3, 24); cont = 1; }
        if (!cont) {
            cross3(va, v3, v2);
            dot = dot3(va, v0);
            if (dot < 0.0 && !is_zero(dot)) { memcpy(v1, v3, 24); cont = 1; }
        }
        if (!cont) break;
        sub3(va, v1, v0);
        sub3(vb, v2, v0);
        cross3(dir, va, vb);
        normalize3(dir);
    }
    /* --- refinePortal --- */
    it = 0;
    for (;;) {
        if (++it > MPR_PORTAL_MAXIT) return -1;
        portal_dir(v1, v2, v3, dir);
        dot = dot3(dir, v1);
        if (is_zero(dot) || dot > 0.0) break; /* portal encapsules origin */
        support_md(g1, g2, dir, v4);
        dot = dot3(v4, dir);
        if (!(is_zero(dot) || dot > 0.0) || portal_reach_tol(v1, v2, v3, v4, dir)) return -1;
        expand_portal(v0, v1, v2, v3, v4);
    }
    /* --- findPenetr --- */
    int iterations = 0;
    for (;;) {
        portal_dir(v1, v2, v3, dir);
        support_md(g1, g2, dir, v4);
        if (portal_reach_tol(v1, v2, v3, v4, dir) || iterations > MPR_MAXIT) {
            *depth = sqrt(point_tri_dist2(origin, v1, v2, v3));
            return 0;
        }
        expand_portal(v0, v1, v2, v3, v4);
        iterations++;
    }
}
static double d_convex(const Geom *A, const Geom *B) {
    double depth;
    if (mpr_penetration(A, B, &depth) == 0) return -depth;
    return ORC_FAR;
}

/* {capsule,cylinder}-cylinder and cylinder-box: closed-form pre-test on the enclosing capsules (a cylinder lies inside
 * the capsule of the same axis, radius and half length) before the portal refinement; enclosures apart by more
 * than 1e-9 => disjoint => what MPR reports for disjoint shapes.  Same rule in the kernels. */
/* Second pre-test for what the enclosing capsules let through: separating axes that are exact for these shapes -- the axis
 * of shape 1 (capsule or cylinder) and the axis / the three face normals of shape 2 (cylinder / box).  Extents along a unit
 * direction n: cylinder h |a.n| + r sqrt(1 - (a.n)^2), capsule h |a.n| + r, box sum |u_i.n| s_i.  A gap of more than 1e-9
 * => disjoint => ORC_FAR, which is what the portal refinement reports for disjoint shapes.  This is an accelerator, not part
 * of [3P]: orc_set_convex_axes_pretest(0) turns it off, and tests/test_oracle_primitives.py checks that no distance changes. */
static int g_convex_axes_pretest = 1;
void orc_set_convex_axes_pretest(int on) { g_convex_axes_pretest = on; }
static int convex_axes_separate(const Geom *A, const Geom *B) {
    double d[3], a[3];
    sub3(d, B->pos, A->pos);
    col3(a, A->mat, 2);
    const double ra = A->size[0], ha = A->size[1];
    const int cap = A->type == G_CAPSULE;
    const double ea = cap ? ha + ra : ha;
    const double da = fabs(dot3(d, a));
    if (B->type == G_BOX) {
        const double *s = B->size;
        double u0[3], u1[3], u2[3];
        col3(u0, B->mat, 0); col3(u1, B->mat, 1); col3(u2, B->mat, 2);
        const double c0 = dot3(a, u0), c1 = dot3(a, u1), c2 = dot3(a, u2);
        if (da - ea - fma(fabs(c2), s[2], fma(fabs(c1), s[1], fabs(c0) * s[0])) > 1e-9) return 1;
        const double n0 = fma(-c0, c0, 1.0), n1 = fma(-c1, c1, 1.0), n2 = fma(-c2, c2, 1.0);
        const double w0 = cap ? ra : ra * sqrt(n0 > 0.0 ? n0 : 0.0);
        const double w1 = cap ? ra : ra * sqrt(n1 > 0.0 ? n1 : 0.0);
        const double w2 = cap ? ra : ra * sqrt(n2 > 0.0 ? n2 : 0.0);
        if (fabs(dot3(d, u0)) - s[0] - fma(ha, fabs(c0), w0) > 1e-9) return 1;
        if (fabs(dot3(d, u1)) - s[1] - fma(ha, fabs(c1), w1) > 1e-9) return 1;
        if (fabs(dot3(d, u2)) - s[2] - fma(ha, fabs(c2), w2) > 1e-9) return 1;
        return 0;
    }
    double b[3];
    col3(b, B->mat, 2);
    const double rb = B->size[0], hb = B->size[1];
    const double c = dot3(a, b), n = fma(-c, c, 1.0), sn = sqrt(n > 0.0 ? n : 0.0), ac = fabs(c);
    if (da - ea - fma(hb, ac, rb * sn) > 1e-9) return 1;
    if (fabs(dot3(d, b)) - hb - fma(ha, ac, cap ? ra : ra * sn) > 1e-9) return 1;
    return 0;
}
static double d_convex_cyl(const Geom *A, const Geom *B) {
    const double pre = (B->type == G_BOX) ? d_capsule_box(A, B) : d_capsule_capsule(A, B);
    if (pre > 1e-9) return ORC_FAR;
    if (g_convex_axes_pretest && convex_axes_separate(A, B)) return ORC_FAR;
    return d_convex(A, B);
}

static double geom_dist(const Geom *A, const Geom *B) {
    switch (A->type) {
        case G_PLANE:
            switch (B->type) {
                case G_SPHERE: return d_plane_sphere(A, B);
                case G_CAPSULE: return d_plane_capsule(A, B);
                case G_CYLINDER: return d_plane_cylinder(A, B);
                case G_BOX: return d_plane_box(A, B);
                case G_MESH: return d_plane_mesh(A, B);
                default: return ORC_FAR;
            }
        case G_SPHERE:
            switch (B->type) {
                case G_SPHERE: return d_sphere_sphere(A, B);
                case G_CAPSULE: return d_sphere_capsule(A, B);
                case G_CYLINDER: return d_sphere_cylinder(A, B);
                case G_BOX: return d_sphere_box(A, B);
                case G_MESH: return d_convex(A, B);   /* [3P] mjc_Convex for every primitive-mesh pair */
                default: return ORC_FAR;
            }
        case G_CAPSULE:
            switch (B->type) {
                case G_CAPSULE: return d_capsule_capsule(A, B);
                case G_CYLINDER: return d_convex_cyl(A, B);
                case G_BOX: return d_capsule_box(A, B);
                case G_MESH: return d_convex(A, B);
                default: return ORC_FAR;
            }
        case G_CYLINDER:
            switch (B->type) {
                case G_CYLINDER: return d_convex_cyl(A, B);
                case G_BOX: return d_convex_cyl(A, B);
                case G_MESH: return d_convex(A, B);
                default: return ORC_FAR;
            }
        case G_BOX:
            if (B->type == G_BOX) return d_box_box(A, B);
            if (B->type == G_MESH) return d_convex(A, B);
            return ORC_FAR;
        default: return ORC_FAR;
    }
}

double orc_geom_dist_mesh(int t1, const double *size1, const double *pos1, const double *mat1,
                          const double *verts, int nvert, const double *pos2, const double *mat2) {
    static const double zero3[3] = { 0.0, 0.0, 0.0 };
    Geom A = { t1, size1, pos1, mat1, NULL, 0 }, B = { G_MESH, zero3, pos2, mat2, verts, nvert };
    return geom_dist(&A, &B);
}

static inline Geom scene_geom(const OrcScene *s, int g, const double *gpos, const double *gmat) {
    Geom G = { s->geom_type[g], s->geom_size + 3 * g, gpos + 3 * g, gmat + 9 * g, NULL, 0 };
    if (s->geom_dataid && s->geom_dataid[g] >= 0) {
        G.verts = s->mesh_vert + 3 * s->mesh_vertadr[s->geom_dataid[g]];
        G.nvert = s->mesh_vertnum[s->geom_dataid[g]];
    }
    return G;
}

double orc_geom_dist(int t1, const double *size1, const double *pos1, const double *mat1,
                     int t2, const double *size2, const double *pos2, const double *mat2) {
    Geom A = { t1, size1, pos1, mat1, NULL, 0 }, B = { t2, size2, pos2, mat2, NULL, 0 };
    return geom_dist(&A, &B);
}

/* [3P] broad phase: bounding spheres, zero margin (thr < 0) */
static inline int bp_cull(const OrcScene *s, int g1, int g2, const double *gpos, const double *gmat) {
    if (s->geom_type[g1] == G_PLANE) {
        double n[3], diff[3];
        col3(n, gmat + 9 * g1, 2);
        sub3(diff, gpos + 3 * g2, gpos + 3 * g1);
        return dot3(diff, n) > s->geom_rbound[g2];
    }
    double diff[3];
    sub3(diff, gpos + 3 * g2, gpos + 3 * g1);
    double rs = s->geom_rbound[g1] + s->geom_rbound[g2];
    if (dot3(diff, diff) > rs * rs) return 1;
    /* second stage, exactly one geom static: its world AABB vs the moving geom's bounding sphere */
    if (s->geom_static[g1] == s->geom_static[g2]) return 0;
    {
        const int gs = s->geom_static[g1] ? g1 : g2, gm = s->geom_static[g1] ? g2 : g1;
        const double *cS = gpos + 3 * gs, *cM = gpos + 3 * gm, *H = s->geom_aabb + 3 * gs, rM = s->geom_rbound[gm];
        return fabs(cM[0] - cS[0]) > H[0] + rM || fabs(cM[1] - cS[1]) > H[1] + rM || fabs(cM[2] - cS[2]) > H[2] + rM;
    }
}

static void pair_dists(const OrcScene *s, const double *gpos, const double *gmat, double *dist) {
    for (int p = 0; p < s->npair; p++) {
        int g1 = s->pair_geom[2 * p], g2 = s->pair_geom[2 * p + 1];
        if (bp_cull(s, g1, g2, gpos, gmat)) { dist[p] = ORC_FAR; continue; }
        Geom A = scene_geom(s, g1, gpos, gmat), B = scene_geom(s, g2, gpos, gmat);
        dist[p] = geom_dist(&A, &B);
    }
}

void orc_pair_dist(const OrcScene *s, const double *qpos, double *dist) {
    double *gpos = (double *)malloc(sizeof(double) * 12 * s->ngeom);
    double *gmat = gpos + 3 * s->ngeom;
    orc_fk(s, qpos, gpos, gmat);
    pair_dists(s, gpos, gmat, dist);
    free(gpos);
}

/* mujoco_ompl_interface.cpp:909-978: invalid iff some contact not in
 * ignored_contacts has dist <= contact_threshold */
static int is_valid_ws(const OrcScene *s, const double *qpos, double *min_dist, double *ws) {
    double *gpos = ws, *gmat = ws + 3 * s->ngeom;
    double *xpos = gmat + 9 * s->ngeom, *xquat = xpos + 3 * s->nbody, *xmat = xquat + 4 * s->nbody;
    fk_bodies(s, qpos, xpos, xquat, xmat, 1);
    fk_geoms(s, xpos, xquat, xmat, gpos, gmat);
    double md = 0.0;   /* deepest penetration: min(0, min over pairs) -- independent of how much the broad phase culls */
    int valid = 1;
    for (int p = 0; p < s->npair; p++) {
        if (s->pair_ignored[p]) continue;
        int g1 = s->pair_geom[2 * p], g2 = s->pair_geom[2 * p + 1];
        if (bp_cull(s, g1, g2, gpos, gmat)) continue;
        Geom A = scene_geom(s, g1, gpos, gmat), B = scene_geom(s, g2, gpos, gmat);
        double d = geom_dist(&A, &B);
        if (d < md) md = d;
        if (d <= s->thr) valid = 0; /* the reference keeps scanning (its break is commented out) */
    }
    if (min_dist) *min_dist = md;
    return valid;
}
static size_t ws_doubles(const OrcScene *s) { return (size_t)12 * s->ngeom + (size_t)16 * s->nbody; }

int orc_is_valid(const OrcScene *s, const double *qpos, double *min_dist) {
    double *ws = (double *)malloc(sizeof(double) * ws_doubles(s));
    int v = is_valid_ws(s, qpos, min_dist, ws);
    free(ws);
    return v;
}

void orc_is_valid_batch(const OrcScene *s, const double *q_active, const double *qpos_env, int64_t N,
                        int64_t samples_per_env, uint8_t *valid, double *min_dist, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel if (nthreads != 1)
#endif
    {
        double *ws = (double *)malloc(sizeof(double) * (ws_doubles(s) + s->nq));
        double *qpos = ws + ws_doubles(s);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < N; i++) {
            memcpy(qpos, qpos_env + (i / samples_per_env) * s->nq, sizeof(double) * s->nq);
            for (int a = 0; a < s->na; a++) qpos[s->active_idx[a]] = q_active[i * s->na + a];
            double md;
            valid[i] = (uint8_t)is_valid_ws(s, qpos, &md, ws);
            if (min_dist) min_dist[i] = md;
        }
        free(ws);
    }
}

/* ------------------------------------------------------------------ */
/* [3P] OMPL DiscreteMotionValidator::checkMotion + StateSpace helpers  */
/* ------------------------------------------------------------------ */
static int valid_active(const OrcScene *s, const double *qpos_env, const double *qa, double *qpos, double *ws) {
    memcpy(qpos, qpos_env, sizeof(double) * s->nq);
    for (int a = 0; a < s->na; a++) qpos[s->active_idx[a]] = qa[a];
    return is_valid_ws(s, qpos, NULL, ws);
}
/* distance in one 1-D subspace: |d| (R^1) or the shorter arc (SO2) */
static inline double dist_dim(const OrcScene *s, int a, double x, double y) {
    double d = fabs(x - y);
    if (s->act_so2[a] && d > ORC_PI) d = 2.0 * ORC_PI - d;
    return d;
}
/* CompoundStateSpace::validSegmentCount = max over 1-D subspaces of
 * ceil(|d_i| / (resolution * extent_i)) */
static int valid_segment_count(const OrcScene *s, const double *qa, const double *qb, double resolution) {
    int nd = 0;
    for (int a = 0; a < s->na; a++) {
        double seg = resolution * s->act_ext[a];
        int c = (int)ceil(dist_dim(s, a, qa[a], qb[a]) / seg);
        if (c > nd) nd = c;
    }
    return nd;
}
/* RealVectorStateSpace / SO2StateSpace ::interpolate, per 1-D subspace */
static inline void interpolate(const OrcScene *s, const double *from, const double *to, double t, double *out) {
    for (int a = 0; a < s->na; a++) {
        double diff = to[a] - from[a];
        if (!s->act_so2[a] || fabs(diff) <= ORC_PI) out[a] = fma(diff, t, from[a]);
        else {
            if (diff > 0.0) diff = 2.0 * ORC_PI - diff; else diff = -2.0 * ORC_PI - diff;
            double v = fma(-diff, t, from[a]);
            if (v > ORC_PI) v -= 2.0 * ORC_PI; else if (v < -ORC_PI) v += 2.0 * ORC_PI;
            out[a] = v;
        }
    }
}
static int check_motion_ws(const OrcScene *s, const double *qpos_env, const double *qa, const double *qb,
                           double resolution, int64_t *n_checks, double *qpos, double *ws) {
    int64_t nc = 0;
    int result = 1;
    nc++;
    if (!valid_active(s, qpos_env, qb, qpos, ws)) { if (n_checks) *n_checks += nc; return 0; }
    int nd = valid_segment_count(s, qa, qb, resolution);
    if (nd >= 2) {
        /* breadth-first bisection order, as OMPL's std::queue of (first,last) */
        int *queue = (int *)malloc(sizeof(int) * 2 * (size_t)(nd + 2));
        double *test = (double *)malloc(sizeof(double) * s->na);
        int head = 0, tail = 0;
        queue[2 * tail] = 1; queue[2 * tail + 1] = nd - 1; tail++;
        while (head < tail) {
            int first = queue[2 * head], second = queue[2 * head + 1];
            int mid = (first + second) / 2;
            interpolate(s, qa, qb, (double)mid / (double)nd, test);
            nc++;
            if (!valid_active(s, qpos_env, test, qpos, ws)) { result = 0; break; }
            head++;
            if (first < mid) { queue[2 * tail] = first; queue[2 * tail + 1] = mid - 1; tail++; }
            if (second > mid) { queue[2 * tail] = mid + 1; queue[2 * tail + 1] = second; tail++; }
        }
        free(queue);
        free(test);
    }
    if (n_checks) *n_checks += nc;
    return result;
}

int orc_check_motion(const OrcScene *s, const double *qpos_env, const double *qa, const double *qb,
                     double resolution, int64_t *n_checks) {
    double *ws = (double *)malloc(sizeof(double) * (ws_doubles(s) + s->nq));
    int r = check_motion_ws(s, qpos_env, qa, qb, resolution, n_checks, ws + ws_doubles(s), ws);
    free(ws);
    return r;
}

void orc_check_motion_batch(const OrcScene *s, const double *qa, const double *qb, const double *qpos_env, int64_t N,
                            int64_t samples_per_env, double resolution, uint8_t *valid, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel if (nthreads != 1)
#endif
    {
        double *ws = (double *)malloc(sizeof(double) * (ws_doubles(s) + s->nq));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < N; i++)
            valid[i] = (uint8_t)check_motion_ws(s, qpos_env + (i / samples_per_env) * s->nq, qa + i * s->na,
                                                qb + i * s->na, resolution, NULL, ws + ws_doubles(s), ws);
        free(ws);
    }
}

/* ------------------------------------------------------------------ */
/* counter-based RNG (splitmix64 finaliser over (seed, stream, counter)) */
/* ------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
uint64_t orc_rng_u64(uint64_t seed, uint64_t stream, uint64_t counter) {
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1));
    return mix64(k + 0x9E3779B97F4A7C15ULL * (counter + 1));
}
double orc_rng_uniform(uint64_t seed, uint64_t stream, uint64_t counter) {
    return (double)(orc_rng_u64(seed, stream, counter) >> 11) * 0x1.0p-53;
}

/* ------------------------------------------------------------------ */
/* [3P] OMPL geometric::RRTConnect::solve / growTree restated           */
/* ------------------------------------------------------------------ */
typedef struct { double *q; int *parent; int n, cap; } Tree;

static double dist_l1(const OrcScene *s, const double *a, const double *b) {
    double d = 0.0;
    for (int i = 0; i < s->na; i++) d += dist_dim(s, i, a[i], b[i]);
    return d;
}
static int nearest(const OrcScene *s, const Tree *t, const double *q) {
    int best = 0;
    double bd = dist_l1(s, t->q, q);
    for (int i = 1; i < t->n; i++) {
        double d = dist_l1(s, t->q + (size_t)i * s->na, q);
        if (d < bd) { bd = d; best = i; } /* ties -> lowest index */
    }
    return best;
}
enum { TRAPPED = 0, ADVANCED = 1, REACHED = 2 };

typedef struct {
    const OrcScene *s; const double *qpos_env; double range, resolution;
    double *qpos, *ws, *xstate; int64_t *n_checks; int xmotion;
} Grow;

static int grow_tree(Grow *g, Tree *tree, int is_start, const double *rstate) {
    const OrcScene *s = g->s;
    int nm = nearest(s, tree, rstate);
    const double *nstate = tree->q + (size_t)nm * s->na;
    int reach = 1;
    const double *dstate = rstate;
    double d = dist_l1(s, nstate, rstate);
    if (d > g->range) {
        interpolate(s, nstate, rstate, g->range / d, g->xstate);
        int same = 1;
        for (int a = 0; a < s->na; a++) if (fabs(g->xstate[a] - nstate[a]) > 2.0 * CCD_EPS) same = 0; /* StateSpace::equalStates */
        if (same) return TRAPPED;
        dstate = g->xstate;
        reach = 0;
    }
    /* node budget (a deviation: OMPL's trees are unbounded): a full tree cannot take the new state, so its motion is not
     * validated either -- before this test sat behind the validation, a query that had filled a tree kept paying for it */
    if (tree->n >= tree->cap) return TRAPPED;
    int ok;
    if (is_start) ok = check_motion_ws(s, g->qpos_env, nstate, dstate, g->resolution, g->n_checks, g->qpos, g->ws);
    else {
        (*g->n_checks)++;
        ok = valid_active(s, g->qpos_env, dstate, g->qpos, g->ws) &&
             check_motion_ws(s, g->qpos_env, dstate, nstate, g->resolution, g->n_checks, g->qpos, g->ws);
    }
    if (!ok) return TRAPPED;
    memcpy(tree->q + (size_t)tree->n * s->na, dstate, sizeof(double) * s->na);
    tree->parent[tree->n] = nm;
    g->xmotion = tree->n;
    tree->n++;
    return reach ? REACHED : ADVANCED;
}

int orc_plan(const OrcScene *s, const double *start, const double *goal, double range, double resolution,
             int max_iters, int max_nodes, uint64_t seed, uint64_t env_id, double *path, int max_path,
             int *path_len, int64_t *n_checks_out, int *n_iters_out) {
    int na = s->na;
    int64_t n_checks = 0;
    double *ws = (double *)malloc(sizeof(double) * (ws_doubles(s) + s->nq + 4 * (size_t)na));
    double *qpos = ws + ws_doubles(s);
    double *qs = qpos + s->nq, *qg = qs + na, *rstate = qg + na, *xstate = rstate + na;
    for (int a = 0; a < na; a++) { qs[a] = start[s->active_idx[a]]; qg[a] = goal[s->active_idx[a]]; }
    int status = 0;
    *path_len = 0;
    /* KinematicPlanner.cpp:181-184: goal invalid -> one row of -5 (passive from start) */
    n_checks++;
    if (!valid_active(s, start, qg, qpos, ws)) {
        status = -5;
        goto done;
    }
    {
        Tree ts, tg;
        ts.cap = tg.cap = max_nodes;
        ts.q = (double *)malloc(sizeof(double) * (size_t)max_nodes * na);
        tg.q = (double *)malloc(sizeof(double) * (size_t)max_nodes * na);
        ts.parent = (int *)malloc(sizeof(int) * max_nodes);
        tg.parent = (int *)malloc(sizeof(int) * max_nodes);
        memcpy(ts.q, qs, sizeof(double) * na); ts.parent[0] = -1; ts.n = 1;
        memcpy(tg.q, qg, sizeof(double) * na); tg.parent[0] = -1; tg.n = 1;
        Grow g = { s, start, range, resolution, qpos, ws, xstate, &n_checks, -1 };
        int start_tree = 1, solved = 0, it = 0;
        int start_motion = -1, goal_motion = -1;
        /* OMPL rejects an invalid start before planning ("Invalid start" -> -4 row) */
        n_checks++;
        if (!valid_active(s, start, qs, qpos, ws)) { max_iters = 0; }
        for (it = 0; it < max_iters && !solved; it++) {
            Tree *tree = start_tree ? &ts : &tg;
            int tgi_start = start_tree;
            start_tree = !start_tree;
            Tree *other = start_tree ? &ts : &tg;
            for (int a = 0; a < na; a++)
                rstate[a] = fma(s->act_hi[a] - s->act_lo[a], orc_rng_uniform(seed, env_id, (uint64_t)it * na + a), s->act_lo[a]);
            int gs = grow_tree(&g, tree, tgi_start, rstate);
            if (gs != TRAPPED) {
                int added = g.xmotion;
                if (gs != REACHED) memcpy(rstate, xstate, sizeof(double) * na);
                int gsc = ADVANCED;
                tgi_start = start_tree;
                while (gsc == ADVANCED) gsc = grow_tree(&g, other, tgi_start, rstate);
                if (gsc == REACHED) {
                    start_motion = start_tree ? g.xmotion : added;
                    goal_motion = start_tree ? added : g.xmotion;
                    /* step back once to avoid the duplicated connection state */
                    if (ts.parent[start_motion] != -1) start_motion = ts.parent[start_motion];
                    else goal_motion = tg.parent[goal_motion];
                    solved = 1;
                }
            }
        }
        if (n_iters_out) *n_iters_out = it;
        if (solved) {
            int n1 = 0, n2 = 0;
            for (int m = start_motion; m != -1; m = ts.parent[m]) n1++;
            for (int m = goal_motion; m != -1; m = tg.parent[m]) n2++;
            if (n1 + n2 > max_path) status = -4;
            else {
                int k = n1 - 1;
                for (int m = start_motion; m != -1; m = ts.parent[m], k--) {
                    double *row = path + (size_t)k * s->nq;
                    memcpy(row, start, sizeof(double) * s->nq); /* passive columns from start (KinematicPlanner.cpp:236-240) */
                    for (int a = 0; a < na; a++) row[s->active_idx[a]] = ts.q[(size_t)m * na + a];
                }
                k = n1;
                for (int m = goal_motion; m != -1; m = tg.parent[m], k++) {
                    double *row = path + (size_t)k * s->nq;
                    memcpy(row, start, sizeof(double) * s->nq);
                    for (int a = 0; a < na; a++) row[s->active_idx[a]] = tg.q[(size_t)m * na + a];
                }
                *path_len = n1 + n2;
            }
        } else status = -4;
        free(ts.q); free(tg.q); free(ts.parent); free(tg.parent);
    }
done:
    if (n_checks_out) *n_checks_out = n_checks;
    if (status != 0 && n_iters_out && status == -5) *n_iters_out = 0;
    free(ws);
    return status;
}

void orc_plan_batch(const OrcScene *s, int64_t E, const double *start, const double *goal, double range, double resolution, int max_iters,
                    int max_nodes, uint64_t seed, uint64_t env_id_base, double *path, int max_path, int32_t *status, int32_t *path_len,
                    int64_t *n_checks, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t e = 0; e < E; e++) {
        double *row = path ? path + (size_t)e * max_path * s->nq : (double *)malloc(sizeof(double) * (size_t)max_path * s->nq);
        int pl = 0, it = 0;
        int64_t nc = 0;
        status[e] = orc_plan(s, start + e * s->nq, goal + e * s->nq, range, resolution, max_iters, max_nodes, seed, env_id_base + (uint64_t)e, row,
                             max_path, &pl, &nc, &it);
        path_len[e] = pl;
        n_checks[e] = nc;
        if (!path) free(row);
    }
}

/* ------------------------------------------------------------------ */
/*  kinematic env.step of the Sawyer obstacle envs */
/* ------------------------------------------------------------------ */
/* Restates the arithmetic AROUND the physics of the reference envs (kind 0 = SawyerPushObstacle, 1 = SawyerLiftObstacle,
 * 2 = SawyerAssemblyObstacle):
 *   _step          env/sawyer/sawyer_{push,lift,assembly}_obstacle.py `_step` (action scaling, desired_state, prev_state;
 *                  Lift: gripper target = gripper qpos + action[-1], sawyer.py:340-342)
 *   _after_step    env/base.py:269-314 (joint-limit clamp, episode length, terminal)
 *   compute_reward sawyer_push_obstacle.py:71-104, sawyer_lift_obstacle.py:93-150, sawyer_assembly_obstacle.py:33-52
 *   _get_obs       env/sawyer/sawyer.py:317-338 + the env's own `_get_obs` (dict order)
 * with `_do_simulation` (MuJoCo position servos, 75 sub-steps) replaced by its kinematic limit: every actuated joint
 * reaches its target clamped to the actuator's ctrlrange, velocities are 0, nothing else moves.  NOT dynamics parity.
 * One deliberate re-ordering: the joint-limit clamp is applied BEFORE obs/reward (in the reference MuJoCo's limit
 * constraint acts inside the physics, and the explicit clamp of _after_step runs after the obs was taken).
 * Lift's grasp test (`has_grasp`: a contact between the can and a left-finger geom AND one with a right-finger geom) is
 * evaluated on the posed geoms: bounding-sphere cull, then the narrow phase; "contact" = the shapes intersect. */
static inline void frame_pos(double *out, const double *xpos, const double *xmat, int b, const double *off) {
    double v[3];
    mat_vec(v, xmat + 9 * b, off);
    add3(out, xpos + 3 * b, v);
}

static int env_touch(const OrcScene *s, int g_obj, int g_fin, const double *xpos, const double *xquat, const double *xmat) {
    /* geoms ordered by type for the narrow phase (MuJoCo orders a contact's geoms by type) */
    int g1 = g_fin, g2 = g_obj;
    if (s->geom_type[g1] > s->geom_type[g2]) { g1 = g_obj; g2 = g_fin; }
    double gp[2][3], gm[2][9], q[4], v[3];
    const int gs[2] = { g1, g2 };
    for (int k = 0; k < 2; k++) {
        const int g = gs[k], b = s->geom_body[g];
        mat_vec(v, xmat + 9 * b, s->geom_pos + 3 * g);
        add3(gp[k], xpos + 3 * b, v);
        quat_mul(q, xquat + 4 * b, s->geom_quat + 4 * g);
        quat2mat(gm[k], q);
    }
    double diff[3];
    sub3(diff, gp[1], gp[0]);
    const double rs = s->geom_rbound[g1] + s->geom_rbound[g2];
    if (dot3(diff, diff) > rs * rs) return 0;
    Geom A = { s->geom_type[g1], s->geom_size + 3 * g1, gp[0], gm[0], NULL, 0 }, B = { s->geom_type[g2], s->geom_size + 3 * g2, gp[1], gm[1], NULL, 0 };
    if (s->geom_dataid && s->geom_dataid[g1] >= 0) { A.verts = s->mesh_vert + 3 * s->mesh_vertadr[s->geom_dataid[g1]]; A.nvert = s->mesh_vertnum[s->geom_dataid[g1]]; }
    if (s->geom_dataid && s->geom_dataid[g2] >= 0) { B.verts = s->mesh_vert + 3 * s->mesh_vertadr[s->geom_dataid[g2]]; B.nvert = s->mesh_vertnum[s->geom_dataid[g2]]; }
    return geom_dist(&A, &B) < 0.0;
}

int orc_env_obs_dim(const OrcEnvDesc *d) {
    if (d->kind == 3) return 3 * d->n_arm + 8;      /* cos, sin, box 2, joint vel, box vel 2, fingertip 2, goal 2 */
    const int base = 2 * d->n_arm + 2 * d->n_grip + 7;
    return base + (d->kind == 0 ? 15 : (d->kind == 1 ? 10 : 13));
}
int orc_env_action_dim(const OrcEnvDesc *d) { return d->n_arm + (d->kind == 1 ? 1 : 0); }

#include "mopa_oracle_dyn.inc"
#include "mopa_oracle_contact.inc"

/* dyn == NULL: the kinematic limit (K4).  dyn != NULL (SURVEY 8 f4b stage A): `_do_simulation` is the contact-free servo
 * dynamics of mopa_oracle_dyn.inc -- nsub sub-steps towards ctrl, qvel / bias_lag carried per env; obs then reports the
 * joint velocities, and the reference's order is kept: physics -> reward / obs -> joint-limit clamp (env/base.py:269-290;
 * a no-op here, the sub-steps stop a joint at the same limits). */
static void env_step_impl(const OrcScene *s, const OrcEnvDesc *d, const OrcDynDesc *dyn, double *qvel, double *bias_lag,
                          double *qpos, double *prev_state, uint8_t *has_prev,
                          int32_t *ep_len, const double *action, int is_planner, int move, double *obs, double *reward,
                          uint8_t *done, uint8_t *success) {
    const int na = d->n_arm;
    if (action && (move & 2)) return;   /* flags: bit 0 = the actuated joints move, bit 1 = env sits this step out */
    move &= 1;
    if (action) {
        double ctrl[16];
        for (int j = 0; j < na; j++) {
            const int adr = d->arm_qpos_idx[j];
            const double prev = (is_planner && *has_prev) ? prev_state[j] : qpos[adr];
            const double a = is_planner ? action[j] : action[j] * d->ac_scale;
            /* Pusher (pusher_obstacle.py:262-266): `desired_state = self._prev_state + action` overrides both scaled forms */
            const double desired = d->kind == 3 ? prev + action[j] : prev + clampd(a, -d->ac_scale, d->ac_scale);
            ctrl[j] = desired;
            prev_state[j] = desired;
        }
        for (int k = na; k < d->n_act; k++) ctrl[k] = qpos[d->act_qpos_idx[k]] + action[na];   /* Lift: gripper_state + action[-1] */
        if (dyn) {
            if (move) {
                double dctrl[ORC_DYN_MAX];
                for (int i = 0; i < dyn->nd; i++) dctrl[i] = 0.0;
                for (int k = 0; k < d->n_act; k++)
                    for (int i = 0; i < dyn->nd; i++)
                        if (dyn->qadr[i] == d->act_qpos_idx[k]) dctrl[i] = clampd(ctrl[k], d->act_lo[k], d->act_hi[k]);
                if (dyn->ct) orc_ct_step(dyn, qpos, qvel, bias_lag, dctrl, dyn->nsub, NULL);
                else orc_dyn_step_obj(dyn, qpos, qvel, bias_lag, dctrl, dyn->nsub, dyn->obj ? qvel + dyn->nd : NULL);
            }
            *has_prev = 1;
        } else {
        if (move)
            for (int k = 0; k < d->n_act; k++) qpos[d->act_qpos_idx[k]] = clampd(ctrl[k], d->act_lo[k], d->act_hi[k]);
        *has_prev = 1;
        if (d->kind != 3)       /* (Pusher: no servo range stops a joint at its limit -- the clamp comes after the obs, as in the reference) */
        for (int i = 0; i < s->nq; i++)
            if (d->qpos_limited[i]) qpos[i] = clampd(qpos[i], d->qpos_min[i], d->qpos_max[i]);
        }
    }
    double *buf = (double *)malloc(sizeof(double) * 16 * s->nbody);
    double *xpos = buf, *xquat = buf + 3 * s->nbody, *xmat = buf + 7 * s->nbody;
    fk_bodies(s, qpos, xpos, xquat, xmat, 0);
    double P[8][3];
    for (int f = 0; f < d->n_frames; f++) frame_pos(P[f], xpos, xmat, d->frame_body[f], d->frame_off + 3 * f);
    const double *eef = P[0];
    const double *eq = xquat + 4 * d->quat_body[0], *oq = xquat + 4 * d->quat_body[1];
    double r = 0.0;
    int succ = 0;
    int o = 0;
    if (d->kind == 3) {
        /* PusherObstacle (env/pusher/pusher_obstacle.py:183-204 `_get_obs`, :223-245 `compute_reward`); frames: 0 site fingertip,
         * 1 body fingertip, 2 body box, 3 body target; the box / goal sliders are the last four qpos entries (:187,:202) */
        const double *tip_site = P[0], *tip = P[1], *box = P[2], *target = P[3];
        double v[3];
        for (int j = 0; j < na; j++) { double sn, cs; orc_sincos(qpos[d->arm_qpos_idx[j]], &sn, &cs); obs[j] = cs; obs[na + j] = sn; }
        o = 2 * na;
        obs[o++] = qpos[s->nq - 2]; obs[o++] = qpos[s->nq - 1];                      /* box qpos */
        for (int j = 0; j < na + 2; j++) obs[o++] = 0.0;                             /* joint and box velocities: the kinematic limit */
        obs[o++] = tip[0]; obs[o++] = tip[1];                                        /* "fingertip": body position, xy */
        obs[o++] = qpos[s->nq - 4]; obs[o++] = qpos[s->nq - 3];                      /* "goal" */
        sub3(v, box, tip_site);
        const double dist_box_to_gripper = norm3(v);
        sub3(v, box, target);
        const double box_to_target = norm3(v);
        double reward_reach = 0.0, reward_push = 0.0;
        if (dist_box_to_gripper < 0.1) reward_reach = 0.1 * (1.0 - orc_tanh_pos(5.0 * dist_box_to_gripper));
        if (box_to_target < 0.1) reward_push = 0.3 * (1.0 - orc_tanh_pos(5.0 * box_to_target));
        r = reward_reach + reward_push;
        if (box_to_target < d->distance_threshold) { r += d->success_reward; succ = 1; }
    } else {
    for (int j = 0; j < na; j++) obs[o++] = qpos[d->arm_qpos_idx[j]];          /* joint_pos */
    for (int j = 0; j < na; j++) obs[o++] = dyn ? dyn_vel_of(dyn, qvel, d->arm_qpos_idx[j]) : 0.0;          /* joint_vel */
    for (int j = 0; j < d->n_grip; j++) obs[o++] = qpos[d->grip_qpos_idx[j]];  /* gripper_qpos */
    for (int j = 0; j < d->n_grip; j++) obs[o++] = dyn ? dyn_vel_of(dyn, qvel, d->grip_qpos_idx[j]) : 0.0;  /* gripper_qvel */
    for (int i = 0; i < 3; i++) obs[o++] = eef[i];                             /* eef_pos */
    obs[o++] = eq[1]; obs[o++] = eq[2]; obs[o++] = eq[3]; obs[o++] = eq[0];    /* eef_quat, xyzw */
    if (d->kind == 0) {
        /* frames: 1 right_eef, 2 left_eef, 3 cube, 4 target */
        const double *rf = P[1], *lf = P[2], *cube = P[3], *target = P[4];
        double grip[3], g2c[3];
        for (int i = 0; i < 3; i++) grip[i] = (rf[i] + lf[i]) / 2.0;
        sub3(g2c, cube, grip);
        const double gripper_to_cube = norm3(g2c);
        const double c2t0 = cube[0] - target[0], c2t1 = cube[1] - target[1];
        const double cube_to_target = sqrt(fma(c2t1, c2t1, c2t0 * c2t0));
        double reward_reach = 0.0, reward_push = 0.0;
        if (gripper_to_cube < 0.1) reward_reach = 0.1 * (1.0 - orc_tanh_pos(10.0 * gripper_to_cube));
        if (cube_to_target < 0.1) reward_push = 0.5 * (1.0 - orc_tanh_pos(5.0 * cube_to_target));
        r = reward_push + reward_reach;
        if (cube_to_target < d->distance_threshold) { r += d->success_reward; succ = 1; }
        for (int i = 0; i < 3; i++) obs[o++] = target[i];                          /* target_pos */
        for (int i = 0; i < 3; i++) obs[o++] = cube[i];                            /* cube_pos */
        obs[o++] = oq[1]; obs[o++] = oq[2]; obs[o++] = oq[3]; obs[o++] = oq[0];    /* cube_quat, xyzw */
        for (int i = 0; i < 3; i++) obs[o++] = eef[i] - cube[i];                   /* gripper_to_cube */
        obs[o++] = c2t0; obs[o++] = c2t1;                                          /* cube_to_target */
    } else if (d->kind == 1) {
        /* frames: 1 cube (the can), 2 bin1 */
        const double *cube = P[1], *bin = P[2];
        double g2c[3];
        sub3(g2c, cube, eef);
        const double reward_reach = (1.0 - orc_tanh_pos(10.0 * norm3(g2c))) * 0.1;
        int touch_l = 0, touch_r = 0;
        if (action) {
            for (int k = 1; k < d->n_touch; k++) {
                const int hit = env_touch(s, d->touch_geom[0], d->touch_geom[k], xpos, xquat, xmat);
                if (k <= d->n_touch_left) touch_l |= hit; else touch_r |= hit;
            }
        }
        const double reward_grasp = (touch_l && touch_r) ? 0.35 : 0.0;
        double reward_lift = 0.0;
        const double z_target = bin[2] + 0.45;
        if (reward_grasp > 0.0) {
            const double z_dist = dmax(z_target - cube[2], 0.0);
            reward_lift = 0.35 + (1.0 - orc_tanh_pos(15.0 * z_dist)) * (0.5 - 0.35);
        }
        r = dmax(dmax(reward_reach, reward_grasp), reward_lift);
        if (reward_grasp > 0.0 && fabs(cube[2] - z_target) < 0.05) { r += d->success_reward; succ = 1; }
        for (int i = 0; i < 3; i++) obs[o++] = cube[i];                            /* cube_pos */
        obs[o++] = oq[1]; obs[o++] = oq[2]; obs[o++] = oq[3]; obs[o++] = oq[0];    /* cube_quat, xyzw */
        for (int i = 0; i < 3; i++) obs[o++] = eef[i] - cube[i];                   /* gripper_to_cube */
    } else {
        /* frames: 1 hole, 2 hole_bottom, 3 pegHead, 4 pegEnd; quat 1 = body "peg" (wxyz, as _get_quat returns it) */
        const double *hole = P[1], *bottom = P[2], *head = P[3], *end = P[4];
        double dh[3], db[3];
        sub3(dh, head, hole);
        sub3(db, head, bottom);
        const double dist = norm3(dh), dist_bottom = norm3(db);
        if (dist < 0.3) r = 0.4 * (1.0 - orc_tanh_pos(15.0 * dist));
        if (dist_bottom < 0.025) { r += d->success_reward; succ = 1; }
        for (int i = 0; i < 3; i++) obs[o++] = hole[i];
        for (int i = 0; i < 3; i++) obs[o++] = head[i];
        for (int i = 0; i < 3; i++) obs[o++] = end[i];
        obs[o++] = oq[0]; obs[o++] = oq[1]; obs[o++] = oq[2]; obs[o++] = oq[3];
    }
    }
    if (action) {
        *ep_len += 1;
        *reward = r;
        *success = (uint8_t)succ;
        *done = (uint8_t)(succ || *ep_len == d->max_episode_steps);
        if (dyn || d->kind == 3)     /* `_after_step`'s clamp, after reward / obs as in the reference */
            for (int i = 0; i < s->nq; i++)
                if (d->qpos_limited[i]) qpos[i] = clampd(qpos[i], d->qpos_min[i], d->qpos_max[i]);
    }
    free(buf);
}

void orc_env_step(const OrcScene *s, const OrcEnvDesc *d, double *qpos, double *prev_state, uint8_t *has_prev,
                  int32_t *ep_len, const double *action, int is_planner, int move, double *obs, double *reward,
                  uint8_t *done, uint8_t *success) {
    env_step_impl(s, d, NULL, NULL, NULL, qpos, prev_state, has_prev, ep_len, action, is_planner, move, obs, reward, done, success);
}

void orc_env_step_dyn(const OrcScene *s, const OrcEnvDesc *d, const OrcDynDesc *dyn, double *qpos, double *qvel, double *bias_lag,
                      double *prev_state, uint8_t *has_prev, int32_t *ep_len, const double *action, int is_planner, int move,
                      double *obs, double *reward, uint8_t *done, uint8_t *success) {
    env_step_impl(s, d, dyn, qvel, bias_lag, qpos, prev_state, has_prev, ep_len, action, is_planner, move, obs, reward, done, success);
}

void orc_env_step_dyn_batch(const OrcScene *s, const OrcEnvDesc *d, const OrcDynDesc *dyn, int64_t E, double *qpos, double *qvel,
                            double *bias_lag, double *prev_state, uint8_t *has_prev, int32_t *ep_len, const double *action,
                            int is_planner, const uint8_t *move_mask, double *obs, double *reward, uint8_t *done,
                            uint8_t *success, int nthreads) {
    const int od = orc_env_obs_dim(d), ad = orc_env_action_dim(d);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int64_t e = 0; e < E; e++)
        env_step_impl(s, d, dyn, qvel + e * (dyn->nd + ((dyn->obj || dyn->ct) ? 6 : 0)), bias_lag + e * dyn->nd, qpos + e * s->nq, prev_state + e * d->n_arm,
                      has_prev + e, ep_len + e, action ? action + e * ad : NULL, is_planner, move_mask ? move_mask[e] : 1,
                      obs + e * od, reward + e, done + e, success + e);
}

/* E envs, rows contiguous; OpenMP across envs (bench.py cpu_baseline of the env-step metric) */
void orc_env_step_batch(const OrcScene *s, const OrcEnvDesc *d, int64_t E, double *qpos, double *prev_state, uint8_t *has_prev,
                        int32_t *ep_len, const double *action, int is_planner, const uint8_t *move_mask, double *obs,
                        double *reward, uint8_t *done, uint8_t *success, int nthreads) {
    const int od = orc_env_obs_dim(d), ad = orc_env_action_dim(d);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int64_t e = 0; e < E; e++)
        orc_env_step(s, d, qpos + e * s->nq, prev_state + e * d->n_arm, has_prev + e, ep_len + e,
                     action ? action + e * ad : NULL, is_planner, move_mask ? move_mask[e] : 1, obs + e * od,
                     reward + e, done + e, success + e);
}

/* ------------------------------------------------------------------ */
/*  damped-least-squares IK of a site position        */
/* ------------------------------------------------------------------ */
/* Restates reference env/inverse_kinematics.py:18-135 (`qpos_from_site_pose`; position target, or position + orientation
 * target: 6 x n Jacobian [jacp; jacr], err = [target_pos - site_xpos; quat2Vel(target_quat * conj(site_xquat))],
 * err_norm = |err_pos| + rot_weight |err_rot|, :38-44,88-107) with
 * `nullspace_method` :274-281: per iteration  err = target - site_xpos;  stop (success) if |err| < tol;
 * J = site position Jacobian w.r.t. the movable joints ([3P] mj_jacSite: hinge column = axis x (p_site - anchor),
 * slide column = axis);  dq = (J^T J + lambda I)^-1 J^T err  (the reference passes `regularization_strength`
 * unconditionally, :113-115, so lambda is always on);  stop (failure) if |err| / |dq| > progress_thresh;
 * |dq| capped at max_update_norm;  qpos[movable] += dq;  forward kinematics.
 * The reference solves with np.linalg.solve (LAPACK LU); J^T J + lambda I is symmetric positive definite and this
 * restatement uses an unpivoted Cholesky factorisation in a fixed operation order (shared with the HIP kernel). */
#define ORC_IK_MAXJ 8

/* (round 6: ONE division per pivot -- inv[j] = 1 / L[j][j] -- and products with it where 44 divisions stood: the kernel's iteration is a chain
 * of dependent operations, a division costs a dozen of them.  Results move by ulps; the reference's own solve is LAPACK LU anyway.) */
static void ik_chol_solve(double H[ORC_IK_MAXJ][ORC_IK_MAXJ], const double *g, double *x) {
    double L[ORC_IK_MAXJ][ORC_IK_MAXJ], y[ORC_IK_MAXJ], inv[ORC_IK_MAXJ];
    for (int j = 0; j < ORC_IK_MAXJ; j++) {
        double d = H[j][j];
        for (int k = 0; k < j; k++) d = fma(-L[j][k], L[j][k], d);
        L[j][j] = sqrt(d);
        inv[j] = 1.0 / L[j][j];
        for (int i = j + 1; i < ORC_IK_MAXJ; i++) {
            double s = H[i][j];
            for (int k = 0; k < j; k++) s = fma(-L[i][k], L[j][k], s);
            L[i][j] = s * inv[j];
        }
    }
    for (int i = 0; i < ORC_IK_MAXJ; i++) {
        double s = g[i];
        for (int k = 0; k < i; k++) s = fma(-L[i][k], y[k], s);
        y[i] = s * inv[i];
    }
    for (int i = ORC_IK_MAXJ - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < ORC_IK_MAXJ; k++) s = fma(-L[k][i], x[k], s);
        x[i] = s * inv[i];
    }
}

void orc_ik_solve(const OrcScene *s, int n_joints, const int32_t *joint_ids, int site_body, const double *site_off,
                  const double *site_quat, double *qpos, const double *target_pos, const double *target_quat, double rot_weight,
                  int max_steps, double tol, double max_update_norm, double progress_thresh, double reg_strength,
                  double *err_norm_out, int32_t *steps_out, uint8_t *success_out) {
    double *buf = (double *)malloc(sizeof(double) * 16 * s->nbody);
    double *xpos = buf, *xquat = buf + 3 * s->nbody, *xmat = buf + 7 * s->nbody;
    double err_norm = 0.0;
    int steps = 0, success = 0;
    const int nrow = target_quat ? 6 : 3;
    double smat_local[9];
    if (site_quat) quat2mat(smat_local, site_quat);
    for (steps = 0; steps < max_steps; steps++) {
        fk_bodies(s, qpos, xpos, xquat, xmat, 0);
        double psite[3], err[6] = {0, 0, 0, 0, 0, 0};
        frame_pos(psite, xpos, xmat, site_body, site_off);
        sub3(err, target_pos, psite);
        err_norm = norm3(err);
        if (target_quat) {
            /* :88-93  site_xquat = mat2Quat(site_xmat); err_rot = quat2Vel(target_quat * conj(site_xquat), 1) */
            double smat[9], sq[4], nq[4], eq[4];
            const double *bm = xmat + 9 * site_body;
            if (site_quat) {
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++)
                        smat[3 * r + c] = fma(bm[3 * r + 2], smat_local[6 + c], fma(bm[3 * r + 1], smat_local[3 + c], bm[3 * r] * smat_local[c]));
            } else memcpy(smat, bm, sizeof smat);
            mj_mat2quat(sq, smat);
            nq[0] = sq[0]; nq[1] = -sq[1]; nq[2] = -sq[2]; nq[3] = -sq[3];
            quat_mul(eq, target_quat, nq);
            mj_quat2vel(err + 3, eq);
            err_norm = err_norm + norm3(err + 3) * rot_weight;
        }
        if (err_norm < tol) { success = 1; break; }
        double J[6][ORC_IK_MAXJ];
        for (int k = 0; k < ORC_IK_MAXJ; k++) for (int r = 0; r < 6; r++) J[r][k] = 0.0;
        for (int k = 0; k < n_joints; k++) {
            const int j = joint_ids[k];
            /* owning body and whether it is an ancestor (or the body) of the site's body */
            int jb = -1;
            for (int b = 0; b < s->nbody; b++)
                if (j >= s->body_jntadr[b] && j < s->body_jntadr[b] + s->body_jntnum[b]) jb = b;
            int on_chain = 0;
            for (int b = site_body; b > 0; b = s->body_parent[b]) if (b == jb) on_chain = 1;
            if (!on_chain) continue;
            /* world axis / anchor of the joint.  For the joints of one body MuJoCo applies them in order; the
             * supported robots carry one joint per body, for which axis and anchor follow from the body pose */
            double axis[3], anchor[3], r[3], c[3];
            mat_vec(axis, xmat + 9 * jb, s->jnt_axis + 3 * j);
            frame_pos(anchor, xpos, xmat, jb, s->jnt_pos + 3 * j);
            if (s->jnt_type[j] == J_SLIDE) { c[0] = axis[0]; c[1] = axis[1]; c[2] = axis[2]; }
            else { sub3(r, psite, anchor); cross3(c, axis, r); }
            J[0][k] = c[0]; J[1][k] = c[1]; J[2][k] = c[2];
            if (s->jnt_type[j] != J_SLIDE) { J[3][k] = axis[0]; J[4][k] = axis[1]; J[5][k] = axis[2]; }   /* [3P] jacr: hinge = axis, slide = 0 */
        }
        double H[ORC_IK_MAXJ][ORC_IK_MAXJ], g[ORC_IK_MAXJ], x[ORC_IK_MAXJ];
        for (int i = 0; i < ORC_IK_MAXJ; i++) {
            for (int j2 = 0; j2 < ORC_IK_MAXJ; j2++) {
                double h = fma(J[2][i], J[2][j2], fma(J[1][i], J[1][j2], J[0][i] * J[0][j2]));
                if (nrow == 6) h = fma(J[5][i], J[5][j2], fma(J[4][i], J[4][j2], fma(J[3][i], J[3][j2], h)));
                H[i][j2] = h;
            }
            H[i][i] = (i < n_joints) ? H[i][i] + reg_strength : 1.0;   /* padding rows: identity */
            double gi = fma(J[2][i], err[2], fma(J[1][i], err[1], J[0][i] * err[0]));
            if (nrow == 6) gi = fma(J[5][i], err[5], fma(J[4][i], err[4], fma(J[3][i], err[3], gi)));
            g[i] = gi;
        }
        ik_chol_solve(H, g, x);
        double un2 = 0.0;
        for (int i = 0; i < n_joints; i++) un2 = fma(x[i], x[i], un2);
        const double update_norm = sqrt(un2);
        if (err_norm / update_norm > progress_thresh) break;
        double sc = 1.0;
        if (update_norm > max_update_norm) sc = max_update_norm / update_norm;
        for (int k = 0; k < n_joints; k++) {
            const int adr = s->jnt_qposadr[joint_ids[k]];
            qpos[adr] = qpos[adr] + ((update_norm > max_update_norm) ? x[k] * sc : x[k]);
        }
    }
    if (steps == max_steps) steps = max_steps - 1;   /* python: `for steps in range(max_steps)` leaves the last index */
    *err_norm_out = err_norm; *steps_out = steps; *success_out = (uint8_t)success;
    free(buf);
}
