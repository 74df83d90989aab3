// mopa_device.hpp -- FP64 geometry for the state-validity kernels (gfx950).
//
// Numerics contract (DESIGN.md "Numerics"): IEEE double everywhere, compiled
// with -ffp-contract=off so that the ONLY fused multiply-adds are the explicit
// fma() calls below; sqrt and division are IEEE correctly rounded; sin/cos is
// mopa_sincos() (Cody-Waite + minimax kernels), never a library call.  With
// that, a state's collision verdict is a pure function of its inputs and is
// reproduced bit-for-bit by any conforming implementation -- which is what the
// parity tests check against the independent CPU restatement.
//
// What each routine stands in for in the reference stack ([3P] = third-party
// library the reference links, source not in its tree):
//   fk_*            [3P] MuJoCo mj_kinematics   (via mj_fwdPosition,
//                   reference motion_planners/src/mujoco_ompl_interface.cpp:932)
//   d_* / mpr_*     [3P] MuJoCo mj_collision narrow phase (mjc_* primitives,
//                   mjc_Convex -> libccd ccdMPRPenetration)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define MOPA_HD __host__ __device__ __forceinline__
#define MOPA_D __device__ __forceinline__

namespace mopa {

constexpr double kFar = 1.0e10;
constexpr double kMinVal = 1e-15;          // mjMINVAL
constexpr double kPi = 3.14159265358979323846;
constexpr double kCcdEps = 2.2204460492503131e-16;
constexpr double kMprTol = 1e-6;           // MuJoCo mpr_tolerance
constexpr int kMprMaxIt = 50;              // MuJoCo mpr_iterations
constexpr int kMprPortalMaxIt = 100;       // guard for libccd's two unbounded loops

enum GeomType : int { G_PLANE = 0, G_SPHERE = 2, G_CAPSULE = 3, G_CYLINDER = 5, G_BOX = 6, G_MESH = 7 };
enum JntType : int { J_FREE = 0, J_BALL = 1, J_SLIDE = 2, J_HINGE = 3 };

// pair type codes, cheapest narrow phase first (the pair list is sorted by this)
enum PairCode : int {
    PC_PLANE_SPHERE = 0, PC_PLANE_CAPSULE, PC_PLANE_CYLINDER, PC_PLANE_BOX, PC_SPHERE_SPHERE, PC_SPHERE_CAPSULE,
    PC_SPHERE_CYLINDER, PC_SPHERE_BOX, PC_CAPSULE_CAPSULE, PC_CAPSULE_BOX, PC_BOX_BOX, PC_CONVEX, PC_PLANE_MESH, PC_CONVEX_MESH,
    PC_COUNT
};

struct V3 { double x, y, z; };
struct Q4 { double w, x, y, z; };

MOPA_HD double dot3(V3 a, V3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
MOPA_HD V3 cross3(V3 a, V3 b) {
    return V3{fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x))};
}
MOPA_HD V3 sub3(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MOPA_HD V3 add3(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MOPA_HD V3 neg3(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// a + b*s
MOPA_HD V3 addscl3(V3 a, V3 b, double s) { return V3{fma(b.x, s, a.x), fma(b.y, s, a.y), fma(b.z, s, a.z)}; }
MOPA_HD double norm3(V3 a) { return sqrt(dot3(a, a)); }
MOPA_HD V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
MOPA_HD void st3(double *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
// M v, M row-major 3x3
MOPA_HD V3 mat_vec(const double *M, V3 v) {
    return V3{fma(M[2], v.z, fma(M[1], v.y, M[0] * v.x)), fma(M[5], v.z, fma(M[4], v.y, M[3] * v.x)),
              fma(M[8], v.z, fma(M[7], v.y, M[6] * v.x))};
}
// M^T v
MOPA_HD V3 matT_vec(const double *M, V3 v) {
    return V3{fma(M[6], v.z, fma(M[3], v.y, M[0] * v.x)), fma(M[7], v.z, fma(M[4], v.y, M[1] * v.x)),
              fma(M[8], v.z, fma(M[5], v.y, M[2] * v.x))};
}
MOPA_HD V3 col3(const double *M, int j) { return V3{M[j], M[3 + j], M[6 + j]}; }
MOPA_HD double dmin(double a, double b) { return (a < b) ? a : b; }
MOPA_HD double dmax(double a, double b) { return (a > b) ? a : b; }
MOPA_HD double clampd(double x, double lo, double hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); }
MOPA_HD double signd(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); }

MOPA_HD Q4 quat_mul(Q4 a, Q4 b) {
    Q4 r;
    r.w = fma(-a.z, b.z, fma(-a.y, b.y, fma(-a.x, b.x, a.w * b.w)));
    r.x = fma(-a.z, b.y, fma(a.y, b.z, fma(a.x, b.w, a.w * b.x)));
    r.y = fma(a.z, b.x, fma(a.y, b.w, fma(-a.x, b.z, a.w * b.y)));
    r.z = fma(a.z, b.w, fma(-a.y, b.x, fma(a.x, b.y, a.w * b.z)));
    return r;
}
// [3P] mju_normalize4
// A product of unit quaternions has |n - 1| <= 1e-15 nearly every time, and is then left alone: that case is decided on n^2 without the
// square root (~20 instructions on the critical path of every body of a kinematic chain).  sqrt is correctly rounded and monotonic, so
// { s : |sqrt(s) - 1| <= mjMINVAL } is an interval of doubles -- its end points below, checked value by value in tests/test_oracle_fk.py.
constexpr double kNormSqLo = 0x1.fffffffffffeep-1, kNormSqHi = 0x1.0000000000009p+0;
MOPA_HD Q4 quat_normalize(Q4 q) {
    const double s = fma(q.z, q.z, fma(q.y, q.y, fma(q.x, q.x, q.w * q.w)));
    if (s >= kNormSqLo && s <= kNormSqHi) return q;
    double n = sqrt(s);
    if (n < kMinVal) return Q4{1.0, 0.0, 0.0, 0.0};
    if (fabs(n - 1.0) > kMinVal) {
        double inv = 1.0 / n;
        q.w *= inv; q.x *= inv; q.y *= inv; q.z *= inv;
    }
    return q;
}
// [3P] mju_quat2Mat
MOPA_HD void quat2mat(double *M, Q4 q) {
    double q00 = q.w * q.w, q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z;
    double q11 = q.x * q.x, q12 = q.x * q.y, q13 = q.x * q.z;
    double q22 = q.y * q.y, q23 = q.y * q.z, q33 = q.z * q.z;
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
// [3P] mju_rotVecQuat
MOPA_HD V3 rot_vec_quat(V3 v, Q4 q) {
    double M[9];
    quat2mat(M, q);
    return mat_vec(M, v);
}

// Deterministic sin/cos: k = rint(x*2/pi); three-term Cody-Waite reduction;
// degree-13/14 minimax kernels on [-pi/4, pi/4]; quadrant fix-up.
MOPA_HD void mopa_sincos(double x, double &sout, double &cout) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P3 = 2.02226624879595063154e-21;
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
    int q = (int)((long long)k & 3);
    if (q == 0) { sout = sn; cout = cs; }
    else if (q == 1) { sout = cs; cout = -sn; }
    else if (q == 2) { sout = -sn; cout = -cs; }
    else { sout = -cs; cout = sn; }
}

// One hinge / slide joint applied to a body pose (the joint loop of [3P] mj_kinematics).  The off-centre correction
// of a hinge (anchor = pos + R jnt_pos before the rotation, pos = anchor - R' jnt_pos after it) is skipped when
// jnt_pos is exactly zero -- every joint of the reference's robots: it is then the identity (up to the sign of a
// zero) and costs two quat->matrix conversions per body.  The CPU checker makes the same choice, so results stay
// bit-identical.  `anchor_zero` must be `jp == (0,0,0)`.
// (sn, cs) = sin / cos of half the hinge angle, computed by the caller (mopa_sincos(0.5 * dq)): the wave-per-state
// path evaluates them for all joints of a state in parallel lanes before the serial walk down the kinematic chain
MOPA_HD void apply_joint_sc(int jt, V3 ax, V3 jp, bool anchor_zero, double dq, double sn, double cs, V3 &pos, Q4 &quat) {
    if (jt == J_SLIDE) {
        const V3 xaxis = rot_vec_quat(ax, quat);
        pos = addscl3(pos, xaxis, dq);
    } else if (jt == J_HINGE) {
        const Q4 ql{cs, ax.x * sn, ax.y * sn, ax.z * sn};
        if (anchor_zero) {
            quat = quat_mul(quat, ql);
        } else {
            const V3 xanchor = add3(rot_vec_quat(jp, quat), pos);
            quat = quat_mul(quat, ql);
            const V3 vec = rot_vec_quat(jp, quat);
            pos = sub3(xanchor, vec);
        }
    }
}
MOPA_HD void apply_joint(int jt, V3 ax, V3 jp, bool anchor_zero, double dq, V3 &pos, Q4 &quat) {
    double sn = 0.0, cs = 1.0;
    if (jt == J_HINGE) mopa_sincos(0.5 * dq, sn, cs);
    apply_joint_sc(jt, ax, jp, anchor_zero, dq, sn, cs, pos, quat);
}
MOPA_HD bool is_zero3(V3 v) { return v.x == 0.0 && v.y == 0.0 && v.z == 0.0; }

// Deterministic exp / tanh for the env reward (never libm / OCML, so CPU oracle and GPU agree bit for bit):
// k = rint(x*log2(e)); two-term Cody-Waite reduction (fdlibm's ln2 split); degree-13 Taylor kernel in Horner
// form on |r| <= ln2/2 (truncation error 4e-18); scaling by 2^k through the exponent bits (|x| < 700).
MOPA_HD double mopa_exp(double x) {
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
    union { unsigned long long u; double d; } sc;
    sc.u = (unsigned long long)((long long)k + 1023) << 52;
    return p * sc.d;
}
// tanh(x) for x >= 0 (absolute error < 3e-16): (1 - e^-2x) / (1 + e^-2x)
MOPA_HD double mopa_tanh_pos(double x) {
    const double t = mopa_exp(-2.0 * x);
    return (1.0 - t) / (1.0 + t);
}

// Deterministic atan2 (IK orientation error; same operation sequence as the CPU checker): a = min/max in [0,1], reduction
// atan(a) = atan(c) + atan((a - c) / (1 + a c)) with c in {0, tan(pi/8), 1} so that |t| <= tan(pi/16), odd Taylor series of
// atan(t) to t^25 in Horner form, quadrant fix-ups.  atan2(0, 0) = 0.
MOPA_HD double mopa_atan2(double y, double x) {
    const double PI = 3.14159265358979311600e+00, PI_2 = 1.57079632679489655800e+00;
    const double T1 = 1.98912367379658006912e-01, T3 = 6.68178637919298919998e-01;
    const double C1 = 4.14213562373095048802e-01, A1 = 3.92699081698724139500e-01, A2 = 7.85398163397448279000e-01;
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

// [3P] mju_mat2Quat: largest-component branch, then mju_normalize4
MOPA_HD Q4 mj_mat2quat(const double *m) {
    Q4 q;
    if (m[0] + m[4] + m[8] > 0.0) {
        q.w = 0.5 * sqrt(1.0 + m[0] + m[4] + m[8]);
        q.x = 0.25 * (m[7] - m[5]) / q.w;
        q.y = 0.25 * (m[2] - m[6]) / q.w;
        q.z = 0.25 * (m[3] - m[1]) / q.w;
    } else if (m[0] > m[4] && m[0] > m[8]) {
        q.x = 0.5 * sqrt(1.0 + m[0] - m[4] - m[8]);
        q.w = 0.25 * (m[7] - m[5]) / q.x;
        q.y = 0.25 * (m[1] + m[3]) / q.x;
        q.z = 0.25 * (m[2] + m[6]) / q.x;
    } else if (m[4] > m[8]) {
        q.y = 0.5 * sqrt(1.0 - m[0] + m[4] - m[8]);
        q.w = 0.25 * (m[2] - m[6]) / q.y;
        q.x = 0.25 * (m[1] + m[3]) / q.y;
        q.z = 0.25 * (m[5] + m[7]) / q.y;
    } else {
        q.z = 0.5 * sqrt(1.0 - m[0] - m[4] + m[8]);
        q.w = 0.25 * (m[3] - m[1]) / q.z;
        q.x = 0.25 * (m[2] + m[6]) / q.z;
        q.y = 0.25 * (m[5] + m[7]) / q.z;
    }
    return quat_normalize(q);
}
// [3P] mju_quat2Vel with dt = 1: rotation axis * angle, angle in (-pi, pi]
MOPA_HD V3 mj_quat2vel(Q4 q) {
    V3 ax{q.x, q.y, q.z};
    const double s = sqrt(fma(ax.z, ax.z, fma(ax.y, ax.y, ax.x * ax.x)));
    if (s < kMinVal) ax = V3{1.0, 0.0, 0.0};
    else if (fabs(s - 1.0) > kMinVal) { const double inv = 1.0 / s; ax.x *= inv; ax.y *= inv; ax.z *= inv; }
    double speed = 2.0 * mopa_atan2(s, q.w);
    if (speed > 3.14159265358979311600e+00) speed = speed - 2.0 * 3.14159265358979311600e+00;
    return V3{ax.x * speed, ax.y * speed, ax.z * speed};
}

// ---------------------------------------------------------------------------
// One posed primitive as the narrow phase sees it.  `p` points at 15 doubles:
// pos[3] mat[9] size[3] (LDS on the device, plain memory on the host).
// ---------------------------------------------------------------------------
constexpr int kGeomStride = 16;   // doubles per posed geom record (15 used, padded to 128 B)
constexpr int GO_POS = 0, GO_MAT = 3, GO_SIZE = 12;

MOPA_HD double d_plane_sphere(const double *P, const double *S) {
    V3 n = col3(P + GO_MAT, 2);
    V3 diff = sub3(ld3(S + GO_POS), ld3(P + GO_POS));
    return dot3(diff, n) - S[GO_SIZE];
}
MOPA_HD double d_plane_capsule(const double *P, const double *C) {
    V3 n = col3(P + GO_MAT, 2), a = col3(C + GO_MAT, 2);
    V3 cp = ld3(C + GO_POS), pp = ld3(P + GO_POS);
    V3 e = addscl3(cp, a, C[GO_SIZE + 1]);
    double d1 = dot3(sub3(e, pp), n) - C[GO_SIZE];
    e = addscl3(cp, a, -C[GO_SIZE + 1]);
    double d2 = dot3(sub3(e, pp), n) - C[GO_SIZE];
    return dmin(d1, d2);
}
MOPA_HD double d_plane_cylinder(const double *P, const double *C) {
    V3 n = col3(P + GO_MAT, 2), a = col3(C + GO_MAT, 2);
    V3 diff = sub3(ld3(C + GO_POS), ld3(P + GO_POS));
    double d0 = dot3(diff, n);
    double na = dot3(n, a);
    double s2 = fma(-na, na, 1.0);
    double sr = (s2 > 0.0) ? sqrt(s2) : 0.0;
    return (d0 - C[GO_SIZE + 1] * fabs(na)) - C[GO_SIZE] * sr;
}
MOPA_HD double d_plane_box(const double *P, const double *B) {
    V3 n = col3(P + GO_MAT, 2);
    V3 diff = sub3(ld3(B + GO_POS), ld3(P + GO_POS));
    double d0 = dot3(diff, n);
    double ext = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) ext = fma(fabs(dot3(n, col3(B + GO_MAT, i))), B[GO_SIZE + i], ext);
    return d0 - ext;
}
MOPA_HD double d_sphere_sphere(const double *A, const double *B) {
    V3 diff = sub3(ld3(B + GO_POS), ld3(A + GO_POS));
    return norm3(diff) - (A[GO_SIZE] + B[GO_SIZE]);
}
MOPA_HD double d_sphere_capsule(const double *S, const double *C) {
    V3 a = col3(C + GO_MAT, 2);
    V3 sp = ld3(S + GO_POS), cp = ld3(C + GO_POS);
    V3 vec = sub3(sp, cp);
    double x = clampd(dot3(a, vec), -C[GO_SIZE + 1], C[GO_SIZE + 1]);
    V3 pt = addscl3(cp, a, x);
    return norm3(sub3(sp, pt)) - (S[GO_SIZE] + C[GO_SIZE]);
}
MOPA_HD double d_capsule_capsule(const double *A, const double *B) {
    V3 a1 = col3(A + GO_MAT, 2), a2 = col3(B + GO_MAT, 2);
    V3 r = sub3(ld3(A + GO_POS), ld3(B + GO_POS));
    double h1 = A[GO_SIZE + 1], h2 = B[GO_SIZE + 1];
    double b = dot3(a1, a2), c = dot3(a1, r), f = dot3(a2, r);
    double denom = fma(-b, b, 1.0);
    double sp = 0.0;
    if (denom > 1e-12) sp = clampd(fma(b, f, -c) / denom, -h1, h1);
    double tp = fma(b, sp, f);
    if (tp < -h2) { tp = -h2; sp = clampd(fma(b, tp, -c), -h1, h1); }
    else if (tp > h2) { tp = h2; sp = clampd(fma(b, tp, -c), -h1, h1); }
    V3 w = addscl3(r, a1, sp);
    w = addscl3(w, a2, -tp);
    return norm3(w) - (A[GO_SIZE] + B[GO_SIZE]);
}
MOPA_HD double d_sphere_box(const double *S, const double *B) {
    V3 v = sub3(ld3(S + GO_POS), ld3(B + GO_POS));
    V3 l = matT_vec(B + GO_MAT, v);
    double hx = B[GO_SIZE], hy = B[GO_SIZE + 1], hz = B[GO_SIZE + 2];
    V3 e{l.x - clampd(l.x, -hx, hx), l.y - clampd(l.y, -hy, hy), l.z - clampd(l.z, -hz, hz)};
    bool inside = (e.x == 0.0) && (e.y == 0.0) && (e.z == 0.0);
    if (inside) {
        double m = dmin(dmin(hx - fabs(l.x), hy - fabs(l.y)), hz - fabs(l.z));
        return -m - S[GO_SIZE];
    }
    return norm3(e) - S[GO_SIZE];
}
MOPA_HD double d_sphere_cylinder(const double *S, const double *C) {
    V3 a = col3(C + GO_MAT, 2);
    V3 v = sub3(ld3(S + GO_POS), ld3(C + GO_POS));
    double z = dot3(v, a);
    V3 w = addscl3(v, a, -z);
    double rho = norm3(w);
    double dr = rho - C[GO_SIZE];
    double dz = fabs(z) - C[GO_SIZE + 1];
    double dp;
    if (dr <= 0.0 && dz <= 0.0) dp = dmax(dr, dz);
    else {
        double er = dmax(dr, 0.0), ez = dmax(dz, 0.0);
        dp = sqrt(fma(ez, ez, er * er));
    }
    return dp - S[GO_SIZE];
}

// exact segment/box distance: bracket the sign change of the piecewise-linear
// derivative of the convex function t -> |p(t) - clamp(p(t))|^2 among its knots.
MOPA_HD double segbox_half_fprime(V3 p0, V3 d, V3 h, double t, V3 &e) {
    double px = fma(d.x, t, p0.x), py = fma(d.y, t, p0.y), pz = fma(d.z, t, p0.z);
    e.x = px - clampd(px, -h.x, h.x);
    e.y = py - clampd(py, -h.y, h.y);
    e.z = pz - clampd(pz, -h.z, h.z);
    return dot3(e, d);
}
MOPA_HD void segbox_knot(V3 p0, V3 d, V3 h, double hi, double p0i, double di, double &tL, double &gL, double &tR, double &gR) {
    if (di == 0.0) return;
#pragma unroll
    for (int sg = 0; sg < 2; sg++) {
        double tk = ((sg ? hi : -hi) - p0i) / di;
        if (!(tk > 0.0 && tk < 1.0)) continue;
        V3 e;
        double gk = segbox_half_fprime(p0, d, h, tk, e);
        // branch-free update of the bracket (keeps the four values in registers)
        const bool upL = (gk < 0.0) && (tk > tL);
        const bool upR = !(gk < 0.0) && (tk < tR);
        tL = upL ? tk : tL; gL = upL ? gk : gL;
        tR = upR ? tk : tR; gR = upR ? gk : gR;
    }
}
MOPA_HD double d_capsule_box(const double *C, const double *B) {
    V3 h = ld3(B + GO_SIZE);
    V3 a = col3(C + GO_MAT, 2);
    V3 v = sub3(ld3(C + GO_POS), ld3(B + GO_POS));
    V3 cl = matT_vec(B + GO_MAT, v);
    V3 al = matT_vec(B + GO_MAT, a);
    double hh = C[GO_SIZE + 1];
    V3 p0 = addscl3(cl, al, -hh);
    V3 d{al.x * (2.0 * hh), al.y * (2.0 * hh), al.z * (2.0 * hh)};
    V3 e;
    double tstar;
    double g0 = segbox_half_fprime(p0, d, h, 0.0, e);
    if (g0 >= 0.0) tstar = 0.0;
    else {
        double g1 = segbox_half_fprime(p0, d, h, 1.0, e);
        if (g1 <= 0.0) tstar = 1.0;
        else {
            double tL = 0.0, gL = g0, tR = 1.0, gR = g1;
            segbox_knot(p0, d, h, h.x, p0.x, d.x, tL, gL, tR, gR);
            segbox_knot(p0, d, h, h.y, p0.y, d.y, tL, gL, tR, gR);
            segbox_knot(p0, d, h, h.z, p0.z, d.z, tL, gL, tR, gR);
            tstar = fma(tR - tL, (-gL) / (gR - gL), tL);
        }
    }
    segbox_half_fprime(p0, d, h, tstar, e);
    double dseg = norm3(e);
    if (dseg > 0.0) return dseg - C[GO_SIZE];
    // the axis segment pierces the box: SAT depth of segment vs box (+ radius)
    V3 m = addscl3(p0, d, 0.5);
    V3 hd{0.5 * d.x, 0.5 * d.y, 0.5 * d.z};
    double best = -kFar;
    best = dmax(best, fabs(m.x) - (h.x + fabs(hd.x)));
    best = dmax(best, fabs(m.y) - (h.y + fabs(hd.y)));
    best = dmax(best, fabs(m.z) - (h.z + fabs(hd.z)));
    const double mm[3] = {m.x, m.y, m.z}, aa[3] = {al.x, al.y, al.z}, hb[3] = {h.x, h.y, h.z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double l2 = fma(aa[k], aa[k], aa[j] * aa[j]);
        if (l2 < 1e-12) continue;
        double tl = fma(mm[k], aa[j], -(mm[j] * aa[k]));
        double ra = fma(hb[k], fabs(aa[j]), hb[j] * fabs(aa[k]));
        best = dmax(best, (fabs(tl) - ra) / sqrt(l2));
    }
    return best - C[GO_SIZE];
}

// 15-axis SAT: max normalised separation (<= 0: minus the minimum translation depth)
MOPA_HD double d_box_box(const double *A, const double *B) {
    const double *Am = A + GO_MAT, *Bm = B + GO_MAT;
    double R[3][3], AR[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            R[i][j] = fma(Am[6 + i], Bm[6 + j], fma(Am[3 + i], Bm[3 + j], Am[i] * Bm[j]));
            AR[i][j] = fabs(R[i][j]);
        }
    V3 tv = matT_vec(Am, sub3(ld3(B + GO_POS), ld3(A + GO_POS)));
    const double t[3] = {tv.x, tv.y, tv.z};
    const double ha[3] = {A[GO_SIZE], A[GO_SIZE + 1], A[GO_SIZE + 2]}, hb[3] = {B[GO_SIZE], B[GO_SIZE + 1], B[GO_SIZE + 2]};
    double best = -kFar;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double rb = fma(hb[2], AR[i][2], fma(hb[1], AR[i][1], hb[0] * AR[i][0]));
        best = dmax(best, fabs(t[i]) - (ha[i] + rb));
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double tb = fma(t[2], R[2][j], fma(t[1], R[1][j], t[0] * R[0][j]));
        double ra = fma(ha[2], AR[2][j], fma(ha[1], AR[1][j], ha[0] * AR[0][j]));
        best = dmax(best, fabs(tb) - (ra + hb[j]));
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            double l2 = fma(R[i2][j], R[i2][j], R[i1][j] * R[i1][j]);
            if (l2 < 1e-12) continue;
            double tl = fma(t[i2], R[i1][j], -(t[i1] * R[i2][j]));
            double ra = fma(ha[i2], AR[i1][j], ha[i1] * AR[i2][j]);
            double rb = fma(hb[j2], AR[i][j1], hb[j1] * AR[i][j2]);
            best = dmax(best, (fabs(tl) - (ra + rb)) / sqrt(l2));
        }
    }
    return best;
}

// ---------------------------------------------------------------------------
// [3P] libccd MPR penetration with MuJoCo's support functions
// ---------------------------------------------------------------------------
MOPA_HD bool is_zero(double x) { return fabs(x) < kCcdEps; }
MOPA_HD bool ccd_eq(double a, double b) {
    double ab = fabs(a - b);
    if (ab < kCcdEps) return true;
    a = fabs(a); b = fabs(b);
    if (b > a) return ab < kCcdEps * b;
    return ab < kCcdEps * a;
}
MOPA_HD bool vec_eq0(V3 a) { return ccd_eq(a.x, 0.0) && ccd_eq(a.y, 0.0) && ccd_eq(a.z, 0.0); }
MOPA_HD V3 normalize3(V3 v) {
    double inv = 1.0 / norm3(v);
    return V3{v.x * inv, v.y * inv, v.z * inv};
}
// A mesh geom's record carries, instead of a size, where its convex hull lives: size[0] = offset (in doubles) of the
// vertex array inside the scene's double blob `aux`, size[1] = number of vertices.
// [3P] mjccd_support for a mesh: exhaustive search over the hull vertices, first maximum wins.
// G > 1 (device only): G neighbouring lanes (an aligned group) evaluate the SAME query and share the vertex search -- lane
// `sub` of the group takes vertices sub, sub + G, ... and the group's (value, index) pairs are folded with "larger value,
// then lower index", which is the first maximum of the sequential scan.  All G lanes return the same vertex.
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ double dpp_quad_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int G, bool MAXIMUM>
__device__ __forceinline__ void coop_fold(double &bd, int &best) {
    if (G >= 2) {
        const double od = dpp_quad_f64<0xB1>(bd);               // quad_perm [1,0,3,2]: lane ^ 1
        const int oi = __builtin_amdgcn_update_dpp(0, best, 0xB1, 0xf, 0xf, false);
        if ((MAXIMUM ? od > bd : od < bd) || (od == bd && oi < best)) { bd = od; best = oi; }
    }
    if (G >= 4) {
        const double od = dpp_quad_f64<0x4E>(bd);               // quad_perm [2,3,0,1]: lane ^ 2
        const int oi = __builtin_amdgcn_update_dpp(0, best, 0x4E, 0xf, 0xf, false);
        if ((MAXIMUM ? od > bd : od < bd) || (od == bd && oi < best)) { bd = od; best = oi; }
    }
    // (the four lanes of a quad now agree; a mirror pairs whole quads, whichever of their lanes meet)
    if (G >= 8) {
        const double od = dpp_quad_f64<0x141>(bd);              // row_half_mirror: lane 7 - l of the aligned eight
        const int oi = __builtin_amdgcn_update_dpp(0, best, 0x141, 0xf, 0xf, false);
        if ((MAXIMUM ? od > bd : od < bd) || (od == bd && oi < best)) { bd = od; best = oi; }
    }
    if (G >= 16) {
        const double od = dpp_quad_f64<0x140>(bd);              // row_mirror: lane 15 - l of the row
        const int oi = __builtin_amdgcn_update_dpp(0, best, 0x140, 0xf, 0xf, false);
        if ((MAXIMUM ? od > bd : od < bd) || (od == bd && oi < best)) { bd = od; best = oi; }
    }
}
#endif
template <int G = 1>
MOPA_HD V3 mesh_support_local(const double *g, V3 ld, const double *aux) {
    const double *V = aux + (int)g[GO_SIZE];
    const int n = (int)g[GO_SIZE + 1];
#if defined(__HIP_DEVICE_COMPILE__)
    if (G > 1) {
        static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "groups are aligned pairs / quads / eights / rows of lanes");
        const int sub = (int)(threadIdx.x & (G - 1));
        int best = sub < n ? sub : 0;
        double bd = dot3(ld, ld3(V + 3 * best));
#pragma unroll 4
        for (int i = best + G; i < n; i += G) {
            const double d = dot3(ld, ld3(V + 3 * i));
            if (d > bd) { bd = d; best = i; }
        }
        coop_fold<G, true>(bd, best);
        return ld3(V + 3 * best);
    }
#endif
    int best = 0;
    double bd = dot3(ld, ld3(V));
    for (int i = 1; i < n; i++) {
        const double d = dot3(ld, ld3(V + 3 * i));
        if (d > bd) { bd = d; best = i; }
    }
    return ld3(V + 3 * best);
}
// MESH = true: the instantiation used for primitive-vs-mesh pairs (kept apart so that scenes without meshes run
// exactly the code they ran before)
template <bool MESH = false, int G = 1>
MOPA_HD V3 support_geom(const double *g, int type, V3 dir, const double *aux = nullptr) {
    V3 ld = matT_vec(g + GO_MAT, dir);
    V3 lr;
    if (MESH && type == G_MESH) {
        lr = mesh_support_local<G>(g, ld, aux);
    } else if (type == G_CAPSULE) {
        lr.x = ld.x * g[GO_SIZE]; lr.y = ld.y * g[GO_SIZE];
        lr.z = fma(ld.z, g[GO_SIZE], signd(ld.z) * g[GO_SIZE + 1]);
    } else if (type == G_CYLINDER) {
        double tmp = sqrt(fma(ld.y, ld.y, ld.x * ld.x));
        if (tmp > kMinVal) {
            double sc = g[GO_SIZE] / tmp;
            lr.x = ld.x * sc; lr.y = ld.y * sc;
        } else { lr.x = 0.0; lr.y = 0.0; }
        lr.z = signd(ld.z) * g[GO_SIZE + 1];
    } else if (type == G_BOX) {
        lr.x = signd(ld.x) * g[GO_SIZE]; lr.y = signd(ld.y) * g[GO_SIZE + 1]; lr.z = signd(ld.z) * g[GO_SIZE + 2];
    } else {
        lr.x = ld.x * g[GO_SIZE]; lr.y = ld.y * g[GO_SIZE]; lr.z = ld.z * g[GO_SIZE];
    }
    return add3(mat_vec(g + GO_MAT, lr), ld3(g + GO_POS));
}
template <bool MESH = false, int G = 1>
MOPA_HD V3 support_md(const double *g1, int t1, const double *g2, int t2, V3 dir, const double *aux = nullptr) {
    V3 s1 = support_geom<false>(g1, t1, dir);          // meshes have the highest type code: always geom 2
    V3 s2 = support_geom<MESH, G>(g2, t2, neg3(dir), aux);
    return sub3(s1, s2);
}
MOPA_HD V3 portal_dir(V3 v1, V3 v2, V3 v3) { return normalize3(cross3(sub3(v2, v1), sub3(v3, v1))); }
MOPA_HD bool portal_reach_tol(V3 v1, V3 v2, V3 v3, V3 v4, V3 dir) {
    double dv1 = dot3(v1, dir), dv2 = dot3(v2, dir), dv3 = dot3(v3, dir), dv4 = dot3(v4, dir);
    double m = dmin(dmin(dv4 - dv1, dv4 - dv2), dv4 - dv3);
    return ccd_eq(m, kMprTol) || m < kMprTol;
}
MOPA_HD void expand_portal(V3 v0, V3 &v1, V3 &v2, V3 &v3, V3 v4) {
    V3 v4v0 = cross3(v4, v0);
    double dot = dot3(v1, v4v0);
    if (dot > 0.0) {
        dot = dot3(v2, v4v0);
        if (dot > 0.0) v1 = v4; else v3 = v4;
    } else {
        dot = dot3(v3, v4v0);
        if (dot > 0.0) v2 = v4; else v1 = v4;
    }
}
// squared distance of the origin to segment x0-b / triangle x0-B-C (libccd, witness form)
MOPA_HD double origin_seg_dist2(V3 x0, V3 b) {
    V3 d = sub3(b, x0);
    V3 a = x0;
    double t = -1.0 * dot3(a, d);
    t = t / dot3(d, d);
    if (t < 0.0 || is_zero(t)) return dot3(x0, x0);
    if (t > 1.0 || ccd_eq(t, 1.0)) return dot3(b, b);
    V3 w = addscl3(a, d, t);
    return dot3(w, w);
}
MOPA_HD double origin_tri_dist2(V3 x0, V3 B, V3 C) {
    V3 d1 = sub3(B, x0), d2 = sub3(C, x0), a = x0;
    double v = dot3(d1, d1), w = dot3(d2, d2);
    double p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
    double s = fma(q, r, -(w * p)) / fma(w, v, -(r * r));
    double t = (fma(-s, r, -q)) / w;
    if ((is_zero(s) || s > 0.0) && (ccd_eq(s, 1.0) || s < 1.0) && (is_zero(t) || t > 0.0) &&
        (ccd_eq(t, 1.0) || t < 1.0) && (ccd_eq(t + s, 1.0) || t + s < 1.0)) {
        V3 wv = addscl3(a, d1, s);
        wv = addscl3(wv, d2, t);
        return dot3(wv, wv);
    }
    double dist = origin_seg_dist2(x0, B);
    double dd = origin_seg_dist2(x0, C);
    if (dd < dist) dist = dd;
    dd = origin_seg_dist2(B, C);
    if (dd < dist) dist = dd;
    return dist;
}
// true + depth on intersection
template <bool MESH = false, int G = 1>
MOPA_HD bool mpr_penetration(const double *g1, int t1, const double *g2, int t2, double &depth, const double *aux = nullptr) {
    V3 v0 = sub3(ld3(g1 + GO_POS), ld3(g2 + GO_POS));
    if (vec_eq0(v0)) v0.x += kCcdEps * 10.0;
    V3 dir = normalize3(neg3(v0));
    V3 v1 = support_md<MESH, G>(g1, t1, g2, t2, dir, aux);
    double dot = dot3(v1, dir);
    if (is_zero(dot) || dot < 0.0) return false;
    dir = cross3(v0, v1);
    if (is_zero(dot3(dir, dir))) {
        if (vec_eq0(v1)) { depth = 0.0; return true; }
        depth = norm3(v1);
        return true;
    }
    dir = normalize3(dir);
    V3 v2 = support_md<MESH, G>(g1, t1, g2, t2, dir, aux);
    dot = dot3(v2, dir);
    if (is_zero(dot) || dot < 0.0) return false;
    dir = normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
    dot = dot3(dir, v0);
    if (dot > 0.0) {
        V3 tmp = v1; v1 = v2; v2 = tmp;
        dir = neg3(dir);
    }
    V3 v3, v4;
    int it = 0;
    for (;;) {
        if (++it > kMprPortalMaxIt) return false;
        v3 = support_md<MESH, G>(g1, t1, g2, t2, dir, aux);
        dot = dot3(v3, dir);
        if (is_zero(dot) || dot < 0.0) return false;
        bool cont = false;
        dot = dot3(cross3(v1, v3), v0);
        if (dot < 0.0 && !is_zero(dot)) { v2 = v3; cont = true; }
        if (!cont) {
            dot = dot3(cross3(v3, v2), v0);
            if (dot < 0.0 && !is_zero(dot)) { v1 = v3; cont = true; }
        }
        if (!cont) break;
        dir = normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
    }
    it = 0;
    for (;;) {
        if (++it > kMprPortalMaxIt) return false;
        dir = portal_dir(v1, v2, v3);
        dot = dot3(dir, v1);
        if (is_zero(dot) || dot > 0.0) break;
        v4 = support_md<MESH, G>(g1, t1, g2, t2, dir, aux);
        dot = dot3(v4, dir);
        if (!(is_zero(dot) || dot > 0.0) || portal_reach_tol(v1, v2, v3, v4, dir)) return false;
        expand_portal(v0, v1, v2, v3, v4);
    }
    int iterations = 0;
    for (;;) {
        dir = portal_dir(v1, v2, v3);
        v4 = support_md<MESH, G>(g1, t1, g2, t2, dir, aux);
        if (portal_reach_tol(v1, v2, v3, v4, dir) || iterations > kMprMaxIt) {
            depth = sqrt(origin_tri_dist2(v1, v2, v3));
            return true;
        }
        expand_portal(v0, v1, v2, v3, v4);
        iterations++;
    }
}
template <bool MESH = false, int G = 1>
MOPA_HD double d_convex(const double *A, int ta, const double *B, int tb, const double *aux = nullptr) {
    double depth;
    if (mpr_penetration<MESH, G>(A, ta, B, tb, depth, aux)) return -depth;
    return kFar;
}
// [3P] mjc_PlaneConvex for a mesh: the hull vertex deepest along -n (first one on ties)
template <int G = 1>
MOPA_HD double d_plane_mesh(const double *P, const double *M, const double *aux) {
    const V3 n = col3(P + GO_MAT, 2);
    const V3 ln = matT_vec(M + GO_MAT, n);
    const double *V = aux + (int)M[GO_SIZE];
    const int nv = (int)M[GO_SIZE + 1];
    int best = 0;
    double bd;
#if defined(__HIP_DEVICE_COMPILE__)
    if (G > 1) {
        const int sub = (int)(threadIdx.x & (G - 1));
        best = sub < nv ? sub : 0;
        bd = dot3(ln, ld3(V + 3 * best));
        for (int i = best + G; i < nv; i += G) {
            const double d = dot3(ln, ld3(V + 3 * i));
            if (d < bd) { bd = d; best = i; }
        }
        coop_fold<G, false>(bd, best);
    } else
#endif
    {
        bd = dot3(ln, ld3(V));
        for (int i = 1; i < nv; i++) {
            const double d = dot3(ln, ld3(V + 3 * i));
            if (d < bd) { bd = d; best = i; }
        }
    }
    const V3 w = add3(mat_vec(M + GO_MAT, ld3(V + 3 * best)), ld3(M + GO_POS));
    return dot3(sub3(w, ld3(P + GO_POS)), n);
}

MOPA_HD int pair_code(int t1, int t2) {
    if (t2 == G_MESH && (t1 == G_SPHERE || t1 == G_CAPSULE || t1 == G_CYLINDER || t1 == G_BOX)) return PC_CONVEX_MESH;
    if (t1 == G_PLANE) {
        if (t2 == G_SPHERE) return PC_PLANE_SPHERE;
        if (t2 == G_CAPSULE) return PC_PLANE_CAPSULE;
        if (t2 == G_CYLINDER) return PC_PLANE_CYLINDER;
        if (t2 == G_BOX) return PC_PLANE_BOX;
        if (t2 == G_MESH) return PC_PLANE_MESH;
        return -1;
    }
    if (t1 == G_SPHERE) {
        if (t2 == G_SPHERE) return PC_SPHERE_SPHERE;
        if (t2 == G_CAPSULE) return PC_SPHERE_CAPSULE;
        if (t2 == G_CYLINDER) return PC_SPHERE_CYLINDER;
        if (t2 == G_BOX) return PC_SPHERE_BOX;
        return -1;
    }
    if (t1 == G_CAPSULE) {
        if (t2 == G_CAPSULE) return PC_CAPSULE_CAPSULE;
        if (t2 == G_CYLINDER) return PC_CONVEX;
        if (t2 == G_BOX) return PC_CAPSULE_BOX;
        return -1;
    }
    if (t1 == G_CYLINDER) {
        if (t2 == G_CYLINDER || t2 == G_BOX) return PC_CONVEX;
        return -1;
    }
    if (t1 == G_BOX && t2 == G_BOX) return PC_BOX_BOX;
    return -1;
}

// Second pre-test before the portal refinement, for what the enclosing capsules let through: separating axes that are exact
// for these shapes -- the axis of shape 1 (capsule or cylinder) and the axis / the three face normals of shape 2 (cylinder /
// box).  Extent of a cylinder (r, h) with axis a along a unit direction n: h |a.n| + r sqrt(1 - (a.n)^2); of a capsule:
// h |a.n| + r; of a box along its own normal i: s_i, along n: sum |u_i.n| s_i.  The wrist and gripper discs (radius 5.5 /
// 3.5 cm, half height 2.5 / 1.5 cm) hover over the table inside their enclosing capsules -- which reach a radius, not a half
// height, beyond the flat faces -- for most of a Push / Assembly episode.  A gap of more than 1e-9 along any of the axes =>
// the shapes are disjoint => what MPR reports for disjoint shapes (same rule, same margin as the capsule pre-test; the
// oracle has the identical function and a switch that turns it off: tests/test_oracle_primitives.py).
MOPA_HD bool convex_axes_separate(const double *A, int ta, const double *B, int tb) {
    const V3 d = sub3(ld3(B + GO_POS), ld3(A + GO_POS));
    const V3 a = col3(A + GO_MAT, 2);
    const double ra = A[GO_SIZE], ha = A[GO_SIZE + 1];
    const bool cap = ta == G_CAPSULE;
    const double ea = cap ? ha + ra : ha;                 // shape 1 along its own axis
    const double da = fabs(dot3(d, a));
    if (tb == G_BOX) {
        const V3 s = ld3(B + GO_SIZE);
        const V3 u0 = col3(B + GO_MAT, 0), u1 = col3(B + GO_MAT, 1), u2 = col3(B + GO_MAT, 2);
        const double c0 = dot3(a, u0), c1 = dot3(a, u1), c2 = dot3(a, u2);
        if (da - ea - fma(fabs(c2), s.z, fma(fabs(c1), s.y, fabs(c0) * s.x)) > 1e-9) return true;
        const double n0 = fma(-c0, c0, 1.0), n1 = fma(-c1, c1, 1.0), n2 = fma(-c2, c2, 1.0);
        const double w0 = cap ? ra : ra * sqrt(n0 > 0.0 ? n0 : 0.0);
        const double w1 = cap ? ra : ra * sqrt(n1 > 0.0 ? n1 : 0.0);
        const double w2 = cap ? ra : ra * sqrt(n2 > 0.0 ? n2 : 0.0);
        if (fabs(dot3(d, u0)) - s.x - fma(ha, fabs(c0), w0) > 1e-9) return true;
        if (fabs(dot3(d, u1)) - s.y - fma(ha, fabs(c1), w1) > 1e-9) return true;
        if (fabs(dot3(d, u2)) - s.z - fma(ha, fabs(c2), w2) > 1e-9) return true;
        return false;
    }
    const V3 b = col3(B + GO_MAT, 2);                     // shape 2 is a cylinder
    const double rb = B[GO_SIZE], hb = B[GO_SIZE + 1];
    const double c = dot3(a, b), n = fma(-c, c, 1.0), sn = sqrt(n > 0.0 ? n : 0.0), ac = fabs(c);
    if (da - ea - fma(hb, ac, rb * sn) > 1e-9) return true;
    if (fabs(dot3(d, b)) - hb - fma(ha, ac, cap ? ra : ra * sn) > 1e-9) return true;
    return false;
}

// Verdict-only shortcut on the other side: the shapes overlap DEEPER than `delta`, so the portal refinement -- whose depth is
// never less than the true penetration depth minus its 1e-6 tolerance -- would report a distance <= -delta + 1e-6.  Shown by a
// witness: a point whose delta-ball lies inside both shapes, i.e. a common point of the two shapes shrunk by delta (cylinder
// (r, h) -> (r - delta, h - delta); capsule -> radius r - delta; box -> half extents s - delta).  Two candidates: the point of
// shape 2's core closest to shape 1's centre, and the point of shape 1's core closest to shape 2's centre.  The wrist disc
// pushed through a bin wall -- the planner's most frequent reason to enter the refinement -- is decided here.  Callers pass
// delta = max(-threshold, 0) + 1e-4 and use the result only for the verdict (the distance itself still needs the refinement).
MOPA_HD bool convex_overlap_deeper_than(const double *A, int ta, const double *B, int tb, double delta) {
    const V3 ca = ld3(A + GO_POS), cb = ld3(B + GO_POS);
    const V3 a = col3(A + GO_MAT, 2);
    const bool cap = ta == G_CAPSULE;
    const double Ra = A[GO_SIZE] - delta, Ha = cap ? A[GO_SIZE + 1] : A[GO_SIZE + 1] - delta;
    if (!(Ra > 0.0) || !(Ha > 0.0)) return false;
    const V3 dab = sub3(cb, ca);                          // centre of 2 seen from centre of 1
    // P2: the point of core 1 closest to centre 2 (for a capsule core: the point of its segment; radius left as slack)
    const double ta2 = dot3(dab, a);
    const double tc = ta2 > Ha ? Ha : (ta2 < -Ha ? -Ha : ta2);
    V3 w{fma(-ta2, a.x, dab.x), fma(-ta2, a.y, dab.y), fma(-ta2, a.z, dab.z)};      // radial part of dab
    const double rho2 = dot3(w, w);
    if (!cap && rho2 > Ra * Ra) { const double k = Ra / sqrt(rho2); w = V3{w.x * k, w.y * k, w.z * k}; }
    if (cap) w = V3{0.0, 0.0, 0.0};
    const V3 p2{fma(tc, a.x, w.x), fma(tc, a.y, w.y), fma(tc, a.z, w.z)};          // relative to ca
    if (tb == G_BOX) {
        const V3 s{B[GO_SIZE] - delta, B[GO_SIZE + 1] - delta, B[GO_SIZE + 2] - delta};
        if (!(s.x > 0.0) || !(s.y > 0.0) || !(s.z > 0.0)) return false;
        // is P2 in the shrunk box?
        const V3 l2 = matT_vec(B + GO_MAT, sub3(p2, dab));
        if (fabs(l2.x) <= s.x && fabs(l2.y) <= s.y && fabs(l2.z) <= s.z) return true;
        // P1: the point of the shrunk box closest to centre 1; is it in core 1?
        const V3 l1 = matT_vec(B + GO_MAT, V3{-dab.x, -dab.y, -dab.z});
        const V3 c1{l1.x > s.x ? s.x : (l1.x < -s.x ? -s.x : l1.x), l1.y > s.y ? s.y : (l1.y < -s.y ? -s.y : l1.y),
                    l1.z > s.z ? s.z : (l1.z < -s.z ? -s.z : l1.z)};
        const V3 p1 = add3(dab, mat_vec(B + GO_MAT, c1));                           // relative to ca
        const double t1 = dot3(p1, a);
        const double tq = cap ? (t1 > Ha ? Ha : (t1 < -Ha ? -Ha : t1)) : t1;
        const V3 r1{fma(-tq, a.x, p1.x), fma(-tq, a.y, p1.y), fma(-tq, a.z, p1.z)};
        return fabs(t1) <= (cap ? 1.0e30 : Ha) && dot3(r1, r1) <= Ra * Ra;
    }
    const V3 b = col3(B + GO_MAT, 2);                     // shape 2 is a cylinder
    const double Rb = B[GO_SIZE] - delta, Hb = B[GO_SIZE + 1] - delta;
    if (!(Rb > 0.0) || !(Hb > 0.0)) return false;
    {   // is P2 in core 2?
        const V3 q = sub3(p2, dab);
        const double tq = dot3(q, b);
        const V3 r{fma(-tq, b.x, q.x), fma(-tq, b.y, q.y), fma(-tq, b.z, q.z)};
        if (fabs(tq) <= Hb && dot3(r, r) <= Rb * Rb) return true;
    }
    // P1: the point of core 2 closest to centre 1; is it in core 1?
    const double tb1 = -dot3(dab, b);
    const double tbc = tb1 > Hb ? Hb : (tb1 < -Hb ? -Hb : tb1);
    V3 wb{fma(-tb1, b.x, -dab.x), fma(-tb1, b.y, -dab.y), fma(-tb1, b.z, -dab.z)};
    const double rb2 = dot3(wb, wb);
    if (rb2 > Rb * Rb) { const double k = Rb / sqrt(rb2); wb = V3{wb.x * k, wb.y * k, wb.z * k}; }
    const V3 p1{dab.x + fma(tbc, b.x, wb.x), dab.y + fma(tbc, b.y, wb.y), dab.z + fma(tbc, b.z, wb.z)};
    const double t1 = dot3(p1, a);
    const double tq = cap ? (t1 > Ha ? Ha : (t1 < -Ha ? -Ha : t1)) : t1;
    const V3 r1{fma(-tq, a.x, p1.x), fma(-tq, a.y, p1.y), fma(-tq, a.z, p1.z)};
    return fabs(t1) <= (cap ? 1.0e30 : Ha) && dot3(r1, r1) <= Ra * Ra;
}
MOPA_HD double convex_deep_delta(double thr) { return (thr < 0.0 ? -thr : 0.0) + 1e-4; }
// verdict of a PC_CONVEX pair (distance <= thr) without the distance: pre-tests on both sides, then the refinement
MOPA_HD bool convex_pair_bad(const double *A, int ta, const double *B, int tb, double thr) {
    const double pre = (tb == G_BOX) ? d_capsule_box(A, B) : d_capsule_capsule(A, B);
    if (pre > 1e-9) return false;
    if (convex_axes_separate(A, ta, B, tb)) return false;
    if (convex_overlap_deeper_than(A, ta, B, tb, convex_deep_delta(thr))) return true;
    return d_convex<false>(A, ta, B, tb) <= thr;
}

// `aux`: the scene's double blob (mesh hull vertices live in it); only the *_MESH codes read it, and only the
// MESH = true instantiation contains them: scenes without a collidable mesh keep running exactly the code (and
// register budget) they had before mesh support existed.
template <bool MESH = false, int G = 1>
MOPA_HD double geom_dist(int code, const double *A, int ta, const double *B, int tb, const double *aux = nullptr) {
    switch (code) {
        case PC_PLANE_SPHERE: return d_plane_sphere(A, B);
        case PC_PLANE_CAPSULE: return d_plane_capsule(A, B);
        case PC_PLANE_CYLINDER: return d_plane_cylinder(A, B);
        case PC_PLANE_BOX: return d_plane_box(A, B);
        case PC_SPHERE_SPHERE: return d_sphere_sphere(A, B);
        case PC_SPHERE_CAPSULE: return d_sphere_capsule(A, B);
        case PC_SPHERE_CYLINDER: return d_sphere_cylinder(A, B);
        case PC_SPHERE_BOX: return d_sphere_box(A, B);
        case PC_CAPSULE_CAPSULE: return d_capsule_capsule(A, B);
        case PC_CAPSULE_BOX: return d_capsule_box(A, B);
        case PC_BOX_BOX: return d_box_box(A, B);
        case PC_CONVEX: {
            // Pre-test before the portal refinement (by far the most expensive narrow-phase routine, and ~90 % of
            // the cylinder pairs the broad phase lets through are in fact disjoint): a cylinder lies inside the capsule
            // of the same axis, radius and half length, and capsule-capsule / capsule-box distances are closed form.
            // Enclosures apart by more than 1e-9  =>  the shapes are disjoint  =>  what MPR would report.
            const double pre = (tb == G_BOX) ? d_capsule_box(A, B) : d_capsule_capsule(A, B);
            if (pre > 1e-9) return kFar;
            if (convex_axes_separate(A, ta, B, tb)) return kFar;
            return d_convex<false>(A, ta, B, tb);
        }
        case PC_PLANE_MESH: return MESH ? d_plane_mesh<G>(A, B, aux) : kFar;
        case PC_CONVEX_MESH: return MESH ? d_convex<MESH, G>(A, ta, B, tb, aux) : kFar;
        default: return kFar;
    }
}

// [3P] broad phase: bounding spheres, zero margin (contact_threshold < 0)
MOPA_HD bool bp_cull(const double *A, int ta, double rba, const double *B, double rbb) {
    V3 diff = sub3(ld3(B + GO_POS), ld3(A + GO_POS));
    if (ta == G_PLANE) return dot3(diff, col3(A + GO_MAT, 2)) > rbb;
    double rs = rba + rbb;
    return dot3(diff, diff) > rs * rs;
}

// Second-stage cull for a pair of one STATIC and one moving geom: the static geom's world AABB (centre = its pos,
// half extents H, computed once per scene by static_aabb_half) against the moving geom's bounding sphere.  Tables,
// bin walls and obstacle posts are long thin boxes whose bounding spheres cover half the workspace; this removes
// ~40-50 % of the narrow-phase work the sphere test lets through (and the expensive box pairs first).
MOPA_HD bool aabb_cull(V3 cS, const double *H, V3 cM, double rM) {
    return fabs(cM.x - cS.x) > H[0] + rM || fabs(cM.y - cS.y) > H[1] + rM || fabs(cM.z - cS.z) > H[2] + rM;
}
// world-AABB half extents of a posed primitive (conservative; + 1e-9 so that rounding can never cut into the shape)
MOPA_HD void static_aabb_half(int type, const double *rec, double rbound, double *H) {
    const double *M = rec + GO_MAT, *sz = rec + GO_SIZE;
    for (int i = 0; i < 3; i++) {
        const double a0 = fabs(M[3 * i]), a1 = fabs(M[3 * i + 1]), a2 = fabs(M[3 * i + 2]);
        double hv;
        if (type == G_BOX) hv = fma(a2, sz[2], fma(a1, sz[1], a0 * sz[0]));
        else if (type == G_SPHERE) hv = sz[0];
        else if (type == G_CAPSULE) hv = fma(a2, sz[1], sz[0]);
        else if (type == G_CYLINDER) { const double s2 = fma(-a2, a2, 1.0); hv = fma(a2, sz[1], sz[0] * ((s2 > 0.0) ? sqrt(s2) : 0.0)); }
        else hv = rbound;   // mesh / anything else: the bounding sphere
        H[i] = hv + 1e-9;
    }
}

// counter-based RNG: splitmix64 finaliser over (seed, stream, counter)
MOPA_HD uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
// (the stream's key once, then a value per counter: the same two steps as rng_uniform)
MOPA_HD uint64_t rng_key(uint64_t seed, uint64_t stream) { return mix64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1)); }
MOPA_HD double rng_uniform_k(uint64_t key, uint64_t counter) {
    const uint64_t r = mix64(key + 0x9E3779B97F4A7C15ULL * (counter + 1));
    return (double)(r >> 11) * 0x1.0p-53;
}
MOPA_HD double rng_uniform(uint64_t seed, uint64_t stream, uint64_t counter) {
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1));
    uint64_t r = mix64(k + 0x9E3779B97F4A7C15ULL * (counter + 1));
    return (double)(r >> 11) * 0x1.0p-53;
}

}  // namespace mopa
