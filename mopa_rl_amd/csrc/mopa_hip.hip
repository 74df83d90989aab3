// mopa_hip.hip -- libmopa_hip.so: scene compilation, HIP kernels and the C ABI
// declared in include/mopa_hip.h.  Target: gfx950 (MI355X), wave64.
//
// Kernels:
//   K1 state validity, two generations
//      k_is_valid_v2 (mopa_valid_v2.inc, production for large batches): one LANE per state, 64-state tiles,
//        FK per lane, cull while the pose is in registers, wave-wide narrow phase from an LDS ring queue;
//      k_is_valid (this file): one wave64 per state -- small batches, single-state API calls and the device
//        routine the planner kernels call.  Scene constants (~10 KB) are staged once per workgroup in LDS.
//   K2 k_check_motion  one wave per segment (OMPL DiscreteMotionValidator semantics)
//   K3 k_rrt_connect   (mopa_planner.inc) one wave per env
//   K4 k_env_step      (mopa_env.inc) one lane per env, kinematic env.step
//   FP64 VALU bound, no MFMA (there is no dense contraction on this path).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <map>
#include <mutex>
#include "../../include/mopa_hip.h"
#include "mopa_device.hpp"
#include "mopa_host.hpp"

using namespace mopa;

// ---------------------------------------------------------------------------
// error plumbing (mopa_host.hpp: fail / HIP_TRY / DeviceGuard / ON_DEVICE / DevBuf, shared with mopa_envdyn.hip)
// ---------------------------------------------------------------------------
static thread_local std::string g_err;
int mopa_fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
constexpr double kV5MaxReach = 32.0;        // metres: beyond this the FP32 broad phase is not used (see mopa_scene_create)
static void plan_register_lds();            // defined with K3 (mopa_planner.inc)

extern "C" const char *mopa_last_error(void) { return g_err.c_str(); }
extern "C" const char *mopa_version(void) { return "mopa_hip 0.1.0 (gfx950)"; }
extern "C" int mopa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------
// device scene: a header of counts/offsets (kernel argument, lives in SGPRs)
// plus one blob of doubles and one of ints that every workgroup copies to LDS.
// ---------------------------------------------------------------------------
struct SceneHdr {
    int na, nq, n_pq, nmb, nmj, nsf, ng, nmg, npair;
    int n_dbl, n_int;            // blob sizes
    // double-blob offsets
    int o_mb_pos, o_mb_quat, o_sf_pos, o_sf_quat, o_sf_mat, o_mj_axis, o_mj_pos, o_mj_ref;
    int o_g_lpos, o_g_lquat, o_g_rbound, o_g_rec, o_act_lo, o_act_hi, o_act_ext;
    int o_act_ref, o_act_hinge;   // per active joint value slot: the joint's reference value (doubles), 1 = hinge (ints)
    int o_g_aabb;   // [ng][3] world-AABB half extents of static geoms (second-stage cull), zeros for moving geoms
    // int-blob offsets
    int o_mb_parent, o_mb_jntadr, o_mb_jntnum, o_mj_type, o_mj_qsrc, o_g_type, o_g_slot, o_g_mb;
    int o_mg_geom, o_chain_adr, o_chain_len, o_chain_items, o_pairs, o_pq_adr, o_act_adr, o_act_so2;
    // second-generation validity kernel (mopa_valid_v2.inc): DFS program + per-geom pair lists
    int n_save, n_gp;
    int v5_ent_cap;   // k_is_valid_v5: survivor-entry buffer capacity per wave
    int o_mesh, has_mesh;   // mesh hull vertices (doubles); has_mesh selects the MESH kernel instantiations
    int o_mb_load, o_mb_save, o_mb_mgadr, o_mb_mgnum, o_mg_padr, o_mg_pnum, o_mg_store, o_gp_word;
    int o_mbr, o_mbd, o_mgr, o_mgd;   // packed per-body / per-geom records (ints: 8 / 4, doubles: 16 / 8)
    // planner FK (mopa_planner.inc: ms_fk): per moving geom slot 8 ints -- [0..3] the chain's bodies, one per byte (0xff past the
    // end), [4] chain length, [5] static frame of the chain root's parent; pfk_maxlen = longest chain (0: a chain is longer than
    // 16 bodies -> the planner keeps the generic walk)
    int o_pfk, pfk_maxlen;
    // k_is_valid_v5, tiles whose 64 states share one env row: the moving bodies no ACTIVE coordinate reaches (a manipulated object's free
    // body and what is welded to it: Assembly's furniture = 21 bodies, 20 geoms) are posed ONCE per tile, one body per lane, level by level,
    // instead of by every lane in its walk.  o_pas_b: the bodies (moving-body ids) in level order, o_pas_lv [n_pas_lv + 1]: level starts,
    // o_mb_pas [nmb]: position in that list or -1 (bit 16: nothing continues from this body's registers), o_pas_g: the moving-geom slots
    // on them, o_mg_pas [nmg]: position or -1.  n_pas_b = 0: off.
    int n_pas_b, n_pas_g, n_pas_lv, o_pas_b, o_pas_lv, o_mb_pas, o_pas_g, o_mg_pas;
    // per-wave LDS slab (in doubles): geom records, qbuf; then worklist (u16)
    int wave_dbl, wave_bytes;
    double thr, range, resolution;
    double nn_eps;   // planner: bound on |FP32 mirror distance - FP64 distance| of the nearest-neighbour sweep (mopa_planner.inc)
};

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

struct StreamScratch {
    DevBuf slab;        // lane-per-state kernels: pose slabs of the launch's waves + [profile words | tile counter]
    DevBuf mpr;         // v5: per-wave ring of deferred cylinder pairs
    DevBuf cen;         // v5, scenes whose FP32 centre table does not fit LDS: the per-wave tables in global memory
    DevBuf k1_ctr;      // validity kernels' tile counter + exit count (self-resetting: tile_ctr_release)
    DevBuf mesh_list;   // [0] = number of rows in mesh_rows, [1] = count, then the states with a mesh pair past the main pass's broad phase
    DevBuf mesh_rows;   // complete records of the mesh pairs within reach ([cap][kMprRow] doubles; k_mesh_rows)
    size_t slab_waves = 0;
    DevBuf mv_cnt, mv_off, mv_env, mv_q, mv_valid, mv_scan;   // expanded motion validation (mopa_motion.inc)
    DevBuf plan_q, plan_p, plan_ctr;                          // planner: both trees of every env, env counter (mopa_planner.inc)
    DevBuf ip_walk;                                           // straight-line pre-check: walk states + verdicts (mopa_paths.inc)
    DevBuf pb_small, pb_rows, pb_act;                         // batched pull-back: verdicts / slots, candidate rows, their active coordinates + verdicts
};

struct MopaScene {
    int device = 0;
    SceneHdr hdr{};
    SceneHdr hdr_mesh{};      // same scene, per-geom pair lists = the mesh pairs only (second pass of the lane-per-state kernels)
    int n_mesh_dbl = 0;       // doubles of the hull-vertex block at hdr.o_mesh (k_mesh_rows stages it in LDS)
    int n_mesh_gp = 0;
    std::vector<double> h_dbl;
    std::vector<int32_t> h_int;
    std::vector<int32_t> h_gp_tab;
    double *d_dbl = nullptr;
    int32_t *d_int = nullptr;
    int lds_bytes = 0;
    // host copies for the single-query forms
    int nq = 0, na = 0, ngeom_model = 0, npair_model = 0;
    std::vector<int32_t> active_idx;
    std::vector<int32_t> pair_slot;   // model pair index -> device pair index or -1
    std::vector<int32_t> geom_model_of_dev;  // (identity; device geoms == model collidable geoms)
    uint64_t seed = 0;
    std::string status = "none";
    // scratch for single-query calls
    double *d_q = nullptr;      // nq + na + big scratch
    uint8_t *d_valid = nullptr;
    double *d_md = nullptr;
    double *d_dbg = nullptr;
    size_t dbg_doubles = 0;
    int n_cu = 256;
    int v2_lds_bytes = 0;
    int use_v2 = 1;
    int32_t *d_gp_tab = nullptr;   // v5: FP32 broad-phase table [n_gp][8]
    int v5_lds_bytes = 0;        // verdict-only instantiations of k_is_valid_v5 (hdr.v5_ent_cap entries per wave)
    int v5_lds_bytes_md = 0, v5_ent_cap_md = 0;   // depth-reporting instantiations
    int use_v5 = 0;
    bool v5_cen_lds = true;   // FP32 centre table of a tile in LDS (false: read back from the pose slab; scenes with many moving geoms)
    bool v2_forced = false;   // MOPA_VALID_KERNEL=v2: lane-per-state kernel for every N >= 64 (tests, A/B runs)
    // Launch scratch, one set PER STREAM: a scene may be driven from several streams at once (validity on one stream while
    // the planner or the previous step's motion check runs on another); calls on the same stream are ordered by the
    // stream.  Buffers only ever grow; an outgrown buffer may still be read by kernels in flight, so it is retired and
    // freed with the scene, never on the hot path.
    std::mutex mu;
    std::map<hipStream_t, StreamScratch> scratch;
    std::vector<void *> retired;
};

static StreamScratch &scratch_for(MopaScene *S, hipStream_t st) {
    std::lock_guard<std::mutex> lock(S->mu);
    return S->scratch[st];       // std::map: references stay valid across later insertions
}
// make `b` hold at least `bytes`; returns a HIP error code
static hipError_t grow(MopaScene *S, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return hipSuccess;
    const size_t want = std::max(bytes, b.cap + b.cap / 2);
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) return e;
    if (b.p) {
        std::lock_guard<std::mutex> lock(S->mu);
        S->retired.push_back(b.p);
    }
    b.p = np;
    b.cap = want;
    return hipSuccess;
}

// Zero a few 8-byte words on a stream.  A kernel, not hipMemsetAsync: these launches are also captured into HIP graphs
// (rollout.py, cfg.use_graphs), and small memset nodes proved unreliable there.
__global__ void k_zero_words(unsigned long long *p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0ull;
}
static hipError_t zero_async(void *p, size_t bytes, hipStream_t st) {
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned long long *>(p), (int)((bytes + 7) / 8));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
MOPA_D void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
MOPA_D bool wave_any(bool p) { return __ballot(p) != 0ull; }
MOPA_D double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = (o < v) ? o : v;
    }
    return v;
}

// A persistent kernel's work counter that needs no zeroing launch: ctr[0] hands out the work items, ctr[1] counts the waves that have
// found none left; the last of the grid's waves to say so puts both back to zero for the next launch on the stream.
MOPA_D void tile_ctr_release(unsigned long long *ctr, int lane) {
    if (lane == 0) {
        const unsigned long long done = atomicAdd(ctr + 1, 1ull);
        if (done + 1ull == (unsigned long long)gridDim.x * (blockDim.x >> 6)) {
            atomicExch(ctr, 0ull);
            atomicExch(ctr + 1, 0ull);
        }
    }
}

struct LdsView {
    const double *dbl;   // shared scene doubles
    const int *ints;     // shared scene ints
    double *grec;        // per-wave posed records of moving geoms [nmg*kGeomStride]
    double *qbuf;        // per-wave joint values: [na active][n_pq passive]
    double *sc;          // per-wave [nmj][2]: sin, cos of half the angle of every moving hinge joint of the state in qbuf
    unsigned short *wl;  // per-wave worklist [npair]
};

MOPA_D void stage_scene(const SceneHdr &h, const double *g_dbl, const int32_t *g_int, double *s_dbl, int *s_int) {
    for (int i = threadIdx.x; i < h.n_dbl; i += blockDim.x) s_dbl[i] = g_dbl[i];
    for (int i = threadIdx.x; i < h.n_int; i += blockDim.x) s_int[i] = g_int[i];
    __syncthreads();
}

MOPA_D LdsView make_view(const SceneHdr &h, unsigned char *smem) {
    LdsView v;
    double *s_dbl = reinterpret_cast<double *>(smem);
    int *s_int = reinterpret_cast<int *>(s_dbl + h.n_dbl);
    int int_pad = (h.n_int + 1) & ~1;
    // (readfirstlane: the wave index is wave-uniform, but derived from threadIdx the compiler keeps it -- and every pointer
    //  computed from it -- in vector registers: two VGPRs per pointer held across the planner's non-inlined validity calls)
    unsigned char *wave_base = reinterpret_cast<unsigned char *>(s_int + int_pad) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * h.wave_bytes;
    v.dbl = s_dbl;
    v.ints = s_int;
    v.grec = reinterpret_cast<double *>(wave_base);
    v.qbuf = v.grec + h.nmg * kGeomStride;
    v.sc = v.qbuf + h.na + h.n_pq + h.na;
    v.wl = reinterpret_cast<unsigned short *>(v.sc + 2 * h.nmj);
    return v;
}

// posed record of geom g: moving geoms live in the wave slab, static ones in the shared blob
MOPA_D const double *geom_rec(const SceneHdr &h, const LdsView &v, int g) {
    int slot = v.ints[h.o_g_slot + g];
    return (slot >= 0) ? (v.grec + slot * kGeomStride) : (v.dbl + h.o_g_rec + g * kGeomStride);
}

// broad phase of one candidate pair: bounding spheres (plane: signed distance), then -- exactly one geom static --
// the static geom's world AABB against the moving geom's bounding sphere
MOPA_D bool pair_culled(const SceneHdr &h, const LdsView &v, int g1, int g2, const double *A, const double *B) {
    const int *I = v.ints;
    const double *D = v.dbl;
    const int t1 = I[h.o_g_type + g1];
    if (bp_cull(A, t1, D[h.o_g_rbound + g1], B, D[h.o_g_rbound + g2])) return true;
    if (t1 == G_PLANE) return false;
    const bool s1 = I[h.o_g_slot + g1] < 0, s2 = I[h.o_g_slot + g2] < 0;
    if (s1 == s2) return false;
    const int gs = s1 ? g1 : g2, gm = s1 ? g2 : g1;
    return aabb_cull(ld3((s1 ? A : B) + GO_POS), D + h.o_g_aabb + 3 * gs, ld3((s1 ? B : A) + GO_POS), D[h.o_g_rbound + gm]);
}

// Forward kinematics for the state whose joint values are in v.qbuf.
// Lane l < nmg walks the ancestor chain of moving geom l (same operation order
// as a parent-first sweep over the body tree) and writes the posed geom.
// fk_one_geom: the per-lane work for moving geom `lane` of the state in v.qbuf -> v.grec.
MOPA_D void fk_one_geom(const SceneHdr &h, const LdsView &v, int lane) {
    // Reads the PACKED per-body records (ints o_mbr: jn, jntadr, -, static-frame id, -, -, -, joint-0 word; doubles o_mbd:
    // pos[3] quat[4] axis[3] jpos[3] ref) -- two levels of dependent LDS reads per body (chain item -> record -> joint value)
    // where the separate tables needed six; the arithmetic and its order are unchanged.
    const double *D = v.dbl;
    const int *I = v.ints;
    const int g = I[h.o_mg_geom + lane];
    const int b = I[h.o_g_mb + g];
    const int cadr = I[h.o_chain_adr + b], clen = I[h.o_chain_len + b];
    V3 pos{0.0, 0.0, 0.0};
    Q4 quat{1.0, 0.0, 0.0, 0.0};
    double mat[9];
    for (int k = 0; k < clen; k++) {
        const int body = I[h.o_chain_items + cadr + k];
        const int *r = I + h.o_mbr + 8 * body;
        const int jn = r[0], ja = r[1], w7 = r[7];
        const double *bd = D + h.o_mbd + 16 * body;
        const int jt0 = w7 & 0x7f, qsrc0 = w7 >> 8;
        if (jn == 1 && jt0 == J_FREE) {
            const double *qp = v.qbuf + qsrc0;
            pos = V3{qp[0], qp[1], qp[2]};
            quat = quat_normalize(Q4{qp[3], qp[4], qp[5], qp[6]});
        } else {
            V3 ppos;
            Q4 pquat;
            if (k == 0) {
                const int sf = r[3];
                ppos = ld3(D + h.o_sf_pos + 3 * sf);
                const double *sq = D + h.o_sf_quat + 4 * sf;
                pquat = Q4{sq[0], sq[1], sq[2], sq[3]};
                const double *sm = D + h.o_sf_mat + 9 * sf;
#pragma unroll
                for (int i = 0; i < 9; i++) mat[i] = sm[i];
            } else {
                ppos = pos;
                pquat = quat;
            }
            pos = add3(ppos, mat_vec(mat, ld3(bd)));
            quat = quat_mul(pquat, Q4{bd[3], bd[4], bd[5], bd[6]});
            if (jn > 0) {
                const double dq = v.qbuf[qsrc0] - bd[13];
                apply_joint_sc(jt0, ld3(bd + 7), ld3(bd + 10), (w7 & 0x80) != 0, dq, v.sc[2 * ja], v.sc[2 * ja + 1], pos, quat);
            }
            for (int j = ja + 1; j < ja + jn; j++) {       // bodies with more than one joint: the separate joint tables
                V3 ax = ld3(D + h.o_mj_axis + 3 * j), jp = ld3(D + h.o_mj_pos + 3 * j);
                double dq = v.qbuf[I[h.o_mj_qsrc + j]] - D[h.o_mj_ref + j];
                apply_joint_sc(I[h.o_mj_type + j], ax, jp, is_zero3(jp), dq, v.sc[2 * j], v.sc[2 * j + 1], pos, quat);
            }
            quat = quat_normalize(quat);
        }
        quat2mat(mat, quat);
    }
    double *rec = v.grec + lane * kGeomStride;
    const double *gd = D + h.o_mgd + 8 * lane;               // lpos[3] lquat[4] rbound of moving geom `lane`
    V3 gp = add3(pos, mat_vec(mat, ld3(gd)));
    Q4 gq = quat_mul(quat, Q4{gd[3], gd[4], gd[5], gd[6]});
    double gm[9];
    quat2mat(gm, gq);
    st3(rec + GO_POS, gp);
#pragma unroll
    for (int i = 0; i < 9; i++) rec[GO_MAT + i] = gm[i];
    // size is constant: copied from the shared record
    const double *srec = D + h.o_g_rec + g * kGeomStride;
    rec[GO_SIZE] = srec[GO_SIZE]; rec[GO_SIZE + 1] = srec[GO_SIZE + 1]; rec[GO_SIZE + 2] = srec[GO_SIZE + 2];
}
MOPA_D void wave_fk(const SceneHdr &h, const LdsView &v, int lane) {
    if (lane < h.nmg) fk_one_geom(h, v, lane);
    wave_sync();
}

// Collision sweep for the posed state.  Returns the wave-uniform verdict.
// WANT_MD: also produce the minimum distance over broad-phase survivors
// (disables the early-out so the minimum is complete).
template <bool WANT_MD, bool MESH = false>
MOPA_D bool wave_collide(const SceneHdr &h, const LdsView &v, int lane, double &min_dist) {
    const int *I = v.ints;
    // 1. broad phase + ballot compaction (order-preserving => worklist stays sorted by pair type)
    int wl_count = 0;
    for (int base = 0; base < h.npair; base += 64) {
        int p = base + lane;
        bool surv = false;
        if (p < h.npair) {
            int pk = I[h.o_pairs + p];
            int g1 = pk & 0xff, g2 = (pk >> 8) & 0xff;
            const double *A = geom_rec(h, v, g1), *B = geom_rec(h, v, g2);
            surv = !pair_culled(h, v, g1, g2, A, B);
        }
        unsigned long long mask = __ballot(surv);
        if (surv) {
            int idx = wl_count + __popcll(mask & ((1ull << lane) - 1ull));
            v.wl[idx] = (unsigned short)p;
        }
        wl_count += __popcll(mask);
    }
    wave_sync();
    // 2. narrow phase over the survivors
    bool bad = false;
    double md = 0.0;   // deepest penetration: min(0, min over pairs)
    for (int base = 0; base < wl_count; base += 64) {
        int i = base + lane;
        if (i < wl_count) {
            int pk = I[h.o_pairs + v.wl[i]];
            int g1 = pk & 0xff, g2 = (pk >> 8) & 0xff, code = (pk >> 16) & 0xff;
            const double *A = geom_rec(h, v, g1), *B = geom_rec(h, v, g2);
            if (!WANT_MD && code == PC_CONVEX) {       // verdict only: the deep-overlap shortcut may stand in for the refinement
                if (convex_pair_bad(A, I[h.o_g_type + g1], B, I[h.o_g_type + g2], h.thr)) bad = true;
            } else {
                double d = geom_dist<MESH>(code, A, I[h.o_g_type + g1], B, I[h.o_g_type + g2], v.dbl);
                if (d < md) md = d;
                if (d <= h.thr) bad = true;
            }
        }
        if (!WANT_MD && wave_any(bad)) break;
    }
    bool any_bad = wave_any(bad);
    if (WANT_MD) min_dist = wave_min(md);
    wave_sync();   // worklist / geom slab are about to be reused
    return !any_bad;
}

// sin/cos of half the joint angle for every moving hinge joint of the state in v.qbuf, one joint per lane: taken out of
// the serial walk down the kinematic chain (it is ~half of a body's critical path there).  Same arithmetic.
MOPA_D void wave_sincos_table(const SceneHdr &h, const LdsView &v, int lane) {
    for (int j = lane; j < h.nmj; j += 64) {
        double sn = 0.0, cs = 1.0;
        if (v.ints[h.o_mj_type + j] == J_HINGE) mopa_sincos(0.5 * (v.qbuf[v.ints[h.o_mj_qsrc + j]] - v.dbl[h.o_mj_ref + j]), sn, cs);
        v.sc[2 * j] = sn;
        v.sc[2 * j + 1] = cs;
    }
    wave_sync();
}

// fill v.qbuf for (env row, active vector)
MOPA_D void wave_load_state(const SceneHdr &h, const LdsView &v, int lane, const double *q_active, const double *qpos_row) {
    if (lane < h.na) v.qbuf[lane] = q_active[lane];
    else if (lane < h.na + h.n_pq) v.qbuf[lane] = qpos_row[v.ints[h.o_pq_adr + lane - h.na]];
    // models with na + n_pq > 64 loop
    for (int i = lane + 64; i < h.na + h.n_pq; i += 64)
        v.qbuf[i] = (i < h.na) ? q_active[i] : qpos_row[v.ints[h.o_pq_adr + i - h.na]];
    wave_sync();
    wave_sincos_table(h, v, lane);
}

// ---------------------------------------------------------------------------
// K1: state validity, one wave per state
// ---------------------------------------------------------------------------
template <bool WANT_MD, bool MESH>
__global__ __launch_bounds__(kBlock) void k_is_valid(SceneHdr h, const double *__restrict__ g_dbl, const int32_t *__restrict__ g_int,
                                                      const double *__restrict__ q_active, const double *__restrict__ qpos_env,
                                                      long long N, long long samples_per_env, unsigned char *__restrict__ valid,
                                                      double *__restrict__ min_dist, const int *__restrict__ env_idx /* nullable: env row of every state */,
                                                      const long long *__restrict__ n_dev /* nullable: the state count lives on the device (<= N) */,
                                                      long long n_small /* with n_dev: this launch serves counts below it only (the lane-per-state launch behind it the others) */) {
    if (n_dev) {
        const long long nd = *n_dev;
        if (nd >= n_small) return;
        if (nd < N) N = nd;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsView v = make_view(h, smem);
    stage_scene(h, g_dbl, g_int, const_cast<double *>(v.dbl), const_cast<int *>(v.ints));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long stride = (long long)gridDim.x * kWavesPerBlock;
    for (long long s = (long long)blockIdx.x * kWavesPerBlock + wave; s < N; s += stride) {
        long long env = env_idx ? (long long)env_idx[s] : s / samples_per_env;
        wave_load_state(h, v, lane, q_active + s * h.na, qpos_env + env * h.nq);
        wave_fk(h, v, lane);
        double md;
        bool ok = wave_collide<WANT_MD, MESH>(h, v, lane, md);
        if (lane == 0) {
            valid[s] = ok ? 1 : 0;
            if (WANT_MD) min_dist[s] = md;
        }
    }
}

// ---------------------------------------------------------------------------
// K2: discrete motion validation, one wave per segment
// ---------------------------------------------------------------------------
// --- OMPL state-space helpers, per 1-D subspace (RealVectorStateSpace(1) / SO2StateSpace) ---
MOPA_D double dist_dim(const SceneHdr &h, const LdsView &v, int a, double x, double y) {
    double d = fabs(x - y);
    if (v.ints[h.o_act_so2 + a] && d > kPi) d = 2.0 * kPi - d;
    return d;
}
MOPA_D double interp_dim(const SceneHdr &h, const LdsView &v, int a, double from, double to, double t) {
    double diff = to - from;
    if (!v.ints[h.o_act_so2 + a] || fabs(diff) <= kPi) return fma(diff, t, from);
    if (diff > 0.0) diff = 2.0 * kPi - diff; else diff = -2.0 * kPi - diff;
    double r = fma(-diff, t, from);
    if (r > kPi) r -= 2.0 * kPi; else if (r < -kPi) r += 2.0 * kPi;
    return r;
}
// CompoundStateSpace::validSegmentCount = max over subspaces of ceil(d_i / (resolution * extent_i))
MOPA_D int valid_segment_count(const SceneHdr &h, const LdsView &v, const double *qa, const double *qb) {
    int nd = 0;
    for (int a = 0; a < h.na; a++) {
        double seg = h.resolution * v.dbl[h.o_act_ext + a];
        int c = (int)ceil(dist_dim(h, v, a, qa[a], qb[a]) / seg);
        if (c > nd) nd = c;
    }
    return nd;
}

__device__ __noinline__ bool plan_state_valid_impl(const SceneHdr *hp, const double *dbl, const int *ints, double *grec,
                                                   double *qbuf, unsigned short *wl, int lane, const double *qa, const double *row);

// (two waves per SIMD: the non-inlined validity routines then save ~40 registers to scratch around every call, and it still pays:
//  wave-per-env pull-back 1.00 -> 0.81 ms per 3000 targets, wave-per-segment motion checks 46 -> 83 M motions/s)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_check_motion(SceneHdr h, const double *__restrict__ g_dbl, const int32_t *__restrict__ g_int,
                                                         const double *__restrict__ qa_all, const double *__restrict__ qb_all,
                                                         const double *__restrict__ qpos_env, long long N, long long samples_per_env,
                                                         unsigned char *__restrict__ valid, int hdr_lds_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsView v = make_view(h, smem);
    // the non-inlined validity routine reads the header through an LDS pointer (a pointer to the by-value kernel
    // argument would live in scratch memory)
    SceneHdr *lh = reinterpret_cast<SceneHdr *>(smem + hdr_lds_off);
    for (int i = threadIdx.x; i < (int)(sizeof(SceneHdr) / 4); i += blockDim.x)
        reinterpret_cast<int *>(lh)[i] = reinterpret_cast<const int *>(&h)[i];
    stage_scene(h, g_dbl, g_int, const_cast<double *>(v.dbl), const_cast<int *>(v.ints));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long stride = (long long)gridDim.x * kWavesPerBlock;
    for (long long s = (long long)blockIdx.x * kWavesPerBlock + wave; s < N; s += stride) {
        const double *qa = qa_all + s * h.na, *qb = qb_all + s * h.na;
        const double *row = qpos_env + (s / samples_per_env) * h.nq;
        int nd = valid_segment_count(h, v, qa, qb);
        bool ok = true;
        // k == nd is the end state (tested first, as OMPL does; also the only test when qa == qb); interior
        // states in index order -- the verdict is order independent.  One call site -> one copy of FK+collision.
        double *tst = v.qbuf + h.na + h.n_pq;   // spare [na] doubles behind the joint-value buffer
        for (int k = nd; k >= (nd > 0 ? 1 : 0) && ok; k--) {
            const double t = (nd > 0) ? (double)k / (double)nd : 1.0;
            if (lane < h.na) tst[lane] = (k == nd) ? qb[lane] : interp_dim(h, v, lane, qa[lane], qb[lane], t);
            for (int i = lane + 64; i < h.na; i += 64) tst[i] = (k == nd) ? qb[i] : interp_dim(h, v, i, qa[i], qb[i], t);
            wave_sync();
            ok = plan_state_valid_impl(lh, v.dbl, v.ints, v.grec, v.qbuf, v.wl, lane, tst, row);
        }
        if (lane == 0) valid[s] = ok ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// debug kernels (parity hooks): posed geoms and per-pair distances of one state
// ---------------------------------------------------------------------------
template <bool MESH>
__global__ __launch_bounds__(kBlock) void k_debug_state(SceneHdr h, const double *__restrict__ g_dbl, const int32_t *__restrict__ g_int,
                                                        const double *__restrict__ q_active, const double *__restrict__ qpos_row,
                                                        double *__restrict__ out_rec /*[ng*16]*/, double *__restrict__ out_dist /*[npair]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsView v = make_view(h, smem);
    stage_scene(h, g_dbl, g_int, const_cast<double *>(v.dbl), const_cast<int *>(v.ints));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave != 0) return;
    wave_load_state(h, v, lane, q_active, qpos_row);
    wave_fk(h, v, lane);
    for (int g = lane; g < h.ng; g += 64) {
        const double *r = geom_rec(h, v, g);
        for (int i = 0; i < kGeomStride; i++) out_rec[g * kGeomStride + i] = (i < 15) ? r[i] : 0.0;
    }
    for (int p = lane; p < h.npair; p += 64) {
        int pk = v.ints[h.o_pairs + p];
        int g1 = pk & 0xff, g2 = (pk >> 8) & 0xff, code = (pk >> 16) & 0xff;
        const double *A = geom_rec(h, v, g1), *B = geom_rec(h, v, g2);
        double d = kFar;
        if (!pair_culled(h, v, g1, g2, A, B))
            d = geom_dist<MESH>(code, A, v.ints[h.o_g_type + g1], B, v.ints[h.o_g_type + g2], v.dbl);
        out_dist[p] = d;
    }
}

#include "mopa_valid_v2.inc"
#include "mopa_valid_v5.inc"

// ---------------------------------------------------------------------------
// host: scene compilation
// ---------------------------------------------------------------------------
namespace {

struct Builder {
    std::vector<double> dbl;
    std::vector<int32_t> ints;
    int add_d(const std::vector<double> &v) { int o = (int)dbl.size(); dbl.insert(dbl.end(), v.begin(), v.end()); return o; }
    int add_i(const std::vector<int32_t> &v) { int o = (int)ints.size(); ints.insert(ints.end(), v.begin(), v.end()); return o; }
};

}  // namespace

extern "C" int mopa_scene_create(const MopaSceneDesc *desc, MopaScene **out) {
    if (!desc || !out) return fail(MOPA_ERR_INVALID_ARG, "null argument");
    const MopaModel &m = desc->model;
    if (m.nq <= 0 || m.nbody <= 0 || m.ngeom < 0) return fail(MOPA_ERR_INVALID_ARG, "empty model");
    if (m.ngeom > 255) return fail(MOPA_ERR_LIMIT, "more than 255 collidable geoms");
    if (m.npair > 65535) return fail(MOPA_ERR_LIMIT, "more than 65535 candidate pairs");
    // The broad phase culls at zero margin: with a threshold > 0 a pair at distance (0, thr] would be reported or not
    // depending on the cull, and MuJoCo's own contact list (dist < margin) would have to be reproduced.  The reference
    // passes negative thresholds (config/sawyer.py:98-100, config/pusher.py:79-81); 0 keeps "any penetration".
    if (desc->contact_threshold > 0.0) return fail(MOPA_ERR_UNSUPPORTED, "contact_threshold > 0 is not supported (the broad phase culls at zero margin)");
    // FP32 broad phase (third-generation kernel): its conservativeness proof assumes coordinates of a few metres (absolute
    // slack 2e-5 m vs the float rounding of a coordinate).  `reach` bounds every model-determined coordinate; larger scenes
    // use the FP64 cull of the second generation.  Free-joint positions come from qpos at run time and are the caller's
    // responsibility (the Sawyer world box is +-1.2 m x 2 m, env/sawyer/sawyer.py:52-53).
    double reach = 0.0;
    {
        std::vector<double> rb(m.nbody, 0.0);
        for (int b = 1; b < m.nbody; b++) {
            const double *p = m.body_pos + 3 * b;
            rb[b] = rb[m.body_parent[b]] + sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            for (int j = m.body_jntadr[b]; j >= 0 && j < m.body_jntadr[b] + m.body_jntnum[b]; j++)
                if (m.jnt_type[j] == J_SLIDE && m.jnt_limited[j]) rb[b] += std::max(fabs(m.jnt_range[2 * j]), fabs(m.jnt_range[2 * j + 1]));
        }
        for (int g = 0; g < m.ngeom; g++) {
            if (m.geom_type[g] == G_PLANE) continue;
            const double *p = m.geom_pos + 3 * g, *z = m.geom_size + 3 * g;
            reach = std::max(reach, rb[m.geom_body[g]] + sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]));
        }
    }

    MopaScene *S = new MopaScene();
    S->nq = m.nq;
    S->seed = desc->seed;
    S->ngeom_model = m.ngeom;
    S->npair_model = m.npair;

    // --- active / passive split (KinematicPlanner.cpp:263-269) ---
    std::vector<char> is_passive(m.nq, 0);
    for (int i = 0; i < desc->n_passive; i++) {
        int a = desc->passive_qpos_idx[i];
        if (a < 0 || a >= m.nq) { delete S; return fail(MOPA_ERR_INVALID_ARG, "passive_qpos_idx out of range"); }
        is_passive[a] = 1;
    }
    std::vector<int> qpos_jnt(m.nq, -1);
    for (int j = 0; j < m.njnt; j++) {
        int w = (m.jnt_type[j] == J_FREE) ? 7 : (m.jnt_type[j] == J_BALL ? 4 : 1);
        for (int k = 0; k < w; k++)
            if (m.jnt_qposadr[j] + k < m.nq) qpos_jnt[m.jnt_qposadr[j] + k] = j;
    }
    std::vector<double> act_lo, act_hi, act_ext;
    std::vector<int32_t> act_adr, act_so2;
    std::vector<int> active_slot(m.nq, -1);
    for (int i = 0; i < m.nq; i++) {
        if (is_passive[i]) continue;
        int j = qpos_jnt[i];
        if (j < 0) { delete S; return fail(MOPA_ERR_INVALID_ARG, "active qpos address without a joint"); }
        if (m.jnt_type[j] == J_FREE || m.jnt_type[j] == J_BALL) {
            delete S;
            return fail(MOPA_ERR_UNSUPPORTED, "free/ball joints cannot be planned over (only hinge/slide are active in the reference scenes)");
        }
        active_slot[i] = (int)act_adr.size();
        act_adr.push_back(i);
        // limited hinge / slide -> R^1 with the joint range; unlimited hinge -> OMPL SO2StateSpace
        // (mujoco_ompl_interface.cpp:227-249): samples in [-pi, pi], maximum extent pi, wrap-around metric
        double lo = m.jnt_range[2 * j], hi = m.jnt_range[2 * j + 1];
        bool so2 = (m.jnt_type[j] == J_HINGE) && !m.jnt_limited[j];
        if (so2) { lo = -kPi; hi = kPi; }
        act_lo.push_back(lo);
        act_hi.push_back(hi);
        act_ext.push_back(so2 ? kPi : hi - lo);
        act_so2.push_back(so2 ? 1 : 0);
    }
    const int na = (int)act_adr.size();
    S->na = na;
    S->active_idx.assign(act_adr.begin(), act_adr.end());

    // --- which bodies matter, which are static ---
    for (int g = 0; g < m.ngeom; g++) {
        int t = m.geom_type[g];
        if (t == G_MESH) {
            const int id = (m.nmesh > 0 && m.geom_dataid) ? m.geom_dataid[g] : -1;
            if (id < 0 || id >= m.nmesh || !m.mesh_vert || m.mesh_vertnum[id] <= 0 ||
                m.mesh_vertadr[id] < 0 || m.mesh_vertadr[id] + m.mesh_vertnum[id] > m.nmeshvert) {
                delete S;
                return fail(MOPA_ERR_INVALID_ARG, "mesh geom " + std::to_string(g) + " without a convex hull (MopaModel.geom_dataid / mesh_vert)");
            }
        } else if (!(t == G_PLANE || t == G_SPHERE || t == G_CAPSULE || t == G_CYLINDER || t == G_BOX)) {
            delete S;
            return fail(MOPA_ERR_UNSUPPORTED, "collidable geom type " + std::to_string(t) + " (ellipsoid/hfield) is not supported");
        }
    }
    std::vector<char> needed(m.nbody, 0), is_static(m.nbody, 0);
    for (int g = 0; g < m.ngeom; g++) {
        int b = m.geom_body[g];
        while (b > 0 && !needed[b]) { needed[b] = 1; b = m.body_parent[b]; }
    }
    needed[0] = 1;
    is_static[0] = 1;
    for (int b = 1; b < m.nbody; b++) is_static[b] = (m.body_jntnum[b] == 0) && is_static[m.body_parent[b]];

    // static world frames (host FK with the same arithmetic as the device path)
    std::vector<double> xpos(3 * m.nbody, 0.0), xquat(4 * m.nbody, 0.0), xmat(9 * m.nbody, 0.0);
    xquat[0] = 1.0;
    quat2mat(&xmat[0], Q4{1.0, 0.0, 0.0, 0.0});
    for (int b = 1; b < m.nbody; b++) {
        if (!needed[b] || !is_static[b]) continue;
        int pid = m.body_parent[b];
        V3 v = mat_vec(&xmat[9 * pid], ld3(m.body_pos + 3 * b));
        V3 p = add3(ld3(&xpos[3 * pid]), v);
        const double *pq = &xquat[4 * pid], *bq = m.body_quat + 4 * b;
        Q4 q = quat_normalize(quat_mul(Q4{pq[0], pq[1], pq[2], pq[3]}, Q4{bq[0], bq[1], bq[2], bq[3]}));
        st3(&xpos[3 * b], p);
        xquat[4 * b] = q.w; xquat[4 * b + 1] = q.x; xquat[4 * b + 2] = q.y; xquat[4 * b + 3] = q.z;
        quat2mat(&xmat[9 * b], q);
    }

    // moving bodies in id (topological) order
    std::vector<int> mb_of_body(m.nbody, -1), sf_of_body(m.nbody, -1);
    std::vector<int> mb_body;
    for (int b = 1; b < m.nbody; b++)
        if (needed[b] && !is_static[b]) { mb_of_body[b] = (int)mb_body.size(); mb_body.push_back(b); }
    const int nmb = (int)mb_body.size();
    std::vector<double> sf_pos, sf_quat, sf_mat;
    auto sf_index = [&](int b) {
        if (sf_of_body[b] < 0) {
            sf_of_body[b] = (int)sf_pos.size() / 3;
            sf_pos.insert(sf_pos.end(), &xpos[3 * b], &xpos[3 * b] + 3);
            sf_quat.insert(sf_quat.end(), &xquat[4 * b], &xquat[4 * b] + 4);
            sf_mat.insert(sf_mat.end(), &xmat[9 * b], &xmat[9 * b] + 9);
        }
        return sf_of_body[b];
    };
    sf_index(0);

    std::vector<double> mb_pos, mb_quat, mj_axis, mj_pos, mj_ref;
    std::vector<int32_t> mb_parent, mb_jntadr, mb_jntnum, mj_type, mj_qsrc, pq_adr;
    std::vector<int> pq_slot(m.nq, -1);
    auto passive_slot = [&](int adr) {
        if (pq_slot[adr] < 0) { pq_slot[adr] = (int)pq_adr.size(); pq_adr.push_back(adr); }
        return na + pq_slot[adr];
    };
    for (int k = 0; k < nmb; k++) {
        int b = mb_body[k];
        int pid = m.body_parent[b];
        mb_parent.push_back(mb_of_body[pid] >= 0 ? mb_of_body[pid] : -(sf_index(pid) + 1));
        mb_pos.insert(mb_pos.end(), m.body_pos + 3 * b, m.body_pos + 3 * b + 3);
        mb_quat.insert(mb_quat.end(), m.body_quat + 4 * b, m.body_quat + 4 * b + 4);
        mb_jntadr.push_back((int)mj_type.size());
        mb_jntnum.push_back(m.body_jntnum[b]);
        for (int j = m.body_jntadr[b]; j < m.body_jntadr[b] + m.body_jntnum[b]; j++) {
            int t = m.jnt_type[j];
            if (t == J_BALL) { delete S; return fail(MOPA_ERR_UNSUPPORTED, "ball joints are not supported (the reference throws as well: mujoco_ompl_interface.cpp:217-229)"); }
            if (t == J_FREE && m.body_jntnum[b] != 1) { delete S; return fail(MOPA_ERR_UNSUPPORTED, "free joint combined with other joints"); }
            mj_type.push_back(t);
            mj_axis.insert(mj_axis.end(), m.jnt_axis + 3 * j, m.jnt_axis + 3 * j + 3);
            mj_pos.insert(mj_pos.end(), m.jnt_pos + 3 * j, m.jnt_pos + 3 * j + 3);
            mj_ref.push_back(t == J_FREE ? 0.0 : m.jnt_ref[j]);
            int adr = m.jnt_qposadr[j];
            if (t == J_FREE) {
                int first = passive_slot(adr);
                for (int c = 1; c < 7; c++) {
                    int sl = passive_slot(adr + c);
                    if (sl != first + c) { delete S; return fail(MOPA_ERR_UNSUPPORTED, "free joint qpos not contiguous in the passive list"); }
                }
                mj_qsrc.push_back(first);
            } else {
                mj_qsrc.push_back(active_slot[adr] >= 0 ? active_slot[adr] : passive_slot(adr));
            }
        }
    }
    const int nmj = (int)mj_type.size();
    const int n_pq = (int)pq_adr.size();

    // ancestor chains (root-most moving ancestor first)
    std::vector<int32_t> chain_adr(nmb), chain_len(nmb), chain_items;
    for (int k = 0; k < nmb; k++) {
        std::vector<int> path;
        for (int c = k; c >= 0; c = mb_parent[c]) path.push_back(c);
        std::reverse(path.begin(), path.end());
        chain_adr[k] = (int)chain_items.size();
        chain_len[k] = (int)path.size();
        chain_items.insert(chain_items.end(), path.begin(), path.end());
    }

    // geoms
    std::vector<double> g_lpos(3 * m.ngeom), g_lquat(4 * m.ngeom), g_rbound(m.ngeom), g_rec((size_t)kGeomStride * m.ngeom, 0.0);
    std::vector<int32_t> g_type(m.ngeom), g_slot(m.ngeom, -1), g_mb(m.ngeom, -1), mg_geom;
    for (int g = 0; g < m.ngeom; g++) {
        int b = m.geom_body[g];
        g_type[g] = m.geom_type[g];
        g_rbound[g] = rbound_of(m.geom_type[g], m.geom_size + 3 * g);
        if (m.geom_type[g] == G_MESH) {   // bounding radius about the geom origin: max |v| over the hull
            const double *V = m.mesh_vert + 3 * (size_t)m.mesh_vertadr[m.geom_dataid[g]];
            double r2 = 0.0;
            for (int i = 0; i < m.mesh_vertnum[m.geom_dataid[g]]; i++) r2 = dmax(r2, dot3(ld3(V + 3 * i), ld3(V + 3 * i)));
            g_rbound[g] = sqrt(r2);
        }
        std::memcpy(&g_lpos[3 * g], m.geom_pos + 3 * g, 24);
        std::memcpy(&g_lquat[4 * g], m.geom_quat + 4 * g, 32);
        double *rec = &g_rec[(size_t)kGeomStride * g];
        std::memcpy(rec + GO_SIZE, m.geom_size + 3 * g, 24);
        if (is_static[b]) {
            V3 gp = add3(ld3(&xpos[3 * b]), mat_vec(&xmat[9 * b], ld3(m.geom_pos + 3 * g)));
            const double *bq = &xquat[4 * b], *lq = m.geom_quat + 4 * g;
            Q4 gq = quat_mul(Q4{bq[0], bq[1], bq[2], bq[3]}, Q4{lq[0], lq[1], lq[2], lq[3]});
            st3(rec + GO_POS, gp);
            quat2mat(rec + GO_MAT, gq);
        } else {
            g_mb[g] = mb_of_body[b];
            g_slot[g] = (int)mg_geom.size();
            mg_geom.push_back(g);
        }
    }
    const int nmg = (int)mg_geom.size();
    if (nmg > 64) { delete S; return fail(MOPA_ERR_LIMIT, "more than 64 moving collidable geoms"); }

    // pairs: drop ignored (mujoco_ompl_interface.cpp:950-960), sort by narrow-phase cost class
    struct PairE { int code, g1, g2, model_idx; };
    std::vector<PairE> pairs;
    for (int p = 0; p < m.npair; p++) {
        int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
        if (g1 < 0 || g1 >= m.ngeom || g2 < 0 || g2 >= m.ngeom) { delete S; return fail(MOPA_ERR_INVALID_ARG, "pair_geom out of range"); }
        int a = m.geom_mjid[g1], b = m.geom_mjid[g2];
        int lo = std::min(a, b), hi = std::max(a, b);
        bool ignored = false;
        for (int i = 0; i < desc->n_ignored; i++)
            if (desc->ignored_pairs[2 * i] == lo && desc->ignored_pairs[2 * i + 1] == hi) ignored = true;
        if (ignored) continue;
        int code = pair_code(m.geom_type[g1], m.geom_type[g2]);
        if (code < 0) { delete S; return fail(MOPA_ERR_UNSUPPORTED, "unsupported geom type pair (must be ordered type1<=type2)"); }
        pairs.push_back(PairE{code, g1, g2, p});
    }
    std::stable_sort(pairs.begin(), pairs.end(), [](const PairE &a, const PairE &b) { return a.code < b.code; });
    std::vector<int32_t> pk(pairs.size());
    S->pair_slot.assign(m.npair, -1);
    for (size_t i = 0; i < pairs.size(); i++) {
        pk[i] = pairs[i].g1 | (pairs[i].g2 << 8) | (pairs[i].code << 16);
        S->pair_slot[pairs[i].model_idx] = (int)i;
    }

    // --- second-generation kernel: DFS program over the moving bodies + per-geom pair lists ---
    std::vector<int32_t> mb_load(nmb, -1), mb_save(nmb, -1), mb_mgadr(nmb, 0), mb_mgnum(nmb, 0);
    int n_save = 0;
    {
        std::vector<char> need_save(nmb, 0);
        for (int k = 0; k < nmb; k++) {
            int pk = mb_parent[k];
            if (pk >= 0 && pk != k - 1) need_save[pk] = 1;
        }
        std::vector<int> save_depth(nmb, 0);   // number of saved proper ancestors
        for (int k = 0; k < nmb; k++) {
            int pk = mb_parent[k];
            save_depth[k] = (pk >= 0) ? save_depth[pk] + (need_save[pk] ? 1 : 0) : 0;
            if (need_save[k]) { mb_save[k] = save_depth[k]; n_save = std::max(n_save, save_depth[k] + 1); }
            if (pk < 0) mb_load[k] = -2;
            else if (pk == k - 1) mb_load[k] = -1;
            else mb_load[k] = mb_save[pk];
        }
    }
    // moving geoms are in geom-id order == body order, so each body's geoms are a contiguous slot range
    for (int mslot = 0; mslot < nmg; mslot++) {
        int k = g_mb[mg_geom[mslot]];
        if (mb_mgnum[k] == 0) mb_mgadr[k] = mslot;
        else if (mb_mgadr[k] + mb_mgnum[k] != mslot) { delete S; return fail(MOPA_ERR_UNSUPPORTED, "moving geoms of a body are not contiguous"); }
        mb_mgnum[k]++;
    }
    for (int k = 1; k < nmb; k++)   // slots must follow body order for the "earlier partner" rule
        if (mb_mgnum[k] && mb_mgnum[k - 1] && mb_mgadr[k] < mb_mgadr[k - 1]) { delete S; return fail(MOPA_ERR_UNSUPPORTED, "geom order does not follow body order"); }
    // Per-owner-geom pair lists for the lane-per-state kernels.  Pairs whose class involves a mesh are kept in a list
    // of their own: the main pass (k_is_valid_v5 / v2) then carries no mesh code at all, and a second, light pass of
    // the MESH instantiation handles the handful of mesh pairs and folds its verdict into the first one's.
    std::vector<int32_t> mg_store(nmg, 0);
    auto build_lists = [&](int mode /*0 = mesh-free pairs, 1 = mesh pairs, 2 = all*/, std::vector<int32_t> &padr, std::vector<int32_t> &pnum, std::vector<int32_t> &words) {
        padr.assign(nmg, 0); pnum.assign(nmg, 0); words.clear();
        std::vector<std::vector<PairE>> own(nmg);
        for (const PairE &e : pairs) {   // already sorted by cost class
            const bool is_mesh = (e.code == PC_PLANE_MESH || e.code == PC_CONVEX_MESH);
            if (mode != 2 && is_mesh != (mode == 1)) continue;
            int s1 = g_slot[e.g1], s2 = g_slot[e.g2];
            int owner = (s2 > s1) ? s2 : s1;
            own[owner].push_back(e);
        }
        for (int mslot = 0; mslot < nmg; mslot++) {
            padr[mslot] = (int)words.size();
            pnum[mslot] = (int)own[mslot].size();
            for (const PairE &e : own[mslot]) {
                int cur = mg_geom[mslot];
                int cur_is_g2 = (e.g2 == cur) ? 1 : 0;
                int partner = cur_is_g2 ? e.g1 : e.g2;
                int pslot = g_slot[partner];
                if (pslot >= 0) mg_store[pslot] = 1;
                words.push_back(partner | (e.code << 8) | (cur_is_g2 << 12) | ((pslot >= 0 ? 1 : 0) << 13) | ((pslot >= 0 ? pslot : 0) << 14));
            }
        }
    };
    std::vector<int32_t> mg_padr, mg_pnum, gp_word, mg_padr_mesh, mg_pnum_mesh, gp_word_mesh;
    build_lists(0, mg_padr, mg_pnum, gp_word);
    build_lists(1, mg_padr_mesh, mg_pnum_mesh, gp_word_mesh);
    // The third-generation kernel culls the mesh pairs too (FP32, a few table entries more) -- not to evaluate them, but
    // to tell the second pass which states have one within reach at all: almost none do, and that pass then skips
    // whole tiles instead of posing every state again for nothing.
    std::vector<int32_t> t5_padr = mg_padr, t5_pnum = mg_pnum, t5_word = gp_word;
    if (!gp_word_mesh.empty()) build_lists(2, t5_padr, t5_pnum, t5_word);

    // v5: FP32 broad-phase table, one 32-byte entry per (owner geom, partner) pair:
    //   [0..2] partner centre (static partners) / a point of the plane,
    //   [3] (owner radius + eps + partner radius)^2 / for a plane: owner radius + eps,
    //   [4..6] world-AABB half extents of a static partner + owner radius + eps / the plane normal,
    //   [7] flags: bits 0..13 = low bits of gp_word (partner gid, code, cur_is_g2, pmov), 14..21 partner slot, 30 plane
    // Within a geom's range the entries are ordered [moving partners | static non-plane partners | planes] (the kernel
    // runs one branch-free loop per group); the group sizes follow the table: tab[8 n_gp + slot] = nmov | nstat<<8 | nplane<<16.
    const size_t n5 = t5_word.size();
    std::vector<int32_t> gp_tab(8 * n5 + 3 * (size_t)nmg, 0);   // entries, then per geom: group counts, then (first entry, count)
    int max_pnum = 0;
    {
        auto f2i = [](double x) { float f = (float)x; int32_t i; std::memcpy(&i, &f, 4); return i; };
        for (int mslot = 0; mslot < nmg; mslot++) {
            max_pnum = std::max(max_pnum, (int)t5_pnum[mslot]);
            gp_tab[8 * n5 + nmg + 2 * mslot] = t5_padr[mslot];
            gp_tab[8 * n5 + nmg + 2 * mslot + 1] = t5_pnum[mslot];
            std::vector<int> order[3];
            for (int p = t5_padr[mslot]; p < t5_padr[mslot] + t5_pnum[mslot]; p++) {
                const int w = t5_word[p];
                const int grp = ((w >> 13) & 1) ? 0 : (m.geom_type[w & 0xff] == G_PLANE ? 2 : 1);
                order[grp].push_back(p);
            }
            gp_tab[8 * n5 + mslot] = (int)order[0].size() | ((int)order[1].size() << 8) | ((int)order[2].size() << 16);
            if (getenv("MOPA_DEBUG"))
                fprintf(stderr, "[mopa]   geom slot %d: %zu moving + %zu static + %zu plane partners\n", mslot, order[0].size(), order[1].size(), order[2].size());
            size_t dst = (size_t)t5_padr[mslot];
            for (int grp = 0; grp < 3; grp++)
                for (int p : order[grp]) {
                    const int w = t5_word[p];
                    const int pg = w & 0xff, pmov = (w >> 13) & 1;
                    int32_t *te = &gp_tab[8 * dst++];
                    // the owner's inflated radius is folded into the entry (an entry belongs to one owner geom): FP32
                    // arithmetic here = what the kernel would do per pair and state
                    const float rg = (float)g_rbound[mg_geom[mslot]] + kCullEps;
                    auto ff2i = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i; };
                    float rs = rg + (float)g_rbound[pg];
                    if (desc->pair_cull_radius) {
                        // a proven bound on the centre distance at which this pair can reach the threshold at all
                        const int own = mg_geom[mslot];
                        for (int pp = 0; pp < m.npair; pp++) {
                            const int a = m.pair_geom[2 * pp], b = m.pair_geom[2 * pp + 1];
                            if (((a == own && b == pg) || (a == pg && b == own)) && desc->pair_cull_radius[pp] > 0.0)
                                rs = std::min(rs, std::nextafter((float)desc->pair_cull_radius[pp], 1.0e30f) + kCullEps);
                        }
                    }
                    te[3] = ff2i(rs * rs);
                    if (pmov) {
                        // [0] (moving partners only; their centre comes from the tile's table): the square of the centre distance
                        // below which the two geoms' INSCRIBED balls (radius r of a sphere / capsule, min(r, h) of a cylinder,
                        // the smallest half extent of a box, centred where the geom is) overlap by more than the threshold
                        // + 0.1 mm -- then the pair's distance is <= the threshold whatever its class computes (closed forms
                        // are exact, SAT reports the true depth, the portal refinement never less than the true depth - 1e-6):
                        // the verdict-only kernels call such a state invalid in the broad phase and drop all its entries
                        auto r_in = [&](int g) -> double {
                            const double *sz = &g_rec[(size_t)kGeomStride * g + GO_SIZE];
                            switch (m.geom_type[g]) {
                                case G_SPHERE: case G_CAPSULE: return sz[0];
                                case G_CYLINDER: return std::min(sz[0], sz[1]);
                                case G_BOX: return std::min(sz[0], std::min(sz[1], sz[2]));
                                default: return 0.0;
                            }
                        };
                        const double ra = r_in(mg_geom[mslot]), rb = r_in(pg);
                        const double reach = ra + rb - (std::max(0.0, -desc->contact_threshold) + 1e-4) - 4.0 * kCullEps;
                        te[0] = ff2i((ra > 0.0 && rb > 0.0 && reach > 0.0) ? (float)(reach * reach) * (1.0f - 1e-6f) : 0.0f);
                    }
                    int flags = w & 0x3fffff;    // gp_word already carries the slot in bits 14..21
                    if (!pmov) {
                        const double *rec = &g_rec[(size_t)kGeomStride * pg];
                        te[0] = f2i(rec[GO_POS]); te[1] = f2i(rec[GO_POS + 1]); te[2] = f2i(rec[GO_POS + 2]);
                        if (m.geom_type[pg] == G_PLANE) {
                            te[4] = f2i(rec[GO_MAT + 2]); te[5] = f2i(rec[GO_MAT + 5]); te[6] = f2i(rec[GO_MAT + 8]);
                            te[3] = ff2i(rg);
                            flags |= 1 << 30;
                        } else {
                            double H[3];
                            static_aabb_half(m.geom_type[pg], rec, g_rbound[pg], H);
                            te[4] = ff2i((float)H[0] + rg); te[5] = ff2i((float)H[1] + rg); te[6] = ff2i((float)H[2] + rg);
                        }
                    }
                    te[7] = flags;
                }
        }
    }
    S->h_gp_tab = gp_tab;

    std::vector<int32_t> mbr(8 * (size_t)nmb, 0), mgr(4 * (size_t)nmg, 0), mgr_mesh(4 * (size_t)nmg, 0);
    std::vector<double> mbd(16 * (size_t)nmb, 0.0), mgd(8 * (size_t)nmg, 0.0);
    for (int k = 0; k < nmb; k++) {
        int ja = mb_jntadr[k], jn = mb_jntnum[k];
        int32_t *r = &mbr[8 * (size_t)k];
        r[0] = jn; r[1] = ja; r[2] = mb_load[k]; r[3] = (mb_parent[k] < 0) ? -(mb_parent[k] + 1) : 0;
        r[4] = mb_save[k]; r[5] = mb_mgadr[k]; r[6] = mb_mgnum[k];
        // joint 0: type (7 bits) | bit 7 = anchor at the body origin (mopa_device.hpp: apply_joint) | value slot << 8
        const bool jp_zero = jn > 0 && mj_pos[3 * (size_t)ja] == 0.0 && mj_pos[3 * (size_t)ja + 1] == 0.0 && mj_pos[3 * (size_t)ja + 2] == 0.0;
        r[7] = (jn > 0) ? ((mj_type[ja] & 0x7f) | (jp_zero ? 0x80 : 0) | (mj_qsrc[ja] << 8)) : 0x7f;
        double *d = &mbd[16 * (size_t)k];
        std::memcpy(d, &mb_pos[3 * (size_t)k], 24);
        std::memcpy(d + 3, &mb_quat[4 * (size_t)k], 32);
        if (jn > 0) {
            std::memcpy(d + 7, &mj_axis[3 * (size_t)ja], 24);
            std::memcpy(d + 10, &mj_pos[3 * (size_t)ja], 24);
            d[13] = mj_ref[ja];
        }
    }
    for (int ms = 0; ms < nmg; ms++) {
        int g = mg_geom[ms];
        int32_t *r = &mgr[4 * (size_t)ms];
        r[0] = g; r[1] = mg_store[ms] | ((m.geom_type[g] == G_BOX ? 1 : 0) << 1); r[2] = mg_padr[ms]; r[3] = mg_pnum[ms];
        int32_t *rm = &mgr_mesh[4 * (size_t)ms];
        rm[0] = r[0]; rm[1] = r[1]; rm[2] = mg_padr_mesh[ms]; rm[3] = mg_pnum_mesh[ms];
        double *d = &mgd[8 * (size_t)ms];
        std::memcpy(d, &g_lpos[3 * (size_t)g], 24);
        std::memcpy(d + 3, &g_lquat[4 * (size_t)g], 32);
        d[7] = g_rbound[g];
    }

    // --- assemble blobs ---
    Builder B;
    SceneHdr &h = S->hdr;
    h.na = na; h.nq = m.nq; h.n_pq = n_pq; h.nmb = nmb; h.nmj = nmj; h.nsf = (int)sf_pos.size() / 3;
    h.ng = m.ngeom; h.nmg = nmg; h.npair = (int)pairs.size();
    h.o_mb_pos = B.add_d(mb_pos); h.o_mb_quat = B.add_d(mb_quat);
    h.o_sf_pos = B.add_d(sf_pos); h.o_sf_quat = B.add_d(sf_quat); h.o_sf_mat = B.add_d(sf_mat);
    h.o_mj_axis = B.add_d(mj_axis); h.o_mj_pos = B.add_d(mj_pos); h.o_mj_ref = B.add_d(mj_ref);
    h.o_g_lpos = B.add_d(g_lpos); h.o_g_lquat = B.add_d(g_lquat); h.o_g_rbound = B.add_d(g_rbound);
    {
        std::vector<double> g_aabb(3 * (size_t)m.ngeom, 0.0);
        for (int g = 0; g < m.ngeom; g++)
            if (g_slot[g] < 0 && m.geom_type[g] != G_PLANE)
                static_aabb_half(m.geom_type[g], &g_rec[(size_t)kGeomStride * g], g_rbound[g], &g_aabb[3 * (size_t)g]);
        h.o_g_aabb = B.add_d(g_aabb);
    }
    if (B.dbl.size() & 1) B.dbl.push_back(0.0);   // 16-byte align the posed records
    // mesh hulls live in the double blob; a mesh geom's record carries (blob offset of its vertices, vertex count)
    // where primitives carry their size (mopa_device.hpp: mesh_support_local / d_plane_mesh)
    if (m.nmesh > 0) {
        h.has_mesh = 1;
        h.o_mesh = B.add_d(std::vector<double>(m.mesh_vert, m.mesh_vert + 3 * (size_t)m.nmeshvert));
        S->n_mesh_dbl = 3 * (int)m.nmeshvert;
        for (int g = 0; g < m.ngeom; g++)
            if (m.geom_type[g] == G_MESH) {
                double *rec = &g_rec[(size_t)kGeomStride * g];
                rec[GO_SIZE] = (double)(h.o_mesh + 3 * m.mesh_vertadr[m.geom_dataid[g]]);
                rec[GO_SIZE + 1] = (double)m.mesh_vertnum[m.geom_dataid[g]];
                rec[GO_SIZE + 2] = 0.0;
            }
    }
    h.o_g_rec = B.add_d(g_rec);
    h.o_act_lo = B.add_d(act_lo); h.o_act_hi = B.add_d(act_hi); h.o_act_ext = B.add_d(act_ext);
    if (B.dbl.size() & 1) B.dbl.push_back(0.0);
    h.o_mbd = B.add_d(mbd); h.o_mgd = B.add_d(mgd);
    if (B.dbl.size() & 1) B.dbl.push_back(0.0);
    h.o_mb_parent = B.add_i(mb_parent); h.o_mb_jntadr = B.add_i(mb_jntadr); h.o_mb_jntnum = B.add_i(mb_jntnum);
    h.o_mj_type = B.add_i(mj_type); h.o_mj_qsrc = B.add_i(mj_qsrc);
    h.o_g_type = B.add_i(g_type); h.o_g_slot = B.add_i(g_slot); h.o_g_mb = B.add_i(g_mb);
    h.o_mg_geom = B.add_i(mg_geom); h.o_chain_adr = B.add_i(chain_adr); h.o_chain_len = B.add_i(chain_len);
    h.o_chain_items = B.add_i(chain_items); h.o_pairs = B.add_i(pk); h.o_pq_adr = B.add_i(pq_adr);
    h.o_act_adr = B.add_i(act_adr); h.o_act_so2 = B.add_i(act_so2);
    {
        std::vector<double> act_ref(8, 0.0);
        std::vector<int32_t> act_hinge(8, 0);
        for (int j = 0; j < nmj; j++)
            if (mj_qsrc[j] >= 0 && mj_qsrc[j] < na && mj_qsrc[j] < 8) {
                act_ref[mj_qsrc[j]] = mj_ref[j];
                act_hinge[mj_qsrc[j]] = mj_type[j] == J_HINGE;
            }
        h.o_act_ref = B.add_d(act_ref);
        h.o_act_hinge = B.add_i(act_hinge);
    }
    h.n_save = n_save; h.n_gp = (int)t5_word.size();
    h.o_mb_load = B.add_i(mb_load); h.o_mb_save = B.add_i(mb_save); h.o_mb_mgadr = B.add_i(mb_mgadr); h.o_mb_mgnum = B.add_i(mb_mgnum);
    h.o_mg_padr = B.add_i(mg_padr); h.o_mg_pnum = B.add_i(mg_pnum); h.o_mg_store = B.add_i(mg_store); h.o_gp_word = B.add_i(gp_word);
    while (B.ints.size() & 7) B.ints.push_back(0);   // 32-byte align the packed records (scalar dwordx8 loads)
    h.o_mbr = B.add_i(mbr); h.o_mgr = B.add_i(mgr);
    {
        // tile-shared passive bodies (see SceneHdr)
        std::vector<char> pas(nmb, 0);
        std::vector<int> lvl(nmb, 0);
        int nlv = 0;
        for (int k = 0; k < nmb; k++) {
            bool p = true, is_free = false;
            for (int j = mb_jntadr[k]; j < mb_jntadr[k] + mb_jntnum[k]; j++) {
                if (mj_type[j] == J_FREE) is_free = true;
                else if (mj_qsrc[j] < na) p = false;
            }
            if (mb_parent[k] >= 0 && !is_free) { p = p && pas[mb_parent[k]]; lvl[k] = lvl[mb_parent[k]] + 1; }
            pas[k] = p ? 1 : 0;
            if (p) nlv = std::max(nlv, lvl[k] + 1);
        }
        std::vector<int32_t> pas_b, pas_lv, mb_pas(nmb, -1), pas_g, mg_pas(nmg, -1);
        for (int L = 0; L < nlv; L++) {
            pas_lv.push_back((int)pas_b.size());
            for (int k = 0; k < nmb; k++)
                if (pas[k] && lvl[k] == L) { mb_pas[k] = (int)pas_b.size(); pas_b.push_back(k); }
        }
        pas_lv.push_back((int)pas_b.size());
        for (int ms = 0; ms < nmg; ms++)
            if (pas[g_mb[mg_geom[ms]]]) { mg_pas[ms] = (int)pas_g.size(); pas_g.push_back(ms); }
        for (int k = 0; k < nmb; k++) {
            if (mb_pas[k] < 0) continue;
            bool cont = false;       // does a body that is NOT posed by the tile continue from this one's registers / saved pose?
            for (int c = 0; c < nmb; c++)
                if (mb_parent[c] == k && !pas[c]) cont = true;
            if (!cont) mb_pas[k] |= 1 << 16;
        }
        const size_t pas_min = std::getenv("MOPA_V5_TILE_MIN") ? (size_t)atoi(std::getenv("MOPA_V5_TILE_MIN")) : 4;      // (A/B knob)
        const bool on = pas_b.size() >= pas_min && pas_b.size() <= 64 && pas_g.size() <= 64 && nlv <= 8 && !std::getenv("MOPA_V5_NO_TILE_POSES");
        h.n_pas_b = on ? (int)pas_b.size() : 0; h.n_pas_g = on ? (int)pas_g.size() : 0; h.n_pas_lv = on ? nlv : 0;
        h.o_pas_b = B.add_i(pas_b); h.o_pas_lv = B.add_i(pas_lv); h.o_mb_pas = B.add_i(mb_pas); h.o_pas_g = B.add_i(pas_g); h.o_mg_pas = B.add_i(mg_pas);
    }
    {
        std::vector<int32_t> pfk(8 * (size_t)nmg, 0);
        int maxlen = 0;
        bool ok = nmb < 255;
        for (int ms = 0; ms < nmg && ok; ms++) {
            const int k = g_mb[mg_geom[ms]];
            const int len = chain_len[k];
            if (len > 16) { ok = false; break; }
            maxlen = std::max(maxlen, len);
            uint32_t w[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            for (int i = 0; i < len; i++) {
                const uint32_t b = (uint32_t)chain_items[chain_adr[k] + i];
                w[i >> 2] = (w[i >> 2] & ~(0xffu << (8 * (i & 3)))) | (b << (8 * (i & 3)));
            }
            for (int i = 0; i < 4; i++) pfk[8 * (size_t)ms + i] = (int32_t)w[i];
            pfk[8 * (size_t)ms + 4] = len;
            const int root = chain_items[chain_adr[k]];
            pfk[8 * (size_t)ms + 5] = (mb_parent[root] < 0) ? -(mb_parent[root] + 1) : 0;
        }
        h.o_pfk = B.add_i(pfk);
        h.pfk_maxlen = ok ? maxlen : 0;
    }
    const int o_mgr_mesh = B.add_i(mgr_mesh), o_gp_word_mesh = B.add_i(gp_word_mesh);
    h.n_dbl = (int)B.dbl.size();
    h.n_int = (int)B.ints.size();
    h.wave_dbl = nmg * kGeomStride + na + n_pq + na + 2 * nmj;   // geom records, joint values, one spare state vector, sin/cos table
    int wl_bytes = (int)((pairs.size() * 2 + 15) & ~size_t(15));
    h.wave_bytes = ((h.wave_dbl * 8 + wl_bytes) + 15) & ~15;
    h.thr = desc->contact_threshold;
    h.range = desc->range;
    h.resolution = desc->resolution > 0.0 ? desc->resolution : 0.005;
    {
        // FP32 mirror of the planner's trees: coordinates rounded to FP32 (|x| <= X: error X 2^-24 each), their difference
        // rounded once more, na terms added with a rounding of the running sum (<= na 2X) each
        double X = kPi;
        for (int i = 0; i < na; i++) X = std::max(X, std::max(std::fabs(act_lo[i]), std::fabs(act_hi[i])));
        const double S1 = 2.0 * X * std::max(na, 1);
        h.nn_eps = (3.0 * na + 2.0) * std::ldexp(1.0, -24) * S1 * 1.5;
    }
    S->h_dbl = B.dbl;
    S->h_int = B.ints;
    S->lds_bytes = h.n_dbl * 8 + ((h.n_int + 1) & ~1) * 4 + kWavesPerBlock * h.wave_bytes;
    if (S->lds_bytes > kMaxLdsBytes) { delete S; return fail(MOPA_ERR_LIMIT, "scene does not fit the 160 KiB LDS"); }
    {
        S->v2_lds_bytes = h.n_dbl * 8 + ((h.n_int + 1) & ~1) * 4 + kWavesPerBlock * kV2LdsPerWave;
        const char *ev = std::getenv("MOPA_VALID_KERNEL");
        S->use_v2 = !(ev && std::string(ev) == "v1") && S->v2_lds_bytes <= kMaxLdsBytes;
        S->v2_forced = ev && std::string(ev) == "v2";
        {
            // largest entry buffer (multiple of 64, 256..1024) that still lets two workgroups share a CU's 160 KiB of LDS;
            // if even the smallest does not fit with the FP32 centre table in LDS, the centres are read back from the slab
            const int fixed = h.n_dbl * 8 + ((h.n_int + 3) & ~3) * 4 + ((8 * h.n_gp + (gp_word_mesh.empty() ? 1 : 3) * nmg + 3) & ~3) * 4;
            const char *ec = std::getenv("MOPA_V5_CENTRES");       // "lds" / "slab": A/B runs and tests
            bool cen_lds = true;
            int cap = kEntCapV5Max;
            for (int attempt = 0; attempt < 2; attempt++) {
                cen_lds = attempt == 0 && gp_word_mesh.empty();   // scenes with mesh pairs: the slab-centre instantiation carries the gate
                if (ec && std::string(ec) == "slab") cen_lds = false;
                if (ec && std::string(ec) == "lds" && gp_word_mesh.empty()) cen_lds = true;   // (mesh scenes: always slab + gate)
                const int n_cen = cen_lds ? nmg : 0;
                cap = kEntCapV5Max;
                while (cap > 256 && fixed + kWavesPerBlock * v5_lds_per_wave(n_cen, cap, true) > 80 * 1024) cap -= 64;
                if (fixed + kWavesPerBlock * v5_lds_per_wave(n_cen, cap, true) <= 80 * 1024) break;
                if (ec && std::string(ec) == "lds") break;
            }
            if (fixed + kWavesPerBlock * v5_lds_per_wave(cen_lds ? nmg : 0, cap, true) > 80 * 1024) {   // one workgroup per CU anyway
                cap = 768;
                cen_lds = gp_word_mesh.empty() && !(ec && std::string(ec) == "slab");
            }
            S->v5_cen_lds = cen_lds;
            // (cap so far: the depth-reporting instantiations; the verdict-only ones have no depth words and take more entries)
            S->v5_ent_cap_md = cap;
            S->v5_lds_bytes_md = fixed + kWavesPerBlock * v5_lds_per_wave(cen_lds ? nmg : 0, cap, true);
            const int budget = std::max(S->v5_lds_bytes_md, 80 * 1024);
            while (cap + 64 <= kEntCapV5Max && fixed + kWavesPerBlock * v5_lds_per_wave(cen_lds ? nmg : 0, cap + 64, false) <= budget) cap += 64;
            h.v5_ent_cap = cap;
            // (the tile's passive poses overlay the entry buffer during the FK phase)
            if ((h.n_pas_b + h.n_pas_g) * 7 * 8 > 4 * std::min(cap, S->v5_ent_cap_md)) { h.n_pas_b = 0; h.n_pas_g = 0; h.n_pas_lv = 0; }
            S->v5_lds_bytes = fixed + kWavesPerBlock * v5_lds_per_wave(cen_lds ? nmg : 0, cap, false);
            if (std::getenv("MOPA_DEBUG"))
                fprintf(stderr, "[mopa] scene: nmg %d nmb %d save slots %d pairs %d (+%d mesh) lds: wave-per-state %d, v2 %d, v5 %d (entry cap %d / %d, fixed %d, centres in %s); tile-posed bodies %d geoms %d levels %d\n", nmg, nmb, n_save,
                        (int)gp_word.size(), (int)gp_word_mesh.size(), S->lds_bytes, S->v2_lds_bytes, S->v5_lds_bytes, cap, S->v5_ent_cap_md, fixed, cen_lds ? "LDS" : "slab", h.n_pas_b, h.n_pas_g, h.n_pas_lv);
        }
        // third generation (FP32 broad phase out of LDS): default wherever it applies; MOPA_VALID_KERNEL=v2 keeps the second
        // ... unless it would get one workgroup per CU where the second generation still gets two (LDS: the FP32 centre
        // table grows with the number of moving geoms; SawyerLift: 19 of them)
        const bool v5_fits2 = S->v5_lds_bytes <= 80 * 1024, v2_fits2 = S->v2_lds_bytes <= 80 * 1024;
        S->use_v5 = !(ev && std::string(ev) == "v2") && S->use_v2 && max_pnum <= 64 && std::max(S->v5_lds_bytes, S->v5_lds_bytes_md) <= kMaxLdsBytes && reach <= kV5MaxReach &&
                    (v5_fits2 || !v2_fits2 || (ev && std::string(ev) == "v5"));
        if (ev && std::string(ev) == "v5") S->v2_forced = true;   // "v5" also forces the lane-per-state path for every N >= 64
    }

    S->hdr_mesh = h;
    S->hdr_mesh.o_mgr = o_mgr_mesh; S->hdr_mesh.o_gp_word = o_gp_word_mesh; S->hdr_mesh.n_gp = (int)gp_word_mesh.size();
    S->n_mesh_gp = (int)gp_word_mesh.size();

    // --- device upload ---
    int ndev = mopa_device_count();
    if (ndev <= 0) { delete S; return fail(MOPA_ERR_HIP, "no HIP device visible: libmopa_hip has no CPU fallback"); }
    if (desc->device >= 0) {
        if (desc->device >= ndev) { delete S; return fail(MOPA_ERR_INVALID_ARG, "device ordinal out of range"); }
        S->device = desc->device;
    } else if (hipGetDevice(&S->device) != hipSuccess) { delete S; return fail(MOPA_ERR_HIP, "hipGetDevice failed"); }
    DeviceGuard guard(S->device);      // the caller's current device is restored on every exit path
    if (!guard.ok()) { delete S; return fail(MOPA_ERR_HIP, "cannot switch to the requested HIP device"); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, S->device) == hipSuccess) S->n_cu = prop.multiProcessorCount;
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 8);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e1 = up((void **)&S->d_dbl, S->h_dbl.data(), S->h_dbl.size() * 8);
    hipError_t e2 = up((void **)&S->d_int, S->h_int.data(), S->h_int.size() * 4);
    if (e2 == hipSuccess) e2 = up((void **)&S->d_gp_tab, S->h_gp_tab.data(), S->h_gp_tab.size() * 4);
    S->dbg_doubles = (size_t)kGeomStride * m.ngeom + pairs.size() + 8;
    hipError_t e3 = hipMalloc((void **)&S->d_q, sizeof(double) * (size_t)(m.nq + na + 8));
    hipError_t e4 = hipMalloc((void **)&S->d_valid, 8);
    hipError_t e5 = hipMalloc((void **)&S->d_md, 8);
    hipError_t e6 = hipMalloc((void **)&S->d_dbg, sizeof(double) * S->dbg_doubles);
    for (hipError_t e : {e1, e2, e3, e4, e5, e6})
        if (e != hipSuccess) { mopa_scene_destroy(S); return fail(MOPA_ERR_HIP, std::string("device allocation: ") + hipGetErrorString(e)); }
    // allow the dynamic LDS size
    // the attribute is per function, not per scene: register the device maximum once so scenes of different sizes can
    // coexist in one process in any creation order (each launch still passes its own, checked, size)
    for (const void *k : {(const void *)k_is_valid<false, false>, (const void *)k_is_valid<true, false>, (const void *)k_is_valid<false, true>,
                          (const void *)k_is_valid<true, true>, (const void *)k_check_motion, (const void *)k_debug_state<false>,
                          (const void *)k_debug_state<true>, (const void *)k_is_valid_v2<false, false>,
                          (const void *)k_is_valid_v2<true, false>, (const void *)k_is_valid_v2<false, true>,
                          (const void *)k_is_valid_v2<true, true>, (const void *)k_is_valid_v5<false, true, false>, (const void *)k_is_valid_v5<true, true, false>,
                          (const void *)k_is_valid_v5<false, false, false>, (const void *)k_is_valid_v5<true, false, false>,
                          (const void *)k_is_valid_v5<false, false, true>, (const void *)k_is_valid_v5<true, false, true>})
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
    plan_register_lds();
    *out = S;
    return MOPA_OK;
}

extern "C" void mopa_scene_destroy(MopaScene *S) {
    if (!S) return;
    DeviceGuard guard(S->device);
    (void)hipDeviceSynchronize();      // nothing of this scene may still be running when its tables go away
    for (void *q : {(void *)S->d_dbl, (void *)S->d_int, (void *)S->d_q, (void *)S->d_valid, (void *)S->d_md, (void *)S->d_dbg, (void *)S->d_gp_tab})
        if (q) (void)hipFree(q);
    for (auto &kv : S->scratch) {
        StreamScratch &sc = kv.second;
        for (DevBuf *b : {&sc.slab, &sc.mpr, &sc.cen, &sc.mesh_list, &sc.mesh_rows, &sc.mv_cnt, &sc.mv_off, &sc.mv_env, &sc.mv_q, &sc.mv_valid, &sc.mv_scan, &sc.plan_q,
                          &sc.plan_p, &sc.plan_ctr, &sc.pb_small, &sc.pb_rows, &sc.pb_act, &sc.ip_walk})
            if (b->p) (void)hipFree(b->p);
    }
    for (void *q : S->retired) (void)hipFree(q);
    delete S;
}

extern "C" int mopa_scene_num_active(const MopaScene *S) { return S ? S->na : -1; }
extern "C" int mopa_scene_active_idx(const MopaScene *S, int32_t *out) {
    if (!S || !out) return fail(MOPA_ERR_INVALID_ARG, "null argument");
    std::memcpy(out, S->active_idx.data(), sizeof(int32_t) * S->na);
    return MOPA_OK;
}
extern "C" int mopa_scene_num_pairs(const MopaScene *S) { return S ? S->hdr.npair : -1; }
extern "C" int mopa_scene_lds_bytes(const MopaScene *S) { return S ? S->lds_bytes : -1; }

extern "C" int mopa_scene_valid_kernel(const MopaScene *S, int64_t N, char *out, int32_t cap) {
    if (!S || !out || cap < 24) return fail(MOPA_ERR_INVALID_ARG, "null argument / buffer under 24 bytes");
    const int64_t v2_min = S->v2_forced ? 64 : std::max<int64_t>(64, (int64_t)S->n_cu * 36);
    const char *name = "k_is_valid";
    if (S->use_v2 && N >= v2_min) name = S->use_v5 ? "k_is_valid_v5" : "k_is_valid_v2";
    std::snprintf(out, (size_t)cap, "%s", name);
    return MOPA_OK;
}

static int grid_for(const MopaScene *S, int64_t N) {
    int64_t blocks = (N + kWavesPerBlock - 1) / kWavesPerBlock;
    int64_t cap = (int64_t)S->n_cu * 8;
    return (int)std::max<int64_t>(1, std::min(blocks, cap));
}

// state validity of N states; env row of state i = env_idx ? env_idx[i] : i / samples_per_env
// n_dev (nullable): the number of states is only known on the device (*n_dev <= N, N then sizes the launch): served by the
// lane-per-state kernel, which reads it when it starts -- no host read-back between the producer of the states and this launch
static int launch_is_valid(MopaScene *S, const double *q_active, const double *qpos_env, int64_t N, int64_t samples_per_env,
                           const int *env_idx, uint8_t *valid, double *min_dist, void *stream, const long long *n_dev = nullptr) {
    if (!S || !valid || (N > 0 && (!q_active || !qpos_env))) return fail(MOPA_ERR_INVALID_ARG, "null argument");
    if (N < 0 || samples_per_env <= 0) return fail(MOPA_ERR_INVALID_ARG, "N < 0 or samples_per_env <= 0");
    if (N == 0) return MOPA_OK;
    ON_DEVICE(S->device);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(kBlock);
    // Kernel choice: the lane-per-state kernel needs ~175 us for a 64-state tile however few tiles there are, the
    // wave-per-state kernel ~23 us per state-wave with 8+ waves per CU in flight -- so small batches (one state per
    // env, e.g. the collision gate of the kinematic env.step) go to the latter.  Measured crossover on MI355X
    // (tools/crossover.py): 8192 states 155 vs 171 us, 12288 states 202 vs 168 us  =>  ~36 states per CU.
    const int64_t v2_min = S->v2_forced ? 64 : std::max<int64_t>(64, (int64_t)S->n_cu * 36);
    if (n_dev && !(S->use_v2 && S->use_v5)) return fail(MOPA_ERR_UNSUPPORTED, "a device-side state count needs the lane-per-state kernel");
    // Small batches with explicit env rows or a device-side count used to be sent to the lane-per-state kernel regardless -- at the latency of
    // one 64-state tile (133 us on Push, 289 us on Assembly) for a few hundred states.  The wave-per-state kernel takes env_idx / n_dev too now;
    // a batch whose count is only known on the device and whose WORST CASE is large gets both launches: each reads the count and one of them
    // leaves at once (the wave-per-state one if the count reached v2_min, the lane-per-state one below it).
    auto kern1 = S->hdr.has_mesh ? (min_dist ? k_is_valid<true, true> : k_is_valid<false, true>)
                                 : (min_dist ? k_is_valid<true, false> : k_is_valid<false, false>);
    long long n_small = 0;         // lane-per-state launch: counts below it are the other launch's
    if (n_dev && !S->v2_forced && N >= v2_min && !std::getenv("MOPA_NO_DUAL_DISPATCH")) {
        n_small = v2_min;
        hipLaunchKernelGGL(kern1, dim3(grid_for(S, v2_min)), block, S->lds_bytes, st, S->hdr, S->d_dbl, S->d_int, q_active, qpos_env, (long long)N,
                           (long long)samples_per_env, valid, min_dist, env_idx, n_dev, n_small);
    }
    if (S->use_v2 && (N >= v2_min || ((env_idx || n_dev) && S->v2_forced))) {
        // one lane per state, 64-state tiles; 2 workgroups per CU keep the pose slab small and L2 resident
        int64_t tiles = (N + 63) / 64;
        int64_t blocks = std::min<int64_t>((tiles + kWavesPerBlock - 1) / kWavesPerBlock, (int64_t)S->n_cu * 2);
        size_t waves = (size_t)blocks * kWavesPerBlock;
        StreamScratch &sc = scratch_for(S, st);
        if (waves > sc.slab_waves) {
            const size_t want = std::max(waves, (size_t)S->n_cu * 2 * kWavesPerBlock);
            HIP_TRY(grow(S, sc.slab, (want * (size_t)(S->hdr.nmg + S->hdr.n_save) * kSlabStride + 16) * sizeof(double)));
            HIP_TRY(grow(S, sc.mpr, want * (size_t)kMprCapV5 * kMprRow * sizeof(double)));
            HIP_TRY(grow(S, sc.cen, want * (size_t)S->hdr.nmg * 3 * 64 * sizeof(float)));
            sc.slab_waves = want;
        }
        if (!sc.k1_ctr.p) {
            // zeroed ONCE, synchronously (never inside a stream capture: a first call under capture fails loudly here instead of baking a
            // memset node into the graph).  INVARIANT the kernels keep (tile_ctr_release): every wave of a launch reaches the release, the
            // last one puts [0] and [1] back to zero -- a kernel edit that adds an early return before the release breaks later launches.
            HIP_TRY(grow(S, sc.k1_ctr, 64));
            HIP_TRY(hipMemset(sc.k1_ctr.p, 0, 64));
        }
        unsigned long long *const d_ctr = sc.k1_ctr.as<unsigned long long>();
        double *const d_slab = sc.slab.as<double>();
        dim3 grid((unsigned)blocks);
#ifdef MOPA_V2_PROFILE
        // [6 profile words | 2 pad] live right behind the slabs of this launch's waves
        double *d_tail = d_slab + (size_t)blocks * kWavesPerBlock * (S->hdr.nmg + S->hdr.n_save) * kSlabStride;
        unsigned long long *d_prof = (unsigned long long *)d_tail;
        (void)zero_async(d_prof, 6 * 8, st);
#endif
        auto kern = min_dist ? k_is_valid_v2<true, false> : k_is_valid_v2<false, false>;   // main lists carry no mesh pair
        long long *mesh_list = nullptr;
        unsigned long long *rows_cnt = nullptr;
        double *mesh_rows = nullptr;
        // rows the gate may hand over per launch (typically 1-2 % of the states have one): beyond it a state goes to the state list
        const long long rows_cap = std::getenv("MOPA_MESH_ROWS_CAP") ? atoll(std::getenv("MOPA_MESH_ROWS_CAP")) : std::min<long long>(std::max<long long>(4096, (long long)N / 2), 1ll << 22);      // (at most 1 GiB of rows; beyond: the state list)
        if (S->use_v5 && !S->v5_cen_lds && S->n_mesh_gp > 0) {
            HIP_TRY(grow(S, sc.mesh_list, ((size_t)N + 2) * sizeof(long long)));
            rows_cnt = sc.mesh_list.as<unsigned long long>();
            mesh_list = sc.mesh_list.as<long long>() + 1;
            HIP_TRY(zero_async(rows_cnt, 2 * sizeof(long long), st));
            if (rows_cap > 0) {
                HIP_TRY(grow(S, sc.mesh_rows, (size_t)rows_cap * kMprRow * sizeof(double)));
                mesh_rows = sc.mesh_rows.as<double>();
            }
        }
        // a device-side count stops the main pass at *n_dev; an ungated mesh pass would still walk all N worst-case rows
        // (uninitialised candidates beyond *n_dev): only the gated form (work list built by the main pass) is served
        if (n_dev && S->n_mesh_gp > 0 && !mesh_list)
            return fail(MOPA_ERR_UNSUPPORTED, "a device-side state count on a scene with mesh pairs needs the gated mesh pass");
        if (S->use_v5) {
            auto k5 = S->v5_cen_lds ? (min_dist ? k_is_valid_v5<true, true, false> : k_is_valid_v5<false, true, false>)
                      : mesh_list   ? (min_dist ? k_is_valid_v5<true, false, true> : k_is_valid_v5<false, false, true>)
                                    : (min_dist ? k_is_valid_v5<true, false, false> : k_is_valid_v5<false, false, false>);
            SceneHdr hk = S->hdr;
            if (min_dist) hk.v5_ent_cap = S->v5_ent_cap_md;
            hipLaunchKernelGGL(k5, grid, block, min_dist ? S->v5_lds_bytes_md : S->v5_lds_bytes, st, hk, S->d_dbl, S->d_int, S->d_gp_tab, q_active, qpos_env,
                               (long long)N, (long long)samples_per_env, valid, min_dist, d_slab, env_idx, sc.mpr.as<double>(), mesh_list, sc.cen.as<float>(),
                               n_dev, d_ctr, mesh_list ? mesh_rows : nullptr, rows_cap, rows_cnt, n_small);
            if (mesh_list && mesh_rows) {
                auto kr = min_dist ? k_mesh_rows<true> : k_mesh_rows<false>;
                // (159 registers: three waves per SIMD -- the rows are latency chains, so all the slots are offered; idle waves leave at once)
                static const int rows_blocks = std::getenv("MOPA_MESH_ROWS_BLOCKS") ? atoi(std::getenv("MOPA_MESH_ROWS_BLOCKS")) : 0;
                hipLaunchKernelGGL(kr, dim3((unsigned)(rows_blocks > 0 ? rows_blocks : 3 * S->n_cu)), block, (size_t)S->n_mesh_dbl * sizeof(double), st, S->hdr, S->d_dbl, (const double *)mesh_rows,
                                   (const unsigned long long *)rows_cnt, rows_cap, S->n_mesh_dbl, valid, min_dist);
            }
        } else
        hipLaunchKernelGGL(kern, grid, block, S->v2_lds_bytes, st, S->hdr, S->d_dbl, S->d_int, q_active, qpos_env, (long long)N,
                           (long long)samples_per_env, valid, min_dist, d_slab, 0, env_idx, (const long long *)nullptr, d_ctr);
        if (S->n_mesh_gp > 0) {
            // second pass: the mesh pairs only (MESH instantiation), verdict AND-ed / depth min-ed into the first pass's
            auto km = min_dist ? k_is_valid_v2<true, true> : k_is_valid_v2<false, true>;
            // (the MESH instantiation holds one wave per SIMD: n_cu workgroups are all that run at once, and the gated pass sizes its tiles by
            //  the waves of the launch -- a second round of workgroups would only find the counter exhausted)
            // (with the row list in front of it the gated pass only serves the states whose rows did not fit: a quarter of the CUs)
            static const int fb_blocks = std::getenv("MOPA_MESH_FALLBACK_BLOCKS") ? atoi(std::getenv("MOPA_MESH_FALLBACK_BLOCKS")) : 0;
            const unsigned fb = fb_blocks > 0 ? (unsigned)fb_blocks : (mesh_rows ? (unsigned)std::max(1, S->n_cu / 4) : (unsigned)S->n_cu);
            const dim3 grid_m(mesh_list ? std::min<unsigned>(grid.x, fb) : grid.x);
            hipLaunchKernelGGL(km, grid_m, block, S->v2_lds_bytes, st, S->hdr_mesh, S->d_dbl, S->d_int, q_active, qpos_env, (long long)N,
                               (long long)samples_per_env, valid, min_dist, d_slab, 1, env_idx, (const long long *)mesh_list, d_ctr);
        }
        HIP_TRY(hipGetLastError());
        if (mesh_list && std::getenv("MOPA_DEBUG_MESH")) {     // diagnostics: how many states the gate lets through
            long long cnt[2] = {0, 0};
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(cnt, mesh_list - 1, sizeof(cnt), hipMemcpyDeviceToHost);
            fprintf(stderr, "[mopa] mesh gate: %lld rows handed to k_mesh_rows (cap %lld), %lld of %lld states go to the second pass\n", cnt[0], rows_cap, cnt[1], (long long)N);
        }
#ifdef MOPA_V2_PROFILE
        {
            unsigned long long hp[6];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hp, d_prof, 48, hipMemcpyDeviceToHost);
            if (hp[5]) fprintf(stderr, "[v2 profile] per tile cycles: body %.0f geom %.0f cull+push %.0f flush %.0f total %.0f (tiles %llu)\n",
                               (double)hp[0] / hp[5], (double)hp[1] / hp[5], (double)hp[2] / hp[5], (double)hp[3] / hp[5], (double)hp[4] / hp[5], hp[5]);
        }
#endif
        return MOPA_OK;
    }
    dim3 grid(grid_for(S, N));
    hipLaunchKernelGGL(kern1, grid, block, S->lds_bytes, st, S->hdr, S->d_dbl, S->d_int, q_active, qpos_env, (long long)N,
                       (long long)samples_per_env, valid, min_dist, env_idx, n_dev, (long long)(1ll << 62));
    HIP_TRY(hipGetLastError());
    return MOPA_OK;
}

extern "C" int mopa_is_valid_batch(MopaScene *S, const double *q_active, const double *qpos_env, int64_t N,
                                   int64_t samples_per_env, uint8_t *valid, double *min_dist, void *stream) {
    return launch_is_valid(S, q_active, qpos_env, N, samples_per_env, nullptr, valid, min_dist, stream);
}

#include "mopa_motion.inc"

extern "C" int mopa_check_motion_batch(MopaScene *S, const double *qa, const double *qb, const double *qpos_env, int64_t N,
                                       int64_t samples_per_env, uint8_t *valid, void *stream) {
    if (!S || !valid || (N > 0 && (!qa || !qb || !qpos_env))) return fail(MOPA_ERR_INVALID_ARG, "null argument");
    if (N < 0 || samples_per_env <= 0) return fail(MOPA_ERR_INVALID_ARG, "N < 0 or samples_per_env <= 0");
    if (N == 0) return MOPA_OK;
    ON_DEVICE(S->device);
    hipStream_t st = (hipStream_t)stream;
    // large batches: expand every segment into its states, validate them with the lane-per-state kernel, AND per segment
    const int64_t big = S->v2_forced ? 64 : std::max<int64_t>(64, (int64_t)S->n_cu * 16);
    if (S->use_v2 && N >= big) return motion_expanded(S, qa, qb, qpos_env, N, samples_per_env, valid, st);
    dim3 grid(grid_for(S, N)), block(kBlock);
    const int hdr_off = (S->lds_bytes + 15) & ~15;
    if (hdr_off + (int)sizeof(SceneHdr) > kMaxLdsBytes) return fail(MOPA_ERR_LIMIT, "motion validation LDS does not fit");
    hipLaunchKernelGGL(k_check_motion, grid, block, hdr_off + sizeof(SceneHdr), st, S->hdr, S->d_dbl, S->d_int, qa, qb, qpos_env,
                       (long long)N, (long long)samples_per_env, valid, hdr_off);
    HIP_TRY(hipGetLastError());
    return MOPA_OK;
}

// split a full qpos into (active vector, env row) on the scene's scratch
static int upload_state(MopaScene *S, const double *qpos_host) {
    std::vector<double> buf(S->nq + S->na);
    std::memcpy(buf.data(), qpos_host, sizeof(double) * S->nq);
    for (int a = 0; a < S->na; a++) buf[S->nq + a] = qpos_host[S->active_idx[a]];
    HIP_TRY(hipMemcpy(S->d_q, buf.data(), sizeof(double) * buf.size(), hipMemcpyHostToDevice));
    return MOPA_OK;
}

extern "C" int mopa_is_valid_state(MopaScene *S, const double *qpos_host, int32_t *valid_out, double *min_dist_out) {
    if (!S || !qpos_host || !valid_out) return fail(MOPA_ERR_INVALID_ARG, "null argument");
    ON_DEVICE(S->device);
    int rc = upload_state(S, qpos_host);
    if (rc) return rc;
    rc = mopa_is_valid_batch(S, S->d_q + S->nq, S->d_q, 1, 1, S->d_valid, min_dist_out ? S->d_md : nullptr, nullptr);
    if (rc) return rc;
    uint8_t v = 0;
    HIP_TRY(hipMemcpy(&v, S->d_valid, 1, hipMemcpyDeviceToHost));
    if (min_dist_out) HIP_TRY(hipMemcpy(min_dist_out, S->d_md, 8, hipMemcpyDeviceToHost));
    *valid_out = v;
    return MOPA_OK;
}

static int run_debug(MopaScene *S, const double *qpos_host) {
    ON_DEVICE(S->device);
    int rc = upload_state(S, qpos_host);
    if (rc) return rc;
    hipLaunchKernelGGL(S->hdr.has_mesh ? k_debug_state<true> : k_debug_state<false>, dim3(1), dim3(kBlock), S->lds_bytes, nullptr, S->hdr, S->d_dbl, S->d_int, S->d_q + S->nq,
                       S->d_q, S->d_dbg, S->d_dbg + (size_t)kGeomStride * S->hdr.ng);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return MOPA_OK;
}

extern "C" int mopa_debug_fk(MopaScene *S, const double *qpos_host, double *gpos, double *gmat) {
    if (!S || !qpos_host || !gpos || !gmat) return fail(MOPA_ERR_INVALID_ARG, "null argument");
    int rc = run_debug(S, qpos_host);
    if (rc) return rc;
    ON_DEVICE(S->device);
    std::vector<double> rec((size_t)kGeomStride * S->hdr.ng);
    HIP_TRY(hipMemcpy(rec.data(), S->d_dbg, rec.size() * 8, hipMemcpyDeviceToHost));
    for (int g = 0; g < S->hdr.ng; g++) {
        std::memcpy(gpos + 3 * g, &rec[(size_t)kGeomStride * g + GO_POS], 24);
        std::memcpy(gmat + 9 * g, &rec[(size_t)kGeomStride * g + GO_MAT], 72);
    }
    return MOPA_OK;
}

extern "C" int mopa_debug_pair_dist(MopaScene *S, const double *qpos_host, double *dist) {
    if (!S || !qpos_host || !dist) return fail(MOPA_ERR_INVALID_ARG, "null argument");
    int rc = run_debug(S, qpos_host);
    if (rc) return rc;
    ON_DEVICE(S->device);
    std::vector<double> d(S->hdr.npair);
    if (!d.empty()) HIP_TRY(hipMemcpy(d.data(), S->d_dbg + (size_t)kGeomStride * S->hdr.ng, d.size() * 8, hipMemcpyDeviceToHost));
    for (int p = 0; p < S->npair_model; p++) dist[p] = (S->pair_slot[p] >= 0) ? d[S->pair_slot[p]] : MOPA_FAR;
    return MOPA_OK;
}

extern "C" const char *mopa_planner_status(const MopaScene *S) { return S ? S->status.c_str() : "none"; }

// The planner entry points are defined in mopa_planner.inc (K3).
#include "mopa_planner.inc"
#include "mopa_pullback.inc"
#include "mopa_ik.inc"
#include "mopa_paths.inc"
