/*
 * mopa_oracle.h -- CPU oracle for the MoPA-RL state-validity hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of the reference path lives in MuJoCo 2.0
 * (closed-source libmujoco200.so) and OMPL (unpinned git HEAD); neither is in
 * /root/reference nor in this image, and the reference ships no tests or
 * golden vectors.  This file restates the published
 * algorithms those libraries implement for the functions the reference calls
 * (file:line citations on each function in mopa_oracle.c) and is pinned only
 * against closed-form geometry, independent numpy/scipy re-derivations and
 * self-consistency invariants (tests/test_oracle_*.py).
 */
#ifndef MOPA_ORACLE_H
#define MOPA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* "no intersection found" sentinel returned by pair distance functions that
 * only resolve penetration (SAT / MPR pairs) and by min_dist when no pair
 * survives the broad phase. */
#define ORC_FAR 1.0e10

typedef struct OrcScene OrcScene;

/* Flat scene description == mopa_rl_amd.mjcf.CompiledModel arrays. */
OrcScene *orc_scene_create(
    int nq, int nbody, const int32_t *body_parent, const double *body_pos, const double *body_quat,
    const int32_t *body_jntadr, const int32_t *body_jntnum,
    int njnt, const int32_t *jnt_type, const int32_t *jnt_qposadr, const double *jnt_axis,
    const double *jnt_pos, const double *jnt_ref, const int32_t *jnt_limited, const double *jnt_range,
    int ngeom, const int32_t *geom_type, const int32_t *geom_body, const int32_t *geom_mjid,
    const double *geom_size, const double *geom_pos, const double *geom_quat,
    int npair, const int32_t *pair_geom,
    int n_passive, const int32_t *passive_qpos_idx,
    int n_ignored, const int32_t *ignored_pairs /* [n,2] MuJoCo geom ids, ordered */,
    double contact_threshold);
/* optional: convex hulls of mesh geoms (type 7); without it mesh pairs report ORC_FAR */
void orc_scene_set_meshes(OrcScene *s, int nmesh, const int32_t *mesh_vertadr, const int32_t *mesh_vertnum, int nmeshvert,
                          const double *mesh_vert /*[nmeshvert,3]*/, const int32_t *geom_dataid /*[ngeom]*/);
void orc_scene_destroy(OrcScene *s);
int orc_num_active(const OrcScene *s);
void orc_active_idx(const OrcScene *s, int32_t *out);

/* deterministic sin/cos shared (as a specification) with the HIP kernels */
void orc_sincos(double x, double *s, double *c);

/* FK: qpos[nq] -> world pose of every collidable geom */
void orc_fk(const OrcScene *s, const double *qpos, double *geom_xpos /*[ngeom,3]*/, double *geom_xmat /*[ngeom,9]*/);
/* FK of every body: xpos[nbody,3], xquat[nbody,4] */
void orc_fk_bodies(const OrcScene *s, const double *qpos, double *xpos, double *xquat);

/* signed distance of two posed primitives (types ordered t1<=t2) */
/* the separating-axis pre-test of the cylinder classes (an accelerator, on by default): 0 = every such pair goes to MPR */
void orc_set_convex_axes_pretest(int on);
double orc_geom_dist(int t1, const double *size1, const double *pos1, const double *mat1,
                     int t2, const double *size2, const double *pos2, const double *mat2);

/* primitive (t1) vs a convex mesh given by its hull vertices in the mesh frame */
double orc_geom_dist_mesh(int t1, const double *size1, const double *pos1, const double *mat1,
                          const double *verts, int nvert, const double *pos2, const double *mat2);

/* per-pair distances for one state; culled pairs get ORC_FAR */
void orc_pair_dist(const OrcScene *s, const double *qpos, double *dist /*[npair]*/);

/* the validity rule; returns 1 valid / 0 invalid; *min_dist = min(0, min signed distance over the non-ignored
 * pairs) = deepest penetration (0 when nothing penetrates) */
int orc_is_valid(const OrcScene *s, const double *qpos, double *min_dist);

/* batch: sample i uses qpos_env[env_of(i)] with active entries replaced by q_active[i] */
void orc_is_valid_batch(const OrcScene *s, const double *q_active /*[N,na]*/, const double *qpos_env /*[E,nq]*/,
                        int64_t N, int64_t samples_per_env, uint8_t *valid, double *min_dist /*nullable*/,
                        int nthreads);

/* OMPL DiscreteMotionValidator restated: 1 = motion valid */
int orc_check_motion(const OrcScene *s, const double *qpos_env, const double *qa, const double *qb,
                     double resolution, int64_t *n_checks);
void orc_check_motion_batch(const OrcScene *s, const double *qa /*[N,na]*/, const double *qb /*[N,na]*/,
                            const double *qpos_env /*[E,nq]*/, int64_t N, int64_t samples_per_env,
                            double resolution, uint8_t *valid, int nthreads);

/* counter-based RNG shared (as a specification) with the HIP planner */
uint64_t orc_rng_u64(uint64_t seed, uint64_t stream, uint64_t counter);
double orc_rng_uniform(uint64_t seed, uint64_t stream, uint64_t counter);

/* RRT-Connect restated; returns status 0 ok / -5 invalid goal / -4 no exact solution.
 * path: [max_path, nq] rows (passive columns copied from start). */
int orc_plan(const OrcScene *s, const double *start /*[nq]*/, const double *goal /*[nq]*/,
             double range, double resolution, int max_iters, int max_nodes, uint64_t seed, uint64_t env_id,
             double *path, int max_path, int *path_len, int64_t *n_checks, int *n_iters);

/* E queries, OpenMP over queries (dynamic schedule: a query that runs out its budget takes 100x a quick one); env ids
 * env_id_base + e.  path [E, max_path, nq] may be NULL (status / length / check counts only). */
void orc_plan_batch(const OrcScene *s, int64_t E, const double *start /*[E,nq]*/, const double *goal /*[E,nq]*/, double range,
                    double resolution, int max_iters, int max_nodes, uint64_t seed, uint64_t env_id_base, double *path, int max_path,
                    int32_t *status /*[E]*/, int32_t *path_len /*[E]*/, int64_t *n_checks /*[E]*/, int nthreads);

/* deterministic exp / tanh(x>=0) shared (as a specification) with the HIP env kernel */
double orc_exp(double x);
double orc_tanh_pos(double x);

/*  kinematic env.step restated -- see the comment on orc_env_step in mopa_oracle.c.
 * Same fields as MopaEnvDesc (include/mopa_hip.h) without the model. */
typedef struct OrcEnvDesc {
    int32_t kind;                                   /* 0 push, 1 lift, 2 assembly (Sawyer), 3 PusherObstacle */
    int32_t n_arm; const int32_t *arm_qpos_idx;
    int32_t n_grip; const int32_t *grip_qpos_idx;
    int32_t n_act; const int32_t *act_qpos_idx; const double *act_lo, *act_hi;   /* position actuators, ctrl order (arm first) */
    int32_t n_frames; const int32_t *frame_body; const double *frame_off;        /* [n_frames], [n_frames,3] */
    int32_t n_quats; const int32_t *quat_body;
    int32_t n_touch; const int32_t *touch_geom; int32_t n_touch_left;            /* lift: [object, left-finger geoms..., right-finger geoms...] */
    const double *qpos_min, *qpos_max; const int32_t *qpos_limited;
    double ac_scale, distance_threshold, success_reward;
    int32_t max_episode_steps;
} OrcEnvDesc;
int orc_env_obs_dim(const OrcEnvDesc *d);
int orc_env_action_dim(const OrcEnvDesc *d);
/* one env, in place; action == NULL: obs only */
void orc_env_step(const OrcScene *s, const OrcEnvDesc *d, double *qpos /*[nq]*/, double *prev_state /*[n_arm]*/,
                  uint8_t *has_prev, int32_t *ep_len, const double *action /*[action_dim] or NULL*/, int is_planner, int move,
                  double *obs /*[obs_dim]*/, double *reward, uint8_t *done, uint8_t *success);
void orc_env_step_batch(const OrcScene *s, const OrcEnvDesc *d, int64_t E, double *qpos, double *prev_state, uint8_t *has_prev,
                        int32_t *ep_len, const double *action, int is_planner, const uint8_t *move_mask, double *obs,
                        double *reward, uint8_t *done, uint8_t *success, int nthreads);


/* (stage A) contact-free servo dynamics of the actuated kinematic tree -- see mopa_oracle_dyn.inc.
 * The tree is given LUMPED: one body per dof (bodies welded to it folded into its inertia), parents before children. */
#define ORC_DYN_MAX 12
/* (stage B, Push) the manipulated object as a free body with penalty contacts -- see mopa_oracle_dyn.inc */
typedef struct OrcObjDesc {
    int32_t qadr;                     /* qpos address of the object's free joint (pos 3, quat 4); its COM is the body origin */
    double mass, inertia[3], damping; /* principal inertia (body frame = principal frame), free-joint damping */
    double half[3], rbound;           /* box half extents, bounding radius */
    int32_t nfeat; const double *feat;   /* [nfeat,3] feature points in the body frame; the first 8 are the vertices */
    int32_t ncol;                     /* colliders: static ones first (body -1, pose in the world), then robot geoms by dynamic body */
    const int32_t *co_body, *co_type;
    const double *co_size /*[n,3]*/, *co_pos /*[n,3]*/, *co_mat /*[n,9]*/, *co_mu /*[n]*/, *co_rbound /*[n]*/;
    double inv_mass, inv_inertia[3];  /* reciprocals, formed once on the host (the integrator multiplies) */
    int32_t precull_every;            /* the full collider scan runs on every precull_every-th sub-step of a call ... */
    double precull_margin;            /* ... with the bounding spheres inflated by this much [m] */
    double kn, dn, eps_v, ct_max;     /* normal stiffness [N/m], normal damping [N s/m], friction regularisation [m/s], cap of the
                                         friction force's viscous coefficient [N s/m] (explicit-step stability) */
} OrcObjDesc;
typedef struct OrcDynDesc {
    int32_t nd;
    const int32_t *parent;                  /* [nd] parent dynamic body, -1 = the fixed base */
    const int32_t *jtype;                   /* [nd] 2 slide, 3 hinge (mjtJoint) */
    const int32_t *qadr;                    /* [nd] qpos address of the dof */
    const double *rel_pos, *rel_quat;       /* [nd,3] [nd,4] body frame in its parent dynamic body's frame (base: world) */
    const double *axis, *jpos, *qref;       /* [nd,3] [nd,3] [nd]  joint axis / anchor in the body frame, reference value */
    const double *mass, *ipos, *inertia;    /* [nd] [nd,3] [nd,6]  lumped; inertia about the COM in body axes: xx yy zz xy xz yz */
    const double *damping, *armature;       /* [nd] */
    const int32_t *limited; const double *lo, *hi;   /* [nd] joint range (inelastic stop) */
    const int32_t *actuated;                /* [nd] a position servo drives this dof: force = clamp(kp ctrl - kp q, force range) */
    const double *kp, *force_lo, *force_hi; /* [nd] */
    const int32_t *gravcomp;                /* [nd] qfrc_applied = the qfrc_bias of the previous mj_forward (the env's gravity compensation) */
    double gravity[3], timestep;
    int32_t nsub;                           /* sub-steps per env.step (frame_dt / timestep) */
    const OrcObjDesc *obj;                  /* NULL: stage A (nothing but the robot moves) */
    const struct OrcCtDesc *ct;             /* stage C: contacts + constraint solver (mopa_oracle_contact.inc); excludes obj */
} OrcDynDesc;

/* (stage C) contacts of the arm, of the manipulated object and between the two, resolved by a soft-constraint
 * solver restated from MuJoCo's published formulation -- see mopa_oracle_contact.inc.  PARITY UNPINNED.
 * Bodies: 0 .. nd-1 the lumped dynamic bodies of the arm, nd the manipulated object (free body), -1 the world.
 * Shapes: the collidable geoms (plane / sphere / capsule / cylinder / box; a mesh enters as its bounding cylinder), posed in
 * the frame of their body.  Features: points (with a radius: sphere-swept) sampled on a shape, in the same body frame.
 * Directed pairs (F, S): the features of shape F are tested against the signed-distance function of shape S. */
#define ORC_CT_MAXCON 16
typedef struct OrcCtDesc {
    int32_t ns;
    const int32_t *sh_body, *sh_type;            /* [ns] */
    const double *sh_size, *sh_pos, *sh_mat, *sh_rbound;   /* [ns,3] [ns,3] [ns,9] [ns] */
    const int32_t *sh_feat0;                     /* [ns + 1] features of shape s: sh_feat0[s] .. sh_feat0[s + 1] - 1 */
    int32_t nf;
    const double *ft_pos, *ft_rad;               /* [nf,3] (body frame of the owning shape) [nf] */
    int32_t np;
    const int32_t *pr_f, *pr_s;                  /* [np] */
    const double *pr_par;                        /* [np,12]: mu, margin, K, B, d0, dmax, width, - (solref / solimp mixed per pair), condim (3 / 4 / 6),
                                                    torsional friction, rolling friction, - (solver 2; the others solve every pair as condim 3) */
    int32_t obj_qadr;                            /* qpos address of the object's free joint; -1: no object */
    double obj_mass, obj_inertia[3], obj_ipos[3], obj_iquat[4], obj_damping;   /* principal inertia at the COM (ipos, iquat in the body frame) */
    double obj_inv_mass, obj_inv_inertia[3];     /* reciprocals for the constraint stage (M^-1) */
    double obj_inv_mass_d, obj_inv_inertia_d[3]; /* 1 / (M + h damping) for the integration */
    int32_t maxcon, maxpair;                     /* contacts kept per env and sub-step (<= ORC_CT_MAXCON) / per directed pair */
    int32_t iterations;                          /* PGS sweeps at most (XML: iterations="50") */
    double tolerance, inv_scale;                 /* stop when improvement * inv_scale < tolerance; inv_scale = 1 / (meaninertia max(1, nv)) */
    int32_t precull_every; double precull_margin;
    int32_t near_every; double near_margin;      /* third culling level: the active pairs within this of contact, re-listed every near_every sub-steps */
    int32_t warmstart;                           /* carry the pyramid forces of persisting contacts into the next sub-step */
    int32_t solver;                              /* 0: projected Gauss-Seidel (pyramidal cones), 1: Newton, pyramidal cones, 2: Newton, ELLIPTIC cones with the
                                                    pairs' condim (3, 4: + torsional, 6: + rolling friction) -- the XML's model */
    int32_t limit_rows;                          /* joint limits as rows of the (Newton) solver instead of an inelastic stop */
    double lim_par[8];                           /* their parameters in a pair record's layout: -, margin 0, K, B, d0, dmax, width, - */
    int32_t noslip_iterations;                   /* sweeps of the noslip pass after the main solve (XML: 5; 0 = none) */
    double noslip_tolerance;                     /* its early exit: improvement * inv_scale below this (MuJoCo default 1e-6) */
    int32_t arena;                               /* solver 2: doubles of the kernel's per-env LDS arena the contact records share (a contact whose record
                                                    does not fit is dropped, as one beyond maxcon is); 0 = unlimited */
} OrcCtDesc;
typedef struct OrcCtStats { int64_t substeps, contacts, sweeps, dropped; int32_t max_contacts; int64_t hot_pairs, active_pairs; } OrcCtStats;
/* n sub-steps with contacts; qvel [nd + 6]: dofs, then the object's (v of its COM, w) in the world; stats may be NULL (accumulated) */
int orc_ct_desc_size(void);      /* sizeof(OrcCtDesc): binding self-check */
void orc_ct_step(const OrcDynDesc *d, double *qpos, double *qvel, double *bias_lag, const double *ctrl, int n, OrcCtStats *stats);
/* the contacts of one configuration (no step): rows of [dist, pos 3, normal 3, shape F, shape S, feature] -> out [maxcon,10]; returns the count */
int orc_ct_contacts(const OrcDynDesc *d, const double *qpos, double *out);
/* qfrc_bias [nd] (RNE with qacc = 0, gravity included) and, if M != NULL, the joint-space inertia [nd,nd] (CRB + armature) */
void orc_dyn_forward(const OrcDynDesc *d, const double *qpos, const double *qvel /*[nd]*/, double *bias, double *M);
/* n sub-steps of mj_step towards ctrl [nd] (already ctrl-range clamped; entries of unactuated dofs ignored); in place */
void orc_dyn_step(const OrcDynDesc *d, double *qpos /*[nq]*/, double *qvel /*[nd]*/, double *bias_lag /*[nd]*/,
                  const double *ctrl /*[nd]*/, int n);
/* the same with the object's velocity state obj_vel [6] = (v world, w world); NULL or d->obj == NULL: the object rests */
void orc_dyn_step_obj(const OrcDynDesc *d, double *qpos, double *qvel, double *bias_lag, const double *ctrl, int n, double *obj_vel);
/* qvel rows are [nd (+ 6 when dyn->obj: the object's velocity)] */
void orc_env_step_dyn(const OrcScene *s, const OrcEnvDesc *d, const OrcDynDesc *dyn, double *qpos, double *qvel, double *bias_lag,
                      double *prev_state, uint8_t *has_prev, int32_t *ep_len, const double *action, int is_planner, int move,
                      double *obs, double *reward, uint8_t *done, uint8_t *success);
void orc_env_step_dyn_batch(const OrcScene *s, const OrcEnvDesc *d, const OrcDynDesc *dyn, int64_t E, double *qpos, double *qvel,
                            double *bias_lag, double *prev_state, uint8_t *has_prev, int32_t *ep_len, const double *action,
                            int is_planner, const uint8_t *move_mask, double *obs, double *reward, uint8_t *done,
                            uint8_t *success, int nthreads);

/*  damped-LS IK of a site pose, one env, in place on qpos -- see mopa_oracle.c.
 * site_quat: orientation of the site in its body's frame (NULL = identity); target_quat NULL = position target only. */
void orc_ik_solve(const OrcScene *s, int n_joints, const int32_t *joint_ids /*model joint ids, <= 8*/, int site_body,
                  const double *site_off /*[3]*/, const double *site_quat /*[4] or NULL*/, double *qpos /*[nq] in/out*/,
                  const double *target_pos /*[3]*/, const double *target_quat /*[4] wxyz or NULL*/, double rot_weight, int max_steps,
                  double tol, double max_update_norm, double progress_thresh, double reg_strength, double *err_norm_out,
                  int32_t *steps_out, uint8_t *success_out);
/* deterministic atan2 shared (as a specification) with the HIP IK kernel */
double orc_atan2(double y, double x);

#ifdef __cplusplus
}
#endif
#endif
