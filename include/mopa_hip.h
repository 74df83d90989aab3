/*
 * mopa_hip.h -- C ABI of libmopa_hip.so, the MI355X (gfx950) implementation of
 * MoPA-RL's state-validity / motion-validation / RRT-Connect hot path.
 *
 * This is the drop-in boundary: each entry point below replaces one piece of
 * what the reference's Cython extension binds (reference
 * motion_planners/planner.pyx:9-52 -> motion_planners/include/KinematicPlanner.h:40-71).
 * Plain pointers and sizes only; no C++ types, no exceptions: every function
 * returns an int status (0 = MOPA_OK) and mopa_last_error() explains failures.
 *
 *   reference interface                              replaced by
 *   -----------------------------------------------  ----------------------------
 *   KinematicPlanner::KinematicPlanner(xml, algo,    mopa_scene_create()
 *     ..., passive_joint_idx, glue_bodies,           (the XML is compiled to the
 *     ignored_contacts, contact_threshold, ...)      flat MopaModel by the host
 *     KinematicPlanner.cpp:42-120                    shim: mopa_rl_amd/mjcf.py)
 *   KinematicPlanner::~KinematicPlanner              mopa_scene_destroy()
 *   KinematicPlanner::isValidState(vector<double>)   mopa_is_valid_state()      (1 state, host ptr)
 *     KinematicPlanner.cpp:253-286 ->                mopa_is_valid_batch()      (N states, device ptrs)
 *     MujocoStateValidityChecker::isValid
 *     mujoco_ompl_interface.cpp:909-978
 *   si->checkMotion (OMPL DiscreteMotionValidator,   mopa_check_motion_batch()
 *     resolution KinematicPlanner.cpp:87)
 *   KinematicPlanner::plan(start, goal, timelimit)   mopa_plan()                (1 query, host ptrs)
 *     KinematicPlanner.cpp:125-251 (RRTConnect)      mopa_plan_batch()          (E queries, device ptrs)
 *   invalid-target back-off of the rollout runner    mopa_pullback_batch()      (E targets, device ptrs)
 *     rl/mopa_rollouts.py:133-143
 *   KinematicPlanner::getPlannerStatus()             mopa_planner_status()
 *     KinematicPlanner.cpp:288-291
 *
 * Memory: "dev" pointers are HIP device pointers (e.g. torch tensor
 * data_ptr() on a ROCm device), "host" pointers are ordinary memory.  The
 * caller owns every buffer; the library never frees caller memory.  All
 * floating point data is IEEE double, C-contiguous.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream); batch calls are
 * asynchronous on that stream.  Streams: a scene keeps its launch scratch per
 * stream, so calls on DIFFERENT streams may be in flight at the same time (e.g.
 * validity on one stream while the planner runs on another); calls on the same
 * stream are ordered by the stream.  Host threads: one at a time per object for
 * the single-query forms (they share a staging buffer); the batch forms may be
 * called from several threads as long as each uses its own stream.  Devices:
 * every entry point runs on its object's device and restores the caller's
 * current device before returning.
 */
#ifndef MOPA_HIP_H
#define MOPA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOPA_OK 0
#define MOPA_ERR_INVALID_ARG 1
#define MOPA_ERR_UNSUPPORTED 2   /* e.g. ellipsoid / height-field geom, ball joint in the model */
#define MOPA_ERR_HIP 3           /* HIP runtime error (no device, launch failure...) */
#define MOPA_ERR_LIMIT 4         /* model exceeds a compile-time capacity */

/* planner result codes == the sentinel rows of KinematicPlanner.cpp:181-184,249-250 */
#define MOPA_PLAN_OK 0
#define MOPA_PLAN_NO_EXACT (-4)
#define MOPA_PLAN_INVALID_GOAL (-5)

/* "culled / ignored / no intersection found" value of mopa_debug_pair_dist outputs */
#define MOPA_FAR 1.0e10

/* geom / joint type codes (MuJoCo's mjtGeom / mjtJoint values) */
enum { MOPA_GEOM_PLANE = 0, MOPA_GEOM_SPHERE = 2, MOPA_GEOM_CAPSULE = 3, MOPA_GEOM_CYLINDER = 5, MOPA_GEOM_BOX = 6, MOPA_GEOM_MESH = 7 };
enum { MOPA_JNT_FREE = 0, MOPA_JNT_BALL = 1, MOPA_JNT_SLIDE = 2, MOPA_JNT_HINGE = 3 };

/* Flat compiled model (host pointers; copied by mopa_scene_create).
 * Same arrays as mopa_rl_amd.mjcf.CompiledModel. */
typedef struct MopaModel {
    int32_t nq, nbody, njnt, ngeom, npair;
    const int32_t *body_parent;   /* [nbody]            */
    const double  *body_pos;      /* [nbody,3]          */
    const double  *body_quat;     /* [nbody,4] wxyz, normalised */
    const int32_t *body_jntadr;   /* [nbody] first joint or -1 */
    const int32_t *body_jntnum;   /* [nbody]            */
    const int32_t *jnt_type;      /* [njnt]             */
    const int32_t *jnt_qposadr;   /* [njnt]             */
    const double  *jnt_axis;      /* [njnt,3] normalised */
    const double  *jnt_pos;       /* [njnt,3]           */
    const double  *jnt_ref;       /* [njnt] qpos0 of hinge/slide */
    const int32_t *jnt_limited;   /* [njnt]             */
    const double  *jnt_range;     /* [njnt,2]           */
    const int32_t *geom_type;     /* [ngeom] collidable geoms only */
    const int32_t *geom_body;     /* [ngeom]            */
    const int32_t *geom_mjid;     /* [ngeom] id among ALL geoms of the model (for ignored_contacts) */
    const double  *geom_size;     /* [ngeom,3]          */
    const double  *geom_pos;      /* [ngeom,3]          */
    const double  *geom_quat;     /* [ngeom,4]          */
    const int32_t *pair_geom;     /* [npair,2] candidate pairs after MuJoCo's static filters, type1<=type2 */
    /* convex hulls of mesh geoms (MuJoCo collides a mesh geom through the convex hull of its vertices):
     * vertices in the geom frame.  nmesh == 0 / NULL pointers for models without collidable meshes. */
    int32_t nmesh, nmeshvert;
    const int32_t *mesh_vertadr;  /* [nmesh] first vertex */
    const int32_t *mesh_vertnum;  /* [nmesh] */
    const double  *mesh_vert;     /* [nmeshvert,3] */
    const int32_t *geom_dataid;   /* [ngeom] mesh id of a MOPA_GEOM_MESH geom, -1 otherwise (may be NULL when nmesh == 0) */
} MopaModel;

/* Everything KinematicPlanner's constructor receives (KinematicPlanner.cpp:42-61). */
typedef struct MopaSceneDesc {
    MopaModel model;
    int32_t n_passive;
    const int32_t *passive_qpos_idx;   /* [n_passive] qpos addresses NOT planned over */
    int32_t n_ignored;
    const int32_t *ignored_pairs;      /* [n_ignored,2] ordered (lo,hi) MuJoCo geom ids */
    double contact_threshold;          /* state invalid iff some non-ignored dist <= this; must be <= 0 (negative in the
                                          reference); > 0 is rejected with MOPA_ERR_UNSUPPORTED: the broad phase culls at zero margin */
    double range;                      /* RRT-Connect maxDistance (KinematicPlanner.cpp:102-104) */
    double resolution;                 /* validity checking resolution; the reference hard-codes 0.005 (:87) */
    uint64_t seed;                     /* KinematicPlanner.cpp:95 */
    int32_t device;                    /* HIP device ordinal, -1 = current */
    const double *pair_cull_radius;    /* nullable, [model.npair]: > 0 = the pair can only reach contact_threshold while its two geom
                                          centres are within this distance (a compile-time bound, tools/prove_separated_pairs.py);
                                          the FP32 broad phase then culls with min(bounding-sphere sum, this).  0 = no bound */
} MopaSceneDesc;

typedef struct MopaScene MopaScene;

typedef struct MopaPlanParams {
    int32_t max_iters;     /* RRT-Connect iteration budget (replaces the wall-clock timelimit) */
    int32_t max_nodes;     /* per-tree node capacity */
    int32_t max_path;      /* rows available per env in `path` */
    uint64_t seed;         /* sample-stream seed; stream id = env_id_base + env index */
    uint64_t env_id_base;
    const uint64_t *env_ids_dev;   /* mopa_plan_batch only, nullable: explicit stream id per query (device pointer, [E]) --
                                      lets a caller plan for a compacted subset of its envs with the streams of the full set */
    const uint64_t *seeds_dev;     /* mopa_plan_batch only, nullable: explicit seed per query (device pointer, [E]) instead of
                                      `seed` -- queries of envs that are at different steps of their own rollouts in one launch */
    int32_t max_workgroups;        /* mopa_plan_batch only: persistent workgroups of the launch.  0 = one per CU (a lone launch ends
                                      with its slowest query, and two budget-exhausting queries on one SIMD slow each other down);
                                      < 0 = as many as the chip holds at once (two per CU: throughput, for launches that overlap
                                      others -- the iteration ladder); > 0 = an explicit cap (asynchronous rollouts leave the main
                                      stream's kernels room this way: a workgroup keeps ~70 KB of LDS while queries are left) */
    int32_t exclusive_cu;          /* mopa_plan_batch only: nonzero = the launch asks for more than half a CU's LDS, so that no other
                                      planner workgroup (of this or of an overlapping launch) shares its CUs -- for a burst of capped
                                      launches that together want one workgroup per CU.  Implied by max_workgroups == 0 */
    /* mopa_plan_batch only, all nullable (device pointers): continuing queries across launches.  A query's outcome is a
       function of its endpoints and sample stream only and the iteration budget merely ends the loop, so a launch with a
       small budget can leave its unsolved queries where a later launch with a larger budget picks them up -- the result is
       that of one launch with the larger budget, without retracing the first iterations (the iteration ladder).
       Saving: tree_q_dev [E][2][max_nodes][na] doubles (+ 8 of padding), tree_p_dev [E][2][max_nodes] int32 hold the trees
       instead of the library's scratch; state_dev [E][4] int64 receives (iterations done | -1 = "no exact solution"
       whatever the budget | -2 = settled: solved or invalid goal; start-tree size, goal-tree size, consumed checks).
       Continuing: resume_tree_q / resume_tree_p / resume_state = those buffers (rows gathered to this launch's query order,
       same max_nodes, same seeds / stream ids as the first launch); a query whose state says -2 is skipped, its outputs are
       not written -- so a launch's whole query list can be continued right behind it on the same stream, no host in between. */
    double *tree_q_dev;
    int32_t *tree_p_dev;
    int64_t *state_dev;
    const double *resume_tree_q;
    const int32_t *resume_tree_p;
    const int64_t *resume_state;
} MopaPlanParams;

const char *mopa_last_error(void);
const char *mopa_version(void);
/* number of visible HIP devices (0 when none); never fails */
int mopa_device_count(void);

int mopa_scene_create(const MopaSceneDesc *desc, MopaScene **out);
void mopa_scene_destroy(MopaScene *scene);
/* introspection used by the host shim and the tests */
int mopa_scene_num_active(const MopaScene *scene);
int mopa_scene_active_idx(const MopaScene *scene, int32_t *out /*[na]*/);
int mopa_scene_num_pairs(const MopaScene *scene);   /* non-ignored candidate pairs checked per state */
int mopa_scene_lds_bytes(const MopaScene *scene);
/* name of the validity kernel mopa_is_valid_batch dispatches for a batch of N states on this scene ("k_is_valid_v5",
 * "k_is_valid_v2" or "k_is_valid"); the benchmark labels its roofline line and selects profiler rows with it */
int mopa_scene_valid_kernel(const MopaScene *scene, int64_t N, char *out, int32_t cap /* >= 24 */);

/* N states: state i = qpos_env[i / samples_per_env] with its active entries replaced by q_active[i].
 * valid[i] = 1 iff no non-ignored pair has dist <= contact_threshold.
 * min_dist (nullable): deepest penetration = min(0, minimum signed distance over the non-ignored pairs), i.e. 0.0 for
 * a state in which nothing penetrates; requesting it disables the per-state early-out. */
int mopa_is_valid_batch(MopaScene *scene, const double *q_active_dev /*[N,na]*/, const double *qpos_env_dev /*[E,nq]*/,
                        int64_t N, int64_t samples_per_env, uint8_t *valid_dev /*[N]*/, double *min_dist_dev /*[N] or NULL*/,
                        void *stream);

/* N segments qa[i] -> qb[i] (active coordinates), OMPL DiscreteMotionValidator semantics:
 * valid[i] = 1 iff qb[i] and every interior state interpolate(qa,qb,k/nd), k=1..nd-1, is valid,
 * nd = max_j ceil(|d_j| / (resolution * extent_j)).
 * Large batches are served by expanding every segment into its states (count, scan, expand), validating them with the
 * lane-per-state kernel and AND-ing per segment; that path reads the total state count back once, i.e. the call
 * synchronises `stream` before it returns (results are still produced asynchronously on the stream). */
int mopa_check_motion_batch(MopaScene *scene, const double *qa_dev /*[N,na]*/, const double *qb_dev /*[N,na]*/,
                            const double *qpos_env_dev /*[E,nq]*/, int64_t N, int64_t samples_per_env,
                            uint8_t *valid_dev /*[N]*/, void *stream);

/* E independent RRT-Connect queries (one per env).  path rows hold full qpos vectors with the passive
 * columns copied from start (KinematicPlanner.cpp:236-240).  status[e] in {0,-4,-5}. */
int mopa_plan_batch(MopaScene *scene, const double *start_dev /*[E,nq]*/, const double *goal_dev /*[E,nq]*/, int64_t E,
                    const MopaPlanParams *params, double *path_dev /*[E,max_path,nq]*/, int32_t *path_len_dev /*[E]*/,
                    int32_t *status_dev /*[E]*/, int64_t *n_checks_dev /*[E] or NULL*/, void *stream);

/* The rollout's invalid-target back-off (rl/mopa_rollouts.py:133-143) for E envs, asynchronous (no read-back unless E * num_trials
 * rows would exceed 1 GiB of scratch; E < 256: one wave per env walks its trials, otherwise all candidate rows of all
 * invalid targets go through one validity launch -- same results): while target[e] (a full
 * qpos row, validated with its own passive entries) is invalid and fewer than num_trials steps were taken,
 *   target[e] += step_size * (cur[e] - target[e]) / ||cur[e] - target[e]||   (Euclidean norm over all nq entries, squares
 * summed left to right).  target is updated in place; n_trials[e] = steps taken, valid[e] = verdict of the final row. */
int mopa_pullback_batch(MopaScene *scene, const double *cur_dev /*[E,nq]*/, double *target_dev /*[E,nq] in/out*/, int64_t E,
                        double step_size, int32_t num_trials, int32_t *n_trials_dev /*[E]*/, uint8_t *valid_dev /*[E]*/,
                        void *stream);

/* ---- single-query convenience forms: what PyKinematicPlanner binds (host pointers, synchronous) ---- */
int mopa_is_valid_state(MopaScene *scene, const double *qpos_host /*[nq]*/, int32_t *valid_out, double *min_dist_out /*nullable*/);
int mopa_plan(MopaScene *scene, const double *start_host /*[nq]*/, const double *goal_host /*[nq]*/,
              const MopaPlanParams *params, double *path_host /*[max_path,nq]*/, int32_t *path_len_out, int32_t *status_out,
              int64_t *n_checks_out /*nullable*/);
/* OMPL-style status string of the last mopa_plan on this scene ("none" before the first). */
const char *mopa_planner_status(const MopaScene *scene);

/* debug / parity hooks: world pose of every collidable geom for one state (host pointers, synchronous) */
int mopa_debug_fk(MopaScene *scene, const double *qpos_host /*[nq]*/, double *geom_xpos_host /*[ngeom,3]*/,
                  double *geom_xmat_host /*[ngeom,9]*/);
/* per-candidate-pair distances (MOPA_FAR for culled / ignored pairs), in MopaModel.pair_geom order */
int mopa_debug_pair_dist(MopaScene *scene, const double *qpos_host /*[nq]*/, double *dist_host /*[npair]*/);

/* ======================================================================================================
 * (BASELINE configs 3-5) batched KINEMATIC env.step for the three Sawyer obstacle envs --
 * the "env-steps/sec" half of the metric.  Restates what the reference envs compute around the physics:
 *   action scaling / desired joint state     `_step` of env/sawyer/sawyer_{push,lift,assembly}_obstacle.py
 *                                            (Lift: gripper target = gripper qpos + action[-1], sawyer.py:340-342)
 *   joint-limit clamp + episode bookkeeping  env/base.py:269-314
 *   reward / success                         sawyer_push_obstacle.py:71-104, sawyer_lift_obstacle.py:93-150,
 *                                            sawyer_assembly_obstacle.py:33-52
 *   observation (dict order)                 env/sawyer/sawyer.py:317-338 + the env's `_get_obs`
 * and REPLACES the physics (`_do_simulation`, 75 MuJoCo sub-steps of position servos) by its kinematic
 * limit: every actuated joint reaches its target clamped to the actuator's ctrlrange, velocities are zero,
 * nothing else moves (no contact forces: the manipulated object never moves).  NOT dynamics parity -- labelled as
 * such wherever it is reported.
 *
 * Frame / quat slots per kind (positions = world position of a point fixed in a body; quats = body orientations):
 *   all       frame 0 = site "grip_site";                           quat 0 = body "right_ee_attchment"
 *   PUSH      frames 1,2 = sites "right_eef","left_eef", 3 = body "cube", 4 = body "target";   quat 1 = "cube"
 *             obs[40]: common 25 + target_pos, cube_pos, cube_quat(xyzw), gripper_to_cube, cube_to_target
 *   LIFT      frames 1 = body "cube" (the can), 2 = body "bin1";                                 quat 1 = "cube"
 *             obs[35]: common 25 + cube_pos, cube_quat(xyzw), gripper_to_cube;  action[8] = 7 arm + gripper
 *             grasp test (`has_grasp`): touch geoms [can mesh, left-finger boxes..., right-finger boxes...]
 *   ASSEMBLY  frames 1..4 = sites "hole","hole_bottom","pegHead","pegEnd";                     quat 1 = "peg"
 *             obs[38]: common 25 + hole, pegHead, pegEnd, peg_quat(wxyz)
 *   common 25 = joint_pos 7, joint_vel 7 (0), gripper_qpos 2, gripper_qvel 2 (0), eef_pos 3, eef_quat 4 (xyzw)
 *   PUSHER    PusherObstacle-v0 (env/pusher/pusher_obstacle.py:183-274; BASELINE config 1's env, joint0 unlimited): four hinges under
 *             torque motors driven by the env's PID loop (env/base.py:200-209); kinematic limit: the joints reach
 *             desired_state = prev + action -- the action UNSCALED and UNCLIPPED, as the reference's `_step` leaves it (:262-266);
 *             the joint-limit clamp comes after reward / obs.  frames 0 = site "fingertip", 1 = body "fingertip", 2 = body "box",
 *             3 = body "target"; quats = "fingertip", "box" (unused); n_grip 0.
 *             obs[20]: cos(theta) 4, sin(theta) 4, box qpos 2 (= qpos[nq-2:]), joint vel 4 (0), box vel 2 (0), fingertip xy 2,
 *             goal 2 (= qpos[nq-4:nq-2]);  reward: 0.1 (1 - tanh 5 d_tip_box) [d < 0.1] + 0.3 (1 - tanh 5 d_box_target) [d < 0.1],
 *             success: d_box_target < distance_threshold (config/pusher.py: 0.05)
 * ====================================================================================================== */
#define MOPA_ENV_PUSH 0
#define MOPA_ENV_LIFT 1
#define MOPA_ENV_ASSEMBLY 2
#define MOPA_ENV_PUSHER 3
#define MOPA_ENV_OBS_DIM 40      /* largest observation (PUSH); see mopa_env_obs_dim */

typedef struct MopaEnvDesc {
    MopaModel model;                 /* body / joint arrays; LIFT also reads the touch geoms and the can's hull */
    int32_t kind;                    /* MOPA_ENV_* */
    int32_t n_arm;                   /* env.ref_joint_pos_indexes (7) */
    const int32_t *arm_qpos_idx;
    int32_t n_grip;                  /* env.ref_gripper_joint_pos_indexes (2) */
    const int32_t *grip_qpos_idx;
    int32_t n_act;                   /* position actuators written by `_do_simulation`, ctrl order: the arm's n_arm, then (LIFT) 2 gripper */
    const int32_t *act_qpos_idx;     /* [n_act] qpos address of the actuated joint */
    const double *act_ctrl_lo;       /* [n_act] ctrlrange (-inf / +inf when not ctrllimited) */
    const double *act_ctrl_hi;
    int32_t n_frames;                /* 5 (PUSH, ASSEMBLY), 3 (LIFT) or 4 (PUSHER), slots as listed above */
    const int32_t *frame_body;       /* [n_frames] */
    const double *frame_off;         /* [n_frames,3] offset in the body frame (site position; 0 for a body frame) */
    int32_t n_quats;                 /* 2 */
    const int32_t *quat_body;
    int32_t n_touch;                 /* LIFT: 1 + left + right collidable-geom indices (into model.geom_*), else 0 */
    const int32_t *touch_geom;
    int32_t n_touch_left;
    const double  *qpos_min;         /* [nq] per-qpos joint limits: _jnt_minimum[jnt_indices] (env/base.py:62-88) */
    const double  *qpos_max;         /* [nq] */
    const int32_t *qpos_limited;     /* [nq] */
    double ac_scale;                 /* config/sawyer.py (0.05) */
    double distance_threshold;       /* PUSH: 0.06 */
    double success_reward;           /* 150 */
    int32_t max_episode_steps;       /* 250 */
    int32_t device;                  /* HIP device ordinal, -1 = current */
} MopaEnvDesc;

typedef struct MopaEnv MopaEnv;

int mopa_env_create(const MopaEnvDesc *desc, MopaEnv **out);
void mopa_env_destroy(MopaEnv *env);
int mopa_env_obs_dim(const MopaEnv *env);      /* 40 / 35 / 38 / 20 */
int mopa_env_action_dim(const MopaEnv *env);   /* n_arm (+1 for LIFT) */

/* One step of E envs (all pointers device, f64 unless noted).  Per env e:
 *   prev = (is_planner && has_prev[e]) ? prev_state[e] : qpos[e, arm]
 *   desired = prev + clip(is_planner ? action[e, :n_arm] : action[e, :n_arm]*ac_scale, -ac_scale, +ac_scale)
 *   (PUSHER: desired = prev + action[e, :n_arm])
 *   LIFT: gripper targets = qpos[e, gripper] + action[e, n_arm]
 *   move_mask[e] (NULL = 1): bit 1 set -> env e sits this call out entirely (nothing read or written);
 *   bit 0 set -> every actuated joint = its target clamped to ctrlrange (kinematic servo); bit 0 clear -> the command is
 *   recorded but nothing moves
 *   prev_state[e] = desired, has_prev[e] = 1; limited qpos entries clipped to their range
 *   FK -> reward, success, obs; ep_len[e] += 1; done[e] = success || ep_len[e] == max_episode_steps
 * action == NULL: no step, only FK -> obs (reward/done/success untouched; used after a reset). */
int mopa_env_step_batch(MopaEnv *env, int64_t E, double *qpos_dev /*[E,nq] in/out*/, double *prev_state_dev /*[E,n_arm] in/out*/,
                        uint8_t *has_prev_dev /*[E] in/out*/, int32_t *ep_len_dev /*[E] in/out*/,
                        const double *action_dev /*[E,action_dim] or NULL*/, int32_t is_planner,
                        const uint8_t *move_mask_dev /*[E] or NULL*/, double *obs_dev /*[E,obs_dim]*/,
                        double *reward_dev /*[E]*/, uint8_t *done_dev /*[E]*/, uint8_t *success_dev /*[E]*/, void *stream);

/* Waypoint execution of the rollout runner (rl/mopa_rollouts.py:152-199) for E envs in ONE launch: env e steps through
 * traj[e, 0 .. path_len[e]-1] (full qpos rows; the action of step k is env.form_action(waypoint) = waypoint - current arm
 * state, is_planner semantics of mopa_env_step_batch) until its path ends or a step reports done, with
 *   smdp_rew[e] += disc_pow[k] * reward_k   (disc_pow[k] = discount_factor^k, supplied by the caller),
 *   smdp_done[e] = done_k, intra[e] = k for the last step taken.  Envs with path_len[e] == 0 are not touched.
 * last_extra (LIFT; NULL otherwise): the policy's gripper action, used as the gripper entry of the LAST waypoint's action
 * (:163-167); earlier waypoints use form_action's gripper difference.
 * rec_* (all four or none): per executed waypoint the obs after the step, the running return and the done flag, and
 * n_exec[e] = steps taken -- the reference's ob_list / meta_rew_list / done_list (input of its reuse_data relabelling).
 * obs / reward / done / success hold the env's last step afterwards, exactly as after that many mopa_env_step_batch calls. */
int mopa_env_exec_batch(MopaEnv *env, int64_t E, double *qpos_dev /*[E,nq] in/out*/, double *prev_state_dev /*[E,n_arm] in/out*/,
                        uint8_t *has_prev_dev /*[E] in/out*/, int32_t *ep_len_dev /*[E] in/out*/,
                        const double *traj_dev /*[E,L,nq]*/, const int64_t *path_len_dev /*[E]*/, int32_t L,
                        const double *disc_pow_dev /*[L]*/, const double *last_extra_dev /*[E] or NULL*/,
                        double *obs_dev /*[E,obs_dim]*/, double *reward_dev /*[E]*/,
                        uint8_t *done_dev /*[E]*/, uint8_t *success_dev /*[E]*/, double *smdp_rew_dev /*[E] in/out*/,
                        uint8_t *smdp_done_dev /*[E] in/out*/, int64_t *intra_dev /*[E] in/out*/,
                        double *rec_ob_dev /*[E,L,obs_dim] or NULL*/, double *rec_rew_dev /*[E,L] or NULL*/,
                        uint8_t *rec_done_dev /*[E,L] or NULL*/, int64_t *n_exec_dev /*[E] or NULL*/, void *stream);

/* ---- servo dynamics inside env.step (stage A: contact-free) ----------------------------------------
 * Replaces the kinematic limit of `_do_simulation` by what the reference runs when the robot touches nothing: nsub = 75
 * sub-steps of { qfrc_applied[arm] = qfrc_bias[arm]; ctrl = desired_state (+ LIFT gripper targets); mj_step } per env.step
 * (env/sawyer/sawyer_push_obstacle.py:186-203, sawyer_lift_obstacle.py:218-236, sawyer_assembly_obstacle.py:121-139,
 * env/base.py:388-392) on the arm's own tree, [3P] MuJoCo 2.0's pipeline restated: composite-rigid-body inertia + armature,
 * RNE bias (gravity included), joint damping integrated implicitly by the Euler step, position servos
 * force = clamp(kp ctrl - kp q, forcerange) (env/assets/xml/common/sawyer_joint_pos_act.xml, gripper_pick_pos_act.xml),
 * timestep 0.002 (sawyer_dependencies.xml:11).  NOT modelled: contacts (manipulated objects do not move), the solver's soft
 * joint-limit constraint (an inelastic stop at the range instead).  The tree is passed LUMPED -- one body per dof, bodies
 * welded to it folded into its inertial (mopa_rl_amd/dynamics.py) -- parents before children, at most 9 dofs. */
/* Stage B (Push): the manipulated object as a free rigid body with PENALTY contacts -- spring-damper normal force and
 * capped regularised Coulomb friction at the object's feature points found inside a collider (and a moving box collider's
 * vertices found inside the object); one-way coupling (the robot moves the object; its reaction on the arm is neglected).
 * NOT MuJoCo's solver (soft convex constraints + elliptic cones + noslip, sawyer_dependencies.xml:11): a labelled stand-in
 * so that env/sawyer/sawyer_push_obstacle.py's task has an object that can be pushed.  Colliders = the geoms MuJoCo pairs
 * with the object (candidate-pair list): static ones (body -1, pose in the world) first, then robot geoms sorted by their
 * dynamic body (pose in that body's frame).  With an object the qvel rows are [nd + 6]: the dofs, then (v, w) world. */
typedef struct MopaObjDesc {
    int32_t qadr;                     /* qpos address of the free joint (pos 3, quat 4); COM = body origin */
    double mass, inertia[3], damping; /* principal inertia in the body frame, free-joint damping */
    double half[3], rbound;           /* box half extents, bounding radius */
    int32_t nfeat; const double *feat;   /* [nfeat,3] feature points (body frame), the 8 vertices first */
    int32_t ncol;
    const int32_t *co_body, *co_type; /* [ncol] dynamic body (-1 static), mjtGeom type (plane / sphere / capsule / cylinder / box) */
    const double *co_size, *co_pos, *co_mat, *co_mu, *co_rbound;   /* [n,3] [n,3] [n,9] [n] [n] */
    double inv_mass, inv_inertia[3];  /* reciprocals (the integrator multiplies) */
    int32_t precull_every;            /* the full collider scan runs on every precull_every-th sub-step of a call (15) ... */
    double precull_margin;            /* ... with bounding spheres inflated by this much (0.15 m); other sub-steps visit its survivors */
    double kn, dn, eps_v, ct_max;     /* normal stiffness, normal damping, friction regularisation, cap of the friction's viscous coefficient */
} MopaObjDesc;

typedef struct MopaDynDesc {
    int32_t nd;
    const int32_t *parent;            /* [nd] parent dynamic body, -1 = fixed base */
    const int32_t *jtype;             /* [nd] 2 slide, 3 hinge */
    const int32_t *qadr;              /* [nd] qpos address of the dof */
    const double *rel_pos;            /* [nd,3] body frame in its parent dynamic body's frame (base: world) */
    const double *rel_quat;           /* [nd,4] wxyz */
    const double *axis, *jpos, *qref; /* [nd,3] [nd,3] [nd] joint axis / anchor (body frame), reference value */
    const double *mass, *ipos;        /* [nd] [nd,3] lumped mass, centre of mass (body frame) */
    const double *inertia;            /* [nd,6] about the COM, body axes: xx yy zz xy xz yz */
    const double *damping, *armature; /* [nd] */
    const int32_t *limited;           /* [nd] */
    const double *lo, *hi;            /* [nd] joint range */
    const int32_t *actuated;          /* [nd] must agree with the env's actuator list (MopaEnvDesc.act_qpos_idx) */
    const double *kp, *force_lo, *force_hi;   /* [nd] servo gain, force range (-inf / +inf when not forcelimited) */
    const int32_t *gravcomp;          /* [nd] the env copies qfrc_bias into qfrc_applied for this dof (the arm's joints) */
    double gravity[3];
    double timestep;                  /* 0.002 */
    int32_t nsub;                     /* int(frame_dt / timestep) = 75 */
    const MopaObjDesc *obj;           /* NULL: stage A (only the robot moves) */
} MopaDynDesc;
int mopa_env_attach_dynamics(MopaEnv *env, const MopaDynDesc *desc);
/* (stage C) contacts of the arm, of the manipulated object (Push: cube, Lift: can, Assembly: furniture; a
 * free rigid body) and between the two, two-way coupled behind ONE soft-constraint solve per sub-step.  Replaces, inside
 * `sim.step()` of the reference's `_do_simulation` loop (env/sawyer/sawyer_push_obstacle.py:186-203, sawyer_lift_obstacle.py:
 * 218-236, sawyer_assembly_obstacle.py:121-139; options env/assets/xml/common/sawyer_dependencies.xml:11), [3P] MuJoCo 2.0's
 * mj_collision + mj_makeConstraint + the constraint solver + mj_Euler.  RESTATED FROM THE PUBLISHED SOLVER, PARITY UNPINNED:
 * soft constraints with solref / solimp impedance; the default pipeline (`solver` 2) is the XML's: ELLIPTIC friction cones
 * (`cone="elliptic"`, condim 3: MuJoCo's primal elliptic-cone cost) in the Newton solver (the XML names no solver = MuJoCo's
 * default) with `iterations="50"` / `tolerance="1e-10"`, then the noslip pass (`noslip_iterations="5"`: friction re-solved
 * without the regulariser inside the friction disc), joint limits as solver rows (`limit_rows`).  Selectable: Newton with
 * pyramidal cones (`solver` 1), projected Gauss-Seidel with pyramidal cones (`solver` 0).  Not restated: torsional / rolling
 * friction (condim 4 / 6 pairs are solved as condim 3), MuJoCo's own pair functions (see "Collision geometry").
 * obj_qadr < 0 (with np = 0, limit_rows = 0): NO contact stage and no object -- the contact-free servo dynamics of stage A in this
 * kernel's 16-lanes-per-env mapping (qvel rows stay [nd]).
 * Collision geometry: sampled feature points of one geom in the exact signed-distance function of the other (plane, sphere,
 * capsule, cylinder, box; the can as the bounding cylinder of its hull).
 * Bodies: 0 .. nd-1 the lumped dynamic bodies of the arm, nd the object, -1 the world.  Shapes are posed in the frame of
 * their body, features likewise.  A directed pair (F, S) tests the features of shape F in the distance function of shape S;
 * pr_par rows: mu, margin, K, B, d0, dmax, width, 0 (MuJoCo's per-pair mix of friction / margin / solref / solimp, formed on
 * the host).  Call after mopa_env_attach_dynamics (without MopaObjDesc); qvel rows become [nd + 6]: the dofs, then the
 * object's (velocity of its COM, angular velocity) in the world. */
#define MOPA_CT_MAXCON 16      /* contacts kept per env and sub-step: <= 16 (solver 0, 2: one contact per lane of an env's 16), <= 8 for solver 1 */
typedef struct MopaCtDesc {
    int32_t ns;
    const int32_t *sh_body, *sh_type;                       /* [ns] */
    const double *sh_size, *sh_pos, *sh_mat, *sh_rbound;    /* [ns,3] [ns,3] [ns,9] [ns] */
    const int32_t *sh_feat0;                                /* [ns + 1] */
    int32_t nf;
    const double *ft_pos, *ft_rad;                          /* [nf,3] [nf] */
    int32_t np;
    const int32_t *pr_f, *pr_s;                             /* [np] */
    const double *pr_par;                                   /* [np,12]: mu, margin, K, B, d0, dmax, width, -, condim (3 / 4 / 6), torsional friction,
                                                               rolling friction, - (solver 2 solves a pair with its condim; 0 / 1 as condim 3) */
    int32_t obj_qadr;                                       /* qpos address of the object's free joint */
    double obj_mass, obj_inertia[3], obj_ipos[3], obj_iquat[4], obj_damping;
    double obj_inv_mass, obj_inv_inertia[3], obj_inv_mass_d, obj_inv_inertia_d[3];
    int32_t maxcon, maxpair, iterations;
    double tolerance, inv_scale;
    int32_t precull_every;
    double precull_margin;
    int32_t near_every;              /* third culling level: active pairs within near_margin of contact, re-listed every near_every sub-steps */
    double near_margin;
    int32_t warmstart;
    int32_t solver;                  /* 2: Newton + elliptic cones (the XML's model; default of the Python host), 1: Newton + pyramidal cones,
                                        0: projected Gauss-Seidel + pyramidal cones */
    int32_t limit_rows;              /* joint limits as rows of the Newton solver (MuJoCo) instead of an inelastic stop (solver 1 / 2) */
    double lim_par[8];               /* their parameters in a pair record's layout: -, margin 0, K, B, d0, dmax, width, - */
    int32_t noslip_iterations;       /* sweeps of the noslip pass after the main solve (0 = none) */
    double noslip_tolerance;
    int32_t arena;                   /* solver 2: doubles of an env's LDS the (variable-size) contact records share; a contact whose record does not
                                        fit is dropped like one beyond maxcon.  0 = what keeps four waves on a CU (mopa_env_contact_arena tells) */
} MopaCtDesc;
int mopa_env_attach_contacts(MopaEnv *env, const MopaCtDesc *desc);
int mopa_ct_desc_size(void);           /* sizeof(MopaCtDesc) as the library was built (binding self-check) */
int mopa_env_contact_arena(const MopaEnv *env);   /* the arena (doubles per env) the attached contact stage runs with; -1 without one */
/* per-env counters of the last stepping launch: [E,4] int32 = contacts summed over the sub-steps, solver sweeps summed,
 * contacts dropped by the caps, largest contact count of a sub-step (NULL: not recorded) */
int mopa_env_set_contact_stats(MopaEnv *env, int32_t *stats_dev);
int mopa_env_dyn_dofs(const MopaEnv *env);      /* nd, or -1 without dynamics */
int mopa_env_dyn_qvel_width(const MopaEnv *env); /* nd (+ 6 with an object), or -1 */
/* mj_forward at (qpos, qvel): bias [E,nd] <- qfrc_bias (what the env reads as gravity compensation before its next
 * sub-step; call after a reset / set_state with qvel = 0); M (optional) <- packed lower triangle of the joint-space
 * inertia [E, nd (nd + 1) / 2]; mask (optional): envs with bit 1 set are skipped. */
int mopa_env_dyn_forward_batch(MopaEnv *env, int64_t E, const double *qpos_dev /*[E,nq]*/, const double *qvel_dev /*[E,nd]*/,
                               double *bias_dev /*[E,nd]*/, double *M_dev /*or NULL*/, const uint8_t *mask_dev /*[E] or NULL*/,
                               void *stream);
/* n sub-steps towards explicit (ctrl-range clamped) servo targets ctrl [E,nd] (entries of unactuated dofs ignored) */
int mopa_env_dyn_substeps_batch(MopaEnv *env, int64_t E, double *qpos_dev /*[E,nq] in/out*/, double *qvel_dev /*[E,nd] in/out*/,
                                double *bias_lag_dev /*[E,nd] in/out*/, const double *ctrl_dev /*[E,nd]*/, int32_t n, void *stream);
/* mopa_env_step_batch with the servo dynamics as `_do_simulation`: same arguments and bookkeeping, plus the carried
 * qvel / bias_lag [E,nd]; move_mask bit 0 clear -> the command is recorded, no sub-step runs.  The obs reports joint_vel /
 * gripper_qvel, and -- as in the reference -- reward and obs are taken BEFORE the joint-limit clamp of env/base.py:269-290.
 * action == NULL: obs refresh only (qvel is read for the obs). */
int mopa_env_step_dyn_batch(MopaEnv *env, int64_t E, double *qpos_dev, double *qvel_dev, double *bias_lag_dev, double *prev_state_dev,
                            uint8_t *has_prev_dev, int32_t *ep_len_dev, const double *action_dev, int32_t is_planner,
                            const uint8_t *move_mask_dev, double *obs_dev, double *reward_dev, uint8_t *done_dev,
                            uint8_t *success_dev, void *stream);

/* ---- bookkeeping of one batched rollout call (rl/mopa_rollouts.py:70-375 + rl/sac_agent.py:148-318 for E envs at once) -------------
 * The elementwise work of mopa_rl_amd/rollout.py::BatchMoPARollout.agent_step between the library's launches, as six
 * one-wave-per-env kernels (mopa_rollstep.inc names the stages).  All pointers are device buffers; bool buffers are bytes. */
typedef struct MopaRolloutStep {
    int64_t E;
    int32_t nq, n_arm, ac_dim, ac_stride, adim, obs_dim, K /* width of the straight-line pre-check: traj is [E, K+1, nq] */;
    int32_t discrete, normal_space;
    double omega, ac_scale, action_range, omega_over_scale, one_minus_omega, range_minus_scale;
    const double *lim_lo, *lim_hi, *lo_state, *hi_state, *lo_shrunk, *hi_shrunk, *safe_q;              /* [nq] */
    /* the env's buffers */
    const double *qpos, *obs, *reward;
    const uint8_t *done, *success;
    uint8_t *has_prev;
    /* the rollout's persistent state */
    uint8_t *busy, *pool_mask, *interp_overflow;
    int64_t *wait_since, *t_dev, *t_env, *pend_type;
    double *q_cur, *q_tgt, *pend_ob, *pend_ac;
    int64_t *c_rl, *c_interp, *c_mp_fail, *c_invalid;
    /* this call's inputs */
    const double *ac;                  /* [E, ac_stride] policy output */
    const int64_t *ac_type_in;         /* [E] (discrete) */
    const double *a_in;                /* [E, n_arm] joint displacement action from the IK (use_ik_target) or NULL */
    /* this call's buffers */
    double *prev_ob, *ac_tr, *a, *extra_ac, *target, *cur_m, *cur_v, *tgt_v, *traj;
    uint8_t *active, *is_pl, *pv, *plan_ok;
    int64_t *ac_type, *path_len;
    const uint8_t *tv, *ok;            /* target valid (after the pull-back); pre-check verdict */
    const int32_t *nst, *tlen;         /* pre-check: steps needed, trajectory length */
    const uint8_t *finished;           /* [E] envs whose planner query was picked up in this call, or NULL */
    double *act0;                      /* [E, adim] */
    uint8_t *flags, *sitting, *stepped, *is_pl_out;
    int64_t *plen_m;
    double *last_extra;                /* [E] or NULL */
    double *rew;
    uint8_t *done_out;
    int64_t *intra;
    double *ob_next;
    uint8_t *success_out;
    const uint8_t *retry_mask;         /* [E] envs waiting for a retry launch, or NULL */
    int64_t *pool_counts;              /* [2] <- number of set entries of pool_mask / retry_mask after the call, or NULL */
} MopaRolloutStep;
int mopa_rollout_stage(const MopaRolloutStep *step, int32_t stage /*0..5*/, void *stream);
int mopa_rollout_step_size(void);      /* sizeof(MopaRolloutStep) as the library was built (binding self-check) */
/* The planner's pick-up of the waiting envs (rl/mopa_rollouts.py:205-209 for E envs: the envs whose straight line is blocked get an RRT-Connect
 * query), one launch: out_count[0] <- number of set bytes of mask [E]; if min_n <= that <= cap: their indices ascending -> out_ids, rows of q_cur / q_tgt
 * [E,nq] -> out_cur / out_tgt [cap,nq], t_env[ids] -> out_steps, t_env[ids] + seed -> out_seeds, and the mask bytes are cleared; otherwise nothing
 * else is written (below min_n the pool waits; above cap the caller chooses which envs go first).  All buffers on the device; bool buffers are bytes. */
int mopa_rollout_pool_pick(int64_t E, int32_t nq, int64_t min_n, int64_t cap, uint8_t *mask_dev, const double *q_cur_dev, const double *q_tgt_dev,
                           const int64_t *t_env_dev, int64_t seed, int64_t *out_ids_dev, double *out_cur_dev, double *out_tgt_dev,
                           int64_t *out_steps_dev, int64_t *out_seeds_dev, int64_t *out_count_dev, void *stream);

/* The arm state the NEXT mopa_env_step_batch call with the same arguments would reach (desired_state clamped to ctrlrange
 * and joint limits), without stepping: input of a collision gate (mopa_is_valid_batch with samples_per_env = 1 -> move_mask). */
int mopa_env_desired_batch(MopaEnv *env, int64_t E, const double *qpos_dev /*[E,nq]*/, const double *prev_state_dev /*[E,n_arm]*/,
                           const uint8_t *has_prev_dev /*[E]*/, const double *action_dev /*[E,action_dim]*/, int32_t is_planner,
                           double *desired_dev /*[E,n_arm]*/, void *stream);

/* ======================================================================================================
 *  batched damped-least-squares IK of a site pose: replaces
 * qpos_from_site_pose(env, site, target_pos, target_quat, joint_names=..., max_steps, tol, ...)   env/inverse_kinematics.py:18-135
 * with nullspace_method :274-281.  Per env and iteration:
 *   err_pos = target_pos - site_xpos;
 *   with an orientation target (:88-93): err_rot = mju_quat2Vel(target_quat * conj(mju_mat2Quat(site_xmat)), 1);
 *   err_norm = |err_pos| (+ rot_weight |err_rot|); success if err_norm < tol;
 *   J = [jacp; jacr] over the movable joints (hinge: axis x (p_site - anchor) / axis, slide: axis / 0) -- 3 x n for a
 *       position target, 6 x n with an orientation target (:38-44,101-105);
 *   dq = (J^T J + regularization_strength I)^-1 J^T err   (always regularised, as the reference calls it, :113-115);
 *   give up if err_norm / |dq| > progress_thresh; |dq| capped at max_update_norm; qpos[movable] += dq.
 * The reference's third mode (target_quat without target_pos) raises inside numpy (6-row Jacobian against a 3-vector)
 * and is not offered.  The linear solve is an unpivoted Cholesky factorisation (the reference uses LAPACK LU through
 * np.linalg.solve): equal to round-off (tests/golden/ref_py_ik.npz: 1e-12 on whole solves, equal step counts).
 * ====================================================================================================== */
typedef struct MopaIkDesc {
    MopaModel model;              /* body / joint arrays */
    int32_t n_joints;             /* movable joints, 1..8 */
    const int32_t *joint_ids;     /* [n_joints] model joint ids (hinge / slide, one per body) */
    int32_t site_body;            /* body carrying the site */
    double site_off[3];           /* site position in that body's frame */
    double site_quat[4];          /* site orientation in that body's frame, wxyz (all zero = identity) */
    int32_t device;               /* HIP device ordinal, -1 = current */
} MopaIkDesc;

typedef struct MopaIk MopaIk;

int mopa_ik_create(const MopaIkDesc *desc, MopaIk **out);
void mopa_ik_destroy(MopaIk *ik);
/* E independent IK problems; qpos rows are updated in place (IKResult.qpos); err_norm / steps / success as IKResult.
 * target_quat_dev NULL = position target only (rot_weight ignored). */
int mopa_ik_solve_batch(MopaIk *ik, int64_t E, double *qpos_dev /*[E,nq] in/out*/, const double *target_pos_dev /*[E,3]*/,
                        const double *target_quat_dev /*[E,4] wxyz or NULL*/, double rot_weight, int32_t max_steps, double tol,
                        double max_update_norm, double progress_thresh, double regularization_strength,
                        double *err_norm_dev /*[E]*/, int32_t *steps_dev /*[E]*/, uint8_t *success_dev /*[E]*/, void *stream);

/* World pose of the IK site for E states: site_pos [E,3], site_mat [E,9] row-major -- what the MoPA+IK rollouts read before
 * they pose the problem (env.sim.data.get_site_xpos / get_site_xmat(config.ik_target), rl/mopa_rollouts.py:91-99,692). */
int mopa_ik_site_pose_batch(MopaIk *ik, int64_t E, const double *qpos_dev /*[E,nq]*/, double *site_pos_dev /*[E,3]*/,
                            double *site_mat_dev /*[E,9]*/, void *stream);

/* The IK problem of the MoPA + IK action space for E envs (MoPARolloutRunner._cart2dispalcement, rl/mopa_rollouts.py:87-99,681-696):
 * target_cart = clip(site_pos + action_range * ac[:3], world box); target_quat (w, x, y, z) = mulQuat(q_site[(w, x, y, y)], ac[3:7] / |ac[3:7]|)
 * with q_site the quaternion of the float32-rounded site matrix (util/env.py:mat2quat, w >= 0).  ac rows are ac_stride (>= 7) doubles apart;
 * world_lo / world_hi: 3 HOST doubles each (env.min_world_size / max_world_size).  One launch, one lane per env. */
int mopa_ik_targets_batch(MopaIk *ik, int64_t E, const double *site_pos_dev /*[E,3]*/, const double *site_mat_dev /*[E,9]*/,
                          const double *ac_dev /*[E,ac_stride]*/, int64_t ac_stride, double action_range, const double *world_lo /*[3] host*/,
                          const double *world_hi /*[3] host*/, double *target_cart_dev /*[E,3]*/, double *target_quat_dev /*[E,4]*/, void *stream);

/* The straight-line pre-check of SACAgent.plan / simple_interpolate (rl/sac_agent.py:198-204, 236-272) for E envs: the line
 * cur -> target is cut into int(s) equal steps, s = max(1, max_j |diff_j| / (0.8 ac_scale)) over the arm joints qpos[:n_arm]
 * (= the scene's active joints), each interior state a running sum from cur and validated (env row = cur); traj[e] = the
 * valid prefix of the walk followed by the exact target, traj_len[e] = its rows, ok[e] = 1 when no interior state was
 * invalid, n_steps[e] = int(s).  K (<= 64) is the fixed row capacity of the walk: callers pass an upper bound of int(s) and
 * check n_steps <= K.  cur must already be clipped (clip_qpos).  Three launches, no read-back. */
int mopa_interpolate_batch(MopaScene *scene, int64_t E, int32_t n_arm, int32_t K, const double *cur_dev /*[E,nq]*/,
                           const double *target_dev /*[E,nq]*/, double ac_scale, double *traj_dev /*[E,K+1,nq]*/,
                           int32_t *traj_len_dev /*[E]*/, uint8_t *ok_dev /*[E]*/, int32_t *n_steps_dev /*[E]*/, void *stream);

/* ===== planner paths -> executable trajectories (reference motion_planners/sampling_based_planner.py:71-99: un-wrap by
 * successive differences; rl/sac_agent.py:205-233: densification of long steps by simple_interpolate from the clipped
 * predecessor).  Three launches around one validity launch; no scene handle (pure arithmetic on caller buffers):
 *   mopa_paths_unwrap_batch    path rows [M, max_path, nq] as mopa_plan_batch wrote them are replaced IN PLACE by the
 *                              un-wrapped rows (row 0 = cur); seg_count[q, i] = interpolation steps of the segment
 *                              row i -> row i+1 (0 = no interpolation needed); n_walk[q] = sum of them; out_len[q] = rows of
 *                              the final trajectory (0 for queries with status != 0)
 *   mopa_paths_walk_batch      the interior states of all long segments, query after query at walk_off[q] (exclusive
 *                              prefix sum of n_walk) -> walk [sum n_walk, nq]; the caller validates them
 *                              (mopa_is_valid_batch on the rows' active columns, samples_per_env 1, rows as env rows)
 *   mopa_paths_assemble_batch  out [M, out_rows, nq]: per waypoint its interior states then the waypoint;
 *                              needs_fallback[q] = 1 when an interior state of q was invalid (the reference then plans
 *                              that segment with the simple / main planner, rl/sac_agent.py:300-306: left to the caller)
 * lo_state / hi_state / lo_shrunk / hi_shrunk [nq]: the agent's joint limits and limits +- joint_margin as the float32
 * arrays the reference holds them in (values widened to double; +-inf for unlimited coordinates) -- `clip_qpos`.
 * nq <= 64; arm coordinates are qpos[:n_arm]. */
int mopa_paths_unwrap_batch(int device, int64_t M, int32_t nq, int32_t n_arm, double *path_dev /*[M,max_path,nq] in/out*/,
                            int32_t max_path, const int32_t *path_len_dev /*[M]*/, const int32_t *status_dev /*[M]*/,
                            const double *cur_dev /*[M,nq]*/, double ac_scale, int32_t interpolate, const double *lo_state_dev,
                            const double *hi_state_dev, const double *lo_shrunk_dev, const double *hi_shrunk_dev,
                            int32_t *seg_count_dev /*[M,max_path]*/, int32_t *n_walk_dev /*[M]*/, int32_t *out_len_dev /*[M]*/,
                            void *stream);
/* the same with the seam rule of the reference's un-wrap loop for models with UNLIMITED joints (sampling_based_planner.py:79-97):
 * bit c of seam_mask marks qpos coordinate c as one of `non_limited_idx`; the planner's states live in (-3.14, 3.14) there
 * (SamplingBasedPlanner.convert_nonlimited -> util/env.py:joint_convert wraps start and goal before planning), and a step with
 * abs(state - pre_state) > 3.14 is taken the short way round: + (3.14 - pre + state + 3.14) when it left through +3.14,
 * - (3.14 - state + pre + 3.14) when through -3.14 (3.14, not pi; sums in this order).  seam_mask 0 == mopa_paths_unwrap_batch. */
int mopa_paths_unwrap_seam_batch(int device, int64_t M, int32_t nq, int32_t n_arm, double *path_dev /*[M,max_path,nq] in/out*/,
                                 int32_t max_path, const int32_t *path_len_dev /*[M]*/, const int32_t *status_dev /*[M]*/,
                                 const double *cur_dev /*[M,nq]*/, double ac_scale, int32_t interpolate, const double *lo_state_dev,
                                 const double *hi_state_dev, const double *lo_shrunk_dev, const double *hi_shrunk_dev,
                                 int32_t *seg_count_dev /*[M,max_path]*/, int32_t *n_walk_dev /*[M]*/, int32_t *out_len_dev /*[M]*/,
                                 uint64_t seam_mask, void *stream);
int mopa_paths_walk_batch(int device, int64_t M, int32_t nq, int32_t n_arm, const double *path_dev, int32_t max_path,
                          const int32_t *path_len_dev, const int32_t *out_len_dev, double ac_scale, const double *lo_state_dev,
                          const double *hi_state_dev, const double *lo_shrunk_dev, const double *hi_shrunk_dev,
                          const int32_t *seg_count_dev, const int64_t *walk_off_dev /*[M]*/, double *walk_dev /*[sum n_walk,nq]*/,
                          void *stream);
int mopa_paths_assemble_batch(int device, int64_t M, int32_t nq, const double *path_dev, int32_t max_path,
                              const int32_t *path_len_dev, const int32_t *out_len_dev, const int32_t *seg_count_dev,
                              const int64_t *walk_off_dev, const double *walk_dev /*nullable when no query has interior states*/,
                              const uint8_t *walk_valid_dev, double *out_dev /*[M,out_rows,nq]*/, int32_t out_rows,
                              uint8_t *needs_fallback_dev /*[M]*/, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MOPA_HIP_H */
