#!/usr/bin/env python3
"""Generate tests/golden/ref_py_*.npz by IMPORTING THE REFERENCE'S PYTHON (build container only).

    python tools/gen_ref_py_golden.py [section ...]        # sections: host agent rollout env ik episode ... (default: all)

The reference's native arithmetic (MuJoCo 2.0, OMPL) is absent from this image, but the layers above it are plain
Python/numpy and can run here once their imports are satisfied (tools/refshim.py: module stubs, an oracle-backed
`PyKinematicPlanner`, a sim-shaped adapter over the oracle's FK).  Every array written below is the OUTPUT OF REFERENCE
CODE -- `util/env.py:joint_convert`, `motion_planners/sampling_based_planner.py:SamplingBasedPlanner.plan`,
`rl/planner_agent.py:PlannerAgent.plan`, `rl/sac_agent.py:SACAgent.{convert2planner_displacement, invert_displacement,
clip_qpos, simple_interpolate, plan, is_planner_ac}`, `rl/mopa_rollouts.py:MoPARolloutRunner.run`,
`env/sawyer/*.py:{step,_step,compute_reward,_get_obs}`, `env/base.py:{step,_after_step}`,
`env/inverse_kinematics.py:{qpos_from_site_pose,nullspace_method}` -- executed unmodified on the inputs stored next to
it.  The fixtures are data (inputs + expected outputs); the reference's source never leaves /root/reference.

What a fixture does and does not pin: the Python-level rows of SURVEY 8 (A9, A10, A11, f1/N1 around the physics, f2,
f3) against the reference's own code.  Validity verdicts, RRT-Connect paths and body/site poses inside those runs come
from this repo's CPU oracle (oracle/mopa_oracle.c), which itself stays unpinned against MuJoCo/OMPL.
"""
import argparse
import os
import sys
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
ROOT = refshim.ROOT
OUT = os.path.join(ROOT, "tests", "golden")

from mopa_rl_amd.scene import ENV_SPECS, default_qpos, load_scene, planner_inputs, qpos_joint_arrays  # noqa: E402

XML = {"SawyerPushObstacle-v0": "sawyer_push_obstacle.xml", "SawyerLiftObstacle-v0": "sawyer_lift_obstacle.xml",
       "SawyerAssemblyObstacle-v0": "sawyer_assembly_obstacle.xml", "PusherObstacle-v0": "pusher_obstacle.xml"}


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"  {name}: {os.path.getsize(path)} B, {len(arrays)} arrays")


# ------------------------------------------------------------------------------------------------------------------
# reference objects built the way rl/trainer.py builds them (without MuJoCo / torch networks)
# ------------------------------------------------------------------------------------------------------------------
def make_config(env_name, **over):
    """config/__init__.py + config/sawyer.py + config/motion_planner.py defaults that the executed code reads."""
    spec = ENV_SPECS[env_name]
    c = SimpleNamespace(
        env=env_name, omega=spec.omega, ac_space_type="piecewise", action_range=spec.action_range, timelimit=spec.timelimit,
        simple_planner_timelimit=spec.simple_planner_timelimit, interpolation=True, joint_margin=spec.joint_margin,
        planner_type="rrt_connect", simple_planner_type="rrt_connect", planner_objective="path_length", threshold=spec.threshold,
        range=spec.range, simple_planner_range=spec.simple_planner_range, contact_threshold=spec.contact_threshold,
        is_simplified=False, simplified_duration=0.01, simple_planner_simplified=False, simple_planner_simplified_duration=0.01,
        seed=1234, device="cpu", use_ik_target=False, discrete_action=False, invalid_target_handling=True,
        num_trials=spec.num_trials, step_size=spec.step_size, discount_factor=0.99, reuse_data=False, max_reuse_data=30,
        ik_target="grip_site", mopa=True, _xml_path=os.path.join(refshim.REFERENCE, "env", "assets", "xml", XML[env_name]))
    c.__dict__.update(over)
    return c


def make_agent(env_name, config, ac_dim=None):
    """A reference `SACAgent` with exactly the planner-facing state `SACAgent.__init__` gives it (rl/sac_agent.py:31-110);
    networks / optimisers / replay are not built (torch-only, not on the path)."""
    from gym import spaces
    from rl.planner_agent import PlannerAgent
    from rl.sac_agent import SACAgent
    pi = planner_inputs(env_name)
    m = pi.model
    idx, lo, hi, lim = qpos_joint_arrays(m)                    # env/base.py:62-88 (float64, per joint)
    n = len(pi.ref_joint_pos_indexes)
    ac_dim = n if ac_dim is None else ac_dim
    ac_space = spaces.Dict([("default", spaces.Box(low=-np.ones(ac_dim), high=np.ones(ac_dim), dtype=np.float32))])
    joint_space = spaces.Dict([("default", spaces.Box(low=lo, high=hi, dtype=np.float32))])    # env/base.py:91-98
    config.passive_joint_idx = pi.passive_joint_idx
    config.ignored_contact_geom_ids = pi.ignored_contacts
    non_limited_idx = np.where(m.jnt_limited[:ac_dim] == 0)[0]    # rl/trainer.py:80-82
    a = object.__new__(SACAgent)
    a._config, a._ac_space = config, ac_space
    a._jnt_indices, a._ref_joint_pos_indexes = list(idx), list(pi.ref_joint_pos_indexes)
    a._joint_space, a._is_jnt_limited = joint_space, lim
    a._jnt_minimum, a._jnt_maximum = joint_space["default"].low, joint_space["default"].high
    a._planner = PlannerAgent(config, ac_space, non_limited_idx, planner_type=config.planner_type,
                              passive_joint_idx=config.passive_joint_idx, ignored_contacts=config.ignored_contact_geom_ids,
                              is_simplified=config.is_simplified, simplified_duration=config.simplified_duration, range_=config.range)
    a._simple_planner = PlannerAgent(config, ac_space, non_limited_idx, planner_type=config.simple_planner_type,
                                     passive_joint_idx=config.passive_joint_idx, ignored_contacts=config.ignored_contact_geom_ids,
                                     goal_bias=1.0, is_simplified=config.simple_planner_simplified,
                                     simplified_duration=config.simple_planner_simplified_duration, range_=config.simple_planner_range)
    a._omega = config.omega
    return a, pi


class Streams:
    """The counter-RNG stream convention of mopa_rl_amd/rollout.py: within one agent step of env e (of E), the first
    main-planner query uses stream e, later main-planner queries (densification fall-back) 2E + e, simple-planner queries
    E + e; the seed is cfg.seed + t."""

    def __init__(self, agent, E, seed, max_nodes, max_path):
        self.main, self.simple = agent._planner.planner.planner, agent._simple_planner.planner.planner
        self.E, self.seed = E, seed
        refshim.PlanCtx.max_nodes, refshim.PlanCtx.max_path = max_nodes, max_path
        refshim.PlanCtx.stream_of = self._stream
        self.begin(0, 0)

    def begin(self, e, t):
        self.e, self.main_calls = e, 0
        refshim.PlanCtx.seed = self.seed + t

    def _stream(self, planner):
        if planner is self.simple:
            return self.E + self.e
        assert planner is self.main
        self.main_calls += 1
        return self.e if self.main_calls == 1 else 2 * self.E + self.e


def pad(rows, L, width):
    out = np.zeros((len(rows), L, width))
    ln = np.zeros(len(rows), dtype=np.int64)
    for k, r in enumerate(rows):
        r = np.asarray(r, dtype=np.float64).reshape(-1, width)
        out[k, :len(r)] = r
        ln[k] = len(r)
    return out, ln


# ------------------------------------------------------------------------------------------------------------------
# section "host": pure-numpy rows (A9 + the displacement maps / clip of A10)
# ------------------------------------------------------------------------------------------------------------------
def gen_host():
    from motion_planners.sampling_based_planner import SamplingBasedPlanner
    from rl.planner_agent import PlannerAgent
    from util.env import joint_convert
    rng = np.random.default_rng(20260928)
    out = {}
    # util/env.py:15-25
    ang = np.concatenate([rng.uniform(-20, 20, 400), np.array([0.0, 3.14, -3.14, 6.28, -6.28, 3.1399999, 3.1400001, 9.42, -9.42, 1e-9, -1e-9]),
                          np.arange(-8, 9) * 3.14])
    out["jc_in"], out["jc_out"] = ang, np.array([joint_convert(float(a)) for a in ang])

    # sampling_based_planner.py:57-100 + planner_agent.py:42-52 on scripted native results (Pusher layout: joint0 unlimited)
    nq = 16

    class Scripted:
        def __init__(self): self.rows, self.seen = None, None
        def plan(self, s, g, t): self.seen = (np.array(s), np.array(g), t); return [list(r) for r in self.rows]

    sbp = object.__new__(SamplingBasedPlanner)
    sbp.planner, sbp.non_limited_idx = Scripted(), [0]
    pa = object.__new__(PlannerAgent)
    pa.planner, pa._config = sbp, SimpleNamespace(timelimit=1.0)
    K, Lmax = 40, 24
    starts, goals, states, slen = np.zeros((K, nq)), np.zeros((K, nq)), np.zeros((K, Lmax, nq)), np.zeros(K, dtype=np.int64)
    conv_s, conv_g, trajs, tlen = np.zeros((K, nq)), np.zeros((K, nq)), np.zeros((K, Lmax, nq)), np.zeros(K, dtype=np.int64)
    flags = np.zeros((K, 3), dtype=np.int64)           # success, valid, exact of PlannerAgent.plan
    pa_traj, pa_len = np.zeros((K, Lmax, nq)), np.zeros(K, dtype=np.int64)
    for k in range(K):
        s = rng.uniform(-3, 3, nq)
        s[0] = rng.uniform(-12, 12)                     # an un-wrapped unlimited joint, several turns out
        g = s + rng.uniform(-1, 1, nq)
        L = int(rng.integers(2, Lmax))
        if k % 10 == 8:
            rows = np.full((1, nq), -5.0)
        elif k % 10 == 9:
            rows = np.full((1, nq), -4.0)
        else:
            # what the native planner returns: states in the WRAPPED space (joint0 in (-pi, pi]), possibly crossing the seam
            w = np.linspace(0, 1, L)[:, None]
            rows = (1 - w) * s + w * g
            j0 = joint_convert(float(s[0])) + np.cumsum(np.r_[0.0, rng.uniform(-0.9, 0.9, L - 1)]) * (3.0 if k % 3 == 0 else 1.0)
            rows[:, 0] = (j0 + np.pi) % (2 * np.pi) - np.pi
        sbp.planner.rows = rows
        tr, st, valid, exact = sbp.plan(s, g, 1.0)
        starts[k], goals[k] = s, g
        states[k, :len(rows)], slen[k] = rows, len(rows)
        conv_s[k], conv_g[k] = sbp.planner.seen[0], sbp.planner.seen[1]
        trajs[k, :len(tr)], tlen[k] = tr, len(tr)
        t2, success, v2, e2 = pa.plan(s, g, 1.0)
        flags[k] = (int(success), int(v2), int(e2))
        pa_traj[k, :len(t2)], pa_len[k] = t2, len(t2)
    out.update(uw_start=starts, uw_goal=goals, uw_states=states, uw_states_len=slen, uw_conv_start=conv_s, uw_conv_goal=conv_g,
               uw_traj=trajs, uw_traj_len=tlen, pa_flags=flags, pa_traj=pa_traj, pa_len=pa_len)

    # rl/sac_agent.py:148-196 (both action-space types), :237-260
    for typ in ("piecewise", "normal"):
        cfg = make_config("SawyerPushObstacle-v0", ac_space_type=typ)
        agent, pi = make_agent("SawyerPushObstacle-v0", cfg)
        ac = np.concatenate([rng.uniform(-1, 1, (100, 7)), np.array([[0.7] * 7, [-0.7] * 7, [0.0] * 7, [1.0] * 7, [-1.0] * 7, [0.6999999] * 7])])
        disp = np.array([agent.convert2planner_displacement(a, 0.05) for a in ac])
        d_in = np.concatenate([rng.uniform(-0.5, 0.5, (100, 7)), rng.uniform(-0.06, 0.06, (40, 7)), np.array([[0.05] * 7, [-0.05] * 7, [0.0] * 7])])
        inv = np.array([agent.invert_displacement(d, 0.05) for d in d_in])
        isp = np.array([agent.is_planner_ac(OrderedDict(default=a)) for a in ac])
        out.update({f"disp_{typ}_ac": ac, f"disp_{typ}_out": disp, f"inv_{typ}_in": d_in, f"inv_{typ}_out": inv, f"ispl_{typ}": isp})
    cfg = make_config("SawyerPushObstacle-v0")
    agent, pi = make_agent("SawyerPushObstacle-v0", cfg)
    q0 = default_qpos("SawyerPushObstacle-v0", pi.model)
    lo, hi = pi.jnt_minimum, pi.jnt_maximum
    Q = np.repeat(q0[None], 64, axis=0)
    Q[:, :7] += rng.normal(0, 0.5, (64, 7))
    for k in range(0, 64, 4):                          # at / beyond a limit, both sides, by ulps and by a lot
        j = k % 7
        Q[k, j] = hi[j] + (0.0, 1e-9, 0.3, -1e-9)[(k // 4) % 4]
        Q[k + 1, j] = lo[j] - (0.0, 1e-9, 0.3, -1e-9)[(k // 4) % 4]
    Q[5, 7] = 0.5                                      # a passive limited joint (gripper slide) out of range
    out["clip_in"], out["clip_out"] = Q, np.array([agent.clip_qpos(q.copy()) for q in Q])
    save("ref_py_host.npz", **out)


# ------------------------------------------------------------------------------------------------------------------
# section "agent": SACAgent.simple_interpolate / plan on the Push scene (A10), validity + RRT-Connect from the oracle
# ------------------------------------------------------------------------------------------------------------------
AGENT_PARAMS = dict(timelimit=0.15, max_nodes=512, max_path=128, seed=1234)


def agent_cases(pi, K, rng):
    q0 = default_qpos(pi.spec.env, pi.model)
    cur = np.repeat(q0[None], K, axis=0)
    cur[:, :7] += rng.normal(0, 0.02, (K, 7))
    cur[3, 0] = pi.jnt_maximum[0] + 0.01                 # start beyond a joint limit: clip_qpos with the margin
    tgt = cur.copy()
    kind = np.arange(K) % 4
    step = np.where(kind[:, None] == 0, rng.uniform(-0.04, 0.04, (K, 7)),            # one interpolation step
                    np.where(kind[:, None] == 1, rng.uniform(-0.5, 0.5, (K, 7)),       # long straight lines
                             rng.uniform(-0.5, 0.5, (K, 7))))
    tgt[:, :7] += step
    far = kind >= 2                                      # towards the table / bin: blocked lines, planner paths
    tgt[far, 1] = cur[far, 1] + rng.uniform(0.35, 0.5, far.sum())
    tgt[far, 3] = cur[far, 3] - rng.uniform(0.3, 0.5, far.sum())
    tgt[:, :7] = np.clip(tgt[:, :7], pi.jnt_minimum, pi.jnt_maximum)
    return cur, tgt


def gen_agent():
    env = "SawyerPushObstacle-v0"
    P = AGENT_PARAMS
    cfg = make_config(env, timelimit=P["timelimit"])
    agent, pi = make_agent(env, cfg)
    K = 64
    st = Streams(agent, K, P["seed"], P["max_nodes"], P["max_path"])
    rng = np.random.default_rng(7)
    cur, tgt = agent_cases(pi, K, rng)
    nq = pi.model.nq
    si, pl = [], []
    si_f, pl_f = np.zeros((K, 3), dtype=np.int64), np.zeros((K, 4), dtype=np.int64)
    tgt_valid = np.zeros(K, dtype=np.int64)
    for k in range(K):
        st.begin(k, 0)
        tgt_valid[k] = int(agent.isValidState(tgt[k]))
        tr, success, valid, exact = agent.simple_interpolate(cur[k].copy(), tgt[k].copy(), 0.05)
        si.append(tr); si_f[k] = (success, valid, exact)
        st.begin(k, 0)
        tr, success, interpolation, valid, exact = agent.plan(cur[k].copy(), tgt[k].copy(), ac_scale=0.05)
        pl.append(np.asarray(tr).reshape(-1, nq)); pl_f[k] = (success, interpolation, valid, exact)
    L = max(max(len(t) for t in si), max(len(t) for t in pl))
    si_t, si_l = pad(si, L, nq)
    pl_t, pl_l = pad(pl, L, nq)
    print("  agent: simple_interpolate ok", int(si_f[:, 0].sum()), "/", K, "| plan success", int(pl_f[:, 0].sum()),
          "interpolation", int((pl_f[:, 0] & pl_f[:, 1]).sum()), "rrt", int((pl_f[:, 0] & (1 - pl_f[:, 1])).sum()),
          "target valid", int(tgt_valid.sum()), "max len", L)
    save("ref_py_agent_push.npz", cur=cur, tgt=tgt, tgt_valid=tgt_valid, si_traj=si_t, si_len=si_l, si_flags=si_f, plan_traj=pl_t,
         plan_len=pl_l, plan_flags=pl_f, params=np.array([P["timelimit"], P["max_nodes"], P["max_path"], P["seed"], 0.05]))


# ------------------------------------------------------------------------------------------------------------------
# reference env objects over FakeSim (kinematic limit of the physics)
# ------------------------------------------------------------------------------------------------------------------
ENV_KW = dict(reward_type="dense", distance_threshold=0.06, success_reward=150.0, frame_skip=1, ctrl_reward_coef=0.0,
              use_robot_indicator=False, use_target_robot_indicator=False)


def make_ref_env(env_name, seed=0, max_episode_steps=250):
    """An instance of the reference's env class (env/sawyer/*.py) whose `sim` is refshim.FakeSim.  `__init__` is not run
    (it loads MuJoCo); the attributes it would set are set here from the same sources (env/base.py:27-98,
    env/sawyer/sawyer.py:19-56).  `_do_simulation` -- one MuJoCo step of the position servos -- becomes: every actuated
    joint reaches its (ctrl-range-clamped) target."""
    if env_name == "PusherObstacle-v0":
        return make_ref_env_pusher(seed=seed, max_episode_steps=max_episode_steps)
    import env.sawyer as ref_envs
    from gym import spaces
    cls = {"SawyerPushObstacle-v0": ref_envs.SawyerPushObstacleEnv, "SawyerLiftObstacle-v0": ref_envs.SawyerLiftObstacleEnv,
           "SawyerAssemblyObstacle-v0": ref_envs.SawyerAssemblyObstacleEnv}[env_name]
    spec = ENV_SPECS[env_name]
    m = load_scene(spec.scene)
    env = object.__new__(cls)
    sim = refshim.FakeSim(m, actuators=[f"pos_{m.jnt_names[j]}" for j in m.act_joint])
    env.sim, env.data = sim, sim.data
    env._kwargs = dict(ENV_KW)
    env._env_config = {"frame_skip": 1, "ctrl_reward": 0.0, "init_randomness": 1e-5, "max_episode_steps": max_episode_steps,
                       "unstable_penalty": 0, "reward_type": "dense", "distance_threshold": ENV_KW["distance_threshold"]}
    env._frame_skip, env._frame_dt = 1, sim.model.opt.timestep          # one pass through the sub-step loop
    env._seed, env.np_random = seed, np.random.RandomState(seed)
    env.render_mode, env._viewer = "no", None
    idx, lo, hi, lim = qpos_joint_arrays(m)
    env.jnt_indices, env._jnt_minimum, env._jnt_maximum, env._is_jnt_limited = list(idx), lo, hi, lim
    env.joint_space = spaces.Dict([("default", spaces.Box(low=lo, high=hi, dtype=np.float32))])
    env.action_space = spaces.Dict([("default", spaces.Box(low=-np.ones(env.dof), high=np.ones(env.dof), dtype=np.float32))])
    env._ac_scale = spec.ac_scale
    env.use_robot_indicator = env.use_target_robot_indicator = False
    env._prev_state, env._i_term = None, 0.0
    env.min_world_size, env.max_world_size = [-1.2, -1.2, 0.0], [1.2, 1.2, 2.0]
    env._fail = env._terminal = env._success = False
    env._episode_reward, env._episode_length, env._episode_time = 0, 0, 0.0
    env._get_reference()
    act_adr = np.array([m.jnt_qposadr[j] for j in m.act_joint])
    act_lo = np.where(m.act_ctrllimited == 1, m.act_ctrlrange[:, 0], -np.inf)
    act_hi = np.where(m.act_ctrllimited == 1, m.act_ctrlrange[:, 1], np.inf)

    def kinematic_limit(a=None):
        sim.data.ctrl[:] = a[:]
        sim.data.qpos[act_adr] = np.minimum(np.maximum(sim.data.ctrl, act_lo), act_hi)
        sim.data.qvel[:] = 0.0
        sim.forward()

    env._do_simulation = kinematic_limit
    return env


PUSHER_KW = dict(reward_type="dense", distance_threshold=0.05, success_reward=150.0, frame_skip=1, ctrl_reward_coef=0.0)   # config/pusher.py


def make_ref_env_pusher(seed=0, max_episode_steps=250):
    """The reference's `PusherObstacleEnv` (env/pusher/pusher_obstacle.py) over refshim.FakeSim, built like make_ref_env.
    Its `_step` drives torque motors through the env's PID loop (`_get_control` + `_do_simulation`, env/base.py:200-209,
    388-400) for int(frame_dt / dt) sub-steps; the kinematic limit of that loop -- the four joints reach `desired_state`,
    nothing else moves, velocities are zero -- replaces the two methods: `_get_control` hands the desired state through,
    `_do_simulation` writes it."""
    from env.pusher import PusherObstacleEnv
    from gym import spaces
    spec = ENV_SPECS["PusherObstacle-v0"]
    m = load_scene(spec.scene)
    env = object.__new__(PusherObstacleEnv)
    sim = refshim.FakeSim(m, actuators=[f"motor_{j}" for j in spec.robot_joints])
    sim.model.opt.timestep = 0.01                    # pusher_gripper.xml:7
    env.sim, env.data = sim, sim.data
    env._kwargs = dict(PUSHER_KW)
    env._env_config = {"frame_skip": 1, "ctrl_reward": 0.0, "init_randomness": 1e-5, "max_episode_steps": max_episode_steps,
                       "unstable_penalty": 0, "reward_type": "dense", "distance_threshold": PUSHER_KW["distance_threshold"],
                       "success_reward": PUSHER_KW["success_reward"]}
    env._frame_skip, env._frame_dt = 1, sim.model.opt.timestep          # one pass through the sub-step loop
    env._seed, env.np_random = seed, np.random.RandomState(seed)
    env.render_mode, env._viewer = "no", None
    idx, lo, hi, lim = qpos_joint_arrays(m)
    env.jnt_indices, env._jnt_minimum, env._jnt_maximum, env._is_jnt_limited = list(idx), lo, hi, lim
    env.joint_space = spaces.Dict([("default", spaces.Box(low=lo, high=hi, dtype=np.float32))])
    env.action_space = spaces.Dict([("default", spaces.Box(low=-np.ones(4), high=np.ones(4), dtype=np.float32))])
    env.joint_names = list(spec.robot_joints)
    env.ref_joint_pos_indexes = [sim.model.get_joint_qpos_addr(x) for x in env.joint_names]
    env.ref_joint_vel_indexes = [sim.model.get_joint_qvel_addr(x) for x in env.joint_names]
    env.ref_indicator_joint_pos_indexes = [sim.model.get_joint_qpos_addr(x + "-goal") for x in env.joint_names]
    env.ref_dummy_joint_pos_indexes = [sim.model.get_joint_qpos_addr(x + "-dummy") for x in env.joint_names]
    env._ac_scale = 0.1
    env.min_world_size, env.max_world_size = [-0.41, -0.41], [0.41, 0.41]
    env._prev_state, env._i_term = None, 0.0
    env._fail = env._terminal = env._success = False
    env._episode_reward, env._episode_length, env._episode_time = 0, 0, 0.0
    env._set_camera_position = env._set_camera_rotation = lambda *a, **k: None
    adr = np.array(env.ref_joint_pos_indexes)

    env._get_control = lambda state, prev_state, target_vel: np.asarray(state, dtype=np.float64).copy()

    def kinematic_limit(a=None):
        sim.data.qpos[adr] = a
        sim.data.qvel[:] = 0.0
        sim.forward()

    env._do_simulation = kinematic_limit
    return env


def flat_ob(ob):
    return np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in ob.values()])


# ------------------------------------------------------------------------------------------------------------------
# section "rollout": rl/mopa_rollouts.py:MoPARolloutRunner.run, env by env, on scripted actions (A11, f2)
# ------------------------------------------------------------------------------------------------------------------
ROLLOUT_PARAMS = dict(timelimit=0.15, max_nodes=512, max_path=128, seed=1234, max_episode_steps=14, num_trials=10)
COUNTERS = ("mp", "rl", "interpolation", "mp_fail", "approximate", "invalid")


def rollout_actions(E, T, n_ac, rng):
    ac = rng.uniform(-1, 1, size=(E, T, n_ac)) * rng.choice([0.6, 0.9, 1.0], size=(E, T, 1))
    for t in (1, 3):                  # far targets towards the table / bin: blocked straight lines, invalid targets
        ac[: E // 2, t, 1] = 1.0
        ac[: E // 2, t, 3] = -1.0
    ac[E // 2: 3 * E // 4, 2, 1] = 1.0
    ac[E // 2: 3 * E // 4, 2, 5] = 1.0
    return ac


def pusher_seam_starts(pi, rng, n):
    """arm poses of the Pusher with joint0 within 0.12 rad of +-3.14 that the validity oracle accepts (straight, the arm lies in
    obstacle 4 / 5 there): rollouts started from them carry the unlimited joint across the seam"""
    from oracle import oracle as O
    orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    q0 = np.asarray(pi.model.qpos0, dtype=np.float64)
    out = []
    while len(out) < n:
        q = q0.copy()
        q[0] = rng.choice([-1.0, 1.0]) * rng.uniform(3.02, 3.13)
        q[1:4] = rng.uniform(-2.5, 2.5, 3)
        if orc.is_valid(q)[0]:
            out.append(q[:4].copy())
    return out


def gen_rollout(env_name="SawyerPushObstacle-v0", tag="push", E=32, T=5, reuse=False, ik=False, discrete=False):
    import util.env as ref_util_env
    from rl.mopa_rollouts import MoPARolloutRunner
    ref_util_env.np = refshim.NumpyCompat()
    P = ROLLOUT_PARAMS
    if env_name == "PusherObstacle-v0":
        P = dict(P, timelimit=0.5)            # the cluttered Pusher scene: 1000 iterations, so that some blocked lines get planned
    cfg = make_config(env_name, timelimit=P["timelimit"], reuse_data=reuse, num_trials=P["num_trials"], use_ik_target=ik,
                      discrete_action=discrete)
    pusher = env_name == "PusherObstacle-v0"
    n_ac = 8 if env_name == "SawyerLiftObstacle-v0" else (4 if pusher else 7)
    agent, pi = make_agent(env_name, cfg, ac_dim=n_ac)
    st = Streams(agent, E, P["seed"], P["max_nodes"], P["max_path"])
    rng = np.random.default_rng(11)
    if ik:
        # MoPA + IK action space (rl/trainer.py:93-125): Cartesian displacement (3) + rotation quaternion (4) in [-1, 1]
        AC = rng.uniform(-1, 1, size=(E, T, 7))
        AC[:, :, :3] *= rng.choice([0.05, 0.3, 1.0], size=(E, T, 1))           # small steps (direct) and long reaches (planner)
        AC[:, :, 3] = np.abs(AC[:, :, 3]) + 0.5                                 # mostly small rotations ...
        AC[: E // 4, :, 3:] = rng.uniform(-1, 1, size=(E // 4, T, 4))            # ... and some arbitrary ones
    elif pusher:
        # joint0 (unlimited) is driven hard one way in half of the envs -- from a start next to +-3.14 (below) that takes the planner's
        # SO(2) coordinate across the seam --, the others get mixed direct / planner actions
        AC = rng.uniform(-1, 1, size=(E, T, n_ac)) * rng.choice([0.5, 0.9, 1.0], size=(E, T, 1))
        AC[: E // 2] *= 0.6
        AC[: E // 2, :, 0] = np.where(np.arange(E // 2)[:, None] % 2 == 0, 1.0, -1.0) * rng.uniform(0.71, 0.8, size=(E // 2, T))
        seam_starts = pusher_seam_starts(pi, rng, E // 2)
    else:
        AC = rollout_actions(E, T, n_ac, rng)
    # --discrete_action (rl/mopa_rollouts.py:86-88): the policy's `ac_type` head, not the action's magnitude, picks the planner
    AC_TYPE = rng.integers(0, 2, size=(E, T)) if discrete else None
    nq = pi.model.nq
    out = dict(ac=AC, qpos_start=np.zeros((E, T, nq)), ep_len_start=np.zeros((E, T), dtype=np.int64), qpos_end=np.zeros((E, T, nq)),
               rew=np.zeros((E, T)), done=np.zeros((E, T), dtype=np.int64), intra=np.zeros((E, T), dtype=np.int64),
               counters=np.zeros((E, T, len(COUNTERS)), dtype=np.int64), ob=None, ob_next=None,
               pulled_back=np.zeros((E, T), dtype=np.int64))
    extra = []              # reuse_data transitions: (e, t, ob, ac, rew, done, intra_steps, ob_next)
    for e in range(E):
        env = make_ref_env(env_name, seed=100 + e, max_episode_steps=P["max_episode_steps"])
        state = {"t": -1}
        if pusher and e < E // 2:
            def reset_at_seam(env=env, arm=seam_starts[e], orig=env.reset):
                orig()
                q = env.sim.data.qpos.copy()
                # towards the seam: the sign of the start equals the sign of this env's drive (even e: +, odd e: -)
                q[:4] = arm
                q[0] = abs(q[0]) * (1.0 if e % 2 == 0 else -1.0)
                env.set_state(q, env.sim.data.qvel.copy())
                return env._get_obs()
            env.reset = reset_at_seam

        def act(ob, is_train=True, return_stds=False, random_exploration=False, e=e, env=env, state=state):
            state["t"] += 1
            t = state["t"]
            st.begin(e, t)
            np.random.seed(1000 * e + t)                    # the reuse_data relabelling draws from the global numpy RNG
            out["qpos_start"][e, t] = env.sim.data.qpos
            out["ep_len_start"][e, t] = env._episode_length
            # the observation the policy sees, copied NOW: the Assembly env's obs dict holds views into the sim's site
            # arrays (`di["pegHead"] = self.sim.data.get_site_xpos(...)`, no copy -- as with mujoco-py), so the dict the runner
            # keeps as `prev_ob` shows the peg where the step took it by the time the transition is read
            fo = flat_ob(ob)
            if out["ob"] is None:
                out["ob"], out["ob_next"] = np.zeros((E, T, len(fo))), np.zeros((E, T, len(fo)))
            out["ob"][e, t] = fo
            if ik:
                return OrderedDict([("default", AC[e, t, :3].copy()), ("quat", AC[e, t, 3:].copy())]), None, None
            a = OrderedDict(default=AC[e, t].copy())
            if discrete:
                a["ac_type"] = np.array([int(AC_TYPE[e, t])])
            if (bool(AC_TYPE[e, t]) if discrete else agent.is_planner_ac(a)):      # will the runner's back-off move this step's target?  (it divides by np.linalg.norm,
                n = len(env.ref_joint_pos_indexes)    # a BLAS dot whose summation order is build-dependent: such steps are compared to round-off)
                tq = env.sim.data.qpos.copy()
                tq[env.ref_joint_pos_indexes] += agent.convert2planner_displacement(a["default"][:n], env._ac_scale)
                tq0 = tq.copy()
                tq = np.clip(tq, env._jnt_minimum[env.jnt_indices], env._jnt_maximum[env.jnt_indices])
                free = np.invert(env._is_jnt_limited[env.jnt_indices])      # (rl/mopa_rollouts.py:121-131: unlimited entries are restored)
                tq[free] = tq0[free]
                out["pulled_back"][e, t] = int(not agent.isValidState(tq))
            return a, None, None

        agent.act = act
        runner = object.__new__(MoPARolloutRunner)
        runner._config, runner._env, runner._env_eval, runner._pi = cfg, env, None, agent
        runner._ik_env = make_ref_env(env_name, seed=900 + e, max_episode_steps=P["max_episode_steps"]) if ik else None
        gen = runner.run(every_steps=1)
        prev_c = {k: 0 for k in COUNTERS}
        t_done = -1
        while True:
            calls = state["t"]
            if calls == T - 1 and t_done == T - 1 and not reuse:
                break
            try:
                if calls == T - 1 and t_done == T - 1:
                    # drain the relabelled transitions of the last step without starting step T
                    agent.act = lambda *a, **k: (_ for _ in ()).throw(StopIteration)
                batch, _ = next(gen)
            except (StopIteration, RuntimeError):
                break
            t = state["t"]
            ob0, ob1 = flat_ob(batch["ob"][0]), flat_ob(batch["ob"][-1])
            if t != t_done:                                 # first yield after an act(): the step's own transition
                t_done = t
                out["ob_next"][e, t] = ob1
                out["rew"][e, t], out["done"][e, t], out["intra"][e, t] = batch["rew"][0], int(batch["done"][0]), batch["intra_steps"][0]
                out["qpos_end"][e, t] = env.sim.data.qpos
                c = dict(gen.gi_frame.f_locals["counter"])
                if any(c[k] < prev_c[k] for k in COUNTERS):
                    prev_c = {k: 0 for k in COUNTERS}       # a new episode started: the counters were re-created
                out["counters"][e, t] = [c[k] - prev_c[k] for k in COUNTERS]
                prev_c = c
            else:                                           # further yields: relabelled sub-trajectories of that step
                extra.append((e, t, ob0, np.asarray(batch["ac"][0]["default"], dtype=np.float64), float(batch["rew"][0]),
                              int(batch["done"][0]), int(batch["intra_steps"][0]), ob1))
    tot = out["counters"].sum(axis=(0, 1))
    print(f"  rollout[{tag}]: counters", dict(zip(COUNTERS, tot.tolist())), "done", int(out["done"].sum()), "max intra", int(out["intra"].max()),
          "relabelled", len(extra))
    if reuse:
        out.update(x_env=np.array([x[0] for x in extra]), x_t=np.array([x[1] for x in extra]), x_ob=np.array([x[2] for x in extra]),
                   x_ac=np.array([x[3] for x in extra]), x_rew=np.array([x[4] for x in extra]), x_done=np.array([x[5] for x in extra]),
                   x_intra=np.array([x[6] for x in extra]), x_ob_next=np.array([x[7] for x in extra]))
    if discrete:
        out["ac_type"] = AC_TYPE
    save(f"ref_py_rollout_{tag}{'_reuse' if reuse else ''}{'_ik' if ik else ''}{'_discrete' if discrete else ''}.npz",
         params=np.array([P["timelimit"], P["max_nodes"], P["max_path"], P["seed"], P["max_episode_steps"], P["num_trials"]]), **out)


# ------------------------------------------------------------------------------------------------------------------
# section "episode": rl/mopa_rollouts.py:MoPARolloutRunner.run_episode (the evaluation loop, :401-678), env by env, scripted actions
# ------------------------------------------------------------------------------------------------------------------
def gen_episode(env_name="SawyerPushObstacle-v0", tag="push", E=24, discrete=False):
    """One whole episode per env through the reference's `run_episode` (max_step 10000, is_train=True): what it returns -- the rollout's
    rew / done lists, ep_info's len / rew / counters / episode_success -- plus the joint state at every policy call and at the end.
    `get_contact_force` (MuJoCo's contact solver, env/base.py:568) is set to 0: no kinematic counterpart."""
    import util.env as ref_util_env
    from rl.mopa_rollouts import MoPARolloutRunner
    ref_util_env.np = refshim.NumpyCompat()
    P = ROLLOUT_PARAMS
    cfg = make_config(env_name, timelimit=P["timelimit"], num_trials=P["num_trials"], discrete_action=discrete, stochastic_eval=False)
    n_ac = 8 if env_name == "SawyerLiftObstacle-v0" else 7
    agent, pi = make_agent(env_name, cfg, ac_dim=n_ac)
    st = Streams(agent, E, P["seed"], P["max_nodes"], P["max_path"])
    rng = np.random.default_rng(23)
    T = P["max_episode_steps"]                      # an agent step takes at least one env step
    AC = rng.uniform(-1, 1, size=(E, T, n_ac)) * rng.choice([0.5, 0.68, 0.9, 1.0], size=(E, T, 1))
    AC[: E // 3, 1::3, 1] = 1.0                     # far targets towards the table / bin: blocked lines, invalid targets, failed plans
    AC[: E // 3, 1::3, 3] = -1.0
    AC_TYPE = rng.integers(0, 2, size=(E, T)) if discrete else None
    nq = pi.model.nq
    out = dict(ac=AC, qpos_start=np.zeros((E, T, nq)), qpos_final=np.zeros((E, nq)), n_steps=np.zeros(E, dtype=np.int64),
               rew=np.zeros((E, T)), done=np.zeros((E, T), dtype=np.int64), pulled_back=np.zeros((E, T), dtype=np.int64),
               ep_len=np.zeros(E, dtype=np.int64), ep_rew=np.zeros(E), ep_success=np.zeros(E, dtype=np.int64),
               counters=np.zeros((E, len(COUNTERS)), dtype=np.int64), ob=None, ob_final=None)
    for e in range(E):
        env = make_ref_env(env_name, seed=300 + e, max_episode_steps=P["max_episode_steps"])
        env.get_contact_force = lambda: 0.0
        env.color_agent = env.reset_color_agent = lambda: None          # geom_rgba only (rendering)
        state = {"t": -1}

        def act(ob, is_train=True, return_stds=False, random_exploration=False, e=e, env=env, state=state):
            state["t"] += 1
            t = state["t"]
            st.begin(e, t)
            out["qpos_start"][e, t] = env.sim.data.qpos
            fo = flat_ob(ob)
            if out["ob"] is None:
                out["ob"], out["ob_final"] = np.zeros((E, T, len(fo))), np.zeros((E, len(fo)))
            out["ob"][e, t] = fo
            a = OrderedDict(default=AC[e, t].copy())
            if discrete:
                a["ac_type"] = np.array([int(AC_TYPE[e, t])])
            if (bool(AC_TYPE[e, t]) if discrete else agent.is_planner_ac(a)):
                n = len(env.ref_joint_pos_indexes)
                tq = env.sim.data.qpos.copy()
                tq[env.ref_joint_pos_indexes] += agent.convert2planner_displacement(a["default"][:n], env._ac_scale)
                tq = np.clip(tq, env._jnt_minimum[env.jnt_indices], env._jnt_maximum[env.jnt_indices])
                out["pulled_back"][e, t] = int(not agent.isValidState(tq))
            return a, None, None

        agent.act = act
        runner = object.__new__(MoPARolloutRunner)
        runner._config, runner._env, runner._env_eval, runner._pi, runner._ik_env = cfg, env, None, agent, None
        rollout, info, _frames = runner.run_episode(max_step=10000, is_train=True, record=False)
        n = state["t"] + 1
        # (the evaluation loop adds ob / ac to its rollout for planner steps only, :577-585 vs :648-653: the lists of rew / done are complete)
        assert len(rollout["rew"]) == n and len(rollout["done"]) == n
        out["n_steps"][e] = n
        out["rew"][e, :n], out["done"][e, :n] = rollout["rew"], np.asarray(rollout["done"], dtype=np.int64)
        out["ob_final"][e] = flat_ob(rollout["ob"][-1])
        out["qpos_final"][e] = env.sim.data.qpos
        out["ep_len"][e], out["ep_rew"][e] = info["len"], info["rew"]
        out["ep_success"][e] = int(info.get("episode_success", 0))
        out["counters"][e] = [info[k] for k in COUNTERS]
    print(f"  episode[{tag}]: agent steps / episode", out["n_steps"].tolist(), "len", out["ep_len"].tolist(), "counters",
          dict(zip(COUNTERS, out["counters"].sum(0).tolist())), "success", int(out["ep_success"].sum()), "pulled back", int(out["pulled_back"].sum()))
    if discrete:
        out["ac_type"] = AC_TYPE
    save(f"ref_py_episode_{tag}{'_discrete' if discrete else ''}.npz",
         params=np.array([P["timelimit"], P["max_nodes"], P["max_path"], P["seed"], P["max_episode_steps"], P["num_trials"]]), **out)


def gen_episodes():
    gen_episode()
    gen_episode("SawyerLiftObstacle-v0", "lift", E=16)


# ------------------------------------------------------------------------------------------------------------------
# section "ik": env/inverse_kinematics.py:qpos_from_site_pose (+ nullspace_method) on the reference env over FakeSim (f3)
# ------------------------------------------------------------------------------------------------------------------
def gen_ik():
    from env.inverse_kinematics import nullspace_method, qpos_from_site_pose
    rng = np.random.default_rng(5)
    out = {}
    # nullspace_method on its own (:274-281): random Jacobians / errors, 3 x 7 and 6 x 7
    Js, ds, xs = [], [], []
    for k in range(40):
        J = rng.normal(0, 0.5, (6 if k % 2 else 3, 7))
        d = rng.normal(0, 0.2, J.shape[0])
        Js.append(np.vstack([J, np.zeros((6 - J.shape[0], 7))])); ds.append(np.r_[d, np.zeros(6 - len(d))])
        xs.append(nullspace_method(J, d, regularization_strength=3e-2))
    out.update(ns_J=np.array(Js), ns_delta=np.array(ds), ns_rows=np.array([6 if k % 2 else 3 for k in range(40)]), ns_out=np.array(xs))
    for env_name, tag in (("SawyerAssemblyObstacle-v0", "assembly"), ("SawyerPushObstacle-v0", "push")):
        env = make_ref_env(env_name, seed=3)
        env.reset()
        K = 48
        nq = env.sim.model.nq
        q_in, tp, tq, use_q = np.zeros((K, nq)), np.zeros((K, 3)), np.zeros((K, 4)), np.zeros(K, dtype=np.int64)
        q_out, en, st, su = np.zeros((K, nq)), np.zeros(K), np.zeros(K, dtype=np.int64), np.zeros(K, dtype=np.int64)
        for k in range(K):
            env.reset()
            q = env.sim.data.qpos.copy()
            q[env.ref_joint_pos_indexes] += rng.normal(0, 0.15, 7)
            env.set_state(q, env.sim.data.qvel.copy())
            site = "grip_site"
            # targets as rl/mopa_rollouts.py:91-99 forms them: current site position + action_range * a, clipped to the world box
            scale = (0.02, 0.1, 0.5)[k % 3]
            target_pos = np.clip(env.sim.data.get_site_xpos(site) + scale * rng.uniform(-1, 1, 3), env.min_world_size, env.max_world_size)
            cur_q = np.zeros(4)
            refshim.mju_mat2Quat(cur_q, env.sim.data.get_site_xmat(site).ravel())
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            ang = (0.05, 0.4, 2.5)[(k // 3) % 3] * rng.uniform(0.5, 1.0)
            dq = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax]
            target_quat = np.zeros(4)
            refshim.mju_mulQuat(target_quat, cur_q, dq)
            use_q[k] = k % 4 != 0
            q_in[k], tp[k], tq[k] = q, target_pos, target_quat
            r = qpos_from_site_pose(env, site, target_pos=target_pos, target_quat=target_quat if use_q[k] else None,
                                    joint_names=env.robot_joints, max_steps=100, tol=1e-2)
            q_out[k], en[k], st[k], su[k] = np.array(r.qpos), r.err_norm, r.steps, int(r.success)
        print(f"  ik[{tag}]: success {int(su.sum())}/{K}  with quat {int(use_q.sum())}  steps max {int(st.max())} mean {st.mean():.1f}")
        out.update({f"{tag}_qpos": q_in, f"{tag}_target_pos": tp, f"{tag}_target_quat": tq, f"{tag}_use_quat": use_q,
                    f"{tag}_qpos_out": q_out, f"{tag}_err_norm": en, f"{tag}_steps": st, f"{tag}_success": su})
    save("ref_py_ik.npz", **out)


# ------------------------------------------------------------------------------------------------------------------
# section "env": the reference env classes' step / _step / compute_reward / _get_obs / _after_step over FakeSim (f1, N1)
# ------------------------------------------------------------------------------------------------------------------
def gen_env():
    from env.inverse_kinematics import qpos_from_site_pose
    for env_name, tag in (("SawyerPushObstacle-v0", "push"), ("SawyerLiftObstacle-v0", "lift"), ("SawyerAssemblyObstacle-v0", "assembly")):
        rng = np.random.default_rng(hash(tag) % 1000)
        E, T, MAXS = 10, 24, 24
        env0 = make_ref_env(env_name, seed=0)
        nq, adim = env0.sim.model.nq, env0.dof
        q0 = np.zeros((E, nq))
        act = np.zeros((E, T, adim)); is_pl = np.zeros((E, T), dtype=np.int64); fresh = np.zeros((E, T), dtype=np.int64)
        n_steps = np.zeros(E, dtype=np.int64)
        obs = None
        rew = np.zeros((E, T)); done = np.zeros((E, T), dtype=np.int64); succ = np.zeros((E, T), dtype=np.int64)
        q_after = np.zeros((E, T, nq)); obs0 = None
        for e in range(E):
            env = make_ref_env(env_name, seed=50 + e, max_episode_steps=MAXS)
            ob = env.reset()
            if e % 2 == 1:
                # start near the task's goal region so that the reward terms are exercised: the reference's own IK moves the
                # gripper to the object (push: above the cube; lift: around the can; assembly: peg head over the hole)
                d = env.sim.data
                if tag == "push":
                    tgt = d.body_xpos[env.cube_body_id] + np.array([-0.04, 0.0, 0.03]) + rng.normal(0, 0.01, 3)
                elif tag == "lift":
                    tgt = d.body_xpos[env.cube_body_id] + np.array([0.0, 0.0, 0.0 if e % 4 == 1 else 0.04]) + rng.normal(0, 0.004, 3)
                else:
                    tgt = d.get_site_xpos("grip_site") + (d.get_site_xpos("hole_bottom") - d.get_site_xpos("pegHead")) + rng.normal(0, 0.01, 3)
                for _ in range(6):      # a few restarts: the solver stops at tol 1e-3 or when it stalls
                    if tag == "assembly":   # the peg swings with the wrist: re-aim at the remaining head-to-hole offset
                        tgt = d.get_site_xpos("grip_site") + (d.get_site_xpos("hole_bottom") - d.get_site_xpos("pegHead")) + (e % 4 == 3) * rng.normal(0, 0.01, 3)
                    qpos_from_site_pose(env, "grip_site", target_pos=tgt, joint_names=env.robot_joints, max_steps=100, tol=1e-3)
                if tag == "lift":       # fingers wide open before closing in on the can
                    q = env.sim.data.qpos.copy()
                    q[env.ref_gripper_joint_pos_indexes] = [-0.0115, -0.0115] if e % 4 == 1 else [0.0208, 0.0208]
                    env.set_state(q, env.sim.data.qvel.copy())
                ob = env._get_obs()
            q0[e] = env.sim.data.qpos
            fo = flat_ob(ob)
            if obs is None:
                obs, obs0 = np.zeros((E, T, len(fo))), np.zeros((E, len(fo)))
            obs0[e] = fo
            for t in range(T):
                pl = int(rng.random() < 0.5)
                if pl:
                    a = rng.uniform(-0.07, 0.07, adim)          # a planner waypoint difference, sometimes beyond +-ac_scale
                else:
                    a = rng.uniform(-1.3, 1.3, adim)
                if adim == 8:
                    a[7] = rng.choice([-1.0, 1.0, rng.uniform(-0.01, 0.01)])      # gripper: bang-bang or a nudge
                if t % 5 == 0:
                    env._reset_prev_state()
                    fresh[e, t] = 1
                ob, r, dn, info = env.step(OrderedDict(default=a.copy()), is_planner=bool(pl))
                act[e, t], is_pl[e, t] = a, pl
                obs[e, t], rew[e, t], done[e, t], succ[e, t] = flat_ob(ob), r, int(dn), int(env._success)
                q_after[e, t] = env.sim.data.qpos
                n_steps[e] = t + 1
                if dn:
                    break
        print(f"  env[{tag}]: steps {int(n_steps.sum())} nonzero rewards {int((rew != 0).sum())} max reward {rew.max():.3f} "
              f"success {int(succ.sum())} done {int(done.sum())} obs_dim {obs.shape[2]}")
        save(f"ref_py_env_{tag}.npz", qpos0=q0, obs0=obs0, action=act, is_planner=is_pl, fresh_prev=fresh, n_steps=n_steps, obs=obs,
             reward=rew, done=done, success=succ, qpos_after=q_after, max_episode_steps=np.array(MAXS))


def gen_env_pusher():
    """PusherObstacleEnv.step on scripted direct / planner actions from set states (joint0 is unlimited: actions carry it across
    +-pi; the box sits near the fingertip / the target in some envs so that both reward terms and the success branch occur)."""
    rng = np.random.default_rng(77)
    E, T, MAXS = 12, 20, 20
    env0 = make_ref_env_pusher()
    nq = env0.sim.model.nq
    q0 = np.zeros((E, nq)); act = np.zeros((E, T, 4)); is_pl = np.zeros((E, T), dtype=np.int64); fresh = np.zeros((E, T), dtype=np.int64)
    n_steps = np.zeros(E, dtype=np.int64)
    obs = np.zeros((E, T, 20)); obs0 = np.zeros((E, 20))
    rew = np.zeros((E, T)); done = np.zeros((E, T), dtype=np.int64); succ = np.zeros((E, T), dtype=np.int64)
    q_after = np.zeros((E, T, nq))
    m = env0.sim._m
    box_base = np.asarray(m.body_pos[m.body_names.index("box")])[:2]
    tgt_base = np.asarray(m.body_pos[m.body_names.index("target")])[:2]
    for e in range(E):
        env = make_ref_env_pusher(seed=50 + e, max_episode_steps=MAXS)
        q = env.sim.data.qpos.copy()
        q[:4] = rng.uniform([-3.0, -2.0, -2.0, -2.0], [3.0, 2.0, 2.0, 2.0])
        if e % 4 == 3:
            q[0] = rng.choice([-3.1, 3.1])            # next to the seam of the unlimited joint
        env.set_state(q, env.sim.data.qvel.copy())
        tip = env.sim.data.get_site_xpos("fingertip")[:2].copy()
        if e % 2 == 1:                                # box next to the fingertip (slider value = world position - body base)
            q[-2:] = np.clip(tip + rng.normal(0, 0.03, 2) - box_base, -0.4, 0.4)
        else:
            q[-2:] = rng.uniform(-0.3, 0.3, 2)
        if e % 3 == 0:                                # goal next to the box: reward_push, and within the threshold for some
            q[-4:-2] = np.clip(q[-2:] + box_base - tgt_base + rng.normal(0, 0.04 if e % 2 else 0.02, 2), -0.4, 0.4)
        else:
            q[-4:-2] = rng.uniform(-0.3, 0.3, 2)
        env.set_state(q, env.sim.data.qvel.copy())
        env._after_reset()
        env._prev_state = None
        q0[e] = env.sim.data.qpos
        obs0[e] = flat_ob(env._get_obs())
        for t in range(T):
            pl = int(rng.random() < 0.5)
            a = rng.uniform(-0.12, 0.12, 4) if pl else rng.uniform(-1.0, 1.0, 4)
            if t % 5 == 0:
                env._reset_prev_state()
                fresh[e, t] = 1
            ob, r, dn, info = env.step(OrderedDict(default=a.copy()), is_planner=bool(pl))
            act[e, t], is_pl[e, t] = a, pl
            obs[e, t], rew[e, t], done[e, t], succ[e, t] = flat_ob(ob), r, int(dn), int(env._success)
            q_after[e, t] = env.sim.data.qpos
            n_steps[e] = t + 1
            if dn:
                break
    print(f"  env[pusher]: steps {int(n_steps.sum())} nonzero rewards {int((rew != 0).sum())} max reward {rew.max():.3f} "
          f"success {int(succ.sum())} done {int(done.sum())} |joint0| max {np.abs(q_after[:, :, 0]).max():.2f}")
    save("ref_py_env_pusher.npz", qpos0=q0, obs0=obs0, action=act, is_planner=is_pl, fresh_prev=fresh, n_steps=n_steps, obs=obs,
         reward=rew, done=done, success=succ, qpos_after=q_after, max_episode_steps=np.array(MAXS))


def gen_rollouts():
    gen_rollout()
    gen_rollout(E=12, T=4, reuse=True)
    gen_rollout("SawyerLiftObstacle-v0", "lift", E=24, T=5)
    gen_rollout("SawyerAssemblyObstacle-v0", "assembly", E=24, T=5)
    gen_rollout("SawyerAssemblyObstacle-v0", "assembly", E=24, T=5, ik=True)
    gen_rollout(E=24, T=5, discrete=True)
    gen_rollout_pusher()


def gen_rollout_pusher():
    gen_rollout("PusherObstacle-v0", "pusher", E=24, T=5)


SECTIONS = OrderedDict(host=gen_host, agent=gen_agent, rollout=gen_rollouts, ik=gen_ik, env=gen_env, env_pusher=gen_env_pusher, rollout_pusher=gen_rollout_pusher, episode=gen_episodes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sections", nargs="*", default=list(SECTIONS))
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for s in a.sections:
        print(f"[{s}]")
        SECTIONS[s]()


if __name__ == "__main__":
    main()
