#!/usr/bin/env python3
"""bench.py -- state-validity (collision-check) throughput of the HIP hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic planner
states: E envs x S states/env per GPU, SawyerPushObstacle-v0 (7-DoF arm, 27
collidable primitives, 241 non-ignored candidate pairs of which 86 are proven unreachable at scene compile time and dropped).  Inputs are resident
in HBM before the timed region.  One process per GPU; envs are sharded across
ranks (weak scaling: per-GPU work is fixed) and every step ends with an RCCL
all-gather of the uint8 validity masks, the only exchange the path has.
`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks
itself (re-exec under torch.distributed.run on 127.0.0.1); under an external
launcher WORLD_SIZE must equal N -- anything else is an error, not a 1-GPU run.

Prints ONE JSON line (rank 0) following the driver's contract, extended with
  roofline      achieved algorithmic GB/s of the validity kernel vs the HBM roof
  cpu_baseline  the CPU oracle (this repo's C restatement -- NOT MuJoCo/OMPL)
                timed on this box's host cores on the same states
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENV = os.environ.get("MOPA_BENCH_ENV", "SawyerPushObstacle-v0")      # (MOPA_BENCH_ENV: the per-scene K1 profiles of tools/profile.sh; the headline is Push)
# algorithmic bytes per validity check = 7 f64 joint values read + 1 verdict byte
# written + the env's qpos row (nq f64) amortised over the S states that share it.
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6    # AMD datasheet vector FP64 (not in the local guide)


def host_cores():
    """CPU threads this process may really use: min(visible CPUs, cgroup v2 quota) -- the GPU boxes show 256 CPUs
    but run the container under a 16-CPU quota, and oversubscribed OpenMP teams are slower than 1 thread."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def make_inputs(torch, pi, E, S, seed, device, mode="mixed", env=None):
    """States for E envs x S states: first half of every env's block uniform in the joint box (what the RRT
    sampler draws), second half near the env's initial pose (what motion validation sees).  Per-env passive
    block: gripper slides uniform in their joint range, the free-jointed object (cube / can / furniture) at its nominal pose
    + U(+-0.05) in xy."""
    from mopa_rl_amd.mjcf import JNT_FREE
    from mopa_rl_amd.scene import default_qpos
    env = env or pi.spec.env
    m = pi.model
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    na = len(pi.ref_joint_pos_indexes)
    lo = torch.tensor(pi.jnt_minimum, dtype=torch.float64, device=device)
    hi = torch.tensor(pi.jnt_maximum, dtype=torch.float64, device=device)
    q0 = torch.tensor(default_qpos(env, m), dtype=torch.float64, device=device)
    u = torch.rand(E, S, na, generator=g, dtype=torch.float64, device=device)
    qa = lo + (hi - lo) * u
    n = torch.randn(E, S, na, generator=g, dtype=torch.float64, device=device) * 0.3
    near = torch.minimum(torch.maximum(q0[pi.ref_joint_pos_indexes] + n, lo), hi)
    half = {"mixed": S // 2, "near": 0, "uniform": S}[mode]
    qa[:, half:, :] = near[:, half:, :]
    rows = q0.repeat(E, 1)
    for name in ("rc_close", "lc_close"):
        if name in m.jnt_names:
            j = m.joint_name2id(name)
            a, (r0, r1) = int(m.jnt_qposadr[j]), m.jnt_range[j]
            rows[:, a] = r0 + (r1 - r0) * torch.rand(E, generator=g, dtype=torch.float64, device=device)
    free = [int(m.jnt_qposadr[j]) for j in range(len(m.jnt_names)) if m.jnt_type[j] == JNT_FREE]
    if free:
        rows[:, free[0]:free[0] + 2] += -0.05 + 0.1 * torch.rand(E, 2, generator=g, dtype=torch.float64, device=device)
    return qa.reshape(E * S, na).contiguous(), rows.contiguous()


def planner_queries(torch, bp, pi, E, device):
    """start = init_qpos + N(0, 0.02) (as `_reset`, env/sawyer/sawyer_push_obstacle.py:36-41), goal = a valid state
    with |dq|_inf <= 0.5 (action_range)."""
    from mopa_rl_amd.scene import default_qpos
    g = torch.Generator(device=device)
    g.manual_seed(99)
    q0 = torch.tensor(default_qpos(ENV, pi.model), dtype=torch.float64, device=device)
    lo = torch.tensor(pi.jnt_minimum, dtype=torch.float64, device=device)
    hi = torch.tensor(pi.jnt_maximum, dtype=torch.float64, device=device)
    start = q0.repeat(E, 1)
    start[:, :7] += 0.02 * torch.randn(E, 7, generator=g, dtype=torch.float64, device=device)
    # rejection-sample goals: 8 candidates per env, keep the first valid one (else the start itself)
    C = 8
    cand = start[:, None, :7] + (torch.rand(E, C, 7, generator=g, dtype=torch.float64, device=device) - 0.5)
    cand = torch.minimum(torch.maximum(cand, lo), hi).reshape(E * C, 7).contiguous()
    ok = bp.is_valid(cand, start.contiguous(), samples_per_env=C).reshape(E, C).bool()
    first = torch.argmax(ok.int(), dim=1)
    goal = start.clone()
    pick = cand.reshape(E, C, 7)[torch.arange(E, device=device), first]
    goal[:, :7] = torch.where(ok.any(dim=1, keepdim=True), pick, start[:, :7])
    return start, goal


def plan_cpu_baseline(pi, start_h, goal_h, prm, status, plen, nchk, n_sample=1024):
    """The oracle's RRT-Connect (kind="port") on the first n_sample queries: one thread, then OpenMP over queries inside the
    oracle (orc_plan_batch, dynamic schedule) on the cores this process may use.  Also the parity check of those queries."""
    from oracle import oracle as O
    orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    n = min(n_sample, len(start_h))
    kw = dict(max_iters=prm["max_iters"], max_nodes=prm["max_nodes"], seed=prm["seed"], max_path=prm["max_path"])
    n1 = min(n, 256)
    t0 = time.perf_counter()
    orc.plan_batch(start_h[:n1], goal_h[:n1], pi.spec.range, 0.005, nthreads=1, **kw)
    t1 = time.perf_counter()
    cores = host_cores()
    st, ln, chk = orc.plan_batch(start_h[:n], goal_h[:n], pi.spec.range, 0.005, nthreads=cores, **kw)
    t2 = time.perf_counter()
    mism = int(((st != np.asarray(status[:n])) | (ln != np.asarray(plen[:n])) | (chk != np.asarray(nchk[:n]))).sum())
    return {"value": n / (t2 - t1), "unit": "plans/s", "cores": cores, "kind": "port", "single_thread_value": n1 / (t1 - t0),
            "sample": f"the first {n} of the same queries through the oracle's orc_plan_batch (same sample streams; OpenMP over queries, dynamic "
                      f"schedule, {cores} threads; single thread: the first {n1}); our C restatement, not OMPL",
            "parity_mismatches_vs_oracle": mism}


def plan_section(torch, bp, pi, E, device, with_cpu=True):
    """BASELINE.json configs[2]: E envs, each one RRT-Connect query (K3, one wave per env).  Reported next to, not
    inside, the headline metric."""
    import time as _t
    start, goal = planner_queries(torch, bp, pi, E, device)
    prm = dict(max_iters=2000, max_nodes=4096, max_path=256, seed=7)
    bp.plan(start, goal, **prm)
    torch.cuda.synchronize()
    reps = 3
    t0 = _t.perf_counter()
    for _ in range(reps):
        path, plen, status, nchk = bp.plan(start, goal, **prm)
    torch.cuda.synchronize()
    dt = (_t.perf_counter() - t0) / reps
    out = {"config": f"{ENV}, {E} envs, one RRT-Connect query each (range {pi.spec.range}, resolution 0.005, "
                     f"{prm['max_iters']} iterations, {prm['max_nodes']} nodes/tree)",
           "plans_per_s": E / dt, "ms_per_batch": dt * 1e3, "consumed_checks_per_s": float(nchk.sum().item()) / dt,
           "success_rate": float((status == 0).float().mean().item()), "mean_path_len": float(plen.float().mean().item()),
           "mean_checks_per_plan": float(nchk.float().mean().item())}
    # A launch lasts as long as its slowest query while most of the chip idles; launches on different streams
    # overlap (per-stream scratch), which is how the asynchronous rollouts use the planner: 4 launches of E queries each
    # (own sample streams: different seeds), each capped to a quarter of the CUs
    from mopa_rl_amd.rollout import _side_streams
    streams = _side_streams(device, 2) + [torch.cuda.Stream(device=device), torch.cuda.current_stream()]
    nl = len(streams)

    def burst():
        for i, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                bp.plan(start, goal, **dict(prm, seed=7 + 13 * i), stream=st, max_workgroups=max(1, torch.cuda.get_device_properties(device).multi_processor_count // nl), exclusive=True)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    burst()
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    burst()
    torch.cuda.synchronize()
    dtc = _t.perf_counter() - t0
    out["concurrent"] = {"launches": nl, "queries": nl * E, "ms_total": dtc * 1e3, "plans_per_s": nl * E / dtc,
                         "note": f"{nl} launches of {E} queries on {nl} streams, each capped to 1/{nl} of the CUs (one workgroup per CU in total: the burst ends with its slowest queries)"}
    # ... and a stream of batches through the iteration ladder (BatchPlanner.plan_laddered): 200 iterations first, the ~3 % it
    # does not solve again with all 2000 on other streams while the next batches' first launches run -- results identical
    # to full-budget launches (tests/test_gpu_parity.py::test_laddered_planning_equals_one_full_launch)
    nb = 24         # (the last retry launch drains alone for one straggler's latency, ~37 ms: the rate grows with the stream's length -- `laddered_long`)
    batches = [dict(start=start, goal=goal, seed=7 + 13 * i) for i in range(nb)]
    # scheduling knobs (results do not depend on them): a first rung of 100 iterations and retry launches of >= 1024 pooled queries
    # measured best over long streams (tools/ladder_grid.py, tools/ladder_nb.py: 534 k plans/s at 24 batches, 629 k at 64; round 4's
    # 200 / 512: 477 k / 529 k); at 24 batches the rate also depends on where the last retry launch lands (110-120 / 1024 / 3-4 streams: 570-587 k)
    lad_kw = dict(max_iters=prm["max_iters"], first_iters=int(os.environ.get("MOPA_LADDER_FIRST_ITERS", "100")), max_nodes=prm["max_nodes"],
                  max_path=prm["max_path"], retry_min=int(os.environ.get("MOPA_LADDER_RETRY_MIN", "1024")),
                  first_stream=streams[0], retry_streams=streams[1:3], retry_exclusive=bool(int(os.environ.get("MOPA_LADDER_EXCL", "0"))))
    bp.plan_laddered(batches, **lad_kw)        # untimed: per-stream scratch (trees of the pooled retry launches) grows to its final size
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    lad = bp.plan_laddered(batches, **lad_kw)
    torch.cuda.synchronize()
    dtl = _t.perf_counter() - t0
    same = all(bool((a == b).all().item()) for a, b in zip(lad[0][1:], (plen, status, nchk)))
    out["laddered"] = {"batches": nb, "queries": nb * E, "ms_total": dtl * 1e3, "ms_per_batch": dtl * 1e3 / nb, "plans_per_s": nb * E / dtl,
                       "first_batch_equals_full_launch": same,
                       "note": f"{nb} batches of {E} queries: first launch with {lad_kw['first_iters']} iterations, unsolved queries (pooled, >= {lad_kw['retry_min']}) again with {prm['max_iters']} on 2 other streams"}
    nbl = 64        # the same ladder over a longer stream: the final drain weighs less
    long_batches = [dict(start=start, goal=goal, seed=7 + 13 * i) for i in range(nbl)]
    bp.plan_laddered(long_batches, **lad_kw)
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    bp.plan_laddered(long_batches, **lad_kw)
    torch.cuda.synchronize()
    dtl2 = _t.perf_counter() - t0
    out["laddered_long"] = {"batches": nbl, "queries": nbl * E, "ms_total": dtl2 * 1e3, "plans_per_s": nbl * E / dtl2}
    if with_cpu:
        out["cpu_baseline"] = plan_cpu_baseline(pi, start.cpu().numpy(), goal.cpu().numpy(), prm, status.cpu().numpy(),
                                                plen.cpu().numpy(), nchk.cpu().numpy())
    return out


def motion_section(torch, bp, pi, qa, rows, S, device):
    """A5: OMPL DiscreteMotionValidator (K2) on the first 262 144 states of the step batch as segment starts; segment =
    a random direction of L1 length `range` (one RRT extension, ~1.5 states) or 5 x range."""
    n = min(len(qa), 1 << 18)
    n -= n % S
    a = qa[:n].contiguous()
    g = torch.Generator(device=device)
    g.manual_seed(2)
    lo = torch.tensor(pi.jnt_minimum, dtype=torch.float64, device=device)
    hi = torch.tensor(pi.jnt_maximum, dtype=torch.float64, device=device)
    out = {"config": f"{ENV}, {n} segments per launch, resolution 0.005 (KinematicPlanner.cpp:87)"}
    for name, step in (("range", pi.spec.range), ("5x_range", 5 * pi.spec.range)):
        d = torch.randn(n, a.shape[1], generator=g, dtype=torch.float64, device=device)
        d = d / d.abs().sum(dim=1, keepdim=True) * step
        b = torch.minimum(torch.maximum(a + d, lo), hi).contiguous()
        for _ in range(2):
            bp.check_motion(a, b, rows, samples_per_env=S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            v = bp.check_motion(a, b, rows, samples_per_env=S)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out[f"motions_per_s_{name}"] = n / dt
        out[f"valid_fraction_{name}"] = float(v.float().mean().item())
    return out


def env_step_section(torch, E, device, steps, with_cpu):
    """The "env-steps/sec" half of BASELINE.json's metric: E kinematic Sawyer envs per task (K4 `k_env_step`),
    KINEMATIC -- the physics of the reference env.step is replaced by its kinematic limit (mopa_rl_amd/kinematic_env.py),
    so this is NOT dynamics parity.  Push (the headline env): the bare step, the step gated by the planner's validity rule on
    the desired state (block_invalid: one K1 launch + one K4 launch + torch glue per step), parity + CPU baseline against the
    oracle.  Lift (grasp test: finger boxes vs the can's hull) and Assembly: the bare step."""
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.scene import planner_inputs
    out = {"label": "kinematic limit of the position servos; NOT dynamics parity (no contact forces, the manipulated object never moves)"}
    g = torch.Generator(device=device)
    g.manual_seed(5)
    for env_name, tag in (("SawyerPushObstacle-v0", "push"), ("SawyerLiftObstacle-v0", "lift"), ("SawyerAssemblyObstacle-v0", "assembly")):
        pi = planner_inputs(env_name)
        blk = {"config": f"{env_name} kinematic env.step, {E} envs, uniform policy actions in [-1,1] (ac_scale {pi.spec.ac_scale})"}
        for key, block in (("steps_per_s", False), ("steps_per_s_collision_gated", True)):
            if block and tag != "push":
                continue
            env = make_env(env_name, E, device=device, seed=11, block_invalid=block, max_episode_steps=1 << 30)
            acts = (torch.rand(steps + 2, E, env.action_dim, generator=g, dtype=torch.float64, device=device) * 2 - 1).contiguous()
            env.reset()
            q_init = env.qpos.clone()
            env.step(acts[0]); env.step(acts[1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(steps):
                env.step(acts[2 + t])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            blk[key] = E * steps / dt
            blk[key.replace("steps_per_s", "us_per_batch")] = dt / steps * 1e6
            if not block and with_cpu and tag == "push":
                # parity + CPU baseline: replay the same rollout through the CPU checker from the same reset state
                from oracle import oracle as O
                orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
                ref = O.OracleEnv(orc, env.facts, E, ac_scale=env.ac_scale, max_episode_steps=1 << 30)
                ref.set_state(q_init.cpu().numpy())
                a_host = acts.cpu().numpy()
                cores = host_cores()
                t0 = time.perf_counter()
                for t in range(steps + 2):
                    ref.step(a_host[t], nthreads=cores)
                dt_cpu = time.perf_counter() - t0
                blk["parity_mismatches_vs_oracle"] = int((env.obs.cpu().numpy().view(np.uint64) != ref.obs.view(np.uint64)).sum()
                                                         + (env.reward.cpu().numpy().view(np.uint64) != ref.reward.view(np.uint64)).sum())
                t0 = time.perf_counter()
                ref.step(a_host[0], nthreads=1)
                dt_1 = time.perf_counter() - t0
                blk["cpu_baseline"] = {"value": E * (steps + 2) / dt_cpu, "unit": "env-steps/s", "cores": cores, "kind": "port",
                                       "single_thread_value": E / dt_1,
                                       "sample": f"the same {steps + 2} x {E} steps through the oracle's orc_env_step_batch "
                                                 "(OpenMP over envs); our C restatement, not MuJoCo"}
        blk["dynamics"] = env_dynamics_block(torch, env_name, E, device, g, with_cpu and tag == "push")
        blk["dynamics_contacts"] = env_dynamics_block(torch, env_name, E, device, g, with_cpu, contacts=True, steps=6)
        if tag == "push":
            # K6 is latency-bound at BASELINE's 4096 envs (64 of 256 CUs hold its waves): its rate with the chip filled, next to it
            sat = env_dynamics_block(torch, env_name, 4 * E, device, g, False, steps=10)
            blk["dynamics_saturated"] = {"envs": 4 * E, "steps_per_s": sat["steps_per_s"], "ms_per_batch": sat["ms_per_batch"]}
        out[tag] = blk
    out["steps_per_s"] = out["push"]["steps_per_s"]
    out["steps_per_s_dynamics"] = out["push"]["dynamics"]["steps_per_s"]
    out["steps_per_s_dynamics_contacts"] = out["push"]["dynamics_contacts"]["steps_per_s"]
    return out


def env_dynamics_block(torch, env_name, E, device, g, with_cpu, steps=20, contacts=False):
    """env.step with `_do_simulation` = the reference's 75 sub-steps of force-limited position servos + gravity compensation
    on the arm's own tree (K6 `k_env_dyn4`, SURVEY.md 8 f4b stage A); contact-free: the manipulated object does not move.
    contacts=True (stage C, K7 `k_env_dyn_ct`): contacts of the arm, of the manipulated object (free rigid body) and between
    the two behind one soft-constraint solve per sub-step -- restated from MuJoCo's published solver, parity unpinned."""
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.scene import planner_inputs
    env = make_env(env_name, E, device=device, seed=11, dynamics=True, contacts=contacts, max_episode_steps=1 << 30)
    acts = (torch.rand(steps + 2, E, env.action_dim, generator=g, dtype=torch.float64, device=device) * 2 - 1).contiguous()
    env.reset()
    q_init = env.qpos.clone()
    env.step(acts[0]); env.step(acts[1])
    torch.cuda.synchronize()
    q_warm = [x.clone() for x in (env.qpos, env.qvel, env.bias_lag, env.prev_state, env.has_prev, env.ep_len)]
    # the same `steps` steps twice from the same state (the GPU boxes are shared: a pass next to another tenant's burst takes 2-3x as
    # long); both times are reported, the rate is the faster pass's
    passes = []
    settle()
    for rep in range(2):
        for dst, src in zip((env.qpos, env.qvel, env.bias_lag, env.prev_state, env.has_prev, env.ep_len), q_warm):
            dst.copy_(src)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for t in range(steps):
            env.step(acts[2 + t])
        ev1.record()
        torch.cuda.synchronize()
        passes.append((time.perf_counter() - t0, ev0.elapsed_time(ev1)))
    unsettle()
    dt, gpu_ms = min(passes)
    nd, nsub = env.dyn.nd, env.dyn.nsub
    bytes_per_step = 6 * nd * 8 + 2 * env.action_dim * 8 + env.obs_dim * 8 + 18       # q / qvel / lagged bias in+out, action, prev_state, obs, flags
    what = (f"servo + contacts (stage C): {len(env.ct.pr_f)} directed geom pairs ({len(env.ct.ft_rad)} feature points in exact signed-distance functions), "
            f"<= {env.ct.maxcon} contacts per env, MuJoCo's solref / solimp impedance model, "
            f"{['Gauss-Seidel + pyramidal cones', 'Newton + pyramidal cones', 'Newton + ELLIPTIC cones (the XML model)'][env.ct.solver]}, <= {env.ct.iterations} iterations at tolerance "
            f"{env.ct.tolerance:g}, noslip pass of {env.ct.noslip_iterations} sweeps, limit rows {env.ct.limit_rows} -- RESTATED FROM THE PUBLISHED SOLVER, PARITY UNPINNED; "
            f"{env.ct.condim_downgraded} pairs ask for condim 4 / 6 and are solved as condim 3") if contacts else (
           f"servo, contact-free: {nsub} sub-steps of h = {env.dyn.timestep} s per env.step on {nd} dofs (RNE bias + CRB inertia + "
           "implicit-damping Euler, kp / forcerange servos, lagged qfrc_bias as gravity compensation); joint limits = inelastic "
           "stop; the manipulated object does not move (no contacts) -- labelled, NOT MuJoCo's constraint solver")
    blk = {"dynamics": what,
           "config": f"{env_name} env.step with servo dynamics, {E} envs, uniform policy actions in [-1,1]",
           "steps_per_s": E * steps / dt, "ms_per_batch": dt / steps * 1e3, "substeps_per_s": E * steps * nsub / dt,
           "gpu_ms_per_batch": gpu_ms / steps, "ms_per_batch_passes": [p[0] / steps * 1e3 for p in passes],
           "algorithmic_bytes_per_env_step": bytes_per_step, "achieved_GBps": E * steps * bytes_per_step / dt / 1e9,
           "bound": ("latency: 16 lanes per env, one wave per SIMD (1024 workgroups of 4 envs): chain walk + contact culling / narrow phase + "
                     "constraint rows + solver iterations of a sub-step; not HBM") if contacts else
                    "latency of the serial chain walk + 9 x 9 solve of a sub-step (four waves share 64 envs; one workgroup per CU: 64 of 256 "
                    "CUs busy at 4096 envs); not HBM"}
    if contacts:
        from mopa_rl_amd import _lib
        stats = torch.zeros(E, 4, dtype=torch.int32, device=device)
        _lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, stats.data_ptr()))
        env.step(acts[2])
        torch.cuda.synchronize()
        st = stats.cpu().numpy().astype(np.float64)
        blk["contacts_per_substep"] = float(st[:, 0].sum() / (E * nsub))
        blk["solver_sweeps_per_substep"] = float(st[:, 1].sum() / (E * nsub))
        blk["contacts_dropped_by_the_cap_per_substep"] = float(st[:, 2].sum() / (E * nsub))
        _lib.check(_lib.lib().mopa_env_set_contact_stats(env._h, None))
    if with_cpu:
        from oracle import oracle as O
        pi = planner_inputs(env_name)
        orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
        n = min(E, 256 if contacts else 1024)
        k = 3 if contacts else 4
        ref = O.OracleEnv(orc, env.facts, n, ac_scale=env.ac_scale, max_episode_steps=1 << 30, dyn=env.dyn, obj=env.obj, ct=env.ct)
        ref.set_state(q_init[:n].cpu().numpy())
        a_host = acts[:, :n].cpu().numpy()
        cores = host_cores()
        t0 = time.perf_counter()
        for t in range(k):
            ref.step(a_host[t], nthreads=cores)
        dt_cpu = time.perf_counter() - t0
        # parity: the GPU envs after the same k steps from the same reset state
        chk = make_env(env_name, n, device=device, seed=11, dynamics=True, contacts=contacts, max_episode_steps=1 << 30)
        chk.set_state(q_init[:n].clone())
        for t in range(k):
            chk.step(acts[t, :n].contiguous())
        torch.cuda.synchronize()
        blk["parity_mismatches_vs_oracle"] = int((chk.obs.cpu().numpy().view(np.uint64) != ref.obs.view(np.uint64)).sum()
                                                 + (chk.qvel.cpu().numpy().view(np.uint64) != ref.qvel.view(np.uint64)).sum()
                                                 + (chk.qpos.cpu().numpy().view(np.uint64) != ref.qpos.view(np.uint64)).sum())
        one = O.OracleEnv(orc, env.facts, 64, ac_scale=env.ac_scale, max_episode_steps=1 << 30, dyn=env.dyn, obj=env.obj, ct=env.ct)
        one.set_state(q_init[:64].cpu().numpy())
        t0 = time.perf_counter()
        one.step(a_host[0][:64], nthreads=1)
        dt_1 = time.perf_counter() - t0
        blk["cpu_baseline"] = {"value": n * k / dt_cpu, "unit": "env-steps/s", "cores": cores, "kind": "port",
                               "single_thread_value": 64 / dt_1,
                               "sample": f"the first {n} envs x {k} steps of the same rollout through the oracle's orc_env_step_dyn_batch "
                                         "(OpenMP over envs); our C restatement of the mj_step pipeline, not MuJoCo"}
        chk.close()
    env.close()
    return blk


SAC_GRAD_FLOATS = 145678 + 2 * 78337       # actor + two critics for obs 40 / ac 7 (the payload of sync_grads)


def rollout_section(torch, env_name, E, device, agent_steps, world=1, async_planner=False, use_ik=False, use_graphs=False, dynamics=False):
    """End-to-end MoPA rollout step (mopa_rl_amd/rollout.py), SURVEY 8d configs 3 / 4: a SAC actor (stock PyTorch, random-init
    obs-256-256-256-(2 x ac) MLP, f32; a = tanh(mu + sigma * eps)) samples the action from the obs inside the timed loop,
    then per env either a direct env step or target / pull-back / straight-line pre-check / RRT-Connect / densification /
    waypoint execution on the kinematic env.  Host-orchestrated; every check / plan / env step a batched launch.
    Every agent step ends with the exchanges a data-parallel run needs (no-ops at world 1): the asynchronous RCCL all-gather
    of the step's transition records (dist.TransitionExchange, overlapped with the next step) and an all-reduce of a
    gradient-sized f32 buffer (what `sync_grads` moves per SAC update, reference util/pytorch.py:153-159)."""
    import torch.distributed as dist
    from mopa_rl_amd.dist import TransitionExchange, all_reduce_mean_
    from mopa_rl_amd.kinematic_env import make_env
    from mopa_rl_amd.rollout import BatchMoPARollout, RolloutConfig
    env = make_env(env_name, E, device=device, seed=21 + int(os.environ.get("RANK", "0")),
                   **({"dynamics": True, "contacts": True} if dynamics else {}))
    env.reset()
    over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("MOPA_BENCH_ROLLOUT", "").split(",") if kv)}   # A/B knob
    if world > 1:
        # main stream + planner side streams + RCCL's stream: the HIP runtime runs 4 hardware queues side by side (streams
        # beyond that share one and serialise), so one planner stream less than the single-GPU default
        over.setdefault("planner_streams", 2)
    if use_graphs:
        over["use_graphs"] = 1          # the fixed-shape halves of a call replayed from HIP graphs (rollout.py)
        # (with replayed calls the host submits faster than three 64-workgroup planner streams drain: 2 x 128 measured steadier,
        #  tools/rollout_ab.sh)
        over.setdefault("planner_streams", 2)
        over.setdefault("planner_workgroups", 128)
    if dynamics:
        # a waypoint is a 75-sub-step physics launch whose time does not depend on how many envs take part: every call advances
        # every env on a path by one waypoint (in the launch that also carries the direct steps), instead of draining towards the
        # longest path of the call (walk_chunk 0: 7.1 k agent steps/s; 1: 23.9 k; 2: 23.5 k; 4: 24.0 k -- profiles/r04)
        over.setdefault("walk_chunk", 1)
    if use_ik:
        over["use_ik_target"] = 1       # MoPA + IK action space (BASELINE config 5): Cartesian displacement + rotation quaternion
    rank = int(os.environ.get("RANK", "0")) if world > 1 else 0
    ro = BatchMoPARollout(env, RolloutConfig.for_env(env_name, async_planner=async_planner, env_id_base=rank * E, env_id_total=world * E, **over))
    if os.environ.get("MOPA_BENCH_PHASES"):
        ro.timing = {}           # per-phase times (each mark synchronises the main stream: slower calls, profiling only)
    torch.manual_seed(8)
    nn = torch.nn
    ad = ro.ac_dim
    actor = nn.Sequential(nn.Linear(env.obs.shape[1], 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                          nn.Linear(256, 2 * ad)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(8)
    tx = TransitionExchange(E, env.obs_dim, ad, device)
    grads = torch.zeros(SAC_GRAD_FLOATS, dtype=torch.float32, device=device)

    def act():
        with torch.no_grad():
            mu, log_std = actor(env.obs.float()).chunk(2, dim=1)
            eps = torch.randn(E, ad, generator=g, dtype=torch.float32, device=device)
            return torch.tanh(mu + torch.exp(log_std.clamp(-10.0, 2.0)) * eps).double()

    def one(k):
        a = act()
        out = ro.agent_step(a)
        tx.pack(k, out["ob"], out["ac"], out["rew"], out["done"], out["intra_steps"], out["ob_next"], stepped=out["stepped"])
        tx.launch(k)
        all_reduce_mean_(grads)
        return out
    # untimed calls first: one for lock-step; the asynchronous mode needs ~25 until planner launches start, run and finish
    # at their steady rate
    warm = int(os.environ.get("MOPA_BENCH_ROLLOUT_WARM", "25")) if async_planner else 1
    for k in range(warm):
        out = one(k)
        env.reset(out["done"].bool() & out["stepped"])
    tx.drain()
    settle()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    n_acc = torch.zeros(2, dtype=torch.int64, device=device)      # agent steps, env steps (read back once, after the loop)
    t0 = time.perf_counter()
    for t in range(agent_steps):
        _c0 = time.perf_counter()
        out = one(warm + t)
        if os.environ.get("MOPA_BENCH_TRACE"):
            torch.cuda.current_stream().synchronize()
            print(f"[rollout {env_name} async={async_planner}] call {t}: {(time.perf_counter() - _c0) * 1e3:.1f} ms, jobs in flight {len(ro._jobs)}", file=sys.stderr)
        st = out["stepped"]
        n_acc[0] += st.sum()
        n_acc[1] += ((out["intra_steps"] + 1) * st).sum()
        env.reset(out["done"].bool() & st)
    if os.environ.get("MOPA_BENCH_PHASES") and getattr(ro, "timing", None):
        print(f"[rollout {env_name} async={async_planner}] phases ms/call:", {k: round(v / (warm + agent_steps) * 1e3, 3) for k, v in ro.timing.items()}, file=sys.stderr)
    gathered = tx.result(warm + agent_steps - 1)
    tx.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    unsettle()
    n_agent_steps, n_env_steps = int(n_acc[0].item()), int(n_acc[1].item())
    c = {k: int(v.sum().item()) for k, v in ro.counters.items()}
    if world > 1:
        t = torch.tensor([dt, float(n_env_steps), float(n_agent_steps)] + [float(c[k]) for k in sorted(c)], dtype=torch.float64, device=device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, n_env_steps, n_agent_steps = float(tmax[0].item()), int(t[1].item()), int(t[2].item())
        c = {k: int(t[3 + i].item()) for i, k in enumerate(sorted(c))}
    mode = ("async_planner: RRT-Connect on side streams, envs waiting for a query sit out (each env's transitions are those of the "
            "lock-step run)") if async_planner else "lock-step: every call waits for its slowest RRT-Connect query"
    if use_graphs:
        mode += "; HIP graphs: action -> target -> pull-back -> pre-check, and execution + bookkeeping, each replayed as one graph launch"
    if use_ik:
        mode += "; IK action space: the actor's Cartesian displacement + quaternion -> joint displacement through the batched damped-LS IK (K5)"
    return {"config": f"{env_name}, {E} envs per GPU x {world} GPU(s), {agent_steps} calls of agent_step; actions sampled by a random-init SAC actor "
                      f"({env.obs.shape[1]}-256-256-256-{2 * ad} MLP, f32, tanh-Gaussian) from the obs (omega 0.7), {'DYNAMICS env (servo dynamics + contacts behind the constraint solver, K7: every env.step is 75 sub-steps; walk_chunk ' + str(ro.cfg.walk_chunk) + ': envs on a path advance one waypoint per call and sit out the policy until it ends)' if dynamics else 'kinematic env'}; {mode}",
            "agent_steps_per_s": n_agent_steps / dt, "env_steps_per_s": n_env_steps / dt, "s_per_agent_step_batch": dt / agent_steps,
            "envs_stepping_per_call": n_agent_steps / (world * agent_steps), "counters": c,
            "exchange": {"transition_record_bytes": tx.width * 4, "all_gather_bytes_per_rank_per_step": tx.bytes_per_step,
                         "gathered_rows": int(gathered["rew"].shape[0]), "grad_all_reduce_bytes": SAC_GRAD_FLOATS * 4,
                         "collectives": "RCCL all_gather_into_tensor (async, double-buffered) + all_reduce(SUM)/world" if world > 1 else "none (1 rank)"}}


def ik_section(torch, device, E=8192):
    """BASELINE.json configs[4]'s IK piece: E damped-LS IK problems (env/inverse_kinematics.py:18-135 restated, K5) on
    SawyerAssemblyObstacle with POSITION + ORIENTATION targets, as the reference's IK action space poses them
    (rl/trainer.py:106-110 adds "quat" for 3-D envs, rl/mopa_rollouts.py:690-705 then always passes target_quat):
    target = grip-site pose of a perturbed arm configuration (reachable by construction), max_steps 100, tol 1e-2.
    The position-only form is timed next to it."""
    from mopa_rl_amd.ik import BatchIK
    from mopa_rl_amd.scene import ENV_SPECS, default_qpos, load_scene
    env = "SawyerAssemblyObstacle-v0"
    m = load_scene(ENV_SPECS[env].scene)
    ik = BatchIK(m, "grip_site", ENV_SPECS[env].robot_joints, device=device.index if device.index is not None else -1)
    g = torch.Generator(device=device)
    g.manual_seed(0)
    q0 = torch.tensor(default_qpos(env, m), device=device).repeat(E, 1)
    q0[:, :7] += 0.2 * torch.randn(E, 7, generator=g, dtype=torch.float64, device=device)
    # targets: solve a throw-away position problem from a perturbed pose, read the site pose it ends in
    qt = q0.clone()
    qt[:, :7] += 0.15 * torch.randn(E, 7, generator=g, dtype=torch.float64, device=device)
    tg = (torch.tensor([0.6, 0.0, 1.1], dtype=torch.float64, device=device)
          + 0.15 * torch.randn(E, 3, generator=g, dtype=torch.float64, device=device)).contiguous()
    ax = torch.randn(E, 3, generator=g, dtype=torch.float64, device=device)
    ax = ax / ax.norm(dim=1, keepdim=True)
    ang = 0.6 * torch.rand(E, 1, generator=g, dtype=torch.float64, device=device)
    dq = torch.cat([torch.cos(ang / 2), torch.sin(ang / 2) * ax], dim=1)
    # current site orientation of q0 from the env kernel's obs would need an env; a fixed nominal orientation composed with
    # a random rotation of up to 0.6 rad poses the same kind of problem
    nominal = torch.tensor([0.0, 0.7071067811865476, 0.7071067811865476, 0.0], dtype=torch.float64, device=device)

    def qmul(a, b):
        aw, ax_, ay, az = a.unbind(-1)
        bw, bx, by, bz = b.unbind(-1)
        return torch.stack([aw * bw - ax_ * bx - ay * by - az * bz, aw * bx + ax_ * bw + ay * bz - az * by,
                            aw * by - ax_ * bz + ay * bw + az * bx, aw * bz + ax_ * by - ay * bx + az * bw], dim=-1)
    tquat = qmul(nominal.expand(E, 4), dq).contiguous()
    out = {"config": f"{env}, {E} IK problems (7 joints, grip_site), max_steps 100, tol 1e-2"}
    for key, quat in (("pos_quat", tquat), ("pos", None)):
        for _ in range(2):
            r = ik.solve(q0.clone(), tg, quat, max_steps=100, tol=1e-2)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            r = ik.solve(q0.clone(), tg, quat, max_steps=100, tol=1e-2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[key] = {"target": "position + orientation (6 x 7 Jacobian)" if quat is not None else "position (3 x 7 Jacobian)",
                    "solves_per_s": E / dt, "ms_per_batch": dt * 1e3, "success_rate": float(r.success.float().mean().item()),
                    "mean_iterations": float((r.steps.float() + 1).mean().item()),
                    "iterations_per_s": float((r.steps.float() + 1).sum().item()) / dt}
    return out


def scenes_section(torch, device, E, S, steps=5):
    """K1 on the other reference scenes at the same batch shape (4096 x 256 mixed states): which kernel the library picks and
    its rate -- Lift runs the gated mesh pass, Assembly and Lift read the FP32 centres from the pose slab."""
    from mopa_rl_amd import _lib
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs
    out = {}
    for env in ("SawyerLiftObstacle-v0", "SawyerAssemblyObstacle-v0", "PusherObstacle-v0"):
        pi = planner_inputs(env)
        sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range,
                        device=device.index if device.index is not None else -1)
        bp = BatchPlanner(sc)
        qa, rows = make_inputs(torch, pi, E, S, 1234, device)
        v = torch.empty(E * S, dtype=torch.uint8, device=device)
        for _ in range(2):
            bp.is_valid(qa, rows, samples_per_env=S, out=v)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(); bp.is_valid(qa, rows, samples_per_env=S, out=v); b.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        out[env] = {"checks_per_s": E * S / (ms * 1e-3), "ms_per_launch": ms, "kernel": sc.valid_kernel(E * S),
                    "pairs_checked_per_state": sc.npair_checked, "valid_fraction": float(v.float().mean().item())}
        sc.close()
    return out


def settle():
    """Before a timed region: collect Python garbage now (a Scene / env of an earlier section that dies INSIDE the region frees its
    device scratch with hipFree, a device-wide wait of tens of milliseconds), and keep the collector out of the region."""
    import gc
    gc.collect()
    gc.disable()


def unsettle():
    import gc
    gc.enable()


def lib_sha256():
    import hashlib
    from mopa_rl_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()


K1_SOURCES = ("mopa_rl_amd/csrc/mopa_hip.hip", "mopa_rl_amd/csrc/mopa_valid_v5.inc", "mopa_rl_amd/csrc/mopa_valid_v2.inc", "mopa_rl_amd/csrc/mopa_device.hpp",
              "mopa_rl_amd/csrc/mopa_host.hpp", "mopa_rl_amd/csrc/mopa_planner.inc", "mopa_rl_amd/csrc/mopa_planner_k3.inc", "mopa_rl_amd/csrc/mopa_pullback.inc", "mopa_rl_amd/csrc/mopa_motion.inc",
              "mopa_rl_amd/csrc/mopa_ik.inc", "mopa_rl_amd/csrc/mopa_paths.inc", "mopa_rl_amd/scenes/sawyer_push_obstacle.json")
# (not hashed: include/mopa_hip.h and the Makefile -- the C ABI's env / dynamics / rollout declarations change with the OTHER translation
#  unit; a change of the scene / validity declarations there comes with a change of mopa_hip.hip or mopa_host.hpp)


def k1_sources_sha256():
    """hash of everything the validity kernel's translation unit (mopa_hip.hip) and the bench scene are built from: a PMC pass stays
    valid for a library whose OTHER translation unit (env / dynamics kernels) changed"""
    import hashlib
    h = hashlib.sha256()
    for rel in K1_SOURCES:
        f = os.path.join(ROOT, rel)
        h.update(rel.encode())
        h.update(open(f, "rb").read() if os.path.exists(f) else b"<missing>")
    return h.hexdigest()


def traffic_record_seal(rec):
    """sha256 over every field of a traffic record but the seal itself (tools/make_traffic_json.py writes it with the record)"""
    import hashlib
    body = {k: v for k, v in rec.items() if k not in ("record_sha256", "source")}
    return hashlib.sha256(json.dumps(body, sort_keys=True).encode()).hexdigest()


def committed_traffic(kernel, n_states):
    """HBM-side traffic / VALU counts of the dominant kernel from the committed rocprofv3 PMC passes (tools/profile.sh ->
    tools/make_traffic_json.py -> profiles/rNN/k_is_valid_traffic.json); they cannot be collected from inside this process.
    A file is used only if it was measured on THIS build of libmopa_hip.so (sha256 stamp) -- or on a build whose validity-kernel
    translation unit and bench scene had the same sources (`k1_sources_sha256`) --, for this kernel and batch size."""
    import glob
    sha, src = lib_sha256(), k1_sources_sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "k_is_valid_traffic.json")), reverse=True):
        t = json.load(open(f))
        if "record_sha256" in t and t["record_sha256"] != traffic_record_seal(t):
            print(f"[bench] {os.path.relpath(f, ROOT)}: record edited after it was written (seal mismatch) -- ignored", file=sys.stderr)
            continue
        if "record_sha256" not in t and os.path.basename(os.path.dirname(f)) >= "r06":
            continue          # (records of round 6 on carry the seal; older rounds' files predate it)
        same_build = t.get("lib_sha256") == sha or (t.get("k1_sources_sha256") is not None and t.get("k1_sources_sha256") == src)
        if same_build and t.get("states_per_launch") == n_states and t.get("kernel", "").startswith(kernel):
            t["source"] = os.path.relpath(f, ROOT)
            return t
    return None


def cpu_baseline(pi, qa_host, rows_host, S, budget_states):
    """Oracle (oracle/mopa_oracle.c, kind="port") on the host cores, on the first `budget_states` states."""
    from oracle import oracle as O
    orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    n = min(budget_states, len(qa_host))
    n -= n % S
    qa = qa_host[:n]
    rows = rows_host[: n // S]
    cores = host_cores()
    t0 = time.perf_counter()
    v1, _ = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=1, want_min_dist=False)
    t1 = time.perf_counter()
    vN, _ = orc.is_valid_batch(qa, rows, samples_per_env=S, nthreads=cores, want_min_dist=False)
    t2 = time.perf_counter()
    return {"single": n / (t1 - t0), "all": n / (t2 - t1), "cores": cores, "n": n, "verdicts": vN}


HEADLINE_MAX_BYTES = 4096


def _g(d, *ks):
    for k in ks:
        d = d.get(k) if isinstance(d, dict) else None
    return d


def _r(x, nd=6):
    """numbers of the headline line to `nd` significant digits (the full-precision values are in the full record)"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def exchange_preflight(torch, dist, world, rank, device):
    """all-gather of a rank-stamped tensor over the initialised backend; raises unless every rank sees exactly the stamps 0 .. world-1 (and
    all ranks agree that they did).  Returns the stamps seen."""
    stamp = torch.full((4,), 1000 + rank, dtype=torch.int64, device=device)
    got = [torch.full((4,), -1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(got, stamp)
    seen = sorted({int(x) - 1000 for g in got for x in g.cpu().tolist()})
    ok = torch.tensor([1 if seen == list(range(world)) else 0], dtype=torch.int64, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        raise SystemExit(f"rank {rank}: exchange preflight failed -- stamps seen {seen}, expected {list(range(world))}: refusing to time a "
                         f"{world}-rank run whose collective does not reach every rank")
    return seen


def headline(out):
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, the parity count and a numeric summary of
    every section -- no prose (the per-section blocks are in gpurun_out/bench_full_n<N>.json and on stderr).  Pure function of
    the full record: tests/test_bench_headline.py holds it under HEADLINE_MAX_BYTES on a canned record."""
    rf = out.get("roofline") or {}
    cb = out.get("cpu_baseline")
    ro_keys = [k for k in out if k.startswith("rollout")]
    env = {k: v for k, v in (out.get("env_step") or {}).items() if isinstance(v, dict) and "steps_per_s" in v}
    h = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    h["config"] = {k: cfg.get(k) for k in ("workload", "envs_per_gpu", "states_per_env", "pairs_checked_per_state", "parallelism") if k in cfg}
    h["valid_fraction"] = out.get("valid_fraction")
    h["exchange"] = {k: _g(out, "exchange", k) for k in ("backend", "ranks", "ranks_seen", "kernel_ms_per_rank")}
    h["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_check",
                                            "traffic_source", "input_resident", "limiter") if k in rf}
    if "valu" in rf:
        h["roofline"]["valu"] = {k: rf["valu"].get(k) for k in ("achieved", "peak", "unit", "frac", "insts_per_check")}
    h["cpu_baseline"] = None if cb is None else {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "single_thread_value")}
    h["parity_mismatches_vs_oracle"] = out.get("parity_mismatches_vs_oracle")
    parity = {"planner": _g(out, "planner", "cpu_baseline", "parity_mismatches_vs_oracle")}
    for k, v in env.items():
        for d in ("", "dynamics", "dynamics_contacts"):
            p = _g(v, d, "parity_mismatches_vs_oracle") if d else v.get("parity_mismatches_vs_oracle")
            if p is not None:
                parity[f"env_{k}" + (f"_{d}" if d else "")] = p
    h["summary"] = {
        "checks_per_s": out.get("value"), "roofline_frac_hbm": rf.get("frac"), "valu_frac": _g(rf, "valu", "frac"),
        "motions_per_s": _g(out, "motion", "motions_per_s_range"),
        "planner": {"ms_per_batch": _g(out, "planner", "ms_per_batch"), "plans_per_s": _g(out, "planner", "plans_per_s"),
                    "laddered_plans_per_s": _g(out, "planner", "laddered", "plans_per_s"),
                    "laddered_long_plans_per_s": _g(out, "planner", "laddered_long", "plans_per_s"), "success_rate": _g(out, "planner", "success_rate"),
                    "cpu_plans_per_s": _g(out, "planner", "cpu_baseline", "value")},
        "scenes_checks_per_s": {k.split("Obstacle")[0].replace("Sawyer", "").lower(): _g(v, "checks_per_s")
                                for k, v in (out.get("scenes") or {}).items() if isinstance(v, dict)},
        "env_steps_per_s": {k: {"kin": _g(v, "steps_per_s"), "dyn": _g(v, "dynamics", "steps_per_s"), "dyn_sat": _g(v, "dynamics_saturated", "steps_per_s"),
                                "ct": _g(v, "dynamics_contacts", "steps_per_s"), "ct_ms": _g(v, "dynamics_contacts", "ms_per_batch"),
                                "ct_cpu": _g(v, "dynamics_contacts", "cpu_baseline", "value"),
                                "ct_dropped": _g(v, "dynamics_contacts", "contacts_dropped_by_the_cap_per_substep")}
                            for k, v in env.items()},
        "ik_solves_per_s": _g(out, "ik", "pos_quat", "solves_per_s"),
        "rollout_agent_steps_per_s": {k[8:] or "lockstep": _g(out[k], "agent_steps_per_s") for k in ro_keys},
        "rollout_envs_stepping_per_call": {k[8:] or "lockstep": _g(out[k], "envs_stepping_per_call") for k in ro_keys},
        "parity_mismatches": parity,
    }
    h["full_record"] = f"gpurun_out/bench_full_n{out.get('n_gpus')}.json"
    h = _r(h)
    if len(json.dumps(h, separators=(",", ":"))) > HEADLINE_MAX_BYTES:       # never again a line the driver cannot keep: drop the summary first
        h["summary"] = {"dropped": "headline over the byte budget; see the full record"}
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--samples", type=int, default=256, help="states per env per step")
    ap.add_argument("--cpu-states", type=int, default=1 << 20, help="states timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--mode", default="mixed", choices=["mixed", "near", "uniform"],
                    help="state distribution: the headline workload is `mixed` (50%% uniform joint-box, 50%% near-init)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-plan", action="store_true", help="skip the RRT-Connect section (config 3)")
    ap.add_argument("--plan-envs", type=int, default=4096)
    ap.add_argument("--graphs", action="store_true", help="also run the HIP-graph replay form of the asynchronous Push rollout")
    ap.add_argument("--no-env", action="store_true", help="skip the kinematic env.step section")
    ap.add_argument("--no-rollout", action="store_true", help="skip the end-to-end rollout sections (Push at 1 GPU; Lift with the "
                    "transition all-gather + gradient all-reduce at any rank count)")
    ap.add_argument("--rendezvous-only", action="store_true", help="test hook: launch / rendezvous / one collective, no GPU work "
                    "(checks that --gpus N really yields N ranks; runs on a CPU-only box with MOPA_BENCH_BACKEND=gloo)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launch the ranks ourselves: one process per GPU under torch.distributed.run (reference rank set-up: rl/main.py:24-35)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import torch
    import torch.distributed as dist
    from mopa_rl_amd import _lib
    from mopa_rl_amd.batch import BatchPlanner
    from mopa_rl_amd.scene import planner_inputs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a "
                         f"{args.gpus}-GPU number from {world} process(es)")
    if args.rendezvous_only:
        backend = os.environ.get("MOPA_BENCH_BACKEND", "nccl")
        seen = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                t = torch.ones(1, device=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
                t = torch.ones(1)
            dist.all_reduce(t)
            seen = int(t.item())
            assert dist.get_world_size() == args.gpus
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "valid-state collision checks/sec", "value": None, "unit": "checks/s", "n_gpus": world,
                              "rendezvous_only": True, "ranks_seen_by_all_reduce": seen,
                              "config": {"parallelism": f"env-shard x{world}"},
                              "exchange": {"backend": backend, "collectives": ["all_gather(uint8 masks)", "all_gather(transitions)",
                                                                               "all_reduce(gradients)"]}}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # test-only knobs: MOPA_BENCH_DEVICE pins every rank to one device and MOPA_BENCH_BACKEND=gloo swaps the collective
    # backend, so that the world > 1 code path can be exercised end to end on a single-GPU box
    dev_index = int(os.environ.get("MOPA_BENCH_DEVICE", local_rank))
    backend = os.environ.get("MOPA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)   # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    # preflight over the REAL backend, before any timing: every rank all-gathers a rank-stamped tensor and must see `world` distinct
    # stamps (a collective that silently ran over fewer ranks, or returned stale buffers, fails here and not in a scaling curve)
    ranks_seen = exchange_preflight(torch, dist, world, rank, device if backend == "nccl" else torch.device("cpu")) if world > 1 else [0]

    pi = planner_inputs(ENV)
    scene = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold,
                       range_=pi.spec.range, seed=0, device=dev_index)
    bp = BatchPlanner(scene)
    E, S = args.envs, args.samples
    N = E * S
    qa, rows = make_inputs(torch, pi, E, S, seed=1234 + rank, device=device, mode=args.mode)
    # Triple-buffered verdict masks: the RCCL all-gather of step k (on RCCL's own stream) overlaps the validity kernels of
    # steps k+1 and k+2; a buffer is only reused after the collective that reads it has completed (mopa_rl_amd/dist.py).
    from mopa_rl_amd.dist import OverlappedGather
    og = OverlappedGather(N, torch.uint8, device)

    def step(k, events=None):
        out_k = og.buffer(k)
        if events is not None:
            events[0].record()
        bp.is_valid(qa, rows, samples_per_env=S, out=out_k)
        if events is not None:
            events[1].record()
        og.launch(k)

    def drain():
        og.drain()

    def barrier():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    settle()
    barrier()
    # kernel-only timing: HIP events on the stream the kernel is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, ev[k])
    barrier()
    elapsed = time.perf_counter() - t0
    unsettle()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    kern_ms_ranks = [kern_ms]
    if world > 1:
        cdev = device if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(km, torch.tensor([kern_ms], dtype=torch.float64, device=cdev))
        kern_ms_ranks = [round(float(x.item()), 6) for x in km]      # a slow rank shows here (the headline's roofline is rank 0's kernel)
    valid = og.local[(args.steps - 1) % og.depth] if args.steps > 0 else og.local[0]

    n_valid = int(valid.sum().item())
    if rank == 0:
        total_checks = world * N * args.steps
        value = total_checks / elapsed
        bytes_per_check = 7 * 8 + 1 + pi.model.nq * 8 / S
        achieved = N * bytes_per_check / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "valid-state collision checks/sec", "value": value, "unit": "checks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{ENV} state validity (FK + collision), {E} envs/GPU x {S} states/env per step, "
                                   + {"mixed": "states 50% uniform joint-box samples + 50% near-init N(0,0.3)", "near": "states near-init N(0,0.3)",
                                      "uniform": "states uniform in the joint box"}[args.mode],
                       "envs_per_gpu": E, "states_per_env": S, "pairs_checked_per_state": scene.npair_checked,
                       "parallelism": f"env-shard x{world}" + (f" + {'RCCL' if backend == 'nccl' else backend} all_gather(uint8 masks), triple-buffered, overlapped with the next steps' kernels" if world > 1 else "")},
            "valid_fraction": n_valid / N,
            "exchange": {"backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None, "ranks": world, "ranks_seen": ranks_seen,
                         "kernel_ms_per_rank": kern_ms_ranks,
                         "collectives": (["all_gather(uint8 masks) per validity step", "all_gather(transition records) + all_reduce(gradient-sized "
                                          "buffer) per rollout call"] if world > 1 else [])},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": scene.valid_kernel(N), "kernel_ms": kern_ms, "bytes_per_check": bytes_per_check,
                         "input_resident": "one input batch re-used by every step (58.7 MB: Infinity-Cache resident after step 1)",
                         "limiter": "fp64-issue",      # what actually bounds K1 (`valu` below when a PMC pass of this build is committed); `bound` / `frac` stay the HBM roofline the contract asks for
                         "note": "path is FP64-VALU bound, not HBM bound (SURVEY.md 8d); see DESIGN.md"},
        }
        t = committed_traffic(scene.valid_kernel(N), N)
        if t is None:
            out["roofline"]["traffic_source"] = None      # no PMC pass committed for this build of the library
        else:
            out["roofline"]["traffic"] = t["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = f"{t['source']} (lib sha256 {t['lib_sha256'][:12]})"
            out["roofline"]["traffic_note"] = t["note"]
            if "valu_insts_per_launch" in t:
                # FP64 issue roof: a wave64 FP64 op holds a 16-lane SIMD for 4 cycles -> CUs*4*clk/4 wave-instr/s
                peak = torch.cuda.get_device_properties(device).multi_processor_count * 4 * 2.4e9 / 4
                ach = t["valu_insts_per_launch"] / (kern_ms * 1e-3)
                out["roofline"]["valu"] = {"achieved": ach, "peak": peak, "unit": "wave-instr/s", "frac": ach / peak,
                                           "insts_per_check": t["valu_insts_per_launch"] / t["states_per_launch"],
                                           "note": "SQ_INSTS_VALU from the committed PMC pass / live kernel time"}
        if not args.no_plan and world == 1:
            out["motion"] = motion_section(torch, bp, pi, qa, rows, S, device)
            out["planner"] = plan_section(torch, bp, pi, args.plan_envs, device, not args.no_cpu)
        if not args.no_env and world == 1:
            out["scenes"] = scenes_section(torch, device, E, S)
            out["env_step"] = env_step_section(torch, args.envs, device, 50, not args.no_cpu)
            out["ik"] = ik_section(torch, device)
        if not args.no_cpu and world == 1:
            cb = cpu_baseline(pi, qa.cpu().numpy(), rows.cpu().numpy(), S, args.cpu_states)
            mism = int((cb["verdicts"] != valid[: cb["n"]].cpu().numpy()).sum())
            out["cpu_baseline"] = {
                "value": cb["all"], "unit": "checks/s", "cores": cb["cores"], "kind": "port",
                "sample": f"first {cb['n']} states of the step batch; C restatement (oracle/), not MuJoCo/OMPL",
                "single_thread_value": cb["single"]}
            out["parity_mismatches_vs_oracle"] = mism
    # the rollout sections run on EVERY rank (their exchanges are collectives): config 3 (Push) at one GPU, config 4
    # (Lift: env shard + all-gather of the rollout transitions + gradient all-reduce) at any rank count
    ro = {}
    if not args.no_rollout:
        if world == 1:
            ro["rollout"] = rollout_section(torch, ENV, args.envs, device, 3)
            ro["rollout_async"] = rollout_section(torch, ENV, args.envs, device, 300, async_planner=True)     # (a retry launch lives ~25 calls: 300 calls = 12 of its cycles)
            # the same at twice the envs per GPU: at 4096 half the envs wait for an RRT-Connect query in any call and the call
            # is bound by host dispatch; the agent-step rate levels off near 8192-16384 resident envs (tools/rollout_envs_sweep.py)
            ro["rollout_async_2x"] = rollout_section(torch, ENV, 2 * args.envs, device, 200, async_planner=True)
            # BASELINE config 1's env, batched (PusherObstacle-v0, joint0 unlimited: wrapped query endpoints, seam rule of the un-wrap)
            ro["rollout_pusher"] = rollout_section(torch, "PusherObstacle-v0", args.envs, device, 200, async_planner=True)
            if args.graphs:          # (HIP-graph replay no longer pays: DESIGN 8; kept behind the flag)
                ro["rollout_async_graphs"] = rollout_section(torch, ENV, args.envs, device, 300, async_planner=True, use_graphs=True)
            # the same rollout where a Push policy could actually be trained: the env with dynamics + contacts (stage C)
            ro["rollout_async_dyn"] = rollout_section(torch, ENV, args.envs, device, 150, async_planner=True, dynamics=True)
        ro["rollout_lift"] = rollout_section(torch, "SawyerLiftObstacle-v0", args.envs, device, 300, world, async_planner=True)
        # BASELINE config 5: SawyerAssemblyObstacle with the IK action space, 8192 envs per GPU
        ro["rollout_assembly_ik"] = rollout_section(torch, "SawyerAssemblyObstacle-v0", 2 * args.envs, device, 120, world, async_planner=True,
                                                    use_ik=True)
    if rank == 0:
        out.update(ro)
        # the full record (every section, all prose) goes to a file + stderr; stdout ends with ONE compact line (< 4 KB)
        full = json.dumps(out)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"bench_full_n{world}.json"), "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
        if os.environ.get("MOPA_BENCH_FULL_STDERR"):
            print("[bench full record] " + full, file=sys.stderr)
            sys.stderr.flush()
        print(json.dumps(headline(out), separators=(",", ":")))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
