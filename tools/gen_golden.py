#!/usr/bin/env python3
"""Generate tests/golden/*.npz: seeded inputs + the CPU oracle's outputs.

The reference cannot produce golden vectors here (its arithmetic is in MuJoCo 2.0 / OMPL, absent from
this image -- SURVEY.md 8c), and it ships none.  These fixtures therefore pin (a) the oracle against
regressions and (b) the HIP path against committed data on the GPU box.  PARITY vs MuJoCo/OMPL: UNPINNED.

    python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import SUPPORTED_ENVS, sample_states  # noqa: E402
from mopa_rl_amd.scene import planner_inputs  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    for env in SUPPORTED_ENVS:
        pi = planner_inputs(env)
        orc = O.OracleScene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
        qa_u, row = sample_states(pi, 192, 2024, "uniform")
        qa_n, _ = sample_states(pi, 192, 2025, "near")
        qa = np.concatenate([qa_u, qa_n])
        valid, md = orc.is_valid_batch(qa, row, samples_per_env=len(qa))
        q8 = []
        gp8, gm8 = [], []
        for i in range(8):
            q = row[0].copy()
            q[pi.ref_joint_pos_indexes] = qa[i * 40]
            gp, gm = orc.fk(q)
            q8.append(q); gp8.append(gp); gm8.append(gm)
        # motion validation
        rng = np.random.default_rng(77)
        qb = np.clip(qa + rng.normal(0, 0.05, qa.shape), pi.jnt_minimum, pi.jnt_maximum)
        mv = orc.check_motion_batch(qa, qb, row, samples_per_env=len(qa))
        # planner queries between valid states
        good = qa_n[valid[len(qa_u):] == 1]   # plan between near-init valid states (as a rollout would)
        starts, goals, stats, plens, nchks, paths = [], [], [], [], [], []
        for k in range(4):
            s, g = row[0].copy(), row[0].copy()
            s[pi.ref_joint_pos_indexes] = good[2 * k]
            g[pi.ref_joint_pos_indexes] = good[2 * k + 1]
            st, path, nchk, _ = orc.plan(s, g, pi.spec.range, 0.005, 400, 4096, seed=99, env_id=k, max_path=512)
            full = np.zeros((512, pi.model.nq)); full[:len(path)] = path
            starts.append(s); goals.append(g); stats.append(st); plens.append(len(path)); nchks.append(nchk); paths.append(full)
        path_out = os.path.join(OUT, f"{pi.spec.scene}.npz")
        np.savez_compressed(
            path_out, q_active=qa, qpos_env=row, valid=valid, min_dist=md, fk_qpos=np.array(q8),
            fk_geom_pos=np.array(gp8), fk_geom_mat=np.array(gm8), motion_qb=qb, motion_valid=mv,
            plan_start=np.array(starts), plan_goal=np.array(goals), plan_status=np.array(stats),
            plan_len=np.array(plens), plan_checks=np.array(nchks), plan_path=np.array(paths),
            plan_params=np.array([400, 4096, 512, 99]))
        print(env, "valid", int(valid.sum()), "/", len(valid), "motion", int(mv.sum()), "plan status", stats, "len", plens,
              os.path.getsize(path_out), "B")


if __name__ == "__main__":
    main()
