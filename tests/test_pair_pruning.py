"""CPU side of the compile-time pair pruning (tools/prove_separated_pairs.py): the committed scenes carry the proven pairs,
and the oracle -- which keeps checking them -- never sees one of them at or below zero distance on sampled states."""
import numpy as np
import pytest

from conftest import SUPPORTED_ENVS, sample_states


def _rbound(t, s):
    """bounding radius about the geom's origin (mjcf.GEOM_*: 2 sphere, 3 capsule, 5 cylinder, 6 box)"""
    return {2: s[0], 3: s[0] + s[1], 5: float(np.hypot(s[0], s[1])), 6: float(np.linalg.norm(s))}[t]


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_committed_never_violating_pairs_hold_on_samples(env, oracle_mod):
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env)
    m = pi.model
    never = m.meta.get("never_violating_pairs", [])
    assert len(never) >= 10
    ign = set(tuple(p) for p in pi.ignored_contacts)
    idx = {(int(a), int(b)): k for k, (a, b) in enumerate(m.pair_geom)}
    for a, b in never:
        ia, ib = int(m.geom_mjid[a]), int(m.geom_mjid[b])
        assert (int(a), int(b)) in idx and (min(ia, ib), max(ia, ib)) not in ign
    orc = oracle_mod.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    rows_of = [idx[(int(a), int(b))] for a, b in never]
    rng = np.random.default_rng(0)
    lo = np.inf
    for mode in ("uniform", "near"):
        qa, rows = sample_states(pi, 400, 3, mode)
        for i in range(len(qa)):
            q = rows[0].copy()
            q[pi.ref_joint_pos_indexes] = qa[i]
            for j in pi.passive_joint_idx:        # passive slides anywhere in their range (the proof covers them)
                jid = [k for k in range(len(m.jnt_names)) if int(m.jnt_qposadr[k]) == int(j) and int(m.jnt_type[k]) == 2]
                if jid and m.jnt_limited[jid[0]]:
                    q[j] = rng.uniform(*m.jnt_range[jid[0]])
            lo = min(lo, orc.pair_dist(q)[rows_of].min())
    assert lo > 1e-4


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_committed_cull_radii_hold_on_samples(env, oracle_mod):
    """`pair_cull_radius` (the tightened broad-phase radius of a pair): whenever the oracle sees the pair at or below the
    contact threshold, the two geom centres are no further apart than the committed radius -- so culling the pair beyond
    it never changes a verdict.  Sampled where it matters: the joints between the two geoms swept over their ranges."""
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env)
    m = pi.model
    cr = m.meta.get("pair_cull_radius")
    assert cr is not None and cr["threshold"] <= pi.spec.contact_threshold + 1e-12
    if not cr["pairs"]:
        return                                                 # (Lift: no pair gains 10 % over its bounding spheres)
    idx = {(int(a), int(b)): k for k, (a, b) in enumerate(m.pair_geom)}
    orc = oracle_mod.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    rows_of = np.array([idx[(int(a), int(b))] for a, b, _ in cr["pairs"]])
    ga = np.array([int(a) for a, _, _ in cr["pairs"]])
    gb = np.array([int(b) for _, b, _ in cr["pairs"]])
    rad = np.array([float(r) for _, _, r in cr["pairs"]])
    rb = lambda gs: np.array([_rbound(int(m.geom_type[g]), m.geom_size[g]) for g in gs])
    assert np.all(rad < rb(ga) + rb(gb))                       # tightened means tighter than the bounding spheres
    rng = np.random.default_rng(1)
    lim = np.array([m.jnt_range[k] if m.jnt_limited[k] else (-np.pi, np.pi) for k in range(len(m.jnt_names))])
    adr = np.array([int(a) for a in m.jnt_qposadr])
    hs = [k for k in range(len(m.jnt_names)) if int(m.jnt_type[k]) in (2, 3)]
    qa, rows = sample_states(pi, 4, 3, "uniform")
    hits = 0
    for i in range(3000):
        q = rows[0].copy()
        for k in hs:
            q[adr[k]] = rng.uniform(*lim[k])
        d = orc.pair_dist(q)[rows_of]
        gp, _ = orc.fk(q)
        cd = np.linalg.norm(gp[ga] - gp[gb], axis=1)
        viol = d <= pi.spec.contact_threshold
        hits += int(viol.sum())
        assert np.all(cd[viol] <= rad[viol]), (env, i)
    assert hits > 0                                            # the property was exercised


@pytest.mark.parametrize("env", SUPPORTED_ENVS)
def test_committed_pruning_is_a_subset_of_what_the_tool_proves_today(env, oracle_mod):
    """a change in the MJCF compile (mjcf.py) must not leave a stale `never_violating_pairs` behind: five committed pairs per
    scene are re-proven here with the tool's own routine (same guard band, smaller budget: the cheap proofs), and the guard
    band the runtime reads is the one the tool proves with"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import prove_separated_pairs as P
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs(env)
    m = pi.model
    band = m.meta.get("prune_guard_band")
    assert band == {"hinge": P.BAND_HINGE, "slide": P.BAND_SLIDE}
    P.MAX_JOINTS = 5            # (the committed scenes were proven with --max-joints 5)
    never = [(int(a), int(b)) for a, b in m.meta.get("never_violating_pairs", [])]
    orc = oracle_mod.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    q0 = np.array(m.qpos0, dtype=np.float64)
    rng = np.random.default_rng(11)
    done = 0
    for k in rng.permutation(len(never)):
        a, b = never[k]
        res, why = P.prove_pair(m, orc, q0, a, b, 60000)
        if res is None and why.startswith("budget"):
            continue                    # an expensive proof: another pair
        assert res is True, f"pair {a} / {b} is committed as never-violating but is not proven today: {why}"
        done += 1
        if done == 5:
            break
    assert done == 5
    # and an unlimited slide can never be part of a proof (its box would be empty)
    assert all(P.joint_box(m, j) is not None or int(m.jnt_type[j]) == 2 for j in range(len(m.jnt_names)))


def test_separation_proofs_are_bound_to_the_geometry_they_were_made_for():
    """the proof keys in a scene's meta carry the hash of the kinematic tree / ranges / geoms / pairs they were derived from: the
    committed scenes match their stamp; a changed joint range drops the proofs at load time (and in tools/compile_scenes.py)"""
    import warnings
    from mopa_rl_amd.mjcf import CompiledModel
    from mopa_rl_amd.scene import ENV_SPECS, scene_path
    for env, spec in ENV_SPECS.items():
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            m = CompiledModel.load(scene_path(spec.scene))
        assert m.meta.get("proof_geometry_sha256") == m.geometry_sha256() and len(m.meta["never_violating_pairs"]) > 0
    m.jnt_range = np.array(m.jnt_range, dtype=np.float64).copy()
    m.jnt_range[0, 1] += 0.1
    assert m.drop_stale_proofs() and "never_violating_pairs" not in m.meta and "never_within_margin_pairs" not in m.meta
    assert not m.drop_stale_proofs()
