"""CPU side of the compile-time pair pruning (tools/prove_separated_pairs.py): the committed scenes carry the proven pairs,
and the oracle -- which keeps checking them -- never sees one of them at or below zero distance on sampled states."""
import numpy as np
import pytest

from conftest import SUPPORTED_ENVS, sample_states


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
