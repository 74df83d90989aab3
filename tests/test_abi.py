"""The C-ABI library loads on a CPU-only box and exports every symbol include/mopa_hip.h declares;
without a GPU the product fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mopa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mopa_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    from mopa_rl_amd import _lib
    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mopa_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == syms


def test_version_and_device_count():
    from mopa_rl_amd import _lib
    L = _lib.lib()
    assert b"gfx950" in L.mopa_version()
    assert L.mopa_device_count() >= 0


def test_no_cpu_fallback_without_gpu():
    import torch
    from mopa_rl_amd import _lib
    from mopa_rl_amd.scene import planner_inputs
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pi = planner_inputs("SawyerPushObstacle-v0")
    with pytest.raises(_lib.MopaError, match="no HIP device"):
        _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, -0.002)


def test_product_never_imports_oracle():
    """Nothing under mopa_rl_amd/ may reference oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "mopa_rl_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "mopa_oracle" not in src, f


def test_bad_arguments_are_status_codes_not_crashes():
    from mopa_rl_amd import _lib
    L = _lib.lib()
    assert L.mopa_scene_create(None, None) == 1          # MOPA_ERR_INVALID_ARG
    assert b"null" in L.mopa_last_error()
    assert L.mopa_scene_num_active(None) == -1
    L.mopa_scene_destroy(None)                           # no-op
    assert L.mopa_planner_status(None) == b"none"


def test_positive_contact_threshold_is_rejected():
    """the broad phase culls at zero margin: a threshold > 0 would make verdicts depend on the cull (include/mopa_hip.h);
    the check precedes device selection, so it is observable without a GPU"""
    from mopa_rl_amd import _lib
    from mopa_rl_amd.scene import planner_inputs
    pi = planner_inputs("SawyerPushObstacle-v0")
    with pytest.raises(_lib.MopaError, match="contact_threshold"):
        _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, 0.001)


def test_rollout_step_struct_layout_matches_the_library():
    """MopaRolloutStep is filled field by field from Python: its ctypes layout must be the one the library was compiled with."""
    import ctypes as C
    from mopa_rl_amd import _lib
    assert _lib.lib().mopa_rollout_step_size() == C.sizeof(_lib.MopaRolloutStep)


def test_contact_desc_struct_layouts_match_the_libraries():
    """MopaCtDesc / OrcCtDesc (about 40 mixed int32 / double / array fields) are mirrored by hand in ctypes on both sides"""
    import ctypes as C
    from mopa_rl_amd import _lib
    from oracle import oracle as O
    assert _lib.lib().mopa_ct_desc_size() == C.sizeof(_lib.MopaCtDesc)
    assert O.lib().orc_ct_desc_size() == C.sizeof(O.OrcCtDesc)
