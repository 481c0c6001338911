"""CPU test: the C-ABI library loads and exports every symbol include/kinetica_hip.h declares
(no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

from kinetica_jl_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "kinetica_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kin_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(L, name), name
    assert b"gfx950" in capi.lib().kin_version()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.KinParams) == 8 * 4 + 4 * 4 + 8 * 3
    assert ctypes.sizeof(capi.KinStats) == 8 * 17


def test_invalid_network_is_rejected_before_touching_the_gpu():
    import numpy as np
    # 3 reactant molecules: unsupported (max_molecularity = 2, network.jl:275-279)
    with pytest.raises(capi.KineticaHipError) as e:
        capi.HipNetwork(3, [0, 1], [0], [3], [0, 1], [1], [1])
    assert e.value.code == capi.KIN_ERR_UNSUPPORTED
    with pytest.raises(capi.KineticaHipError) as e:
        capi.HipNetwork(3, [0, 1], [7], [1], [0, 1], [1], [1])
    assert e.value.code == capi.KIN_ERR_INVALID_ARG
    # a valid network without a GPU must fail loudly with KIN_ERR_DEVICE, never fall back
    if capi.device_count() == 0:
        with pytest.raises(capi.KineticaHipError) as e:
            capi.HipNetwork(2, [0, 1], [0], [1], [0, 1], [1], [1])
        assert e.value.code == capi.KIN_ERR_DEVICE
