"""CPU test: the C-ABI library loads and exports every symbol include/kinetica_hip.h declares
(no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

from kinetica_jl_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kinetica_hip.h")


def _declared():
    text = open(os.path.join(ROOT, "include", "kinetica_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kin_[A-Za-z0-9_]+)\s*\(", text)))


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


def test_struct_layouts_match_header(tmp_path):
    """Field offsets and sizes of kin_params / kin_stats as the C compiler lays them out from the header itself,
    against the ctypes mirrors (the Julia structs of INTEGRATION.md list the same fields in the same order)."""
    import subprocess
    fields = {"kin_params": capi.KinParams, "kin_stats": capi.KinStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in fields.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in fields.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    # the header declares no field the mirrors lack
    import re
    text = open(HEADER).read()
    for cname, cls in fields.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        declared = [n.strip() for decl in body.split(";") if decl.strip()
                    for n in decl.strip().split(None, 1)[1].split(",")]
        assert declared == [f for f, _ in cls._fields_], cname


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
