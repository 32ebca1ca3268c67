"""The C-ABI library loads and exports every symbol include/mifwi.h declares, the ctypes table
matches the header's parameter counts, and the product fails loudly without a HIP device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mifwi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:int64_t|int|const char \*)\s*\*?\s*(mifwi_\w+)\s*\(([^;]*?)\)\s*;", src, re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_header_symbols_exported_and_bound():
    from physicsbasedfwi2_amd import _lib
    lib = _lib.load()
    decl = _header_functions()
    assert len(decl) >= 13
    for name, nargs in decl.items():
        assert hasattr(lib, name), "%s declared in mifwi.h but not exported" % name
        assert name in _lib.SIGNATURES, "%s has no ctypes signature" % name
        assert len(_lib.SIGNATURES[name][1]) == nargs, name
    for name in _lib.SIGNATURES:
        assert name in decl, "%s bound but not declared in mifwi.h" % name
    assert lib.mifwi_version() == 2


def test_struct_layouts_match_header():
    from physicsbasedfwi2_amd import _lib
    assert ctypes.sizeof(_lib.AcousticDesc) == 11 * 4
    assert ctypes.sizeof(_lib.ElasticDesc) == 12 * 4
    assert ctypes.sizeof(_lib.AcousticLayout) == 4 * 4 + 4 * 8
    assert ctypes.sizeof(_lib.ElasticLayout) == 4 * 4 + 4 * 8


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device error path")
def test_no_cpu_fallback():
    from physicsbasedfwi2_amd import _lib, acoustic, elastic
    from physicsbasedfwi2_amd._lib import MifwiError
    lib = _lib.load()
    assert lib.mifwi_device_count() == 0
    h = ctypes.c_void_p()
    d = _lib.AcousticDesc(8, 8, 4, 1, 1, 1, 1, 1.0, 1.0, 0, 0)
    rc = lib.mifwi_acoustic_plan_create(ctypes.byref(h), 0, ctypes.byref(d))
    assert rc == -2 and b"no CPU fallback" in lib.mifwi_last_error()
    with pytest.raises(MifwiError):
        acoustic.propagate(torch.ones(8, 8), torch.zeros(4, 1, 1), torch.zeros(8), torch.zeros(8),
                           torch.zeros(1, 1, 1, dtype=torch.int32), torch.ones(1, 1, 1),
                           torch.zeros(1, 1, 1, dtype=torch.int32), torch.ones(1, 1, 1))
    with pytest.raises(MifwiError):
        elastic.propagate(torch.ones(5, 8, 8), torch.zeros(4, 1, 1), torch.zeros(6, 8),
                          torch.zeros(6, 8), torch.zeros(1, 1, 1, dtype=torch.int32),
                          torch.ones(1, 1, 1), torch.zeros(1, 1, 1, dtype=torch.int32),
                          torch.ones(1, 1, 1), 0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "physicsbasedfwi2_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M), f
                assert "liboracle" not in txt, f
