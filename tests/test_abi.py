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
    assert lib.mifwi_version() == 7


def test_struct_layouts_match_header(tmp_path):
    """sizeof and every field offset of the four C structs, as gcc lays out include/mifwi.h, against the ctypes
    mirrors in _lib.py."""
    import subprocess
    from physicsbasedfwi2_amd import _lib
    structs = {"mifwi_acoustic_desc": _lib.AcousticDesc, "mifwi_elastic_desc": _lib.ElasticDesc,
               "mifwi_acoustic_layout": _lib.AcousticLayout, "mifwi_elastic_layout": _lib.ElasticLayout}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mifwi.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append("return 0; }")
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)
    assert ctypes.sizeof(_lib.ElasticDesc) == 14 * 4


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
