"""Build libmifwi.so (hand-written HIP, gfx950 only) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libmifwi.so")
# the same kernels with the timing / fault-injection ablations compiled in (-DMIFWI_ABLATIONS): loaded only through
# MIFWI_LIB, by tools/latency_floor.py and tests/test_fallback_gpu.py; always built together with LIB
LIB_ABLATIONS = os.path.join(_HERE, "libmifwi_ablations.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",           # arithmetic is an explicit fmaf chain (bitwise vs oracle)
    "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(LIB_ABLATIONS):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_ABLATIONS))
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(_HERE, "..", "include", "mifwi.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """`out`: build another copy (e.g. an ablation build loaded through MIFWI_LIB) instead of the in-tree library."""
    if out is None and not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-o", out or LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if out is None and "-DMIFWI_ABLATIONS" not in extra_flags:
        build(force=True, verbose=verbose, extra_flags=list(extra_flags) + ["-DMIFWI_ABLATIONS"], out=LIB_ABLATIONS)
    return out or LIB


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True,
          extra_flags=[a for a in sys.argv[1:] if a.startswith(("-R", "-save", "-D"))])
