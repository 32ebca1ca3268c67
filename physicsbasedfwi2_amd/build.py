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


def _compile_and_link(out, extra_flags, verbose):
    """One hipcc -c per source (in parallel: the two big kernel files take ten seconds each), then one link."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    flags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    with tempfile.TemporaryDirectory(prefix="mifwi_build_") as tmp:
        jobs = []
        for src in sources():
            obj = os.path.join(tmp, os.path.basename(src) + ".o")
            jobs.append(([HIPCC] + flags + ["-c", "-o", obj, src], obj))
        if verbose:
            for cmd, _ in jobs:
                print(" ".join(cmd))
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(lambda j: subprocess.check_call(j[0]), jobs))
        link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for _, o in jobs]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """`out`: build another copy (e.g. an ablation build loaded through MIFWI_LIB) instead of the in-tree library."""
    if out is None and not force and not needs_build():
        return LIB
    if out is None and "-DMIFWI_ABLATIONS" not in extra_flags:
        # libmifwi.so and its -DMIFWI_ABLATIONS twin side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=2) as ex:
            a = ex.submit(_compile_and_link, LIB, list(extra_flags), verbose)
            b = ex.submit(_compile_and_link, LIB_ABLATIONS, list(extra_flags) + ["-DMIFWI_ABLATIONS"], verbose)
            a.result(); b.result()
        return LIB
    _compile_and_link(out or LIB, list(extra_flags), verbose)
    return out or LIB


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True,
          extra_flags=[a for a in sys.argv[1:] if a.startswith(("-R", "-save", "-D"))])
