#!/usr/bin/env python3
"""BASELINE config 5's per-GPU share at its full length (elastic 1000x3000, free surface, 16 shots, 5000 steps, one
MI355X; time-checkpointed: the 4.8 TB of f32 snapshot planes cannot be resident) -> profiles/<round>_c5_full_length.json.

    python tools/c5_full_length.py [OUT.json]       (on the GPU box; ~1.5 minutes per snapshot format)

Per format: the full report of `python bench.py --workload elastic_seam --steps 2 --warmup 1`, the per-kernel times of
a short run with resident snapshots, resident_equivalent_ms = 5000 x (forward+save + adjoint) of that short run, and
checkpointing_overhead = ms_per_step / resident_equivalent_ms - 1 (the second forward sweep of the checkpointed pass).
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(fmt):
    env = dict(os.environ, MIFWI_EL_SNAP=fmt)
    detail = tempfile.NamedTemporaryFile(suffix=".json", delete=False).name
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "elastic_seam", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--detail", detail], env=env, capture_output=True, text=True, timeout=1100)
    if res.returncode != 0:
        raise SystemExit("bench failed (%s):\n%s" % (fmt, res.stderr[-2000:]))
    with open(detail) as fh:
        d = json.load(fh)
    os.unlink(detail)
    keep = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "check", "memory", "kernels_note")
            if k in d}
    keep["kernels_us_per_step"] = {k: round(v["avg_step_s"] * 1e6, 1) for k, v in d["kernels"].items()}
    nt = d["config"]["nt"]
    keep["resident_equivalent_ms"] = round(nt * sum(v["avg_step_s"] for v in d["kernels"].values()) * 1e3, 1)
    keep["checkpointing_overhead"] = round(d["ms_per_step"] / keep["resident_equivalent_ms"] - 1.0, 3)
    return keep


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", bench.PROFILE_ROUND + "_c5_full_length.json")
    doc = {"csrc_sha16": bench.csrc_sha16(), "f32": run("f32"), "bf16": run("bf16"),
           "note": "BASELINE config 5's per-GPU share at its full length: elastic 1000x3000, free surface, 16 shots, 5000 steps, one "
                   "MI355X, time-checkpointed (the 4.8 TB of f32 snapshot planes cannot be resident): python bench.py --workload "
                   "elastic_seam --steps 2 --warmup 1 [MIFWI_EL_SNAP=bf16].  resident_equivalent_ms = 5000 x (forward+save + adjoint) "
                   "kernel time of a short run with resident snapshots; the overhead is the second forward sweep of the checkpointed "
                   "backward pass."}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: {"value": v["value"], "ms_per_step": v["ms_per_step"], "overhead": v["checkpointing_overhead"],
                          "kernels": v["kernels_us_per_step"], "verified": v["check"].get("verified")}
                      for k, v in doc.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
