#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of ONE bench workload (one pass with FETCH_SIZE, one with WRITE_SIZE;
--output-format csv; `bench.py --workload W [--grid G --nt N --shots S] --steps 1 --warmup 0 --no-cpu-baseline
--no-also --no-verify`) into an entry of profiles/<round>_pmc_traffic.json: measured HBM bytes per INTERIOR
cell-step of the forward(+save) and adjoint(+imaging) time loops = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 /
(interior cells x steps) - FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, section HBM).
A time loop may be one launch (single-launch kernels) or many (two launches per step, a few shots per pass):
all launches of the label's kernels are summed.  The file carries the fingerprint of the kernel sources;
bench.py quotes it only while the fingerprint matches.

usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json --workload W [--grid NZxNX] [--nt N] [--shots S]
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

# label -> regexes on the kernel name (namespace and "void " stripped).  SAVE = true instantiations only: the
# observed-data forward run of the bench set-up (SAVE = false) is not part of the gradient pass.
LABELS = {
    "acoustic": {
        "forward+save": [r"^ac_cluster<1,", r"^ac_step<\d+, \d+, true, false>"],
        "adjoint+imaging": [r"^ac_cluster<2,", r"^ac_step<\d+, \d+, false, true>"],
    },
    "elastic": {
        "forward+save": [r"^el_cluster_fwd<true", r"^el_step_v<\d+, \d+, [12]>", r"^el_step_s<\d+, \d+, [12]>",
                         r"^el_fwd_fused<[12]>"],
        "adjoint+imaging": [r"^el_cluster_adj<", r"^el_adj_s<", r"^el_adj_v$", r"^el_adj_fused<", r"^el_adj_walk<", r"^el_inject_adjsrc$"],
    },
}


def clean(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def totals(path, counter):
    acc = defaultdict(float)
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + path)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[clean(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("out")
    ap.add_argument("--workload", required=True, choices=sorted(bench.WORKLOADS))
    ap.add_argument("--grid", default="")
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--shots", type=int, default=0)
    ap.add_argument("--suffix", default="", help="appended to the entry's key (\"_cpml\": the acoustic pass ran with the C-PML)")
    a, _ = ap.parse_known_args()            # tools/pmc_one.sh hands over the bench arguments as they are
    cls = bench.WORKLOADS[a.workload]
    nz, nx = (int(v) for v in a.grid.lower().split("x")) if a.grid else (cls.nz, cls.nx)
    nt, ns = a.nt or cls.nt, a.shots or cls.shots_per_gpu
    physics = a.workload.split("_")[0]
    key = "%s_%dx%d%s" % (physics, nz, nx, "_fs" if getattr(cls, "free_surface", False) else "") + a.suffix
    fetch, write = totals(a.fetch_dir, "FETCH_SIZE"), totals(a.write_dir, "WRITE_SIZE")
    try:
        with open(a.out) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        doc = {}
    sha = bench.csrc_sha16()
    if doc.get("csrc_sha16") != sha:
        doc = {}                                    # entries of other kernels do not mix with these
    doc["csrc_sha16"] = sha
    doc["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                   cwd=bench.ROOT).stdout.strip() or doc.get("commit", "")
    entry = {}
    for label, pats in LABELS[physics].items():
        names = sorted(n for n in set(fetch) | set(write) if any(re.search(p, n) for p in pats))
        if not names:
            continue
        rd = sum(2.0 * 1024.0 * fetch.get(n, 0.0) for n in names)
        wr = sum(1024.0 * write.get(n, 0.0) for n in names)
        entry[label] = {"kernels": names, "read_bytes": rd, "write_bytes": wr,
                        "bytes_per_cell_step": (rd + wr) / (float(nz) * nx * ns * nt),
                        "read_bytes_per_cell_step": rd / (float(nz) * nx * ns * nt),
                        "write_bytes_per_cell_step": wr / (float(nz) * nx * ns * nt),
                        "units": "%d x %d interior cells x %d shots x %d steps, one gradient pass" % (nz, nx, ns, nt),
                        "note": "2*FETCH_SIZE + WRITE_SIZE (KiB counters), all launches of these kernels"}
    if not entry:
        raise SystemExit("no time-loop kernel of %s found in the counter files" % a.workload)
    doc[key] = entry
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print(json.dumps({key: entry}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
