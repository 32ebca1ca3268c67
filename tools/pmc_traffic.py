#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0` (one with FETCH_SIZE, one with
WRITE_SIZE; --output-format csv) into profiles/r01_pmc_traffic.json: HBM bytes per cell-step of the
time-loop kernels, 2 x FETCH_SIZE + WRITE_SIZE (both in KiB; FETCH_SIZE counts 64 B per 128-B
request on gfx950 - MI355X_MICROARCH.md, section HBM).
usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

KERNELS = {   # kernel-name prefix -> (workload class, bench kernel label)
    "ac_cluster<1,": (bench.AcousticMarmousi, "forward+save"),
    "ac_cluster<2,": (bench.AcousticMarmousi, "adjoint+imaging"),
    "el_cluster_fwd<true": (bench.ElasticMarmousi, "forward+save"),
    "el_cluster_adj<": (bench.ElasticMarmousi, "adjoint+imaging"),
}


def totals(path, counter):
    acc = defaultdict(float)
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + path)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            for pre in KERNELS:
                if name.startswith(pre):
                    acc[pre] += float(r["Counter_Value"])
    return acc


if __name__ == "__main__":
    fetch, write = totals(sys.argv[1], "FETCH_SIZE"), totals(sys.argv[2], "WRITE_SIZE")
    out = {}
    for pre, (cls, label) in KERNELS.items():
        if pre not in fetch and pre not in write:
            continue
        P = cls.pml if cls is bench.AcousticMarmousi else 0
        cells = (cls.nz + 2 * P) * (cls.nx + 2 * P) * cls.shots_per_gpu
        steps = cls.nt
        rd, wr = 2.0 * fetch.get(pre, 0.0) * 1024.0, write.get(pre, 0.0) * 1024.0
        out.setdefault(cls.name, {})[label] = {
            "kernel": pre, "read_bytes": rd, "write_bytes": wr,
            "bytes_per_cell_step": (rd + wr) / (cells * steps),
            "note": "one pass of the time loop over all shots; 2*FETCH_SIZE + WRITE_SIZE (KiB counters)"}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))
