#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv):
2 x FETCH_SIZE + WRITE_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md).  Optional CELLS divides by
the cells one launch updates; with STEPS too, the kernel's TOTAL bytes are divided by CELLS x STEPS (a time
step split into several launches, e.g. shots taken a few at a time).
usage: pmc_kernel_bytes.py FETCH_DIR WRITE_DIR [CELLS [STEPS]]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            tot[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


if __name__ == "__main__":
    ft, fc = per_kernel(sys.argv[1], "FETCH_SIZE")
    wt, wc = per_kernel(sys.argv[2], "WRITE_SIZE")
    cells = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    steps = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    for k in sorted(set(ft) | set(wt), key=lambda k: -(2 * ft.get(k, 0) + wt.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        rd, wr = 2048.0 * ft.get(k, 0.0) / n, 1024.0 * wt.get(k, 0.0) / n
        if rd + wr < 1e6:
            continue
        extra = "  %.1f + %.1f B/cell" % (rd / cells, wr / cells) if cells else ""
        if cells and steps:
            extra = "  %.1f + %.1f B/cell-step" % (rd * n / (cells * steps), wr * n / (cells * steps))
        print("%-40s n=%-5d read %.3e write %.3e B/launch%s" % (k[:40], n, rd, wr, extra))
