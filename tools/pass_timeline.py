#!/usr/bin/env python3
"""Timeline of ONE gradient pass from a rocprofv3 --kernel-trace run of bench.py: every kernel between the start of the
last forward time loop and the end of the last adjoint time loop, with its start offset, duration and the idle gap in
front of it.  usage: python tools/pass_timeline.py DIR [elastic|acoustic]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
phys = sys.argv[2] if len(sys.argv) > 2 else "elastic"
fwd_re, adj_re = (r"el_cluster_fwd<true", r"el_cluster_adj<") if phys == "elastic" else (r"ac_cluster<1,", r"ac_cluster<2,")
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
clean = lambda n: re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)[:70]
fw = [i for i, r in enumerate(rows) if re.search(fwd_re, r[2])]
ad = [i for i, r in enumerate(rows) if re.search(adj_re, r[2])]
i0, i1 = fw[-1], ad[-1]
prev_fw_end = rows[ad[-2]][1] if len(ad) > 1 else rows[i0][0]
t0 = rows[i0][0]
print("pass: %.3f ms from the start of the forward loop to the end of the adjoint loop; %.3f ms from the end of the previous adjoint loop"
      % ((rows[i1][1] - t0) / 1e6, (rows[i1][1] - prev_fw_end) / 1e6))
last_end = prev_fw_end
j0 = ad[-2] + 1 if len(ad) > 1 else i0
busy = 0
for s, e, n in rows[j0:i1 + 1]:
    print("%10.3f ms  +%8.1f us gap  %10.1f us  %s" % ((s - t0) / 1e6, (s - last_end) / 1e3, (e - s) / 1e3, clean(n)))
    last_end = max(last_end, e)
    busy += e - s
print("kernels busy %.3f ms" % (busy / 1e6))
