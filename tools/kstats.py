#!/usr/bin/env python3
"""Print the el_* / ac_* rows of a rocprofv3 --kernel-trace --stats output directory: calls, average and total time."""
import csv
import glob
import sys

for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if n.startswith(("el_", "ac_", "mat", "misfit", "reparam", "coef_")):
                print("%-30s calls %7s avg %9.1f us total %9.2f ms" % (n[:30], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                       float(r["TotalDurationNs"]) / 1e6))
