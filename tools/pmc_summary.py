#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    for f in glob.glob(path + "/*/*_counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, cs in acc.items():
            if not name.startswith(("el_", "ac_")):
                continue
            n = max(len(v) for v in cs.values())
            print("%-28s n=%d " % (name[:28], n) + " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(cs.items())))
