"""Copy the round's summaries from gpurun_out/ (scratch, merged back from the GPU box) into profiles/ (tracked) and stamp
the commit they were measured at: the GPU box has no .git, so its `commit` fields arrive empty.  The kernel-source
fingerprint (`csrc_sha16`) inside each file is what bench.py trusts; the commit is for the reader.
usage: python tools/adopt_profiles.py [r03] [commit]"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
COMMIT = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(
    ["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
NAMES = ["bench_default_output.json", "bench_default_kernel_stats.csv", "pmc_traffic.json", "latency_floor.json",
         "issue_counters.json",
         "seam_bench_output.json", "seam_kernel_stats.csv"]

for n in NAMES:
    src = os.path.join(ROOT, "gpurun_out", f"{R}_{n}")
    dst = os.path.join(ROOT, "profiles", f"{R}_{n}")
    if not os.path.exists(src):
        print("missing", src)
        continue
    if n in ("pmc_traffic.json", "latency_floor.json", "issue_counters.json"):
        doc = json.load(open(src))
        if not doc.get("commit"):
            doc["commit"] = COMMIT
        json.dump(doc, open(dst, "w"), indent=1, sort_keys=True)
        open(dst, "a").write("\n")
    else:
        shutil.copyfile(src, dst)
    print("adopted", dst)
