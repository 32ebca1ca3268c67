#!/usr/bin/env python3
"""Instruction counts of the single-launch time-loop kernels from ONE rocprofv3 --pmc pass of a bench workload
(`rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d DIR -- python bench.py
--workload W --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify`) -> an entry of
profiles/<round>_issue_counters.json: vector / scalar / LDS instructions per wave and time step.  bench.py turns them
into the bound it quotes for an LDS-resident time loop: issue cycles of the SHIPPED kernel (guide: a wave's vector
instruction costs 4 issue cycles, MI355X_MICROARCH.md "vector-instruction ISSUE cost") over the cycles of a step.

usage: issue_counters.py PMC_DIR OUT.json --workload W [--nt N] [--suffix _cpml]
"""
import argparse
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pmc_traffic  # noqa: E402

LABELS = {"acoustic": {"forward+save": [r"^ac_cluster<1,"], "adjoint+imaging": [r"^ac_cluster<2,"]},
          "elastic": {"forward+save": [r"^el_cluster_fwd<true"], "adjoint+imaging": [r"^el_cluster_adj<"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_dir")
    ap.add_argument("out")
    ap.add_argument("--workload", required=True, choices=sorted(bench.WORKLOADS))
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--suffix", default="", help="appended to the entry's key (\"_cpml\": the pass ran with BENCH_ABSORBING=cpml)")
    a = ap.parse_args()
    cls = bench.WORKLOADS[a.workload]
    physics = a.workload.split("_")[0]
    key = "%s_%dx%d" % (physics, cls.nz, cls.nx) + a.suffix
    nt = a.nt or cls.nt
    tot = {c: pmc_traffic.totals(a.pmc_dir, c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    try:
        doc = json.load(open(a.out))
    except (OSError, ValueError):
        doc = {}
    sha = bench.csrc_sha16()
    if doc.get("csrc_sha16") != sha:
        doc = {}
    doc["csrc_sha16"] = sha
    doc["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                   cwd=bench.ROOT).stdout.strip() or doc.get("commit", "")
    entry = {}
    for label, pats in LABELS[physics].items():
        names = sorted(n for n in tot["SQ_WAVES"] if any(re.search(p, n) for p in pats))
        if not names:
            continue
        waves = sum(tot["SQ_WAVES"][n] for n in names)
        steps = nt - (1 if label.startswith("adjoint") and physics == "acoustic" else 0)
        per = {c: sum(tot[c].get(n, 0.0) for n in names) / (waves * steps) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
        entry[label] = {"kernels": names, "waves": waves, "steps": steps, "valu_per_wave_step": per["SQ_INSTS_VALU"],
                        "salu_per_wave_step": per["SQ_INSTS_SALU"], "lds_per_wave_step": per["SQ_INSTS_LDS"]}
    if not entry:
        raise SystemExit("no single-launch time-loop kernel of %s in the counter files" % a.workload)
    doc[key] = entry
    json.dump(doc, open(a.out, "w"), indent=1, sort_keys=True)
    print(json.dumps({key: entry}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
