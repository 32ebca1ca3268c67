#!/usr/bin/env python3
"""Minimal driver for profiling: forward-only (no snapshots) + one gradient pass, small nt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
cls = {"elastic": bench.ElasticMarmousi, "acoustic": bench.AcousticMarmousi}[os.environ.get("PROF_WL", "elastic")]
kw = {}
if os.environ.get("TUNE_GRID") and cls is bench.ElasticMarmousi:
    kw["grid"] = tuple(int(v) for v in os.environ["TUNE_GRID"].split("x"))
if os.environ.get("TUNE_SHOTS"):
    kw["shots"] = int(os.environ["TUNE_SHOTS"])
wl = cls(torch.device("cuda:0"), 0, 1, nt=nt, **kw)     # constructor runs one forward (no save)
if mode == "both":
    wl.step(False)
torch.cuda.synchronize()
print("done")
