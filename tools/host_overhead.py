#!/usr/bin/env python3
"""Where the host time of one bench iteration goes (cProfile over a few acoustic steps; GPU box only)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "acoustic_marmousi"
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS[name](dev, 0, 1)
    wl.step(False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        wl.step(False)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
