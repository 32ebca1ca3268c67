#!/usr/bin/env python3
"""Where the non-kernel time of an elastic gradient pass goes: wall time of the pass against the durations of its two
time-loop kernels (HIP events), plus a cProfile of the host side.  usage: python tools/host_gaps.py [shots]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

shots = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
wl = bench.ElasticMarmousi(dev, 0, 1, shots=shots)
for _ in range(3):
    wl.step(False)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    wl.step(True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
tf, tb = wl.kernel_times()
print("shots %d: wall %.3f ms per pass, forward loop %.3f ms, adjoint loop %.3f ms, everything else %.3f ms"
      % (shots, wall * 1e3, tf * wl.nt * 1e3, tb * wl.nt * 1e3, (wall - (tf + tb) * wl.nt) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    wl.step(False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
