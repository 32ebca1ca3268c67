#!/usr/bin/env python3
"""Sweep the acoustic kernel's tile parameters on the bench workload (GPU box only)."""
import itertools
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(nt, reps=3):
    dev = torch.device("cuda:0")
    wl = bench.AcousticMarmousi(dev, 0, 1, nt=nt)
    wl.step(False)
    torch.cuda.synchronize()
    wl._ev = []
    for _ in range(reps):
        wl.step(True)
    torch.cuda.synchronize()
    tf, tb = wl.kernel_times()
    return tf * 1e6, tb * 1e6


if __name__ == "__main__":
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    combos = list(itertools.product([16, 32, 64], [1, 2], [1, 2, 4]))
    print("lx rz gs  fwd+save_us  adj_us")
    for lx, rz, gs in combos:
        os.environ["MIFWI_AC_LX"] = str(lx)
        os.environ["MIFWI_AC_RZ"] = str(rz)
        os.environ["MIFWI_AC_GS"] = str(gs)
        try:
            tf, tb = run(nt)
            print("%2d %d %d   %7.2f   %7.2f" % (lx, rz, gs, tf, tb), flush=True)
        except Exception as e:  # noqa: BLE001
            print(lx, rz, gs, "FAILED", e, flush=True)
