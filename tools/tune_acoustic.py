#!/usr/bin/env python3
"""A/B the acoustic kernels' tuning knobs on the bench workload (GPU box only).
usage: tune_acoustic.py NT "K1=V1,K2=V2;K1=V3,..."   (each ';'-separated group is one configuration,
e.g. "MIFWI_AC_CLUSTER=0,MIFWI_AC_LX=64,MIFWI_AC_RZ=2;MIFWI_AC_CLUSTER=1")
"""
import os
import sys

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
    del wl
    torch.cuda.empty_cache()
    return tf * 1e6, tb * 1e6


if __name__ == "__main__":
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    configs = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
    print("config -> fwd+save_step_us adj_step_us")
    for cfg in configs:
        keys = []
        for kv in [c for c in cfg.split(",") if c]:
            k, v = kv.split("=")
            os.environ[k] = v
            keys.append(k)
        try:
            tf, tb = run(nt)
            print("%-50s %8.2f %8.2f" % (cfg or "(default)", tf, tb), flush=True)
        except Exception as e:  # noqa: BLE001
            print(cfg, "FAILED", repr(e)[:300], flush=True)
        for k in keys:
            os.environ.pop(k, None)
