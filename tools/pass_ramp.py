#!/usr/bin/env python3
"""Wall time per gradient pass of the elastic headline workload, pass by pass (clock ramp, host-side overheads of the
bench loop): python tools/pass_ramp.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.ElasticMarmousi(dev, 0, 1)


def loop(n, body, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    print("%-46s %.2f ms per pass" % (label, (time.perf_counter() - t0) / n * 1e3))


ts = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wl.step(False)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("first passes, synchronised:", " ".join("%.2f" % t for t in ts))
loop(20, lambda: wl.step(False), "step(False), no sync between passes")
loop(20, lambda: wl.step(True), "step(True)")
keep = []


def with_events():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(); g, l = wl.step(True); ev[1].record(); ev[2].record(); keep.append(ev)
    return g, l


loop(20, with_events, "+ the bench's three events")
acc = []


def with_sums():
    g, l = with_events()
    acc.append(l.detach()); acc.append(g.detach().abs().sum(dtype=torch.float64))


loop(20, with_sums, "+ loss / gradient-sum bookkeeping")
acc.clear()
loop(20, lambda: acc.append(with_events()[1].detach()), "events + keep the loss only")
acc.clear()
loop(20, lambda: acc.append(with_events()[0].detach()), "events + keep the gradient tensor only")
acc.clear()
loop(20, lambda: acc.append(with_events()[0].detach().abs().sum()), "events + |grad| sum in float32")
acc.clear()
loop(20, lambda: acc.append(with_events()[0].detach().abs().sum(dtype=torch.float64)), "events + |grad| sum in float64")
