"""Sub-phases of the C-PML block of ac_cluster<1 | 2, ..., PML> from the stamps 1, 11, 12, 13, (15,) 14, 2 of the ablation build's trace."""
import sys
import numpy as np
rows, keep = [], False
for line in open(sys.argv[1]):
    if line.startswith("#"):
        keep = ("mode=%s" % (sys.argv[2] if len(sys.argv) > 2 else "1")) in line
        continue
    if keep:
        rows.append([int(x) for x in line.split()])
a = np.array(rows, dtype=np.int64).reshape(-1, 64, 16, 16)[-1][4:60]      # [step][wave][stamp]
adjoint = len(sys.argv) > 2 and sys.argv[2] == "2"
seq = [1, 11, 12, 13, 15, 14, 2] if adjoint else [1, 11, 12, 13, 14, 2]
names = (["phase a (P, Zb)", "sync 1", "phase b (Q, Pb)", "sync 2", "phase c (e)", "sync 3", "interior updates"] if adjoint
         else ["psi loops", "sync 1", "zeta loops", "sync 2", "interior updates"])
for nm, (i, j) in zip(names, zip(seq[:-1], seq[1:])):
    d = (a[:, :, j] - a[:, :, i]).mean(axis=0)
    print("%-18s" % nm + "".join("%7.0f" % x for x in d))
