"""Per-phase cycle table from the time stamps an ablation build of the elastic single-launch forward kernel
writes (MIFWI_LIB=.../libmifwi_ablations.so MIFWI_EL_CL_TRACE=<file>): one workgroup (slab NW/2 of the first
shot), steps 64..127, 8 waves x 12 stamps.  usage: python tools/cluster_trace.py <file>"""
import sys
import numpy as np

NAMES = ["V interior", "poll S halo", "wait barrier A", "V boundary", "wait barrier B", "S interior", "poll V halo",
         "receivers", "wait barrier C", "S boundary", "wait barrier D"]
NAMES_AC = ["sampling", "interior slots", "poll halo", "wait barrier A", "slot 0 (boundary rows)", "wait barrier B",
            "adjoint: inject + image", "publish", "snapshots + prefetch", "collective check"]
NAMES_ADJ = ["A: E + publish", "wait barrier 1", "B interior", "poll E halo", "wait barrier 2", "B boundary",
             "wait barrier 3", "receivers", "C: D + publish + grad", "wait barrier 4", "D interior", "poll D halo",
             "request S", "wait barrier 5", "D boundary"]


def main(path, brief=False):
    blocks, cur = [], []
    for line in open(path):
        if line.startswith("#"):
            if cur:
                blocks.append((head, cur))
            head, cur = line.strip(), []
        else:
            cur.append([int(x) for x in line.split()])
    if cur:
        blocks.append((head, cur))
    for head, rows in blocks:
        adj = "adj" in head
        names = NAMES_AC if "ac_cluster" in head else NAMES_ADJ if adj else NAMES
        nw = 16 if "waves=16" in head else 8
        a = np.array(rows, dtype=np.int64).reshape(64, nw, 16)[:, :, :len(names) + 1]
        if not a.any():
            continue
        a = a[4:60]                                        # steps with every stamp written
        d = np.diff(a, axis=2).astype(np.float64)          # [step][wave][phase]
        step = (a[1:, :, 0] - a[:-1, :, 0]).mean()
        if brief:                                          # one line per kernel: phase means over the waves
            m = d.mean(axis=(0, 1))
            print(head.split("|")[0].strip("# ")[:32], "step %.0f |" % step, " ".join("%s %.0f" % (nm.split(":")[0][:14], x) for nm, x in zip(names, m)))
            continue
        print(head, "| s_memtime ticks per step %.0f" % step)
        print("%-22s" % "phase" + "".join("%7s" % ("w%d" % w) for w in range(a.shape[1])) + "    max")
        for k, nm in enumerate(names):
            m = d[:, :, k].mean(axis=0)
            print("%-22s" % nm + "".join("%7.0f" % x for x in m) + "%7.0f" % m.max())
        print("%-22s" % "sum" + "".join("%7.0f" % x for x in d.mean(axis=0).sum(axis=1)))


if __name__ == "__main__":
    main(sys.argv[1], "--brief" in sys.argv)
