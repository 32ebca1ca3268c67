import sys, numpy as np
blocks, cur, head = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur: blocks.append((head, cur))
        head, cur = line.strip(), []
    else:
        cur.append([int(x) for x in line.split()])
if cur: blocks.append((head, cur))
for head, rows in blocks:
    if "adj" not in head: continue
    a = np.array(rows, dtype=np.int64).reshape(64, 8, 16)[4:60, :, :13]
    d = np.diff(a, axis=2).mean(axis=0)
    names = ["A", "bar1", "B int", "pollE", "bar2", "B bnd", "bar3", "recv:zero", "recv:bar", "recv:atom", "recv:bar2", "recv:read"]
    print(head)
    for k, nm in enumerate(names): print("%-10s" % nm, " ".join("%6.0f" % x for x in d[:, k]))
