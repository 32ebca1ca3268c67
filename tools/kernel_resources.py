#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage):
python tools/kernel_resources.py physicsbasedfwi2_amd/csrc/mifwi_elastic.hip [filter] [-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from physicsbasedfwi2_amd import build  # noqa: E402

src = sys.argv[1]
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
flags = [f for f in build.FLAGS if f != "-shared"] + extra
out = subprocess.run([build.HIPCC] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", name)
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-46s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "sSpill", "vSpill", "scratch", "occ", "LDS"))
for k, v in rows.items():
    if filt and not any(f in k for f in filt):
        continue
    print("%-46s %5d %5d %6d %6d %7d %4d %7d" % (k[:46], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("SGPRs Spill", -1),
                                                 v.get("VGPRs Spill", -1), v.get("ScratchSize", -1),
                                                 v.get("Occupancy", -1), v.get("LDS Size", -1)))
