#!/usr/bin/env python3
"""Memory-instruction mix of the kernels of one source file whose mangled name contains a filter string:
python tools/asm_mix.py physicsbasedfwi2_amd/csrc/mifwi_acoustic.hip ac_cluster   (flat_* that should be ds_* / global_* show up here)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from physicsbasedfwi2_amd import build  # noqa: E402

src, filt = sys.argv[1], sys.argv[2:]
flags = [f for f in build.FLAGS if f != "-shared"]
out = "/tmp/asm_mix.s"
subprocess.run([build.HIPCC] + flags + ["--cuda-device-only", "-S", "-o", out, src], check=True, capture_output=True)
name, rows = None, collections.OrderedDict()
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        continue
    if line.startswith("\t.amdhsa_kernel") or line.startswith(".Lfunc_end"):
        name = None
    if name and all(f in name for f in filt):
        m = re.match(r"^\s+(flat_load|flat_store|global_load|global_store|ds_read|ds_write|scratch_load|scratch_store|v_rcp|s_barrier|v_readlane|v_writelane)", line)
        if m:
            rows.setdefault(name, collections.Counter())[m.group(1)] += 1
for n, c in rows.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    print(re.sub(r"\(anonymous namespace\)::|void |\(.*", "", d), dict(c))
