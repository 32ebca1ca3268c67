#!/bin/bash
# soak of the single-launch kernels: many gradient passes of ONE job, results compared bit for bit pass to pass (bench's
# `bitwise_repeatable`).  The hand-off time-out and the fall-back are exercised deterministically by
# tests/test_fallback_gpu.py (ablation bits 64 / 128 in a child process, MIFWI_TEST_FAKE_TIMEOUT=1/2), not by
# crowding the GPU with a second job: that never provoked one (profiles/r02_soak.txt: fallbacks 0 in both jobs).
timeout -k 10 400 python bench.py --workload elastic_marmousi --steps 150 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/soak_el.json 2> gpurun_out/soak_el.err || { tail -3 gpurun_out/soak_el.err; exit 1; }
timeout -k 10 400 python bench.py --workload acoustic_marmousi --steps 300 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/soak_ac.json 2> gpurun_out/soak_ac.err || { tail -3 gpurun_out/soak_ac.err; exit 1; }
python - <<'PY'
import json
for n in ("el", "ac"):
    d = json.loads(open("gpurun_out/soak_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["steps"], "passes", round(d["value"]), d["check"])
PY
