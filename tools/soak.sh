#!/bin/bash
# soak of the single-launch kernels: many gradient passes, results compared bit for bit pass to pass (bench's
# `bitwise_repeatable`), then two such jobs at once on the one GPU (workgroups of one job may not be resident when the
# other holds the CUs: hand-off time-outs and the fall-back happen for real; results must not change)
timeout -k 10 400 python bench.py --workload elastic_marmousi --steps 150 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/soak_el.json 2> gpurun_out/soak_el.err || { tail -3 gpurun_out/soak_el.err; exit 1; }
timeout -k 10 400 python bench.py --workload acoustic_marmousi --steps 300 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/soak_ac.json 2> gpurun_out/soak_ac.err || { tail -3 gpurun_out/soak_ac.err; exit 1; }
python - <<'PY'
import json
for n in ("el", "ac"):
    d = json.loads(open("gpurun_out/soak_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["steps"], "passes", round(d["value"]), d["check"])
PY
echo "--- two jobs at once"
timeout -k 10 600 python bench.py --workload elastic_marmousi --steps 20 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/soak_2a.json 2> gpurun_out/soak_2a.err &
P1=$!
timeout -k 10 600 python bench.py --workload acoustic_marmousi --steps 40 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/soak_2b.json 2> gpurun_out/soak_2b.err &
P2=$!
wait $P1; R1=$?; wait $P2; R2=$?
echo "exit codes $R1 $R2"
python - <<'PY'
import json
for n in ("2a", "2b"):
    try:
        d = json.loads(open("gpurun_out/soak_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), d["check"])
    except Exception as e:
        print(n, "no result", e)
PY
grep -h "falling back" gpurun_out/soak_2a.err gpurun_out/soak_2b.err | sort | uniq -c
