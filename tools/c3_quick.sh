#!/bin/bash
# quick C3 timing (+ optional phase trace of the forward time loop): tools/c3_quick.sh [trace]
timeout -k 10 300 python bench.py --workload elastic_marmousi --steps 5 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/c3.json 2> gpurun_out/c3.err || { tail -5 gpurun_out/c3.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/c3.json').read().strip().splitlines()[-1])
print(round(d['value']), {k:round(v['us_per_step'],2) for k,v in d['kernels'].items()}, d['check'])"
if [ "$1" = trace ]; then bash tools/trace_c3.sh | tail -14; fi
