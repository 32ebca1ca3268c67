#!/bin/bash
timeout -k 10 300 python bench.py --workload acoustic_marmousi --steps 5 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/c2.json 2> gpurun_out/c2.err || { tail -5 gpurun_out/c2.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/c2.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], {k:round(v['us_per_step'],2) for k,v in d['kernels'].items()}, d['check'])"
