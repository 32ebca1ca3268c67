#!/bin/bash
# usage: tools/ab_bench.sh "ENV=.. ENV=.." ...   - times one bench workload (WL, GRID, SHOTS, NT from the environment; default: the SEAM-sized elastic one) once per environment string
# the first run on a fresh box is 2-3 % slow (clocks, cold caches): one throw-away run first
timeout -k 10 200 python bench.py --workload ${WL:-elastic_seam} ${GRID:+--grid $GRID} --nt 20 ${SHOTS:+--shots $SHOTS} --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2>&1
i=0
for E in "$@"; do i=$((i+1)); env $E timeout -k 10 200 python bench.py --workload ${WL:-elastic_seam} ${GRID:+--grid $GRID} --nt ${NT:-90} ${SHOTS:+--shots $SHOTS} --steps 5 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/ab_$i.json 2>gpurun_out/ab_err.log; python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), {k:round(v['avg_step_s']*1e6,1) for k,v in d['kernels'].items()}, d['check'])" gpurun_out/ab_$i.json "$E"; done
