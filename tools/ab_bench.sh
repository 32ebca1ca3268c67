#!/bin/bash
# usage: tools/ab_bench.sh "ENV=.. ENV=.." ...   - times one bench workload (WL, GRID, SHOTS, NT from the environment; default: the SEAM-sized elastic one) once per environment string ("X=" = defaults)
# the first run on a fresh box is 2-3 % slow (clocks, cold caches): one throw-away run first
ARGS="--workload ${WL:-elastic_seam} ${GRID:+--grid $GRID} ${SHOTS:+--shots $SHOTS} --no-cpu-baseline --no-also --no-verify"
timeout -k 10 200 python bench.py $ARGS --nt 20 --steps 2 --warmup 1 > /dev/null 2>&1
i=0
for E in "$@"; do i=$((i+1)); env $E timeout -k 10 300 python bench.py $ARGS --nt ${NT:-90} --steps ${STEPS:-5} --warmup 3 > gpurun_out/ab_$i.json 2>gpurun_out/ab_err.log || { echo "FAILED: $E"; tail -5 gpurun_out/ab_err.log; continue; }; python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), {k:round(v['us_per_step'],1) for k,v in d['kernels'].items()}, 'frac', {k:round(v['frac_of_hbm_peak'],3) for k,v in d['kernels'].items()}, 'loss %.6g gsum %.6g rep %s' % (d['check']['loss'], d['check']['grad_abs_sum'], d['check']['bitwise_repeatable']))" gpurun_out/ab_$i.json "$E"; done
