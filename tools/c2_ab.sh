#!/bin/bash
# usage: tools/c2_ab.sh "ENV=.." ...  - the C2 workload (acoustic 174x500 x 29 shots x 2000 steps) once per environment string
ARGS="--workload acoustic_marmousi --no-cpu-baseline --no-also"
timeout -k 10 200 python bench.py $ARGS --steps 2 --warmup 1 > /dev/null 2>&1
for E in "$@"; do
  env $E timeout -k 10 300 python bench.py $ARGS --steps ${STEPS:-8} --warmup 3 2>gpurun_out/ab_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$E', d['config']['workload'][-8:], round(d['value']), d['ms_per_step'], {k:round(v['us_per_step'],2) for k,v in d['kernels'].items()}, 'loss %.6g rep %s ver %s' % (d['check']['loss'], d['check']['bitwise_repeatable'], d['check'].get('verified')))" || tail -3 gpurun_out/ab_err.log
done
