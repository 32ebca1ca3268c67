#!/bin/bash
# usage: tools/ab_lib.sh LIB ...  - the headline workload (or WL / GRID / SHOTS / NT) once per library build (MIFWI_LIB), same box;
# "-" = the in-tree libmifwi.so
ARGS="--workload ${WL:-elastic_marmousi} ${GRID:+--grid $GRID} ${SHOTS:+--shots $SHOTS} ${NT:+--nt $NT} --no-cpu-baseline --no-also --no-verify"
timeout -k 10 200 python bench.py $ARGS --steps 2 --warmup 1 > /dev/null 2>&1
for L in "$@"; do
  if [ "$L" = "-" ]; then unset MIFWI_LIB; else export MIFWI_LIB=$GRAFT_REPO_ROOT/$L; fi
  timeout -k 10 300 python bench.py $ARGS --steps ${STEPS:-8} --warmup 3 2>gpurun_out/ab_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', round(d['value']), d['ms_per_step'], {k:round(v['us_per_step'],3) for k,v in d['kernels'].items()}, 'loss %.6g rep %s' % (d['check']['loss'], d['check']['bitwise_repeatable']))"
done
