#!/bin/bash
# the elastic grids and batch sizes the reference's prop() variants run through the default plans:
# per-step kernel times and the gradient-pass rate, one line each
for C in "100x300 5" "100x300 6" "150x294 6" "170x396 6" "190x324 4" "100x300 32" "150x294 28" "170x396 32"; do
  set -- $C
  timeout -k 10 300 python bench.py --workload elastic_marmousi --grid $1 --shots $2 --nt ${NT:-2500} --steps 3 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/refgrid.json 2> gpurun_out/refgrid.err || { echo "$C FAILED"; tail -3 gpurun_out/refgrid.err; continue; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/refgrid.json').read().strip().splitlines()[-1])
k=d['kernels']
print('%-9s %2s shots | %7.1f Mcells*steps/s | fwd %6.2f us adj %6.2f us per step | %s | fallbacks %s verified %s' % (sys.argv[1], sys.argv[2], d['value'], k['forward+save']['us_per_step'], k['adjoint+imaging']['us_per_step'], d['config'].get('kernel_family','?'), d['check'].get('fallbacks'), d['check'].get('verified')))" $1 $2
done
