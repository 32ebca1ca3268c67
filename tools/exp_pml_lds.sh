#!/bin/bash
# A/B on one box: the acoustic single-launch C-PML with the layer's arrays in LDS (default) and through global memory
# (MIFWI_AC_PML_LDS=0); parity tests first, then the C2 bench at layer widths 10 and 20.
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_acoustic_gpu.py tests/test_full_size_gpu.py tests/test_compat_gpu.py -x -q -k "cpml or pml or acoustic or scalar" > gpurun_out/pml_lds_tests.log 2>&1 || { tail -30 gpurun_out/pml_lds_tests.log; exit 1; }
tail -3 gpurun_out/pml_lds_tests.log
for W in 10 20; do
  for L in 1 0; do
    echo "== W=$W MIFWI_AC_PML_LDS=$L"
    MIFWI_AC_PML_LDS=$L BENCH_ABSORBING=cpml BENCH_PML_WIDTH=$W timeout -k 10 200 python bench.py --workload acoustic_marmousi --steps 3 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/pml_lds_${W}_${L}.json 2> gpurun_out/pml_lds_${W}_${L}.err || { tail -5 gpurun_out/pml_lds_${W}_${L}.err; exit 1; }
    python -c "
import json,sys
d=json.loads(open('gpurun_out/pml_lds_${W}_${L}.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], {k:(round(v['us_per_step'],2), v.get('alg_B_per_cell')) for k,v in d['kernels'].items()}, d.get('check'))
"
  done
done
