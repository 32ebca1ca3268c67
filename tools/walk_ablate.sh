#!/bin/bash
# usage: tools/walk_ablate.sh  - traffic and time of el_adj_walk with one stream switched off at a time (ablation build; wrong results)
export MIFWI_LIB=$GRAFT_REPO_ROOT/physicsbasedfwi2_amd/libmifwi_ablations.so
export MIFWI_EL_FUSED_ADJ=2 MIFWI_EL_GS=${GS:-4} MIFWI_EL_WALK_ROWS=${ROWS:-126}
for D in 0 1 2 4 8 16 32 63; do
  echo "== MIFWI_WALK_DBG=$D"
  MIFWI_WALK_DBG=$D bash tools/pmc_one.sh walk$D gpurun_out/t_walk_$D.json --workload elastic_seam --nt 24 --timing-only 2>&1 | grep -A8 "adjoint+imaging" | grep "bytes_per_cell_step"
  MIFWI_WALK_DBG=$D timeout -k 10 300 python bench.py --workload elastic_seam --nt 40 --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-verify --timing-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  adj us/step', round(d['kernels']['adjoint+imaging']['us_per_step'],1))"
done
