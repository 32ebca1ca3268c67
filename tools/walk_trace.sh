#!/bin/bash
# usage: tools/walk_trace.sh  - cycles per phase of el_adj_walk (ablation build, s_memtime stamps), with all streams and with none
export MIFWI_LIB=$GRAFT_REPO_ROOT/physicsbasedfwi2_amd/libmifwi_ablations.so
export MIFWI_EL_FUSED_ADJ=2 MIFWI_EL_GS=${GS:-4} MIFWI_EL_WALK_ROWS=${ROWS:-126}
rm -f gpurun_out/walk_trace.txt
for D in 0 63 1 16; do
  echo "MIFWI_WALK_DBG=$D" >> gpurun_out/walk_trace.txt
  MIFWI_WALK_DBG=$D MIFWI_WALK_TRACE=gpurun_out/walk_trace.txt timeout -k 10 300 python bench.py --workload ${WL:-elastic_seam} ${GRID:+--grid $GRID} ${SHOTS:+--shots $SHOTS} --nt 12 --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-verify --timing-only > /dev/null 2>gpurun_out/walk_trace.err || tail -3 gpurun_out/walk_trace.err
done
cat gpurun_out/walk_trace.txt
