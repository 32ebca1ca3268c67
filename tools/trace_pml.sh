#!/bin/bash
# sub-phase stamps of the C-PML block of the acoustic single-launch kernels on C2 (ablation build), edge slab 0 and an
# interior slab, at the layer widths in $WIDTHS.  ABL_LIB = the ablation build to trace (default: the in-tree one)
ABL=${ABL_LIB:-physicsbasedfwi2_amd/libmifwi_ablations.so}
for W in ${WIDTHS:-10 20}; do
 for S in ${SLABS:-0 3}; do
  rm -f gpurun_out/cl_trace_ac.txt
  BENCH_ABSORBING=cpml BENCH_PML_WIDTH=$W MIFWI_LIB=$GRAFT_REPO_ROOT/$ABL MIFWI_AC_CL_TRACE=$GRAFT_REPO_ROOT/gpurun_out/cl_trace_ac.txt MIFWI_AC_CL_DBG=$(((S + 1) << 8)) \
    timeout -k 10 300 python bench.py --workload acoustic_marmousi --nt 400 --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-verify --timing-only > gpurun_out/trace_bench.json 2> gpurun_out/trace.err || { tail -5 gpurun_out/trace.err; exit 1; }
  for M in 1 2; do
    echo "== W=$W slab $S mode $M"; python tools/trace_pml.py gpurun_out/cl_trace_ac.txt $M
  done
  python tools/cluster_trace.py gpurun_out/cl_trace_ac.txt --brief | tail -30
 done
done
