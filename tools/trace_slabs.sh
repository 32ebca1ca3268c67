#!/bin/bash
# phase table of the elastic single-launch kernels on C3 for every slab of the first shot in turn (ablation build; the
# dbg bits 8-11 select the traced slab).  Per slab: mean over waves of every phase - which slab waits in which poll
for S in ${SLABS:-0 1 2 3 4 5 6 7}; do
  rm -f gpurun_out/cl_trace.txt
  MIFWI_LIB=$GRAFT_REPO_ROOT/physicsbasedfwi2_amd/libmifwi_ablations.so MIFWI_EL_CL_TRACE=$GRAFT_REPO_ROOT/gpurun_out/cl_trace.txt MIFWI_EL_CL_DBG=$(((S + 1) << 8)) \
    timeout -k 10 300 python bench.py --workload elastic_marmousi ${SHOTS:+--shots $SHOTS} --nt 400 --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-verify --timing-only > gpurun_out/trace_bench.json 2> gpurun_out/trace.err || { tail -5 gpurun_out/trace.err; exit 1; }
  echo "== slab $S"; python tools/cluster_trace.py gpurun_out/cl_trace.txt ${BRIEF---brief}
done
