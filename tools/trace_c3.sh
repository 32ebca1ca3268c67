#!/bin/bash
# phase time stamps of the elastic single-launch forward kernel on the C3 workload (ablation build)
rm -f gpurun_out/cl_trace.txt
MIFWI_LIB=$GRAFT_REPO_ROOT/physicsbasedfwi2_amd/libmifwi_ablations.so MIFWI_EL_CL_TRACE=$GRAFT_REPO_ROOT/gpurun_out/cl_trace.txt MIFWI_EL_CL_DBG=${DBG:-0} \
  timeout -k 10 300 python bench.py --workload elastic_marmousi ${SHOTS:+--shots $SHOTS} --nt 400 --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-verify --timing-only > gpurun_out/trace_bench.json 2> gpurun_out/trace.err || { tail -5 gpurun_out/trace.err; exit 1; }
python tools/cluster_trace.py gpurun_out/cl_trace.txt | tail -40
