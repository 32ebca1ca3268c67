#!/bin/bash
# Round profile set, run on the GPU box (gpurun -- bash tools/profile_round.sh [r03]); results under gpurun_out/,
# to be copied into profiles/:
#   <R>_bench_default_output.json        python bench.py
#   <R>_bench_default_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command (no CPU baseline)
#   <R>_pmc_traffic.json                 FETCH_SIZE / WRITE_SIZE passes of the four bench workloads (tools/pmc_traffic.py)
#   <R>_latency_floor.json               tools/latency_floor.py (ablation build)
#   <R>_issue_counters.json              SQ instruction counts of the single-launch kernels (tools/issue_counters.py)
#   <R>_seam_kernel_stats.csv            rocprofv3 stats of the SEAM-sized sample
R=${1:-r04}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/latency_floor.py $O/${R}_latency_floor.json > $O/latency_floor.log 2>&1 || { tail -5 $O/latency_floor.log; exit 1; }
cp $O/${R}_latency_floor.json profiles/${R}_latency_floor.json        # bench.py reads profiles/
pmc() {  # name, bench args..., -- pmc_traffic args
    local name=$1; shift
    local bargs=() ; while [ "$1" != "--" ]; do bargs+=("$1"); shift; done; shift
    for C in FETCH_SIZE WRITE_SIZE; do
        rm -rf $O/pmc_${name}_$C
        timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${name}_$C -- python bench.py "${bargs[@]}" --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; return 1; }
    done
    python tools/pmc_traffic.py $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $O/${R}_pmc_traffic.json "$@" > /dev/null || return 1
    find $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE -name "*.csv" -size +200k -delete
}
issue() {  # workload: one --pmc pass of the SQ instruction counters
    rm -rf $O/pmc_issue_$1
    timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_issue_$1 -- python bench.py --workload $1 --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; return 1; }
    python tools/issue_counters.py $O/pmc_issue_$1 $O/${R}_issue_counters.json --workload $1 > /dev/null || return 1
    find $O/pmc_issue_$1 -name "*.csv" -size +200k -delete
}
issue elastic_marmousi && BENCH_ABSORBING=sponge BENCH_PML_WIDTH=20 issue acoustic_marmousi || exit 1
# the acoustic single-launch kernels with the second-order C-PML (bench.py's `also` entry "..._cpml20")
rm -rf $O/pmc_issue_cpml
BENCH_ABSORBING=cpml timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_issue_cpml -- python bench.py --workload acoustic_marmousi --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
python tools/issue_counters.py $O/pmc_issue_cpml $O/${R}_issue_counters.json --workload acoustic_marmousi --suffix _cpml > /dev/null || exit 1
find $O/pmc_issue_cpml -name "*.csv" -size +200k -delete
cp $O/${R}_issue_counters.json profiles/${R}_issue_counters.json
pmc el100 --workload elastic_marmousi -- --workload elastic_marmousi &&
BENCH_ABSORBING=sponge BENCH_PML_WIDTH=20 pmc ac174 --workload acoustic_marmousi -- --workload acoustic_marmousi &&
pmc ac174c --workload acoustic_marmousi -- --workload acoustic_marmousi --suffix _cpml &&
pmc el350 --workload elastic_marmousi --grid 350x1700 --nt 60 -- --workload elastic_marmousi --grid 350x1700 --nt 60 &&
pmc seam --workload elastic_seam --nt 24 -- --workload elastic_seam --nt 24 || exit 1
cp $O/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json
python bench.py --detail $O/${R}_bench_default_detail.json > $O/${R}_bench_default_output.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
rm -rf $O/prof_default
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python bench.py --no-cpu-baseline > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
find $O/prof_default -name "*kernel_stats.csv" -exec cp {} $O/${R}_bench_default_kernel_stats.csv \;
find $O/prof_default -name "*kernel_trace.csv" -delete
rm -rf $O/prof_seam
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seam -- python bench.py --workload elastic_seam --nt 90 --steps 3 --warmup 3 --no-cpu-baseline --no-also --no-verify > $O/${R}_seam_bench_output.json 2> $O/prof.log || { tail -5 $O/prof.log; exit 1; }
find $O/prof_seam -name "*kernel_stats.csv" -exec cp {} $O/${R}_seam_kernel_stats.csv \;
find $O/prof_seam -name "*kernel_trace.csv" -delete
echo PROFILE_ROUND_DONE
