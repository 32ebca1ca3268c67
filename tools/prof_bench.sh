#!/bin/bash
# usage: tools/prof_bench.sh "ENV=.." ...  - rocprofv3 kernel stats of one bench workload (WL, GRID, SHOTS, NT) per environment string ("X=" = defaults)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for E in "$@"; do i=$((i+1)); export $E; rm -rf gpurun_out/prof_$i
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$i -- python bench.py --workload ${WL:-elastic_seam} ${GRID:+--grid $GRID} --nt ${NT:-60} ${SHOTS:+--shots $SHOTS} --steps 3 --warmup 3 --no-cpu-baseline --no-also --no-verify > gpurun_out/prof.log 2>&1 || { tail -5 gpurun_out/prof.log; exit 1; }
echo "== $E"; find gpurun_out/prof_$i -name "*kernel_stats.csv" | xargs head -8 | cut -d, -f1,2,4 | grep -v "^\"Name" | sed 's/(anonymous namespace):://g' | cut -c1-110; for v in $(echo $E | tr ' ' '\n' | cut -d= -f1); do unset $v; done; done
