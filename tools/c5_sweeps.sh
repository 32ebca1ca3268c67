#!/bin/bash
# usage: tools/c5_sweeps.sh "ENV=.." ...  - per-kernel times of a short time-checkpointed run of the C5 share (rocprofv3 --stats):
# el_fwd_fused<0> = the no-save sweep, <1> = the saving sweep, el_adj_s / el_adj_v
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for E in "$@"; do i=$((i+1)); rm -rf gpurun_out/c5s_$i
  env $E MIFWI_EL_SNAPSHOT_BUDGET_GB=60 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5s_$i -- python bench.py --workload elastic_seam --nt 150 --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-verify > gpurun_out/c5s.log 2>&1 || { tail -3 gpurun_out/c5s.log; continue; }
  echo "== $E"; python - gpurun_out/c5s_$i <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if n.startswith("el_"):
            print("  %-28s calls %6s avg %9.1f us total %9.1f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  find gpurun_out/c5s_$i -name "*kernel_trace.csv" -delete
done
