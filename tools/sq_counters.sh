#!/bin/bash
# usage: tools/sq_counters.sh WORKLOAD [bench args]  - SQ instruction / wait counters of the time-loop kernels (one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WL=$1; shift
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf gpurun_out/sq_$i
  timeout -k 10 300 rocprofv3 --pmc $G --output-format csv -d gpurun_out/sq_$i -- python bench.py --workload $WL "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify > gpurun_out/sq.log 2>&1 || { tail -3 gpurun_out/sq.log; exit 1; }
  python - gpurun_out/sq_$i <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if n.startswith(("el_cluster", "ac_cluster", "el_fwd_fused", "el_adj", "el_step")):
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, d in sorted(acc.items()):
    print(n, {k: "%.4g" % v for k, v in sorted(d.items())})
PY
  find gpurun_out/sq_$i -name "*.csv" -size +100k -delete
done
