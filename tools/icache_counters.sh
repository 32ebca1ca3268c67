#!/bin/bash
# instruction-cache counters of the single-launch kernels on C3 (one rocprofv3 --pmc pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ic
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d gpurun_out/ic -- python bench.py --workload elastic_marmousi --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify > gpurun_out/ic.log 2>&1 || { tail -3 gpurun_out/ic.log; exit 1; }
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if n.startswith(("el_cluster", "ac_cluster")):
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, d in sorted(acc.items()):
    print(n, {k: "%.4g" % v for k, v in sorted(d.items())})
PY
find gpurun_out/ic -name "*.csv" -size +100k -delete
