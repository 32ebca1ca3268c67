#!/bin/bash
# A/B on one box: LDS row pitch of the elastic single-launch forward kernel (MIFWI_EL_PL_SKEW), time and bank conflicts
for S in 0 1; do
  echo "== MIFWI_EL_PL_SKEW=$S"
  MIFWI_EL_PL_SKEW=$S bash tools/c3_quick.sh || exit 1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in 0 1; do
  export MIFWI_EL_PL_SKEW=$S
  rm -rf gpurun_out/sqx
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/sqx -- python bench.py --workload elastic_marmousi --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify > gpurun_out/sq.log 2>&1 || { tail -3 gpurun_out/sq.log; exit 1; }
  python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/sqx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if n.startswith("el_cluster"):
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
print("SKEW", os.environ["MIFWI_EL_PL_SKEW"])
for n, d in sorted(acc.items()):
    print(" ", n, {k: "%.4g" % v for k, v in sorted(d.items())})
PY
  find gpurun_out/sqx -name "*.csv" -size +100k -delete
done
