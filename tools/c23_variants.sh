#!/bin/bash
# C3 (elastic) and C2 (acoustic) timing of the in-tree library and of every variant under physicsbasedfwi2_amd/_variants
one() {
  timeout -k 10 300 python bench.py --workload $1 --steps 5 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/v.json 2> gpurun_out/v.err || { tail -3 gpurun_out/v.err; return 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/v.json').read().strip().splitlines()[-1])
print('$1', round(d['value']), {k:round(v['us_per_step'],2) for k,v in d['kernels'].items()}, d['check']['verified'], d['check']['bitwise_repeatable'])"
}
echo "== in-tree"; one elastic_marmousi && one acoustic_marmousi || exit 1
for V in physicsbasedfwi2_amd/_variants/*.so; do
  echo "== $V"; export MIFWI_LIB=$GRAFT_REPO_ROOT/$V; one elastic_marmousi && one acoustic_marmousi || exit 1
done
unset MIFWI_LIB
echo "== in-tree again"; one elastic_marmousi
