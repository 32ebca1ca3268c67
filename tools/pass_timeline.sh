#!/bin/bash
# kernel timeline of one gradient pass of a bench workload (default: the elastic headline): tools/pass_timeline.sh [workload] [physics]
W=${1:-elastic_marmousi}; P=${2:-elastic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --workload $W --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-verify > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
python tools/pass_timeline.py gpurun_out/tl $P > gpurun_out/pass_timeline_$W.txt; cat gpurun_out/pass_timeline_$W.txt
find gpurun_out/tl -name "*.csv" -size +500k -delete
