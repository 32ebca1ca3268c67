#!/bin/bash
# usage: [ENV=..] tools/pmc_one.sh NAME OUT.json bench-args...   - FETCH_SIZE / WRITE_SIZE passes of ONE bench workload
# with the environment as it is (A/B of a kernel variant's HBM traffic); prints the entry of tools/pmc_traffic.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; name=$1; out=$2; shift 2
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${name}_$C
    timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${name}_$C -- python bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-verify > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
done
python tools/pmc_traffic.py $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $out "$@" | grep -E "bytes_per_cell_step|kernels|\"el_|\"ac_|adjoint|forward"
find $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE -name "*.csv" -size +200k -delete
