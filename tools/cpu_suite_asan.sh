#!/bin/bash
# The CPU test suite against the oracle built with -fsanitize=address,undefined ("race
# detection / sanitizers").  CPU container only - never on the GPU box (gpurun refuses sanitizer runs there).
#   usage: tools/cpu_suite_asan.sh [pytest args]      default: every non-GPU test that drives the C oracle
set -e
cd "$(dirname "$0")/.."
make -C oracle -B asan > /dev/null
export ORACLE_ASAN=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
# python itself leaks by design; the interesting reports are overflows / UB inside liboracle
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
if [ $# -eq 0 ]; then
    set -- tests/test_oracle_properties.py tests/test_misfit_oracle.py tests/test_reference_pins.py -k "not time_convergence"
fi
exec python -m pytest -x -q -m "not gpu" -p no:cacheprovider "$@"
