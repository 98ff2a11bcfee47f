#!/bin/bash
# The CPU suite (-m "not gpu") against AddressSanitizer + UBSan builds of the host library and of
# the oracle (sanitizers for the host side; GPU sanitizers are not available
# on this pool).  python itself is not instrumented: the sanitizer runtime is preloaded, leak
# detection is off (the interpreter never frees everything), everything else aborts the run.
set -e
cd "$(dirname "$0")/.."
python pg_strom_amd/build.py --sanitized
make -s -C oracle liboracle_asan.so
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
export STROM_HIP_LIBRARY=$PWD/pg_strom_amd/libstrom_hip_asan.so
export STROM_ORACLE_LIBRARY=$PWD/oracle/liboracle_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD="$ASAN_LIB $UBSAN_LIB" python -m pytest tests -q -x -m "not gpu" -p no:cacheprovider "$@"
