#!/bin/bash
# CPU-only: the oracle (and the tests that drive it) under AddressSanitizer + UBSan.  GPU sanitizers are not available on the
# pool; the C restatement is the part of the test infrastructure that indexes raw buffers, so it is the part worth checking.
# Builds the instrumented library to /tmp and points the ctypes front-end at it through DPENV_ORACLE_SO.
set -e
cd "$(dirname "$0")/.."
SO=/tmp/libdpenv_oracle_asan.so
gcc -O1 -g -std=c11 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -fopenmp -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared -o $SO oracle/dpenv_oracle.c -lm
DPENV_ORACLE_SO=$SO ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests -x -q -m "not gpu" "$@"
