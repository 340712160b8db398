#!/bin/bash
# Same-call A/B of library builds on the closed loop: tools/ab_libs.sh "<time_closed_loop args>" lib1.so lib2.so ...   (two interleaved passes;
# rows digests by tools/ab_bits.py first: identical digests = bit-identical rows)
args=$1; shift
for lib in "$@"; do echo "== bits $lib"; DPENV_LIB=$PWD/$lib python3 tools/ab_bits.py 2>/dev/null; done
for pass in 1 2; do
  for lib in "$@"; do echo "== time pass $pass $lib"; DPENV_LIB=$PWD/$lib python3 tools/time_closed_loop.py $args 2>/dev/null; done
done
