#!/bin/bash
# A/B of the HEADLINE leg only (dpenv_step at 65 536 envs, graph replay) across library builds, interleaved: tools/ab_headline.sh REPS a.so b.so ...
# (run on the GPU box; DEV_FAST builds are enough - they hold the shipped instantiation).  Prints us per env step (wall) and the HIP-event kernel time.
REPS=$1; shift
mkdir -p gpurun_out
for rep in $(seq $REPS); do
  for v in "$@"; do
    DPENV_LIB=$PWD/$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --side-legs 0 --eager-loop 0 --steps 5000 --warmup 500 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-32s' % '$v', 'step %.4f us' % (d['ms_per_step']*1e3), 'kernel (events) %.4f us' % d['roofline']['avg_launch_us'], 'value %.4e' % d['value'])" || exit 1
  done
done
