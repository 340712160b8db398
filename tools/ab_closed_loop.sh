#!/bin/bash
# A/B of every closed-loop launch form across library builds, interleaved in one call: tools/ab_closed_loop.sh REPS a.so b.so ...
# (tools/time_closed_loop.py per build and repetition; DEV_FAST builds hold the shipped env / network shape)
REPS=$1; shift
mkdir -p gpurun_out/ab_closed_loop
for rep in $(seq $REPS); do
  for v in "$@"; do
    tag=$(basename $v .so)
    DPENV_LIB=$PWD/$v timeout -k 10 300 python3 tools/time_closed_loop.py --out gpurun_out/ab_closed_loop/${tag}_$rep.json > /dev/null 2>&1 || exit 1
  done
done
python3 - "$@" <<'PY'
import glob, json, os, sys
for v in sys.argv[1:]:
    tag = os.path.basename(v)[:-3]
    runs = [json.load(open(f)) for f in sorted(glob.glob('gpurun_out/ab_closed_loop/%s_*.json' % tag))]
    print(tag)
    for k in runs[0]:
        if isinstance(runs[0][k], dict):
            vals = [r[k]['us_per_step_median'] for r in runs]
            print('   %-28s median us per env step: %s   mean %.3f' % (k, ' '.join('%.3f' % x for x in vals), sum(vals) / len(vals)))
PY
