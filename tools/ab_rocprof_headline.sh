#!/bin/bash
# The headline kernel's rocprofv3 duration (dispatch to completion) for several library builds ON ONE BOX: tools/ab_rocprof_headline.sh a.so b.so ...
# (50-step graphs, kernel trace; the durations of different boxes differ by more than most code changes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_rocprof; mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
  tag=$(basename $v .so)_$rep
  export DPENV_LIB=$GRAFT_REPO_ROOT/$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 bench.py --graph-steps 50 --steps 50 --warmup 50 --no-cpu-baseline --no-fused --side-legs 0 --eager-loop 0 > $O/$tag.json 2> $O/$tag.err
  python3 - $O/$tag $tag <<'PY'
import csv, glob, sys, statistics, json
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f[0])) if 'step_kernel<' in r['Kernel_Name']]
b = json.loads(open(sys.argv[1] + '.json').readline())
print('%-14s %5d dispatches: avg %.0f median %.0f min %d p95 %.0f ns; bench spacing in this traced run %.2f us' % (sys.argv[2], len(d), statistics.mean(d), statistics.median(d), min(d), sorted(d)[int(0.95 * len(d))], b['ms_per_step'] * 1e3))
PY
  rm -rf $O/$tag
done
done
