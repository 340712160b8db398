#!/bin/bash
# round-5 GPU session s: HEAD at the end of round 5 - whole GPU suite, smoke(), the default bench line and
# the driver's form of it.  A step that times out ends the session (no GPU step after a hung one).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "$name exit $rc" >&2; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: session ends here" >&2; exit $rc; fi; return $rc; }
step tests 1100 python3 -m pytest tests -q -m gpu -x > $O/gputests.txt 2>&1 || { tail -n 30 $O/gputests.txt; exit 1; }
tail -n 2 $O/gputests.txt
step smoke 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
step bench 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
step bench_driver 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python3 -c "
import json
for f in ('bench_default','bench_driver_form'):
    d=json.loads(open('$O/'+f+'.json').readline()); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
    pe=d.get('vessel_classes',{}).get('per_env',{})
    if pe: print('  per-env', pe['registers']['step_us'], pe['registers']['roofline']['frac'], pe['lds_image']['step_us'], {k:(v['step_us'], v.get('closed_loop_us_per_step_f16')) for k,v in pe['config2_workload'].items() if isinstance(v,dict)})
"
