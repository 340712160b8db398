#!/bin/bash
# One GPU session = one gpurun call:   gpurun --timeout 1100 -- 'bash tools/gpu_session.sh TAG STEP [STEP ...]'
# Steps run in order, outputs under gpurun_out/TAG/.  A step that times out or is killed ends the session (no GPU step after a hung one); a
# step that FAILS ends it too (its last lines are shown), unless its name ends in `?`.  This replaces the per-session scripts of round 5
# (tools/r05?_run.sh): every one of them was a list of these steps.
#   tests[=ARGS]     python3 -m pytest tests -q -m gpu -x [ARGS: files, -k ..., --deselect ...]        (limit 1100 s)
#   smoke            __graft_entry__.smoke()
#   bench            the default bench line (+ bench_side.json)            bench_driver   the driver's form: --steps 20 --warmup 5
#   profile          tools/profile_round.sh TAG (kernel trace + stats, PMC passes, summary)             profile_shard  tools/profile_shard.sh TAG
#   soak[=ARGS]      tools/soak_determinism.py ARGS                        soak_fused[=ARGS]  tools/soak_fused.py ARGS
#   box=PRESET       tools/compare_cybersea_box.py PRESET                  sweep          tools/batch_sweep.sh
#   run:NAME:LIMIT:COMMAND...   anything else (quote it)
TAG=${1:?usage: gpu_session.sh TAG STEP...}; shift
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}; O=gpurun_out/$TAG; mkdir -p $O
step() {      # name limit command...   -> $O/name.txt (stdout + stderr)
    local name=$1 lim=$2; shift 2; local soft=0; case $name in *\?) soft=1; name=${name%\?};; esac
    timeout -k 10 $lim "$@" > $O/$name.txt 2>&1; local rc=$?
    echo "[$TAG] $name: exit $rc"; tail -n 3 $O/$name.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$TAG] $name timed out: the session ends here"; exit $rc; fi
    if [ $rc -ne 0 ] && [ $soft -eq 0 ]; then tail -n 30 $O/$name.txt; echo "[$TAG] $name failed: the session ends here"; exit $rc; fi
}
bench_line() {   # name limit args...  -> $O/name.json (the ONE stdout line), $O/name.err, $O/name_side.json
    local name=$1 lim=$2; shift 2
    timeout -k 10 $lim python3 bench.py --side-json $O/${name}_side.json "$@" > $O/$name.json 2> $O/$name.err; local rc=$?
    echo "[$TAG] $name: exit $rc, line $(wc -c < $O/$name.json) B, stderr $(wc -c < $O/$name.err) B"
    if [ $rc -ne 0 ]; then tail -n 20 $O/$name.err; exit $rc; fi
    python3 -c "import json; d=json.loads(open('$O/$name.json').readline()); print('   value %.4e env-steps/s, %.4f us per step, roofline frac %.3f' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['frac']))"
}
for s in "$@"; do
    arg=; case $s in *=*) arg=${s#*=}; s=${s%%=*};; esac
    case $s in
    tests|tests\?) case "$arg" in *tests/*) step $s 1100 python3 -m pytest -q -m gpu -x $arg;; *) step $s 1100 python3 -m pytest tests -q -m gpu -x $arg;; esac;;
    smoke) step smoke 300 python3 -c "import __graft_entry__ as g; g.smoke()";;
    bench) bench_line bench_default 500;;
    bench_driver) bench_line bench_driver_form 300 --steps 20 --warmup 5;;
    profile) step profile 1000 bash tools/profile_round.sh $TAG;;
    profile_shard) step profile_shard 1000 bash tools/profile_shard.sh $TAG;;
    soak|soak\?) step $s 900 python3 tools/soak_determinism.py $arg;;
    soak_fused|soak_fused\?) step $s 700 python3 tools/soak_fused.py $arg;;
    box) step box_$arg 200 python3 tools/compare_cybersea_box.py $arg;;
    sweep) step sweep 900 bash tools/batch_sweep.sh;;
    run:*) IFS=: read -r _ name lim cmd <<< "$s${arg:+=$arg}"; step $name $lim bash -c "$cmd";;
    *) echo "unknown step $s"; exit 2;;
    esac
done
