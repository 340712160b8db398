#!/bin/bash
# Round profile: kernel trace + stats, HBM traffic PMC passes, SQ counters.  Run via gpurun (tools/gpu_session.sh TAG profile); outputs under
# gpurun_out/prof_$1/summary (copy what is to be judged into profiles/).  One program directly after `--` in every rocprofv3 call, PMC passes
# separate from the kernel trace, every pass under its own time limit and CHAINED: a pass that fails or hangs ends the round - nothing runs on the
# GPU after it, and nothing is summarised from partial data.
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; mkdir -p $O
B="python3 bench.py --side-json $O/side.json"
pass() {      # name limit command...
    local name=$1 lim=$2; shift 2
    timeout -k 10 $lim "$@" > $O/bench_$name.json 2> $O/$name.err; local rc=$?
    echo "pass $name: exit $rc"
    if [ $rc -ne 0 ]; then tail -n 5 $O/$name.err; echo "profile round $TAG ends at pass $name"; exit $rc; fi
}
# (--eager-loop 0 in every traced pass: the eager-loop record launches the headline kernel ~14 000 times EAGERLY - dispatch duration 6.0 us under the
# tracer - which would turn the per-kernel average of the default, graph-replayed command into an average over two launch forms; --multi-handle 0
# (round 6): the independent-chains record launches the headline instantiation at 16 384 ... 131 072 envs)
pass kt 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --no-cpu-baseline --eager-loop 0 --multi-handle 0
# the headline kernel's own duration, unstretched: (i) eager launches (the host spaces them: every dispatch's timestamps are its own), (ii) a
# 50-step graph replayed; the plain run above traces every node of the one long graph and is kept for the other kernels' durations
pass kt_eager 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_eager -- $B --no-graph --steps 250 --warmup 50 --no-cpu-baseline --no-fused
pass kt_g50 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_g50 -- $B --graph-steps 50 --steps 50 --warmup 50 --no-cpu-baseline --no-fused
P="--steps 250 --warmup 50 --no-cpu-baseline --eager-loop 0 --multi-handle 0"
pass pf 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B $P
pass pw 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B $P
pass sq 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- $B $P
pass mf 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_mfma -- $B $P
pass l2 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- $B $P
pass plain 400 $B
cp $O/side.json $O/bench_plain_side.json
# condense on the box (the raw traces exceed what gpurun copies back), keep the summaries only
python3 tools/summarize_profile.py $TAG $O/summary && rm -rf $O/kt $O/kt_eager $O/kt_g50 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma $O/pmc_l2
ls $O $O/summary
