#!/bin/bash
# Round profile: kernel trace + stats, HBM traffic PMC passes, SQ counters. Run via gpurun; outputs under gpurun_out/prof_$1
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --eager-loop 0 > $O/bench_kt.json 2> $O/kt.err
# (--eager-loop 0 in every traced pass: the eager-loop record launches the headline kernel ~14 000 times EAGERLY - dispatch duration 6.0 us under the
# tracer - which would turn the per-kernel average of the default, graph-replayed command into an average over two launch forms; pass (i) below is
# the eager form on its own)
# the headline kernel's own duration, unstretched: (i) eager launches (the host spaces them: every dispatch's timestamps are its own), (ii) a
# 50-step graph replayed; (iii) the plain run above traces every node of the one long graph and is kept for the other kernels' durations
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_eager -- python3 bench.py --no-graph --steps 250 --warmup 50 --no-cpu-baseline --no-fused > $O/bench_kt_eager.json 2> $O/kt_eager.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_g50 -- python3 bench.py --graph-steps 50 --steps 50 --warmup 50 --no-cpu-baseline --no-fused > $O/bench_kt_g50.json 2> $O/kt_g50.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 250 --warmup 50 --no-cpu-baseline --eager-loop 0 > $O/bench_pf.json 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 250 --warmup 50 --no-cpu-baseline --eager-loop 0 > $O/bench_pw.json 2> $O/pw.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 250 --warmup 50 --no-cpu-baseline --eager-loop 0 > $O/bench_sq.json 2> $O/sq.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 250 --warmup 50 --no-cpu-baseline --eager-loop 0 > $O/bench_mf.json 2> $O/mf.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 bench.py --steps 250 --warmup 50 --no-cpu-baseline --eager-loop 0 > $O/bench_l2.json 2> $O/l2.err
python3 bench.py > $O/bench_plain.json 2> $O/plain.err
# condense on the box (the raw traces exceed what gpurun copies back), keep the summaries only
python3 tools/summarize_profile.py $TAG $O/summary && rm -rf $O/kt $O/kt_eager $O/kt_g50 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma $O/pmc_l2
ls $O $O/summary
