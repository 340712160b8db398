#!/bin/bash
# Round profile: kernel trace + stats, HBM traffic PMC passes, SQ counters. Run via gpurun; outputs under gpurun_out/prof_$1
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline > $O/bench_kt.json 2> $O/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_pf.json 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_pw.json 2> $O/pw.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_sq.json 2> $O/sq.err
python3 bench.py > $O/bench_plain.json 2> $O/plain.err
ls $O
