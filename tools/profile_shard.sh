#!/bin/bash
# Shard-size profile round: the kernels that are the default at config 4's shard size under the config-2 workload.
#   gpurun -- 'bash tools/profile_shard.sh r05a'       outputs: gpurun_out/prof_<tag>/summary/ (copy to profiles/)
# One program directly after `--` in every rocprofv3 call; PMC passes separate from the kernel trace; steps joined with && so that
# nothing runs on the GPU after a failed step.
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; mkdir -p $O
P="python3 tools/profile_shard.py"
$P > $O/plain.json 2> $O/plain.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $P > $O/kt.json 2> $O/kt.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P --reps 1 > $O/pf.json 2> $O/pf.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P --reps 1 > $O/pw.json 2> $O/pw.err &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- $P --reps 1 > $O/sq.json 2> $O/sq.err &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_mfma -- $P --reps 1 > $O/mf.json 2> $O/mf.err &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- $P --reps 1 > $O/l2.json 2> $O/l2.err &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $O/pmc_flat -- $P --reps 1 > $O/fl.json 2> $O/fl.err
echo "last pass exit: $?"
python3 tools/summarize_shard_profile.py $TAG $O/summary && rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma $O/pmc_l2 $O/pmc_flat
ls $O $O/summary; for f in $O/*.err; do tail -n 2 $f; done
