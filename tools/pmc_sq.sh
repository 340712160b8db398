#!/bin/bash
# SQ counter passes for the step and rollout kernels (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc/p1 -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/pmc/p1.json 2> gpurun_out/pmc/p1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc/p2 -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/pmc/p2.json 2> gpurun_out/pmc/p2.err
python3 - <<'PY'
import csv, glob, collections, statistics as st
for p in ('p1','p2'):
    f=glob.glob('gpurun_out/pmc/%s/*/*_counter_collection.csv'%p)
    if not f: print(p,'no csv'); continue
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.defaultdict(list)
    for r in rows:
        k=r['Kernel_Name']
        kn='step' if 'step_kernel' in k else 'rollout' if 'rollout_kernel' in k else None
        if kn: agg[(kn,r['Counter_Name'])].append(float(r['Counter_Value']))
    for k in sorted(agg): print(p, k, 'median %.0f'%st.median(agg[k]), 'n', len(agg[k]))
PY
