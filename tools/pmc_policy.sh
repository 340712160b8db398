#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcp/p1 -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/pmcp/p1.json 2> gpurun_out/pmcp/p1.err
rocprofv3 --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmcp/p2 -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/pmcp/p2.json 2> gpurun_out/pmcp/p2.err
tail -2 gpurun_out/pmcp/p2.err
python3 - <<'PY'
import csv, glob, collections, statistics as st
for p in ('p1','p2'):
    f=glob.glob('gpurun_out/pmcp/%s/*/*_counter_collection.csv'%p)
    if not f: print(p,'no csv'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'policy_rollout' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    w=st.median(agg.get('SQ_WAVES',[1024]))
    for k in sorted(agg): print(p, k, 'per wave per step %.1f'%(st.median(agg[k])/w/50), 'n', len(agg[k]))
PY
