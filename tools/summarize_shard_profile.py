#!/usr/bin/env python3
"""Condense a tools/profile_shard.sh output directory (gpurun_out/prof_<tag>) into <dst>/<tag>_summary.json + <tag>_kernel_stats.csv:
per kernel INSTANTIATION (template arguments kept: the reset-wave step kernel, the critic-wave closed loop and the 128- / 256-env
geometries are different kernels) - dispatch duration, registers / LDS / scratch from the kernel trace, and the PMC counters per
wave per env-step and per launch, HBM bytes with the gfx950 FETCH_SIZE correction (x 2: MI355X_MICROARCH.md, HBM / rocprofv3)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import statistics as st
import sys

tag = sys.argv[1]
src = 'gpurun_out/prof_%s' % tag
dst = sys.argv[2] if len(sys.argv) > 2 else 'profiles'
os.makedirs(dst, exist_ok=True)
plain = json.loads(open(src + '/plain.json').read().strip().splitlines()[-1])
T, CH, N, BIG = plain['T'], plain['chunk'], plain['envs'], None
for k, v in plain['legs'].items():
    if k.startswith('closed_big'):
        BIG = v['envs']
PREC = {'0': 'f16', '1': 'f32', '2': 'f32_actor'}


def kname(k):
    """(short name, env steps per launch, envs) of a kernel this workload launches, None for anything else (torch fills, copies)"""
    m = re.search(r'policy_rollout_ws_kernel<([^>]*)>', k)
    if m:
        a = [re.sub(r'[^0-9a-z]', '', x) for x in m.group(1).split(',')]     # MODE, EXT, KA, ROLES, PREC, GROUPS
        grp = int(a[5]) if len(a) > 5 else 4
        rnd = len(a) > 6 and a[6] in ('true', '1')
        name = 'policy_rollout_ws_kernel<roles=%s,%s,%d_envs_per_workgroup%s>' % (a[3], PREC.get(a[4], a[4]), 64 * grp, ',randomised' if rnd else '')
        return name, T, (N if grp == 2 else (BIG or N))
    m = re.search(r'dpenv::step_kernel<([^>]*)>', k)
    if m:
        a = [x.strip() for x in m.group(1).split(',')]
        ves = {'0': 'shared', '1': 'class_lds', '2': 'per_env_registers', '3': 'per_env_lds_image', '4': 'per_env_randomised'}.get(a[2], a[2]) if len(a) >= 4 else ''
        return 'step_kernel<%s>%s%s' % (','.join(a), ' ' + ves if ves else '', ' (reset wave)' if a[-1] == 'true' and len(a) >= 4 else ''), 1, N
    if 'dpenv::rollout_ws_kernel' in k:
        return 'rollout_ws_kernel', CH, N
    if 'dpenv::rollout_kernel' in k:
        return 'rollout_kernel', CH, N
    for n_ in ('reset_kernel', 'pack_policy_kernel'):
        if n_ in k:
            return n_, 1, N
    return None


out = {'tag': tag, 'plain_run': plain,
       'commands': {'kernel_trace': 'rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/profile_shard.py',
                    'pmc': 'rocprofv3 --pmc <one counter set per pass> --output-format csv -- python3 tools/profile_shard.py --reps 1'},
       'correction': 'gfx950: FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> x 2 (MI355X_MICROARCH.md); WRITE_SIZE exact; both KiB',
       'kernels': {}}
ks = glob.glob(src + '/kt/*/*_kernel_stats.csv')
if ks:
    shutil.copy(ks[0], '%s/%s_kernel_stats.csv' % (dst, tag))
rows = list(csv.DictReader(open(glob.glob(src + '/kt/*/*_kernel_trace.csv')[0])))
by = collections.defaultdict(list)
for r in rows:
    kn = kname(r['Kernel_Name'])
    if kn:
        by[kn].append(r)
for (name, spl, envs), rs in sorted(by.items()):
    d = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    r0 = rs[0]
    k = {'full_name': r0['Kernel_Name'], 'env_steps_per_launch': spl, 'envs': envs, 'dispatches': len(d), 'avg_ns': st.mean(d), 'median_ns': st.median(d),
         'min_ns': d[0], 'max_ns': d[-1], 'median_us_per_env_step': st.median(d) * 1e-3 / spl}
    for col, key in (('VGPR_Count', 'vgpr'), ('Accum_VGPR_Count', 'agpr'), ('SGPR_Count', 'sgpr'), ('LDS_Block_Size', 'lds_bytes'),
                     ('Scratch_Size', 'scratch_bytes_per_lane'), ('Private_Segment_Size', 'scratch_bytes_per_lane'),
                     ('Workgroup_Size_X', 'workgroup'), ('Grid_Size_X', 'grid')):
        if col in r0 and key not in k:
            k[key] = r0[col]
    out['kernels'][name] = k
for sub in sorted(glob.glob(src + '/pmc_*')):
    f = glob.glob(sub + '/*/*_counter_collection.csv')
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        kn = kname(r['Kernel_Name'])
        if kn:
            agg[(kn[0], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (name, c), v in agg.items():
        if name in out['kernels']:
            out['kernels'][name].setdefault('pmc_median_per_launch', {})[c] = st.median(v)
for name, k in out['kernels'].items():
    p = k.get('pmc_median_per_launch', {})
    spl, envs = k['env_steps_per_launch'], k['envs']
    if 'FETCH_SIZE' in p and 'WRITE_SIZE' in p:
        k['hbm_bytes_per_launch'] = (2 * p['FETCH_SIZE'] + p['WRITE_SIZE']) * 1024
        k['hbm_bytes_per_env_step'] = k['hbm_bytes_per_launch'] / envs / spl
    if p.get('TCC_HIT_sum', 0) + p.get('TCC_MISS_sum', 0) > 0:
        k['l2_hit_rate'] = p['TCC_HIT_sum'] / (p['TCC_HIT_sum'] + p['TCC_MISS_sum'])
    if p.get('SQ_WAVES'):
        w = p['SQ_WAVES']
        k['per_wave_per_env_step'] = {c: p[c] / w / spl for c in p if c.startswith('SQ_') and c != 'SQ_WAVES'}
        if p.get('SQ_WAVE_CYCLES'):
            k['share_of_wave_cycles'] = {c: p[c] / p['SQ_WAVE_CYCLES'] for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU',
                                                                                  'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES') if c in p}
json.dump(out, open('%s/%s_summary.json' % (dst, tag), 'w'), indent=1)
for name, k in out['kernels'].items():
    print('%-72s %6d x  median %10.1f us  (%7.3f us/env-step)  vgpr %s agpr %s lds %s scratch %s' % (
        name, k['dispatches'], k['median_ns'] * 1e-3, k['median_us_per_env_step'], k.get('vgpr'), k.get('agpr'), k.get('lds_bytes'), k.get('scratch_bytes_per_lane')))
