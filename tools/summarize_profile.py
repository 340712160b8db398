#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into profiles/<tag>_*.{csv,json}."""
import collections
import csv
import glob
import json
import shutil
import statistics as st
import sys

import os

tag = sys.argv[1]
src = 'gpurun_out/prof_%s' % tag
dst = sys.argv[2] if len(sys.argv) > 2 else 'profiles'       # on the GPU box: a directory under gpurun_out/, copied to profiles/ afterwards
os.makedirs(dst, exist_ok=True)
out = {'tag': tag, 'n_envs': 65536,
       'commands': {'kernel_trace': 'rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --eager-loop 0 --multi-handle 0 (the eager-loop record launches the headline kernel eagerly ~14 000 times: its dispatches are profiled on their own, headline_kernel_duration.eager)',
                    'pmc': 'rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_*> --output-format csv -- python3 bench.py --steps 250 --warmup 50 --no-cpu-baseline --eager-loop 0 (one pass per counter set)'},
       'correction': 'gfx950: FETCH_SIZE tallies 128-B requests of wide coalesced reads at 64 B -> x2 (MI355X_MICROARCH.md, HBM); '
                     'WRITE_SIZE exact; both in KiB', 'kernels': {}}
shutil.copy(glob.glob(src + '/kt/*/*_kernel_stats.csv')[0], '%s/%s_kernel_stats.csv' % (dst, tag))
rows = list(csv.DictReader(open(glob.glob(src + '/kt/*/*_kernel_trace.csv')[0])))


import re

PREC = {'0': 'f16', '1': 'f32', '2': 'f32_actor'}


def kname(k):
    m = re.search(r'policy_rollout_ws_kernel<([^>]*)>', k)
    if m:
        # <MODE, EXT, KA, ROLES, PREC, GROUPS>: one entry per arithmetic and workgroup geometry (round 3)
        a = [x.strip() for x in m.group(1).split(',')]
        prec = PREC.get(re.sub(r'[^0-9]', '', a[4]) if len(a) > 4 else '0', '?')
        grp = re.sub(r'[^0-9]', '', a[5]) if len(a) > 5 else '4'
        rnd = len(a) > 6 and a[6] in ('true', '1')          # round 5: the instantiation with the randomisation's hull re-draw (a side record launches it)
        return 'policy_rollout_ws_kernel/%s/%d_envs_per_workgroup%s' % (prec, 64 * int(grp or 4), '/randomised' if rnd else '')
    for n in ('policy_rollout_x_kernel', 'policy_rollout_kernel', 'gae_kernel', 'gae_finalize_kernel',
              'adv_apply_kernel', 'pack_policy_kernel'):
        if n in k:
            return n
    m = re.search(r'dpenv::step_kernel<([^>]*)>', k)
    if m:
        # <MODE, EXT, VES, RESETW>: the headline is the shared-hull kernel without a reset wave; the bench's side records (round 5) also launch
        # the class / per-env / randomised instantiations and the reset-wave forms - different kernels, kept apart
        a = [x.strip() for x in m.group(1).split(',')]
        if len(a) >= 4 and (a[2], a[3]) != ('0', 'false'):
            ves = {'0': 'shared', '1': 'class_lds', '2': 'per_env_registers', '3': 'per_env_lds_image', '4': 'per_env_randomised'}.get(a[2], a[2])
            return 'step_kernel/%s%s' % (ves, '/reset_wave' if a[3] == 'true' else '')
        return 'step_kernel'
    return 'rollout_ws_kernel' if 'dpenv::rollout_ws_kernel' in k else 'rollout_kernel' if 'rollout_kernel' in k else None


ALL = sorted({kname(r['Kernel_Name']) for r in rows} - {None})
for kn in ALL:
    ks = [r for r in rows if kname(r['Kernel_Name']) == kn]
    if not ks:
        continue
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks]
    out['kernels'][kn] = {'full_name': ks[0]['Kernel_Name'], 'dispatches': len(d), 'avg_ns': st.mean(d), 'median_ns': st.median(d),
                          'min_ns': min(d), 'max_ns': max(d), 'vgpr': ks[0]['VGPR_Count'], 'sgpr': ks[0]['SGPR_Count'],
                          'lds_bytes': ks[0]['LDS_Block_Size'], 'workgroup': ks[0]['Workgroup_Size_X'], 'grid': ks[0]['Grid_Size_X']}
for name in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_mfma', 'pmc_l2'):
    f = glob.glob(src + '/%s/*/*_counter_collection.csv' % name)
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        kn = kname(r['Kernel_Name'])
        if kn:
            agg[(kn, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (kn, c), v in agg.items():
        out['kernels'].setdefault(kn, {}).setdefault('pmc_median_per_launch', {})[c] = st.median(v)
steps_per_launch = collections.defaultdict(lambda: 1, {'step_kernel': 1, 'rollout_kernel': 50, 'rollout_ws_kernel': 50, 'policy_rollout_kernel': 50, 'policy_rollout_x_kernel': 50,
                    'gae_kernel': 400, 'gae_finalize_kernel': 1, 'adv_apply_kernel': 400, 'pack_policy_kernel': 1})
for kn, k in out['kernels'].items():
    if kn.startswith('policy_rollout_ws_kernel'):
        # the bench launches these with T = 50 (closed-loop legs) and T = 400 (config 5): per-step figures use the launch's own T,
        # recovered from the duration ratio is fragile - the PMC passes run `--steps 250` where only the T = 50 and T = 400 launches
        # exist; medians are dominated by the more frequent T = 50 launches
        steps_per_launch[kn] = 50
    p = k.get('pmc_median_per_launch', {})
    if 'FETCH_SIZE' in p and 'WRITE_SIZE' in p:
        k['hbm_bytes_per_launch'] = (2 * p['FETCH_SIZE'] + p['WRITE_SIZE']) * 1024
        k['hbm_bytes_per_env_step'] = k['hbm_bytes_per_launch'] / 65536 / steps_per_launch[kn]
    if 'TCC_HIT_sum' in p and 'TCC_MISS_sum' in p and p['TCC_HIT_sum'] + p['TCC_MISS_sum'] > 0:
        k['l2_hit_rate'] = p['TCC_HIT_sum'] / (p['TCC_HIT_sum'] + p['TCC_MISS_sum'])
    if 'SQ_WAVES' in p:
        w = p['SQ_WAVES']
        k['per_wave_per_env_step'] = {c: p[c] / w / steps_per_launch[kn] for c in p if c.startswith('SQ_') and c != 'SQ_WAVES'}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in p and 'SQ_BUSY_CYCLES' in p and p['SQ_BUSY_CYCLES'] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs that issue MFMAs; SQ_BUSY_CYCLES is per SE (32 on this part):
        # matrix-pipe share of the kernel = MFMA cycles / (4 SIMDs x 256 CUs x kernel cycles)
        k['mfma_cycles_per_wave_per_env_step'] = p['SQ_VALU_MFMA_BUSY_CYCLES'] / p.get('SQ_WAVES', 1) / steps_per_launch[kn]
    if kn in ('step_kernel', 'rollout_kernel', 'rollout_ws_kernel') or kn.startswith('step_kernel/'):
        k['algorithmic_bytes_per_launch'] = 177 * 65536 * steps_per_launch[kn]
try:
    out['bench_line'] = json.loads(open(src + '/bench_plain.json').read())
except Exception as e:
    out['bench_line'] = str(e)
try:
    out['bench_side_records'] = json.loads(open(src + '/bench_plain_side.json').read())      # round 6: the side records are a file of their own
except Exception:
    pass
# the headline kernel's duration by launch form (round 4): what the tracer does to a dependent chain depends on how it is launched
hd = {'what': 'step_kernel duration (End - Start timestamp of rocprofv3 --kernel-trace) by launch form, beside the HIP-event SPACING of back-to-back '
              'dependent launches that bench.py reports, traced and untraced.  Three different clocks (profiles/LAB_NOTES.md, round 4): (1) the waves\' '
              'own span, first wave in to last store issued, read with s_memrealtime inside a -DDPENV_STEP_TRACE build: 3.67 us at 65 536 envs '
              '(profiles/r04_step_placement.txt); (2) the command processor\'s dispatch-to-completion interval that rocprofv3 reports: it adds the launch '
              'ramp and the end-of-kernel cache write-back / completion signal, 5.3-6.1 us in EVERY launch form, eager included, and the tracer\'s own '
              'per-dispatch signals space the launches out (bench spacing in the traced runs: 6.0-15 us); (3) the untraced spacing of back-to-back '
              'launches, 5.05 us: shorter than (2) because, untraced, the processing of dispatch k + 1 overlaps the tail of kernel k.  roofline.frac of '
              'the bench line uses (3), the only one of the three that is throughput; pricing the kernel by (2) (median of the 50-step-graph pass) '
              'gives a fraction 5-6 % lower, by (1) 38 % higher.'}
for form, sub, cmd in (('eager', 'kt_eager', 'bench.py --no-graph --steps 250 --warmup 50 --no-cpu-baseline --no-fused'),
                       ('graph_50_steps', 'kt_g50', 'bench.py --graph-steps 50 --steps 50 --warmup 50 --no-cpu-baseline --no-fused'),
                       ('one_long_graph', 'kt', 'bench.py --no-cpu-baseline --eager-loop 0 --multi-handle 0')):
    f = glob.glob(src + '/%s/*/*_kernel_trace.csv' % sub)
    if not f:
        continue
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f[0])) if kname(r['Kernel_Name']) == 'step_kernel']
    if not d:
        continue
    rec = {'command': 'rocprofv3 --kernel-trace --stats --output-format csv -- python3 ' + cmd, 'dispatches': len(d), 'avg_ns': st.mean(d),
           'median_ns': st.median(d), 'min_ns': min(d), 'p95_ns': sorted(d)[int(0.95 * len(d))], 'max_ns': max(d)}
    try:
        b = json.loads(open(src + '/bench_%s.json' % sub).read())
        rec['bench_spacing_us_in_that_traced_run'] = b['roofline']['avg_launch_us']
    except Exception:
        pass
    hd[form] = rec
try:
    hd['untraced_spacing_us'] = out['bench_line']['roofline']['avg_launch_us']
    if 'eager' in hd:
        e = hd['eager']['avg_ns'] * 1e-3
        hd['frac_by_untraced_spacing'] = out['bench_line']['roofline']['frac']
        for form in ('eager', 'graph_50_steps', 'one_long_graph'):
            if form in hd:
                hd['frac_by_rocprof_duration_' + form] = {'avg': 177 * 65536 / (hd[form]['avg_ns'] * 1e-9) / 8e12, 'median': 177 * 65536 / (hd[form]['median_ns'] * 1e-9) / 8e12}
except Exception:
    pass
out['headline_kernel_duration'] = hd
json.dump(out, open('%s/%s_summary.json' % (dst, tag), 'w'), indent=1)
if 'step_kernel' in out['kernels'] and 'hbm_bytes_per_launch' in out['kernels']['step_kernel']:
    tl = {'n_envs': 65536, 'hbm_bytes_per_launch': out['kernels']['step_kernel']['hbm_bytes_per_launch'],
          'source': 'profiles/%s_summary.json' % tag, 'tag': tag}
    if 'graph_50_steps' in hd:
        f_ = hd['graph_50_steps']
        tl['step_kernel_duration_ns'] = {'avg': f_['avg_ns'], 'median': f_['median_ns'], 'min': f_['min_ns'],
                                         'source': 'profiles/%s_summary.json headline_kernel_duration.graph_50_steps (rocprofv3 --kernel-trace, dispatch-to-completion '
                                                   'interval of %d dispatches in 50-step graphs: launch ramp + waves + end-of-kernel write-back)' % (tag, f_['dispatches'])}
    json.dump(tl, open('%s/traffic_latest.json' % dst, 'w'))
print(json.dumps({k: v for k, v in out['kernels'].items() if 'policy' in k or 'gae' in k}, indent=1)[:6000])
