#!/usr/bin/env python3
"""Where and when the workgroups of dpenv_step run, by batch size (32 768 envs slower than 65 536).

  python3 tools/step_placement.py [--sizes 8192,16384,...] [--workload headline|config2|both] [--steps 250]

Per size and workload: us per step of a HIP graph of `--steps` dependent dpenv_step launches (median / min of 9 replays,
HIP events on the launch stream).  With a trace build of the library (DPENV_LIB=build/wsdiag/trace<B>.so from
`tools/build_kernels_variant.sh trace<B> BLOCK=<B> -DDPENV_STEP_TRACE`) also, from the kernel's own records of the last 8
launches (HW_REG_XCC_ID, HW_REG_HW_ID, s_memrealtime at entry / after the stores were issued, 100 MHz):
  * placement: workgroups per XCD, CUs used, workgroups per CU and per SIMD (max / histogram)
  * timing per launch: first-to-last workgroup START spread, workgroup duration (median / p95 / max), kernel span (first start to
    last end), gap from the last end of launch k to the first start of launch k + 1
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def decode(hw):
    """gfx9 HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]"""
    return {'wave': hw & 15, 'simd': (hw >> 4) & 3, 'pipe': (hw >> 6) & 3, 'cu': (hw >> 8) & 15, 'sh': (hw >> 12) & 1, 'se': (hw >> 13) & 7}


def analyse(tr, ring):
    """tr: [ring][wg][4] uint32"""
    nwg = tr.shape[1]
    xcc = tr[..., 0] & 15
    hw = tr[..., 1]
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    cu_key = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    t0 = tr[..., 2].astype(np.int64)
    t1 = tr[..., 3].astype(np.int64)
    # 32-bit wrap of the 100 MHz counter (43 s period): unwrap relative to the ring's minimum
    base = t0.min()
    t0 = (t0 - base) & 0xffffffff
    t1 = (t1 - base) & 0xffffffff
    order = np.argsort(t0.min(axis=1))                      # launches in time order
    t0, t1, cu_key, simd_key, xcc = t0[order], t1[order], cu_key[order], simd_key[order], xcc[order]
    per = []
    for k in range(ring):
        cus, wg_per_cu = np.unique(cu_key[k], return_counts=True)
        simds, wg_per_simd = np.unique(simd_key[k], return_counts=True)
        dur = (t1[k] - t0[k]) * 10.0                        # ns
        rec = {'workgroups': int(nwg), 'xcds_used': int(len(np.unique(xcc[k]))),
               'wg_per_xcd': np.bincount(xcc[k], minlength=8).tolist(),
               'cus_used': int(len(cus)), 'wg_per_cu_hist': np.bincount(wg_per_cu).tolist(),
               'simds_used': int(len(simds)), 'wg_per_simd_hist': np.bincount(wg_per_simd).tolist(),
               'start_spread_ns': float((t0[k].max() - t0[k].min()) * 10.0),
               'wg_ns_median': float(np.median(dur)), 'wg_ns_p95': float(np.percentile(dur, 95)), 'wg_ns_max': float(dur.max()),
               'span_ns': float((t1[k].max() - t0[k].min()) * 10.0)}
        if k + 1 < ring:
            rec['gap_to_next_ns'] = float((t0[k + 1].min() - t1[k].max()) * 10.0)
            rec['period_ns'] = float((t0[k + 1].min() - t0[k].min()) * 10.0)
            # does workgroup b of launch k+1 land on the XCD / CU that ran workgroup b of launch k (state re-read from that XCD's L2)?
            rec['same_xcd_as_next_frac'] = float((xcc[k] == xcc[k + 1]).mean())
            rec['same_cu_as_next_frac'] = float((cu_key[k] == cu_key[k + 1]).mean())
        # duration by co-residency: workgroups alone on their SIMD / sharing it
        cnt = dict(zip(simds.tolist(), wg_per_simd.tolist()))
        share = np.array([cnt[s] for s in simd_key[k].tolist()])
        rec['wg_ns_median_by_wg_per_simd'] = {int(c): float(np.median(dur[share == c])) for c in np.unique(share)}
        cntc = dict(zip(cus.tolist(), wg_per_cu.tolist()))
        sharec = np.array([cntc[s] for s in cu_key[k].tolist()])
        rec['wg_ns_median_by_wg_per_cu'] = {int(c): float(np.median(dur[sharec == c])) for c in np.unique(sharec)}
        per.append(rec)
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='8192,16384,24576,32768,40960,49152,57344,65536')
    ap.add_argument('--workload', default='both')
    ap.add_argument('--steps', type=int, default=250)
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    import torch
    import ml4ca_amd
    from ml4ca_amd import _lib
    lib = _lib.load()
    trace = hasattr(lib, 'dpenv_debug_set_step_trace')
    dev = torch.device('cuda', 0)
    RING = 8
    block = int(os.environ.get('DPENV_TRACE_BLOCK', '64'))
    out = {'lib': os.environ.get('DPENV_LIB', 'product'), 'trace': trace, 'rows': []}
    print('library: %s   trace records: %s' % (out['lib'], trace))
    for wl in (['headline', 'config2', 'config2_one_wave'] if args.workload == 'both' else args.workload.split(',')):
        for n in [int(x) for x in args.sizes.split(',')]:
            if wl == 'headline':
                env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=False, time_limit=False, seed=1)
            else:
                # config 2 / 4 workload: termination and auto-reset on; `_one_wave`: the re-draw on the env wave (config.step_one_wave)
                env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=True, auto_reset=True, seed=4, step_one_wave=wl.endswith('one_wave'))
            g = torch.Generator(device=dev)
            g.manual_seed(7)
            actions = torch.randn((50, n, 7), generator=g, device=dev) * 0.6065
            obs = torch.empty((n, 9), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
            env.reset()

            def chunk():
                for k in range(args.steps):
                    env.step(actions[k % 50], out=(obs, rew, done))

            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for k in range(20):
                    env.step(actions[k], out=(obs, rew, done))
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                chunk()
            for _ in range(4):
                gr.replay()
            ts = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record()
                torch.cuda.synchronize(dev)
                ts.append(e0.elapsed_time(e1) * 1e3 / args.steps)
            row = {'workload': wl, 'envs': n, 'us_per_step_median': float(np.median(ts)), 'us_per_step_min': float(min(ts))}
            line = '%-16s envs %6d  step %6.3f us (min %6.3f)' % (wl, n, row['us_per_step_median'], row['us_per_step_min'])
            if trace and wl == 'headline':
                nwg = (n + block - 1) // block
                buf = torch.zeros((RING, nwg, 4), dtype=torch.int32, device=dev)
                lib.dpenv_debug_set_step_trace.restype = C.c_int
                lib.dpenv_debug_set_step_trace.argtypes = [C.c_void_p]
                assert lib.dpenv_debug_set_step_trace(buf.data_ptr()) == 0
                gr.replay()
                torch.cuda.synchronize(dev)
                assert lib.dpenv_debug_set_step_trace(None) == 0
                tr = buf.cpu().numpy().view(np.uint32)
                per = analyse(tr, RING)
                row['launches'] = per
                mid = per[RING // 2]
                gaps = [p['gap_to_next_ns'] for p in per if 'gap_to_next_ns' in p]
                periods = [p['period_ns'] for p in per if 'period_ns' in p]
                line += ('  | wg %d  xcds %d  cus %d  wg/cu hist %s  wg/simd hist %s | start spread %4.0f ns  wg median %4.0f p95 %4.0f max %4.0f ns  '
                         'span %4.0f ns  gap %4.0f ns  period %4.0f ns | same xcd next %.2f same cu next %.2f | wg ns by wg/simd %s' % (
                             mid['workgroups'], mid['xcds_used'], mid['cus_used'], mid['wg_per_cu_hist'], mid['wg_per_simd_hist'],
                             np.median([p['start_spread_ns'] for p in per]), np.median([p['wg_ns_median'] for p in per]),
                             np.median([p['wg_ns_p95'] for p in per]), np.median([p['wg_ns_max'] for p in per]),
                             np.median([p['span_ns'] for p in per]), np.median(gaps), np.median(periods),
                             mid.get('same_xcd_as_next_frac', -1), mid.get('same_cu_as_next_frac', -1), mid['wg_ns_median_by_wg_per_simd']))
            print(line, flush=True)
            out['rows'].append(row)
            del gr, env
    if args.json:
        json.dump(out, open(args.json, 'w'))


if __name__ == '__main__':
    main()
