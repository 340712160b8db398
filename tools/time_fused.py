#!/usr/bin/env python3
"""dpenv_rollout: the two-wave kernel (an env wave + a row wave per 64 envs, shipped) against the one-wave kernel (config.step_one_wave),
same call: us per env step of 50-step launches, headline workload (no resets) and config-2 workload (termination + auto-reset).
    python3 tools/time_fused.py [--envs 65536,32768,16384]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', default='65536,32768,16384')
    args = ap.parse_args()
    import numpy as np
    import torch
    import ml4ca_amd
    dev = torch.device('cuda', 0)
    T = 50
    for n in [int(x) for x in args.envs.split(',')]:
        g = torch.Generator(device=dev).manual_seed(3)
        actions = torch.randn((T, n, 7), generator=g, device=dev) * 0.6065
        for wl, kw in (('headline', dict(terminate=False, time_limit=False)), ('config2', dict(terminate=True, auto_reset=True))):
            res = {}
            for rep in range(2):
                for one_wave in (False, True):
                    env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, seed=1, step_one_wave=one_wave, **kw)
                    env.reset()
                    out = env.rollout(actions)
                    for _ in range(10):
                        env.rollout(actions, out=out)
                    ts = []
                    for _ in range(15):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(4):
                            env.rollout(actions, out=out)
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3 / (4 * T))
                    res.setdefault(one_wave, []).append(float(np.median(ts)))
                    del env
            print('%-9s envs %6d   two waves %s us per step   one wave %s' % (wl, n, ' / '.join('%.3f' % x for x in res[False]),
                                                                               ' / '.join('%.3f' % x for x in res[True])), flush=True)


if __name__ == '__main__':
    main()
