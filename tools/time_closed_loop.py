#!/usr/bin/env python3
"""Closed-loop launch forms side by side (same call, same box): us per env step of dpenv_policy_rollout for every arithmetic x launch
form at 65 536 envs (256-env workgroups: an env and a network wave per SIMD) and 32 768 envs (128-env workgroups: a SIMD per wave).
    python tools/time_closed_loop.py [--envs 65536,32768] [--steps 50] [--reps 8] [--out gpurun_out/closed_loop_forms.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', default='65536,32768')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--warm', type=int, default=40, help='untimed launches first: the MFMA-heavy forms wobble by 10-30 %% for the first ~20 launches of a process (clock / power ramp)')
    ap.add_argument('--out', default='')
    ap.add_argument('--forms', default='f16:two_wave,f16:one_wave,f32_actor:two_wave,f32_actor:one_wave,f32:two_wave,f32:one_wave')
    args = ap.parse_args()
    import torch
    import ml4ca_amd
    from ml4ca_amd import DpenvError
    from ml4ca_amd.policy import ActorCritic, policy_rollout
    dev = torch.device('cuda', 0)
    res = {}
    for n in [int(x) for x in args.envs.split(',')]:
        env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, auto_reset=True, seed=1)
        ac = ActorCritic(9, 7, (80, 80, 80), seed=0, device=dev)
        for spec in args.forms.split(','):
            prec, form = spec.split(':')
            try:
                ac.upload(env, precision=prec, launch_form=form)
            except DpenvError as e:
                res['%d/%s' % (n, spec)] = 'refused: %s' % e
                print('%7d %-22s refused (%s)' % (n, spec, str(e)[:60]), flush=True)
                continue
            env.reset()
            out = policy_rollout(env, args.steps, sample=True)
            for _ in range(args.warm):
                policy_rollout(env, args.steps, sample=True, out=out)
            torch.cuda.synchronize()
            best = 1e9
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                policy_rollout(env, args.steps, sample=True, out=out)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / args.steps * 1e6)
            ts.sort()
            res['%d/%s' % (n, spec)] = {'us_per_step_median': ts[len(ts) // 2], 'us_per_step_min': ts[0], 'env_steps_per_s': n / (ts[len(ts) // 2] * 1e-6)}
            print('%7d %-22s %7.2f us/step (min %6.2f)  %.3g env-steps/s' % (n, spec, ts[len(ts) // 2], ts[0], n / (ts[len(ts) // 2] * 1e-6)), flush=True)
            assert bool(torch.isfinite(out['logp']).all())
        del env
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
