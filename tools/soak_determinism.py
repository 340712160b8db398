#!/usr/bin/env python3
"""Soak: the closed loop at full size (65 536 envs, auto-reset, drifting current, in-kernel noise) run TWICE from the same seed in every
network arithmetic / launch form - every row of every launch must be identical between the two runs (any timing-dependent fault, e.g. a
missing wait state beside the matrix pipe, shows up as a difference), everything finite, no fault bits.
Usage: python tools/soak_determinism.py [launches=60] [randomise=0] [preset=no_loss] [current=0]    current C > 0 (round 6): every reset also draws the
episode's current, 0.2 +- C m/s from any direction (dpenv_set_current_randomisation: the shared training form with one class, the general per-env
kernels with randomised hulls); the currents join the digest.  randomise R > 0 (round 5): hulls re-drawn at every reset inside
the launches (the RND instantiations of the two-wave kernels, the function-call draw of the one-wave kernels); the table of hulls joins the digest.
preset thrust_loss: the nominal hull carries inflow thrust-loss coefficients (with R > 0 the general per-env kernels apply them; with R = 0 every env runs on
the preset itself as the handle's one class: the shared training form)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4ca_amd
from ml4ca_amd.policy import ActorCritic, policy_rollout

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
randomise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
preset = sys.argv[3] if len(sys.argv) > 3 else 'no_loss'
cur_rand = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
nominal = ml4ca_amd.default_vessel(preset)
T = 50
# round 3: every arithmetic in both launch forms, at 65 536 envs (256-env workgroups) and 32 768 envs (128-env workgroups); the one-wave form
# of an arithmetic must also give the digests of its two-wave form (same rows bit for bit)
ref = {}
for n, prec, form in [(nn, p, f) for nn in (65536, 32768) for p in ('f16', 'f32_actor', 'f32') for f in ('two_wave', 'one_wave')]:
    digests = []
    for run in range(2):
        env = ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True, seed=5, current=True, current_drift=True, max_ep_len=120,
                                         vessel_params=nominal if preset != 'no_loss' else None)
        ActorCritic(9, 7, (80, 80, 80), seed=1, device=env.device).upload(env, precision=prec, launch_form=form)
        env.set_current(torch.full((n,), 0.2, device=env.device), torch.full((n,), 2.356, device=env.device))
        if randomise > 0:
            env.set_vessel_randomisation(randomise, nominal=nominal)
        if cur_rand > 0:
            env.set_current_randomisation(cur_rand, 3.14159)
        env.reset()
        dig = []
        for _ in range(launches):
            out = policy_rollout(env, T, sample=True)
            assert bool(torch.isfinite(out['rew']).all()) and bool(torch.isfinite(out['val']).all()) and not bool((out['done'] & 4).any())
            dig.append(tuple(int(out[k].view(torch.int32 if out[k].dtype == torch.float32 else out[k].dtype).to(torch.int64).sum())
                             for k in ('obs', 'act', 'rew', 'val', 'logp', 'done', 'boot')))
        st, ctr = env.get_state()
        dig.append((int(st.view(torch.int32).to(torch.int64).sum()), int(ctr.to(torch.int64).sum())))
        if randomise > 0 or preset != 'no_loss':
            dig.append((int(env.get_vessel_params().view(torch.int32).to(torch.int64).sum()),))
        if cur_rand > 0:
            dig.append(tuple(int(x.view(torch.int32).to(torch.int64).sum()) for x in env.get_current() + env.get_current_mean()))
        digests.append(dig)
    same = digests[0] == digests[1]
    cross = ref.setdefault((n, prec), digests[0]) == digests[0]
    print('%-9s %-8s %d launches x %d steps x %d envs%s: two runs %s%s' % (prec, form, launches, T, n, (' (hulls re-drawn)' if randomise > 0 else '') + (' (currents re-drawn)' if cur_rand > 0 else '') + (' [%s]' % preset if preset != 'no_loss' else ''), 'IDENTICAL' if same else 'DIFFER',
                                                                        '' if form == 'two_wave' else (', = the two-wave form' if cross else ', DIFFERS from the two-wave form')))
    if not cross:
        sys.exit(1)
    if not same:
        bad = [i for i, (a, b) in enumerate(zip(*digests)) if a != b]
        print('  first differing launch:', bad[0])
        sys.exit(1)
