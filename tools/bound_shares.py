#!/usr/bin/env python3
"""Which of the six state bounds ends the episodes of the config-2 workload (customEnv.py:207-213: |x~|, |y~| > 8 m, |psi~| > 45 deg,
|u| > 1.4, |v| > 0.30, |r| > 0.52 - the velocity bounds are the vessel's steady speeds WITH thrust losses, customEnv.py:17,26), under each
preset of the build-owned plant (dpenv_default_vessel_ex: no-loss = the default, thrust-loss), for two policies: the initial Gaussian policy
(std e^-0.5 around a zero mean, core.py:83) and the thesis' trained actor with its own exploration noise.  Runs on the GPU:
    python tools/bound_shares.py [n_envs] [steps]  > profiles/r05_bound_shares.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4ca_amd
from ml4ca_amd.policy import ActorCritic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
dev = torch.device('cuda', 0)
d = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'final_policy.npz'))
trained = ActorCritic.from_tensors({k.replace('.', '/'): d[k] for k in d.files if '.' in k}, device=dev)
NAMES = ('x~ 8 m', 'y~ 8 m', 'psi~ 45 deg', 'u 1.4 m/s', 'v 0.30 m/s', 'r 0.52 rad/s')
print('%d envs x %d steps of the config-2 workload (final / ext / cont_ang, terminate + auto-reset on, training resets, time limit 400); '
      'share of the TERMINATED episodes ended by each bound (an episode can exceed several at once)' % (n, T))
for preset in ('no_loss', 'thrust_loss'):
    for policy in ('initial Gaussian', 'trained actor + its noise'):
        env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=True, auto_reset=True, seed=3, vessel_params=ml4ca_amd.default_vessel(preset))
        b = torch.tensor(env.real_ss_bounds, device=dev)
        g = torch.Generator(device=dev).manual_seed(5)
        obs = env.reset().clone()
        fin = torch.zeros((n, 9), device=dev)
        hits = torch.zeros(6, device=dev)
        first = torch.zeros(6, device=dev)
        n_term = n_cut = 0
        ret = 0.0
        umax = torch.zeros(3, device=dev)
        for t in range(T):
            noise = torch.randn((n, 7), generator=g, device=dev)
            if policy.startswith('initial'):
                act = noise * 0.6065
            else:
                mu, _ = trained.forward_ref(obs.float())
                act = mu + torch.exp(trained.log_std) * noise
            obs, rew, done, _ = env.step(act.contiguous(), final_obs=fin)
            term = (done & 1) != 0
            cut = ((done & 2) != 0) & ~term
            if bool(term.any()):
                over = fin[term][:, :6].abs() > b
                hits += over.float().sum(0)
                # the bound exceeded by the largest factor = the one that "ended" the episode when several are over
                lead = (fin[term][:, :6].abs() / b).argmax(1)
                first += torch.bincount(lead, minlength=6).float()
            n_term += int(term.sum())
            n_cut += int(cut.sum())
            ret += float(rew.mean())
            umax = torch.maximum(umax, obs[:, 3:6].abs().max(0).values)
        tot = max(n_term, 1)
        print('\npreset %-11s policy %-26s: %7d episodes terminated, %6d reached the time limit; reward per step %.3f; largest |u| |v| |r| seen at an episode start or later %.2f %.2f %.2f' % (
            preset, policy, n_term, n_cut, ret / T, float(umax[0]), float(umax[1]), float(umax[2])))
        print('    bound          ' + ''.join('%14s' % s for s in NAMES))
        print('    exceeded       ' + ''.join('%13.1f%%' % (100.0 * float(h) / tot) for h in hits))
        print('    by most        ' + ''.join('%13.1f%%' % (100.0 * float(h) / tot) for h in first))
        del env
