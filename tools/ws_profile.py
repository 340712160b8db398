#!/usr/bin/env python3
"""Where the two waves of the closed-loop rollout spend their time (MI355X).

Needs a library built with -DDPENV_WS_PROFILE (the kernel then accumulates s_memtime around its hand-over waits and
leaves the totals in rows 0-4 of the logp block):

    cd ml4ca_amd/csrc && hipcc --offload-arch=gfx950 $(grep -m1 '^CXXFLAGS' Makefile | cut -d= -f2- | sed 's/$(BLOCK)/64/') \
        -DDPENV_WS_PROFILE -shared -o ../../build/ab/prof.so dpenv_kernels.hip dpenv_policy.hip dpenv_api.hip
    DPENV_LIB=$PWD/build/ab/prof.so python tools/ws_profile.py
"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4ca_amd
from ml4ca_amd.policy import ActorCritic, policy_rollout

n, T = 65536, 50
env = ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True)
ActorCritic(9, 7, (80, 80, 80), device=env.device).upload(env)
env.reset()
noise = torch.randn((T, n, 7), device=env.device)
for _ in range(3):
    out = policy_rollout(env, T, noise=noise)
lp = out['logp'].double()
e_tot, m_tot = float(lp[2].mean()), float(lp[4].mean())
print('env wave:     waits for the actor mean %4.1f %%, for the value %4.1f %%, busy %4.1f %%' % (
    100 * float(lp[0].mean()) / e_tot, 100 * float(lp[1].mean()) / e_tot, 100 * (1 - float((lp[0] + lp[1]).mean()) / e_tot)))
print('network wave: waits for the observation %4.1f %%, busy %4.1f %%' % (100 * float(lp[3].mean()) / m_tot, 100 * (1 - float(lp[3].mean()) / m_tot)))
