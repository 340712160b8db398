#!/usr/bin/env python3
"""Where the two waves of the closed-loop rollout spend their time (MI355X).

Needs a library built with -DDPENV_WS_PROFILE (the kernel then accumulates s_memtime around its hand-over waits and
leaves the totals in rows 0-4 of the logp block):

    cd ml4ca_amd/csrc && hipcc --offload-arch=gfx950 $(grep -m1 '^CXXFLAGS' Makefile | cut -d= -f2- | sed 's/$(BLOCK)/64/') \
        -DDPENV_WS_PROFILE -shared -o ../../build/ab/prof.so dpenv_kernels.hip dpenv_policy.hip dpenv_policy_ws.hip dpenv_policy_x.hip dpenv_policy_xws1.hip dpenv_policy_xws2.hip dpenv_api.hip
    (add -DDPENV_DEV_FAST to build the shipped configuration only: seconds instead of minutes)
    DPENV_LIB=$PWD/build/ab/prof.so python tools/ws_profile.py
"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4ca_amd
from ml4ca_amd.policy import ActorCritic, policy_rollout

n, T = int(os.environ.get('WS_PROFILE_ENVS', 65536)), 50
prec = os.environ.get('WS_PROFILE_PRECISION', 'f16')           # f16 | f32_actor | f32 (round 3: every arithmetic has the two-wave form)
env = ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True)
ActorCritic(9, 7, (80, 80, 80), device=env.device).upload(env, precision=prec, launch_form='two_wave')
print('%d envs, %s' % (n, prec))
env.reset()
for _ in range(int(os.environ.get('WS_PROFILE_WARM', 30))):          # clocks ramp over the first ~20 launches
    out = policy_rollout(env, T, sample=True)
lp = out['logp'].double()
e_tot, m_tot = float(lp[2].mean()), float(lp[4].mean())
print('env wave:     waits for the actor mean %4.1f %%, for the value %4.1f %%, busy %4.1f %%' % (
    100 * float(lp[0].mean()) / e_tot, 100 * float(lp[1].mean()) / e_tot, 100 * (1 - float((lp[0] + lp[1]).mean()) / e_tot)))
print('network wave: waits for the observation %4.1f %%, busy %4.1f %%' % (100 * float(lp[3].mean()) / m_tot, 100 * (1 - float(lp[3].mean()) / m_tot)))
# s_memtime tick -> us from the launch's own duration (timed with events below)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
out2 = policy_rollout(env, T, sample=True)
ev1.record()
torch.cuda.synchronize()
tick_us = ev0.elapsed_time(ev1) * 1e3 / float(out2['logp'].double()[2].mean())
print('per step (us): launch %.2f | network wave: actor %.2f, critic (+ pre-reset critic) %.2f, wait obs %.2f | env wave: env_step %.2f, noise %.2f, '
      'wait mu %.2f, wait v %.2f; mu -> env_step %.2f, env_step -> obs posted %.2f, obs posted -> v wait done %.2f' % (e_tot * tick_us / T, float(lp[5].mean()) * tick_us / T, float(lp[6].mean()) * tick_us / T,
                                      float(lp[3].mean()) * tick_us / T, float(lp[7].mean()) * tick_us / T, float(lp[8].mean()) * tick_us / T,
                                      float(lp[0].mean()) * tick_us / T, float(lp[1].mean()) * tick_us / T,
                                      float(lp[9].mean()) * tick_us / T, float(lp[10].mean()) * tick_us / T, float(lp[11].mean()) * tick_us / T))
