#!/usr/bin/env python3
"""Closed-loop box test (results/all_plots/box_test) of the thesis' trained policy in the build-owned plant.
Runs on the GPU: deterministic policy (test_policy.py:90), 4-corner setpoint sequence, 250 s; prints tracking errors,
IAE and energy-equivalent work.  A soft validation of the plant only: the policy was trained in Cybersea."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4ca_amd
from ml4ca_amd import evaluate as EV
from ml4ca_amd.policy import ActorCritic, policy_rollout

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'final_policy.npz'))
ac = ActorCritic.from_tensors({k.replace('.', '/'): d[k] for k in d.files if '.' in k}, device='cuda:0')
env = ml4ca_amd.BatchedRevoltEnv(n, terminate=False, time_limit=False)
ac.upload(env)
start = torch.zeros((3, n), device=env.device)
init = torch.zeros((6, n), device=env.device)
env.reset(init=init, new_ref=start.clone())
steps, refs = EV.box_schedule(start)
T = 1250
out = policy_rollout(env, T, noise=None, switch_steps=steps, refs=refs)
obs, act = out['obs'], out['act']
iae_tot, iae_cum = EV.iae(obs)
w = EV.work(EV.commanded_thrust(act))
print('IAE (thesis box test, RL in Cybersea ~ 30-40): %.2f' % float(iae_tot.mean()))
print('work W* bow/port/star: %s' % [round(float(x), 1) for x in w.mean(0)])
e = obs[:, 0, :3].cpu().numpy()
for t in list(steps) + [T - 1]:
    k = max(t - 1, 0)
    print('t=%6.1fs body-frame error before switch: x %.2f m  y %.2f m  yaw %.1f deg' % (k * 0.2, e[k, 0], e[k, 1], np.degrees(e[k, 2])))
print('reward mean %.3f (max 3.5)' % float(out['rew'].mean()))
print('thrust cmd range', float(act[..., :3].min()), float(act[..., :3].max()))
print('--- first leg (+5 m North at t = 10 s): N(t) in this plant vs Cybersea record (box_test/bagfile__RL_observer_eta_ned.csv)')
cyb = {15: 0.17, 20: 1.91, 25: 3.93, 30: 4.79, 40: 5.02, 50: 5.00}
for tt in (15, 20, 25, 30, 40, 50):
    k = int(tt / 0.2)
    print('t=%3d s  N = %.2f m   (Cybersea %.2f)   u = %.2f m/s' % (tt, 5.0 + e[k, 0], cyb[tt], float(obs[k, 0, 3])))
print('--- second leg (-5 m East at t = 60 s): E(t)')
cyb2 = {70: -0.58, 90: -4.68, 109: -5.08}
for tt in (70, 90, 109):
    k = int(tt / 0.2)
    print('t=%3d s  E = %.2f m   (Cybersea %.2f)' % (tt, -5.0 + e[k, 1], cyb2[tt]))
