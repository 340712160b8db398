#!/usr/bin/env python3
"""Soak of the two-wave env kernels of round 4 (LDS mailboxes between an env wave and a row / reset wave): dpenv_rollout (rollout_ws_kernel)
and dpenv_step with auto-reset (step_kernel<.., RESETW>) at 65 536 and 32 768 envs with termination, auto-reset, reset_acts, drifting current and
setpoint switches, run TWICE from the same seed, and once in the one-wave form (config.step_one_wave): every row of every launch and the final
state must be identical in all three runs - a race in the hand-over would show as a difference.
Usage: python tools/soak_fused.py [launches=300] [randomise=0]     randomise R > 0 (round 5): dpenv_set_vessel_randomisation(R) - the reset wave / row wave
also draw and hand over the hulls (VES_ENV_RND instantiations); the table of hulls joins the digest"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4ca_amd

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
randomise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
T = 50
dev = torch.device('cuda', 0)


def digest(*ts):
    return tuple(int(t.view(torch.int32 if t.dtype == torch.float32 else t.dtype).to(torch.int64).sum()) for t in ts)


ok = True
for n in (65536, 32768):
    g = torch.Generator(device=dev).manual_seed(11)
    actions = torch.randn((T, n, 7), generator=g, device=dev) * 0.7
    refs = torch.randn((3, 3, n), generator=g, device=dev)
    runs = []
    for one_wave in (False, False, True):
        env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, auto_reset=True, terminate=True, max_ep_len=34, seed=9, reset_acts=True, current=True,
                                         current_drift=True, step_one_wave=one_wave)
        env.set_current(torch.full((n,), 0.2, device=dev), torch.full((n,), 2.356, device=dev))
        if randomise > 0:
            env.set_vessel_randomisation(randomise)
        env.reset()
        dig = []
        out = None
        for k in range(launches):
            if k % 3 == 2:
                # one launch in three through the one-launch-per-step path (reset wave), setpoints handed over on some steps
                for t in range(T):
                    o, r, d, _ = env.step(actions[t], new_ref=refs[t % 3] if t % 17 == 0 else None)
                dig.append(digest(o, r, d))
            else:
                out = env.rollout(actions, switch_steps=(0, 16, 33), refs=refs, out=out)
                assert bool(torch.isfinite(out[1]).all()) and not bool((out[2] & 4).any())
                dig.append(digest(*out))
        st, ctr = env.get_state()
        dig.append(digest(st, ctr))
        if randomise > 0:
            dig.append(digest(env.get_vessel_params()))
        runs.append(dig)
        del env
    same = runs[0] == runs[1]
    cross = runs[0] == runs[2]
    print('%d envs%s, %d launches x %d steps (two thirds fused, one third single steps): two runs %s; one-wave kernels %s' % (
        n, ', hulls re-drawn at every reset (+-%g %%)' % (100 * randomise) if randomise > 0 else '', launches, T, 'IDENTICAL' if same else 'DIFFER', 'IDENTICAL' if cross else 'DIFFER'), flush=True)
    ok = ok and same and cross
sys.exit(0 if ok else 1)
