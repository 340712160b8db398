#!/usr/bin/env python3
"""Station-keeping records of the reference's simulator as a low-speed hull-force pin.

    python3 -B tests/golden/gen_cybersea_dynpos.py            (build container only: reads /root/reference, writes numbers)

results/all_plots/dyn_pos/ holds 32 Cybersea runs of 60 s: the pseudo-inverse allocator and the RL allocator with integral
action, each holding station at the origin, heading 0, in a 0.2 m/s current (current_box_test/plot_pos.py:78) that comes from
16 directions - the angle in the file name (plot_pos.py:24 calls the list `headings`; the recorded heading is 0 +- 3 deg in
every run, it is the CURRENT that turns: -158 ... 180 deg).  Each run has the commands Cybersea received: stern efforts
(starboard first, box_test/plot_act.py:52-53) and pod angles at 5-7 Hz, bow throttle and angle (90 deg) - SURVEY appendix E.

A vessel that holds station does not accelerate on average, so the mean of the thrust it was given IS minus the mean force
and moment of the water on its hull at that relative flow angle: a 16-point measurement of the current's force and moment on
Cybersea's hull at 0.2 m/s - the quantity DESIGN.md section 3 could only argue from one heading before.  The commands are
noisy (both allocators chatter), so the series is kept, not just its mean: the thrust law is non-linear (K n|n|, and the
thrust-loss preset's inflow term), and the tests evaluate it per sample.

Stored, per run, on the env's 0.2 s grid over t = 15 ... 59.7 s (the first 15 s are the transient from rest): commands in force
(zero-order hold) n [%] and alpha [rad] in env order bow, port, star; pose relative to the setpoint (N, E [m], heading [rad]).
Data only: tests/golden/cybersea_dynpos.npz.
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

REF = '/root/reference/results/all_plots/dyn_pos'
OUT = os.path.dirname(os.path.abspath(__file__))
ANGLES = [-158, -135, -113, -90, -68, -45, -23, 0, 23, 45, 68, 90, 113, 135, 158, 180]      # dyn_pos/plot_pos.py:24
ALLOCATORS = ['pseudo', 'RLintegral']                                                       # dyn_pos/plot_pos.py:25,27
T0, DT = 15.0, 0.2


def main():
    n_all, a_all, pose_all, lab = [], [], [], []
    tq = None
    for m in ALLOCATORS:
        for h in ANGLES:
            p = os.path.join(REF, 'bagfile__%s%ddeg_' % (m, h))
            g = lambda s: np.genfromtxt(p + s, delimiter=',', skip_header=1)
            eta, ref = g('observer_eta_ned.csv'), g('reference_filter_state_desired.csv')
            st, an, bw = g('thrusterAllocation_stern_thruster_setpoints.csv'), g('thrusterAllocation_pod_angle_input.csv'), g('bow_control.csv')
            t0 = eta[0, 0]
            te = (eta[:, 0] - t0) * 1e-9
            if tq is None:
                tq = np.arange(T0, 59.75, DT)
            assert te[-1] > tq[-1], (m, h, te[-1])

            def zoh(t, x):
                return x[np.clip(np.searchsorted((t - t0) * 1e-9, tq, side='right') - 1, 0, len(t) - 1)]

            n = np.stack([zoh(bw[:, 0], bw[:, 1]), zoh(st[:, 0], st[:, 2]), zoh(st[:, 0], st[:, 1])], 1)          # bow, port, star [%]
            a = np.radians(np.stack([zoh(bw[:, 0], bw[:, 2]), zoh(an[:, 0], an[:, 1]), zoh(an[:, 0], an[:, 2])], 1))
            pose = np.stack([np.interp(tq, te, eta[:, 1]) - ref[-1, 1], np.interp(tq, te, eta[:, 2]) - ref[-1, 2],
                             np.radians(np.interp(tq, te, eta[:, 6]) - ref[-1, 3])], 1)
            n_all.append(n); a_all.append(a); pose_all.append(pose); lab.append((ALLOCATORS.index(m), h))
    out = os.path.join(OUT, 'cybersea_dynpos.npz')
    np.savez_compressed(out, t=tq, n=np.asarray(n_all, np.float32), alpha=np.asarray(a_all, np.float32), pose=np.asarray(pose_all, np.float32),
                        allocator=np.array([l[0] for l in lab], np.int32), current_dir_deg=np.array([l[1] for l in lab], np.float32),
                        current_speed=np.float32(0.2), allocators=np.array(ALLOCATORS))
    print('wrote %s: %d runs x %d samples (%d bytes)' % (out, len(lab), len(tq), os.path.getsize(out)))


if __name__ == '__main__':
    main()
