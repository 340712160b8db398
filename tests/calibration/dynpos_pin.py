#!/usr/bin/env python3
"""The reference's 32 recorded Cybersea station-keeping runs against the BUILD-OWNED plant (soft pin, no parity claim).

tests/golden/cybersea_dynpos.npz (tests/golden/gen_cybersea_dynpos.py from results/all_plots/dyn_pos/): two allocators hold station at
heading 0 in a 0.2 m/s current from 16 directions, with the thruster commands Cybersea received.  A vessel that holds station does not
accelerate on average, so for a plant that agreed with Cybersea's the recorded commands, applied to a hull AT REST over ground in that current,
would give zero mean acceleration.  What they give in the oracle's plant is measured here with the plant itself as the probe - one 10 ms
sub-step from rest per recorded sample, dv / h times the mass matrix = the net wrench:

    thrust[k]   mean wrench of the recorded commands in still water                        (the plant's thrust law alone)
    net[k]      the same in the run's current, hull at rest over ground                     (thrust + hull force at that relative flow)
    hull[k]     = net - thrust_in_current ... obtained as the net wrench of ZERO commands   (what the plant's hull asks the thrusters to cancel)

so net = thrust' + hull with thrust' the thrust in the current (the thrust-loss preset's inflow term sees the relative flow), and the question
DESIGN.md section 3 left open - "Cybersea's low-speed drag is about half of this hull's, or its thrust at small commands is above K n|n|" - is the
ratio  -thrust' / hull  per axis over 16 flow angles and two allocators that use the thrusters in entirely different ways.

    python tests/calibration/dynpos_pin.py          (CPU: float64 oracle plant; prints the table of DESIGN.md section 3)

Lives under tests/ because it drives oracle/ code.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, 'tests', 'golden')


def vessel(preset):
    from oracle import oracle as O
    v = np.zeros(O.NPARAM, np.float64)
    getattr(O.lib(), {'thrust_loss': 'dpo_thrust_loss_vessel_f64', 'dynpos_fit': 'dpo_dynpos_fit_vessel_f64'}.get(preset, 'dpo_default_vessel_f64'))(O._p(v))
    return v


def wrenches(preset='no_loss'):
    """per run: thrust (still water), thrust_c (in the current), hull, net = thrust_c + hull   [32, 3] each (Fx, Fy [N], Mz [N m], body frame),
    and the standard error of the mean of thrust_c (the commands chatter: batch means over 5 s)"""
    from oracle import oracle as O
    d = np.load(os.path.join(G, 'cybersea_dynpos.npz'))
    v = vessel(preset)
    h = 0.01
    orc = O.Oracle(O.make_config(n_substeps=1, substep_dt=h, terminate=0, current_enabled=1), np.float64, vessel=v)
    P = O.P if hasattr(O, 'P') else None
    m11, m22, m23, m33 = v[0], v[1], v[2], v[3]
    M = np.array([[m11, 0, 0], [0, m22, m23], [0, m23, m33]])
    n, al, pose = d['n'].astype(np.float64), d['alpha'].astype(np.float64), d['pose'].astype(np.float64)
    V, beta = float(d['current_speed']), np.radians(d['current_dir_deg'].astype(np.float64))
    R, T = n.shape[0], n.shape[1]
    out = {k: np.zeros((R, 3)) for k in ('thrust', 'thrust_c', 'hull', 'net', 'se')}

    def probe(eta, cmd_n, cmd_a, cur):
        _, nu1 = orc.plant(eta, np.zeros(3), cmd_n, cmd_a, current=cur)
        return M @ (nu1 / h)                 # from rest over ground: no Coriolis term of nu, dv / h = M^-1 (tau - hull(nu_r))

    for k in range(R):
        cur = np.array([V, beta[k]])
        th, ne, hu = np.zeros((T, 3)), np.zeros((T, 3)), np.zeros((T, 3))
        for t in range(T):
            eta = np.array([0.0, 0.0, pose[k, t, 2]])
            th[t] = probe(eta, n[k, t], al[k, t], np.zeros(2))
            ne[t] = probe(eta, n[k, t], al[k, t], cur)
            hu[t] = probe(eta, np.zeros(3), al[k, t], cur)
        out['thrust'][k], out['net'][k], out['hull'][k] = th.mean(0), ne.mean(0), hu.mean(0)
        tc = ne - hu
        out['thrust_c'][k] = tc.mean(0)
        nb = T // 25
        out['se'][k] = tc[:nb * 25].reshape(nb, 25, 3).mean(1).std(0, ddof=1) / np.sqrt(nb)
    out['allocator'], out['angle'] = d['allocator'], d['current_dir_deg']
    out['drift'] = np.stack([np.polyfit(d['t'], pose[k, :, j], 1)[0] for k in range(R) for j in range(3)]).reshape(R, 3)
    out['offset'] = pose.mean(1)
    return out


def ratios(w, sel=None):
    """least-squares s per axis in  -thrust_c = s * hull  over the selected runs (+ its standard error from the residuals)"""
    sel = np.ones(len(w['angle']), bool) if sel is None else sel
    s, e = np.zeros(3), np.zeros(3)
    for j in range(3):
        x, y = w['hull'][sel, j], -w['thrust_c'][sel, j]
        s[j] = (x * y).sum() / (x * x).sum()
        r = y - s[j] * x
        e[j] = np.sqrt((r * r).sum() / (len(x) - 1) / (x * x).sum())
    return s, e


def main():
    names = ('pseudo-inverse', 'RL + integral')
    for preset in ('no_loss', 'thrust_loss', 'dynpos_fit'):
        w = wrenches(preset)
        print('preset %s: mean wrench over t = 15 ... 60 s of the commands Cybersea received, in the oracle plant (body frame; Fx, Fy [N], Mz [N m])' % preset)
        print('%-15s %6s  %26s  %26s  %26s  %s' % ('allocator', 'flow', 'thrust in the current', 'hull asks for', 'net (0 = agreement)', 'recorded drift [mm/s, mdeg/s]'))
        for k in range(len(w['angle'])):
            print('%-15s %6.0f  [% 6.2f % 6.2f % 6.2f] +-%.1f  [% 6.2f % 6.2f % 6.2f]       [% 6.2f % 6.2f % 6.2f]       % .0f % .0f % .0f' % (
                names[w['allocator'][k]], w['angle'][k], *w['thrust_c'][k], w['se'][k, 1], *(-w['hull'][k]), *w['net'][k],
                w['drift'][k, 0] * 1e3, w['drift'][k, 1] * 1e3, np.degrees(w['drift'][k, 2]) * 1e3))
        for a in (0, 1, None):
            sel = (w['allocator'] == a) if a is not None else None
            s, e = ratios(w, sel)
            rms = np.sqrt((w['net'][sel if sel is not None else slice(None)] ** 2).mean(0))
            print('   %-15s thrust given / thrust the plant\'s hull asks for: Fx %.2f +- %.2f   Fy %.2f +- %.2f   Mz %.2f +- %.2f     rms net wrench [%.2f %.2f %.2f]' % (
                names[a] if a is not None else 'both', s[0], e[0], s[1], e[1], s[2], e[2], *rms))
        print()


if __name__ == '__main__':
    main()
