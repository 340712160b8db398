#!/usr/bin/env python3
"""The DYNPOS-FIT preset of the BUILD-OWNED plant (dpenv_default_vessel_ex(DPENV_VESSEL_DYNPOS_FIT); soft pins, no parity claim).

The default hull (calibrate_plant.py) is fitted to three things the reference records about its plant: the free drift in a 0.2 m/s current, the RL
box test on the recorded setpoints, the steady surge / yaw speeds (customEnv.py:13-14).  Round 6 brought two more into reach that the default
does not meet:
  (d) the reference's 32 recorded Cybersea STATION-KEEPING runs in that current from 16 directions (results/all_plots/dyn_pos/;
      tests/golden/cybersea_dynpos.npz, tests/calibration/dynpos_pin.py): holding station, the mean thrust given is minus the mean force of the
      water on the hull - the default hull asks for 1.5-2 x the sway force and 2.5-7 x the yaw moment the recorded commands (priced by the
      reference's own K n|n|) delivered;
  (e) the recorded steady SWAY speed without losses, 0.35 m/s (customEnv.py:14): the default hull reaches 0.29 (bow thruster at full thrust, stern
      thrust that keeps the heading).
This script refits the sway-yaw part of the hull (m22, m33, Yv, Yvv, Yr, Nv, Nr, Nrr, Nuv, Yur; surge terms and m11 stay, so the thrust-loss
preset's numbers - a surge fit - carry over unchanged) to (a)-(e) jointly: Nelder-Mead from the default hull on calibrate_plant's own cost + the mean
squared station-keeping residual in Fy and Mz (per 2 N / 2 N m) + the sway pin.  The result trades 0.01-0.02 m on rows (a), (b) and the open-loop
replay for (e) met exactly and the yaw-moment residual of (d) halved - which is why it is a PRESET and not the new default.

    python tests/calibration/fit_dynpos_preset.py            the table of DESIGN.md section 3 for the default hull and the shipped preset
    python tests/calibration/fit_dynpos_preset.py fit        re-run the fit (2-3 minutes) and print the fitted vector next to the shipped one

Lives under tests/ because it drives oracle/ code.
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O                                  # noqa: E402
from tests.calibration import dynpos_pin as DP                   # noqa: E402
from tests.calibration import fit_thrust_loss_preset as F        # noqa: E402
from tests.calibration import replay_cybersea as RC              # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
FREE = ('YV', 'YVV', 'YR', 'NV', 'NR', 'NRR', 'NUV', 'YUR', 'M22', 'M33')          # what the fit may move (XU, XUU, M11, M23 stay)
W_DYN, W_SWAY = 0.1, 0.5


def _cal():
    """calibrate_plant.py as a module (it is a script with a __main__ part): its records, its simulators, its cost"""
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        spec = importlib.util.spec_from_file_location('calibrate_plant', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'calibrate_plant.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def shipped():
    v = np.zeros(O.NPARAM, np.float64)
    O.lib().dpo_dynpos_fit_vessel_f64(O._p(v))
    return v


def default():
    return O.Oracle(O.make_config(), np.float64).vessel.copy()


class StationKeeping(object):
    """the 32 runs' mean thrust (hull-independent under the no-loss law) and relative flow: the residual of a candidate hull in closed form"""

    def __init__(self):
        w = DP.wrenches('no_loss')
        self.thrust, self.allocator = w['thrust_c'], w['allocator']
        d = np.load(os.path.join(G, 'cybersea_dynpos.npz'))
        psi = d['pose'][:, :, 2].astype(np.float64).mean(1)
        beta, V = np.radians(d['current_dir_deg'].astype(np.float64)), float(d['current_speed'])
        vN, vE = V * np.cos(beta), V * np.sin(beta)
        self.u = -(np.cos(psi) * vN + np.sin(psi) * vE)
        self.v = -(-np.sin(psi) * vN + np.cos(psi) * vE)

    def net(self, vec):
        """thrust given + force of the water on the hull at rest over ground in the run's current (r = 0), [32, 3]"""
        u, v = self.u, self.v
        m11, m22 = vec[0], vec[1]
        Xu, Xuu, Yv, Yvv, _, Nv = vec[4:10]
        Nuv = vec[24]
        need = np.stack([(Xu + Xuu * np.abs(u)) * u, (Yv + Yvv * np.abs(v)) * v, (Nv + Nuv * u) * v - (m11 - m22) * u * v], 1)
        return self.thrust - need


def rows(vec, cal=None, sk=None, replay_windows=None):
    """every soft pin of the plant for one hull vector"""
    cal = cal or _cal()
    sk = sk or StationKeeping()
    W = replay_windows or RC.load_windows()
    ed = cal.simulate_drift(vec) - cal.drift['pose'][1:]
    eb = cal.simulate_box(vec) - cal.box['pose'][1:]
    net = sk.net(vec)
    rep = RC.errors(RC.replay_numpy(W, vessel=vec), W)
    return dict(drift_pos=float(np.sqrt(np.mean(ed[:, 0] ** 2 + ed[:, 1] ** 2))), drift_yaw_deg=float(np.degrees(np.sqrt(np.mean(ed[:, 2] ** 2)))),
                box_N=float(np.sqrt(np.mean(eb[:, 0] ** 2))), box_E=float(np.sqrt(np.mean(eb[:, 1] ** 2))), box_yaw_deg=float(np.degrees(np.sqrt(np.mean(eb[:, 2] ** 2)))),
                replay_10s=(float(rep[50][0]), float(rep[50][1])), dynpos_rms=[float(x) for x in np.sqrt((net ** 2).mean(0))],
                manoeuvres={k: float(x) for k, x in F.manoeuvres(vec).items()})


def fit(maxfev=700):
    from scipy.optimize import minimize
    cal, sk = _cal(), StationKeeping()
    base = default()
    names = list(cal.NAMES)
    th0 = np.array([base[cal.IDX[n]] for n in names])
    free = [names.index(n) for n in FREE]

    def sway_quick(vec):
        # the steady sway of F.sway_speed in closed form (bow 9 N at lx 1.08, stern sideways thrust at lx -1.12 balancing the hull's Nv v): the
        # fit's inner loop cannot afford the 120 s simulation; the shipped vector is checked with the real manoeuvre (rows())
        Yv, Yvv, Nv = vec[6], vec[7], vec[9]
        vs = np.linspace(0.05, 0.6, 551)
        Fs = np.clip((9.0 * 1.08 - Nv * vs) / 1.12, -41.0, 41.0)
        return float(vs[np.argmin(np.abs(9.0 + Fs - (Yv + Yvv * vs) * vs))])

    def J(x):
        th = th0.copy()
        th[free] = x
        j = cal.cost(th, base)
        if j >= 1e5:
            return j
        vec = cal.vessel_from(th, base)
        net = sk.net(vec)
        return j + W_DYN * (np.mean((net[:, 1] / 2.0) ** 2) + np.mean((net[:, 2] / 2.0) ** 2)) + W_SWAY * ((sway_quick(vec) - 0.35) / 0.03) ** 2 * 0.1

    x0 = th0[free]
    res = minimize(J, x0, method='Nelder-Mead', options=dict(maxfev=maxfev, xatol=1e-2, fatol=1e-4, adaptive=True,
                   initial_simplex=np.vstack([x0] + [x0 + np.eye(len(x0))[i] * (0.25 * abs(x0[i]) + 5.0) for i in range(len(x0))])))
    th = th0.copy()
    th[free] = res.x
    return cal.vessel_from(th, base), dict(zip(names, th))


def main():
    cal, sk, W = _cal(), StationKeeping(), RC.load_windows()
    hulls = [('default hull', default()), ('dynpos-fit preset (shipped)', shipped())]
    if len(sys.argv) > 1 and sys.argv[1] == 'fit':
        vec, th = fit()
        print('fitted:', {k: round(float(v), 2) for k, v in th.items()})
        hulls.append(('fitted now (unrounded)', vec))
    idx = cal.IDX
    for tag, vec in hulls:
        r = rows(vec, cal, sk, W)
        print('%s: %s' % (tag, {k: round(float(vec[i]), 2) for k, i in idx.items()}))
        print('   free drift 60 s: %.2f m / %.1f deg rms | box test (thesis\' actor, recorded setpoints): %.2f m N, %.2f m E, %.1f deg | open-loop replay 10 s: %.2f m / %.1f deg' % (
            r['drift_pos'], r['drift_yaw_deg'], r['box_N'], r['box_E'], r['box_yaw_deg'], r['replay_10s'][0], r['replay_10s'][1]))
        print('   station keeping, 32 runs: rms net wrench Fx %.2f N, Fy %.2f N, Mz %.2f N m | steady surge %+.2f / %+.2f m/s, sway %.3f m/s (recorded 0.35), yaw %.3f rad/s (0.60)' % (
            r['dynpos_rms'][0], r['dynpos_rms'][1], r['dynpos_rms'][2], r['manoeuvres']['surge_ahead'], r['manoeuvres']['surge_astern'], r['manoeuvres']['sway'], r['manoeuvres']['yaw']))


if __name__ == '__main__':
    main()
