#!/usr/bin/env python3
"""The THRUST-LOSS preset of the BUILD-OWNED plant (dpenv_default_vessel_ex(DPENV_VESSEL_THRUST_LOSS); soft pins, no parity claim).

The reference records TWO sets of steady full-thrust speeds of its plant (customEnv.py:13-18): without thrust losses surge +2.20 / -1.60 m/s,
sway +-0.35 m/s, yaw +-0.60 rad/s - the default hull is fitted to those (calibrate_plant.py) - and WITH thrust losses +1.4 / -1.1 m/s,
+-0.30 m/s, +-0.52 rad/s, which are also the velocity bounds it trains with (customEnv.py:26: "these are its REAL limits").  This script
derives a second parameter vector for the second set.  What it may touch: the thrust gains K (forward and reverse slots, DPENV_P_KF_* /
DPENV_P_KR_*: a thrust LOSS is a property of the thrusters) - the hull (mass, damping) stays the calibrated one, so the free-drift record,
which involves no thrust, is reproduced exactly as before.  Manoeuvres (float64 oracle plant, steady state after 120 s):
    surge ahead / astern   stern thrusters +-100 % at azimuth 0, bow off
    sway                   bow +100 % at its fixed 90 deg, stern azimuths at 90 deg with the thrust that keeps the heading (r = 0)
    yaw                    bow +100 % at 90 deg, stern +100 % at -90 deg ("rotating stern azimuths only", customEnv.py:13)
Three gains (stern forward, stern reverse, bow) are determined by surge ahead, surge astern and sway.  The yaw pin is then an OUTCOME: one
constant gain per thruster and direction cannot also meet it - at 1.4 m/s ahead the stern thrusters must deliver 44 % of their bollard
thrust, in the yaw manoeuvre (no inflow) the record wants 87 % - i.e. Cybersea's loss depends on the inflow speed, which nothing in the
reference tree describes.

    python tests/calibration/fit_thrust_loss_preset.py        prints the vector and the table of DESIGN.md section 3
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402

P = dict(KF_BOW=12, KF_PORT=13, KF_STAR=14, KR_BOW=15, KR_PORT=16, KR_STAR=17)
PINS_NO_LOSS = dict(surge_ahead=2.20, surge_astern=-1.60, sway=0.35, yaw=0.60)          # customEnv.py:14
PINS_LOSS = dict(surge_ahead=1.4, surge_astern=-1.1, sway=0.30, yaw=0.52)               # customEnv.py:17


def steady(vessel, n_pct, alpha, seconds=120.0):
    orc = O.Oracle(O.make_config(terminate=0), np.float64, vessel=vessel)
    e, v = np.zeros(3), np.zeros(3)
    for _ in range(int(seconds / 0.2)):
        e, v = orc.plant(e, v, n_pct, alpha)
    return v


def bisect(f, lo, hi, it=60):
    flo = f(lo)
    for _ in range(it):
        mid = 0.5 * (lo + hi)
        fm = f(mid)
        if (fm > 0) == (flo > 0):
            lo, flo = mid, fm
        else:
            hi = mid
    return 0.5 * (lo + hi)


def sway_speed(vessel, sign=1.0):
    """steady sway with the bow thruster at full thrust and the stern thrust that keeps the heading: (v, stern percent)"""
    al = [np.pi / 2, np.pi / 2, np.pi / 2]
    ns = bisect(lambda n: steady(vessel, [sign * 100.0, sign * n, sign * n], al)[2] * sign, 0.0, 100.0, it=40)
    return steady(vessel, [sign * 100.0, sign * ns, sign * ns], al)[1], ns


def manoeuvres(vessel):
    return dict(surge_ahead=steady(vessel, [0, 100, 100], [np.pi / 2, 0, 0])[0],
                surge_astern=steady(vessel, [0, -100, -100], [np.pi / 2, 0, 0])[0],
                sway=sway_speed(vessel)[0], sway_to_port=sway_speed(vessel, -1.0)[0],
                yaw=steady(vessel, [100, 100, 100], [np.pi / 2, -np.pi / 2, -np.pi / 2])[2])


def fit(base):
    v = np.array(base, np.float64)

    def with_gains(kf_s=None, kr_s=None, k_b=None):
        w = v.copy()
        if kf_s is not None:
            w[P['KF_PORT']] = w[P['KF_STAR']] = kf_s
        if kr_s is not None:
            w[P['KR_PORT']] = w[P['KR_STAR']] = kr_s
        if k_b is not None:
            w[P['KF_BOW']] = w[P['KR_BOW']] = k_b
        return w

    kf_s = bisect(lambda k: steady(with_gains(kf_s=k), [0, 100, 100], [np.pi / 2, 0, 0])[0] - PINS_LOSS['surge_ahead'], 1e-5, v[P['KF_PORT']])
    v = with_gains(kf_s=kf_s)
    kr_s = bisect(lambda k: -steady(with_gains(kr_s=k), [0, -100, -100], [np.pi / 2, 0, 0])[0] + PINS_LOSS['surge_astern'], 1e-5, v[P['KR_PORT']])
    v = with_gains(kr_s=kr_s)
    k_b = bisect(lambda k: sway_speed(with_gains(k_b=k))[0] - PINS_LOSS['sway'], 1e-5, v[P['KF_BOW']], it=30)
    return with_gains(k_b=k_b)


if __name__ == '__main__':
    base = O.Oracle(O.make_config(), np.float64).vessel.copy()
    loss = fit(base)
    # four significant digits are what goes into dpenv_default_vessel_ex / dpo (the speeds below are those of the ROUNDED vector)
    for k in P.values():
        loss[k] = float('%.4g' % loss[k])
    print('thrust-loss preset: K forward (bow, port, star) = %s   K reverse = %s   [N per percent squared]' % (
        ', '.join('%.4g' % loss[k] for k in (12, 13, 14)), ', '.join('%.4g' % loss[k] for k in (15, 16, 17))))
    print('share of the bollard thrust of the no-loss preset: stern ahead %.2f, stern astern %.2f, bow %.2f' % (
        loss[13] / base[13], loss[16] / base[16], loss[12] / base[12]))
    print('%-14s %10s %10s   %10s %10s' % ('manoeuvre', 'no-loss', 'recorded', 'loss', 'recorded'))
    m0, m1 = manoeuvres(base), manoeuvres(loss)
    for key in ('surge_ahead', 'surge_astern', 'sway', 'sway_to_port', 'yaw'):
        pin = 'sway' if key == 'sway_to_port' else key
        sgn = -1.0 if key == 'sway_to_port' else 1.0
        print('%-14s %10.3f %10.2f   %10.3f %10.2f' % (key, m0[key], sgn * PINS_NO_LOSS[pin], m1[key], sgn * PINS_LOSS[pin]))
    try:
        from tests.calibration import replay_cybersea as RC
        W = RC.load_windows()
        for name, vec in (('no-loss (default)', base), ('thrust-loss', loss)):
            e = RC.errors(RC.replay_oracle(W, vessel=vec), W)
            print('open-loop replay of the recorded Cybersea commands, %-18s %s' % (name + ':', '  '.join('%4.1f s: %.2f m %5.1f deg' % (h * 0.2, e[h][0], e[h][1]) for h in RC.HORIZONS)))
    except Exception as ex:      # pragma: no cover
        print('replay skipped:', ex)
