#!/usr/bin/env python3
"""The THRUST-LOSS preset of the BUILD-OWNED plant (dpenv_default_vessel_ex(DPENV_VESSEL_THRUST_LOSS); soft pins, no parity claim).

The reference records TWO sets of steady full-thrust speeds of its plant (customEnv.py:13-18): without thrust losses surge +2.20 / -1.60 m/s,
sway +-0.35 m/s, yaw +-0.60 rad/s - the default hull is fitted to those (calibrate_plant.py) - and WITH thrust losses +1.4 / -1.1 m/s,
+-0.30 m/s, +-0.52 rad/s, which are also the velocity bounds it trains with (customEnv.py:26: "these are its REAL limits").  This script
derives the second parameter vector.  What it may touch: the THRUSTERS - a thrust loss is theirs - i.e. the reverse gain of the stern
thrusters (DPENV_P_KR_*: from the no-loss astern speed, -1.60 m/s, which the default hull with its symmetric gains does not meet) and the
inflow-loss coefficients (DPENV_P_KLF_* / DPENV_P_KLR_*: F = K n|n| - Kl |n| u_a, the linear open-water characteristic, Fossen 2011 eq. 9.7).
The hull (mass, damping) stays the calibrated one, so the free-drift record, which involves no thrust, is reproduced exactly as before.
Manoeuvres (float64 oracle plant, steady state after 120 s):
    surge ahead / astern   stern thrusters +-100 % at azimuth 0, bow off
    sway                   bow +100 % at its fixed 90 deg, stern azimuths at 90 deg with the thrust that keeps the heading (r = 0)
    yaw                    bow +100 % at 90 deg, stern +100 % at -90 deg ("rotating stern azimuths only", customEnv.py:13)
Three numbers are determined by three pins: stern reverse gain by -1.60 m/s (no loss), stern Kl forward by +1.4 m/s, stern Kl reverse by
-1.1 m/s.  Sway and yaw are then OUTCOMES: an inflow-type loss that explains the surge speeds leaves the stern thrusters ~77 % of their bollard
thrust in the yaw manoeuvre - a constant gain reduced to fit +1.4 m/s would leave 44 % (yaw 0.35 rad/s; round 5's first form of this preset).
The bow thruster keeps its gain and gets no loss coefficient: the recorded sway needs all of its thrust.

    python tests/calibration/fit_thrust_loss_preset.py        prints the vector and the table of DESIGN.md section 3
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402

P = dict(KF_BOW=12, KF_PORT=13, KF_STAR=14, KR_BOW=15, KR_PORT=16, KR_STAR=17, KLF_BOW=26, KLF_PORT=27, KLF_STAR=28, KLR_BOW=29, KLR_PORT=30, KLR_STAR=31)
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


def no_loss_of(vessel):
    """the same thrusters with the loss switched off (what the reference measured 'with no thrust losses activated')"""
    w = np.array(vessel, np.float64)
    w[26:32] = 0.0
    return w


def fit(base):
    v = np.array(base, np.float64)

    def with_(**kw):
        w = v.copy()
        for k, x in kw.items():
            for name in (k + '_PORT', k + '_STAR'):
                w[P[name]] = x
        return w

    # stern reverse gain: -1.60 m/s astern without losses (customEnv.py:14)
    kr = bisect(lambda k: -steady(with_(KR=k), [0, -100, -100], [np.pi / 2, 0, 0])[0] + PINS_NO_LOSS['surge_astern'], 1e-5, v[P['KF_PORT']])
    v = with_(KR=kr)
    # inflow-loss coefficients of the stern thrusters: +1.4 ahead, -1.1 astern with losses (customEnv.py:17)
    klf = bisect(lambda k: PINS_LOSS['surge_ahead'] - steady(with_(KLF=k), [0, 100, 100], [np.pi / 2, 0, 0])[0], 0.0, 0.5)
    v = with_(KLF=klf)
    klr = bisect(lambda k: steady(with_(KLR=k), [0, -100, -100], [np.pi / 2, 0, 0])[0] - PINS_LOSS['surge_astern'], 0.0, 0.5)
    return with_(KLR=klr)


if __name__ == '__main__':
    base = O.Oracle(O.make_config(), np.float64).vessel.copy()
    loss = fit(base)
    # four significant digits are what goes into dpenv_default_vessel_ex / the oracle (the speeds below are those of the ROUNDED vector)
    for k in (P['KR_PORT'], P['KR_STAR'], P['KLF_PORT'], P['KLF_STAR'], P['KLR_PORT'], P['KLR_STAR']):
        loss[k] = float('%.4g' % loss[k])
    print('thrust-loss preset: stern K forward %.4g, K reverse %.4g [N per percent^2]; Kl forward %.4g, Kl reverse %.4g [N per percent per m/s]; bow unchanged' % (
        loss[13], loss[16], loss[27], loss[30]))
    print('%-14s %10s %10s   %12s %10s   %12s %10s' % ('manoeuvre', 'default', 'recorded', 'loss preset,', 'recorded', 'loss preset', 'recorded'))
    print('%-14s %10s %10s   %12s %10s   %12s %10s' % ('', '(no loss)', 'no loss', 'loss OFF', 'no loss', '', 'with loss'))
    m0, m1, m2 = manoeuvres(base), manoeuvres(no_loss_of(loss)), manoeuvres(loss)
    for key in ('surge_ahead', 'surge_astern', 'sway', 'sway_to_port', 'yaw'):
        pin = 'sway' if key == 'sway_to_port' else key
        sgn = -1.0 if key == 'sway_to_port' else 1.0
        print('%-14s %10.3f %10.2f   %12.3f %10.2f   %12.3f %10.2f' % (key, m0[key], sgn * PINS_NO_LOSS[pin], m1[key], sgn * PINS_NO_LOSS[pin], m2[key], sgn * PINS_LOSS[pin]))
    stern = steady(loss, [100, 100, 100], [np.pi / 2, -np.pi / 2, -np.pi / 2])
    print('stern thrust in the yaw manoeuvre: %.0f %% of the bollard thrust (inflow %.2f m/s)' % (100 * (1 - loss[27] * 100 * 1.12 * stern[2] / (loss[13] * 1e4)), 1.12 * stern[2]))
    try:
        from tests.calibration import replay_cybersea as RC
        W = RC.load_windows()
        for name, vec in (('default (no loss)', base), ('loss preset, loss off', no_loss_of(loss)), ('loss preset', loss)):
            e = RC.errors(RC.replay_oracle(W, vessel=vec), W)
            print('open-loop replay of the recorded Cybersea commands, %-22s %s' % (name + ':', '  '.join('%4.1f s: %.2f m %5.1f deg' % (h * 0.2, e[h][0], e[h][1]) for h in RC.HORIZONS)))
    except Exception as ex:      # pragma: no cover
        print('replay skipped:', ex)
