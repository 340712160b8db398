#!/usr/bin/env python3
"""Open-loop replay of recorded Cybersea runs through the BUILD-OWNED plant (soft validation, no parity claim).

tests/golden/cybersea_replay.npz holds seven recorded runs of the reference's simulator - box test, large setpoint
changes, box test in a 0.2 m/s current, under the QP, pseudo-inverse and RL allocators - with the thruster commands
that produced them.  Every 5 s a 10 s window starts from the recorded pose (velocity: finite difference of the
record), the recorded commands are applied open loop, and the predicted pose is compared with the record after
2 / 5 / 10 s.  Unlike the closed-loop box comparison no controller is there to hide a plant mismatch.

    python tests/calibration/replay_cybersea.py          (CPU: float64 oracle plant; prints the table in DESIGN.md section 3)

Lives under tests/ because it drives oracle/ code.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, 'tests', 'golden')
HORIZONS = (10, 25, 50)            # env steps of 0.2 s: 2, 5, 10 s
STRIDE, START = 25, 10


def load_windows(runs=None, length=50):
    """All windows of the chosen runs as one batch: eta0, nu0 [n,3]; actions [length, n, 7] for the final/cont_ang env
    (thrust/100 and the sin/cos heads of the recorded azimuths - scale_and_clip and atan2 give the commands back);
    truth [length, n, 3]; current [2, n]; run index [n]."""
    d = np.load(os.path.join(G, 'cybersea_replay.npz'))
    names = [str(x) for x in d['runs']]
    eta0, nu0, act, truth, cur, idx = [], [], [], [], [], []
    for k, r in enumerate(names):
        if runs is not None and r not in runs:
            continue
        pose, nu = d[r + '_pose'].astype(np.float64), d[r + '_nu'].astype(np.float64)
        n, a = d[r + '_n'].astype(np.float64), d[r + '_a'].astype(np.float64)
        assert np.allclose(a[:, 0], np.pi / 2, atol=1e-6)                    # the bow azimuth never moved
        A = np.concatenate([n / 100.0, np.sin(a[:, 1:2]), np.cos(a[:, 1:2]), np.sin(a[:, 2:3]), np.cos(a[:, 2:3])], 1)
        i = START
        while i + length < len(pose):
            eta0.append(pose[i]); nu0.append(nu[i]); act.append(A[i:i + length]); truth.append(pose[i + 1:i + length + 1])
            cur.append(d[r + '_current']); idx.append(k)
            i += STRIDE
    return dict(eta0=np.array(eta0), nu0=np.array(nu0), act=np.array(act).transpose(1, 0, 2).copy(),
                truth=np.array(truth).transpose(1, 0, 2).copy(), current=np.array(cur).T.copy(), run=np.array(idx), names=names)


def replay_oracle(W, vessel=None, dtype=np.float64):
    """pose [length, n, 3] predicted by the oracle env (plant live, no termination) under the recorded commands"""
    from oracle import oracle as O
    n = W['eta0'].shape[0]
    orc = O.Oracle(O.make_config(terminate=0, current_enabled=1), dtype, vessel=vessel)
    st, ctr = orc.new_state(n)
    orc.reset(st, ctr, init=np.concatenate([W['eta0'].T, W['nu0'].T], 0).astype(dtype), ref=np.zeros((3, n), dtype))
    cur = W['current'].astype(dtype)
    out = np.zeros(W['truth'].shape)
    for t in range(W['act'].shape[0]):
        orc.step(st, ctr, W['act'][t].astype(dtype), current=cur)
        out[t] = st[0:3].T
    return out


def replay_numpy(W, vessel=None, h=0.01, n_substeps=20):
    """The same replay with the plant written out in NumPy (vectorised over the windows): an independent second
    implementation of the model equations of DESIGN.md section 3, used to cross-check the C oracle's plant."""
    from oracle import oracle as O
    v = np.asarray(O.Oracle(O.make_config(), np.float64).vessel if vessel is None else vessel, np.float64)
    m11, m22, m23, m33 = v[0:4]
    Xu, Xuu, Yv, Yvv, Yr, Nv, Nr, Nrr = v[4:12]
    Kf, Kr, lx, ly = v[12:15], v[15:18], v[18:21], v[21:24]
    Nuv, Yur = v[24], v[25]
    det = m22 * m33 - m23 * m23
    i22, i23, i33 = m33 / det, -m23 / det, m22 / det
    N, E, psi = (W['eta0'][:, k].copy() for k in range(3))
    u, vv, r = (W['nu0'][:, k].copy() for k in range(3))
    vcN, vcE = W['current'][0] * np.cos(W['current'][1]), W['current'][0] * np.sin(W['current'][1])
    out = np.zeros(W['truth'].shape)
    for t in range(W['act'].shape[0]):
        A = W['act'][t]
        n = np.clip(A[:, 0:3] * 100.0, -100.0, 100.0)
        al = np.stack([np.full(len(N), np.pi / 2), np.arctan2(A[:, 3], A[:, 4]), np.arctan2(A[:, 5], A[:, 6])], 1)
        F = np.where(n >= 0, Kf, Kr) * n * np.abs(n)
        tx, ty = (np.cos(al) * F).sum(1), (np.sin(al) * F).sum(1)
        tn = ((lx * np.sin(al) - ly * np.cos(al)) * F).sum(1)
        c, s = np.cos(psi), np.sin(psi)
        u, vv = u - (c * vcN + s * vcE), vv - (-s * vcN + c * vcE)              # relative velocity
        for _ in range(n_substeps):
            c13, c23 = -(m22 * vv + m23 * r), m11 * u
            fx = tx - c13 * r - (Xu + Xuu * np.abs(u)) * u
            fy = ty - c23 * r - ((Yv + Yvv * np.abs(vv)) * vv + (Yr + Yur * u) * r)
            fn = tn + (c13 * u + c23 * vv) - ((Nv + Nuv * u) * vv + (Nr + Nrr * np.abs(r)) * r)
            u, vv, r = u + h * fx / m11, vv + h * (i22 * fy + i23 * fn), r + h * (i23 * fy + i33 * fn)
            N, E = N + h * (c * u - s * vv + vcN), E + h * (s * u + c * vv + vcE)
            d = h * r
            psi = psi + d
            c, s = c * (1 - 0.5 * d * d) - s * d, s * (1 - 0.5 * d * d) + c * d    # second-order rotation, re-seeded per step
        c, s = np.cos(psi), np.sin(psi)
        u, vv = u + (c * vcN + s * vcE), vv + (-s * vcN + c * vcE)
        out[t, :, 0], out[t, :, 1], out[t, :, 2] = N, E, psi
    return out


def errors(pred, W, sel=None):
    """{horizon_steps: (rms position error [m], rms yaw error [deg])}"""
    res = {}
    for h in HORIZONS:
        e = pred[h - 1] - W['truth'][h - 1]
        if sel is not None:
            e = e[sel]
        res[h] = (float(np.sqrt((e[:, :2] ** 2).sum(1).mean())), float(np.degrees(np.sqrt((e[:, 2] ** 2).mean()))))
    return res


def constant_velocity(W):
    """the no-model predictor: keep the initial velocity over ground and yaw rate"""
    c, s = np.cos(W['eta0'][:, 2]), np.sin(W['eta0'][:, 2])
    rate = np.stack([c * W['nu0'][:, 0] - s * W['nu0'][:, 1], s * W['nu0'][:, 0] + c * W['nu0'][:, 1], W['nu0'][:, 2]], 1)
    k = np.arange(1, W['truth'].shape[0] + 1)[:, None, None] * 0.2
    return W['eta0'][None] + k * rate[None]


if __name__ == '__main__':
    W = load_windows()
    pred, cv = replay_oracle(W), constant_velocity(W)
    fmt = lambda r: '  '.join('%4.1f s: %.2f m %5.1f deg' % (h * 0.2, r[h][0], r[h][1]) for h in HORIZONS)
    print('%-22s %4s  %s' % ('run', 'win', 'open-loop prediction error of the default hull (rms over windows)'))
    for k, name in enumerate(W['names']):
        sel = W['run'] == k
        print('%-22s %4d  %s' % (name, sel.sum(), fmt(errors(pred, W, sel))))
    print('%-22s %4d  %s' % ('all', len(W['run']), fmt(errors(pred, W))))
    print('%-22s %4d  %s' % ('constant velocity', len(W['run']), fmt(errors(cv, W))))
    print('%-22s %4d  %s' % ('stay put', len(W['run']), fmt(errors(np.repeat(W['eta0'][None], W['truth'].shape[0], 0), W))))
