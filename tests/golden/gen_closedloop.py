#!/usr/bin/env python3
"""Closed-loop fixtures: the reference's OWN env classes stepped around a plant that moves.

    python3 -B tests/golden/gen_closedloop.py            (build container only: needs /root/reference)

The reference env (specific/customEnv.py Revolt / RevoltSimple / RevoltLimited / RevoltFinal, imported read-only
behind the gym / keras stubs of tools/gen_golden.py) is constructed on oracle/twin_shim.TwinShim, an object with the
reference's plant seam val()/step() (digitwin.py:50-114,213-219) over the oracle's float64 plant.  env.reset() and
env.step() then run exactly as in training (customEnv.py:92-194): each recorded step holds the pre-step bookkeeping,
the action, the optional new_ref, and the reference's obs / reward / done.  The plant is build-owned (Cybersea is
closed source), so what these vectors pin is the composition AROUND it - order of command writes and plant reads,
previous-thrust lag, late setpoints, the reset handshake - on trajectories that really move, to float64 round-off.

Lives under tests/ because it drives oracle/ code (test infrastructure); one interpreter per variant (quirk Q9).
Writes data only: tests/golden/closedloop_<mode>.npz.
"""
import os
import subprocess
import sys

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
E, T = 6, 160


def f32(x):
    return np.asarray(x, np.float32).astype(np.float64)


def gen(mode):
    from tools import gen_golden as GG
    from oracle.twin_shim import TwinShim
    GG.install_stubs()
    sys.path.insert(0, GG.WW)
    import specific.customEnv as CE
    cls_name, kw, act_dim = GG.MODES[mode]
    ext = mode != 'simple'                     # the reference's simple + extended state raises (customEnv.py:319)
    od = 9 if ext else 6
    rng = np.random.RandomState(4321 + sorted(GG.MODES).index(mode))
    np.random.seed(99 + sorted(GG.MODES).index(mode))          # the reference's reset samplers draw from np.random
    R = dict(action=np.zeros((E, T, act_dim)), new_ref=np.zeros((E, T, 3)), use_new_ref=np.zeros((E, T), np.uint8),
             eta=np.zeros((E, T, 3)), nu=np.zeros((E, T, 3)), prev_thrust=np.zeros((E, T, 3)),
             angles=np.zeros((E, T, 3)), ref=np.zeros((E, T, 3)), obs=np.zeros((E, T, od)), reward=np.zeros((E, T)),
             done=np.zeros((E, T), np.uint8), eta_after=np.zeros((E, T, 3)), nu_after=np.zeros((E, T, 3)),
             obs0=np.zeros((E, od)), init_eta=np.zeros((E, 3)), init_nu=np.zeros((E, 3)),
             reset_substeps=np.zeros(E, np.int64))
    for e in range(E):
        twin = TwinShim()
        env = getattr(CE, cls_name)(twin, extended_state=ext, **kw)
        # episodes 0-3: training reset (sampled pose and velocity); 4-5: testing reset on the 5 m circle
        if e >= 4:
            env.testing = True
        obs0 = env.reset() if e < 4 else env.reset(fixed_point=e - 3)
        R['obs0'][e] = np.asarray(obs0).ravel()
        R['init_eta'][e], R['init_nu'][e] = twin.eta, twin.nu
        R['reset_substeps'][e] = sum(v for m, f, v in twin.log if m == 'step')
        assert twin.n_substeps_run == 0                       # the reset handshake holds the plant (StateResetOn)
        # action scripts
        if e == 0:
            A = rng.normal(0.0, 0.607, size=(T, act_dim))                       # the initial policy's noise
        elif e == 1:
            A = np.cumsum(rng.normal(0.0, 0.08, size=(T, act_dim)), 0)          # slow random walk: sustained motion
        elif e == 2:
            A = np.tile(rng.uniform(0.6, 1.0, size=act_dim), (T, 1))            # hard over: runs into the bounds
            A[:, 0] *= -1.0
        elif e == 3:
            A = rng.normal(0.0, 0.3, size=(T, act_dim)) + 0.5 * np.sin(np.arange(T)[:, None] / 9.0 + np.arange(act_dim))
        else:
            A = rng.normal(0.0, 0.4, size=(T, act_dim))
        A = f32(A)
        for t in range(T):
            use = (e in (3, 5)) and (t % 37 == 5)
            nr = f32(rng.uniform(-4, 4, size=3) * np.array([1, 1, 0.1]))
            R['action'][e, t] = A[t]
            R['new_ref'][e, t], R['use_new_ref'][e, t] = nr, use
            R['eta'][e, t], R['nu'][e, t] = twin.eta, twin.nu
            R['prev_thrust'][e, t] = np.asarray(env.prev_thrust, float)
            R['angles'][e, t] = np.asarray(env.current_angles, float)
            R['ref'][e, t] = np.asarray(env.EF.get_NED_ref(), float).ravel()
            o, r, d, _ = env.step(A[t].copy(), new_ref=(list(nr) if use else None))
            R['obs'][e, t] = np.asarray(o).ravel()
            R['reward'][e, t] = float(np.asarray(r).ravel()[0])
            R['done'][e, t] = int(bool(d))
            R['eta_after'][e, t], R['nu_after'][e, t] = twin.eta, twin.nu
        assert twin.n_substeps_run == 20 * T
    R['vessel'] = TwinShim().vessel
    R['n_substeps'], R['substep_dt'] = np.array([20]), np.array([0.01])
    np.savez_compressed(os.path.join(OUT, 'closedloop_%s.npz' % mode), **R)
    print(mode, 'done fraction %.3f' % R['done'].mean(), 'max |eta|', np.abs(R['eta']).reshape(-1, 3).max(0))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        gen(sys.argv[1])
    else:
        for m in ('full', 'simple', 'limited', 'final_wrap', 'final_cont'):
            subprocess.check_call([sys.executable, '-B', os.path.abspath(__file__), m])
