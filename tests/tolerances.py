"""Parity tolerances (north star: outputs match the CPU reference to 1e-5 relative in fp32).

|a - b| <= RTOL * max(|b|, floor).  The floor is the natural scale of the inputs the quantity is
computed from: a body-frame error of a few mm is the difference of products of metre-sized
numbers, a reward near zero is the sum of O(1) terms, so their fp32 rounding error scales with
those inputs, not with the (cancelled) result.
"""
import numpy as np

RTOL_F32 = 1e-5
ATOL_F64 = 1e-11            # float64 oracle vs float64 reference: identical formulae, libm noise only

# obs = [x~, y~, psi~, u, v, r, n_bow/100, n_port/100, n_star/100]
OBS_FLOOR = np.array([1.0, 1.0, 0.1, 0.1, 0.1, 0.1, 0.01, 0.01, 0.01])
REWARD_FLOOR = 1.0          # sum of four O(1) parts (max 3.5 per step, customEnv.py reward)
PARTS_FLOOR = np.array([0.1, 1.0, 0.1, 0.1])
THRUST_FLOOR = 1.0          # percent, range +-100
ANGLE_FLOOR = 0.1           # rad, range +-pi
ETA_FLOOR = np.array([1.0, 1.0, 0.1])
NU_FLOOR = np.array([0.1, 0.1, 0.1])
TAU_FLOOR = 1.0             # N / Nm


def assert_close(a, b, floor, rtol=RTOL_F32, what=''):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    tol = rtol * np.maximum(np.abs(b), floor)
    err = np.abs(a - b)
    bad = ~(err <= tol)
    if bad.any():
        i = np.unravel_index(np.argmax(err / tol), err.shape)
        raise AssertionError('%s: %d/%d outside tolerance; worst at %s: got %r want %r (err %.3e tol %.3e)'
                             % (what, bad.sum(), bad.size, i, a[i], b[i], err[i], tol[i]))
