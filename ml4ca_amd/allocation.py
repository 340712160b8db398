"""Pseudo-inverse thrust allocation for one tau = [Fx, Fy, Mz] (BASELINE.json config 1; CPU plumbing).

The reference tree has no pseudo-inverse allocator of its own (src/qp is an SLSQP solver and the "IPI"
baseline of the thesis ran in a DNV GL ROS node that is not in the repository - SURVEY section 0.2), so this
is built from what IS in the reference: the effectiveness matrix B(alpha) (src/sl/SupervisedTau.py:42-52,
src/qp/ROS/qp_allocator/src/qp_allocator.py:156-158), the thrust law F = K n|n| with the simulator's
constants (qp_allocator.py:51-55) and the force -> percent map n = sgn(F/K) sqrt(|F/K|) (qp_allocator.py:287-288).
Thruster order here is the ROS/QP order: port, starboard, bow (qp_allocator.py:69-70).  float64, NumPy.
"""
import numpy as np

LX = np.array([-1.12, -1.12, 1.08])            # qp_allocator.py:69
LY = np.array([-0.15, 0.15, 0.0])              # qp_allocator.py:70
K_SIM = np.array([0.00205, 0.00205, 0.0009])   # qp_allocator.py:54
F_MAX = np.array([20.5, 20.5, 9.0])            # qp_allocator.py:52
ROS_TO_ENV = [2, 0, 1]                         # (port, star, bow) -> env order (bow, port, star), customEnv.py:48-50


def effectiveness(alpha, lx=LX, ly=LY):
    """B(alpha), SupervisedTau.py:49-52."""
    a = np.asarray(alpha, dtype=np.float64)
    c, s = np.cos(a), np.sin(a)
    return np.stack([c, s, lx * s - ly * c])


def force_to_percent(F, K=K_SIM):
    """qp_allocator.py:287-288."""
    fk = np.asarray(F, dtype=np.float64) / K
    return np.sign(fk) * np.sqrt(np.abs(fk))


def percent_to_force(n, K=K_SIM):
    n = np.asarray(n, dtype=np.float64)
    return K * n * np.abs(n)


def pinv_allocate(tau, alpha, K=K_SIM, saturate=False):
    """Minimum-norm forces F = B(alpha)^+ tau for fixed azimuths, and the percent commands that produce them.
    Returns (n_pct[3], F[3]); with saturate=True forces are clipped to +-F_MAX first (then B F != tau)."""
    B = effectiveness(alpha)
    F = np.linalg.pinv(B) @ np.asarray(tau, dtype=np.float64)
    if saturate:
        F = np.clip(F, -F_MAX, F_MAX)
    return force_to_percent(F, K), F
