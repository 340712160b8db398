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


class QPAllocator(object):
    """CPU restatement of the reference's nonlinear "QP" thrust allocator (SLSQP), the baseline the thesis compares the
    RL allocator with: QPTA.solve_QP and the command conversion of tau_controller_callback_func
    (src/qp/ROS/qp_allocator/src/qp_allocator.py:108-234, 264-320).  ROS order: port, starboard, bow.

    Decision vector x = [F_port, F_star, F_bow, a_port, a_star, s1, s2, s3]; minimise 0.5 z'Qz with
    z = [s, |F|^1.5, |a - a_prev| (weight 0.25), |F - F_prev| (weight 0.25)] subject to B(a) F - tau - s = 0, force-rate and
    azimuth-rate limits per 0.2 s step, force bounds, |a| <= 2 pi, |s| <= s_bnd.  The bow azimuth is fixed at pi/2.
    A failed solve keeps the previous thruster state (:267-269).  ``retry`` reproduces the node's loop that widens the
    slack bound by 1 N until SLSQP succeeds (:209-226; it only runs when wall-clock time has elapsed, and with SciPy
    versions whose ``success`` is a Python bool)."""

    def __init__(self, simulation=False, retry=False, dt=0.20):
        self.dt = dt
        self.simulation = simulation
        self.retry = retry
        self.max_force_rate = [10.0 / 2.0, 10.0 / 2.0, 4.0 / 2.0]                           # :57
        self.max_rotational_rate = [np.pi / 6.0 / 2.0, np.pi / 6.0 / 2.0, np.pi / 32.0 / 2.0]  # :58
        self.bow_angle_fixed = np.pi / 2                                                    # :65
        self.previous_thruster_state = [0, 0, 0, 0, 0, self.bow_angle_fixed]                # :66  [N, N, N, rad, rad, rad]

    def solve(self, tau_d, s_bnd=1.0, x0=None):
        from scipy.optimize import minimize
        tau = np.asarray(tau_d, dtype=np.float64).reshape(3)
        s_t = self.previous_thruster_state
        lx, ly = LX, LY

        # Cost (:116-150, :203): J = 1/2 sum_k w_k z_k^2 over the eleven terms
        #   z = [ slack (3) | |F|^1.5 (3) | |d alpha| port, star | |d F| (3) ],  w = [1 1 1 | 1 1 1 | 1/4 1/4 | 1/4 1/4 1/4]
        # i.e. slack^2 + |F|^3 (power) + quarter-weighted squared changes of azimuth and force against the state in force.
        prev_F, prev_alpha = np.asarray(s_t[0:3], dtype=np.float64), np.asarray(s_t[3:5], dtype=np.float64)
        weight = np.concatenate([np.ones(6), np.full(5, 0.25)])

        def objective(x):
            F, alpha, slack = x[0:3], x[3:5], x[5:8]
            z = np.concatenate([slack, np.abs(F) ** 1.5, np.abs(alpha - prev_alpha), np.abs(F - prev_F)])
            return 0.5 * np.dot(weight * z, z)

        hp = np.pi / 2
        cons = [                                                                            # :156-192
            {'type': 'eq', 'fun': lambda x: np.cos(x[3]) * x[0] + np.cos(x[4]) * x[1] + np.cos(hp) * x[2] - x[5] - float(tau[0])},
            {'type': 'eq', 'fun': lambda x: np.sin(x[3]) * x[0] + np.sin(x[4]) * x[1] + np.sin(hp) * x[2] - x[6] - float(tau[1])},
            {'type': 'eq', 'fun': lambda x: (lx[0] * np.sin(x[3]) - ly[0] * np.cos(x[3])) * x[0] +
                (lx[1] * np.sin(x[4]) - ly[1] * np.cos(x[4])) * x[1] + (lx[2] * np.sin(hp) - ly[2] * np.cos(hp)) * x[2] - x[7] - float(tau[2])},
        ]
        for i in range(3):
            cons.append({'type': 'ineq', 'fun': lambda x, i=i: self.max_force_rate[i] - (x[i] - s_t[i])})
            cons.append({'type': 'ineq', 'fun': lambda x, i=i: self.max_force_rate[i] + (x[i] - s_t[i])})
        for i in range(2):
            cons.append({'type': 'ineq', 'fun': lambda x, i=i: self.max_rotational_rate[i] + (x[3 + i] - s_t[3 + i])})
            cons.append({'type': 'ineq', 'fun': lambda x, i=i: self.max_rotational_rate[i] - (x[3 + i] - s_t[3 + i])})

        def bounds(sb):                                                                     # :198-200
            return ((-F_MAX[0], F_MAX[0]), (-F_MAX[1], F_MAX[1]), (-F_MAX[2], F_MAX[2]), (-2 * np.pi, 2 * np.pi),
                    (-2 * np.pi, 2 * np.pi), (-sb, sb), (-sb, sb), (-sb, sb))

        if x0 is None:
            x0 = np.concatenate([prev_F, prev_alpha, np.zeros(3)])                          # start at the state in force, no slack (:203)
        sol = minimize(objective, x0, method='SLSQP', bounds=bounds(s_bnd), constraints=cons)   # :206
        tries = 0
        while self.retry and not bool(sol.success) and tries < 100:                         # :209-226
            s_bnd += 1.0
            x0 = np.concatenate([prev_F, prev_alpha, sol.x[5:8]])                           # keep the slack reached (:212-222)
            sol = minimize(objective, x0, method='SLSQP', bounds=bounds(s_bnd), constraints=cons)
            tries += 1
        x = sol.x
        x[np.where(np.abs(x) < 0.01)] = 0.0                                                 # :232
        return x, bool(sol.success)

    def allocate(self, tau_d):
        """One 5 Hz allocation step (:264-320): returns (n_pct [port, star, bow-as-published], angles_deg [port, star])
        and updates the carried thruster state."""
        x, ok = self.solve(tau_d)
        sol = x if ok else self.previous_thruster_state                                     # :267-269
        F = np.array([sol[0], sol[1], sol[2]], dtype=np.float64)
        alpha = np.array([sol[3], sol[4], self.bow_angle_fixed], dtype=np.float64)
        alpha = np.mod(alpha + np.pi, 2 * np.pi) - np.pi                                    # mapToPi :101-106,277
        n = force_to_percent(F, K_SIM)                                                      # :287-288
        bow = float(n[2]) if self.simulation else float(np.clip(n[2] * 2.5, -100.0, 100.0))  # :303-308
        self.previous_thruster_state = [float(F[0]), float(F[1]), float(F[2]), float(alpha[0]), float(alpha[1]),
                                        float(alpha[2])]                                    # :316-318
        return np.array([n[0], n[1], bow]), np.degrees(alpha[:2]), ok
