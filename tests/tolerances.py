"""Parity tolerances (north star: outputs match the CPU reference to 1e-5 relative in fp32).

    |a - b| <= RTOL * max(|b|, floor)

Where the floors come from.  Two correct fp32 evaluations of the same formula differ by the roundings on the way, and a rounding
is half a unit in the last place of the number being rounded - which is an INPUT-sized number when the result is a cancelled
difference (a body-frame error of millimetres is the difference of products of metre-sized coordinates; a reward near zero is a
sum of O(1) parts).  So for a quantity computed from inputs of magnitude S through about K roundings the absolute error is up to
K * ulp32(S), whatever the size of the result, and a relative test needs the floor

    floor = K * ulp32(S) / RTOL

below which the tolerance stops shrinking.  `DERIVATION` holds (S, K) for every quantity, the constants below are
`derive_floor(S, K)` rounded UP to one digit, and tests/test_oracle_golden.py::test_tolerance_floors_follow_from_the_ulp_argument
recomputes them, checks on 20 000 random transitions that the fp32 oracle stays inside the tolerance against the float64 oracle
with a factor two to spare (the other fp32 implementation - the HIP kernel - gets the other half) and that no floor is more than
an order of magnitude looser than that measurement needs.  SURVEY section 7 proposes one blanket floor of 1e-2; the bench line and
tests/test_gpu_parity.py::test_error_against_the_survey_floor report the error under that floor too (it is met by every
quantity that is not a cancelled difference of metre-sized inputs, and cannot be met by those in fp32 by ANY implementation:
the float32 oracle itself misses it against the float64 oracle, see the test).
"""
import numpy as np

RTOL_F32 = 1e-5
ATOL_F64 = 1e-11            # float64 oracle vs float64 reference: identical formulae, libm noise only
SURVEY_FLOOR = 1e-2         # |a - b| <= 1e-5 * max(|b|, 1e-2)


def ulp32(x):
    """unit in the last place of the float32 binade that holds |x|"""
    return float(np.spacing(np.float32(abs(x))))


def derive_floor(S, K):
    return K * ulp32(S) / RTOL_F32


def _round_up_1digit(x):
    e = 10.0 ** np.floor(np.log10(x))
    return float(np.ceil(x / e - 1e-9) * e)


# quantity -> (S: magnitude of the largest input, K: roundings on the chain), with the reason for S
DERIVATION = {
    # x~ = cos(psi) (N - N_r) + sin(psi) (E - E_r) (errorFrame.py:25-32): |N|, |E| <= 8 (customEnv.py:26) and a setpoint up to 8 away
    # -> differences and products in the binade of 8..16; the plant adds 20 position increments first.
    'x': (8.0, 10), 'y': (8.0, 10),
    # psi~ = psi - psi_r, |psi| up to pi: 20 increments of psi, one subtraction, the wrap
    'psi': (np.pi, 4),
    # u, v, r: three COUPLED equations (Coriolis m11 u r, cross-flow Y_ur u r, N_uv u v), so the input scale of each is the largest
    # of them, surge at up to 1.4 m/s (customEnv.py:26); 20 sub-steps whose increments are ~1e-2 of the state each, i.e. the
    # roundings that matter are those of the accumulations: one per sub-step, growing like sqrt(20) + the force evaluation
    'u': (1.4, 8), 'v': (1.4, 8), 'r': (1.4, 8),
    # previous thrust / 100: one division of a clipped command in [-100, 100] (the constant below, 0.01, is tighter than this gives)
    'n': (1.0, 1),
    # reward = r_vel + r_pos + r_thr + r_dn + r_dalpha (customEnv.py:263), max 3.5; r_pos is a Gaussian of the pose error with slope
    # 2 exp(-1/2) = 1.2 per metre and 14 per radian, so it inherits the x~ / psi~ errors above: 1.2 * 10 ulp(8) + 14 * 4 ulp(pi)
    # = 2.5e-5 worst case = ~100 ulp(3.5); the measured worst case is a fifth of that (the steep part of the Gaussian is a small
    # part of the state space), K = 40
    'reward': (3.5, 40),
    # parts: vel (norm of nu, inherits nu: 1.4 -> 8), pos (as reward), thrust penalty (sum of |n| / 100: ~1), derivative penalties (<= 1.5)
    'part_vel': (1.4, 8), 'part_pos': (3.5, 40), 'part_thr': (1.0, 8), 'part_der': (1.5, 8),
    'thrust_cmd': (100.0, 1),      # percent, a clipped product
    'angle_cmd': (np.pi, 4),       # rad: atan2 of the two heads / pi * pi, or a wrap
    'eta_NE': (8.0, 10), 'eta_psi': (np.pi, 4),
    'tau': (20.5, 4),              # N / Nm: sums of three K n |n| <= 20.5 N (qp_allocator.py:51-55) times sin / cos and lever arms ~1
}

# obs = [x~, y~, psi~, u, v, r, n_bow/100, n_port/100, n_star/100]
OBS_FLOOR = np.array([1.0, 1.0, 0.1, 0.1, 0.1, 0.1, 0.01, 0.01, 0.01])
OBS_KEYS = ('x', 'y', 'psi', 'u', 'v', 'r', 'n', 'n', 'n')
REWARD_FLOOR = 1.0
PARTS_FLOOR = np.array([0.1, 1.0, 0.1, 0.1])
PARTS_KEYS = ('part_vel', 'part_pos', 'part_thr', 'part_der')
THRUST_FLOOR = 0.8          # percent, range +-100
ANGLE_FLOOR = 0.1           # rad, range +-pi
ETA_FLOOR = np.array([1.0, 1.0, 0.1])
NU_FLOOR = np.array([0.1, 0.1, 0.1])
TAU_FLOOR = 0.8             # N / Nm


def derived_floors():
    """{quantity: floor} recomputed from DERIVATION, rounded up to one digit - what the constants above must equal"""
    return {k: _round_up_1digit(derive_floor(S, K)) for k, (S, K) in DERIVATION.items()}


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


def done_agrees(gdone, odone, oobs, bounds):
    """done bits must be equal, except where an observation sits within fp32 rounding of a termination
    bound (strict > on a value 1 ulp either side may legitimately flip).  Returns the mask of envs that agree;
    raises if a disagreement is NOT explained by a bound."""
    gdone, odone = np.asarray(gdone), np.asarray(odone)
    same = gdone == odone
    if not same.all():
        b = np.asarray(bounds, np.float64)
        margin = np.abs(np.abs(np.asarray(oobs)[:, :6].astype(np.float64)) - b[None, :]) / b[None, :]
        borderline = margin.min(1) < 2e-6
        bad = ~same & ~borderline
        assert not bad.any(), 'done bits differ away from any bound at envs %s' % np.nonzero(bad)[0][:8]
    return same


class ErrorLedger(object):
    """Per-quantity record of |a - b| between two evaluations of env.step (a: the HIP kernel or the fp32 oracle, b: the oracle it
    is checked against), under SURVEY's floor, under the test floor, and in ulps of the largest input of the quantity."""

    QUANT = [('obs.x~', 'x'), ('obs.y~', 'y'), ('obs.psi~', 'psi'), ('obs.u', 'u'), ('obs.v', 'v'), ('obs.r', 'r'), ('obs.thrust/100', 'n')]

    def __init__(self):
        self.rec = {}

    def _add(self, name, key, a, b, floor, scale):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        err = np.abs(a - b)
        r = self.rec.setdefault(name, {'rel_err_floor_1e-2': 0.0, 'rel_err_test_floor': 0.0, 'test_floor': float(floor), 'max_abs_err': 0.0,
                                       'err_in_ulps_of_largest_input': 0.0, 'derivation_S_K': list(DERIVATION[key])})
        r['rel_err_floor_1e-2'] = max(r['rel_err_floor_1e-2'], float((err / np.maximum(np.abs(b), SURVEY_FLOOR)).max()))
        r['rel_err_test_floor'] = max(r['rel_err_test_floor'], float((err / np.maximum(np.abs(b), floor)).max()))
        r['max_abs_err'] = max(r['max_abs_err'], float(err.max()))
        # in ulps of the largest input ACTUALLY present (never below a quarter of the nominal S: a heading of 1e-3 rad is still
        # advanced by yaw-rate increments of 1e-2), and in ulps of the nominal S itself - the unit K of DERIVATION is counted in
        sc = np.maximum(np.abs(np.asarray(scale, np.float64)), 0.25 * DERIVATION[key][0])
        ulps = err / np.spacing(sc.astype(np.float32)).astype(np.float64)
        r['err_in_ulps_of_largest_input'] = max(r['err_in_ulps_of_largest_input'], float(ulps.max()))
        r['err_in_ulps_of_S'] = max(r.get('err_in_ulps_of_S', 0.0), float(err.max()) / ulp32(DERIVATION[key][0]))

    def add_step(self, obs_a, rew_a, parts_a, state_a, obs_b, rew_b, parts_b, state_b, pre_state):
        """obs [n, 9], rew [n], parts [n, 4], state / pre_state canonical [15, n] (post / pre step)"""
        post = np.asarray(state_b, np.float64)
        pre = np.asarray(pre_state, np.float64)
        # largest input per env: positions and setpoints for the body-frame errors, headings for psi~, the velocity triple for nu
        s_pos = np.maximum.reduce([np.abs(post[0]), np.abs(post[1]), np.abs(pre[6]), np.abs(pre[7]), np.abs(post[0] - pre[6]), np.abs(post[1] - pre[7])])
        s_psi = np.maximum(np.abs(post[2]), np.abs(pre[8]))
        s_nu = np.maximum.reduce([np.abs(pre[3]), np.abs(pre[4]), np.abs(pre[5]), np.abs(post[3]), np.abs(post[4]), np.abs(post[5])])
        scales = [s_pos, s_pos, s_psi, s_nu, s_nu, s_nu, np.ones_like(s_nu)]
        for k, (name, key) in enumerate(self.QUANT):
            if k < 6:
                self._add(name, key, obs_a[:, k], obs_b[:, k], OBS_FLOOR[k], scales[k])
            else:
                self._add(name, key, obs_a[:, 6:9], obs_b[:, 6:9], OBS_FLOOR[6], np.ones_like(obs_b[:, 6:9]))
        s_rew = np.maximum(np.abs(np.asarray(parts_b, np.float64)).max(1), 1.0)
        self._add('reward', 'reward', rew_a, rew_b, REWARD_FLOOR, s_rew)
        for k, key in enumerate(PARTS_KEYS):
            self._add('reward.' + key[5:], key, parts_a[:, k], parts_b[:, k], PARTS_FLOOR[k], np.maximum(np.abs(np.asarray(parts_b)[:, k]), DERIVATION[key][0] * 0.5))
        self._add('state.thrust_cmd', 'thrust_cmd', state_a[9:12], state_b[9:12], THRUST_FLOOR, np.maximum(np.abs(post[9:12]), 1.0))
        self._add('state.N,E', 'eta_NE', state_a[0:2], state_b[0:2], ETA_FLOOR[0], np.broadcast_to(s_pos, post[0:2].shape))

    def report(self):
        return self.rec
