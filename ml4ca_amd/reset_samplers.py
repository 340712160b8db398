"""Host-side reset samplers of the single-env adapters.

Mirror of the reference's reset helpers (src/rl/windows_workspace/specific/misc/simtools.py:81-123)
for the testing-mode starts, which the reference draws on the host with numpy's global RNG.  The
batched training reset does NOT use these: it samples on device with Philox (dpenv_kernels.hip).
"""
import math

import numpy as np

# simtools.py:94-95: six evaluation starts on the 5 m circle, bearing (compass) and initial heading [deg]
FIXED_BEARINGS = (0.0, math.pi / 4, math.pi / 2, math.pi, 5 * math.pi / 4, 3 * math.pi / 2)
FIXED_HEADINGS_DEG = (0.0, 0.0, -15.0, 15.0, 0.0, -15.0)


def _uniform3(limits):
    lim = np.asarray(limits, dtype=np.float64)
    assert lim.shape == (3,)
    return tuple(np.random.uniform(-lim, lim))


def get_pose_on_state_space(bounds=(5, 5, math.pi / 18), fraction=1.0):
    """simtools.py:109-115: N, E, psi ~ U(+-fraction * bounds)."""
    return _uniform3(np.asarray(bounds, dtype=np.float64) * fraction)


def get_vel_on_state_space(bounds=(2.2, 0.35, 0.60), fraction=1.0):
    """simtools.py:117-123: u, v, r ~ U(+-fraction * bounds)."""
    return _uniform3(np.asarray(bounds, dtype=np.float64) * fraction)


def get_random_pose_on_radius(r=5, angle=5 * math.pi / 180):
    """simtools.py:81-88: random point on the radius-r circle, heading ~ U(+-angle)."""
    theta = np.random.random() * 2 * math.pi
    return r * math.sin(theta), r * math.cos(theta), np.random.uniform(-angle, angle)


def get_fixed_pose_on_radius(n, r=5, angle=5 * math.pi / 180):
    """simtools.py:91-107: n-th fixed evaluation start (n wraps modulo 6 like the reference)."""
    n = n % len(FIXED_BEARINGS)
    ang = math.pi / 2 - FIXED_BEARINGS[n]      # compass bearing -> unit-circle angle
    return r * math.sin(ang), r * math.cos(ang), FIXED_HEADINGS_DEG[n] * math.pi / 180
