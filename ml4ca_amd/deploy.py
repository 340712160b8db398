"""Deployment-side adapters of the RL allocator: what the thesis' ROS node wraps around the trained actor
(src/rl/ROS/rl_allocator/src/rl_allocator.py, utils.py:88-115, errorFrame.py), without ROS.

Host logic (NumPy, float64), not on the accelerated path: SURVEY section 8(f) rank 4.  It exists so that a policy
trained on the batched env can be driven exactly the way the vessel's node drives it - state assembly with the radian
wrap the node uses (unlike the training env, quirk Q1), network order -> ROS thruster order, default commands for the
thrusters a variant does not control, the optional body-frame integral action, and the message fields published.

  network order  [n_bow, n_port, n_star, (a_bow,) a_port, a_star]       (customEnv.py:47-61)
  ROS order      [n_port, n_star, n_bow, a_port, a_star, a_bow]         (rl_allocator.py:92-106)
"""
import numpy as np

ROS_ORDER = ('n_port', 'n_star', 'n_bow', 'a_port', 'a_star', 'a_bow')

# rl_allocator.py:92-106
ACT_BND = {'simple': [100.0] * 3, 'limited': [100.0] * 3 + [np.pi / 2] * 2, 'final': [100.0] * 3 + [np.pi] * 2,
           'full': [100.0] * 3 + [np.pi] * 3}
ACT_MAP = {'simple': {0: 2, 1: 0, 2: 1}, 'limited': {0: 2, 1: 0, 2: 1, 3: 3, 4: 4}, 'final': {0: 2, 1: 0, 2: 1, 3: 3, 4: 4},
           'full': {0: 2, 1: 0, 2: 1, 3: 5, 4: 3, 5: 4}}
ACT_DEF = {'simple': [0, 0, 0, np.pi / 2, -3 * np.pi / 4, 3 * np.pi / 4], 'limited': [0, 0, 0, np.pi / 2, 0, 0],
           'final': [0, 0, 0, np.pi / 2, 0, 0], 'full': [0] * 6}


def wrap_angle(angle, deg=False):
    """errorFrame.py:14-25 of the ROS package: the node's wrap defaults to RADIANS (the training env's to degrees)."""
    ref = 180.0 if deg else np.pi
    return np.mod(np.asarray(angle, np.float64) + ref, 2 * ref) - ref


def shortest_path(a_prev, a, deg=False):
    """rl_allocator.py:284-299: signed shortest angular distance from a_prev to a."""
    ref = 180.0 if deg else np.pi
    shortest = np.mod(np.mod(a - a_prev, 2 * ref) + 2 * ref, 2 * ref)
    return shortest - 2 * ref if shortest > ref else shortest


def to_ros_order(action, variant='final', cont_ang=True):
    """Raw network output(s) [..., act_dim] -> thruster commands in the ROS order [..., 6] (percent, radians):
    handle_continuous_angles (rl_allocator.py:275-283), scale_and_clip (:222-226), defaults and maps (:228-250)."""
    a = np.asarray(action, np.float64)
    lead = a.shape[:-1]
    a = a.reshape(-1, a.shape[-1])
    bnd = np.asarray(ACT_BND[variant], np.float64)
    if variant == 'final' and cont_ang:
        a = np.concatenate([a[:, 0:3], np.arctan2(a[:, 3:4], a[:, 4:5]) / bnd[-1], np.arctan2(a[:, 5:6], a[:, 6:7]) / bnd[-1]], 1)
    a = np.clip(a * bnd, -bnd, bnd)
    out = np.zeros((a.shape[0], 6))
    for i, default in enumerate(ACT_DEF[variant]):          # (1) defaults through the FULL map
        out[:, ACT_MAP['full'][i]] = default
    for i in range(a.shape[1]):                             # (2) the variant's own commands
        out[:, ACT_MAP[variant][i]] = a[:, i]
    return out.reshape(lead + (6,))


def publishable(u, simulation=True):
    """utils.py:88-115: the fields of the three messages the node publishes for one command vector in ROS order."""
    u = np.asarray(u, np.float64)
    msg = {'pod_angle.port': float(np.rad2deg(u[3])), 'pod_angle.star': float(np.rad2deg(u[4])),
           'stern.port_effort': float(u[0]), 'stern.star_effort': float(u[1]), 'bow.lin_act_bow': 2}
    if simulation:
        msg['bow.position_bow'] = int(np.rad2deg(u[5]))
        msg['bow.throttle_bow'] = float(u[2])
    else:
        msg['bow.position_bow'] = 45                                           # utils.py:112
        msg['bow.throttle_bow'] = float(np.clip(float(u[2]) * 2.5, -100.0, 100.0))   # utils.py:113
    return msg


class BodyFrameIntegrator(object):
    """rl_allocator.py:252-273: integral action added to the body-frame error once the vessel has dwelt near the
    setpoint.  Leaving the 5 m / 140 deg box resets it; after 5 s inside, it integrates 0.05 * error * step, clipped to
    [0.5 m, 1 m, pi/32 rad].  `now` replaces the node's wall clock."""
    GAIN = np.array([0.05, 0.05, 0.05])
    BOUND = np.array([0.5, 1.0, np.pi / 32])

    def __init__(self, now=0.0):
        self.value = np.zeros(3)
        self.time_arrival = float(now)

    def update(self, err, step, now):
        err = np.asarray(err, np.float64)
        if abs(err[0]) > 5.0 or abs(err[1]) > 5.0 or abs(err[2]) > np.deg2rad(140):
            self.value = np.zeros(3)
            self.time_arrival = float(now)
        elif (now - self.time_arrival) > 5.0:
            self.value = np.clip(self.value + step * self.GAIN * err, -self.BOUND, self.BOUND)
        return err + self.value


class RLAllocatorNode(object):
    """The node's callbacks as plain methods (rl_allocator.py:168-220).  `actor(state[9]) -> action[act_dim]` is any
    callable: ActorCritic.forward_ref on the CPU, or policy_forward on the GPU for one env."""

    def __init__(self, actor, variant='final', cont_ang=True, integrator=False, simulation=True, now=0.0):
        if variant == 'simple':
            raise ValueError('no simple environment was trained with the extended state vector')   # rl_allocator.py:126
        self.actor, self.variant, self.cont_ang = actor, variant, cont_ang
        self.simulation = simulation
        self.state = np.zeros(9)
        self.velocities = np.zeros(3)
        self.prev_thrust_state = np.zeros(6)
        self.pos = [0.0, 0.0, 0.0]
        self.ref = [0.0, 0.0, 0.0]
        self.integrator = BodyFrameIntegrator(now) if integrator else None
        self.time_prev = float(now)
        self.h = 0.0

    def _error(self):
        """errorFrame.py transform: rotation by the wrapped heading, yaw error wrapped in radians"""
        e = [a - b for a, b in zip(self.pos, self.ref)]
        rot = float(wrap_angle(self.pos[2]))
        c, s = np.cos(rot), np.sin(rot)
        return np.array([c * e[0] + s * e[1], -s * e[0] + c * e[1], float(wrap_angle(e[2]))])

    def _error_states(self, step, now):
        err = self._error()
        if self.integrator is None:
            return err
        return self.integrator.update(err, step, now)

    def on_eta(self, north, east, heading_deg, now):
        """eta_obs_callback (rl_allocator.py:168-178); note the node integrates here too, with its default step 0.1"""
        self.pos = [float(north), float(east), float(wrap_angle(np.deg2rad(heading_deg)))]
        self.state[0:3] = self._error_states(0.1, now)

    def on_nu(self, u, v, r):
        """nu_obs_callback (:180-187)"""
        self.velocities = np.array([u, v, r], np.float64)
        self.state[3:6] = self.velocities

    def on_reference(self, north, east, heading_deg, now):
        """state_desired_callback (:189-220): returns (u in ROS order, message fields)"""
        self.h = now - self.time_prev
        self.time_prev = now
        self.ref = [float(north), float(east), float(np.deg2rad(heading_deg))]
        self.state[0:3] = self._error_states(self.h, now)
        self.state[3:6] = self.velocities
        u = to_ros_order(np.asarray(self.actor(self.state.copy()), np.float64), self.variant, self.cont_ang)
        self.prev_thrust_state = u.copy()
        self.state[-3:] = np.array([u[2], u[0], u[1]]) / 100.0          # back to network order (:218)
        return u, publishable(u, self.simulation)
