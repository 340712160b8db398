"""Gym-style host mirror of the reference environment over libdpenv.so.

Mirrors the reference's environment interface (paths relative to the reference root,
ENV = src/rl/windows_workspace/specific/customEnv.py):

  * ``BatchedRevoltEnv``  - N environments stepped by one HIP kernel launch; ``reset()`` / ``step()``
    take and return ROCm torch tensors.  This is what a vectorised PPO rollout consumes.
  * ``Revolt`` / ``RevoltSimple`` / ``RevoltLimited`` / ``RevoltFinal`` - single-env adapters with
    the reference constructors' signatures (ENV:22-32,331,355,377), NumPy in / NumPy out, and the
    attribute set the reference's PPO loop and evaluation harness read (ppo.py:202-206,
    train.py:70-73, test_policy.py:114-178) so that a Spinning-Up style loop runs unchanged.
    ``digitwin`` is accepted and ignored: the py4j/Cybersea bridge (digitwin.py) is replaced by
    the kernel's own plant.

PyTorch is plumbing here (device memory, streams); all env arithmetic runs in libdpenv.so.
"""
import ctypes as C
import math
import weakref

import numpy as np

from . import _lib
from . import reset_samplers as simtools

_VARIANTS = {'full': _lib.FULL, 'simple': _lib.SIMPLE, 'limited': _lib.LIMITED, 'final': _lib.FINAL}
_NAMES = {'full': 'full', 'simple': 'revoltsimple', 'limited': 'revoltlimited', 'final': 'revoltfinal'}   # ENV:37,335,360,385


def variant_constants(variant, cont_ang=False):
    """Per-variant constants exactly as the reference constructors leave them (ENV:55-65,337-349,361-371,386-399)."""
    pi = math.pi
    if variant == 'full':
        return dict(num_actions=6, real_action_bounds=[100] * 3 + [pi] * 3,
                    real_ss_bounds=[8.0, 8.0, pi / 2, 1.4, 0.30, 0.52],
                    default_actions={0: 0, 1: 0, 2: 0, 3: 0, 4: 0, 5: 0},
                    valid_action_indices=[0, 1, 2, 3, 4, 5],
                    act_2_act_map={0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5},
                    act_2_act_map_inv={0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5})
    if variant == 'simple':
        return dict(num_actions=3, real_action_bounds=[100] * 3,
                    real_ss_bounds=[8.0, 8.0, pi / 2, 1.75, 0.30, 0.51],
                    default_actions={0: 0, 1: 0, 2: 0, 3: pi / 2, 4: -3 * pi / 4, 5: 3 * pi / 4},
                    valid_action_indices=[0, 1, 2],
                    act_2_act_map={0: 0, 1: 1, 2: 2}, act_2_act_map_inv={0: 0, 1: 1, 2: 2})
    if variant in ('limited', 'final'):
        ab = pi / 2 if variant == 'limited' else pi
        return dict(num_actions=7 if (variant == 'final' and cont_ang) else 5,
                    real_action_bounds=[100] * 3 + [ab] * 2,
                    real_ss_bounds=[8.0, 8.0, 45 * pi / 180, 1.4, 0.30, 0.52],
                    default_actions={0: 0, 1: 0, 2: 0, 3: pi / 2, 4: 0, 5: 0},
                    valid_action_indices=[0, 1, 2, 4, 5],
                    act_2_act_map={0: 0, 1: 1, 2: 2, 4: 3, 5: 4},
                    act_2_act_map_inv={0: 0, 1: 1, 2: 2, 3: 4, 4: 5})
    raise ValueError('unknown variant %r' % (variant,))


class Box(object):
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency): what ppo.py:202-206 reads."""

    def __init__(self, low, high, dtype=np.float64):
        self.low, self.high, self.shape, self.dtype = low, high, low.shape, dtype

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


def _torch():
    import torch
    return torch


class BatchedRevoltEnv(object):
    """N independent ReVolt DP environments on one MI355X (one wavefront lane per environment).

    step()/reset() semantics per environment are Revolt.step / Revolt.reset (ENV:92-194).
    Differences forced by batching, all explicit:
      * ``done`` is a uint8 tensor of DPENV_DONE_* bits (1 terminal = the reference's ``d``,
        2 time limit = ppo.py:304's ``traj_len == max_ep_len``, 4 non-finite state).
      * with ``auto_reset=True`` finished envs are re-sampled inside the same launch (the batched
        form of ppo.py:305-322) and ``obs`` holds the new episode's first observation; the terminal
        observation is available through ``final_obs``.
      * reset() takes explicit per-env init / ref tensors instead of **init dicts.
    """

    def __init__(self, n_envs, variant='final', extended_state=True, cont_ang=True, device='cuda:0',
                 testing=False, realtime=False, max_ep_len=800, auto_reset=False, terminate=True,
                 wrap_mode='reference', seed=0, env_id_base=0, obs_dtype='float32', current=False,
                 vessel_params=None, layout='aos', reset_fraction=0.8, time_limit=True, hold_plant=False,
                 current_drift=False, current_tau=100.0, current_sigma_v=0.02, current_sigma_beta=5.0 * math.pi / 180.0,
                 n_steps=None, reset_acts=False, step_one_wave=False, per_env_lds=False):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError('BatchedRevoltEnv needs a ROCm device: the env.step path is a HIP kernel and has no CPU fallback')
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.variant = variant
        k = variant_constants(variant, cont_ang)
        self.name = _NAMES[variant]
        self.extended_state = bool(extended_state)
        self.cont_ang = bool(cont_ang) and variant == 'final'
        self.testing = bool(testing)
        self.reset_actions = bool(reset_acts)                        # ENV:30,179-188
        self.n_envs = int(n_envs)
        self.num_actions = k['num_actions']
        self.num_states = 9 if extended_state else 6                 # ENV:44
        self.real_action_bounds = k['real_action_bounds']
        self.real_ss_bounds = k['real_ss_bounds']
        self.default_actions = k['default_actions']
        self.valid_action_indices = k['valid_action_indices']
        self.act_2_act_map = k['act_2_act_map']
        self.act_2_act_map_inv = k['act_2_act_map_inv']
        self.n_steps = 1 if (testing and realtime) else 20           # ENV:79-80
        if n_steps is not None:                                      # any other agent rate: n_steps plant sub-steps of 10 ms
            self.n_steps = int(n_steps)
        self.dt = 0.01 * self.n_steps                                # ENV:81
        self.max_ep_len = int(max_ep_len * 10.0 / self.n_steps)      # ENV:83
        self.vel_rew_coeffs = [0.5, 0.5, 1.0]                        # ENV:78
        self.covar = np.array([[1.0, 0.0], [0.0, 25.0]])             # ENV:86-87
        self.observation_space = Box(-np.ones(self.num_states), np.ones(self.num_states))   # ENV:55-64
        self.action_space = Box(-np.ones(self.num_actions), np.ones(self.num_actions))

        cfg = _lib.default_config()
        cfg.n_envs = self.n_envs
        cfg.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cfg.variant = _VARIANTS[variant]
        cfg.extended_state = int(self.extended_state)
        cfg.cont_ang = int(self.cont_ang)
        cfg.n_substeps = self.n_steps
        cfg.substep_dt = 0.01
        cfg.wrap_mode = {'reference': _lib.WRAP_REFERENCE, 'radians': _lib.WRAP_RADIANS}[wrap_mode]
        cfg.terminate = int(bool(terminate))
        cfg.max_ep_len = self.max_ep_len if time_limit else 0
        cfg.auto_reset = int(bool(auto_reset))
        lay = {'aos': _lib.AOS, 'soa': _lib.SOA}[layout]
        cfg.action_layout = lay
        cfg.obs_layout = lay
        self.obs_torch_dtype = {'float32': torch.float32, 'bfloat16': torch.bfloat16}[str(obs_dtype).replace('torch.', '')]
        cfg.obs_dtype = _lib.BF16 if self.obs_torch_dtype == torch.bfloat16 else _lib.F32
        cfg.current_enabled = int(bool(current))
        cfg.seed = int(seed)
        cfg.env_id_base = int(env_id_base)
        cfg.reset_fraction = float(reset_fraction)
        cfg.hold_plant = int(bool(hold_plant))
        cfg.current_drift = int(bool(current_drift))
        cfg.current_tau = float(current_tau)
        cfg.current_sigma_v = float(current_sigma_v)
        cfg.current_sigma_beta = float(current_sigma_beta)
        cfg.reset_acts = int(self.reset_actions)                     # drawn inside the reset kernels, ENV:179-188
        cfg.step_one_wave = int(bool(step_one_wave))                 # A/B switch: dpenv_step's re-draw on the env wave (dpenv.h)
        cfg.per_env_lds = int(bool(per_env_lds))                     # A/B switch: per-env blocks through an LDS image in dpenv_step (dpenv.h)
        self.cfg = cfg
        self.layout = layout
        self.auto_reset = bool(auto_reset)

        n_classes = 1
        vp = None
        if vessel_params is not None:
            vp_np = np.ascontiguousarray(np.asarray(vessel_params, dtype=np.float32).reshape(-1, _lib.NPARAM))
            n_classes = vp_np.shape[0]
            vp = vp_np.ctypes.data_as(C.POINTER(C.c_float))
        self.n_classes = n_classes
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dpenv_create(C.byref(cfg), vp, n_classes, C.byref(h)))
        self._h = h
        n, od = self.n_envs, self.num_states
        oshape = (n, od) if layout == 'aos' else (od, n)
        self._obs = [torch.empty(oshape, dtype=self.obs_torch_dtype, device=self.device) for _ in range(2)]
        self._flip = 0
        self._f32, self._u8 = torch.float32, torch.uint8
        self._rew = torch.empty(n, dtype=torch.float32, device=self.device)
        self._done = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._final_obs = None
        self._io = _lib.StepIO()
        self._io.struct_size = C.sizeof(_lib.StepIO)
        self._io_ref = C.byref(self._io)
        self._ok = {}
        self._dev_index = cfg.device
        self._raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        self._ashape, self._oshape = tuple(self.action_shape), tuple(self.obs_shape)
        self._n1, self._r3, self._p4 = (n,), (3, n), (4, n)
        self._step_ex = self.lib.dpenv_step_ex

    # -- plumbing -----------------------------------------------------------------------------
    # What a Python `for` over step() pays per call besides the launch is this plumbing (bench.py `eager_loop`): the stream handle and
    # the argument checks.  Both are kept cheap: the raw stream of the CURRENT stream context comes from torch's C binding where it
    # has one (a graph capture's side stream included), and a tensor object that passed the checks once is recognised by identity (a
    # weak reference, so a recycled id() cannot alias another tensor) and not checked again - shape, dtype, device and contiguity of a
    # live tensor object do not change short of resize_() / set_().
    def _stream(self):
        raw = self._raw_stream
        if raw is not None:
            return C.c_void_p(raw(self._dev_index))
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t, shape, dtype, what):
        """Argument check of a tensor handed to the C ABI: type, device, dtype, shape, contiguity - ONCE per tensor object (a weak reference keyed by
        id()): a tensor that passed is not looked at again, which is what keeps a Python `for` over step() at the GPU's rate (bench_side.json
        eager_loop: the full check costs ~0.4 us per tensor, four tensors per call, against a 4.9 us kernel).  The price: a tensor that
        is CHANGED IN PLACE between calls - t_(), transpose_(), resize_(), as_strided_(), set_(), `.data = ...` - is not re-checked, and the kernels
        would then read or write through a pointer whose extent no longer matches.  Do not reshape tensors you pass to reset / step / rollout in
        place; make a new tensor (a new object is checked)."""
        if t is None:
            return None
        ok = self._ok.get(id(t))
        if ok is not None and ok[0]() is t and ok[1] == shape and ok[2] is dtype:
            return t
        torch = _torch()
        if not (isinstance(t, torch.Tensor) and t.device == self.device and t.dtype == dtype
                and tuple(t.shape) == tuple(shape) and t.is_contiguous()):
            raise ValueError('%s must be a contiguous %s tensor of shape %s on %s (got %s)' % (
                what, dtype, tuple(shape), self.device,
                (tuple(t.shape), t.dtype, t.device) if isinstance(t, torch.Tensor) else type(t)))
        if len(self._ok) > 512:
            self._ok.clear()
        self._ok[id(t)] = (weakref.ref(t), tuple(shape), dtype)
        return t

    @staticmethod
    def _ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.dpenv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def action_shape(self):
        return (self.n_envs, self.num_actions) if self.layout == 'aos' else (self.num_actions, self.n_envs)

    @property
    def obs_shape(self):
        return (self.n_envs, self.num_states) if self.layout == 'aos' else (self.num_states, self.n_envs)

    # -- Gym-style API ------------------------------------------------------------------------
    def reset(self, mask=None, init=None, new_ref=None, fraction=None, out=None):
        """Revolt.reset (ENV:135-194) for the envs in ``mask`` (uint8 [n], None = all).

        init: float32 [6, n] rows N, E, psi, u, v, r (the reference's **init, ENV:141,152); None draws the
        training sample (ENV:143-145).  new_ref: float32 [3, n] setpoints (ENV:155-156); fraction: ENV:135.
        Returns the observation of every env."""
        torch = _torch()
        n = self.n_envs
        if fraction is not None:
            _lib.check(self.lib.dpenv_set_reset_fraction(self._h, float(fraction)), self._h)
        mask = self._chk(mask, (n,), torch.uint8, 'mask')
        init = self._chk(init, (6, n), torch.float32, 'init')
        new_ref = self._chk(new_ref, (3, n), torch.float32, 'new_ref')
        obs = out if out is not None else self._next_obs()
        self._chk(obs, self.obs_shape, self.obs_torch_dtype, 'out')
        _lib.check(self.lib.dpenv_reset(self._h, self._ptr(mask), self._ptr(init), self._ptr(new_ref), self._ptr(obs),
                                        self._stream()), self._h)
        return obs

    def _next_obs(self):
        self._flip ^= 1
        return self._obs[self._flip]

    def step(self, action, new_ref=None, out=None, reward_parts=None, final_obs=None):
        """Revolt.step (ENV:92-133) for all envs: returns (obs, reward, done_bits, info).

        action: float32 [n, act_dim] (or [act_dim, n] with layout='soa'); new_ref: float32 [3, n], visible from
        the NEXT observation (ENV:131).  out: optional (obs, reward, done) tensors to write into (e.g. rows of a
        [T, n, .] rollout buffer); otherwise internal buffers are reused (obs is double-buffered).
        Tensors are checked (device, dtype, shape, contiguity) the first time the OBJECT is seen, not on every call: do not change the shape or
        strides of a tensor in place (t_(), resize_(), set_() ...) between calls - pass a new tensor instead (_chk)."""
        f32, chk = self._f32, self._chk
        chk(action, self._ashape, f32, 'action')
        if out is None:
            obs, rew, done = self._next_obs(), self._rew, self._done
        else:
            obs, rew, done = out
            chk(obs, self._oshape, self.obs_torch_dtype, 'out[0]')
            chk(rew, self._n1, f32, 'out[1]')
            chk(done, self._n1, self._u8, 'out[2]')
        io = self._io
        io.action = action.data_ptr()
        io.new_ref = chk(new_ref, self._r3, f32, 'new_ref').data_ptr() if new_ref is not None else None
        io.obs = obs.data_ptr()
        io.reward = rew.data_ptr()
        io.done = done.data_ptr()
        io.reward_parts = chk(reward_parts, self._p4, f32, 'reward_parts').data_ptr() if reward_parts is not None else None
        io.final_obs = chk(final_obs, self._oshape, self.obs_torch_dtype, 'final_obs').data_ptr() if final_obs is not None else None
        rc = self._step_ex(self._h, self._io_ref, self._stream())
        if rc:
            _lib.check(rc, self._h)
        return obs, rew, done, {'None': 0}

    def rollout(self, actions, switch_steps=(), refs=None, out=None):
        """T env steps in one launch (dpenv_rollout): exactly T successive step() calls with actions[t], passing
        refs[k] as new_ref at step switch_steps[k] (setpoint sequences, test_policy.py:127,148-153), but with the
        state kept in registers between steps.  actions: float32 [T, n, act_dim] (or [T, act_dim, n] for 'soa').
        Returns (obs [T, n, obs_dim], reward [T, n], done_bits [T, n]); obs[t] is what step t returned."""
        torch = _torch()
        T = int(actions.shape[0])
        n = self.n_envs
        self._chk(actions, (T,) + self.action_shape, torch.float32, 'actions')
        k = len(switch_steps)
        if k > 8:
            raise ValueError('at most 8 setpoint switches per rollout launch')
        if k:
            self._chk(refs, (k, 3, n), torch.float32, 'refs')
        if out is None:
            obs = torch.empty((T,) + self.obs_shape, dtype=self.obs_torch_dtype, device=self.device)
            rew = torch.empty((T, n), dtype=torch.float32, device=self.device)
            done = torch.empty((T, n), dtype=torch.uint8, device=self.device)
        else:
            obs, rew, done = out
            self._chk(obs, (T,) + self.obs_shape, self.obs_torch_dtype, 'out[0]')
            self._chk(rew, (T, n), torch.float32, 'out[1]')
            self._chk(done, (T, n), torch.uint8, 'out[2]')
        io = _lib.RolloutIO()
        io.struct_size = C.sizeof(_lib.RolloutIO)
        io.T = T
        io.actions = actions.data_ptr()
        io.obs = obs.data_ptr()
        io.reward = rew.data_ptr()
        io.done = done.data_ptr()
        io.n_switch = k
        for j, st in enumerate(switch_steps):
            io.switch_step[j] = int(st)
        io.refs = refs.data_ptr() if k else None
        _lib.check(self.lib.dpenv_rollout(self._h, C.byref(io), self._stream()), self._h)
        return obs, rew, done

    # -- state access (parity tests, checkpoints) ----------------------------------------------
    def get_state(self):
        torch = _torch()
        st = torch.empty((_lib.NSTATE, self.n_envs), dtype=torch.float32, device=self.device)
        ctr = torch.empty((2, self.n_envs), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.dpenv_get_state(self._h, self._ptr(st), self._ptr(ctr), self._stream()), self._h)
        return st, ctr

    def set_state(self, state=None, counters=None):
        torch = _torch()
        state = self._chk(state, (_lib.NSTATE, self.n_envs), torch.float32, 'state')
        counters = self._chk(counters, (2, self.n_envs), torch.int32, 'counters')
        _lib.check(self.lib.dpenv_set_state(self._h, self._ptr(state), self._ptr(counters), self._stream()), self._h)

    def get_rng_counters(self):
        """(noise_ctr, drift_ctr): int32 [n] draws made so far of the in-kernel exploration noise and of the current drift (uint32
        bits).  With get_state() and get_current() a complete checkpoint: restoring all of them reproduces sampled rollouts."""
        torch = _torch()
        nc = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        dc = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.dpenv_get_rng_counters(self._h, self._ptr(nc), self._ptr(dc), self._stream()), self._h)
        return nc, dc

    def set_rng_counters(self, noise_ctr=None, drift_ctr=None):
        torch = _torch()
        self._chk(noise_ctr, (self.n_envs,), torch.int32, 'noise_ctr')
        self._chk(drift_ctr, (self.n_envs,), torch.int32, 'drift_ctr')
        _lib.check(self.lib.dpenv_set_rng_counters(self._h, self._ptr(noise_ctr), self._ptr(drift_ctr), self._stream()), self._h)

    def get_obs_thrust(self):
        """float32 [n, 4]: thrust columns of the observation the last closed-loop launch ended with (a mid-episode checkpoint needs
        them: the observation lags the stored thrust command by one step, customEnv.py:196-205,126).  Every call that changes the state
        keeps them: rollouts leave those of the last observation they returned, set_state those of an observation rebuilt from the state
        (previous thrust / 100), a reset those of the envs it re-draws, step() those of the observation it returns while a policy is in
        force.  None only while they are stale: after a step() WITHOUT a policy uploaded, or after a masked reset of a handle whose
        columns were already stale - the next closed-loop launch then starts from the state block alone."""
        torch = _torch()
        t = torch.empty((self.n_envs, 4), dtype=torch.float32, device=self.device)
        rc = self.lib.dpenv_get_obs_thrust(self._h, self._ptr(t), self._stream())
        return t if rc == _lib.OK else None

    def set_obs_thrust(self, t):
        self._chk(t, (self.n_envs, 4), _torch().float32, 'obs_thrust')
        _lib.check(self.lib.dpenv_set_obs_thrust(self._h, self._ptr(t), self._stream()), self._h)

    def set_current(self, vc, beta, present_only=False):
        """Per-env constant current: speed [m/s] and NED direction [rad] (results/.../current_box_test/plot_pos.py:78).  They are
        both the present value and the mean a drifting current reverts to; present_only=True restores only the present values
        (what get_current returned: checkpoints of a drifting current)."""
        torch = _torch()
        self._chk(vc, (self.n_envs,), torch.float32, 'vc')
        self._chk(beta, (self.n_envs,), torch.float32, 'beta')
        fn = self.lib.dpenv_set_current_present if present_only else self.lib.dpenv_set_current
        _lib.check(fn(self._h, self._ptr(vc), self._ptr(beta), self._stream()), self._h)

    def get_current(self):
        """Present (vc, beta) of every env; differs from what set_current gave only with current_drift."""
        torch = _torch()
        vc = torch.empty(self.n_envs, dtype=torch.float32, device=self.device)
        beta = torch.empty(self.n_envs, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dpenv_get_current(self._h, self._ptr(vc), self._ptr(beta), self._stream()), self._h)
        return vc, beta

    def get_current_mean(self):
        """(vc, beta) the drift reverts to: what set_current gave, or what the last randomised reset drew (set_current_randomisation)."""
        torch = _torch()
        vc = torch.empty(self.n_envs, dtype=torch.float32, device=self.device)
        beta = torch.empty(self.n_envs, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dpenv_get_current_mean(self._h, self._ptr(vc), self._ptr(beta), self._stream()), self._h)
        return vc, beta

    def set_current_randomisation(self, vc_range, beta_range, vc_nominal=None, beta_nominal=None):
        """Per-episode randomisation of the current through the reset path (dpenv_set_current_randomisation): every reset starts its
        episode in V_c = max(0, vc_nominal + vc_range u1), beta_c = beta_nominal + beta_range u2, u ~ U[-1, 1), keyed (seed; global env
        id, episode).  Nominals: float32 [n] device tensors, default the means in force (set_current).  Both ranges 0: off."""
        torch = _torch()
        for t, what in ((vc_nominal, 'vc_nominal'), (beta_nominal, 'beta_nominal')):
            if t is not None:
                self._chk(t, (self.n_envs,), torch.float32, what)
        _lib.check(self.lib.dpenv_set_current_randomisation(self._h, self._ptr(vc_nominal), self._ptr(beta_nominal), float(vc_range),
                                                            float(beta_range), self._stream()), self._h)

    def set_vessel_class(self, class_id):
        torch = _torch()
        self._chk(class_id, (self.n_envs,), torch.int32, 'class_id')
        if int(class_id.min()) < 0 or int(class_id.max()) >= self.n_classes:
            raise ValueError('class ids must be in [0, %d)' % self.n_classes)
        _lib.check(self.lib.dpenv_set_vessel_class(self._h, self._ptr(class_id), self._stream()), self._h)

    def set_vessel_params(self, params, keep_randomisation=False):
        """Per-env hull / thruster parameters (dpenv_set_vessel_params_ex): float32 [NPARAM, n] device tensor, row p = parameter
        _lib.P[...] of every env - the constants the reference hard-codes for its one vessel (qp_allocator.py:51-55,69-70;
        SupervisedTau.py:35-36,69-71), the build-owned plant's mass / damping terms and (rows 26..31) the thrusters' inflow-loss
        coefficients.  A single length-NPARAM vector is given to every env.  None returns to the vessel classes / the single class given
        to the constructor.  Stream-ordered (no read-back; may be recorded into a graph).  keep_randomisation=True installs the table
        while set_vessel_randomisation stays in force: the restore path of a checkpoint (get_state + get_vessel_params) taken mid-episode."""
        torch = _torch()
        if params is not None and not (hasattr(params, 'dim') and params.dim() == 2):
            one = torch.as_tensor(np.asarray(params.cpu() if hasattr(params, 'cpu') else params, np.float32).reshape(_lib.NPARAM), device=self.device)
            params = one[:, None].expand(_lib.NPARAM, self.n_envs).contiguous()
        self._chk(params, (_lib.NPARAM, self.n_envs), torch.float32, 'params')
        flags = _lib.VESSEL_KEEP_RANDOMISATION if keep_randomisation else 0
        _lib.check(self.lib.dpenv_set_vessel_params_ex(self._h, self._ptr(params), flags, self._stream()), self._h)

    def get_vessel_params(self):
        """float32 [NPARAM, n]: the per-env parameter vectors in force (set explicitly, or drawn by the randomisation)."""
        torch = _torch()
        out = torch.empty((_lib.NPARAM, self.n_envs), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dpenv_get_vessel_params(self._h, self._ptr(out), self._stream()), self._h)
        return out

    def set_vessel_randomisation(self, rel_range, nominal=None):
        """Domain randomisation through the reset path (dpenv_set_vessel_randomisation): every reset starts its episode on a hull with
        parameter p = nominal[p] * (1 + rel_range[p] * u), u ~ U[-1, 1), keyed (seed; global env id, episode).  rel_range: a float (the
        same relative half-range for every parameter), a length-NPARAM sequence, or None to stop re-drawing; nominal: a parameter
        vector, default class 0."""
        if rel_range is None:
            _lib.check(self.lib.dpenv_set_vessel_randomisation(self._h, None, None, self._stream()), self._h)
            return
        rr = np.zeros(_lib.NPARAM, np.float32)
        if np.isscalar(rel_range):
            rr[:_lib.NPARAM_USED] = float(rel_range)
        else:
            rr[:] = np.asarray(rel_range, np.float32).reshape(_lib.NPARAM)
        nom = None
        if nominal is not None:
            nom_np = np.ascontiguousarray(np.asarray(nominal, np.float32).reshape(_lib.NPARAM))
            nom = nom_np.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(self.lib.dpenv_set_vessel_randomisation(self._h, nom, rr.ctypes.data_as(C.POINTER(C.c_float)), self._stream()), self._h)

    def render(self):
        pass   # ENV:246-247


def thrust_map(n_pct, alpha, params=None):
    """tau = B(alpha) F(n) on device (SupervisedTau.py:42-83): n_pct, alpha float32 [3, n] in env order
    (bow, port, star); returns float32 [3, n] = Fx, Fy, Mz."""
    torch = _torch()
    lib = _lib.load()
    assert n_pct.shape == alpha.shape and n_pct.shape[0] == 3 and n_pct.is_contiguous() and alpha.is_contiguous()
    assert n_pct.dtype == torch.float32 and alpha.dtype == torch.float32 and n_pct.is_cuda
    tau = torch.empty_like(n_pct)
    p = None
    if params is not None:
        pa = np.ascontiguousarray(params, dtype=np.float32)
        assert pa.shape == (_lib.NPARAM,)
        p = pa.ctypes.data_as(C.POINTER(C.c_float))
    s = C.c_void_p(torch.cuda.current_stream(n_pct.device).cuda_stream)
    with torch.cuda.device(n_pct.device):
        _lib.check(lib.dpenv_thrust_map(p, C.c_void_p(n_pct.data_ptr()), C.c_void_p(alpha.data_ptr()),
                                        C.c_void_p(tau.data_ptr()), n_pct.shape[1], s))
    return tau


class _ErrorFrameView(object):
    """What test_policy.py:117-121 reads from env.EF (errorFrame.py:15-23), backed by the device state."""

    def __init__(self, env):
        self._env = env

    def _row(self, a, b):
        st, _ = self._env._benv.get_state()
        return [float(x) for x in st[a:b, 0].cpu().numpy()]

    def get_NED_pos(self):
        return self._row(_lib.S['N'], _lib.S['N'] + 3)

    def get_NED_ref(self):
        return self._row(_lib.S['REF_N'], _lib.S['REF_N'] + 3)

    def get_pose(self):
        return [float(x) for x in self._env.state()[:3]]

    def update(self, pos=None, ref=None):
        st, _ = self._env._benv.get_state()
        torch = _torch()
        if pos:
            st[_lib.S['N']:_lib.S['N'] + 3, 0] = torch.tensor(pos, dtype=torch.float32)
        if ref:
            st[_lib.S['REF_N']:_lib.S['REF_N'] + 3, 0] = torch.tensor(ref, dtype=torch.float32)
        self._env._benv.set_state(st.contiguous(), None)


class Revolt(object):
    """Single-environment adapter with the reference's constructor and Gym API (ENV:11-325)."""

    _variant = 'full'

    def __init__(self, digitwin=None, num_actions=None, num_states=6, real_ss_bounds=None, testing=False,
                 realtime=False, max_ep_len=800, extended_state=False, reset_acts=False, cont_ang=False,
                 device='cuda:0', seed=0, wrap_mode='reference'):
        torch = _torch()
        self.dTwin = digitwin   # kept for attribute compatibility; never called
        self._benv = BatchedRevoltEnv(1, variant=self._variant, extended_state=extended_state, cont_ang=cont_ang,
                                      device=device, testing=testing, realtime=realtime, max_ep_len=max_ep_len,
                                      auto_reset=False, terminate=True, wrap_mode=wrap_mode, seed=seed,
                                      time_limit=False)
        b = self._benv
        for k in ('name', 'extended_state', 'cont_ang', 'testing', 'num_actions', 'num_states', 'real_action_bounds',
                  'real_ss_bounds', 'default_actions', 'valid_action_indices', 'act_2_act_map', 'act_2_act_map_inv',
                  'n_steps', 'dt', 'max_ep_len', 'vel_rew_coeffs', 'covar', 'observation_space', 'action_space'):
            setattr(self, k, getattr(b, k))
        self.reset_actions = reset_acts
        self.covar_inv = np.linalg.inv(self.covar)
        self.EF = _ErrorFrameView(self)
        self.metadata = {'render.modes': ['human']}
        self._act = torch.zeros(b.action_shape, dtype=torch.float32, device=b.device)
        self._ref = torch.zeros((3, 1), dtype=torch.float32, device=b.device)
        self._init = torch.zeros((6, 1), dtype=torch.float32, device=b.device)
        self._rng = np.random
        # step() I/O in PINNED HOST memory that the kernel reads and writes directly (hipHostMalloc memory is mapped into the device's
        # address space at the same address): one launch and one stream synchronisation per step instead of an H2D copy, a launch and
        # three D2H copies with a synchronisation each.  The C ABI wants device-ACCESSIBLE pointers, which these are.
        self._h_act = torch.zeros(b.action_shape, dtype=torch.float32).pin_memory()
        self._h_ref = torch.zeros((3, 1), dtype=torch.float32).pin_memory()
        self._h_obs = torch.zeros(b.obs_shape, dtype=torch.float32).pin_memory()
        self._h_rew = torch.zeros(1, dtype=torch.float32).pin_memory()
        self._h_done = torch.zeros(1, dtype=torch.uint8).pin_memory()
        self._np_act, self._np_ref = self._h_act.numpy(), self._h_ref.numpy()
        self._np_obs, self._np_rew, self._np_done = self._h_obs.numpy(), self._h_rew.numpy(), self._h_done.numpy()
        self._sio = _lib.StepIO()
        self._sio.struct_size = C.sizeof(_lib.StepIO)
        self._sio.action = self._h_act.data_ptr()
        self._sio.obs = self._h_obs.data_ptr()
        self._sio.reward = self._h_rew.data_ptr()
        self._sio.done = self._h_done.data_ptr()

    # ENV:92-133
    def step(self, action, new_ref=None):
        torch = _torch()
        b = self._benv
        if b.obs_torch_dtype != torch.float32:
            raise ValueError('the single-env adapter returns float64 observations made from float32 rows')
        a = np.asarray(action, dtype=np.float32)
        if a.size != b.num_actions:
            raise ValueError('action must have %d elements for the %s env (got shape %s)' % (b.num_actions, b.name, a.shape))
        self._np_act[...] = a.reshape(b.action_shape)
        self._sio.new_ref = None
        if new_ref is not None:
            self._np_ref[...] = np.asarray(new_ref, dtype=np.float32).reshape(3, 1)
            self._sio.new_ref = self._h_ref.data_ptr()
        stream = torch.cuda.current_stream(b.device)
        _lib.check(b.lib.dpenv_step_ex(b._h, C.byref(self._sio), C.c_void_p(stream.cuda_stream)), b._h)
        stream.synchronize()                     # the kernel's stores to the pinned rows are visible to the host from here
        o = self._np_obs.reshape(-1).astype(np.float64)
        r = float(self._np_rew[0])
        d = bool(int(self._np_done[0]) & (_lib.DONE_TERMINAL | _lib.DONE_FAULT))
        return o, r, d, {'None': 0}

    # ENV:135-194
    def reset(self, new_ref=None, fraction=0.8, fixed_point=None, **init):
        torch = _torch()
        if not init:
            N, E, Y, u, v, r = 0, 0, 0, 0, 0, 0
            if not self.testing:
                N, E, Y = simtools.get_pose_on_state_space(self.real_ss_bounds[0:3], fraction=fraction)
                u, v, r = simtools.get_vel_on_state_space(self.real_ss_bounds[3:], fraction=0.30 * fraction)
            elif fixed_point is None:
                N, E, Y = simtools.get_random_pose_on_radius()
            else:
                N, E, Y = simtools.get_fixed_pose_on_radius(n=fixed_point)
        else:
            # the reference forwards 'Module.Feature' keys to the plant (ENV:152,159-161)
            N, E = init.get('Hull.PosNED', [0, 0])[:2]
            Y = init.get('Hull.PosAttitude', [0, 0, 0])[2]
            nu6 = init.get('Hull.VelocityNu', [0] * 6)
            u, v, r = nu6[0], nu6[1], nu6[5]
        self._init.copy_(torch.tensor([N, E, Y, u, v, r], dtype=torch.float32).reshape(6, 1))
        nr = None
        if self.testing and new_ref is not None:   # ENV:155-156
            self._ref.copy_(torch.tensor(new_ref, dtype=torch.float32).reshape(3, 1))
            nr = self._ref
        obs = self._benv.reset(init=self._init, new_ref=nr)
        if self.reset_actions:
            # ENV:179-188: previous thrust ~ clip(N(0, 0.1) * 100)
            a = np.clip(self._rng.normal(loc=0.0, scale=0.1, size=3) * 100.0, -100.0, 100.0)
            st, _ = self._benv.get_state()
            st[_lib.S['PT_BOW']:_lib.S['PT_BOW'] + 3, 0] = torch.tensor(a, dtype=torch.float32)
            self._benv.set_state(st.contiguous(), None)
            return self.state_extended() if self.extended_state else self.state()
        return obs.float().cpu().numpy().reshape(-1).astype(np.float64)

    # ENV:196-213 (host-side views of the device state; not on the hot path)
    def _state9(self):
        st, _ = self._benv.get_state()
        s = st[:, 0].cpu().numpy().astype(np.float64)
        eN, eE = s[0] - s[6], s[1] - s[7]
        deg = self._benv.cfg.wrap_mode == _lib.WRAP_REFERENCE
        ref = 180.0 if deg else math.pi
        wrap = lambda x: np.mod(x + ref, 2 * ref) - ref
        rot = wrap(s[2])
        x = math.cos(rot) * eN + math.sin(rot) * eE
        y = -math.sin(rot) * eN + math.cos(rot) * eE
        return np.array([x, y, wrap(s[2] - s[8]), s[3], s[4], s[5], s[9] / 100.0, s[10] / 100.0, s[11] / 100.0])

    def state(self):
        return self._state9()[:6]

    def state_extended(self):
        return self._state9()

    def is_terminal(self):
        return bool(np.any(np.abs(self.state()) > np.array(self.real_ss_bounds)))

    # ENV:215-244 (pure host helpers used by the evaluation harness, test_policy.py:160)
    def scale_and_clip(self, action):
        bnds = np.array(self.real_action_bounds)
        return np.clip(np.multiply(action, bnds), -bnds, bnds).tolist()

    def handle_continuous_angles(self, action):
        assert self.name.lower() == 'revoltfinal' and self.cont_ang is True
        a_port = np.arctan2(action[3], action[4]) / self.real_action_bounds[3]
        a_star = np.arctan2(action[5], action[6]) / self.real_action_bounds[3]
        return np.hstack((action[0:3], np.array([a_port, a_star])))

    def wrap_stern_angles(self, action):
        assert self.name.lower() == 'revoltfinal' and self.cont_ang is False
        b = self.real_action_bounds[3]
        w = lambda x: (np.mod(x * b + math.pi, 2 * math.pi) - math.pi) / b
        return np.hstack((action[0:3], np.array([w(action[3]), w(action[4])])))

    def render(self):
        pass

    def close(self):
        self._benv.close()


class RevoltSimple(Revolt):
    _variant = 'simple'

    def __init__(self, digitwin=None, testing=False, realtime=False, max_ep_len=800, extended_state=False,
                 reset_acts=False, cont_ang=False, **kw):
        super().__init__(digitwin=digitwin, testing=testing, realtime=realtime, max_ep_len=max_ep_len,
                         extended_state=extended_state, reset_acts=reset_acts, cont_ang=False, **kw)


class RevoltLimited(Revolt):
    _variant = 'limited'

    def __init__(self, digitwin=None, testing=False, realtime=False, max_ep_len=800, extended_state=False,
                 reset_acts=False, cont_ang=False, **kw):
        super().__init__(digitwin=digitwin, testing=testing, realtime=realtime, max_ep_len=max_ep_len,
                         extended_state=extended_state, reset_acts=reset_acts, cont_ang=False, **kw)


class RevoltFinal(Revolt):
    _variant = 'final'

    def __init__(self, digitwin=None, testing=False, realtime=False, max_ep_len=800, extended_state=False,
                 reset_acts=False, cont_ang=False, **kw):
        super().__init__(digitwin=digitwin, testing=testing, realtime=realtime, max_ep_len=max_ep_len,
                         extended_state=extended_state, reset_acts=reset_acts, cont_ang=cont_ang, **kw)


ENVIRONMENTS = {'simple': RevoltSimple, 'limited': RevoltLimited, 'final': RevoltFinal, 'full': Revolt}   # trainer.py:15
