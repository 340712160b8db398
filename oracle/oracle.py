"""ctypes front-end of the CPU oracle (oracle/libdpenv_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package ml4ca_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libdpenv_oracle.so')

FULL, SIMPLE, LIMITED, FINAL = 0, 1, 2, 3
WRAP_REFERENCE, WRAP_RADIANS = 0, 1
NSTATE, NPARAM = 15, 32
S = dict(N=0, E=1, PSI=2, U=3, V=4, R=5, REF_N=6, REF_E=7, REF_PSI=8,
         PT_BOW=9, PT_PORT=10, PT_STAR=11, A_BOW=12, A_PORT=13, A_STAR=14)
MODES = {  # fixture name -> (variant, cont_ang)
    'full': (FULL, 0), 'simple': (SIMPLE, 0), 'limited': (LIMITED, 0),
    'final_wrap': (FINAL, 0), 'final_cont': (FINAL, 1),
}


class Config(C.Structure):
    _fields_ = [('variant', C.c_int32), ('extended_state', C.c_int32), ('cont_ang', C.c_int32),
                ('n_substeps', C.c_int32), ('substep_dt', C.c_double), ('wrap_mode', C.c_int32),
                ('terminate', C.c_int32), ('max_ep_len', C.c_int32), ('auto_reset', C.c_int32),
                ('current_enabled', C.c_int32), ('seed', C.c_uint64), ('env_id_base', C.c_int64),
                ('reset_fraction', C.c_double), ('current_drift', C.c_int32), ('current_tau', C.c_double),
                ('current_sigma_v', C.c_double), ('current_sigma_beta', C.c_double), ('reset_acts', C.c_int32)]


def make_config(variant=FINAL, extended_state=1, cont_ang=1, n_substeps=20, substep_dt=0.01,
                wrap_mode=WRAP_REFERENCE, terminate=1, max_ep_len=0, auto_reset=0, current_enabled=0,
                seed=0, env_id_base=0, reset_fraction=0.8, current_drift=0, current_tau=100.0, current_sigma_v=0.02,
                current_sigma_beta=5.0 * np.pi / 180.0, reset_acts=0):
    return Config(variant, extended_state, cont_ang, n_substeps, substep_dt, wrap_mode, terminate,
                  max_ep_len, auto_reset, current_enabled, seed, env_id_base, reset_fraction,
                  current_drift, current_tau, current_sigma_v, current_sigma_beta, int(reset_acts))


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ('dpenv_oracle.c', 'dpenv_oracle_impl.h', 'dpenv_oracle.h')]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libdpenv_oracle.so'], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get('DPENV_ORACLE_SO')          # an instrumented build (tools/oracle_sanitize.sh)
        if alt:
            _lib = C.CDLL(alt)
        else:
            build()
            _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle(object):
    """Thin array-in/array-out wrapper; dtype selects the f64 or f32 build."""

    def __init__(self, cfg, dtype=np.float64, vessel=None):
        self.cfg = cfg
        self.dtype = np.dtype(dtype)
        self.sfx = 'f64' if self.dtype == np.float64 else 'f32'
        self.L = lib()
        self.act_dim = self._f('dpo_act_dim')(C.byref(cfg))
        self.obs_dim = self._f('dpo_obs_dim')(C.byref(cfg))
        if vessel is None:
            vessel = np.zeros(NPARAM, self.dtype)
            self._f('dpo_default_vessel')(_p(vessel))
        self.vessel = np.ascontiguousarray(vessel, self.dtype)

    def _f(self, name):
        return getattr(self.L, '%s_%s' % (name, self.sfx))

    def _a(self, x, shape=None):
        if x is None:
            return None
        x = np.ascontiguousarray(x, self.dtype)
        if shape is not None:
            assert x.shape == tuple(shape), (x.shape, shape)
        return x

    def decode(self, action, ang_in):
        a = self._a(action, (self.act_dim,))
        ai = self._a(ang_in, (3,))
        t = np.zeros(3, self.dtype)
        ao = np.zeros(3, self.dtype)
        self._f('dpo_decode')(C.byref(self.cfg), _p(a), _p(ai), _p(t), _p(ao))
        return t, ao

    def thrust_map(self, n_pct, alpha, vessel=None):
        v = self.vessel if vessel is None else self._a(vessel, (NPARAM,))
        n = self._a(n_pct, (3,))
        al = self._a(alpha, (3,))
        tau = np.zeros(3, self.dtype)
        self._f('dpo_thrust_map')(_p(v), _p(n), _p(al), _p(tau))
        return tau

    def plant(self, eta, nu, n_pct, alpha, current=None):
        e = self._a(eta, (3,)).copy()
        v = self._a(nu, (3,)).copy()
        cur = self._a(current, (2,))
        self._f('dpo_plant')(C.byref(self.cfg), _p(self.vessel), _p(e), _p(v), _p(self._a(n_pct, (3,))),
                             _p(self._a(alpha, (3,))), _p(cur))
        return e, v

    def obs(self, eta, nu, ref, prev_thrust):
        o = np.zeros(self.obs_dim, self.dtype)
        self._f('dpo_obs')(C.byref(self.cfg), _p(self._a(eta, (3,))), _p(self._a(nu, (3,))),
                           _p(self._a(ref, (3,))), _p(self._a(prev_thrust, (3,))), _p(o))
        return o

    def reward(self, obs, thrust_now, ang_cur, ang_prev):
        parts = np.zeros(4, self.dtype)
        self._f('dpo_reward')(C.byref(self.cfg), _p(self._a(obs, (self.obs_dim,))), _p(self._a(thrust_now, (3,))),
                              _p(self._a(ang_cur, (3,))), _p(self._a(ang_prev, (3,))), _p(parts))
        return parts

    def done(self, obs):
        return int(self._f('dpo_done')(C.byref(self.cfg), _p(self._a(obs, (self.obs_dim,)))))

    def sample_reset(self, gid, episode):
        e = np.zeros(3, self.dtype)
        v = np.zeros(3, self.dtype)
        self._f('dpo_sample_reset')(C.byref(self.cfg), C.c_int64(gid), C.c_uint32(episode), _p(e), _p(v))
        return e, v

    # ---- batched state machine -------------------------------------------------
    def policy_noise(self, gids, draw, adim):
        """Exploration noise xi [len(gids), adim] of action number `draw` (int or array) of the envs with global ids `gids`."""
        gids = np.asarray(gids, np.int64).reshape(-1)
        draws = np.broadcast_to(np.asarray(draw, np.uint32), gids.shape)
        out = np.zeros((gids.size, adim), self.dtype)
        fn = self._f('dpo_policy_noise')
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int32, C.c_void_p]
        row = np.zeros(8, self.dtype)
        for j in range(gids.size):
            fn(C.byref(self.cfg), int(gids[j]), int(draws[j]), adim, _p(row))
            out[j] = row[:adim]
        return out

    def new_state(self, n):
        return np.zeros((NSTATE, n), self.dtype), np.zeros((2, n), np.int32)

    def draw_vessel(self, rand_tab, gid, episode):
        """The randomisation's hull of episode `episode` of env `gid`: rand_tab = [nominal (32) | relative half-range (32)]."""
        rt = self._a(rand_tab, (2 * NPARAM,))
        p = np.zeros(NPARAM, self.dtype)
        self._f('dpo_draw_vessel')(C.byref(self.cfg), _p(rt), C.c_int64(int(gid)), C.c_uint32(int(episode)), _p(p))
        return p

    def _vessel_env(self, vessel_env, rand_tab, n):
        """per-env parameter table [NPARAM, n] (updated IN PLACE by resets when rand_tab is given) and the randomisation table"""
        if vessel_env is not None:
            assert vessel_env.dtype == self.dtype and vessel_env.flags.c_contiguous and vessel_env.shape == (NPARAM, n)
        if rand_tab is not None:
            assert vessel_env is not None
        return vessel_env, self._a(rand_tab, (2 * NPARAM,))

    def draw_current(self, gid, episode, nom_v, nom_b, range_v, range_b):
        """(V_c, beta_c) of episode `episode` of env `gid` under the per-episode randomisation of the current."""
        out = np.zeros(2, self.dtype)
        fn = self._f('dpo_draw_current')
        ft = C.c_double if self.sfx == 'f64' else C.c_float
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, ft, ft, ft, ft, C.c_void_p]
        fn(C.byref(self.cfg), int(gid), int(episode), float(nom_v), float(nom_b), float(range_v), float(range_b), _p(out))
        return out

    def _cur_rand(self, cur_rand, current, current_mean, n):
        """cur_rand = (range_v, range_b, nominal_v [n], nominal_b [n]) -> the oracle's flat table; current / current_mean [2][n] are then
        updated IN PLACE by every reset (pass contiguous arrays of the oracle's dtype)"""
        if cur_rand is None:
            return None
        rv, rb, nv, nb = cur_rand
        for x in (current, current_mean):
            assert x is not None and x.dtype == self.dtype and x.flags.c_contiguous and x.shape == (2, n)
        return np.ascontiguousarray(np.concatenate([[rv, rb], np.asarray(nv).reshape(n), np.asarray(nb).reshape(n)]), self.dtype)

    def reset(self, state, counters, mask=None, init=None, ref=None, vessel_env=None, rand_tab=None, current=None, current_mean=None,
              cur_rand=None):
        n = state.shape[1]
        obs = np.zeros((n, self.obs_dim), self.dtype)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        ve, rt = self._vessel_env(vessel_env, rand_tab, n)
        cr = self._cur_rand(cur_rand, current, current_mean, n)
        self._f('dpo_reset')(C.byref(self.cfg), C.c_int32(n), _p(state), _p(counters), _p(m),
                             _p(self._a(init, (6, n))), _p(self._a(ref, (3, n))), _p(obs), _p(ve), _p(rt),
                             _p(current if cr is not None else None), _p(current_mean if cr is not None else None), _p(cr))
        return obs

    def step(self, state, counters, action, new_ref=None, plant_override=None, current=None,
             want_parts=False, want_final_obs=False, current_mean=None, drift_ctr=None, vessel_env=None, rand_tab=None, cur_rand=None):
        """current [2][n] is updated IN PLACE when the config enables drift (pass a contiguous array of the
        oracle's dtype together with current_mean [2][n] and drift_ctr uint32[n])."""
        n = state.shape[1]
        assert state.dtype == self.dtype and state.flags.c_contiguous and counters.dtype == np.int32
        a = self._a(action, (n, self.act_dim))
        obs = np.zeros((n, self.obs_dim), self.dtype)
        rew = np.zeros(n, self.dtype)
        done = np.zeros(n, np.uint8)
        parts = np.zeros((n, 4), self.dtype) if want_parts else None
        fobs = np.zeros((n, self.obs_dim), self.dtype) if want_final_obs else None
        if current is not None and self.cfg.current_drift:
            assert current.dtype == self.dtype and current.flags.c_contiguous and current.shape == (2, n)
            assert drift_ctr is not None and drift_ctr.dtype == np.uint32 and current_mean is not None
            cur = current
        else:
            cur = self._a(current, (2, n))
        cr = self._cur_rand(cur_rand, current, current_mean, n)         # (asserts that current / current_mean can be written in place)
        ve, rt = self._vessel_env(vessel_env, rand_tab, n)
        self._f('dpo_step')(C.byref(self.cfg), _p(self.vessel), C.c_int32(n), _p(state), _p(counters), _p(a),
                            _p(self._a(new_ref, (3, n))), _p(self._a(plant_override, (6, n))),
                            _p(cur), _p(obs), _p(rew), _p(done), _p(parts), _p(fobs),
                            _p(current_mean if cr is not None else self._a(current_mean, (2, n))), _p(drift_ctr), _p(ve), _p(rt), _p(cr))
        out = [obs, rew, done]
        if want_parts:
            out.append(parts)
        if want_final_obs:
            out.append(fobs)
        return tuple(out)

    def step_into(self, state, counters, action, obs, rew, done, new_ref=None):
        """step() writing into caller-provided arrays: nothing but the C loop runs (used for CPU timing)."""
        n = state.shape[1]
        self._f('dpo_step')(C.byref(self.cfg), _p(self.vessel), C.c_int32(n), _p(state), _p(counters), _p(action),
                            _p(new_ref), None, None, _p(obs), _p(rew), _p(done), None, None, None, None, None, None, None)

    def discount_cumsum(self, x, discount):
        x = self._a(x)
        y = np.zeros_like(x)
        fn = self._f('dpo_discount_cumsum')
        fn.argtypes = [C.c_void_p, C.c_int32, C.c_double if self.sfx == 'f64' else C.c_float, C.c_void_p]
        fn(_p(x), x.shape[0], discount, _p(y))
        return y

    def gae(self, rew, val, end=None, boot=None, last_val=None, gamma=0.99, lam=0.97):
        T, n = rew.shape
        rew = self._a(rew)
        val = self._a(val, (T, n))
        adv = np.zeros((T, n), self.dtype)
        ret = np.zeros((T, n), self.dtype)
        e = None if end is None else np.ascontiguousarray(end, np.uint8)
        ft = C.c_double if self.sfx == 'f64' else C.c_float
        fn = self._f('dpo_gae')
        fn.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_int32, ft, ft, C.c_void_p, C.c_void_p]
        fn(_p(rew), _p(val), _p(e), _p(self._a(boot, (T, n))), _p(self._a(last_val, (n,))), T, n, gamma, lam,
           _p(adv), _p(ret))
        return adv, ret

    def normalize_adv(self, adv):
        a = self._a(adv).copy()
        ms = np.zeros(2, self.dtype)
        fn = self._f('dpo_normalize_adv')
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        fn(_p(a), a.size, _p(ms))
        return a, ms


def set_threads(n):
    """Threads of the batched CPU step loop (OpenMP over envs); returns the count in force."""
    return int(lib().dpo_set_threads(int(n)))


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().dpo_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)
