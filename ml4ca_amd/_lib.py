"""ctypes binding of libdpenv.so (C ABI in include/dpenv.h).

There is deliberately no fallback: if the HIP library is missing or no MI355X is usable the
import / constructor raises.  Nothing here touches oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPENV_LIB overrides the path (A/B builds of the kernel library during tuning); still no fallback of any kind
LIB_PATH = os.environ.get('DPENV_LIB') or os.path.join(_HERE, 'lib', 'libdpenv.so')

OK, EINVAL, ENODEV, ENOMEM, EHIP = 0, -1, -2, -3, -4
FULL, SIMPLE, LIMITED, FINAL = 0, 1, 2, 3
AOS, SOA = 0, 1
WRAP_REFERENCE, WRAP_RADIANS = 0, 1
F32, BF16 = 0, 1
ACT_LEAKY_RELU, ACT_TANH = 0, 1
POLICY_F16, POLICY_F32, POLICY_F32_ACTOR = 0, 1, 2
LAUNCH_AUTO, LAUNCH_ONE_WAVE, LAUNCH_TWO_WAVE = 0, 1, 2
DONE_TERMINAL, DONE_TIMELIMIT, DONE_FAULT = 1, 2, 4
NSTATE, NPARAM, NPARAM_USED, MAX_CLASSES = 15, 32, 32, 64
ABI_VERSION = 5
VESSEL_KEEP_RANDOMISATION = 1          # dpenv_set_vessel_params_ex flag

# canonical state rows (dpenv.h DPENV_S_*)
S = dict(N=0, E=1, PSI=2, U=3, V=4, R=5, REF_N=6, REF_E=7, REF_PSI=8,
         PT_BOW=9, PT_PORT=10, PT_STAR=11, A_BOW=12, A_PORT=13, A_STAR=14)
# vessel parameter slots (dpenv.h DPENV_P_*)
P = dict(M11=0, M22=1, M23=2, M33=3, XU=4, XUU=5, YV=6, YVV=7, YR=8, NV=9, NR=10, NRR=11,
         KF_BOW=12, KF_PORT=13, KF_STAR=14, KR_BOW=15, KR_PORT=16, KR_STAR=17,
         LX_BOW=18, LX_PORT=19, LX_STAR=20, LY_BOW=21, LY_PORT=22, LY_STAR=23, NUV=24, YUR=25,
         KLF_BOW=26, KLF_PORT=27, KLF_STAR=28, KLR_BOW=29, KLR_PORT=30, KLR_STAR=31)


class Config(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('n_envs', C.c_int32), ('device', C.c_int32), ('variant', C.c_int32),
                ('extended_state', C.c_int32), ('cont_ang', C.c_int32), ('n_substeps', C.c_int32),
                ('substep_dt', C.c_float), ('wrap_mode', C.c_int32), ('terminate', C.c_int32),
                ('max_ep_len', C.c_int32), ('auto_reset', C.c_int32), ('action_layout', C.c_int32),
                ('obs_layout', C.c_int32), ('obs_dtype', C.c_int32), ('current_enabled', C.c_int32),
                ('seed', C.c_uint64), ('env_id_base', C.c_int64), ('reset_fraction', C.c_float),
                ('hold_plant', C.c_int32), ('current_drift', C.c_int32), ('current_tau', C.c_float),
                ('current_sigma_v', C.c_float), ('current_sigma_beta', C.c_float), ('reset_acts', C.c_int32), ('step_one_wave', C.c_int32),
                ('per_env_lds', C.c_int32)]


class StepIO(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('action', C.c_void_p), ('new_ref', C.c_void_p), ('obs', C.c_void_p),
                ('reward', C.c_void_p), ('done', C.c_void_p), ('reward_parts', C.c_void_p), ('final_obs', C.c_void_p)]


class RolloutIO(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('T', C.c_int32), ('actions', C.c_void_p), ('obs', C.c_void_p),
                ('reward', C.c_void_p), ('done', C.c_void_p), ('n_switch', C.c_int32), ('switch_step', C.c_int32 * 8),
                ('refs', C.c_void_p)]


class Mlp(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('sizes', C.c_int32 * 6), ('W', C.c_void_p * 5), ('b', C.c_void_p * 5)]


class PolicyRolloutIO(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('T', C.c_int32), ('noise', C.c_void_p), ('obs', C.c_void_p),
                ('act', C.c_void_p), ('reward', C.c_void_p), ('value', C.c_void_p), ('logp', C.c_void_p),
                ('done', C.c_void_p), ('boot', C.c_void_p), ('last_obs', C.c_void_p), ('last_value', C.c_void_p),
                ('n_switch', C.c_int32), ('switch_step', C.c_int32 * 8), ('refs', C.c_void_p), ('sample', C.c_int32),
                ('reset_at_end', C.c_int32)]


class PolicyDesc(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('pi', C.POINTER(Mlp)), ('v', C.POINTER(Mlp)), ('log_std', C.c_void_p),
                ('activation', C.c_int32), ('leak', C.c_float), ('precision', C.c_int32), ('launch_form', C.c_int32),
                ('device_pointers', C.c_int32), ('reserved', C.c_int32)]


# every symbol include/dpenv.h declares: name -> (restype, argtypes)
_VP, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    'dpenv_abi_version': (C.c_int, []),
    'dpenv_default_config': (C.c_int, [C.POINTER(Config)]),
    'dpenv_default_vessel': (C.c_int, [C.POINTER(C.c_float)]),
    'dpenv_default_vessel_ex': (C.c_int, [C.c_int32, C.POINTER(C.c_float)]),
    'dpenv_act_dim': (C.c_int, [C.POINTER(Config)]),
    'dpenv_obs_dim': (C.c_int, [C.POINTER(Config)]),
    'dpenv_create': (C.c_int, [C.POINTER(Config), C.POINTER(C.c_float), _I32, C.POINTER(_VP)]),
    'dpenv_destroy': (C.c_int, [_VP]),
    'dpenv_last_error': (C.c_char_p, [_VP]),
    'dpenv_set_reset_fraction': (C.c_int, [_VP, _F]),
    'dpenv_set_vessel_class': (C.c_int, [_VP, _VP, _VP]),
    'dpenv_set_vessel_params': (C.c_int, [_VP, _VP, _VP]),
    'dpenv_set_vessel_params_ex': (C.c_int, [_VP, _VP, C.c_uint32, _VP]),
    'dpenv_get_vessel_params': (C.c_int, [_VP, _VP, _VP]),
    'dpenv_set_vessel_randomisation': (C.c_int, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float), _VP]),
    'dpenv_set_current': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_set_current_present': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_get_current': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_get_current_mean': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_set_current_randomisation': (C.c_int, [_VP, _VP, _VP, _F, _F, _VP]),
    'dpenv_reset': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    'dpenv_step': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'dpenv_step_ex': (C.c_int, [_VP, C.POINTER(StepIO), _VP]),
    'dpenv_rollout': (C.c_int, [_VP, C.POINTER(RolloutIO), _VP]),
    'dpenv_set_policy': (C.c_int, [_VP, C.POINTER(Mlp), C.POINTER(Mlp), C.POINTER(C.c_float), _F]),
    'dpenv_set_policy_ex': (C.c_int, [_VP, C.POINTER(Mlp), C.POINTER(Mlp), C.POINTER(C.c_float), C.c_int32, _F]),
    'dpenv_set_policy_desc': (C.c_int, [_VP, C.POINTER(PolicyDesc), _VP]),
    'dpenv_policy_forward': (C.c_int, [_VP, _VP, _VP, _VP, _I32, _VP]),
    'dpenv_policy_rollout': (C.c_int, [_VP, C.POINTER(PolicyRolloutIO), _VP]),
    'dpenv_get_state': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_set_state': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_get_rng_counters': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_set_rng_counters': (C.c_int, [_VP, _VP, _VP, _VP]),
    'dpenv_get_obs_thrust': (C.c_int, [_VP, _VP, _VP]),
    'dpenv_get_policy_launch': (C.c_int, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'dpenv_get_policy_launch_ex': (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    'dpenv_release_policy_graphs': (C.c_int, [_VP]),
    'dpenv_set_obs_thrust': (C.c_int, [_VP, _VP, _VP]),
    'dpenv_thrust_map': (C.c_int, [C.POINTER(C.c_float), _VP, _VP, _VP, _I32, _VP]),
    'dpenv_gae': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _F, _F, _VP, _VP, _VP]),
    'dpenv_gae_workspace_bytes': (C.c_int64, [_I32]),
    'dpenv_gae_stats': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _F, _F, _VP, _VP, _VP, _VP, _VP]),
    'dpenv_adv_apply_stats': (C.c_int, [_VP, _I64, _VP, C.c_double, _VP]),
    'dpenv_adv_sum': (C.c_int, [_VP, _I64, _VP, _VP]),
    'dpenv_adv_sumsq': (C.c_int, [_VP, _I64, _VP, _VP, _VP]),
    'dpenv_adv_apply': (C.c_int, [_VP, _I64, _VP, _VP, _VP]),
}

_lib = None


def load():
    """Load libdpenv.so; raise (never fall back) if it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('libdpenv.so not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                          'or `make -C ml4ca_amd/csrc` (hipcc, gfx950). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    if lib.dpenv_abi_version() != ABI_VERSION:
        raise ImportError('libdpenv.so ABI %d != binding ABI %d' % (lib.dpenv_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class DpenvError(RuntimeError):
    pass


def check(code, handle=None):
    if code != OK:
        msg = load().dpenv_last_error(handle)
        raise DpenvError('libdpenv error %d: %s' % (code, msg.decode() if msg else '?'))


def default_config():
    cfg = Config()
    check(load().dpenv_default_config(C.byref(cfg)))
    return cfg


VESSEL_NO_LOSS, VESSEL_THRUST_LOSS, VESSEL_DYNPOS_FIT = 0, 1, 2        # DYNPOS_FIT is a flag: 3 = both


def default_vessel(kind='no_loss'):
    """Parameter vector of a preset of the build-owned plant: 'no_loss' (the default hull) or 'thrust_loss' (dpenv.h: DPENV_VESSEL_*)."""
    import numpy as np
    p = (C.c_float * NPARAM)()
    check(load().dpenv_default_vessel_ex({'no_loss': VESSEL_NO_LOSS, 'thrust_loss': VESSEL_THRUST_LOSS, 'dynpos_fit': VESSEL_DYNPOS_FIT,
                                         'dynpos_fit_thrust_loss': VESSEL_DYNPOS_FIT | VESSEL_THRUST_LOSS}[kind], p))
    return np.array(p[:], dtype=np.float32)
