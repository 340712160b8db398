"""Shared fixtures of the GPU parity tests: build the HIP env and the fp32 oracle from one spec,
draw plausible random states, and move data between the two."""
import numpy as np

from oracle import oracle as O

MODES = {  # name -> (env variant string, cont_ang)
    'full': ('full', False), 'simple': ('simple', False), 'limited': ('limited', False),
    'final_wrap': ('final', False), 'final_cont': ('final', True),
}


def make_pair(mode, n, ext=True, dtype=np.float32, device='cuda:0', **kw):
    """Returns (BatchedRevoltEnv, Oracle) configured identically."""
    import ml4ca_amd
    variant, cont = MODES[mode]
    ovar, ocont = O.MODES[mode]
    env = ml4ca_amd.BatchedRevoltEnv(
        n, variant=variant, extended_state=ext, cont_ang=cont, device=device,
        auto_reset=kw.get('auto_reset', False), terminate=kw.get('terminate', True),
        wrap_mode=kw.get('wrap_mode', 'reference'), seed=kw.get('seed', 0), env_id_base=kw.get('env_id_base', 0),
        obs_dtype=kw.get('obs_dtype', 'float32'), current=kw.get('current', False),
        vessel_params=kw.get('vessel_params'), layout=kw.get('layout', 'aos'),
        reset_fraction=kw.get('reset_fraction', 0.8), time_limit=kw.get('time_limit', True),
        max_ep_len=kw.get('max_ep_len', 800), hold_plant=kw.get('hold_plant', False),
        current_drift=kw.get('current_drift', False), current_tau=kw.get('current_tau', 100.0),
        current_sigma_v=kw.get('current_sigma_v', 0.02), current_sigma_beta=kw.get('current_sigma_beta', 5.0 * np.pi / 180.0),
        n_steps=kw.get('n_steps'), testing=kw.get('testing', False), realtime=kw.get('realtime', False),
        reset_acts=kw.get('reset_acts', False), step_one_wave=kw.get('step_one_wave', False), per_env_lds=kw.get('per_env_lds', False))
    cfg = O.make_config(variant=ovar, extended_state=int(ext), cont_ang=ocont, n_substeps=env.n_steps,
                        wrap_mode=O.WRAP_RADIANS if kw.get('wrap_mode') == 'radians' else O.WRAP_REFERENCE,
                        terminate=int(kw.get('terminate', True)),
                        max_ep_len=env.max_ep_len if kw.get('time_limit', True) else 0,
                        auto_reset=int(kw.get('auto_reset', False)), current_enabled=int(kw.get('current', False)),
                        seed=kw.get('seed', 0), env_id_base=kw.get('env_id_base', 0),
                        reset_fraction=kw.get('reset_fraction', 0.8), current_drift=int(kw.get('current_drift', False)),
                        current_tau=kw.get('current_tau', 100.0), current_sigma_v=kw.get('current_sigma_v', 0.02),
                        current_sigma_beta=kw.get('current_sigma_beta', 5.0 * np.pi / 180.0), reset_acts=int(kw.get('reset_acts', False)))
    vessel = None
    if kw.get('vessel_params') is not None and np.asarray(kw['vessel_params']).ndim == 1:
        vessel = np.asarray(kw['vessel_params'], dtype)
    orc = O.Oracle(cfg, dtype, vessel=vessel)
    return env, orc


def random_state(rng, n, spread=1.0):
    """Plausible mid-episode states (float32-representable), canonical [15][n] layout."""
    st = np.zeros((O.NSTATE, n), np.float32)
    st[0:2] = rng.uniform(-7, 7, size=(2, n)) * spread
    st[2] = rng.uniform(-0.7, 0.7, size=n) * spread
    st[3] = rng.uniform(-1.2, 1.2, size=n)
    st[4] = rng.uniform(-0.28, 0.28, size=n)
    st[5] = rng.uniform(-0.45, 0.45, size=n)
    st[6:8] = rng.uniform(-3, 3, size=(2, n)) * (rng.uniform(size=n) < 0.5)
    st[8] = rng.uniform(-0.3, 0.3, size=n) * (rng.uniform(size=n) < 0.5)
    st[9:12] = rng.uniform(-100, 100, size=(3, n))
    st[12:15] = rng.uniform(-np.pi, np.pi, size=(3, n))
    return st


def random_actions(rng, n, act_dim, scale=0.8):
    return rng.normal(0.0, scale, size=(n, act_dim)).astype(np.float32)


def to_dev(x, device='cuda:0'):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def random_hulls(rng, n, rel=0.15, base=None, loss=0.0):
    """[NPARAM, n] float32: every env its own parameter vector, each of the 26 parameters of the default hull scaled by an
    independent factor in [1 - rel, 1 + rel] (the constants the reference fixes once for its one vessel: qp_allocator.py:51-55,69-70,
    SupervisedTau.py:35-36,69-71, and the build-owned mass / damping terms).  loss > 0: the six inflow thrust-loss coefficients
    (rows 26..31) uniform in [0, loss) as well - every fourth env keeps none."""
    import ml4ca_amd
    base = np.asarray(ml4ca_amd.default_vessel() if base is None else base, np.float32)
    tab = np.zeros((O.NPARAM, n), np.float32)
    tab[:26] = base[:26, None] * (1.0 + rel * rng.uniform(-1, 1, size=(26, n))).astype(np.float32)
    if loss > 0:
        tab[26:32] = (loss * rng.uniform(0, 1, size=(6, n))).astype(np.float32)
        tab[26:32, ::4] = 0.0
    return tab


def rand_table(rel, nominal=None):
    """[nominal (32) | relative half-range (32)] as the library holds it (float32)"""
    import ml4ca_amd
    rt = np.zeros(2 * O.NPARAM, np.float32)
    rt[:O.NPARAM] = np.asarray(ml4ca_amd.default_vessel() if nominal is None else nominal, np.float32)
    rt[O.NPARAM:2 * O.NPARAM] = np.float32(rel)
    return rt
