#!/usr/bin/env python3
"""Fixtures for SURVEY section 8 row f-2: the reference's OWN evaluation harness and energy metric.

    python3 -B tests/golden/gen_run_rl_policy.py            (build container only: needs /root/reference)

(1) tests/golden/run_rl_policy.npz - `run_RL_policy` of spinup/utils/test_policy.py:97-186, imported read-only (behind stub
    modules for joblib / tensorflow / spinup.utils.logx's own tensorflow + mpi imports, which the function never touches on this
    path) and executed as it is: it takes `get_action` as a callable, so the trained actor of the thesis
    (tests/golden/final_policy.npz, the fixture tools/gen_golden.py extracted from the TF checkpoint bundle) is evaluated by a NumPy
    float64 MLP (core.py:29-33: dense layers, leaky-relu 0.2, deterministic mean as test_policy.py:90 picks it).  The env is the
    reference's RevoltFinal(testing=True, extended_state=True, cont_ang=True) (specific/customEnv.py) on oracle/twin_shim.TwinShim
    - the reference's plant seam over the oracle's float64 plant (the real plant, Cybersea, is closed source: the plant is ours,
    everything around it the reference's).  Recorded per episode and step, with test_setpoint_changes False and True: observation,
    reward, (ned_pos, ned_ref), action_vec, EpRet, EpLen - the three lists the function returns plus what its logger stored.
(2) tests/golden/energy.npz - `power()` of results/all_plots/box_test/plot_act.py:133-135 and the trapezoid of :184-211.  The
    script is a plotting script that reads CSVs at import, so only the `power` function and its four constant tables are taken
    out of its syntax tree (ast: the FunctionDef named power and the assignments to rps_max / diameters / KQ_0 / rho) and executed;
    the trapezoid loop is re-run with that function on the recorded RL box-test commands (tests/golden/cybersea_replay.npz holds
    them) and on a random series.  Data only is written.

Lives under tests/ because it drives oracle/ code (test infrastructure).
"""
import ast
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


class _Any(object):
    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


class _Logger(object):
    """stands in for spinup.utils.logx.EpochLogger (a stdout / file logger): keeps what run_RL_policy stores"""
    last = None

    def __init__(self, *a, **k):
        self.rows = []
        _Logger.last = self

    def store(self, **kw):
        self.rows.append(kw)

    def log_tabular(self, *a, **k):
        pass

    def dump_tabular(self, *a, **k):
        pass


def numpy_actor(path):
    """deterministic mean of mlp_gaussian_policy (core.py:29-33,80-86) in float64 from the extracted checkpoint tensors"""
    d = np.load(path)
    Ws, bs, i = [], [], 0
    while True:
        name = 'pi.dense%s' % ('' if i == 0 else '_%d' % i)          # the fixture stores pi/dense_1/kernel as pi.dense_1.kernel
        if name + '.kernel' not in d.files:
            break
        Ws.append(d[name + '.kernel'].astype(np.float64))
        bs.append(d[name + '.bias'].astype(np.float64))
        i += 1

    def get_action(o):
        x = np.asarray(o, np.float64).reshape(-1)
        for W, b in zip(Ws[:-1], bs[:-1]):
            x = x @ W + b
            x = np.where(x > 0, x, 0.2 * x)                       # tf.nn.leaky_relu default alpha
        return x @ Ws[-1] + bs[-1]

    return get_action


def gen_run_rl_policy():
    from tools import gen_golden as GG
    from oracle.twin_shim import TwinShim
    GG.install_stubs()
    GG.install_tf_mpi_stubs()
    sys.modules['joblib'] = types.ModuleType('joblib')
    logx = types.ModuleType('spinup.utils.logx')
    logx.EpochLogger = _Logger
    logx.restore_tf_graph = _Any()
    sys.path.insert(0, GG.WW)
    import spinup.utils                                             # the package itself (its __init__ is empty)
    sys.modules['spinup.utils.logx'] = logx
    spinup.utils.logx = logx
    import spinup.utils.test_policy as TP
    import specific.customEnv as CE
    get_action = numpy_actor(os.path.join(OUT, 'final_policy.npz'))
    out = {}
    max_ep_len = 400
    # with test_setpoint_changes the reference has five setpoints for the episodes it is asked for (test_policy.py:127): a sixth
    # episode raises IndexError at :149, so that run has five
    for tag, changes, num_episodes in (('plain', False, 6), ('setpoints', True, 5)):
        twin = TwinShim()
        env = CE.RevoltFinal(twin, testing=True, extended_state=True, cont_ang=True)
        np.random.seed(7)                                           # nothing random is drawn on this path (fixed points); belt and braces
        data, ned_pos, action_data = TP.run_RL_policy(env, get_action, max_ep_len=max_ep_len, num_episodes=num_episodes, render=False,
                                                      test_setpoint_changes=changes)
        rows = _Logger.last.rows
        assert len(data) == num_episodes and len(rows) == num_episodes
        L = max(len(e) for e in data)
        obs = np.full((num_episodes, L, 9), np.nan)
        rew = np.full((num_episodes, L), np.nan)
        pos = np.full((num_episodes, L, 3), np.nan)
        ref = np.full((num_episodes, L, 3), np.nan)
        vec = np.full((num_episodes, L, 6), np.nan)
        n_rec = np.zeros(num_episodes, np.int64)
        for k in range(num_episodes):
            n_rec[k] = len(data[k])
            assert len(ned_pos[k]) == n_rec[k] and len(action_data[k]) == n_rec[k]
            for t in range(n_rec[k]):
                obs[k, t] = np.asarray(data[k][t][0], np.float64).ravel()
                rew[k, t] = float(np.asarray(data[k][t][1]).ravel()[0])
                pos[k, t] = np.asarray(ned_pos[k][t][0], np.float64).ravel()
                ref[k, t] = np.asarray(ned_pos[k][t][1], np.float64).ravel()
                vec[k, t] = np.asarray(action_data[k][t], np.float64).ravel()
        out[tag + '_obs'], out[tag + '_rew'], out[tag + '_ned_pos'], out[tag + '_ned_ref'], out[tag + '_action_vec'] = obs, rew, pos, ref, vec
        out[tag + '_n_recorded'] = n_rec
        out[tag + '_EpRet'] = np.array([float(np.asarray(r['EpRet']).ravel()[0]) for r in rows])
        out[tag + '_EpLen'] = np.array([int(r['EpLen']) for r in rows])
        print(tag, 'EpLen', out[tag + '_EpLen'], 'EpRet', np.round(out[tag + '_EpRet'], 2))
    out['max_ep_len'] = np.int64(max_ep_len)
    out['vessel'] = TwinShim().vessel
    np.savez_compressed(os.path.join(OUT, 'run_rl_policy.npz'), **out)


def reference_power():
    """the FunctionDef `power` and its constant tables out of the plotting script's syntax tree (the script itself cannot be
    imported: it reads CSVs and opens figures at module level)"""
    src = open(os.path.join(REF, 'results/all_plots/box_test/plot_act.py')).read()
    tree = ast.parse(src)
    want = {'rps_max', 'diameters', 'KQ_0', 'rho'}
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == 'power':
            keep.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id in want:
            keep.append(node)
    assert {getattr(n, 'name', None) or n.targets[0].id for n in keep} == want | {'power'}
    ns = {'np': np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), 'plot_act.py[power]', 'exec'), ns)
    return ns['power'], {k: ns[k] for k in want}


def gen_energy():
    power, consts = reference_power()

    def cumulative_work(t, n, which):
        # plot_act.py:184-211: work_elements then np.cumsum
        el = []
        for j in range(len(t) - 1):
            dt = t[j + 1] - t[j]
            el.append((power(n[j + 1], which) + power(n[j], which)) / 2 * dt)
        return np.cumsum(el)

    rng = np.random.RandomState(11)
    T = 300
    t = np.cumsum(rng.uniform(0.15, 0.25, size=T))
    n = rng.uniform(-100, 100, size=(T, 3))
    out = dict(t_rand=t, n_rand=n, p_rand=np.stack([power(n[:, 0], 'bow'), power(n[:, 1], 'stern'), power(n[:, 2], 'stern')], 1),
               w_rand=np.stack([cumulative_work(t, n[:, 0], 'bow'), cumulative_work(t, n[:, 1], 'stern'), cumulative_work(t, n[:, 2], 'stern')], 1))
    # uniform 5 Hz grid (what the rollout blocks are): 400 steps of a smooth command
    tg = np.arange(401) * 0.2
    ng = 60.0 * np.sin(tg[:, None] / np.array([7.0, 11.0, 13.0])) + np.array([10.0, -20.0, 5.0])
    out.update(t_grid=tg, n_grid=ng,
               w_grid=np.stack([cumulative_work(tg, ng[:, 0], 'bow'), cumulative_work(tg, ng[:, 1], 'stern'), cumulative_work(tg, ng[:, 2], 'stern')], 1))
    for k, v in consts.items():
        out['const_' + k] = np.array([v['bow'], v['stern']]) if isinstance(v, dict) else np.float64(v)
    np.savez_compressed(os.path.join(OUT, 'energy.npz'), **out)
    print('energy: W* of the random series', out['w_rand'][-1], 'of the grid series', out['w_grid'][-1])


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('all', 'energy'):
        gen_energy()
    if what in ('all', 'run'):
        gen_run_rl_policy()
