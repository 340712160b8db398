"""CPU tests of the trained-policy import path: tensor-bundle reader/writer, ActorCritic construction from the
reference's variable names, fp32 torch forward pass against the committed float64 expectations of the shipped
model, and the evaluation metrics."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, 'tests', 'golden', 'final_policy.npz')
REF_CKPT = '/root/reference/src/rl/windows_workspace/data/finalmodel/finconttothighbowder_s0/tf1_save/variables/variables'


def fixture_tensors():
    d = np.load(FIX)
    return {k.replace('.', '/'): d[k] for k in d.files if '.' in k}, d


def test_bundle_roundtrip(tmp_path):
    from ml4ca_amd.tf_checkpoint import read_bundle, write_bundle
    rng = np.random.RandomState(0)
    t = {'pi/dense/kernel': rng.normal(size=(9, 80)).astype(np.float32), 'pi/dense/bias': rng.normal(size=80).astype(np.float32),
         'pi/log_std': rng.normal(size=7).astype(np.float32), 'beta1_power': np.float32(0.5).reshape(()),
         'global_step': np.array(7, np.int64), 'v/dense_3/kernel': rng.normal(size=(80, 1)).astype(np.float32)}
    write_bundle(str(tmp_path / 'variables'), t)
    r = read_bundle(str(tmp_path / 'variables'))
    assert sorted(r) == sorted(t)
    for k in t:
        assert r[k].shape == np.asarray(t[k]).shape and np.array_equal(r[k], t[k]), k
    with pytest.raises(ValueError):
        (tmp_path / 'bad.index').write_bytes(b'\\x00' * 64)
        (tmp_path / 'bad.data-00000-of-00001').write_bytes(b'')
        read_bundle(str(tmp_path / 'bad'))


@pytest.mark.skipif(not os.path.exists(REF_CKPT + '.index'), reason='reference tree only exists in the build container')
def test_reader_on_the_reference_checkpoint_matches_fixture():
    from ml4ca_amd.tf_checkpoint import read_bundle
    b = read_bundle(REF_CKPT)
    assert sum(v.size for v in b.values()) == 84529          # 28 175 parameters + Adam slots
    t, _ = fixture_tensors()
    assert sum(v.size for v in t.values()) == 28175
    for k, v in t.items():
        assert np.array_equal(b[k], v), k


def test_actor_critic_from_reference_names_and_forward():
    import torch
    from ml4ca_amd.policy import ActorCritic
    t, d = fixture_tensors()
    ac = ActorCritic.from_tensors(t)
    assert (ac.obs_dim, ac.act_dim, ac.hidden_sizes, ac.leak) == (9, 7, (80, 80, 80), 0.2)   # config.json of the shipped run
    mu, v = ac.forward_ref(torch.tensor(d['obs'], dtype=torch.float32))
    assert np.allclose(mu.numpy(), d['mu'], rtol=2e-5, atol=2e-5)
    assert np.allclose(v.numpy(), d['v'], rtol=2e-5, atol=2e-3)
    # on the setpoint at rest the critic predicts ~ the discounted sum of the 3.5/step maximum reward
    assert 250 < float(v[0]) < 350
    sd = ac.state_dict()
    assert all(np.array_equal(sd[k], t[k]) for k in t)
    with pytest.raises(ValueError):
        ActorCritic.from_tensors({'x': np.zeros(3)})


def test_evaluation_metrics():
    import torch
    from ml4ca_amd import evaluate as EV
    T, n = 11, 3
    obs = torch.zeros((T, n, 9))
    obs[:, 0, 0] = 5.0                       # constant 5 m error -> normalised error 1 -> IAE = duration
    obs[:, 1, 2] = float(np.deg2rad(25.0))   # constant 25 deg heading error -> 1
    tot, cum = EV.iae(obs, dt=0.2)
    assert np.allclose(tot.numpy(), [2.0, 2.0, 0.0], atol=1e-6) and cum.shape == (T, n)
    # the reference's own formula on the same series (common.py:56-74)
    t = np.arange(T) * 0.2
    e = np.sqrt(((obs[:, 0, :3].numpy() / np.array([5, 5, np.deg2rad(25)])) ** 2).sum(1))
    assert abs(sum((e[i] + e[i + 1]) / 2 * (t[i + 1] - t[i]) for i in range(T - 1)) - 2.0) < 1e-6
    thr = torch.zeros((T, n, 3))
    thr[:, 0, 0] = 100.0
    thr[:, 1, 1] = -50.0
    w = EV.work(thr, dt=0.2)
    p_bow = 0.02 * 2 * np.pi * 1025 * 0.06 ** 5 * 33.0 ** 3          # plot_act.py:128-135 at n = 100 %
    # sgn(n) * (n ...)^3 is |.|^3: the reference's power is positive for reverse thrust as well
    p_port = 0.036 * 2 * np.pi * 1025 * 0.15 ** 5 * (0.5 * 11.0) ** 3
    assert np.allclose(w[0].numpy(), [p_bow * 2.0, 0, 0], rtol=1e-5) and np.allclose(w[1].numpy(), [0, p_port * 2.0, 0], rtol=1e-5)
    steps, refs = EV.box_schedule(torch.zeros((3, 2)))
    assert steps == (50, 300, 550, 700, 950) and refs.shape == (5, 3, 2)
    assert np.allclose(refs[2, :, 0].numpy(), [5.0, -5.0, -np.pi / 4])
    assert np.allclose(EV.commanded_thrust(torch.tensor([[[2.0, -0.5, 0.1]]])).numpy(), [[[100.0, -50.0, 10.0]]])


@pytest.mark.parametrize('kind', ['plain', 'integral'])
def test_ros_node_adapter_against_the_imported_node(kind):
    """src/rl/ROS/rl_allocator/src/rl_allocator.py run behind ROS stubs (tools/gen_golden.py rosnode[_integral]) with the
    trained actor: 420 callback rounds - approach, dwell near the setpoint (the integral action winds up to its
    bounds), a kick outside the 5 m box (reset), a setpoint change.  ml4ca_amd.deploy.RLAllocatorNode must reproduce
    the node's state vector, its command vector in ROS order, the integrator and every published message field."""
    from ml4ca_amd import deploy as DP
    d = np.load(os.path.join(G, 'ros_rl_node_%s.npz' % kind))
    pol = np.load(os.path.join(G, 'final_policy.npz'))
    Wb = [(pol['pi.dense%s.kernel' % k].astype(np.float64), pol['pi.dense%s.bias' % k].astype(np.float64)) for k in ('', '_1', '_2', '_3')]

    def actor(x):
        for i, (w, b) in enumerate(Wb):
            x = x @ w + b
            if i < 3:
                x = np.where(x > 0, x, 0.2 * x)
        return x

    node = DP.RLAllocatorNode(actor, variant='final', cont_ang=True, integrator=(kind == 'integral'), simulation=True,
                              now=float(d['t0'][0]))
    T = d['pose'].shape[0]
    for k in range(T):
        now = float(d['t'][k])
        node.on_eta(d['pose'][k, 0], d['pose'][k, 1], d['pose'][k, 2], now)
        node.on_nu(*d['nu'][k])
        u, msg = node.on_reference(d['ref'][k, 0], d['ref'][k, 1], d['ref'][k, 2], now)
        assert abs(node.h - d['h'][k]) < 1e-9
        assert np.allclose(node.state, d['state'][k], rtol=0, atol=1e-9), (k, np.abs(node.state - d['state'][k]).max())
        assert np.allclose(u, d['u'][k], rtol=0, atol=1e-9), k
        if node.integrator is not None:
            assert np.allclose(node.integrator.value, d['integ'][k], rtol=0, atol=1e-12), k
        assert np.allclose([msg['pod_angle.port'], msg['pod_angle.star']], d['pod'][k], rtol=0, atol=1e-9)
        assert np.allclose([msg['stern.port_effort'], msg['stern.star_effort']], d['stern'][k], rtol=0, atol=1e-9)
        assert np.allclose([msg['bow.throttle_bow'], msg['bow.position_bow'], msg['bow.lin_act_bow']], d['bow'][k], rtol=0, atol=1e-9)
    if kind == 'integral':
        assert np.allclose(np.abs(d['integ']).max(0), DP.BodyFrameIntegrator.BOUND)      # wound up to every bound
        assert (np.abs(d['integ'][262:300]).max() == 0.0)                                 # and reset by the kick
    else:
        assert np.abs(d['integ']).max() == 0.0


def test_ros_order_maps_and_hardware_bow_mapping():
    from ml4ca_amd import deploy as DP
    # limited: 5 outputs, stern angles scale to +-pi/2, bow fixed at pi/2 (rl_allocator.py:92-106)
    u = DP.to_ros_order(np.array([0.5, -0.25, 2.0, 1.0, -0.5]), 'limited', cont_ang=False)
    assert np.allclose(u, [-25.0, 100.0, 50.0, np.pi / 2, -np.pi / 4, np.pi / 2])
    # full: 6 outputs, a_bow is the network's 4th output
    u = DP.to_ros_order(np.array([0.1, 0.2, 0.3, 0.5, -0.5, 0.25]), 'full', cont_ang=False)
    assert np.allclose(u, [20.0, 30.0, 10.0, -np.pi / 2, np.pi / 4, np.pi / 2])
    # batched input keeps its leading shape
    assert DP.to_ros_order(np.zeros((4, 3, 7)), 'final', True).shape == (4, 3, 6)
    # on the vessel the bow thruster gets 2.5x the command, clipped, at a fixed 45 % position (utils.py:112-113)
    m = DP.publishable(np.array([10.0, -10.0, 60.0, 0.1, -0.1, np.pi / 2]), simulation=False)
    assert m['bow.throttle_bow'] == 100.0 and m['bow.position_bow'] == 45
    assert abs(DP.shortest_path(np.radians(170), np.radians(-170)) - np.radians(20)) < 1e-12
    with pytest.raises(ValueError):
        DP.RLAllocatorNode(lambda s: s, variant='simple')


def test_iae_matches_the_reference_function_on_the_recorded_box_test():
    """evaluate.iae_series against tests/golden/iae.npz: the reference's own IAE (results/all_plots/common.py:60-74, imported by
    tools/gen_golden_iae.py) on the recorded Cybersea RL box test prepared as box_test/plot_pos.py does (per-second averages,
    normalisation [5, 5, 25]), and on an irregular time base.  float64: identical formulae, 1e-12."""
    import torch
    from ml4ca_amd import evaluate as EV
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'iae.npz'))
    integ, cum = EV.iae_series(torch.from_numpy(d['eta']), torch.from_numpy(d['ref']), torch.from_numpy(d['time']), norm=tuple(d['norm']))
    assert np.abs(integ.numpy() - d['integrals']).max() < 1e-12 and np.abs(cum.numpy() - d['cumsum']).max() < 1e-11
    assert abs(float(cum[-1]) - 58.1992) < 1e-3                       # the thesis' RL box-test figure from the record
    integ2, cum2 = EV.iae_series(torch.from_numpy(d['a2']), torch.from_numpy(d['b2']), torch.from_numpy(d['t2']), norm=(1.0, 1.0, 1.0))
    assert np.abs(integ2.numpy() - d['integrals2']).max() < 1e-12 and np.abs(cum2.numpy() - d['cumsum2']).max() < 1e-10
    # the [T, n] batched form used on rollout blocks = the series form per env
    T, n = 50, 7
    rng = np.random.RandomState(1)
    obs = torch.from_numpy(rng.normal(size=(T, n, 9)).astype(np.float32))
    tot, cumb = EV.iae(obs, dt=0.2)
    for i in range(n):
        e = torch.stack([obs[:, i, 0], obs[:, i, 1], torch.rad2deg(obs[:, i, 2])], -1).double()
        _, c = EV.iae_series(e, torch.zeros_like(e), torch.arange(T, dtype=torch.float64) * 0.2)
        assert abs(float(tot[i]) - float(c[-1])) < 1e-4 * float(c[-1])


def test_energy_metric_matches_the_reference_power_fixture():
    """SURVEY 8 f-2 / VERDICT r02 item 4: evaluate.thruster_power / evaluate.work against tests/golden/energy.npz, produced by the
    reference's own power() (results/all_plots/box_test/plot_act.py:133-135, taken out of the plotting script's syntax tree by
    tests/golden/gen_run_rl_policy.py) and its trapezoid (:184-211): a random command series on a non-uniform time base and a
    smooth one on the 5 Hz grid the rollout blocks have.  float64, 1e-12 relative."""
    import torch
    from ml4ca_amd import evaluate as EV
    d = np.load(os.path.join(G, 'energy.npz'))
    assert np.allclose(d['const_rps_max'], [EV.RPS_MAX['bow'], EV.RPS_MAX['stern']]) and np.allclose(d['const_diameters'], [EV.DIAMETER['bow'], EV.DIAMETER['stern']])
    assert np.allclose(d['const_KQ_0'], [EV.KQ0['bow'], EV.KQ0['stern']]) and float(d['const_rho']) == EV.RHO
    n = torch.from_numpy(d['n_rand'])
    p = np.stack([EV.thruster_power(n[:, 0], 'bow').numpy(), EV.thruster_power(n[:, 1], 'stern').numpy(), EV.thruster_power(n[:, 2], 'stern').numpy()], 1)
    assert np.abs(p - d['p_rand']).max() <= 1e-12 * np.abs(d['p_rand']).max()
    cum = EV.work(n, time=torch.from_numpy(d['t_rand']), cumulative=True).numpy()
    assert cum.shape == d['w_rand'].shape and np.abs(cum - d['w_rand']).max() <= 1e-12 * np.abs(d['w_rand']).max()
    tot = EV.work(torch.from_numpy(d['n_grid'])[:, None, :], dt=0.2).numpy()[0]
    assert np.abs(tot - d['w_grid'][-1]).max() <= 1e-12 * np.abs(d['w_grid'][-1]).max()


def test_policy_yardstick_reproduces_the_checkpoint_fixture():
    """oracle/policy_ref.py (the float64 restatement of core.py:29-46,80-107 that the network-arithmetic tests measure against) on the
    thesis' shipped checkpoint: mu and v of tests/golden/final_policy.npz (an independent NumPy evaluation made when the checkpoint
    bundle was read, tools/gen_golden.py), and the log-likelihood formula against a direct evaluation."""
    from oracle import policy_ref as PR
    d = np.load(os.path.join(G, 'final_policy.npz'))
    params = {k.replace('.', '/'): d[k] for k in d.files if '.' in k}
    mu, v = PR.actor_critic(params, d['obs'])
    assert np.abs(mu - d['mu']).max() < 1e-5 * (np.abs(d['mu']).max() + 1) and np.abs(v - d['v']).max() < 1e-5 * (np.abs(d['v']).max() + 1)
    rng = np.random.RandomState(0)
    x = mu + np.exp(params['pi/log_std']) * rng.standard_normal(mu.shape)
    lp = PR.gaussian_likelihood(x, mu, params['pi/log_std'])
    z = (x - mu) / (np.exp(params['pi/log_std'].astype(np.float64)) + PR.LIKELIHOOD_EPS)
    want = (-0.5 * z * z - params['pi/log_std'] - 0.5 * PR.LOG_2PI).sum(1)
    assert np.abs(lp - want).max() < 1e-12
    # the constants are the float32 values the reference's graph holds (test below); with the double-precision log(2 pi) the seven
    # action dimensions would differ by 3.5 x 3.1e-8
    want64 = (-0.5 * z * z - params['pi/log_std'] - 0.5 * np.log(2 * np.pi)).sum(1)
    assert 0.9e-7 < np.abs(lp - want64).max() < 1.3e-7
    # relu = leaky with slope 0, tanh: the other --activation choices (train.py:24,31)
    assert np.all(PR.mlp(np.array([[-1.0]]), [np.eye(1), np.eye(1)], [np.zeros(1), np.zeros(1)], 'relu') == 0.0)
    assert abs(PR.mlp(np.array([[0.5]]), [np.eye(1), np.eye(1)], [np.zeros(1), np.zeros(1)], 'tanh')[0, 0] - np.tanh(0.5)) < 1e-15


# ---- the network GRAPH pinned to the reference's own saved_model.pb (tests/golden/gen_final_graph.py) -----------------------------
REF_MODEL = '/root/reference/src/rl/windows_workspace/data/finalmodel/finconttothighbowder_s0/tf1_save'


def graph_fixture():
    import json
    rec = json.load(open(os.path.join(G, 'final_graph.json')))
    nodes = {}
    for n, nd in rec['forward_nodes'].items():
        attr = {}
        for k, v in nd['attr'].items():
            attr[k] = np.array(v['data'], dtype=v['dtype']).reshape(v['shape']) if isinstance(v, dict) else v
        nodes[n] = {'op': nd['op'], 'inputs': nd['inputs'], 'attr': attr}
    return rec, nodes, np.load(os.path.join(G, 'final_graph_vectors.npz'))


def test_the_saved_graph_is_what_the_library_and_the_yardstick_assume():
    """Facts of the reference's GraphDef (not of config.json, not of prose): layer order, activation form and slope, likelihood
    constants, signature, the tensor test_policy.py:90 feeds to the env."""
    from oracle import policy_ref as PR
    rec, nodes, _ = graph_fixture()
    assert rec['nodes_in_file'] == 19898 and len(nodes) == 112
    assert rec['signature'] == {'inputs': {'x': 'Placeholder:0'}, 'outputs': {'pi': 'pi/add:0', 'v': 'v/Squeeze:0'}}
    assert rec['tensor_fed_to_the_env_by_test_policy_py_90'] == 'pi/dense_3/BiasAdd' and 'pi/dense_3/BiasAdd' in nodes
    for net, out_units in (('pi', 7), ('v', 1)):
        chain = rec['describe'][net]
        assert [l['layer'] for l in chain] == [net + '/dense', net + '/dense_1', net + '/dense_2', net + '/dense_3']
        assert all(l['ops'] == ['MatMul', 'BiasAdd', 'LeakyRelu'] and l['form'] == 'Maximum(Mul(alpha, x), x)' for l in chain[:-1])
        assert chain[-1]['ops'] == ['MatMul', 'BiasAdd']                                   # no activation behind the output layer
        assert not any(l['transpose_a'] or l['transpose_b'] for l in chain)                # x @ W, W stored [in][out] as the kernels take it
        assert chain[0]['input'] == 'Placeholder'
    # the slope: float32(0.2) - what ActorCritic's default, the kernel's `leak` argument (a float) and the yardstick use
    assert rec['hidden_activation'] == ['leaky', float(np.float32(0.2))]
    assert PR.LEAKY_SLOPE == float(np.float32(0.2))
    c = rec['scalar_float_constants']
    assert c['pi/add_1/y'] == PR.LIKELIHOOD_EPS == float(np.float32(1e-8))                 # core.py:44
    assert c['pi/add_3/y'] == PR.LOG_2PI == float(np.float32(np.log(2 * np.pi)))           # core.py:45
    assert c['pi/mul_2/x'] == -0.5 and c['pi/pow/y'] == 2.0 and c['pi/mul_1/x'] == 2.0
    # the likelihood subgraph is the cited formula: Sum(-0.5 * (((a - mu) / (exp(log_std) + eps))^2 + 2 log_std + log 2 pi), axis 1)
    assert nodes['pi/Sum']['op'] == 'Sum' and nodes['pi/Sum']['inputs'][0] == 'pi/mul_2'
    assert nodes['pi/truediv']['op'] == 'RealDiv' and nodes['pi/truediv']['inputs'] == ['pi/sub', 'pi/add_1']
    assert nodes['pi/sub']['inputs'] == ['Placeholder_1', 'pi/dense_3/BiasAdd'] and nodes['pi/add_1']['inputs'] == ['pi/Exp_1', 'pi/add_1/y']
    # the sample: mu + N(0, 1) * exp(log_std)  (core.py:85)
    assert nodes['pi/add']['inputs'] == ['pi/dense_3/BiasAdd', 'pi/mul'] and nodes['pi/mul']['inputs'] == ['pi/random_normal', 'pi/Exp']
    # PPO clip ratio 0.2 (ppo.py:233: clip_ratio) sits in the loss as 1.2 / 0.8
    assert abs(c['mul/x'] - 1.2) < 1e-6 and abs(c['mul_1/x'] - 0.8) < 1e-6


def test_yardstick_and_torch_reference_against_an_execution_of_the_saved_graph():
    """tf_graph.evaluate runs the reference's own forward graph (fixture nodes) on the checkpoint's variables (final_policy.npz).
    oracle/policy_ref.py - the yardstick of every network-arithmetic claim - must BE that function (1e-12), and the fp32 torch
    reference of ActorCritic must be within 1e-5 of the output scale."""
    import torch
    from ml4ca_amd import tf_graph as TG
    from ml4ca_amd.policy import ActorCritic
    from oracle import policy_ref as PR
    rec, nodes, vec = graph_fixture()
    t, _ = fixture_tensors()
    obs, act, xi = vec['obs'], vec['act'], vec['xi']
    mu, v = TG.evaluate(nodes, ['pi/dense_3/BiasAdd', 'v/Squeeze'], {'Placeholder': obs}, t)
    logp, = TG.evaluate(nodes, ['pi/Sum'], {'Placeholder': obs, 'Placeholder_1': act}, t)
    pi, logp_pi = TG.evaluate(nodes, ['pi/add', 'pi/Sum_1'], {'Placeholder': obs}, t, rng_normal=lambda shape: xi.reshape(shape))
    # the stored vectors were made from the reference's files (bundle + graph) in the build container: same numbers from the fixtures
    for a, k in ((mu, 'mu_f64'), (v, 'v_f64'), (logp, 'logp_f64'), (pi, 'pi_f64'), (logp_pi, 'logp_pi_f64')):
        assert np.array_equal(a, vec[k]), k
    # the yardstick IS the graph's function
    pmu, pv = PR.actor_critic(t, obs)
    assert np.abs(pmu - mu).max() < 1e-12 and np.abs(pv - v).max() < 1e-10 * max(1.0, np.abs(v).max())
    assert np.abs(PR.gaussian_likelihood(act, pmu, t['pi/log_std']) - logp).max() < 1e-9 * np.abs(logp).max()
    assert np.abs(PR.gaussian_likelihood(pi, pmu, t['pi/log_std']) - logp_pi).max() < 1e-9 * np.abs(logp_pi).max()
    # the graph in its own arithmetic type (float32, NumPy's summation order) against its real-number function: the size of what
    # "fp32" leaves open - and the torch fp32 reference, and ActorCritic's defaults, inside 1e-5 of the output scale
    assert np.abs(vec['mu_f32'] - mu).max() < 1e-5 * max(1.0, np.abs(mu).max())
    assert np.abs(vec['v_f32'] - v).max() < 1e-5 * np.abs(v).max()
    ac = ActorCritic.from_tensors(t)
    assert np.float32(ac.leak) == np.float32(rec['hidden_activation'][1]) and ac.activation == 'leaky'
    tmu, tv = ac.forward_ref(torch.tensor(obs))
    assert np.abs(tmu.numpy() - mu).max() < 1e-5 * max(1.0, np.abs(mu).max())
    assert np.abs(tv.numpy() - v).max() < 1e-5 * np.abs(v).max()
    tlp = ac.logp_ref(torch.tensor(act), tmu).numpy()
    assert np.abs(tlp - logp).max() < 2e-5 * np.abs(logp).max()
    # an op outside the forward pass is refused, not guessed
    bad = dict(nodes, extra={'op': 'Conv2D', 'inputs': ['Placeholder'], 'attr': {}})
    with pytest.raises(NotImplementedError):
        TG.evaluate(bad, ['extra'], {'Placeholder': obs}, t)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_MODEL, 'saved_model.pb')), reason='reference tree only exists in the build container')
def test_graph_reader_on_the_reference_file_matches_the_fixture():
    from ml4ca_amd import tf_graph as TG
    from ml4ca_amd.policy import ActorCritic
    rec, nodes, _ = graph_fixture()
    g = TG.read_saved_model(os.path.join(REF_MODEL, 'saved_model.pb'))
    assert len(g.order) == rec['nodes_in_file'] and TG.ancestors(g, rec['fetches']) == rec['order']
    for n in rec['order']:
        assert g.nodes[n]['op'] == nodes[n]['op'] and g.nodes[n]['inputs'] == nodes[n]['inputs'], n
    ac = ActorCritic.from_saved_model(REF_MODEL)
    assert (ac.obs_dim, ac.act_dim, ac.hidden_sizes, ac.activation) == (9, 7, (80, 80, 80), 'leaky') and np.float32(ac.leak) == np.float32(0.2)
    t, _ = fixture_tensors()
    for k, v in ac.state_dict().items():
        assert np.array_equal(v, t[k]), k
