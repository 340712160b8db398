#!/usr/bin/env python3
"""Fixture that pins the network GRAPH to the reference's own file.

    python3 -B tests/golden/gen_final_graph.py            (build container only: needs /root/reference)

Reads src/rl/windows_workspace/data/finalmodel/finconttothighbowder_s0/tf1_save/saved_model.pb - the GraphDef that
tf.saved_model.simple_save wrote for the thesis' final policy (spinup/utils/logx.py:161-228) - with ml4ca_amd/tf_graph.py (no
TensorFlow anywhere) and writes DATA only:

tests/golden/final_graph.json
  * `forward_nodes`: every node the five tensors {pi/dense_3/BiasAdd (the tensor test_policy.py:90 feeds to the env), pi/add (the
    sample, the signature's 'pi'), pi/Sum (logp of a given action), pi/Sum_1 (logp of the sample), v/Squeeze} depend on:
    name, op, inputs, attributes (constants as numbers) - 112 of the file's 19 898 nodes (the rest is optimiser, gradients, saver);
  * `signature` (the serving signature's tensor names), `describe` (per layer: op order, leaky-relu form and alpha),
    `scalar_float_constants` of the model scopes (1e-8, log 2 pi as the float32 the graph holds, the PPO clip bounds ...);
tests/golden/final_graph_vectors.npz
  * 64 observations x actions and what the reference's graph computes for them - mu, v, logp - evaluated by tf_graph.evaluate on the
    checkpoint's variables in float64 (the real-number function) and in float32 (the graph's own arithmetic type).

What this pins: topology, activation form and slope, likelihood constants and output tensors come from the reference's file, and
oracle/policy_ref.py is checked against an execution of that very graph (tests/test_policy_import_cpu.py).  What it cannot pin:
TensorFlow's kernels themselves (its MatMul summation order) - the fp32 graph evaluation here uses NumPy's.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
MODEL = '/root/reference/src/rl/windows_workspace/data/finalmodel/finconttothighbowder_s0/tf1_save'
FETCH = ['pi/dense_3/BiasAdd', 'pi/add', 'pi/Sum', 'pi/Sum_1', 'v/Squeeze']


def jsonable(v):
    if isinstance(v, np.ndarray):
        return {'dtype': str(v.dtype), 'shape': list(v.shape), 'data': v.reshape(-1).tolist()}
    if isinstance(v, tuple):
        return list(v)
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    return v


def main():
    from ml4ca_amd import tf_graph as G
    from ml4ca_amd import tf_checkpoint as TC
    g = G.read_saved_model(os.path.join(MODEL, 'saved_model.pb'))
    names = G.ancestors(g, FETCH)
    fwd = {}
    for n in names:
        nd = g.nodes[n]
        attr = {k: jsonable(v) for k, v in nd['attr'].items() if k in ('value', 'alpha', 'transpose_a', 'transpose_b', 'keep_dims', 'squeeze_dims', 'T', 'dtype')}
        fwd[n] = {'op': nd['op'], 'inputs': nd['inputs'], 'attr': attr}
    desc = G.describe_actor_critic(g)
    consts = {k: v for k, v in desc.pop('scalar_float_constants').items()
              if not k.startswith(('gradients', 'save')) and 'Initializer' not in k}
    ops = {}
    for n in g.order:
        ops[g.nodes[n]['op']] = ops.get(g.nodes[n]['op'], 0) + 1
    rec = {'source': 'src/rl/windows_workspace/data/finalmodel/finconttothighbowder_s0/tf1_save/saved_model.pb',
           'generator': 'tests/golden/gen_final_graph.py (ml4ca_amd/tf_graph.py; no TensorFlow)',
           'nodes_in_file': len(g.order), 'op_histogram_of_file': ops,
           'signature': g.signature, 'fetches': FETCH, 'order': names, 'forward_nodes': fwd,
           'describe': desc, 'scalar_float_constants': consts,
           'hidden_activation': list(G.hidden_activation(desc)),
           'tensor_fed_to_the_env_by_test_policy_py_90': 'pi/dense_3/BiasAdd'}
    json.dump(rec, open(os.path.join(OUT, 'final_graph.json'), 'w'), indent=0, sort_keys=True)

    var = TC.read_bundle(os.path.join(MODEL, 'variables', 'variables'))
    rng = np.random.RandomState(20261004)
    obs = (rng.standard_normal((64, 9)) * np.array([3, 3, 0.4, 0.5, 0.15, 0.2, 0.5, 0.5, 0.5])).astype(np.float32)
    out = {'obs': obs}
    for tag, dt in (('f64', np.float64), ('f32', np.float32)):
        mu, v = G.evaluate(g.nodes, ['pi/dense_3/BiasAdd', 'v/Squeeze'], {'Placeholder': obs}, var, dtype=dt)
        if tag == 'f64':
            act = (mu + np.exp(var['pi/log_std'].astype(np.float64)) * rng.standard_normal(mu.shape)).astype(np.float32)
            out['act'] = act
        logp, = G.evaluate(g.nodes, ['pi/Sum'], {'Placeholder': obs, 'Placeholder_1': out['act']}, var, dtype=dt)
        xi = rng.standard_normal(mu.shape) if tag == 'f64' else out['xi']
        out['xi'] = np.asarray(xi, np.float64)
        pi, logp_pi = G.evaluate(g.nodes, ['pi/add', 'pi/Sum_1'], {'Placeholder': obs}, var, dtype=dt, rng_normal=lambda shape: np.asarray(xi).reshape(shape))
        out.update({'mu_' + tag: mu, 'v_' + tag: v, 'logp_' + tag: logp, 'pi_' + tag: pi, 'logp_pi_' + tag: logp_pi})
    np.savez_compressed(os.path.join(OUT, 'final_graph_vectors.npz'), **out)
    print('wrote final_graph.json (%d forward nodes of %d) and final_graph_vectors.npz' % (len(names), len(g.order)))
    print('hidden activation', rec['hidden_activation'], 'constants', {k: v for k, v in consts.items() if k.startswith('pi/')})


if __name__ == '__main__':
    main()
