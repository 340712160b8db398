"""TEST INFRASTRUCTURE (oracle/): float64 restatement of the reference's actor-critic, the yardstick of the network-arithmetic
claims (DPENV_POLICY_F32: mu, v, logp within 1e-5 of the output scale).

Follows spinup/algos/tf1/ppo/core.py:
  mlp                   :29-33   dense layers x W + b, hidden activation on all but the last
  gaussian_likelihood   :42-46   -0.5 (((x - mu) / (exp(log_std) + 1e-8))^2 + 2 log_std + log(2 pi)), summed over the action
  mlp_gaussian_policy   :80-92   mu = mlp(x), log_std a free parameter, pi = mu + N(0, 1) exp(log_std)
  mlp_actor_critic      :98-107  'pi' and 'v' scopes, v = squeeze(mlp(x, hidden + [1]))
Parameters are taken in the reference's variable naming (pi/dense{,_1,..}/{kernel,bias}, pi/log_std, v/dense*), i.e. what
ml4ca_amd.policy.ActorCritic.state_dict() / tf_checkpoint.read_bundle() give, and evaluated in float64 with the fp32 VALUES of the
parameters - the exact real-number function the reference's fp32 TensorFlow graph approximates.

PINNED TO THE REFERENCE'S OWN GRAPH FILE (round 4): tests/golden/final_graph.json holds the forward part of
data/finalmodel/finconttothighbowder_s0/tf1_save/saved_model.pb (read without TensorFlow by ml4ca_amd/tf_graph.py,
tests/golden/gen_final_graph.py) - node list, op order MatMul -> BiasAdd -> Maximum(alpha x, x), alpha, the likelihood's constants, the
signature and the tensor test_policy.py:90 picks - and tests/test_policy_import_cpu.py executes THAT graph node by node on the
checkpoint's variables and requires these functions to agree with it to 1e-12.  The constants below are the float32 values the graph
holds.  NOT pinned: TensorFlow's own kernels (no TensorFlow in this image, so no vector it computed exists) - i.e. its fp32 summation
order, which is inside the 1e-5 the arithmetic claim is stated at.  Only tests/ import this.
"""
import numpy as np

LEAKY_SLOPE = 0.20000000298023224      # pi/dense*/LeakyRelu/alpha: float32(0.2), tf.nn.leaky_relu's default (train.py:24,31 'leaky')
LIKELIHOOD_EPS = 9.99999993922529e-09  # pi/add_1/y: float32(1e-8)  (core.py:44 EPS)
LOG_2PI = 1.8378770351409912           # pi/add_3/y: float32(log(2 pi))  (core.py:45)


def _layers(params, scope):
    Ws, bs, i = [], [], 0
    while True:
        name = '%s/dense%s' % (scope, '' if i == 0 else '_%d' % i)
        if name + '/kernel' not in params:
            break
        Ws.append(np.asarray(params[name + '/kernel'], np.float64))
        bs.append(np.asarray(params[name + '/bias'], np.float64))
        i += 1
    return Ws, bs


def mlp(x, Ws, bs, activation='leaky', leak=LEAKY_SLOPE):
    """core.py:29-33"""
    x = np.asarray(x, np.float64)
    for W, b in zip(Ws[:-1], bs[:-1]):
        z = x @ W + b
        x = np.tanh(z) if activation == 'tanh' else np.where(z > 0, z, (0.0 if activation == 'relu' else leak) * z)
    return x @ Ws[-1] + bs[-1]


def actor_critic(params, obs, activation='leaky', leak=LEAKY_SLOPE):
    """(mu [n, act_dim], v [n]) - core.py:80-107, deterministic part"""
    pW, pb = _layers(params, 'pi')
    vW, vb = _layers(params, 'v')
    return mlp(obs, pW, pb, activation, leak), mlp(obs, vW, vb, activation, leak)[:, 0]


def gaussian_likelihood(x, mu, log_std):
    """core.py:42-46"""
    x, mu, log_std = np.asarray(x, np.float64), np.asarray(mu, np.float64), np.asarray(log_std, np.float64)
    pre = -0.5 * (((x - mu) / (np.exp(log_std) + LIKELIHOOD_EPS)) ** 2 + 2 * log_std + LOG_2PI)
    return pre.sum(axis=1)
