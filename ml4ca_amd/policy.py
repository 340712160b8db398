"""Actor-critic of the reference PPO (spinup/algos/tf1/ppo/core.py:29-33,80-107) for the in-kernel rollout.

``ActorCritic`` holds the fp32 parameters in the reference's layout (dense kernels W[in][out], biases, log_std;
variable names pi/dense{,_1,_2,_3}/{kernel,bias}, pi/log_std, v/dense*) as torch tensors, offers an fp32 torch
forward pass (the numerics reference for tests and for the PPO update, which stays host-side glue), and uploads
itself to a ``BatchedRevoltEnv`` where libdpenv evaluates it on the matrix cores inside the rollout launch
(dpenv_policy.hip).
"""
import ctypes as C
import math

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


class ActorCritic(object):
    def __init__(self, obs_dim=9, act_dim=7, hidden_sizes=(80, 80, 80), leak=0.2, log_std_init=-0.5, seed=0, device='cpu',
                 activation='leaky'):
        """activation: the reference's --activation choices (train.py:24,31): 'leaky' (slope `leak`, default 0.2 as
        tf.nn.leaky_relu), 'relu' (= leaky with slope 0) or 'tanh'."""
        torch = _torch()
        if not (len(set(hidden_sizes)) == 1 and 1 <= len(hidden_sizes) <= 4):
            raise ValueError('equal hidden widths, 1..4 hidden layers (got %r)' % (tuple(hidden_sizes),))
        if activation not in ('leaky', 'relu', 'tanh'):
            raise ValueError("activation must be 'leaky', 'relu' or 'tanh' (got %r)" % (activation,))
        if activation == 'relu':
            leak = 0.0
        self.activation = activation
        self.obs_dim, self.act_dim, self.hidden_sizes, self.leak = obs_dim, act_dim, tuple(hidden_sizes), float(leak)
        g = torch.Generator().manual_seed(seed)
        self.device = torch.device(device)

        def net(out_dim):
            sizes = [obs_dim] + list(hidden_sizes) + [out_dim]
            Ws, bs = [], []
            for i in range(len(sizes) - 1):
                # tf.layers.dense default kernel initializer: glorot uniform, zero bias
                lim = math.sqrt(6.0 / (sizes[i] + sizes[i + 1]))
                Ws.append(((torch.rand((sizes[i], sizes[i + 1]), generator=g) * 2 - 1) * lim).to(self.device))
                bs.append(torch.zeros(sizes[i + 1], device=self.device))
            return Ws, bs

        self.pi_W, self.pi_b = net(act_dim)
        self.v_W, self.v_b = net(1)
        self.log_std = torch.full((act_dim,), float(log_std_init), device=self.device)      # core.py:83

    @classmethod
    def from_tensors(cls, tensors, leak=0.2, device='cpu', activation='leaky'):
        """Build from a {name: array} dict with the reference's variable names (core.py:103-106 scopes 'pi' and 'v',
        tf.layers.dense naming dense, dense_1, ...; pi/log_std) - e.g. the output of tf_checkpoint.read_bundle."""
        torch = _torch()

        def layers(scope):
            Ws, bs, i = [], [], 0
            while True:
                name = '%s/dense%s' % (scope, '' if i == 0 else '_%d' % i)
                if name + '/kernel' not in tensors:
                    break
                Ws.append(torch.tensor(np.asarray(tensors[name + '/kernel'], np.float32), device=device))
                bs.append(torch.tensor(np.asarray(tensors[name + '/bias'], np.float32), device=device))
                i += 1
            return Ws, bs

        pW, pb = layers('pi')
        vW, vb = layers('v')
        if not pW or not vW:
            raise ValueError("no 'pi/dense*/kernel' / 'v/dense*/kernel' variables found")
        hidden = tuple(int(w.shape[1]) for w in pW[:-1])
        ac = cls(int(pW[0].shape[0]), int(pW[-1].shape[1]), hidden, leak=leak, device=device, activation=activation)
        ac.pi_W, ac.pi_b, ac.v_W, ac.v_b = pW, pb, vW, vb
        ac.log_std = torch.tensor(np.asarray(tensors['pi/log_std'], np.float32), device=device)
        return ac

    @classmethod
    def from_tf_checkpoint(cls, prefix, leak=0.2, device='cpu', activation='leaky'):
        """Load a reference-trained model, e.g. '<model dir>/tf1_save/variables/variables' (logx.py:161-228)."""
        from .tf_checkpoint import read_bundle
        return cls.from_tensors(read_bundle(prefix), leak=leak, device=device, activation=activation)

    @classmethod
    def from_saved_model(cls, model_dir, device='cpu'):
        """Load a reference-trained model from its `tf1_save` directory (logx.py:161-228: `saved_model.pb` + `variables/`), taking the
        hidden activation and its slope from the saved GRAPH (tf_graph.py) instead of from the caller: the thesis' models are
        Maximum(0.2 x, x) (tf.nn.leaky_relu of TensorFlow 1.12), which is also how the kernels evaluate it."""
        import os
        from .tf_checkpoint import read_bundle
        from . import tf_graph
        g = tf_graph.read_saved_model(os.path.join(model_dir, 'saved_model.pb'))
        desc = tf_graph.describe_actor_critic(g)
        act, alpha = tf_graph.hidden_activation(desc)
        for net in ('pi', 'v'):                            # the shape this library evaluates: dense stacks, no activation behind the last layer
            for l in desc[net][:-1]:
                if len(l['ops']) != 3:
                    raise ValueError('unsupported saved graph: hidden layer %r is %r, expected MatMul -> BiasAdd -> activation' % (l['layer'], l['ops']))
            if desc[net][-1]['ops'] != ['MatMul', 'BiasAdd']:
                raise ValueError('unsupported saved graph: an activation (%r) follows the last layer %r' % (desc[net][-1]['ops'], desc[net][-1]['layer']))
        ac = cls.from_tensors(read_bundle(os.path.join(model_dir, 'variables', 'variables')), leak=float(np.float32(alpha)), device=device,
                              activation=act)
        ac.graph = desc
        return ac

    def state_dict(self):
        """{reference variable name: numpy array} (inverse of from_tensors)."""
        out = {}
        for scope, Ws, bs in (('pi', self.pi_W, self.pi_b), ('v', self.v_W, self.v_b)):
            for i, (W, b) in enumerate(zip(Ws, bs)):
                name = '%s/dense%s' % (scope, '' if i == 0 else '_%d' % i)
                out[name + '/kernel'] = W.detach().cpu().numpy()
                out[name + '/bias'] = b.detach().cpu().numpy()
        out['pi/log_std'] = self.log_std.detach().cpu().numpy()
        return out

    def parameters(self):
        return self.pi_W + self.pi_b + self.v_W + self.v_b + [self.log_std]

    def _mlp(self, x, Ws, bs):
        torch = _torch()
        for W, b in zip(Ws[:-1], bs[:-1]):
            x = torch.tanh(x @ W + b) if self.activation == 'tanh' else torch.nn.functional.leaky_relu(x @ W + b, self.leak)
        return x @ Ws[-1] + bs[-1]

    def forward_ref(self, obs):
        """fp32 torch reference: (mu [n, act_dim], v [n])."""
        return self._mlp(obs, self.pi_W, self.pi_b), self._mlp(obs, self.v_W, self.v_b)[:, 0]

    def logp_ref(self, act, mu):
        """gaussian_likelihood, core.py:42-46."""
        torch = _torch()
        pre = -0.5 * (((act - mu) / (torch.exp(self.log_std) + 1e-8)) ** 2 + 2 * self.log_std + math.log(2 * math.pi))
        return pre.sum(dim=1)

    def upload(self, env, precision='f16', launch_form='auto'):
        """Pack into the env's library handle (dpenv_set_policy_desc); call again after each PPO update.

        precision: 'f16' (fast mode: f16 weights / activations, f32 accumulation), 'f32' (split-f16 arithmetic within 1e-5 of an
        fp32 evaluation - the reference's networks are fp32, core.py:29-33) or 'f32_actor' (the actor - mu, action, logp - as in
        'f32', the critic as in 'f16': the PPO ratio is exact, values carry the fast mode's ~5e-4; ~1.5 x the speed of 'f32').
        launch_form: 'auto' | 'one_wave' | 'two_wave'.
        Parameters that live on the env's device are handed over as DEVICE pointers: one packing kernel on the current stream,
        no host copy and no synchronisation; parameters elsewhere (CPU tensors) go through a host copy."""
        torch = _torch()
        lib = env.lib
        on_dev = all(p.is_cuda and p.device == env.device and p.dtype == torch.float32 and p.is_contiguous() for p in self.parameters())
        keep = []

        def mk(Ws, bs):
            m = _lib.Mlp()
            m.n_layers = len(Ws)
            sizes = [Ws[0].shape[0]] + [w.shape[1] for w in Ws]
            for i, sz in enumerate(sizes):
                m.sizes[i] = int(sz)
            for i, (W, b) in enumerate(zip(Ws, bs)):
                if on_dev:
                    m.W[i], m.b[i] = W.data_ptr(), b.data_ptr()
                else:
                    Wn = np.ascontiguousarray(W.detach().float().cpu().numpy())
                    bn = np.ascontiguousarray(b.detach().float().cpu().numpy())
                    keep.extend([Wn, bn])
                    m.W[i], m.b[i] = Wn.ctypes.data, bn.ctypes.data
            return m

        pi, v = mk(self.pi_W, self.pi_b), mk(self.v_W, self.v_b)
        d = _lib.PolicyDesc()
        d.struct_size = C.sizeof(_lib.PolicyDesc)
        d.pi, d.v = C.pointer(pi), C.pointer(v)
        if on_dev:
            d.log_std = self.log_std.data_ptr()
        else:
            ls = np.ascontiguousarray(self.log_std.detach().float().cpu().numpy())
            keep.append(ls)
            d.log_std = ls.ctypes.data
        d.activation = _lib.ACT_TANH if self.activation == 'tanh' else _lib.ACT_LEAKY_RELU
        d.leak = float(self.leak)
        d.precision = {'f16': _lib.POLICY_F16, 'f32': _lib.POLICY_F32, 'f32_actor': _lib.POLICY_F32_ACTOR}[precision]
        d.launch_form = {'auto': _lib.LAUNCH_AUTO, 'one_wave': _lib.LAUNCH_ONE_WAVE, 'two_wave': _lib.LAUNCH_TWO_WAVE}[launch_form]
        d.device_pointers = 1 if on_dev else 0
        with torch.cuda.device(env.device):
            _lib.check(lib.dpenv_set_policy_desc(env._h, C.byref(d), env._stream()), env._h)
            if not on_dev:
                torch.cuda.current_stream(env.device).synchronize()      # the host arrays in `keep` must outlive the copies
        env._has_policy = True
        self.precision = precision
        return self


def policy_launch_form(env):
    """('two_wave' | 'one_wave', envs per workgroup) that the uploaded policy's rollouts run in (what 'auto' resolved to)."""
    tw, epw = C.c_int32(0), C.c_int32(0)
    _lib.check(env.lib.dpenv_get_policy_launch(env._h, C.byref(tw), C.byref(epw)), env._h)
    return ('two_wave' if tw.value else 'one_wave'), int(epw.value)


def policy_launch_info(env):
    """dict(two_wave, envs_per_workgroup, waves_per_64_envs, precision) of the uploaded policy as the LIBRARY resolved it
    (dpenv_get_policy_launch_ex): waves 1 = one-wave form, 2 = env + network wave, 3 = env + actor + critic wave."""
    out = (C.c_int32 * 4)()
    _lib.check(env.lib.dpenv_get_policy_launch_ex(env._h, out), env._h)
    return dict(two_wave=bool(out[0]), envs_per_workgroup=int(out[1]), waves_per_64_envs=int(out[2]),
                precision={_lib.POLICY_F16: 'f16', _lib.POLICY_F32: 'f32', _lib.POLICY_F32_ACTOR: 'f32_actor'}[int(out[3])])


def release_policy_graphs(env):
    """dpenv_release_policy_graphs: the HIP graphs that recorded closed-loop launches of this env are gone; uploads of any layout are accepted again."""
    _lib.check(env.lib.dpenv_release_policy_graphs(env._h), env._h)


def policy_forward(env, obs):
    """Deterministic actor mean and critic value for obs [n, obs_dim] on the env's device (dpenv_policy_forward)."""
    torch = _torch()
    n = obs.shape[0]
    env._chk(obs, (n, env.num_states), torch.float32, 'obs')
    mu = torch.empty((n, env.num_actions), dtype=torch.float32, device=env.device)
    v = torch.empty(n, dtype=torch.float32, device=env.device)
    _lib.check(env.lib.dpenv_policy_forward(env._h, env._ptr(obs), env._ptr(mu), env._ptr(v), n, env._stream()), env._h)
    return mu, v


def policy_rollout(env, T, noise=None, switch_steps=(), refs=None, out=None, sample=None, reset_at_end=False):
    """T steps of (actor -> sample -> env.step -> critic) in ONE launch: the rollout loop ppo.py:289-322 for every env.

    noise: float32 [T, n, act_dim] standard-normal draws (a = mu + exp(log_std) * noise, core.py:85), or None.  With noise None:
    sample=True draws the exploration noise inside the kernel (Philox keyed by the seed, the global env id and the number of
    actions the env has sampled so far - like tf.random_normal inside the reference's graph, core.py:85, but reproducible and
    independent of the rank count); sample=False / None is the deterministic policy a = mu (test_policy.py:90).  Returns a dict of blocks: obs [T,n,od] (policy inputs), act [T,n,ad],
    rew, val, logp, boot [T,n], done [T,n] uint8, last_obs [n,od], last_val [n].  GAE: rollout.gae(rew, val, end=done, boot=boot).
    reset_at_end: the reference's epoch boundary (ppo.py:305-322): every env is cut and re-drawn after step T-1 (dpenv.h)."""
    torch = _torch()
    n, od, ad = env.n_envs, env.num_states, env.num_actions
    dev = env.device
    if noise is not None:
        env._chk(noise, (T, n, ad), torch.float32, 'noise')
    k = len(switch_steps)
    if k:
        env._chk(refs, (k, 3, n), torch.float32, 'refs')
    f32 = torch.float32
    if out is None:
        out = dict(obs=torch.empty((T, n, od), dtype=env.obs_torch_dtype, device=dev), act=torch.empty((T, n, ad), dtype=f32, device=dev),
                   rew=torch.empty((T, n), dtype=f32, device=dev), val=torch.empty((T, n), dtype=f32, device=dev),
                   logp=torch.empty((T, n), dtype=f32, device=dev), boot=torch.empty((T, n), dtype=f32, device=dev),
                   done=torch.empty((T, n), dtype=torch.uint8, device=dev),
                   last_obs=torch.empty((n, od), dtype=env.obs_torch_dtype, device=dev), last_val=torch.empty(n, dtype=f32, device=dev))
    io = _lib.PolicyRolloutIO()
    io.struct_size = C.sizeof(_lib.PolicyRolloutIO)
    io.T = int(T)
    io.noise = noise.data_ptr() if noise is not None else None
    io.obs, io.act, io.reward = out['obs'].data_ptr(), out['act'].data_ptr(), out['rew'].data_ptr()
    io.value, io.logp, io.done = out['val'].data_ptr(), out['logp'].data_ptr(), out['done'].data_ptr()
    io.boot, io.last_obs, io.last_value = out['boot'].data_ptr(), out['last_obs'].data_ptr(), out['last_val'].data_ptr()
    io.n_switch = k
    for j, st in enumerate(switch_steps):
        io.switch_step[j] = int(st)
    io.refs = refs.data_ptr() if k else None
    io.sample = 1 if (sample and noise is None) else 0
    io.reset_at_end = 1 if reset_at_end else 0
    _lib.check(env.lib.dpenv_policy_rollout(env._h, C.byref(io), env._stream()), env._h)
    return out
