"""Read the GRAPH of a TensorFlow-1 SavedModel (`saved_model.pb`) without TensorFlow.

The reference saves its trained actor-critic with `tf.saved_model.simple_save` (spinup/utils/logx.py:161-228): the weights go to the
tensor bundle `tf1_save/variables/` (read by `tf_checkpoint.read_bundle`), the graph - which ops, in which order, with which attributes
and constants - to `tf1_save/saved_model.pb`.  `config.json` only names the activation FUNCTION (`tf.nn.leaky_relu`); that its slope is
0.2, that the layers are MatMul -> BiasAdd -> LeakyRelu with none behind the last, which tensor `test_policy.py:90` feeds to the env
(`pi/dense_3/BiasAdd`, the mean, not the sample) and the constants of the likelihood (1e-8, log 2 pi; core.py:42-46) are facts of the
GraphDef.  This module extracts them so that `ActorCritic.from_saved_model` takes topology and slope from the reference's own file
and a test pins `oracle/policy_ref.py` to it.

Wire format (protobuf, hand-walked like tf_checkpoint._parse_proto):
  SavedModel   {1: schema_version, 2: repeated MetaGraphDef}
  MetaGraphDef {1: meta_info_def, 2: GraphDef, 3: saver_def, 4: collection_def, 5: map<string, SignatureDef>}
  GraphDef     {1: repeated NodeDef, 2: library, 4: versions}
  NodeDef      {1: name, 2: op, 3: repeated input, 4: device, 5: map<string, AttrValue>}
  AttrValue    {1: list, 2: s, 3: i, 4: f (fixed32), 5: b, 6: type, 7: shape, 8: TensorProto}
  TensorProto  {1: dtype, 2: shape{2: dim{1: size}}, 4: tensor_content, 5: float_val, 6: double_val, 7: int_val, 10: int64_val}
  SignatureDef {1: map inputs<string, TensorInfo>, 2: map outputs, 3: method_name};  TensorInfo {1: name, 2: dtype, 3: shape}
"""
import struct

import numpy as np

from .tf_checkpoint import _parse_proto, _varint

_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}


def _s(b):
    return bytes(b).decode('utf-8', 'replace')


def _map_entries(fields):
    """protobuf map<k, v> = repeated {1: key, 2: value}"""
    for raw in fields:
        e = _parse_proto(raw)
        yield _s(e[1][0]), e.get(2, [b''])[0]


def _packed_floats(vals, fmt, size):
    out = []
    for v in vals:
        if isinstance(v, (bytes, bytearray, memoryview)):          # packed
            out += [struct.unpack_from(fmt, v, k)[0] for k in range(0, len(v), size)]
        else:                                                       # one fixed32 / fixed64 field per value
            out.append(struct.unpack(fmt, struct.pack('<I' if size == 4 else '<Q', v))[0])
    return out


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v              # negative ints travel as 64-bit two's complement varints


def _packed_varints(vals):
    out = []
    for v in vals:
        if isinstance(v, (bytes, bytearray, memoryview)):
            pos = 0
            while pos < len(v):
                x, pos = _varint(v, pos)
                out.append(_signed(x))
        else:
            out.append(_signed(v))
    return out


def _tensor(raw):
    t = _parse_proto(raw)
    dtype = _DT.get(t.get(1, [0])[0])
    shape = []
    if 2 in t:
        for dim in _parse_proto(t[2][0]).get(2, []):
            shape.append(_parse_proto(dim).get(1, [0])[0])
    if dtype is None:
        return None
    n = int(np.prod(shape)) if shape else 1
    if 4 in t and len(t[4][0]):
        arr = np.frombuffer(bytes(t[4][0]), dtype=dtype)
    elif 5 in t:
        arr = np.array(_packed_floats(t[5], '<f', 4), dtype=dtype)
    elif 6 in t:
        arr = np.array(_packed_floats(t[6], '<d', 8), dtype=dtype)
    elif 7 in t:
        arr = np.array(_packed_varints(t[7]), dtype=np.int64).astype(dtype)
    elif 10 in t:
        arr = np.array(_packed_varints(t[10]), dtype=np.int64).astype(dtype)
    else:
        arr = np.zeros(n, dtype)
    if arr.size == 1 and n > 1:
        arr = np.repeat(arr, n)                                      # TensorProto splat form
    return arr.reshape(shape)


def _attr(raw):
    a = _parse_proto(raw)
    if 1 in a:                                   # AttrValue.ListValue {3: i (packed or repeated varints), 4: f, 2: s}
        lv = _parse_proto(a[1][0])
        if 3 in lv:
            return _packed_varints(lv[3])
        if 4 in lv:
            return _packed_floats(lv[4], '<f', 4)
        if 2 in lv:
            return [_s(x) for x in lv[2]]
        return []
    if 8 in a:
        return _tensor(a[8][0])
    if 4 in a:
        return struct.unpack('<f', struct.pack('<I', a[4][0]))[0]
    if 3 in a:
        return _signed(a[3][0])
    if 5 in a:
        return bool(a[5][0])
    if 2 in a:
        return _s(a[2][0])
    if 6 in a:
        return ('dtype', a[6][0])
    return None


class Graph:
    """nodes: {name: {'op', 'inputs' [names, control inputs dropped, ':k' output suffix stripped], 'attr' {name: value}}} in file
    order (`order`); signature: {'inputs': {key: tensor name}, 'outputs': {...}} of the serving signature simple_save wrote."""

    def __init__(self, nodes, order, signature):
        self.nodes, self.order, self.signature = nodes, order, signature

    def op(self, name):
        return self.nodes[name]['op']

    def producers(self, name):
        return self.nodes[name]['inputs']

    def const(self, name):
        """value of a Const node (following Identity / read ops)"""
        nd = self.nodes[name]
        while nd['op'] in ('Identity',):
            nd = self.nodes[nd['inputs'][0]]
        if nd['op'] != 'Const':
            raise KeyError('%s is not a constant (%s)' % (name, nd['op']))
        return nd['attr'].get('value')


def read_saved_model(path):
    """Graph of the first MetaGraphDef of `path` (a saved_model.pb)."""
    buf = memoryview(open(path, 'rb').read())
    sm = _parse_proto(buf)
    if 2 not in sm:
        raise ValueError('%s holds no MetaGraphDef: not a SavedModel' % path)
    mg = _parse_proto(sm[2][0])
    gd = _parse_proto(mg[2][0])
    nodes, order = {}, []
    for raw in gd.get(1, []):
        nd = _parse_proto(raw)
        name = _s(nd[1][0])
        ins = []
        for i in nd.get(3, []):
            i = _s(i)
            if i.startswith('^'):
                continue
            ins.append(i.split(':')[0])
        attr = {}
        for k, v in _map_entries(nd.get(5, [])):
            attr[k] = _attr(v)
        nodes[name] = {'op': _s(nd[2][0]), 'inputs': ins, 'attr': attr}
        order.append(name)
    signature = {'inputs': {}, 'outputs': {}}
    for _, sraw in _map_entries(mg.get(5, [])):
        sd = _parse_proto(sraw)
        for field, key in ((1, 'inputs'), (2, 'outputs')):
            for k, traw in _map_entries(sd.get(field, [])):
                signature[key][k] = _s(_parse_proto(traw)[1][0])
    return Graph(nodes, order, signature)


def mlp_chain(g, scope):
    """The dense stack of variable scope `scope` ('pi' / 'v') as the graph has it, first layer to last:
    [{'layer': 'pi/dense', 'ops': ['MatMul', 'BiasAdd', 'LeakyRelu'], 'alpha': 0.2, 'form': 'Maximum(Mul(alpha, x), x)', ...}, ...]
    (core.py:29-33: `tf.layers.dense(x, units=h, activation=activation)`).  TensorFlow 1.12's tf.nn.leaky_relu is not one op but
    `Maximum(alpha * x, x)` under a `LeakyRelu` name scope (alpha a Const) - later versions emit a `LeakyRelu` op with an `alpha`
    attribute; both are recognised, as are single-op Relu / Tanh."""
    layers = []
    i = 0
    while True:
        name = '%s/dense%s' % (scope, '' if i == 0 else '_%d' % i)
        if name + '/MatMul' not in g.nodes:
            break
        mm = g.nodes[name + '/MatMul']
        rec = {'layer': name, 'input': mm['inputs'][0], 'kernel': mm['inputs'][1],
               'transpose_a': bool(mm['attr'].get('transpose_a', False)), 'transpose_b': bool(mm['attr'].get('transpose_b', False))}
        ba = name + '/BiasAdd'
        if not (ba in g.nodes and g.nodes[ba]['op'] == 'BiasAdd' and g.nodes[ba]['inputs'][0] == name + '/MatMul'):
            raise ValueError('unsupported graph: layer %r has no BiasAdd on its MatMul (%r)' % (name, g.nodes.get(ba)))
        rec['bias'] = g.nodes[ba]['inputs'][1]
        ops, out = ['MatMul', 'BiasAdd'], ba
        users = [n for n in g.order if n.startswith(name + '/') and ba in g.nodes[n]['inputs']]
        if users:
            lk = name + '/LeakyRelu'
            if lk in g.nodes and g.nodes[lk]['op'] == 'Maximum':
                mul = g.nodes[lk]['inputs'][0]
                if not (g.nodes[lk]['inputs'][1] == ba and g.nodes[mul]['op'] == 'Mul' and g.nodes[mul]['inputs'][1] == ba):
                    raise ValueError('unsupported graph: %r is not Maximum(Mul(alpha, x), x) of its BiasAdd: %r' % (lk, g.nodes[lk]))
                if sorted(users) != sorted([mul, lk]):
                    raise ValueError('unsupported graph: BiasAdd of layer %r has other users than its leaky-relu: %r' % (name, users))
                rec['alpha'] = float(np.asarray(g.const(g.nodes[mul]['inputs'][0])).reshape(-1)[0])
                rec['form'] = 'Maximum(Mul(alpha, x), x)'
                ops.append('LeakyRelu')
                out = lk
            else:
                if len(users) != 1:
                    raise ValueError('unsupported graph: BiasAdd of layer %r feeds %d ops (%r), expected one activation' % (name, len(users), users))
                a = g.nodes[users[0]]
                ops.append(a['op'])
                if 'alpha' in a['attr']:
                    rec['alpha'] = float(a['attr']['alpha'])
                rec['form'] = a['op']
                out = users[0]
        rec['ops'], rec['output'] = ops, out
        layers.append(rec)
        i += 1
    for a, b in zip(layers[:-1], layers[1:]):              # layer k + 1 reads layer k's output
        if b['input'] != a['output']:
            raise ValueError('unsupported graph: layer %r does not read the output of layer %r (%r)' % (b['layer'], a['layer'], (a, b)))
    return layers


def ancestors(g, fetches):
    """names of every node the fetches depend on (data inputs only), in file order"""
    need, stack = set(), [f.split(':')[0] for f in fetches]
    while stack:
        n = stack.pop()
        if n in need:
            continue
        need.add(n)
        if g.nodes[n]['op'] in ('VariableV2', 'Placeholder'):
            continue                                        # a variable's value comes from the bundle, not from its initialiser
        stack += g.nodes[n]['inputs']
    return [n for n in g.order if n in need]


def evaluate(nodes, fetches, feeds, variables, dtype=np.float64, rng_normal=None):
    """Execute the reference's own forward graph in NumPy: `nodes` = {name: {'op', 'inputs', 'attr'}} (Graph.nodes or the JSON fixture
    made from them), `feeds` = {placeholder name: array}, `variables` = {variable name: array} (tf_checkpoint.read_bundle).  Only the ops
    the actor-critic's forward pass, sampling and likelihood use (core.py:29-107) exist here; anything else raises.  `dtype` float32
    follows the graph's own arithmetic type, float64 gives the real-number function it approximates."""
    memo = {}

    def val(name):
        name = name.split(':')[0]
        if name in memo:
            return memo[name]
        nd = nodes[name]
        op, ins = nd['op'], nd['inputs']
        if op == 'Placeholder':
            r = np.asarray(feeds[name], dtype)
        elif op == 'VariableV2':
            r = np.asarray(variables[name], dtype)
        elif op == 'Const':
            v = np.asarray(nd['attr']['value'])
            r = v.astype(dtype) if v.dtype.kind == 'f' else v
        elif op == 'Identity':
            r = val(ins[0])
        elif op == 'MatMul':
            a, b = val(ins[0]), val(ins[1])
            r = (a.T if nd['attr'].get('transpose_a') else a) @ (b.T if nd['attr'].get('transpose_b') else b)
        elif op in ('BiasAdd', 'Add'):
            r = val(ins[0]) + val(ins[1])
        elif op == 'Sub':
            r = val(ins[0]) - val(ins[1])
        elif op == 'Mul':
            r = val(ins[0]) * val(ins[1])
        elif op == 'RealDiv':
            r = val(ins[0]) / val(ins[1])
        elif op == 'Maximum':
            r = np.maximum(val(ins[0]), val(ins[1]))
        elif op == 'Pow':
            r = np.power(val(ins[0]), val(ins[1]))
        elif op == 'Exp':
            r = np.exp(val(ins[0]))
        elif op == 'Tanh':
            r = np.tanh(val(ins[0]))
        elif op == 'Relu':
            r = np.maximum(val(ins[0]), 0)
        elif op == 'LeakyRelu':
            x = val(ins[0])
            r = np.where(x > 0, x, dtype(nd['attr'].get('alpha', 0.2)) * x)
        elif op == 'Sum':
            r = val(ins[0]).sum(axis=tuple(np.atleast_1d(val(ins[1])).tolist()), keepdims=bool(nd['attr'].get('keep_dims', False)))
        elif op == 'Squeeze':
            dims = nd['attr'].get('squeeze_dims')                 # list(int); empty = every dimension of size one
            r = np.squeeze(val(ins[0]), axis=tuple(int(k) for k in dims) if dims else None)
        elif op == 'Shape':
            r = np.array(val(ins[0]).shape, np.int64)
        elif op == 'RandomStandardNormal':
            if rng_normal is None:
                raise ValueError('%s: the graph samples here; pass rng_normal(shape)' % name)
            r = np.asarray(rng_normal(tuple(int(k) for k in val(ins[0]))), dtype)
        else:
            raise NotImplementedError('op %s (%s) is not part of the actor-critic forward graph' % (op, name))
        memo[name] = r
        return r

    return [val(f) for f in fetches]


def describe_actor_critic(g):
    """What a TensorFlow-free consumer needs to know of the reference's saved graph (JSON-able)."""
    out = {'signature': g.signature, 'pi': mlp_chain(g, 'pi'), 'v': mlp_chain(g, 'v')}
    acts = {l['ops'][2] for net in ('pi', 'v') for l in out[net] if len(l['ops']) > 2}
    alphas = {l['alpha'] for net in ('pi', 'v') for l in out[net] if 'alpha' in l}
    out['hidden_activation_ops'] = sorted(acts)
    out['leaky_alpha'] = sorted(alphas)
    # float constants of the graph outside the optimiser / gradient / save scopes: the likelihood's 1e-8 and log(2 pi), the clip ratio ...
    consts = {}
    for n in g.order:
        nd = g.nodes[n]
        if nd['op'] != 'Const' or n.startswith(('gradients', 'save', 'Adam', 'beta')) or '/Adam' in n or 'Initializer' in n or 'gradients' in n:
            continue
        v = nd['attr'].get('value')
        if isinstance(v, np.ndarray) and v.dtype in (np.float32, np.float64) and v.size == 1:
            consts[n] = float(v.reshape(-1)[0])
    out['scalar_float_constants'] = consts
    return out


def hidden_activation(desc):
    """('leaky' | 'relu' | 'tanh', slope) from describe_actor_critic()'s record; raises on a graph this library has no arithmetic for"""
    ops, alphas = desc['hidden_activation_ops'], desc['leaky_alpha']
    if ops == ['LeakyRelu'] and len(alphas) == 1:
        return 'leaky', alphas[0]
    if ops == ['Relu']:
        return 'relu', 0.0
    if ops == ['Tanh']:
        return 'tanh', 0.0
    raise ValueError('hidden activations %s (alpha %s): not one of leaky-relu / relu / tanh' % (ops, alphas))
