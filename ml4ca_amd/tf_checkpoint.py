"""Read (and, for tests, write) TensorFlow "tensor bundle" checkpoints without TensorFlow.

The reference saves its trained actor-critic with tf.saved_model.simple_save
(src/rl/windows_workspace/spinup/utils/logx.py:161-228) which leaves
``tf1_save/variables/variables.index`` + ``variables.data-00000-of-00001``.  The index is a LevelDB-format table
(prefix-compressed key/value blocks, 48-byte footer with magic 0xdb4775248b80fb57) whose values are
BundleEntryProto messages {1: dtype, 2: shape{2: dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c};
the data file is the raw little-endian tensor bytes.  Only what that format needs is implemented: uncompressed
blocks, float32/int32/int64 tensors, a single shard.
"""
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block_entries(buf, off, size):
    blk = buf[off:off + size]
    if buf[off + size] != 0:
        raise ValueError('compressed table blocks are not supported')
    n_restarts = struct.unpack_from('<I', blk, len(blk) - 4)[0]
    end = len(blk) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(blk[pos:pos + vlen])
        pos += vlen


def _parse_proto(buf):
    """flat {field: [values]} for varint / length-delimited / fixed32 / fixed64 fields"""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.setdefault(field, []).append(v)
    return out


def read_bundle(prefix):
    """{variable name: ndarray} for the bundle ``prefix``.index / ``prefix``.data-00000-of-00001."""
    idx = open(prefix + '.index', 'rb').read()
    if len(idx) < 48 or struct.unpack_from('<Q', idx, len(idx) - 8)[0] != _MAGIC:
        raise ValueError('%s.index is not a tensor-bundle index (bad magic)' % prefix)
    footer = idx[-48:]
    _, p = _varint(footer, 0)          # metaindex handle offset
    _, p = _varint(footer, p)          # metaindex handle size
    ioff, p = _varint(footer, p)
    isize, p = _varint(footer, p)
    data = np.memmap(prefix + '.data-00000-of-00001', dtype=np.uint8, mode='r')
    out = {}
    for _, handle in _block_entries(idx, ioff, isize):
        boff, q = _varint(handle, 0)
        bsize, q = _varint(handle, q)
        for key, val in _block_entries(idx, boff, bsize):
            if key == b'':
                continue                                   # BundleHeaderProto
            e = _parse_proto(val)
            dtype = _DTYPES.get(e.get(1, [0])[0])
            if dtype is None:
                continue
            shape = []
            if 2 in e:
                for dim in _parse_proto(e[2][0]).get(2, []):
                    shape.append(_parse_proto(dim).get(1, [0])[0])
            if e.get(3, [0])[0] != 0:
                raise ValueError('multi-shard bundles are not supported')
            off, size = e.get(4, [0])[0], e.get(5, [0])[0]
            arr = np.frombuffer(bytes(data[off:off + size]), dtype=dtype).reshape(shape)
            out[key.decode()] = arr.copy()
    return out


# ---- minimal writer (tests only: round-trips the reader without needing the reference's files) -----------------
def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _field(num, wt, payload):
    return _enc_varint((num << 3) | wt) + payload


def _block(entries):
    body = bytearray()
    for k, v in entries:                                   # no prefix sharing: every entry is a restart point
        body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
    restarts, pos = [], 0
    for k, v in entries:
        restarts.append(pos)
        pos += len(_enc_varint(0)) + len(_enc_varint(len(k))) + len(_enc_varint(len(v))) + len(k) + len(v)
    for r in restarts:
        body += struct.pack('<I', r)
    body += struct.pack('<I', len(restarts))
    return bytes(body)


def write_bundle(prefix, tensors):
    inv = {np.dtype(v): k for k, v in _DTYPES.items()}
    data = bytearray()
    entries = [(b'', _field(1, 0, _enc_varint(1)))]        # header: num_shards = 1
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        a = a if a.flags.c_contiguous else a.copy()
        shape = b''.join(_field(2, 2, _enc_varint(len(d)) + d) for d in (_field(1, 0, _enc_varint(s)) for s in a.shape))
        val = _field(1, 0, _enc_varint(inv[a.dtype])) + _field(2, 2, _enc_varint(len(shape)) + shape) + \
            _field(4, 0, _enc_varint(len(data))) + _field(5, 0, _enc_varint(a.nbytes))
        entries.append((name.encode(), val))
        data += a.tobytes()
    blk = _block(entries)
    out = bytearray(blk) + b'\x00' + b'\x00\x00\x00\x00'                 # block + type + (unchecked) crc
    meta_off = len(out)
    meta = _block([])
    out += meta + b'\x00' + b'\x00\x00\x00\x00'
    idx_off = len(out)
    index = _block([(entries[-1][0] + b'\x00', _enc_varint(0) + _enc_varint(len(blk)))])
    out += index + b'\x00' + b'\x00\x00\x00\x00'
    footer = _enc_varint(meta_off) + _enc_varint(len(meta)) + _enc_varint(idx_off) + _enc_varint(len(index))
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    open(prefix + '.index', 'wb').write(bytes(out) + footer)
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
