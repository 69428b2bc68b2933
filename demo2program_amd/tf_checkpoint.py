"""TensorFlow checkpoint (V2 "tensor bundle") reader / writer without TensorFlow, and the mapping
between the reference's variable names and this build's parameter names.

The reference saves with `tf.train.Saver(max_to_keep=100)` (trainer.py:114,180-184) and restores
trainable variables from `--checkpoint` (trainer.py:115,142-147; evaler.py:82-99).  TF 1.3 writes
the V2 format: `<prefix>.index` -- an SSTable (the LevelDB table format: prefix-compressed key /
value blocks, an index block, a 48-byte footer ending in the magic 0xdb4775248b80fb57) whose ""
key holds a BundleHeaderProto and whose other keys are variable names with BundleEntryProto
values (dtype, shape, shard, offset, size, crc32c) -- and `<prefix>.data-NNNNN-of-MMMMM` with
the raw little-endian tensor bytes.

UNVERIFIED against a checkpoint written by TensorFlow: none is available offline (the released
checkpoints are external links, README.md:186-210).  The format code is tested by round trip
and by hand-assembled byte strings; the variable-name table follows TF-1.3 naming conventions
(slim `Conv/weights`, `fully_connected/weights`, `BatchNorm/{beta,gamma,moving_mean,
moving_variance}`, `basic_lstm_cell/{kernel,bias}`) and can be overridden with a JSON map --
`import_checkpoint` reports exactly which names it could not match.
"""
import json
import os
import re
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------- varints
def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(value):
    out = bytearray()
    value &= (1 << 64) - 1
    while value >= 0x80:
        out.append((value & 0x7f) | 0x80)
        value >>= 7
    out.append(value)
    return bytes(out)


# ------------------------------------------------------------------------------------- crc32c
_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), bytewise table; used for the small index blocks, optional for tensors."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = crc ^ 0xffffffff
    tab = _CRC_TABLE
    for b in bytes(data):
        c = tab[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


# ------------------------------------------------------------------------------------- protobuf
def _parse_fields(buf):
    """-> list of (field number, wire type, value); value = int (varint / fixed) or bytes."""
    out, pos = [], 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((num, wt, v))
    return out


def _parse_entry(buf):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto): 1 dtype, 2 shape
    (TensorShapeProto: 2 = repeated Dim{1 size}), 3 shard_id, 4 offset, 5 size, 6 crc32c."""
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for num, _, v in _parse_fields(buf):
        if num == 1:
            e['dtype'] = v
        elif num == 2:
            for n2, _, dim in _parse_fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _, x in _parse_fields(dim):
                        if n3 == 1:
                            size = x - (1 << 64) if x >= 1 << 63 else x
                    e['shape'].append(size)
        elif num == 3:
            e['shard_id'] = v
        elif num == 4:
            e['offset'] = v
        elif num == 5:
            e['size'] = v
        elif num == 6:
            e['crc32c'] = v
        elif num == 7:
            e['sliced'] = True
    return e


def _entry_bytes(dtype_id, shape, offset, size, crc):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(int(s)) for s in shape))
    out = b'\x08' + _put_varint(dtype_id) + b'\x12' + _put_varint(len(dims)) + dims
    # shard_id 0 is the proto default and is omitted
    if offset:
        out += b'\x20' + _put_varint(offset)
    out += b'\x28' + _put_varint(size) + b'\x35' + struct.pack('<I', crc)
    return out


# ------------------------------------------------------------------------------------- SSTable
def _read_block(data, offset, size):
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError('compressed table block (type %d): not supported' % ctype)
    num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key, out = 0, b'', []
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path):
    """-> (header fields, {variable name: entry dict}) of a `<prefix>.index` file."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow V2 checkpoint index (bad table magic)' % path)
    footer = data[-48:]
    _, pos = _get_varint(footer, 0)          # metaindex handle (offset, size): unused
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries, header = {}, None
    for _, handle in _read_block(data, ioff, isize):
        boff, p2 = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p2)
        for key, value in _read_block(data, boff, bsize):
            if key == b'':
                header = _parse_fields(value)
            else:
                entries[key.decode()] = _parse_entry(value)
    return header, entries


def read_bundle(prefix, verify_crc=False):
    """All tensors of the checkpoint `<prefix>` -> {name: numpy array}."""
    header, entries = read_index(prefix + '.index')
    num_shards = next((v for num, _, v in (header or []) if num == 1), 1)     # BundleHeaderProto.num_shards
    shards, out = {}, {}
    for name, e in entries.items():
        if e['sliced']:
            raise ValueError('%s is a partitioned variable: not supported' % name)
        if e['dtype'] not in _DTYPES:
            raise ValueError('%s: unsupported dtype enum %d' % (name, e['dtype']))
        sid = e['shard_id']
        if sid not in shards:
            path = '%s.data-%05d-of-%05d' % (prefix, sid, num_shards)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode='r')
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if verify_crc and e['crc32c'] is not None and masked_crc32c(raw.tobytes()) != e['crc32c']:
            raise ValueError('%s: crc32c mismatch' % name)
        out[name] = np.frombuffer(raw.tobytes(), dtype=np.dtype(_DTYPES[e['dtype']]).newbyteorder('<')).reshape(
            e['shape']).copy()
    return out


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for i, (key, value) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        out += key[shared:] + value
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, block_bytes=4096, with_crc=True):
    """Writes `<prefix>.index` + `<prefix>.data-00000-of-00001` for {name: array} (one shard, keys
    sorted as the table requires)."""
    names = sorted(tensors)
    offset, items = 0, [(b'', b'\x08\x01\x1a\x02\x08\x01')]      # BundleHeaderProto: num_shards 1, version.producer 1
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in names:
            a = np.asarray(tensors[name])              # (ascontiguousarray would turn a scalar into [1])
            if not a.flags.c_contiguous:
                a = a.copy()
            if a.dtype not in _DTYPE_IDS:
                raise ValueError('%s: dtype %s not supported' % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
            f.write(raw)
            crc = masked_crc32c(raw) if with_crc else 0
            items.append((name.encode(), _entry_bytes(_DTYPE_IDS[a.dtype], a.shape, offset, len(raw), crc)))
            offset += len(raw)
    out, index_items, cur, cur_bytes = bytearray(), [], [], 0

    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        block = _build_block(cur)
        handle = _put_varint(len(out)) + _put_varint(len(block))
        trailer = b'\x00'
        out.extend(block + trailer + struct.pack('<I', masked_crc32c(block + trailer)))
        index_items.append((cur[-1][0], handle))     # any key >= the block's last key separates it
        cur, cur_bytes = [], 0

    for it in items:
        cur.append(it)
        cur_bytes += len(it[0]) + len(it[1])
        if cur_bytes >= block_bytes:
            flush()
    flush()
    meta = _build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out.extend(meta + b'\x00' + struct.pack('<I', masked_crc32c(meta + b'\x00')))
    index = _build_block(index_items, restart_interval=1)
    index_handle = _put_varint(len(out)) + _put_varint(len(index))
    out.extend(index + b'\x00' + struct.pack('<I', masked_crc32c(index + b'\x00')))
    footer = meta_handle + index_handle
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC))
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------------- names
def variable_names(config):
    """{this build's parameter / moving-statistic name: the reference graph's variable name} for
    the full model (scopes of models/model_full.py:216-599; leaf names by TF-1.3 convention)."""
    from .config import n_conv
    names = {}

    def bn(ours, scope):
        names[ours + '/beta'] = scope + '/bn_act/BatchNorm/beta'
        names[ours + '/gamma'] = scope + '/bn_act/BatchNorm/gamma'
        names['moving_mean/' + ours] = scope + '/bn_act/BatchNorm/moving_mean'
        names['moving_var/' + ours] = scope + '/bn_act/BatchNorm/moving_variance'

    for l in range(1, n_conv(config) + 1):
        scope = 'Demo_Encoder/State_Encoder/conv%d' % l
        names['conv%d/W' % l] = scope + '/Conv/weights'
        names['conv%d/b' % l] = scope + '/Conv/biases'
        bn('conv%d' % l, scope)
    for ours, scope in (('demo_lstm', 'Demo_Encoder/rnn/basic_lstm_cell'),
                        ('second_lstm', 'SecondPathEncoder/rnn/basic_lstm_cell')):
        names[ours + '/kernel'] = scope + '/kernel'
        names[ours + '/bias'] = scope + '/bias'
    for ours, scope in (('rn_h', 'demo_h_summary/rn_pool'), ('rn_c', 'demo_c_summary/rn_pool')):
        for fc in ('fc1', 'fc2'):
            names['%s/%s/W' % (ours, fc)] = '%s/%s/fully_connected/weights' % (scope, fc)
            names['%s/%s/b' % (ours, fc)] = '%s/%s/fully_connected/biases' % (scope, fc)
            bn('%s/%s' % (ours, fc), '%s/%s' % (scope, fc))
    for ours, scope in (('prog', 'Program_Decoder'), ('act', 'Action_Decoder'), ('per', 'Per_Decoder')):
        if ours != 'per':
            names[ours + '/embedding'] = scope + '/Token_Embedding/embedding_map'
        names[ours + '/lstm/kernel'] = scope + '/dynamic_decoder/basic_lstm_cell/kernel'
        names[ours + '/lstm/bias'] = scope + '/dynamic_decoder/basic_lstm_cell/bias'
        names[ours + '/proj'] = scope + '/dynamic_decoder/output_projection/kernel'
    # Per_Encoder returns a closure; its fc is only created when the closure is CALLED, and that happens
    # in get_DecoderHelper after the `with tf.variable_scope('Per_Encoder')` block has exited
    # (models/model_full.py:308-316,412-414,455-456), i.e. directly under LSTM_Decoder's 'Per_Decoder'
    # scope -- unlike Token_Embedding, whose embedding_map is created inside its own scope (:288-296)
    names['per/fc/W'] = 'Per_Decoder/fc2/fully_connected/weights'
    names['per/fc/b'] = 'Per_Decoder/fc2/fully_connected/biases'
    bn('per/fc', 'Per_Decoder/fc2')
    return names


def _moving_key(name):
    kind, scope = name.split('/', 1)
    return scope, 0 if kind == 'moving_mean' else 1


def export_checkpoint(model, prefix, global_step=0, name_map=None):
    """Writes the model's parameters and batch-norm moving statistics under the reference's names."""
    names = dict(variable_names(model.config))
    names.update(name_map or {})
    P = model.params.to_numpy('p')
    tensors = {'global_step': np.asarray(global_step, np.int64)}
    for ours, theirs in names.items():
        if ours.startswith('moving_'):
            scope, i = _moving_key(ours)
            if scope in model.moving:
                tensors[theirs] = model.moving[scope][i].cpu().numpy()
        elif ours in P:
            tensors[theirs] = P[ours]
    write_bundle(prefix, tensors)
    return sorted(tensors)


def import_checkpoint(prefix, model, name_map=None, strict=True):
    """Loads a TF V2 checkpoint into `model` (parameters + moving statistics).  Optimizer slot
    variables (`.../Adam`, `.../Adam_1`, beta power accumulators) in the file are ignored, like the
    reference's restore of trainable variables.  -> global step (0 if the file has none)."""
    import torch
    names = dict(variable_names(model.config))
    if isinstance(name_map, str):
        with open(name_map) as f:
            name_map = json.load(f)
    names.update(name_map or {})
    tensors = read_bundle(prefix)
    params, missing, wrong = {}, [], []
    for ours in model.params.shapes:
        theirs = names.get(ours)
        if theirs not in tensors:
            missing.append('%s <- %s' % (ours, theirs))
            continue
        t = tensors[theirs]
        if tuple(t.shape) != tuple(model.params.shapes[ours]):
            wrong.append('%s: checkpoint %s, model %s' % (theirs, t.shape, model.params.shapes[ours]))
            continue
        params[ours] = t.astype(np.float32)
    if (missing or wrong) and strict:
        slot = re.compile(r'/(Adam(_1)?|ExponentialMovingAverage)$|^(beta[12]_power|global_step|OptimizeLoss)')
        unused = sorted(n for n in tensors if n not in names.values() and not slot.search(n))
        raise KeyError('TF checkpoint %s does not match the model.\n  not found: %s\n  wrong shape: %s\n'
                       '  unmatched variables in the file: %s\n(pass name_map={our name: their name} to override)'
                       % (prefix, missing[:8], wrong[:8], unused[:12]))
    full = model.params.to_numpy('p')
    full.update(params)
    model.params.load(full)
    for scope, (mm, mv) in model.moving.items():
        for i, (kind, dst) in enumerate((('moving_mean/', mm), ('moving_var/', mv))):
            theirs = names.get(kind + scope)
            if theirs in tensors:
                dst.copy_(torch.from_numpy(np.array(tensors[theirs], dtype=np.float32)))
    step = tensors.get('global_step')
    return int(step) if step is not None else 0


def is_tf_checkpoint(path):
    return isinstance(path, str) and os.path.exists(path + '.index')
