"""TensorFlow V2 checkpoint reader / writer (demo2program_amd/tf_checkpoint.py): format round trips,
hand-assembled byte strings of the table / protobuf layers, and the variable-name table.  No
TensorFlow-written file is available offline, so none of this is pinned to TensorFlow itself."""
import struct

import numpy as np
import pytest

from demo2program_amd import tf_checkpoint as T
from demo2program_amd.config import make_config
from demo2program_amd.params import param_shapes


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert T.crc32c(b'\x00' * 32) == 0x8a9136aa
    assert T.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert T.crc32c(bytes(range(32))) == 0x46dd794e
    assert T.crc32c(b'123456789') == 0xe3069283
    c = T.crc32c(b'123456789')
    assert T.masked_crc32c(b'123456789') == ((((c >> 15) | (c << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


def test_varints_and_entry_proto_bytes():
    for v in (0, 1, 127, 128, 300, 2 ** 32, 2 ** 63 - 1):
        enc = T._put_varint(v)
        assert T._get_varint(enc, 0) == (v, len(enc))
    assert T._put_varint(300) == b'\xac\x02'
    # dtype DT_FLOAT (1), shape [3, 5], offset 60, size 60, crc 0x01020304, field by field
    raw = b'\x08\x01' + b'\x12\x08' + b'\x12\x02\x08\x03' + b'\x12\x02\x08\x05' + b'\x20\x3c' + b'\x28\x3c' + \
        b'\x35' + struct.pack('<I', 0x01020304)
    assert T._entry_bytes(1, (3, 5), 60, 60, 0x01020304) == raw
    e = T._parse_entry(raw)
    assert (e['dtype'], e['shape'], e['offset'], e['size'], e['crc32c'], e['shard_id']) == (1, [3, 5], 60, 60, 0x01020304, 0)
    # an entry with shard_id 2 and a slice field is recognised
    e = T._parse_entry(b'\x08\x03\x18\x02\x3a\x00')
    assert e['dtype'] == 3 and e['shard_id'] == 2 and e['sliced']


def test_block_prefix_compression_round_trip():
    items = [(b'', b'h'), (b'a/b/weights', b'1'), (b'a/b/weights/Adam', b'22'), (b'a/c', b''), (b'b', b'x' * 300)]
    items += [(('k/%03d' % i).encode(), bytes([i])) for i in range(40)]      # crosses two restart points
    block = T._build_block(items)
    assert T._read_block(block + b'\x00' + b'\x00' * 4, 0, len(block)) == items
    # second entry shares 0 bytes (restart), third shares the whole previous key
    assert block[:3] == b'\x00\x00\x01' and b'\x0b\x05\x02/Adam22' in block
    with pytest.raises(ValueError):
        T._read_block(block + b'\x01' + b'\x00' * 4, 0, len(block))          # snappy-compressed block


def test_bundle_round_trip(tmp_path):
    rs = np.random.RandomState(0)
    tensors = {'Demo_Encoder/rnn/basic_lstm_cell/kernel': rs.randn(48, 64).astype(np.float32),
               'Demo_Encoder/rnn/basic_lstm_cell/bias': rs.randn(64).astype(np.float32),
               'global_step': np.asarray(1234, np.int64), 'lens': np.arange(7, dtype=np.int32),
               'flag': np.array([True, False]), 'wide': rs.randn(3, 3, 4, 16)}
    tensors.update({'many/%03d/weights' % i: rs.randn(5, 3).astype(np.float32) for i in range(120)})
    prefix = str(tmp_path / 'model-1234')
    T.write_bundle(prefix, tensors, block_bytes=512)                          # many data blocks
    assert T.is_tf_checkpoint(prefix) and not T.is_tf_checkpoint(prefix + '.npz')
    data = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', data[-8:])[0] == T.MAGIC and len(data) > 48
    header, entries = T.read_index(prefix + '.index')
    assert (1, 0, 1) in header                                                 # num_shards = 1
    assert len(entries) == len(tensors)
    back = T.read_bundle(prefix, verify_crc=True)
    assert set(back) == set(tensors)
    for n, a in tensors.items():
        assert back[n].dtype == a.dtype and back[n].shape == a.shape and np.array_equal(back[n], a), n
    # corruption is detected
    with open(prefix + '.data-00000-of-00001', 'r+b') as f:
        f.seek(10)
        f.write(b'\xff')
    with pytest.raises(ValueError, match='crc32c'):
        T.read_bundle(prefix, verify_crc=True)
    with pytest.raises(ValueError, match='magic'):
        bad = tmp_path / 'bad.index'
        bad.write_bytes(b'\x00' * 64)
        T.read_index(str(bad))


@pytest.mark.parametrize('preset', ['karel', 'vizdoom'])
def test_variable_name_table_covers_every_parameter(preset):
    cfg = make_config(preset)
    names = T.variable_names(cfg)
    shapes = param_shapes(cfg)
    assert set(shapes) <= set(names)
    theirs = [names[n] for n in shapes]
    assert len(set(theirs)) == len(theirs)
    assert names['conv1/W'] == 'Demo_Encoder/State_Encoder/conv1/Conv/weights'
    assert names['rn_c/fc2/gamma'] == 'demo_c_summary/rn_pool/fc2/bn_act/BatchNorm/gamma'
    assert names['prog/proj'] == 'Program_Decoder/dynamic_decoder/output_projection/kernel'
    assert names['moving_var/per/fc'] == 'Per_Decoder/fc2/bn_act/BatchNorm/moving_variance'
    assert ('conv5/W' in names) == (preset == 'vizdoom')


# Variables of the reference's Karel full model, derived by hand by walking the variable scopes of
# /root/reference/models/model_full.py (and models/ops.py for the layer helpers), NOT from the table under
# test: State_Encoder (:216-231) inside Demo_Encoder (:235-258); tf.nn.dynamic_rnn's default scope 'rnn' +
# BasicLSTMCell's 'basic_lstm_cell' (TF 1.3 names its variables kernel / bias); SecondPathEncoder called
# with scope='SecondPathEncoder' (:395); rn_pool (:333-349) inside SummarizeFeature scopes demo_h_summary /
# demo_c_summary (:399-404) -- the avgpool summaries (:380-385) own no variables; LSTM_Decoder (:444-471)
# under 'Program_Decoder' / 'Action_Decoder' / 'Per_Decoder' (:508-592) with Token_Embedding's
# embedding_map created INSIDE its scope (:283-296), the decoder cell and Dense(name='output_projection')
# inside dynamic_decode(scope='dynamic_decoder'), and Per_Encoder's fc2 created when its closure is
# called from get_DecoderHelper, after the 'Per_Encoder' scope has been left (:308-316,412-414).
# slim.conv2d -> 'Conv/{weights,biases}', slim.fully_connected -> 'fully_connected/{weights,biases}',
# contrib.layers.batch_norm under bn_act -> 'bn_act/BatchNorm/{beta,gamma,moving_mean,moving_variance}'.
REFERENCE_KAREL_VARIABLES = sorted(
    ['Demo_Encoder/State_Encoder/conv%d/Conv/%s' % (l, w) for l in (1, 2, 3) for w in ('weights', 'biases')] +
    ['Demo_Encoder/State_Encoder/conv%d/bn_act/BatchNorm/%s' % (l, w) for l in (1, 2, 3)
     for w in ('beta', 'gamma', 'moving_mean', 'moving_variance')] +
    ['%s/rnn/basic_lstm_cell/%s' % (s, w) for s in ('Demo_Encoder', 'SecondPathEncoder') for w in ('kernel', 'bias')] +
    ['%s/rn_pool/%s/fully_connected/%s' % (s, f, w) for s in ('demo_h_summary', 'demo_c_summary')
     for f in ('fc1', 'fc2') for w in ('weights', 'biases')] +
    ['%s/rn_pool/%s/bn_act/BatchNorm/%s' % (s, f, w) for s in ('demo_h_summary', 'demo_c_summary')
     for f in ('fc1', 'fc2') for w in ('beta', 'gamma', 'moving_mean', 'moving_variance')] +
    ['%s/Token_Embedding/embedding_map' % s for s in ('Program_Decoder', 'Action_Decoder')] +
    ['%s/dynamic_decoder/%s' % (s, w) for s in ('Program_Decoder', 'Action_Decoder', 'Per_Decoder')
     for w in ('basic_lstm_cell/kernel', 'basic_lstm_cell/bias', 'output_projection/kernel')] +
    ['Per_Decoder/fc2/fully_connected/%s' % w for w in ('weights', 'biases')] +
    ['Per_Decoder/fc2/bn_act/BatchNorm/%s' % w for w in ('beta', 'gamma', 'moving_mean', 'moving_variance')])


def test_variable_name_table_equals_the_scope_walk_of_the_reference():
    """an export/import round trip cannot catch a wrong name (both sides use the table)."""
    names = T.variable_names(make_config('karel'))
    assert sorted(names.values()) == REFERENCE_KAREL_VARIABLES
    assert len(REFERENCE_KAREL_VARIABLES) == 6 + 12 + 4 + 8 + 16 + 2 + 9 + 2 + 4
