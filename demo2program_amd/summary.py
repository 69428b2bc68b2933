"""Scalar summaries as TensorBoard event files, without TensorFlow.

The reference writes `tf.summary.scalar` values through `tf.summary.FileWriter(train_dir)` (trainer.py:111-116,
170-178; tags at models/model_full.py:1139-1176: "loss/<name>" in the 'train' collection, "test_loss/<name>" in the
'test' collection).  This writer produces the same file format -- TFRecord framing (length, masked CRC-32C of the
length, payload, masked CRC-32C of the payload) around serialised `Event` messages -- for scalar values only; the
reference's text / image / histogram summaries (model_full.py:1178-1297) are not emitted.

Event / Summary wire format (tensorflow/core/util/event.proto, framework/summary.proto):
  Event   { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
  Summary { repeated Value value = 1; }   Value { string tag = 1; float simple_value = 2; }
"""
import os
import socket
import struct
import time

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1        # CRC-32C (Castagnoli), reflected
            tab.append(c)
        _CRC_TABLE = tab
    return _CRC_TABLE


def crc32c(data):
    tab = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time, step=None, scalars=None, file_version=None):
    msg = _varint((1 << 3) | 1) + struct.pack('<d', float(wall_time))
    if step is not None:
        msg += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        msg += _field_bytes(3, file_version.encode())
    if scalars:
        summary = b''
        for tag, value in scalars:
            v = _field_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack('<f', float(value))
            summary += _field_bytes(1, v)
        msg += _field_bytes(5, summary)
    return msg


def frame_record(payload):
    head = struct.pack('<Q', len(payload))
    return head + struct.pack('<I', masked_crc32c(head)) + payload + struct.pack('<I', masked_crc32c(payload))


class SummaryWriter(object):
    """`add_scalars({'loss/loss': 1.2, ...}, step)`; one events.out.tfevents.* file per writer, as FileWriter."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'ab')
        self._f.write(frame_record(encode_event(time.time(), file_version='brain.Event:2')))
        self._f.flush()

    def add_scalars(self, scalars, step):
        items = [(t, v) for t, v in sorted(scalars.items()) if v is not None]
        if items:
            self._f.write(frame_record(encode_event(time.time(), step=step, scalars=items)))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    """Parses an event file written by this module (or TensorFlow) back into [(step, {tag: value})]; checks every
    CRC.  Used by the tests."""
    out = []
    with open(path, 'rb') as f:
        data = f.read()
    pos = 0

    def varint(buf, i):
        n = shift = 0
        while True:
            b = buf[i]
            i += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, i

    def fields(buf):
        i = 0
        while i < len(buf):
            key, i = varint(buf, i)
            num, wt = key >> 3, key & 7
            if wt == 0:
                val, i = varint(buf, i)
            elif wt == 1:
                val, i = buf[i:i + 8], i + 8
            elif wt == 5:
                val, i = buf[i:i + 4], i + 4
            else:
                n, i = varint(buf, i)
                val, i = buf[i:i + n], i + n
            yield num, wt, val

    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack('<Q', head)
        assert struct.unpack('<I', data[pos + 8:pos + 12])[0] == masked_crc32c(head), 'length CRC'
        payload = data[pos + 12:pos + 12 + n]
        assert struct.unpack('<I', data[pos + 12 + n:pos + 16 + n])[0] == masked_crc32c(payload), 'payload CRC'
        pos += 16 + n
        step, scalars = 0, {}
        for num, wt, val in fields(payload):
            if num == 2:
                step = val
            elif num == 5:
                for n2, _, v in fields(val):
                    if n2 == 1:
                        tag, value = None, None
                        for n3, w3, x in fields(v):
                            if n3 == 1:
                                tag = x.decode()
                            elif n3 == 2 and w3 == 5:
                                (value,) = struct.unpack('<f', x)
                        if tag is not None:
                            scalars[tag] = value
        if scalars:
            out.append((step, scalars))
    return out
