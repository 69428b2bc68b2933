#!/usr/bin/env python
"""Per-launch time distribution of the persistent forward kernels over thousands of launches: an occasional stall of
tens of milliseconds (a hand-off that arrives late, yet inside the bounded wait) would not show up in an average."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import call  # noqa: E402
from check_lstm_wide import bench_lens, mk  # noqa: E402


def dist(name, fn, n=1500):
    fn(); fn()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    t = np.array([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)])
    slow = np.nonzero(t > 3 * np.median(t))[0]
    print('%-44s median %7.1f us  p99 %7.1f  max %9.1f  launches > 3x median: %d %s'
          % (name, np.median(t), np.percentile(t, 99), t.max(), slow.size, list(slow[:8])))
    return t


if __name__ == '__main__':
    build.build_library()
    K.set_lstm_persistent(True)
    enc = mk(320, 20, masked=True, init=False, seed=11)
    lens = bench_lens(320, 20)
    enc['lens_host'], enc['lens'] = lens, torch.from_numpy(lens.astype(np.int32)).cuda()
    act, per, prog = mk(320, 20, seed=13), mk(320, 20, seed=14), mk(32, 50, seed=15)
    order = K.lstm_row_order(lens)
    for wide, xl in ((0, 1), (1, 1), (1, 0)):
        call.d2p_lstm_persist_set_fwd_wide(wide, 0, 0, xl)
        tag = ('wide, L2-local' if xl else 'wide, write-through') if wide else 'narrow'
        enc.pop('row_order', None)
        dist(tag + ': encoder', lambda: K.lstm_seq_fwd_multi([enc]))
        dist(tag + ': act alone', lambda: K.lstm_seq_fwd_multi([act]))
        dist(tag + ': act + prog', lambda: K.lstm_seq_fwd_multi([act, prog]))
        if wide:
            enc['row_order'] = order
            dist(tag + ': encoder sorted', lambda: K.lstm_seq_fwd_multi([enc]))
            dist(tag + ': triple', lambda: K.lstm_seq_fwd_multi([act, per, prog]))
    call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, 1)
    print('error word: 0x%x' % K.lstm_persist_error())
