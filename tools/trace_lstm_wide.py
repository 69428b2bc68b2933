#!/usr/bin/env python
"""Timeline of one workgroup of the wide-tile persistent forward kernel (lstm_persist.hip): shader-clock stamps per
phase of MFMA wave 0 and of the publish wave, printed as intervals; and how many workgroups found their row domain on
one XCD (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import call, load  # noqa: E402
from check_lstm_wide import bench_lens, mk  # noqa: E402

MAXT, KK = 512, 8


def trace(name, seqs, block, sort=False):
    buf = torch.zeros(2 * MAXT * KK, dtype=torch.int64, device='cuda')
    K.set_lstm_persistent(True)
    for q in seqs:
        q.pop('row_order', None)
        if sort and q.get('lens') is not None:
            q['row_order'] = K.lstm_row_order(q['lens_host'])
    K.lstm_seq_fwd_multi(seqs)
    torch.cuda.synchronize()
    load().d2p_lstm_persist_wide_local_wgs(1)
    call.d2p_lstm_persist_set_trace(buf.data_ptr(), buf.numel() * 8, block)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.lstm_seq_fwd_multi(seqs)
    e1.record()
    torch.cuda.synchronize()
    call.d2p_lstm_persist_set_trace(None, 0, 0)
    nloc = load().d2p_lstm_persist_wide_local_wgs(1)
    ms = e0.elapsed_time(e1)
    tr = buf.cpu().reshape(2, MAXT, KK)
    c = tr[0]
    n = int((c[:, 0] != 0).sum().item())
    if n < 2:
        print('no trace recorded')
        return
    total = (c[n - 1, 4] - c[0, 0]).item()
    print('== %s block %d: %d ticks, %d clocks total (%.0f per tick), call %.1f us, %d workgroups on L2-local hand-offs'
          % (name, block, n, total, total / n, ms * 1e3, nloc))
    e = tr[1]
    print('   MFMA wave 0: tick | ->wait done | ->chain done | ->barrier A | ->barrier B || publish wave: barriers | store issue | drain')
    lo = max(0, n // 2 - 4)
    for i in range(lo, min(n, lo + 8)):
        print('   %4d | %6d | %6d | %6d | %6d || %6d | %6d | %6d'
              % (i, c[i, 1] - c[i, 0], c[i, 2] - c[i, 1], c[i, 3] - c[i, 2], c[i, 4] - c[i, 3],
                 e[i, 1] - e[i, 0], e[i, 2] - e[i, 1], e[i, 3] - e[i, 2]))

    def avg(a, b, t):
        return (t[4:n - 1, b] - t[4:n - 1, a]).float().mean().item()
    print('   avg: ->wait %.0f, ->chain %.0f, ->barrier A %.0f, ->barrier B %.0f || publish: barriers %.0f, issue %.0f, drain %.0f'
          % (avg(0, 1, c), avg(1, 2, c), avg(2, 3, c), avg(3, 4, c), avg(0, 1, e), avg(1, 2, e), avg(2, 3, e)))


if __name__ == '__main__':
    build.build_library()
    enc = mk(320, 20, masked=True, init=True, seed=12)
    lens = bench_lens(320, 20)
    enc['lens_host'], enc['lens'] = lens, torch.from_numpy(lens.astype(np.int32)).cuda()
    act, per, prog = mk(320, 20, seed=13), mk(320, 20, seed=14), mk(32, 50, seed=15)
    for xl in (1, 0):
        call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, xl)
        print('#### xcd_local = %d' % xl)
        trace('encoder 320 x 20 unsorted', [enc], 0)
        trace('encoder 320 x 20 unsorted', [enc], 3)
        trace('encoder 320 x 20 sorted', [enc], 0, sort=True)
        trace('encoder 320 x 20 sorted', [enc], 7, sort=True)
        trace('triple: act', [act, per, prog], 0)
        trace('triple: prog', [act, per, prog], 7)
        trace('prog alone', [prog], 0)
    call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, 1)
