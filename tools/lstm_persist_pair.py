#!/usr/bin/env python
"""Two sequences in one persistent launch (d2p_lstm_seq_*_multi, nseq = 2) against the two launches back to back:
forward and backward times for the action + program decoder shapes of the three configs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import kernels as K  # noqa: E402


def mk(M, T, U=512, seed=0):
    g = torch.Generator().manual_seed(seed + M)
    o = dict(M=M, U=U, n_steps=T,
             z=(torch.rand(T * M, 4 * U, generator=g) - 0.5).cuda(),
             Wh=((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.1).cuda(),
             h0=(torch.rand(M, U, generator=g) - 0.5).cuda(), c0=(torch.rand(M, U, generator=g) - 0.5).cuda(),
             hout=torch.zeros(T, M, U, device='cuda'), cs=torch.zeros(T, M, U, device='cuda'))
    b = dict(M=M, U=U, n_steps=T, z=o['z'], Wh=o['Wh'], c0=o['c0'], cs=o['cs'],
             dhout=(torch.rand(T, M, U, generator=g) - 0.5).cuda(), dz=torch.zeros(T * M, 4 * U, device='cuda'),
             dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'))
    return o, b


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    for (Ma, Ta, Mb, Tb) in [(320, 20, 32, 50), (320, 20, 32, 32), (400, 20, 16, 32), (320, 20, 320, 20)]:
        fa, ba = mk(Ma, Ta, seed=1)
        fb, bb = mk(Mb, Tb, seed=2)
        K.lstm_seq_fwd_multi([fa]); K.lstm_seq_fwd_multi([fb])      # cs for the backward runs
        res = {}
        res['fwd a'] = timed(lambda: K.lstm_seq_fwd_multi([fa]))
        res['fwd b'] = timed(lambda: K.lstm_seq_fwd_multi([fb]))
        res['fwd pair'] = timed(lambda: K.lstm_seq_fwd_multi([fa, fb]))
        res['bwd a'] = timed(lambda: K.lstm_seq_bwd_multi([ba]))
        res['bwd b'] = timed(lambda: K.lstm_seq_bwd_multi([bb]))
        res['bwd pair'] = timed(lambda: K.lstm_seq_bwd_multi([ba, bb]))
        print('a: %d rows x %d steps, b: %d rows x %d steps | ' % (Ma, Ta, Mb, Tb) +
              '  '.join('%s %.0f us' % kv for kv in res.items()) + ' | err %d' % K.lstm_persist_error(True))


def triple():
    """backward of the three decoders: perception + action (320 rows x 20 steps) + program (32 rows x 50 steps)"""
    qs = [mk(320, 20, seed=1), mk(320, 20, seed=3), mk(32, 50, seed=2)]
    for f, _ in qs:
        K.lstm_seq_fwd_multi([f])
    b = [q[1] for q in qs]
    t_single = [timed(lambda i=i: K.lstm_seq_bwd_multi([b[i]])) for i in range(3)]
    t_pair = timed(lambda: K.lstm_seq_bwd_multi([b[1], b[2]]))
    t_triple = timed(lambda: K.lstm_seq_bwd_multi(b))
    print('bwd three decoders: singles %.0f + %.0f + %.0f us; one + pair %.0f + %.0f = %.0f us; all three in one launch '
          '%.0f us | err %d' % (t_single[0], t_single[1], t_single[2], t_single[0], t_pair, t_single[0] + t_pair,
                               t_triple, K.lstm_persist_error(True)))


if __name__ == '__main__':
    main()
    triple()
