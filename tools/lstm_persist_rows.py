#!/usr/bin/env python
"""Persistent LSTM kernels, time per step against the number of rows (= phases per row domain): U = 512, 20 steps.
Forward: 64 column tiles x 4 row domains; backward: 32 column tiles x 8 row domains."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import kernels as K  # noqa: E402


def mk(M, T, U=512):
    g = torch.Generator().manual_seed(M)
    f = dict(M=M, U=U, n_steps=T, z=(torch.rand(T * M, 4 * U, generator=g) - 0.5).cuda(),
             Wh=((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.1).cuda(),
             h0=(torch.rand(M, U, generator=g) - 0.5).cuda(), c0=(torch.rand(M, U, generator=g) - 0.5).cuda(),
             hout=torch.zeros(T, M, U, device='cuda'), cs=torch.zeros(T, M, U, device='cuda'))
    b = dict(M=M, U=U, n_steps=T, z=f['z'], Wh=f['Wh'], c0=f['c0'], cs=f['cs'],
             dhout=(torch.rand(T, M, U, generator=g) - 0.5).cuda(), dz=torch.zeros(T * M, 4 * U, device='cuda'),
             dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'))
    return f, b


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


def phases(M, domains):
    rs = (M + 15) // 16
    rt = min(domains, rs)
    return '%d..%d' % (rs // rt, -(-rs // rt))


def main():
    T = 20
    for M in (16, 32, 64, 96, 128, 144, 160, 192, 256, 320, 384, 400, 512):
        f, b = mk(M, T)
        tf = min(timed(lambda: K.lstm_seq_fwd_multi([f])) for _ in range(3))
        tb = min(timed(lambda: K.lstm_seq_bwd_multi([b])) for _ in range(3))
        print('M=%3d  fwd %3.0f us = %5.2f us/step (%s phases)   bwd %3.0f us = %5.2f us/step (%s phases)   err %d' % (
            M, tf, tf / T, phases(M, 4), tb, tb / T, phases(M, 8), K.lstm_persist_error(True)))


if __name__ == '__main__':
    main()
