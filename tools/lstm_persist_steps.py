#!/usr/bin/env python
"""Persistent LSTM launches at 5..80 time steps: per-time-step rate and fixed cost per call (forward and backward, 320 and 32 rows)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demo2program_amd import kernels as K
from lstm_persist_rows import mk, timed
for M in (320, 32):
    rows = []
    for T in (5, 10, 20, 40, 80):
        f, b = mk(M, T)
        tf = min(timed(lambda: K.lstm_seq_fwd_multi([f])) for _ in range(3))
        tb = min(timed(lambda: K.lstm_seq_bwd_multi([b])) for _ in range(3))
        rows.append((T, tf, tb))
    print('M=%d ' % M + '  '.join('T=%d fwd %.0f bwd %.0f' % r for r in rows))
    (t0, f0, b0), (t1, f1, b1) = rows[1], rows[-1]
    sf, sb = (f1 - f0) / (t1 - t0), (b1 - b0) / (t1 - t0)
    print('   fwd: %.2f us/step + %.0f us per call;  bwd: %.2f us/step + %.0f us per call' % (sf, f0 - sf * t0, sb, b0 - sb * t0))
