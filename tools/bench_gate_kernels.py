#!/usr/bin/env python
"""HBM-roofline measurement of the standalone LSTM gate kernels K3/K4 (d2p_lstm_gate_fwd/bwd).
Algorithmic bytes per row (U=512): fwd 14336 B (read 4U pre-activations + U c_prev,
write U c + U h), bwd 26624 B (read dh, dc, 4U z, c_prev, c; write 4U dz + U dc).
Run on the GPU box; prints GB/s and the fraction of 8.0 TB/s (spec) / 6.29 TB/s (measured copy)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    build.build_library()
    U = 512
    print('| rows M | fwd us | fwd GB/s | frac 8.0 TB/s | bwd us | bwd GB/s | frac 8.0 TB/s |')
    print('|---|---|---|---|---|---|---|')
    for M in (320, 3200, 32000, 128000):
        z = torch.randn(M, 4 * U, device='cuda')
        c_prev = torch.randn(M, U, device='cuda')
        c_out, h_out = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
        dh, dc = torch.randn(M, U, device='cuda'), torch.randn(M, U, device='cuda')
        dz = torch.empty(M, 4 * U, device='cuda')
        reps = 200 if M <= 3200 else 30
        # rotate over several buffers > 256 MiB Infinity Cache for the large sizes is implicit:
        # M = 128000 touches 1.8 GB (fwd) / 3.4 GB (bwd) per launch
        tf = timed(lambda: K.lstm_gate_fwd(z, c_prev, None, None, 0, c_out, None, h_out), reps)
        tb = timed(lambda: K.lstm_gate_bwd(z, c_prev, c_out, dh, None, None, 0, dc, dz, None), reps)
        bf, bb = M * 14336.0 / tf / 1e9, M * 26624.0 / tb / 1e9
        print('| %d | %.1f | %.0f | %.2f | %.1f | %.0f | %.2f |' % (M, tf * 1e6, bf, bf / 8000, tb * 1e6, bb, bb / 8000))


if __name__ == '__main__':
    main()
