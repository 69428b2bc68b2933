#!/usr/bin/env python
"""Timeline of one workgroup of the persistent LSTM kernels (lstm_persist.hip): shader-clock stamps
per phase of MFMA wave 0 and of the epilogue wave, printed as intervals (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import call  # noqa: E402
from check_lstm_persist import Seq  # noqa: E402

MAXT, KK = 512, 8


def trace(which, M, U, T, masked, block):
    s = Seq(M, U, T, masked, True, seed=3)
    buf = torch.zeros(2 * MAXT * KK, dtype=torch.int64, device='cuda')
    K.set_lstm_persistent(True)
    s.fwd()
    if which == 'bwd':
        s.bwd()
    torch.cuda.synchronize()
    call.d2p_lstm_persist_set_trace(buf.data_ptr(), buf.numel() * 8, block)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    (s.fwd if which == 'fwd' else s.bwd)()
    e1.record()
    torch.cuda.synchronize()
    call.d2p_lstm_persist_set_trace(None, 0, 0)
    ms = e0.elapsed_time(e1)
    tr = buf.cpu().reshape(2, MAXT, KK)
    c = tr[0]
    n = int((c[:, 0] != 0).sum().item())
    if n < 2:
        print('no trace recorded')
        return
    total = (c[n - 1, 4] - c[0, 0]).item()
    print('== %s M=%d U=%d T=%d masked=%d block %d: %d ticks, %d clocks total, call %.1f us (incl. pack/memset)'
          % (which, M, U, T, masked, block, n, total, ms * 1e3))
    print('   clocks per tick %.0f; per step %.0f' % (total / n, total / T))
    e = tr[1]
    print('   MFMA wave 0: tick | flag wait | loads+chain | P+barrier A | epilogue+barrier B || publish wave: both barriers | store+DMA issue | drain')
    lo = max(0, n // 2 - 6)
    for i in range(lo, min(n, lo + 14)):
        print('   %4d | %6d | %6d | %6d | %6d || %6d | %6d | %6d'
              % (i, c[i, 1] - c[i, 0], c[i, 2] - c[i, 1], c[i, 3] - c[i, 2], c[i, 4] - c[i, 3],
                 e[i, 1] - e[i, 0], e[i, 2] - e[i, 1], e[i, 3] - e[i, 2]))

    def avg(a, b, t):
        return (t[4:n - 1, b] - t[4:n - 1, a]).float().mean().item()
    print('   avg: flag wait %.0f, loads+chain %.0f, P+barrier A %.0f, epilogue+barrier B %.0f || publish: barriers %.0f, issue %.0f, drain %.0f'
          % (avg(0, 1, c), avg(1, 2, c), avg(2, 3, c), avg(3, 4, c), avg(0, 1, e), avg(1, 2, e), avg(2, 3, e)))


if __name__ == '__main__':
    build.build_library()
    if len(sys.argv) > 1:
        # which M T masked block[,block...]   e.g.  bwd 320 20 0 0,4   (block b of a 256-workgroup launch is in row domain b % 8)
        which, M, T, masked = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) != 0
        for b in sys.argv[5].split(','):
            trace(which, M, 512, T, masked, int(b))
        sys.exit(0)
    for which in ('fwd', 'bwd'):
        trace(which, 320, 512, 20, False, 0)
        trace(which, 320, 512, 20, False, 77)
    trace('fwd', 32, 512, 40, False, 0)
