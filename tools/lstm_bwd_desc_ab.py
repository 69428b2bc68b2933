#!/usr/bin/env python
"""Backward persistent kernel: stores / prefetches through buffer descriptors (d2p_lstm_persist_set_bwd_desc, round 4)
against 64-bit pointers (round 3): us per launch for one sorted 320-row sequence and the three decoders' launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402
from lstm_sorted_plan_sweep import seq, timed  # noqa: E402

if __name__ == '__main__':
    build.build_library()
    lib = load()
    g = torch.Generator().manual_seed(5)
    one = [seq(320, 20, 512, g, 'lens')]
    three = [seq(320, 20, 512, g, 'mask'), seq(320, 20, 512, g, 'mask'), seq(32, 50, 512, g, None)]
    outs = {}
    for rep in range(2):
        for on in (0, 1):
            lib.d2p_lstm_persist_set_bwd_desc(on)
            a, b = timed(lambda: K.lstm_seq_bwd_multi(one)), timed(lambda: K.lstm_seq_bwd_multi(three))
            outs[on] = [t.clone() for t in (one[0]['dz'], one[0]['dh0'], one[0]['dc0'], one[0]['db'], three[2]['dz'])]
            print('descriptors %d: one sorted sequence %.1f us, three decoders %.1f us' % (on, a, b), flush=True)
    lib.d2p_lstm_persist_set_bwd_desc(0)
    same = all(torch.equal(x, y) for x, y in zip(outs[0], outs[1]))
    print('results bit-identical: %s; error word 0x%x' % (same, K.lstm_persist_error(True)))
