#!/usr/bin/env python
"""The length-sorted backward launch under different planner cost models (d2p_lstm_persist_set_plan_cost): the kernel
time of one 320-row, 20-step, U = 512 sequence with lengths uniform in [8, 20] (the bench's demonstrations), and of the
three decoders' launch (320 + 320 + 32 rows)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402


def seq(M, T, U, g, mode):
    lens_h = torch.randint(8, T + 1, (M,), generator=g).int() if mode else None
    d = dict(M=M, U=U, n_steps=T, z=(torch.rand(T * M, 4 * U, generator=g) * 2 - 1).cuda(),
             Wh=((torch.rand(U, 4 * U, generator=g) * 2 - 1) * 0.05).cuda(), c0=torch.zeros(M, U, device='cuda'),
             cs=torch.rand(T, M, U, generator=g).cuda(), dhout=(torch.rand(T, M, U, generator=g) * 2 - 1).cuda(),
             dz=torch.zeros(T * M, 4 * U, device='cuda'), dh0=torch.zeros(M, U, device='cuda'),
             dc0=torch.zeros(M, U, device='cuda'), db=torch.zeros(4 * U, device='cuda'))
    if mode == 'lens':
        d['lens'] = lens_h.cuda()
    if mode:
        d['row_order'] = K.lstm_row_order(lens_h.numpy())
    return d


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    build.build_library()
    lib = load()
    g = torch.Generator().manual_seed(5)
    one = [seq(320, 20, 512, g, 'lens')]
    three = [seq(320, 20, 512, g, 'mask'), seq(320, 20, 512, g, 'mask'), seq(32, 50, 512, g, None)]
    lib.d2p_lstm_persist_set_sorted(0)
    print('unsorted: one sequence %.1f us, three decoders %.1f us' % (timed(lambda: K.lstm_seq_bwd_multi(one)),
                                                                   timed(lambda: K.lstm_seq_bwd_multi(three))))
    lib.d2p_lstm_persist_set_sorted(1)
    for ph, fl in ((3.3, 6.3), (2.8, 6.3), (2.8, 7.0), (3.3, 7.5), (3.0, 5.5), (2.5, 6.3), (3.6, 6.3), (3.3, 9.0),
                   (3.3, 4.0)):
        lib.d2p_lstm_persist_set_plan_cost(ph, fl)
        print('phase %.1f us, floor %.1f us: one sequence %.1f us, three decoders %.1f us' % (
            ph, fl, timed(lambda: K.lstm_seq_bwd_multi(one)), timed(lambda: K.lstm_seq_bwd_multi(three))), flush=True)
    lib.d2p_lstm_persist_set_plan_cost(3.3, 6.3)
    assert K.lstm_persist_error(True) == 0


if __name__ == '__main__':
    main()
