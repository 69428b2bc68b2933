#!/usr/bin/env python
"""What a weight-gradient GEMM costs beside a persistent recurrence, by GEMM plan: the backward recurrence of a 320-row,
20-step, U = 512 sequence on one stream, two d2p_gemm_f32_tn_rows products (512 x 2048 over 4480 listed rows: the second
encoder's dWx and dWh) on another, started together -- the time until both streams are done against the two alone.
Plans are forced with d2p_gemm_force_plan (tile, splits): the register footprint of a tile decides how many of its
workgroups fit beside the recurrence's waves (256 of a SIMD's 512 registers each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402
from demo2program_amd.models.model_full import pick_concurrent_stream  # noqa: E402


def main():
    build.build_library()
    lib = load()
    g = torch.Generator().manual_seed(1)
    M, T, U = 320, 20, 512
    z = (torch.rand(T * M, 4 * U, generator=g) * 2 - 1).cuda()
    Wh = ((torch.rand(U, 4 * U, generator=g) * 2 - 1) * 0.05).cuda()
    c0 = torch.zeros(M, U, device='cuda')
    cs = torch.rand(T, M, U, generator=g).cuda()
    dhout = (torch.rand(T, M, U, generator=g) * 2 - 1).cuda()
    dz = torch.zeros(T * M, 4 * U, device='cuda')
    dh0, dc0, db = torch.zeros(M, U, device='cuda'), torch.zeros(M, U, device='cuda'), torch.zeros(4 * U, device='cuda')
    lens_h = torch.randint(8, T + 1, (M,), generator=g).int()
    lens = lens_h.cuda()
    order = K.lstm_row_order(lens_h.numpy())
    seq = [dict(M=M, U=U, n_steps=T, z=z, Wh=Wh, c0=c0, lens=lens, cs=cs, dhout=dhout, dz=dz, dh0=dh0, dc0=dc0, db=db,
                row_order=order)]
    R, Kn = T * M, 4480
    A = (torch.rand(R, U, generator=g) - 0.5).cuda()
    B = (torch.rand(R, 4 * U, generator=g) - 0.5).cuda()
    rows = torch.randperm(R, generator=g)[:Kn].sort().values.int().cuda()
    C1, C2 = torch.empty(U, 4 * U, device='cuda'), torch.empty(U, 4 * U, device='cuda')
    main_s = torch.cuda.current_stream()
    side = pick_concurrent_stream()

    def rec():
        K.lstm_seq_bwd_multi(seq)

    def gemms():
        K.gemm_tn_rows(U, 4 * U, Kn, A, U, rows, B, 4 * U, rows, C1, 4 * U)
        K.gemm_tn_rows(U, 4 * U, Kn, A, U, rows, B, 4 * U, rows, C2, 4 * U)

    def timed(do_rec, do_gemm, reps=20):
        tot = 0.0
        for i in range(reps + 3):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(main_s)
            side.wait_event(e0)
            if do_gemm:
                with torch.cuda.stream(side):
                    gemms()
            if do_rec:
                rec()
            e1.record(side)
            main_s.wait_event(e1)
            e2.record(main_s)
            torch.cuda.synchronize()
            if i >= 3:
                tot += e0.elapsed_time(e2)
        return tot / reps * 1e3

    print('recurrence alone: %.1f us' % timed(True, False))
    for tile, name, opts in ((-1, 'auto (64x64, 32-deep slabs)', 0), (-1, 'auto, 16-deep slabs', 16), (0, '64x64', 0),
                             (7, '32x32 wave-split', 0), (4, '128x64', 0), (1, '128x128', 0)):
        for sp in ((0,) if tile < 0 else ((1, 2, 4) if tile == 7 else (4, 8))):
            lib.d2p_gemm_force_plan(tile, sp)
            lib.d2p_gemm_set_option(opts)
            try:
                alone = timed(False, True)
                both = timed(True, True)
                print('%-30s splits %d: two GEMMs alone %.1f us, beside the recurrence %.1f us' % (name, sp, alone, both),
                      flush=True)
            except Exception as ex:
                print('%-30s splits %d: %s' % (name, sp, str(ex)[:80]))
    lib.d2p_gemm_force_plan(-1, 0)
    lib.d2p_gemm_set_option(0)
    assert K.lstm_persist_error(True) == 0


if __name__ == '__main__':
    main()
