#!/usr/bin/env python
"""Row-list GEMMs (d2p_gemm_f32_rows) against the dense launch over all rows, at the shapes of the first encoder's input
projection ('nn', 6400 x 2048 x 512) and input gradient ('nt', 6400 x 512 x 2048) with 70 % of the rows listed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demo2program_amd import build  # noqa: E402
build.build_library()
from demo2program_amd import kernels as K  # noqa: E402


def timed(fn, reps=50):
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
    R, frac = 6400, 0.7
    g = torch.Generator().manual_seed(1)
    keep = torch.rand(R, generator=g) < frac
    rows = torch.nonzero(keep).view(-1).int().cuda()
    n = rows.numel()
    for kind, N, Kd in (('nn', 2048, 512), ('nt', 512, 2048)):
        A = torch.randn(R, Kd, device='cuda')
        B = torch.randn(Kd, N, device='cuda') if kind == 'nn' else torch.randn(N, Kd, device='cuda')
        C = torch.zeros(R, N, device='cuda')
        bias = torch.randn(N, device='cuda') if kind == 'nn' else None
        dense = timed(lambda: K.gemm_raw(kind, R, N, Kd, A, Kd, B, B.shape[1], C, N, bias=bias))
        listed = timed(lambda: K.gemm_rows(kind, n, N, Kd, A, Kd, B, B.shape[1], C, N, rows, bias=bias))
        short = timed(lambda: K.gemm_raw(kind, n, N, Kd, A, Kd, B, B.shape[1], C, N, bias=bias))
        print('%s %dx%dx%d: dense all rows %.1f us | %d listed rows %.1f us | dense over %d contiguous rows %.1f us'
              % (kind, R, N, Kd, dense, n, listed, n, short))


def skinny():
    """The first encoder's input gradient at the Karel geometry: [rows, 48] = dz [rows, 2048] . Wx^T -- a tiny output
    with a long K; plans forced through d2p_gemm_force_plan(tile, splits)."""
    from demo2program_amd.lib import call
    R, N, Kd = 6400, 48, 2048
    g = torch.Generator().manual_seed(2)
    rows = torch.nonzero(torch.rand(R, generator=g) < 0.7).view(-1).int().cuda()
    n = rows.numel()
    A = torch.randn(R, Kd, device='cuda')
    B = torch.randn(N, Kd, device='cuda')
    C = torch.zeros(R, N, device='cuda')
    print('nt %d listed rows x %d x %d, default plan: %.1f us; dense all rows: %.1f us' % (
        n, N, Kd, timed(lambda: K.gemm_rows('nt', n, N, Kd, A, Kd, B, Kd, C, N, rows)),
        timed(lambda: K.gemm_raw('nt', R, N, Kd, A, Kd, B, Kd, C, N))))
    for tile, name in ((0, '64x64'), (4, '128x64'), (7, '32x32 KSR')):
        for splits in (1, 2, 4, 8):
            call.d2p_gemm_force_plan(tile, splits)
            t = timed(lambda: K.gemm_rows('nt', n, N, Kd, A, Kd, B, Kd, C, N, rows))
            print('  tile %-9s splits %d: %.1f us' % (name, splits, t))
    call.d2p_gemm_force_plan(-1, 0)


if __name__ == '__main__':
    main()
    skinny()
