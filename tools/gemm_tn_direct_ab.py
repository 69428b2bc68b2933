#!/usr/bin/env python
"""The weight-gradient products over row lists (d2p_gemm_f32_tn_rows): the register-direct kernel (gemm_tn_direct_kernel)
against the staged kernel (d2p_gemm_set_option bit 6), with the error of both against an fp64 product.

  python tools/gemm_tn_direct_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    build.build_library()
    lib = load()
    g = torch.Generator().manual_seed(1)
    R = 6720
    for (M, N, Kn) in ((512, 2048, 4480), (512, 2048, 4448), (512, 2048, 4128), (512, 2048, 6400), (560, 2048, 4480), (48, 2048, 4480),
                       (512, 2048, 1600), (512, 512, 3200), (1024, 2048, 4480), (256, 2048, 4480), (500, 2048, 4480)):
        lda = 1024 if M <= 1024 else M
        A = (torch.rand(R, lda, generator=g) - 0.5).cuda()
        B = (torch.rand(R, N, generator=g) - 0.5).cuda()
        rows = torch.randperm(R, generator=g)[:Kn].sort().values.int().cuda()
        rows2 = torch.randperm(R, generator=g)[:Kn].sort().values.int().cuda()
        ref = (A[rows.long(), :M].double().t() @ B[rows2.long()].double())
        line = '%4d x %4d x %4d:' % (M, N, Kn)
        for opt, name in ((64, 'staged'), (256, 'direct 64x64'), (0, 'direct')):
            lib.d2p_gemm_set_option(opt)
            C = torch.full((M, N), float('nan'), device='cuda')
            K.gemm_tn_rows(M, N, Kn, A, lda, rows, B, N, rows2, C, N)
            err = (C.double() - ref).abs().max().item()
            C2 = torch.ones(M, N, device='cuda')
            K.gemm_tn_rows(M, N, Kn, A, lda, rows, B, N, rows2, C2, N, accumulate=True)
            err2 = (C2.double() - 1.0 - ref).abs().max().item()
            t = timed(lambda: K.gemm_tn_rows(M, N, Kn, A, lda, rows, B, N, rows2, C, N))
            line += '  %s %6.1f us %5.1f TF/s err %.1e / %.1e' % (name, t, 2.0 * M * N * Kn / t * 1e-6, err, err2)
        print(line, flush=True)
    lib.d2p_gemm_set_option(0)


if __name__ == '__main__':
    main()
