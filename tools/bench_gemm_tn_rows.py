#!/usr/bin/env python
"""Micro-benchmark of the weight-gradient GEMMs over row lists (d2p_gemm_f32_tn_rows): C[M, N] = A[rows]^T B[rows]
at the shapes of a training step, for forced tile / split plans (d2p_gemm_force_plan).
usage: tools/bench_gemm_tn_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402


def main():
    build.build_library()
    lib = load()
    g = torch.Generator().manual_seed(1)
    R = 6720
    for (M, N, Kn) in ((512, 2048, 4512), (48, 2048, 4512), (512, 2048, 896)):
        A = (torch.rand(R, M, generator=g) - 0.5).cuda()
        B = (torch.rand(R, N, generator=g) - 0.5).cuda()
        rows = torch.randperm(R, generator=g)[:Kn].sort().values.int().cuda()
        C = torch.empty(M, N, device='cuda')
        for tile, name in ((-1, 'auto'), (0, '64x64'), (4, '128x64'), (1, '128x128')):
            for sp in ((0,) if tile < 0 else (1, 2, 4, 8, 16)):
                lib.d2p_gemm_force_plan(tile, sp)
                try:
                    for _ in range(3):
                        K.gemm_tn_rows(M, N, Kn, A, M, rows, B, N, rows, C, N)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        K.gemm_tn_rows(M, N, Kn, A, M, rows, B, N, rows, C, N)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / 20
                    print('%4dx%4dx%4d %8s splits %2d: %7.1f us  %6.1f TFLOP/s' %
                          (M, N, Kn, name, sp, us, 2.0 * M * N * Kn / us / 1e6), flush=True)
                except Exception as ex:          # a plan the shape does not take
                    print('%4dx%4dx%4d %8s splits %2d: %s' % (M, N, Kn, name, sp, str(ex)[:60]))
        lib.d2p_gemm_force_plan(-1, 0)


if __name__ == '__main__':
    main()
