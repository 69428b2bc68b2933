#!/usr/bin/env python
"""Row-list GEMMs of the second encoder (nn: x . Wx over the active rows, 2048 x 512; nt: dz . Wx^T, 512 x 2048) as a
function of the NUMBER of listed rows: is the launch's time a staircase in the workgroup count (256 CUs)?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demo2program_amd import build  # noqa: E402
build.build_library()
from demo2program_amd import kernels as K  # noqa: E402
from bench_gemm_rows import timed  # noqa: E402


def main():
    R = 6400
    g = torch.Generator().manual_seed(1)
    perm = torch.randperm(R, generator=g)
    for kind, N, Kd in (('nn', 2048, 512), ('nt', 512, 2048)):
        A = torch.randn(R, Kd, device='cuda')
        B = torch.randn(Kd, N, device='cuda') if kind == 'nn' else torch.randn(N, Kd, device='cuda')
        C = torch.zeros(R, N, device='cuda')
        bias = torch.randn(N, device='cuda') if kind == 'nn' else None
        for n in (3584, 3840, 4096, 4224, 4352, 4429, 4480, 4608, 4864, 5120):
            rows = perm[:n].sort().values.int().cuda()
            t = timed(lambda: K.gemm_rows(kind, n, N, Kd, A, Kd, B, B.shape[1], C, N, rows, bias=bias))
            print('%s %5d rows x %4d x %4d: %6.1f us  %6.1f TFLOP/s' % (kind, n, N, Kd, t, 2.0 * n * N * Kd / t / 1e6), flush=True)


if __name__ == '__main__':
    main()
