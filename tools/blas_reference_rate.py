#!/usr/bin/env python
"""What the vendor library reaches on the step's large fp32 products (calibration only: the product path never calls it).
torch.mm in fp32 = rocBLAS / hipBLASLt sgemm; run under rocprofv3 --kernel-trace --stats to see the kernels it picks.

  python tools/blas_reference_rate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    from demo2program_amd import build, kernels as K
    build.build_library()
    shapes = [('tn', 512, 2048, 6400), ('tn', 512, 2048, 4480), ('nn', 6400, 2048, 512), ('nn', 4480, 2048, 512),
              ('nt', 6400, 512, 2048), ('tn', 512, 512, 3200), ('nn', 3200, 512, 512), ('nn', 8192, 8192, 8192)]
    for kind, M, N, Kd in shapes:
        if kind == 'tn':
            a, b = torch.randn(Kd, M, device='cuda'), torch.randn(Kd, N, device='cuda')
            f = lambda: torch.mm(a.t(), b)          # noqa: E731
        elif kind == 'nn':
            a, b = torch.randn(M, Kd, device='cuda'), torch.randn(Kd, N, device='cuda')
            f = lambda: torch.mm(a, b)              # noqa: E731
        else:
            a, b = torch.randn(M, Kd, device='cuda'), torch.randn(N, Kd, device='cuda')
            f = lambda: torch.mm(a, b.t())          # noqa: E731
        c = torch.empty(M, N, device='cuda')
        if M * N * Kd < 2 ** 36:
            lda, ldb = a.shape[1], b.shape[1]
            g = lambda: K.gemm_raw(kind, M, N, Kd, a, lda, b, ldb, c, N)      # noqa: E731
        else:
            g = None
        t = timed(f)
        line = '%s %5d x %5d x %5d: library %7.1f us %6.1f TFLOP/s' % (kind, M, N, Kd, t * 1e6, 2.0 * M * N * Kd / t * 1e-12)
        if g is not None:
            try:
                t2 = timed(g)
                line += ' | this repo %7.1f us %6.1f TFLOP/s' % (t2 * 1e6, 2.0 * M * N * Kd / t2 * 1e-12)
            except Exception as e:      # noqa: BLE001
                line += ' | this repo: %s' % e
        print(line, flush=True)


if __name__ == '__main__':
    main()
