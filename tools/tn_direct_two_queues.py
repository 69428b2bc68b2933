#!/usr/bin/env python
"""The step's tail: N large A^T B weight-gradient products (512 x 2048 x 4448 over row lists) alone on the chip --
one after the other on one stream (what the side queue does behind the last recurrence) against dealt out over two
streams, for the library in D2P_LIB_PATH (the LDS request of gemm_tn_direct_kernel decides whether two of its
workgroups share a CU: 100 KB = never, by design beside the recurrences; 64 KB = two per CU -- built for the measurement with a -D
variant of gemm.hip's `lds_req`, `python demo2program_amd/build.py --source gemm.hip --variant ...`; result: DESIGN_APPENDIX A.6)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402


def main():
    build.build_library()
    g = torch.Generator().manual_seed(1)
    R, M, N, Kn, NP = 6720, 512, 2048, 4448, 4
    A = [(torch.rand(R, M, generator=g) - 0.5).cuda() for _ in range(NP)]
    B = [(torch.rand(R, N, generator=g) - 0.5).cuda() for _ in range(NP)]
    rows = torch.randperm(R, generator=g)[:Kn].sort().values.int().cuda()
    C = [torch.zeros(M, N, device='cuda') for _ in range(NP)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def one_stream():
        for i in range(NP):
            K.gemm_tn_rows(M, N, Kn, A[i], M, rows, B[i], N, rows, C[i], N)

    def two_streams():
        main = torch.cuda.current_stream()
        s1.wait_stream(main)
        s2.wait_stream(main)
        for i in range(NP):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                K.gemm_tn_rows(M, N, Kn, A[i], M, rows, B[i], N, rows, C[i], N)
        main.wait_stream(s1)
        main.wait_stream(s2)

    for name, fn in (('one stream', one_stream), ('two streams', two_streams)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / 30
        print('%s %-11s %d products: %7.1f us  (%.1f TF/s)' % (os.path.basename(os.environ.get('D2P_LIB_PATH', 'libd2p_hip.so')), name, NP, t,
                                                              NP * 2.0 * M * N * Kn / t * 1e-6))


if __name__ == '__main__':
    main()
