#!/usr/bin/env python
"""The decoders' small gradient products at config 2's shapes: the grouped launch (d2p_small_pair_products) against the
per-decoder GEMM launches it replaces.  us per call, one stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from bench_conv_wide import timed  # noqa: E402


def main():
    build.build_library()
    U = 512
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.rand(*s, generator=g) - 0.5).cuda()
    pp = []
    for R in (60, 7, 51):
        pp.append((R, U, r(R + 1, 4 * U), r(R, U), r(U, 4 * U), torch.empty(U, 4 * U, device='cuda'), torch.empty(R, U, device='cuda')))
    t_new = min(timed(lambda: K.small_pair_products(pp), 50) for _ in range(3))

    def old():
        for R, _, S, A, Wx, G1, G2 in pp:
            K.matmul_tn(A, S[:R], out=G1)
            K.matmul_nt(S[:R], Wx, out=G2)
    t_old = min(timed(old, 50) for _ in range(3))
    print('pair products (3 decoders): grouped %.1f us | six GEMM launches %.1f us' % (t_new * 1e6, t_old * 1e6))


if __name__ == '__main__':
    main()
