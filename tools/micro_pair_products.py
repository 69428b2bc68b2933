#!/usr/bin/env python
"""Device time of d2p_small_pair_products by problem set (HIP events around 200 back-to-back launches of prepared calls)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import PairProductsDesc, load  # noqa: E402


def main():
    build.build_library()
    lib = load()
    U = 512
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.rand(*s, generator=g) - 0.5).cuda()
    keep = []

    def prob(R):
        t = (r(R + 1, 4 * U), r(R, U), r(U, 4 * U), torch.empty(U, 4 * U, device='cuda'), torch.empty(R, U, device='cuda'))
        keep.append(t)
        return (R,) + t
    for name, Rs in (('R = 60, 7, 51', (60, 7, 51)), ('R = 60', (60,)), ('R = 7', (7,)), ('R = 51', (51,)), ('R = 176', (176,))):
        ps = [prob(R) for R in Rs]
        arr = (PairProductsDesc * len(ps))()
        for d, (R, S, A, Wx, G1, G2) in zip(arr, ps):
            d.R, d.U, d.N4 = R, U, 4 * U
            d.S, d.A, d.Wx, d.G1, d.G2 = S.data_ptr(), A.data_ptr(), Wx.data_ptr(), G1.data_ptr(), G2.data_ptr()
        pa = ctypes.cast(arr, ctypes.c_void_p)
        st = K.current_stream()
        for _ in range(5):
            lib.d2p_small_pair_products(len(ps), pa, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            lib.d2p_small_pair_products(len(ps), pa, st)
        e1.record()
        torch.cuda.synchronize()
        print('%-16s %.1f us per launch' % (name, e0.elapsed_time(e1) * 1e3 / 200), flush=True)


if __name__ == '__main__':
    main()
