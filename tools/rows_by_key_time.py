#!/usr/bin/env python
"""Microseconds of d2p_embedding_scatter_add_oob0 at the decoders' shapes: rows_by_key_kernel (+ combine) against the one-hot
GEMM (d2p_gemm_set_option bit 7).   python tools/rows_by_key_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402


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
    return e0.elapsed_time(e1) * 1e3 / reps


build.build_library()
from demo2program_amd.lib import call  # noqa: E402
OFF = len(sys.argv) > 1 and sys.argv[1] == 'onehot'
call.d2p_gemm_set_option(128 if OFF else 0)
print('rows_by_key_kernel %s' % ('off (one-hot GEMM)' if OFF else 'on'))
g = torch.Generator().manual_seed(1)
for n, rows, E in ((6400, 9, 2048), (1568, 53, 2048), (6400, 52, 512)):
    ids = torch.randint(0, rows, (n,), generator=g, dtype=torch.int32).cuda()
    x = torch.randn(n, E, generator=g).cuda()
    out = torch.empty(rows, E, device='cuda')
    print('  n=%d keys=%d E=%d: %.1f us (%.0f GB/s of rows)' % (n, rows, E, timed(lambda: K.embedding_scatter_add(ids, x, out)),
                                                            n * E * 4 / timed(lambda: K.embedding_scatter_add(ids, x, out)) * 1e-3))
