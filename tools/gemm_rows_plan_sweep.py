#!/usr/bin/env python
"""The two large row-list products of a step (x . Wx of the second encoder: 'nn' 4480 x 2048 x 512; its input gradient
dz . Wx^T: 'nt' 4480 x 512 x 2048) under forced tile / split plans (d2p_gemm_force_plan).   python tools/gemm_rows_plan_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import call  # noqa: E402


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


build.build_library()
g = torch.Generator().manual_seed(1)
R, n = 6400, 4480
rows = torch.randperm(R, generator=g)[:n].sort().values.int().cuda()
for kind, N, Kd in (('nn', 2048, 512), ('nt', 512, 2048)):
    A = torch.randn(R, Kd, device='cuda')
    B = torch.randn(Kd, N, device='cuda') if kind == 'nn' else torch.randn(N, Kd, device='cuda')
    C = torch.zeros(R, N, device='cuda')
    call.d2p_gemm_force_plan(-1, 0)
    print('%s %d x %d x %d: automatic plan %.1f us' % (kind, n, N, Kd, timed(lambda: K.gemm_rows(kind, n, N, Kd, A, Kd, B, B.shape[1], C, N, rows))))
    for opt in (0, 1):
        call.d2p_gemm_set_option(opt)
        for tile, name in ((0, '64x64'), (4, '128x64'), (1, '128x128')):
            line = '   option %d tile %-8s' % (opt, name)
            for sp in (1, 2, 4):
                call.d2p_gemm_force_plan(tile, sp)
                t = timed(lambda: K.gemm_rows(kind, n, N, Kd, A, Kd, B, B.shape[1], C, N, rows))
                line += '  splits %d: %6.1f us' % (sp, t)
            print(line, flush=True)
    call.d2p_gemm_set_option(0)
    call.d2p_gemm_force_plan(-1, 0)
