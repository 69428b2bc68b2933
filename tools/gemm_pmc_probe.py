#!/usr/bin/env python
"""Runs the three large GEMM shapes of a training step 10 times each (for rocprofv3 --pmc passes:
tools/profile_gemm_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402

build.build_library()
g = torch.Generator().manual_seed(0)
for kind, M, N, Kd in (('nn', 6400, 2048, 512), ('nt', 6400, 512, 2048), ('tn', 512, 2048, 6400)):
    if kind == 'nn':
        A, B = torch.rand(M, Kd, generator=g).cuda(), torch.rand(Kd, N, generator=g).cuda()
        fn = lambda: K.matmul_nn(A, B, out=C)
    elif kind == 'nt':
        A, B = torch.rand(M, Kd, generator=g).cuda(), torch.rand(N, Kd, generator=g).cuda()
        fn = lambda: K.matmul_nt(A, B, out=C)
    else:
        A, B = torch.rand(Kd, M, generator=g).cuda(), torch.rand(Kd, N, generator=g).cuda()
        fn = lambda: K.matmul_tn(A, B, out=C)
    C = torch.empty(M, N, device='cuda')
    K.SCRATCH.reserve(16 * M * N * 4)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
