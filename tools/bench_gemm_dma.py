#!/usr/bin/env python
"""A/B of the staged (gemm_mfma_kernel) and LDS-DMA (gemm_dma_kernel) forms of the dense fp32 GEMM on the large
shapes of one training step (run on the GPU box).  For each shape: us and TFLOP/s per tile variant (best of
several interleaved rounds) and whether the result is bit-identical to the automatic plan's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402

SHAPES = [  # (kind, M, N, K, label)
    ('nn', 6400, 2048, 512, 'z   = X Wx     (6400x2048, K=512)'),
    ('nt', 6400, 512, 2048, 'dX  = dZ Wx^T  (6400x512, K=2048)'),
    ('tn', 512, 2048, 6400, 'dW  = X^T dZ   (512x2048, K=6400)'),
    ('tn', 512, 2048, 6080, 'dWh = H^T dZ   (512x2048, K=6080)'),
    ('nn', 1600, 2048, 512, 'prog x-proj    (1600x2048, K=512)'),
    ('nn', 3200, 512, 512, 'rn fc2 fwd     (3200x512, K=512)'),
    ('tn', 512, 512, 3200, 'rn fc2 dW      (512x512, K=3200)'),
    ('nn', 4096, 4096, 4096, 'square 4096'),
]
VARIANTS = [('auto', -1), ('64x64', 0), ('128x64', 4), ('128x128', 1), ('dma64 s4', 8), ('dma64 s3', 9),
            ('dma128x64 s3', 10), ('dma128 s2', 11), ('dma128 s3', 12)]


def timed(fn, reps=20):
    for _ in range(2):
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
    lib = load()
    g = torch.Generator().manual_seed(0)
    splits = [int(a) for a in sys.argv[1:]] or [0]
    for kind, M, N, Kd, label in SHAPES:
        if kind == 'nn':
            A, B = torch.rand(M, Kd, generator=g).cuda() - 0.5, torch.rand(Kd, N, generator=g).cuda() - 0.5
            fn = lambda: K.matmul_nn(A, B, out=C)
        elif kind == 'nt':
            A, B = torch.rand(M, Kd, generator=g).cuda() - 0.5, torch.rand(N, Kd, generator=g).cuda() - 0.5
            fn = lambda: K.matmul_nt(A, B, out=C)
        else:
            A, B = torch.rand(Kd, M, generator=g).cuda() - 0.5, torch.rand(Kd, N, generator=g).cuda() - 0.5
            fn = lambda: K.matmul_tn(A, B, out=C)
        C = torch.empty(M, N, device='cuda')
        K.SCRATCH.reserve(16 * M * N * 4)
        fl = 2.0 * M * N * Kd
        print(label)
        for sp in splits:
            res, same = {}, {}
            lib.d2p_gemm_force_plan(0, sp)
            fn()
            ref = C.clone()
            for rnd in range(3):
                for name, tile in VARIANTS:
                    lib.d2p_gemm_force_plan(tile, sp if tile >= 0 else 0)
                    C.zero_()
                    res.setdefault(name, []).append(timed(fn))
                    if tile >= 0:
                        same[name] = bool(torch.equal(C, ref))
            lib.d2p_gemm_force_plan(-1, 0)
            print('  split %d: ' % sp + '  '.join('%s %.0fus %.0fTF%s' % (
                n, min(t) * 1e6, fl / min(t) / 1e12, '' if same.get(n, True) else ' DIFF') for n, t in res.items()))
        err = (C.double() - (A.double() @ B.double() if kind == 'nn' else A.double() @ B.double().t() if kind == 'nt'
                             else A.double().t() @ B.double())).abs().max().item()
        print('  max |err| vs fp64 of the last variant: %.3g' % err)


if __name__ == '__main__':
    main()
