#!/usr/bin/env python
"""Within-process A/B of the dense GEMM entry points on the shapes of one training step
(run on the GPU box).  Prints us and TFLOP/s per shape for each knob setting, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402

SHAPES = [  # (kind, M, N, K, label)
    ('nn', 320, 512, 512, 'rn fc1 P/Q      (320x512, K=512)'),
    ('nt', 320, 512, 512, 'rn fc1 dfeat    (320x512, K=512)'),
    ('tn', 512, 512, 320, 'rn fc1 dW       (512x512, K=320)'),
    ('tn', 512, 2048, 6400, 'dW  = X^T dZ   (512x2048, K=6400)'),
    ('nn', 6400, 2048, 512, 'z   = X Wx     (6400x2048, K=512)'),
    ('nt', 6400, 512, 2048, 'dX  = dZ Wx^T  (6400x512, K=2048)'),
    ('nn', 3200, 512, 512, 'rn fc2 fwd     (3200x512, K=512)'),
    ('tn', 512, 512, 3200, 'rn fc2 dW      (512x512, K=3200)'),
    ('nn', 1600, 2048, 512, 'prog x-proj    (1600x2048, K=512)'),
    ('nn', 6400, 2048, 48, 'demo x-proj    (6400x2048, K=48)'),
]


def timed(fn, reps=30):
    for _ in range(3):
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
    only = len(sys.argv) > 1 and sys.argv[1] == 'small'
    if only:
        del SHAPES[3:]
        SHAPES.extend([('nn', 3200, 512, 512, 'rn fc2 fwd (3200x512, K=512)'), ('tn', 512, 512, 3200, 'rn fc2 dW'),
                       ('tn', 512, 2048, 320, 'h0 dWh (512x2048, K=320)'), ('nn', 1568, 2048, 512, 'prog x-proj'),
                       ('tn', 48, 2048, 6400, 'demo dWx (48x2048, K=6400)'), ('nt', 6400, 48, 2048, 'demo dX'),
                       ('nn', 6400, 2048, 48, 'demo x-proj K=48'), ('nt', 3200, 512, 512, 'rn fc2 dy1')])
    build.build_library()
    lib = load()
    g = torch.Generator().manual_seed(0)
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
        fl = 2.0 * M * N * Kd
        variants = [('auto', -1, 0), ('64x64', 0, 0), ('64x64 s2', 0, 2), ('64x64 s4', 0, 4), ('128x64', 4, 0),
                    ('128x64 s2', 4, 2), ('128x128', 1, 0), ('128x128 s2', 1, 2), ('128x128 s4', 1, 4),
                    ('128x128 s8', 1, 8)]
        res = {}
        ws_need = 16 * M * N * 4
        K.SCRATCH.reserve(ws_need)
        if not only:
            variants = variants + [('auto select', -1, 0), ('64x64 s1', 0, 1), ('64x64 s3', 0, 3), ('64x64 s6', 0, 6), ('64x64 s8', 0, 8), ('128x64 s4', 4, 4)]
        if only:
            variants = variants + [('auto select', -1, 0),('64x64 bk32', 0, 0), ('64x64 bk32 s2', 0, 2), ('64x64 bk32 s4', 0, 4),
                                   ('32x32ksr', 7, 0), ('32x32ksr s2', 7, 2), ('32x32ksr s4', 7, 4),
                                   ('128x64', 4, 0), ('128x64 s2', 4, 2), ('128x32', 2, 0), ('128x32 s4', 2, 4)]
        for rnd in range(2):
            for name, tile, sp in variants:
                if sp and Kd // sp < 64:
                    continue
                if only and name.split()[0] not in ('auto', '64x64', '32x32ksr', '128x64', '128x32'):
                    continue
                opt = 1 if 'bk32' in name else 0
                opt |= 4 if 'select' in name else 0
                lib.d2p_gemm_set_option(opt)
                lib.d2p_gemm_force_plan(tile, sp)
                res.setdefault(name, []).append(timed(fn, 20))
        lib.d2p_gemm_force_plan(-1, 0)
        lib.d2p_gemm_set_option(0)
        print(label)
        print('    ' + '  '.join('%s %.0fus %.0fTF' % (n, min(t) * 1e6, fl / min(t) / 1e12) for n, t in res.items()))


if __name__ == '__main__':
    main()
