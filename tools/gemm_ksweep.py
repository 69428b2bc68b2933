#!/usr/bin/env python
"""GEMM time over K at a fixed 6400 x 2048 output, per forced tile: slope (main-loop rate) and intercept (per-launch cost)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import kernels as K
from demo2program_amd.lib import load
lib = load()
def timed(fn, reps=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
M, N = 6400, 2048
C = torch.empty(M, N, device='cuda')
for name, tile in [('64x64', 0), ('128x64', 4), ('128x128', 1), ('dma64s4', 8), ('dma64s3', 9), ('dma128x64s3', 10), ('dma128s3', 12)]:
    row = []
    for Kd in (128, 256, 512, 1024, 2048, 4096):
        A = torch.rand(M, Kd, device='cuda') - 0.5; B = torch.rand(Kd, N, device='cuda') - 0.5
        lib.d2p_gemm_force_plan(tile, 1)
        t = min(timed(lambda: K.matmul_nn(A, B, out=C)) for _ in range(3))
        row.append((Kd, t))
    b = (row[-1][1] - row[2][1]) / (row[-1][0] - row[2][0])
    a = row[2][1] - b * 512
    print(name, ' '.join('K=%d %.0fus' % r for r in row), '| slope %.4f us/K -> %.0f TF inner, intercept at K=512 fit %.1f us' % (b, 2.0 * M * N / b / 1e6, a))
lib.d2p_gemm_force_plan(-1, 0)
