#!/usr/bin/env python
"""The write-bound floor of the 6400 x 2048 GEMM output: fill / copy / K = 32..128 GEMMs, and equal-flop shapes with smaller outputs."""
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
D = torch.empty(M, N, device='cuda')
print('fill 52MB: %.1f us' % min(timed(lambda: C.fill_(1.0)) for _ in range(3)))
print('copy 52MB: %.1f us' % min(timed(lambda: D.copy_(C)) for _ in range(3)))
E = torch.empty(8, device='cuda')
print('tiny fill: %.1f us' % min(timed(lambda: E.fill_(1.0)) for _ in range(3)))
for Kd in (32, 64, 128):
    A = torch.rand(M, Kd, device='cuda') - 0.5; B = torch.rand(Kd, N, device='cuda') - 0.5
    for name, tile in [('64x64', 0), ('128x128', 1), ('dma64s3', 9)]:
        lib.d2p_gemm_force_plan(tile, 1)
        print('K=%d %s: %.1f us' % (Kd, name, min(timed(lambda: K.matmul_nn(A, B, out=C)) for _ in range(3))))
# smaller output, same flops: 1600 x 2048 x 2048
lib.d2p_gemm_force_plan(-1, 0)
for (m, n, k) in [(1600, 2048, 2048), (6400, 2048, 512), (6400, 512, 2048), (3200, 2048, 1024)]:
    A = torch.rand(m, k, device='cuda') - 0.5; B = torch.rand(k, n, device='cuda') - 0.5
    Cc = torch.empty(m, n, device='cuda')
    t = min(timed(lambda: K.matmul_nn(A, B, out=Cc)) for _ in range(3))
    print('nn %dx%dx%d: %.1f us %.0f TF' % (m, n, k, t, 2.0 * m * n * k / t / 1e6))
