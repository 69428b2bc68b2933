#!/usr/bin/env python
"""Microseconds of the batch-norm-folding conv launches against the plain ones at BASELINE config 4's sizes
(6 400 frames: conv1 80x80x4 -> 40x40x16 with statistics; conv2 40x40x16 -> 20x20x32 with the input affine and
statistics; conv2's weight gradient with the input affine).   python tools/bench_conv_bn.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
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
B, G, T = 32, 10, 20
N = int(sys.argv[1]) if len(sys.argv) > 1 else B * G * T
B = N // (G * T)
g = torch.Generator().manual_seed(1)
xu = torch.randint(0, 256, (N, 80, 80, 4), generator=g, dtype=torch.uint8).cuda()
w1, b1 = (torch.randn(3, 3, 4, 16, generator=g) * 0.02).cuda(), torch.zeros(16).cuda()
w2, b2 = (torch.randn(3, 3, 16, 32, generator=g) * 0.1).cuda(), torch.zeros(32).cuda()
a1_ext = torch.empty(N * 1600 * 16 + G * 16, device='cuda')       # (+ the pad pixels of the input affine)
a1, pad = a1_ext[:N * 1600 * 16].view(N, 40, 40, 16), a1_ext[N * 1600 * 16:].view(G, 16)
a2 = torch.empty(N, 20, 20, 32, device='cuda')
S1, S2 = K.conv_bn_slices((N, 80, 80, 4), 16, G, T), K.conv_bn_slices((N, 40, 40, 16), 32, G, T)
st1 = torch.zeros(G * S1 * 16 * 2, dtype=torch.float64, device='cuda')
st2 = torch.zeros(G * S2 * 32 * 2, dtype=torch.float64, device='cuda')
sc, sh = torch.rand(G, 16, device='cuda') + 0.5, torch.randn(G, 16, device='cuda')
pad.copy_(-sh / sc)
print('frames %d, slices per index: conv1 %d, conv2 %d' % (N, S1, S2))
print('conv1 fwd   plain %.1f us | + statistics %.1f us' % (
    timed(lambda: K.conv_fwd(xu, w1, b1, act=1, out=a1)),
    timed(lambda: K.conv_fwd_bn(xu, w1, b1, G, T, S1, st1, act=1, out=a1))))
print('conv2 fwd   plain %.1f us | + statistics %.1f us | + statistics + input affine %.1f us' % (
    timed(lambda: K.conv_fwd(a1, w2, b2, act=1, out=a2)),
    timed(lambda: K.conv_fwd_bn(a1, w2, b2, G, T, S2, st2, act=1, out=a2)),
    timed(lambda: K.conv_fwd_bn(a1, w2, b2, G, T, S2, st2, act=1, out=a2, in_affine=(sc, sh, pad)))))
dy2 = torch.randn(N, 20, 20, 32, generator=g).cuda() if False else torch.randn(N, 20, 20, 32, device='cuda')
dw = torch.empty(3, 3, 16, 32, device='cuda')
print('conv2 wgrad plain %.1f us | + input affine %.1f us' % (
    timed(lambda: K.conv_wgrad(a1, dy2, dw)), timed(lambda: K.conv_wgrad_bn(a1, dy2, dw, G, T, (sc, sh)))))
y = torch.empty(N * 1600, 16, device='cuda')
mean, rstd = torch.zeros(G, 16, device='cuda'), torch.ones(G, 16, device='cuda')
gam, bet = torch.ones(16, device='cuda'), torch.zeros(16, device='cuda')
print('batch norm of conv1 (separate launches): statistics + apply %.1f us; apply alone %.1f us' % (
    timed(lambda: K.bn_fwd(a1.view(N * 1600, 16), gam, bet, G, T * 1600, y=y)),
    timed(lambda: K.bn_apply_fwd(a1.view(N * 1600, 16), gam, bet, mean, rstd, G, T * 1600, y=y))))
Sd = K.conv_dgrad_bn_slices((N, 40, 40, 16), 32, G, T)
std = torch.zeros(G * Sd * 16 * 2, dtype=torch.float64, device='cuda')
dx1 = torch.empty(N, 40, 40, 16, device='cuda')
print('conv2 dgrad plain %.1f us | + conv1 batch-norm-backward sums %.1f us (slices %d); the separate sums pass: see below' % (
    timed(lambda: K.conv_dgrad(dy2, w2, (N, 40, 40, 16), dx=dx1)),
    timed(lambda: K.conv_dgrad_bn(dy2, w2, (N, 40, 40, 16), a1, mean, rstd, G, T, std, Sd, dx=dx1)), Sd))
coef = torch.empty(G, 16, 4, device='cuda')
dg, db = torch.empty(16, device='cuda'), torch.empty(16, device='cuda')
da1 = torch.empty(N * 1600, 16, device='cuda')
dw1, dbias = torch.empty(3, 3, 4, 16, device='cuda'), torch.empty(16, device='cuda')
a1f, dx1f = a1.view(N * 1600, 16), dx1.view(N * 1600, 16)
print('conv1 backward: batch-norm backward (sums + apply) %.1f us + weight gradient %.1f us | sums + coefficients %.1f us '
      '(coefficients from the producer\'s sums %.1f us) + folded weight gradient %.1f us' % (
          timed(lambda: K.bn_bwd(a1f, dx1f, gam, mean, rstd, G, T * 1600, True, dg, db, dx=da1, dbias=dbias)),
          timed(lambda: K.conv_wgrad(xu, da1.view(N, 40, 40, 16), dw1)),
          timed(lambda: K.bn_bwd_coef(a1f, dx1f, gam, mean, rstd, G, T * 1600, coef, dg, db)),
          timed(lambda: K.bn_bwd_coef(a1f, dx1f, gam, mean, rstd, G, T * 1600, coef, dg, db, sums=(std, Sd))),
          timed(lambda: K.conv_wgrad_bnbwd(xu, a1, dx1, coef, G, T, dw1, dbias))))
