#!/usr/bin/env python
"""The 48-channel State_Encoder layers at BASELINE config 4's size (6 400 frames; config 5: 8 000): the wide back end
(conv_wide.hip) against the implicit-GEMM path, per direction.  us per call, TFLOP/s, fraction of the fp32 MFMA peak."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402

PEAK = 157.3
LAYERS = [(20, 32), (10, 48), (5, 48)]


def timed(fn, reps=20):
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
    build.build_library()
    lib = load()
    B, G, T = (32, 10, 20) if len(sys.argv) < 2 or sys.argv[1] != 'k25' else (16, 25, 20)
    N = B * G * T
    g = torch.Generator().manual_seed(0)
    for H, Cin in LAYERS:
        Ho = (H + 1) // 2
        x_ext = torch.empty(N * H * H * Cin + G * Cin, device='cuda')
        x = x_ext[:N * H * H * Cin].view(N, H, H, Cin)
        x.copy_((torch.rand(N, H, H, Cin, generator=g) - 0.5))
        sc, sh = (torch.rand(G, Cin, generator=g) + 0.5).cuda(), (torch.rand(G, Cin, generator=g) - 0.5).cuda()
        pad = x_ext[N * H * H * Cin:].view(G, Cin)
        pad.copy_(-sh / sc)
        w = (torch.rand(3, 3, Cin, 48, generator=g) - 0.5).cuda()
        b = torch.zeros(48).cuda()
        dy = (torch.rand(N, Ho, Ho, 48, generator=g) - 0.5).cuda()
        y = torch.empty(N, Ho, Ho, 48, device='cuda')
        dx = torch.empty(N, H, H, Cin, device='cuda')
        dw = torch.empty(3, 3, Cin, 48, device='cuda')
        fl = 2.0 * N * Ho * Ho * 9 * Cin * 48
        S = K.conv_bn_slices((N, H, H, Cin), 48, G, T)
        st = torch.zeros(max(1, G * S * 48 * 2), dtype=torch.float64, device='cuda')
        fns = [('fwd', lambda: K.conv_fwd(x, w, b, act=1, out=y)),
               ('wgrad', lambda: K.conv_wgrad(x, dy, dw)),
               ('wgrad+affine', lambda: K.conv_wgrad_bn(x, dy, dw, G, T, (sc, sh))),
               ('dgrad', lambda: K.conv_dgrad(dy, w, (N, H, H, Cin), dx=dx))]
        if S > 0:
            fns.insert(1, ('fwd+stats', lambda: K.conv_fwd_bn(x, w, b, G, T, S, st, act=1, out=y)))
            fns.insert(2, ('fwd+stats+affine', lambda: K.conv_fwd_bn(x, w, b, G, T, S, st, act=1, out=y, in_affine=(sc, sh, pad))))
        print('conv %dx%dx%d -> 48  (N=%d, %.2f GFLOP, %.1f us at peak; S=%d)' % (H, H, Cin, N, fl / 1e9, fl / PEAK / 1e6, S))
        for name, fn in fns:
            out = []
            for label, sel in (('gemm', 0), ('default', 2)):
                if label == 'gemm' and ('stats' in name or 'affine' in name):
                    continue
                lib.d2p_conv_set_direct(sel, sel, sel)
                t = min(timed(fn), timed(fn))
                out.append('%s %.1f us %.1f TF/s (%.3f)' % (label, t * 1e6, fl / t / 1e12, fl / t / 1e12 / PEAK))
            lib.d2p_conv_set_direct(2, 2, 2)
            print('  %-18s ' % name + ' | '.join(out), flush=True)


if __name__ == '__main__':
    main()
