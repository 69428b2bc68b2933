#!/usr/bin/env python
"""Within-process A/B of the conv entry points (direct kernels vs implicit-im2col GEMM) on the
layer geometries of the Karel and ViZDoom demonstration encoders.  Run on the GPU box.
Prints us per call, TFLOP/s and algorithmic GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402

KAREL = [(6400, 8, 8, 16, 16, False), (6400, 4, 4, 16, 32, False), (6400, 2, 2, 32, 48, False)]
VIZDOOM = [(6400, 80, 80, 4, 16, True), (6400, 40, 40, 16, 32, False)]


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
    which = sys.argv[1] if len(sys.argv) > 1 else 'karel'
    shapes = KAREL if which == 'karel' else VIZDOOM if which == 'vizdoom' else KAREL + VIZDOOM
    g = torch.Generator().manual_seed(0)
    for N, H, W, Cin, Cout, u8 in shapes:
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        if u8:
            x = torch.randint(0, 256, (N, H, W, Cin), generator=g, dtype=torch.uint8).cuda()
        else:
            x = (torch.rand(N, H, W, Cin, generator=g) - 0.5).cuda()
        w = (torch.rand(3, 3, Cin, Cout, generator=g) - 0.5).cuda()
        b = torch.zeros(Cout).cuda()
        dy = (torch.rand(N, Ho, Wo, Cout, generator=g) - 0.5).cuda()
        y = torch.empty(N, Ho, Wo, Cout, device='cuda')
        dx = torch.empty(N, H, W, Cin, device='cuda')
        dw = torch.empty(3, 3, Cin, Cout, device='cuda')
        fl = 2.0 * N * Ho * Wo * 9 * Cin * Cout
        xb = x.numel() * x.element_size()
        by = {'fwd': xb + y.numel() * 4, 'dgrad': dy.numel() * 4 + dx.numel() * 4, 'wgrad': xb + dy.numel() * 4}
        fns = {'fwd': lambda: K.conv_fwd(x, w, b, act=1, out=y),
               'wgrad': lambda: K.conv_wgrad(x, dy, dw)}
        if not u8:
            fns['dgrad'] = lambda: K.conv_dgrad(dy, w, (N, H, W, Cin), dx=dx)
        print('conv %dx%dx%d -> %d  (N=%d%s)' % (H, W, Cin, Cout, N, ', u8' if u8 else ''))
        for name, fn in fns.items():
            out = []
            lib.d2p_conv_set_direct(0, 0, 0)
            t = min(timed(fn), timed(fn))
            out.append('gemm %.1fus %.1fTF %.0fGB/s' % (t * 1e6, fl / t / 1e12, by[name] / t / 1e9))
            lib.d2p_conv_set_direct(2, 2, 2)
            knobs = {'fwd': [(1, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0), (8, 0, 0)],
                     'dgrad': [(0, 1, 0), (0, 2, 0), (0, 4, 0), (0, 8, 0)],
                     'wgrad': [(0, 0, 128), (0, 0, 256), (0, 0, 512), (0, 0, 1024), (0, 0, 2048)]}[name]
            for kn in knobs:
                lib.d2p_conv_direct_tune(*kn)
                t = min(timed(fn), timed(fn))
                out.append('direct%s %.1fus %.1fTF %.0fGB/s' % (str(max(kn)), t * 1e6, fl / t / 1e12, by[name] / t / 1e9))
            lib.d2p_conv_direct_tune(-1, 2, -1)          # back to the automatic settings
            print('  %-5s ' % name + ' | '.join(out))


if __name__ == '__main__':
    main()
