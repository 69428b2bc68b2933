#!/usr/bin/env python
"""The 16 -> 32 layer's input gradient at config 4's size (6 400 frames of 40x40x16 <- 20x20x32), plain and with the first
layer's batch-norm-backward sums folded in: the block-form kernel of conv_wide.hip against the row-strip kernel
(d2p_conv_set_direct(2, 3, 2))."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402
from bench_conv_wide import timed  # noqa: E402


def main():
    build.build_library()
    lib = load()
    B, G, T = 32, 10, 20
    N, H, Cin, Cout = B * G * T, 40, 16, 32
    g = torch.Generator().manual_seed(0)
    dy = (torch.rand(N, 20, 20, Cout, generator=g) - 0.5).cuda()
    w = (torch.rand(3, 3, Cin, Cout, generator=g) - 0.5).cuda()
    act = (torch.rand(N, H, H, Cin, generator=g) - 0.5).cuda()
    mu, rs = (torch.rand(G, Cin, generator=g) - 0.5).cuda(), (torch.rand(G, Cin, generator=g) + 0.5).cuda()
    dx = torch.empty(N, H, H, Cin, device='cuda')
    fl = 2.0 * N * 400 * 9 * Cin * Cout
    by = dy.numel() * 4 + dx.numel() * 4
    res = {}
    for label, sel in (('row-strip', 3), ('block form', 2)):
        lib.d2p_conv_set_direct(2, sel, 2)
        S = K.conv_dgrad_bn_slices((N, H, H, Cin), Cout, G, T)
        st = torch.zeros(G * S * Cin * 2, dtype=torch.float64, device='cuda')
        t0 = min(timed(lambda: K.conv_dgrad(dy, w, (N, H, H, Cin), dx=dx)) for _ in range(2))
        ref = dx.clone()
        t1 = min(timed(lambda: K.conv_dgrad_bn(dy, w, (N, H, H, Cin), act, mu, rs, G, T, st, S, dx=dx)) for _ in range(2))
        res[label] = (ref, st.view(G, S, Cin, 2).sum(1).clone())
        print('%-11s plain %.1f us (%.0f GB/s algorithmic, %.1f TF/s) | + batch-norm sums %.1f us (%.0f GB/s incl. the activation read)  S=%d'
              % (label, t0 * 1e6, by / t0 / 1e9, fl / t0 / 1e12, t1 * 1e6, (by + act.numel() * 4) / t1 / 1e9, S), flush=True)
    lib.d2p_conv_set_direct(2, 2, 2)
    # ---- forward: statistics out, the input through the first layer's batch-norm apply
    xe = torch.empty(N * H * H * Cin + G * Cin, device='cuda')
    x = xe[:N * H * H * Cin].view(N, H, H, Cin)
    x.copy_(act)
    sc, sh = (torch.rand(G, Cin, generator=g) + 0.5).cuda(), (torch.rand(G, Cin, generator=g) - 0.5).cuda()
    pad = xe[N * H * H * Cin:].view(G, Cin)
    pad.copy_(-sh / sc)
    bias = torch.zeros(Cout, device='cuda')
    y = torch.empty(N, 20, 20, Cout, device='cuda')
    outs = {}
    for label, sel in (('gather kernel', 3), ('block form', 2)):
        lib.d2p_conv_set_direct(sel, 2, 2)
        S = K.conv_bn_slices((N, H, H, Cin), Cout, G, T)
        st = torch.zeros(G * S * Cout * 2, dtype=torch.float64, device='cuda')
        t0 = min(timed(lambda: K.conv_fwd(x, w, bias, act=1, out=y)) for _ in range(2))
        t1 = min(timed(lambda: K.conv_fwd_bn(x, w, bias, G, T, S, st, act=1, out=y, in_affine=(sc, sh, pad))) for _ in range(2))
        outs[label] = (y.clone(), st.view(G, S, Cout, 2).sum(1).clone())
        print('forward %-13s plain %.1f us (%.1f TF/s, %.3f of peak) | + statistics + input affine %.1f us (%.3f of peak)  S=%d'
              % (label, t0 * 1e6, fl / t0 / 1e12, fl / t0 / 1e12 / 157.3, t1 * 1e6, fl / t1 / 1e12 / 157.3, S), flush=True)
    lib.d2p_conv_set_direct(2, 2, 2)
    ya, yb = outs['gather kernel'], outs['block form']
    print('forward: max |y difference| %.3e (scale %.3e); statistics: max rel difference %.3e' %
          (float((ya[0] - yb[0]).abs().max()), float(ya[0].abs().max()), float(((ya[1] - yb[1]).abs() / (ya[1].abs() + 1e-3)).max())))
    a, b = res['row-strip'], res['block form']
    print('max |dx difference| %.3e (scale %.3e); sums: max rel difference %.3e' %
          (float((a[0] - b[0]).abs().max()), float(a[0].abs().max()), float(((a[1] - b[1]).abs() / (a[1].abs() + 1e-3)).max())))


if __name__ == '__main__':
    main()
