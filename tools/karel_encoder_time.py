#!/usr/bin/env python
"""Microseconds per call of the one-launch Karel State_Encoder forward (d2p_karel_encoder_fwd) and of the separate
launches it replaces (conv -> batch norm x 3 + transpose), at the headline geometry (B=32, k=10, T=20, uint8 frames).

  python tools/karel_encoder_time.py [B] [G] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402


def timed(fn, reps=50):
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


def main():
    build.build_library()
    B, G, T = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (32, 10, 20)
    NF = B * G * T
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(NF, 8, 8, 16, generator=g) < 0.15).to(torch.uint8).cuda()
    w = [(torch.randn(3, 3, ci, co, generator=g) * 0.1).cuda() for ci, co in ((16, 16), (16, 32), (32, 48))]
    b = [torch.zeros(c, device='cuda') for c in (16, 32, 48)]
    gam = [torch.ones(c, device='cuda') for c in (16, 32, 48)]
    bet = [torch.zeros(c, device='cuda') for c in (16, 32, 48)]
    a = [torch.empty(NF, 4, 4, 16, device='cuda'), torch.empty(NF, 2, 2, 32, device='cuda'),
         torch.empty(NF, 1, 1, 48, device='cuda')]
    y = [torch.empty(NF * 16, 16, device='cuda'), torch.empty(NF * 4, 32, device='cuda'), torch.empty(NF, 48, device='cuda')]
    stat = lambda: [torch.empty(G, c, device='cuda') for c in (16, 32, 48)]       # noqa: E731
    mean, rstd, var = stat(), stat(), stat()
    feats_tm = torch.empty(T, B * G, 48, device='cuda')
    ws = torch.empty(max(1, K._load_lib().d2p_karel_encoder_ws_bytes(B, G, T)), dtype=torch.uint8, device='cuda')

    def fused():
        K.karel_encoder_fwd(x, B, G, T, w, b, gam, bet, a, y[:2], feats_tm, mean, rstd, var, ws)

    def chain():
        cur = x
        for l, (cout, hw) in enumerate(((16, 4), (32, 2), (48, 1))):
            K.conv_fwd(cur, w[l], b[l], act=1, out=a[l])
            K.bn_fwd(a[l].view(NF * hw * hw, cout), gam[l], bet[l], G, T * hw * hw, y=y[l], mean=mean[l], rstd=rstd[l])
            cur = y[l].view(NF, hw, hw, cout)
        K.transpose_rt(y[2].view(B * G, T, 48), B * G, T, 48, out=feats_tm)
    print('separate launches: %.1f us' % timed(chain))
    if K.karel_encoder_ok(B, G, T):
        print('one launch:        %.1f us' % timed(fused))
        nwg = 256
        tr = torch.zeros(nwg, 10, dtype=torch.int64, device='cuda')
        from demo2program_amd.lib import call
        call.d2p_karel_encoder_set_trace(tr.data_ptr())
        fused()
        torch.cuda.synchronize()
        call.d2p_karel_encoder_set_trace(None)
        t = tr.cpu().double()
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        names = ['start', 'layer 1 products', 'statistics 1', 'normalise 1', 'layer 2 products', 'statistics 2',
                 'normalise 2', 'layer 3 products', 'statistics 3', 'normalise 3 + transpose']
        print('%d workgroups; stamps in us from the first workgroup\'s start (mean / max over workgroups):' % t.shape[0])
        for i in range(10):
            d = (t[:, i] - (t[:, i - 1] if i else t0)) / 100.0
            print('   %-26s +%6.2f / %6.2f   (at %6.2f / %6.2f)' % (names[i], d.mean(), d.max(), ((t[:, i] - t0) / 100).mean(),
                                                                  ((t[:, i] - t0) / 100).max()))
    # ---- backward: the chain of the separate launches against d2p_karel_encoder_bwd (+ its combine launch)
    fused()
    dfeat_tm = torch.randn(T, B * G, 48, generator=g).cuda()
    dfeat = torch.empty(B * G, T, 48, device='cuda')
    dw = [torch.empty_like(t) for t in w]
    db, dgam, dbet = ([torch.empty(c, device='cuda') for c in (16, 32, 48)] for _ in range(3))
    da = [torch.empty(NF * hw * hw, c, device='cuda') for c, hw in ((16, 4), (32, 2), (48, 1))]
    dx = [None, torch.empty(NF, 4, 4, 16, device='cuda'), torch.empty(NF, 2, 2, 32, device='cuda')]
    xin = [x, y[0].view(NF, 4, 4, 16), y[1].view(NF, 2, 2, 32)]

    def chain_bwd():
        dy = K.transpose_rt(dfeat_tm, T, B * G, 48, out=dfeat).view(NF, 48)
        for l, (cin, cout, hw, hin) in reversed(list(enumerate(((16, 16, 4, 8), (16, 32, 2, 4), (32, 48, 1, 2))))):
            K.bn_bwd(a[l].view(NF * hw * hw, cout), dy.view(NF * hw * hw, cout), gam[l], mean[l], rstd[l], G, T * hw * hw, True,
                     dgam[l], dbet[l], dx=da[l], dbias=db[l])
            K.conv_wgrad(xin[l], da[l].view(NF, hw, hw, cout), dw[l])
            if l > 0:
                dy = K.conv_dgrad(da[l].view(NF, hw, hw, cout), w[l], (NF, hin, hin, cin), dx=dx[l])
    print('backward, separate launches: %.1f us' % timed(chain_bwd))
    if K.karel_encoder_bwd_ok(B, G, T):
        wsb = torch.empty(K._load_lib().d2p_karel_encoder_bwd_ws_bytes(B, G, T), dtype=torch.uint8, device='cuda')

        def fused_bwd():
            K.karel_encoder_bwd(x, dfeat_tm, B, G, T, w, gam, bet, a, mean, rstd, dw, db, dgam, dbet, wsb)
        print('backward, one launch + combine: %.1f us' % timed(fused_bwd))
        tr = torch.zeros(256, 16, dtype=torch.int64, device='cuda')
        from demo2program_amd.lib import call
        call.d2p_karel_encoder_set_trace(tr.data_ptr())
        fused_bwd()
        torch.cuda.synchronize()
        call.d2p_karel_encoder_set_trace(None)
        t = tr.cpu().double()
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        names = ['start', 'loads -> LDS', 'sums 3', 'exchange 3', 'apply 3', 'dgrad 3 + wgrad 3', 'sums 2 + zero fill',
                 'exchange 2', 'apply 2', 'dgrad 2 + wgrad 2', 'sums 1', 'exchange 1', 'apply 1', 'wgrad 1 products',
                 'tree + slab']
        print('%d workgroups; stamps in us from the first workgroup\'s start (mean / max over workgroups):' % t.shape[0])
        for i in range(15):
            d = (t[:, i] - (t[:, i - 1] if i else t0)) / 100.0
            print('   %-26s +%6.2f / %6.2f   (at %6.2f / %6.2f)' % (names[i], d.mean(), d.max(), ((t[:, i] - t0) / 100).mean(),
                                                                  ((t[:, i] - t0) / 100).max()))
    print('error word: 0x%x' % K.lstm_persist_error())


if __name__ == '__main__':
    main()
