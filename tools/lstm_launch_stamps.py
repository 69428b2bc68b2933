#!/usr/bin/env python
"""Where a persistent recurrence's launch time goes OUTSIDE its steady tick loop (the per-call
fixed cost).  Runs on the GPU box against a DIAGNOSTIC build of the library (lstm_persist.hip with -DD2P_PS_STAMPS,
`python demo2program_amd/build.py --stamps`): wave 0 (MFMA) and wave 4 (publish / prefetch) of every workgroup of every
persistent launch leave the chip-wide 100 MHz counter at the launch's boundaries --

    0 entry | 1 tables + initial state in LDS (weights requested) | 3 last tick done
    4 bias-gradient exchange done (backward) | 5 final stores issued | 6 final stores out | 7 geometry word

(nothing inside the tick loops: a stamp there changes hipcc's schedule of the loop).  Printed per launch: kernel span
(first entry -> last exit), the spread of the entries (dispatch), and per row domain -- its slowest workgroup --
prologue, the tick loop (with the first tick's pipeline fill), us per tick, the epilogue pieces, exit.  The launch's
critical path is the workgroup that exits last; what it spends outside its tick loop is the launch's fixed cost.

    python tools/lstm_launch_stamps.py [--mode step|seq] [--steps 3]"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ['D2P_LIB_PATH'] = os.path.join(ROOT, 'demo2program_amd', 'csrc', 'libd2p_hip_stamps.so')

import torch  # noqa: E402

MAXB, K = 512, 8
TICK_NS = 10.0           # s_memrealtime: 100 MHz


def decode(buf, n_launches):
    a = buf.cpu().numpy().reshape(-1, MAXB, 2, K).astype(np.int64)
    out = []
    for s in range(min(n_launches, a.shape[0])):
        g = a[s]
        live = g[:, 0, 0] != 0
        nb = int(live.sum())
        if nb == 0:
            continue
        m = g[live, 0, :]            # MFMA wave 0 of every workgroup
        p = g[live, 1, :]            # publish wave
        geom = m[:, 7]
        nrs, T = geom & 0xff, (geom >> 8) & 0xff
        dom, nt, seq = (geom >> 16) & 0xff, (geom >> 24) & 0xff, (geom >> 40) & 0xff
        t0 = min(m[:, 0].min(), p[:, 0].min())
        end_m = np.where(m[:, 6] != 0, m[:, 6], m[:, 3])
        end = np.maximum(end_m, p[:, 6])
        out.append(dict(blocks=nb, m=m, p=p, nrs=nrs, T=T, dom=dom, nt=nt, seq=seq, t0=t0, end=end))
    return out


def us(x):
    return x * TICK_NS * 1e-3


def describe(L, name):
    m, p, t0, end = L['m'], L['p'], L['t0'], L['end']
    span = us(end.max() - t0)
    ticks = L['nrs'] * L['T']
    has4 = bool((m[:, 4] != 0).any())
    lines = []
    lines.append('%s: %d workgroups, span %.1f us; entries spread over %.1f us (median %.1f); exits spread %.1f us'
                 % (name, L['blocks'], span, us(m[:, 0].max() - t0), us(np.median(m[:, 0]) - t0),
                    us(end.max() - end.min())))
    # per (sequence, domain): one line for the domain's slowest workgroup
    keys = sorted(set(zip(L['seq'].tolist(), L['dom'].tolist())))
    lines.append('   seq dom | nrs  T ticks | entry  prolog   loop  /tick |  db   stores drain |  exit  (us; slowest workgroup of the domain; publish wave: prologue, exit)')
    worst = None
    for (sq, d) in keys:
        sel = np.where((L['seq'] == sq) & (L['dom'] == d))[0]
        i = sel[np.argmax(end[sel])]
        n = int(ticks[i])
        steady = us(m[i, 3] - m[i, 1]) / max(n, 1)
        db = us(m[i, 4] - m[i, 3]) if has4 else 0.0
        st5 = m[i, 4] if has4 else m[i, 3]
        lines.append('   %3d %3d | %3d %3d %5d | %5.1f  %6.1f %6.1f %6.2f | %4.1f %6.1f %6.1f | %6.1f   (pub %5.1f, %6.1f)'
                     % (sq, d, L['nrs'][i], L['T'][i], n, us(m[i, 0] - t0), us(m[i, 1] - m[i, 0]),
                        us(m[i, 3] - m[i, 1]), steady, db, us(m[i, 5] - st5), us(m[i, 6] - m[i, 5]), us(end[i] - t0),
                        us(p[i, 1] - p[i, 0]), us(p[i, 6] - t0)))
        if worst is None or end[i] > end[worst[0]]:
            worst = (i, steady, n)
    i, steady, n = worst
    fixed = span - us(m[i, 3] - m[i, 1])
    lines.append('   critical workgroup: tick loop %.1f us (%d ticks x %.2f); outside the loop %.1f us = entry %.1f + prologue %.1f '
                 '+ epilogue %.1f' % (us(m[i, 3] - m[i, 1]), n, steady, fixed, us(m[i, 0] - t0), us(m[i, 1] - m[i, 0]),
                                     us(end[i] - m[i, 3])))
    return '\n'.join(lines), span, fixed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='step')
    ap.add_argument('--preset', default='karel')
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    from demo2program_amd import lib as L_
    lib = L_.load()
    lib.d2p_lstm_persist_set_stamps.restype = ctypes.c_int
    lib.d2p_lstm_persist_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.d2p_lstm_persist_stamp_launches.restype = ctypes.c_int
    slots = 64
    buf = torch.zeros(slots * MAXB * 2 * K, dtype=torch.int64, device='cuda')

    def collect(fn, label):
        buf.zero_()
        torch.cuda.synchronize()
        lib.d2p_lstm_persist_set_stamps(buf.data_ptr(), buf.numel() * 8)
        fn()
        torch.cuda.synchronize()
        n = lib.d2p_lstm_persist_stamp_launches()
        lib.d2p_lstm_persist_set_stamps(None, 0)
        Ls = decode(buf, n)
        print('==== %s: %d persistent launches stamped' % (label, len(Ls)))
        return Ls

    if args.mode == 'seq':
        from demo2program_amd import kernels as Kn
        from check_lstm_persist import Seq
        Kn.set_lstm_persistent(True)
        # the bench's two-point fit (north_star_targets.recurrent_step: one sequence, T and 4T steps, back-to-back calls)
        # next to the stamped launch: call period - kernel span = what the CALL adds around the kernel
        for (M, T, masked) in ((320, 20, False), (320, 80, False), (320, 20, True)):
            s = Seq(M, 512, T, masked, True, seed=3)
            s.fwd()
            for _ in range(3):
                s.bwd()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                s.bwd()
            e1.record()
            torch.cuda.synchronize()
            period = e0.elapsed_time(e1) * 100.0
            Ls = collect(s.bwd, 'one sequence M=%d U=512 T=%d masked=%d, backward' % (M, T, masked))
            for L in Ls:
                txt, span, fixed = describe(L, 'backward')
                print(txt)
                print('   back-to-back call period %.1f us (events, this build); kernel span %.1f us' % (period, span))
        return

    from demo2program_amd.config import make_config
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    config = make_config(args.preset)
    trainer = Trainer(config, make_train_dir=False)
    batches = [make_batch(config, seed=123 + i) for i in range(4)]
    for b in batches:
        b['s_h'] = b['s_h'].astype(np.uint8)
    feeds = [trainer.model.get_feed_dict(b) for b in batches]
    with trainer.step_stream():              # (the loop's own high-priority stream, as Trainer.train / bench.py)
        for i in range(20):
            trainer.train_step(feeds[i % 4])
    torch.cuda.synchronize()

    def steps():
        with trainer.step_stream():
            for i in range(args.steps):
                trainer.train_step(feeds[i % 4])
    Ls = collect(steps, '%d training steps (%s)' % (args.steps, args.preset))
    per = len(Ls) // args.steps
    names = ['fwd encoder 1', 'fwd encoder 2', 'fwd decoders', 'bwd decoders', 'bwd encoder 2', 'bwd encoder 1']
    tot_span, tot_fixed = 0.0, 0.0
    for j, L in enumerate(Ls[(args.steps - 1) * per:]):          # the last step
        txt, span, fixed = describe(L, 'launch %d (%s)' % (j, names[j] if per == 6 and j < 6 else '?'))
        print(txt)
        tot_span += span
        tot_fixed += fixed
    print('last step: %d launches, spans %.1f us, of which outside the critical workgroups\' tick loops %.1f us'
          % (per, tot_span, tot_fixed))
    # the same sums over all stamped steps
    for st in range(args.steps):
        sp = [describe(L, '')[1:] for L in Ls[st * per:(st + 1) * per]]
        print('  step %d: spans %s | fixed %s' % (st, ' '.join('%.1f' % a for a, _ in sp), ' '.join('%.1f' % b for _, b in sp)))


if __name__ == '__main__':
    main()
