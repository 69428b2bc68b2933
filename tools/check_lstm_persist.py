#!/usr/bin/env python
"""Persistent LSTM sequence kernels (lstm_persist.hip) against the one-launch-per-step kernels
(lstm_step.hip) on the GPU box: forward outputs must be BIT-IDENTICAL (same K split, same
summation order), backward within fp32 round-off (two accumulators per wave instead of one);
then microseconds per time step of both back ends (graph replay, so no host launch cost).

A stale hand-off (the failure mode of an in-launch producer/consumer protocol) shows up here as a
forward mismatch; every case is repeated to catch rare ones."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402


class Seq(object):
    def __init__(self, M, U, T, masked, with_init, seed):
        g = torch.Generator().manual_seed(seed)
        self.M, self.U, self.T = M, U, T
        self.z0 = ((torch.rand(T * M, 4 * U, generator=g) - 0.5) * 2).cuda()
        self.z = self.z0.clone()
        self.Wh = ((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.2).cuda()
        self.h0 = ((torch.rand(M, U, generator=g) - 0.5)).cuda() if with_init else None
        self.c0 = ((torch.rand(M, U, generator=g) - 0.5)).cuda() if with_init else None
        if masked:
            lens = torch.randint(0, T + 1, (M,), generator=g)
            lens[0] = T
            self.lens = lens.to(torch.int32).cuda()
        else:
            self.lens = None
        self.hout, self.cs = torch.empty(T, M, U, device='cuda'), torch.empty(T, M, U, device='cuda')
        self.hf, self.cf = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
        self.dz = torch.empty_like(self.z)
        self.dhout = (torch.rand(T, M, U, generator=g) - 0.5).cuda()
        self.dhf, self.dcf = (torch.rand(M, U, generator=g) - 0.5).cuda(), (torch.rand(M, U, generator=g) - 0.5).cuda()
        self.dh0, self.dc0 = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')

    def fwd(self, fresh=True):
        M, U, T = self.M, self.U, self.T
        if fresh:
            self.z.copy_(self.z0)
        K.lstm_seq_fwd(self.z, 4 * U, M * 4 * U, M, U, T, self.Wh, self.h0, self.c0, self.lens,
                       self.hout, self.cs, self.hf, self.cf)

    def bwd(self):
        M, U, T = self.M, self.U, self.T
        K.lstm_seq_bwd(self.z, 4 * U, M * 4 * U, M, U, T, self.Wh, self.c0, self.lens, self.cs,
                       self.dhout, self.dhf, self.dcf, self.dz, self.dh0, self.dc0)

    def fwd_outputs(self):
        return [t.clone() for t in (self.z, self.hout, self.cs, self.hf, self.cf)]

    def bwd_outputs(self):
        return [t.clone() for t in (self.dz, self.dh0, self.dc0)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def check(M, U, T, masked, with_init, reps=5):
    s = Seq(M, U, T, masked, with_init, seed=M * 7 + U + T)
    K.set_lstm_persistent(False)
    s.fwd()
    ref_f = s.fwd_outputs()
    s.bwd()
    ref_b = s.bwd_outputs()
    torch.cuda.synchronize()
    K.set_lstm_persistent(True)
    bad = 0
    worst_b = 0.0
    for rep in range(reps):
        for t in (s.hout, s.cs, s.hf, s.cf, s.dz, s.dh0, s.dc0):
            t.fill_(float('nan'))
        s.fwd()
        got_f = s.fwd_outputs()
        s.bwd()
        got_b = s.bwd_outputs()
        torch.cuda.synchronize()
        err = K.lstm_persist_error()
        if err:
            print('   !! persistent hand-off error word 0x%08x' % (err & 0xffffffff))
            bad += 1
        for name, a, b in zip(('z', 'hout', 'cs', 'h_final', 'c_final'), got_f, ref_f):
            if not torch.equal(a, b):
                nd = (a != b).sum().item()
                print('   !! forward %s differs in %d elements (max |d| %.3e) rep %d'
                      % (name, nd, (a - b).abs().nan_to_num(nan=1e30).max().item(), rep))
                bad += 1
        for name, a, b in zip(('dz', 'dh0', 'dc0'), got_b, ref_b):
            d = (a - b).abs().nan_to_num(nan=1e30).max().item()
            scale = b.abs().max().item() + 1e-30
            worst_b = max(worst_b, d / scale)
            if d > 2e-5 * scale + 1e-7:
                print('   !! backward %s: max |d| %.3e (scale %.3e) rep %d' % (name, d, scale, rep))
                bad += 1
    print('M=%4d U=%3d T=%3d masked=%d init=%d : %s (backward rel diff %.1e)'
          % (M, U, T, masked, with_init, 'OK' if not bad else 'FAILED', worst_b))
    return bad


def bench(M, U, T, masked, reps=20):
    s = Seq(M, U, T, masked, True, seed=1)
    out = {}
    for name, on in (('per-step', False), ('persistent', True)):
        K.set_lstm_persistent(on)
        s.fwd()
        f = timed(graphed(lambda: s.fwd(fresh=False)), reps) / T
        b = timed(graphed(s.bwd), reps) / T
        out[name] = (f, b)
        gf = 2.0 * M * 4 * U * U / 1e3       # MFLOP per step -> TFLOP/s = gf / us / 1e3
        print('   M=%d U=%d T=%d masked=%d %-10s fwd %6.2f us/step (%5.1f TF/s)   bwd %6.2f us/step (%5.1f TF/s)'
              % (M, U, T, masked, name, f, gf / f / 1e3, b, gf / b / 1e3))
    K.set_lstm_persistent(True)
    return out


if __name__ == '__main__':
    build.build_library()
    quick = '--quick' in sys.argv
    bad = 0
    cases = [(12, 64, 6, 1, 1), (12, 64, 6, 0, 0), (35, 128, 4, 1, 1), (80, 256, 5, 0, 1), (48, 512, 4, 1, 0),
             (320, 512, 20, 1, 1), (320, 512, 20, 0, 0), (32, 512, 40, 0, 1), (400, 512, 12, 1, 1),
             (320, 64, 30, 1, 1), (512, 512, 8, 0, 1)]
    for c in cases:
        bad += check(*c, reps=2 if quick else 5)
    if not quick:
        bad += check(320, 512, 200, 1, 1, reps=10)     # long sequence: many hand-offs
    print('=== timing (graph replay)')
    from demo2program_amd.lib import call
    bench(320, 512, 20, False)
    bench(320, 512, 20, True)
    bench(32, 512, 48, False)
    bench(400, 512, 20, True)
    err = K.lstm_persist_error()
    print('error word: 0x%x' % err)
    print('RESULT: %s' % ('PASS' if not bad and not err else 'FAIL'))
    sys.exit(1 if bad or err else 0)
