#!/usr/bin/env python
"""Micro-benchmark / ablation of the fused recurrent-step kernels (run on the GPU box).
Times d2p_lstm_seq_fwd / _bwd with torch events over T steps: full kernel, MFMA part
skipped, epilogue skipped, both skipped (launch floor); single stream and two independent
LSTMs on two streams (co-residency test)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402


class Seq(object):
    def __init__(self, M, U, T, masked, seed):
        g = torch.Generator().manual_seed(seed)
        self.M, self.U, self.T = M, U, T
        self.z = (torch.rand(T * M, 4 * U, generator=g) - 0.5).cuda()
        self.Wh = ((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.1).cuda()
        self.h0, self.c0 = torch.zeros(M, U, device='cuda'), torch.zeros(M, U, device='cuda')
        self.lens = torch.full((M,), T, dtype=torch.int32).cuda() if masked else None
        self.hout, self.cs = torch.empty(T, M, U, device='cuda'), torch.empty(T, M, U, device='cuda')
        self.hf, self.cf = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
        self.dz = torch.empty_like(self.z)
        self.dhout = torch.randn(T, M, U, device='cuda')
        self.dh0, self.dc0 = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')

    def fwd(self):
        M, U, T = self.M, self.U, self.T
        K.lstm_seq_fwd(self.z, 4 * U, M * 4 * U, M, U, T, self.Wh, self.h0, self.c0, self.lens,
                       self.hout, self.cs, self.hf, self.cf)

    def bwd(self):
        M, U, T = self.M, self.U, self.T
        K.lstm_seq_bwd(self.z, 4 * U, M * 4 * U, M, U, T, self.Wh, self.c0, self.lens, self.cs,
                       self.dhout, self.hf, self.cf, self.dz, self.dh0, self.dc0)


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


def ablation(M, U, T, masked, reps=20):
    lib = load()
    s = Seq(M, U, T, masked, 0)
    print('M=%d U=%d T=%d masked=%s  (us per step)' % (M, U, T, masked))
    for name, flags in (('full', 0), ('no_mfma', 1), ('no_epilogue', 2), ('launch_only', 3)):
        lib.d2p_lstm_debug_flags(flags)
        print('   %-12s fwd %7.2f   bwd %7.2f' % (name, timed(s.fwd, reps) / T, timed(s.bwd, reps) / T))
    lib.d2p_lstm_debug_flags(0)


def graphed(fn):
    """Capture fn into a hipGraph (removes host launch cost) and return a replay callable."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def dual(M, U, T, reps=20):
    a, b = Seq(M, U, T, False, 1), Seq(M, U, T, False, 2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both(which):
        def run():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                getattr(a, which)()
            with torch.cuda.stream(s2):
                getattr(b, which)()
            cur.wait_stream(s1)
            cur.wait_stream(s2)
        return run

    def serial(which):
        def run():
            getattr(a, which)()
            getattr(b, which)()
        return run

    for which in ('fwd', 'bwd'):
        t_ser = timed(graphed(serial(which)), reps) / T
        t_par = timed(graphed(both(which)), reps) / T
        print('   two LSTMs %s (graph replay): serial %7.2f us/step-pair   2 streams %7.2f us/step-pair'
              % (which, t_ser, t_par))


if __name__ == '__main__':
    build.build_library()
    lib = load()
    for pipe in (0, 1):
        lib.d2p_lstm_set_tiling(256, 256, pipe)
        print('=== forward kernel: %s' % ('pipelined <=256 regs' if pipe else 'all loads up front'))
        ablation(320, 512, 20, False)
        dual(320, 512, 20)
    lib.d2p_lstm_set_tiling(256, 256, 0)
    ablation(32, 512, 50, False)
    t = Seq(32, 512, 50, False, 3)
    print('M=32 graph replay: fwd %.2f bwd %.2f us/step' % (timed(graphed(t.fwd), 20) / 50, timed(graphed(t.bwd), 20) / 50))
    t = Seq(320, 512, 20, True, 3)
    print('M=320 masked graph replay: fwd %.2f bwd %.2f us/step' % (timed(graphed(t.fwd), 20) / 20, timed(graphed(t.bwd), 20) / 20))
