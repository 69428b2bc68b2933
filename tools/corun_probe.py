#!/usr/bin/env python
"""When do workgroups of a second queue become resident beside a persistent recurrent launch?  A probe launch (every
workgroup records the wall clock of its first instruction) is issued on the side stream right after a persistent launch
was issued on the main stream; two more probes bracket the persistent launch on the main stream.  Reported: when the
side probe's workgroups started, as a fraction of the persistent launch's duration (0 = with it, 1 = after it), by
workgroup size, register footprint and LDS of the probe."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import call  # noqa: E402
from demo2program_amd.models.model_full import pick_concurrent_stream  # noqa: E402
from check_lstm_wide import mk  # noqa: E402


def main():
    build.build_library()
    K.set_lstm_persistent(True)
    src = torch.rand(1024, device='cuda')
    sink = torch.zeros(1024, device='cuda')
    main_s = torch.cuda.current_stream()
    side = pick_concurrent_stream()
    act, per, prog = mk(320, 20, seed=13), mk(320, 20, seed=14), mk(32, 50, seed=15)
    # backward sequences
    def bw(q):
        M, T, U = q['M'], q['n_steps'], q['U']
        return dict(M=M, U=U, n_steps=T, z=q['z'], Wh=q['Wh'], c0=q.get('c0'), cs=q['cs'],
                    dhout=torch.rand(T, M, U, device='cuda'), dz=torch.zeros(T * M, 4 * U, device='cuda'),
                    dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'))
    for q in (act, per, prog):
        K.lstm_seq_fwd_multi([q])
    bact, bper, bprog = bw(act), bw(per), bw(prog)
    launches = [('wide fwd, one sequence (59 KB LDS)', lambda: K.lstm_seq_fwd_multi([act])),
                ('wide fwd, three sequences (81 KB LDS)', lambda: K.lstm_seq_fwd_multi([act, per, prog])),
                ('bwd, one sequence', lambda: K.lstm_seq_bwd_multi([bact])),
                ('bwd, three sequences', lambda: K.lstm_seq_bwd_multi([bact, bper, bprog]))]
    call.d2p_lstm_persist_set_fwd_wide(0, 0, 0, -1)
    launches.insert(0, ('narrow fwd, one sequence (208 registers)', lambda: K.lstm_seq_fwd_multi([act])))

    bufs = {}

    def probe(stream, blocks, threads, regs, lds, key):
        out = bufs.get(key)
        if out is None:
            out = bufs[key] = torch.zeros(blocks, dtype=torch.int64, device='cuda')
            torch.cuda.synchronize()             # (the fill runs on the current stream: not behind the launch under test)
        with torch.cuda.stream(stream):
            call.d2p_probe_clock(blocks, threads, regs, lds, out.data_ptr(), src.data_ptr(), sink.data_ptr(),
                                 stream.cuda_stream)
        return out

    for name, fn in launches:
        if not name.startswith('narrow'):
            call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, -1)
        fn(); fn()
        torch.cuda.synchronize()
        print('== beside %s' % name)
        for (threads, regs, lds) in ((64, 4, 0), (128, 4, 0), (256, 4, 0), (512, 4, 0), (256, 48, 0), (256, 80, 0),
                                    (256, 120, 0), (256, 200, 0), (128, 80, 0), (64, 80, 0), (256, 80, 32768), (256, 48, 32768),
                                    (256, 48, 65536)):
            fr = []
            for rep in range(3):
                torch.cuda.synchronize()
                a = probe(main_s, 8, 64, 4, 0, 'a')
                ev = torch.cuda.Event()
                ev.record(main_s)
                fn()
                side.wait_event(ev)
                b = probe(side, 512, threads, regs, lds, 'b')
                c = probe(main_s, 8, 64, 4, 0, 'c')
                torch.cuda.synchronize()
                t0, t1 = float(a.min()), float(c.min())
                tb = b.cpu().numpy().astype(np.float64)
                fr.append(((np.percentile(tb, 10) - t0) / (t1 - t0), (np.median(tb) - t0) / (t1 - t0),
                           (t1 - t0) / 100.0))
            f10, f50, us = np.median([f[0] for f in fr]), np.median([f[1] for f in fr]), np.median([f[2] for f in fr])
            print('   probe %3d threads, ~%3d registers, %5d B LDS: 10 %% of its workgroups started at %.2f, half at %.2f '
                  'of the launch (%.0f us)' % (threads, regs, lds, f10, f50, us))
    print('error word: 0x%x' % K.lstm_persist_error())


if __name__ == '__main__':
    main()
