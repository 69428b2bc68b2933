#!/usr/bin/env python
"""Can two persistent LSTM launches run side by side on two streams?  Launches pairs (A on the main stream,
B on a concurrent stream) with a varying head start for A and reports hand-off time-outs
(d2p_lstm_persist_error) and the pair's time against the two launches back to back."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import kernels as K  # noqa: E402
from demo2program_amd.models.model_full import pick_concurrent_stream  # noqa: E402


def make(M, U, T):
    g = torch.Generator().manual_seed(M)
    z = (torch.rand(T, M, 4 * U, generator=g) - 0.5).cuda()
    Wh = ((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.1).cuda()
    return dict(z=z, Wh=Wh, hout=torch.empty(T, M, U, device='cuda'), cs=torch.empty(T, M, U, device='cuda'), M=M, U=U, T=T)


def launch(q):
    K.lstm_seq_fwd(q['z'].clone(), 4 * q['U'], q['M'] * 4 * q['U'], q['M'], q['U'], q['T'], q['Wh'], None, None, None,
                   q['hout'], q['cs'], None, None)


def main():
    side = pick_concurrent_stream()
    mainst = torch.cuda.current_stream()
    for (Ma, Mb) in [(320, 32), (320, 320), (32, 32)]:
        A, B = make(Ma, 512, 20), make(Mb, 512, 50 if Mb == 32 else 20)
        launch(A); launch(B)
        with torch.cuda.stream(side):
            launch(B)
        torch.cuda.synchronize()
        K.lstm_persist_error(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(A); launch(B); e1.record(); torch.cuda.synchronize()
        serial = e0.elapsed_time(e1) * 1e3
        for delay in (0, 20000, 100000, 400000):     # spin cycles on the side stream before B starts
            errs, times = 0, []
            for rep in range(5):
                torch.cuda.synchronize()
                e0.record(mainst)
                side.wait_stream(mainst)
                launch(A)
                with torch.cuda.stream(side):
                    if delay:
                        torch.cuda._sleep(delay)
                    launch(B)
                mainst.wait_stream(side)
                e1.record(mainst)
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3)
                err = K.lstm_persist_error(True)
                errs += 1 if err else 0
            print('A: M=%d  B: M=%d  head start %7d cycles: pair %.0f us (back to back %.0f us), time-outs in %d of 5 runs' % (
                Ma, Mb, delay, min(times), serial, errs))


if __name__ == '__main__':
    main()
