#!/usr/bin/env python
"""Same-process A/B of HIP stream priorities on the headline step (round 2 measured "main at -1, side at 0" as noise at a
4.4 ms step; round 6's stamps show the encoders' backward recurrences ENTERING 17-27 us late behind side-queue workgroups):

    default      main = the default stream, side = a fresh priority-0 stream
    main-high    the whole step on a priority -1 stream, side at 0
    side-low     main default, side = hipStreamCreateWithPriority(..., +1) wrapped as an ExternalStream
    both         main -1, side +1
    main-fresh   the whole step on a fresh priority-0 stream (is it the priority, or leaving the null stream?)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from demo2program_amd import build
    from demo2program_amd.config import make_config
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    build.build_library()
    preset = sys.argv[1] if len(sys.argv) > 1 else 'karel'
    config = make_config(preset)
    trainer = Trainer(config, make_train_dir=False)
    batches = [make_batch(config, seed=123 + i) for i in range(4)]
    for b in batches:
        b['s_h'] = b['s_h'].astype(np.uint8)
    feeds = [trainer.model.get_feed_dict(b) for b in batches]
    m = trainer.model
    hip = ctypes.CDLL('libamdhip64.so')
    lo, hi = ctypes.c_int(0), ctypes.c_int(0)
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    print('hipDeviceGetStreamPriorityRange: least %d, greatest %d' % (lo.value, hi.value))

    def low_stream():
        h = ctypes.c_void_p()
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, lo.value)       # hipStreamNonBlocking
        assert rc == 0, rc
        return torch.cuda.ExternalStream(h.value)

    def block(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            trainer.train_step(feeds[i % 4])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    high = torch.cuda.Stream(priority=-1)
    fresh = torch.cuda.Stream(priority=0)
    lowst = low_stream()
    side0 = None
    res = {}
    for rnd in range(4):
        for name in ('default', 'main-fresh', 'main-high', 'side-low', 'both'):
            main_st = high if name in ('main-high', 'both') else (fresh if name == 'main-fresh' else torch.cuda.default_stream())
            with torch.cuda.stream(main_st):
                if name in ('side-low', 'both'):
                    m._side = lowst
                else:
                    if side0 is None:
                        m._side = None
                        block(3)
                        side0 = m._side
                    m._side = side0
                block(20)
                res.setdefault(name, []).append(block(200 if preset == 'karel' else 60))
            torch.cuda.synchronize()
            print('round %d %-10s %.4f ms/step' % (rnd, name, res[name][-1]), flush=True)
    for name, v in res.items():
        print('%-10s mean %.4f  min %.4f' % (name, sum(v) / len(v), min(v)))


if __name__ == '__main__':
    main()
