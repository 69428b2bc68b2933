#!/usr/bin/env python
"""The training step's MAIN-queue timeline without a profiler: Model.mark() records HIP timing events on the main stream
at the boundaries of its dependent chain (encoder, x-projection, each recurrence, the relation networks, the joins); the
interval between two consecutive marks is the device time from the completion of everything before the first to the
completion of everything before the second.  rocprofv3's kernel trace slows the host enough to put host-made gaps into
its timeline (the step is enqueued ~100 launches at 8-15 us each); this costs ~20 event records per step.

    python tools/step_marks.py [--preset karel] [--steps 50] [--side 1|0]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--preset', default='karel')
    ap.add_argument('--steps', type=int, default=50)
    args = ap.parse_args()
    from demo2program_amd import build
    from demo2program_amd.config import make_config
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    build.build_library()
    config = make_config(args.preset)
    trainer = Trainer(config, make_train_dir=False)
    batches = [make_batch(config, seed=123 + i) for i in range(4)]
    for b in batches:
        b['s_h'] = b['s_h'].astype(np.uint8)
    feeds = [trainer.model.get_feed_dict(b) for b in batches]
    m = trainer.model
    for i in range(20):
        trainer.train_step(feeds[i % 4])
    torch.cuda.synchronize()
    # un-marked reference
    import time
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.train_step(feeds[i % 4])
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / args.steps * 1e3
    rows = {}
    order = []
    t0 = time.perf_counter()
    per_step = []
    for i in range(args.steps):
        m._marks = []
        m.mark('step:start')
        trainer.train_step(feeds[i % 4])
        m.mark('step:adam')
        per_step.append(m._marks)
        m._marks = None
    torch.cuda.synchronize()
    marked = (time.perf_counter() - t0) / args.steps * 1e3
    for marks in per_step:
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            key = '%s -> %s' % (n0, n1)
            if key not in rows:
                rows[key] = []
                order.append(key)
            rows[key].append(e0.elapsed_time(e1) * 1e3)
    print('%s: %.4f ms/step unmarked, %.4f ms/step with marks' % (args.preset, plain, marked))
    tot = 0.0
    for key in order:
        v = np.asarray(rows[key])
        tot += float(np.median(v))
        print('  %-40s median %8.1f us   mean %8.1f   (p10 %7.1f, p90 %7.1f)' % (key, np.median(v), v.mean(),
                                                                                np.percentile(v, 10), np.percentile(v, 90)))
    print('  sum of medians %.1f us' % tot)


if __name__ == '__main__':
    main()
