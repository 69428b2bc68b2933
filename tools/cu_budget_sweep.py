#!/usr/bin/env python
"""The training step against the number of CUs the persistent recurrences are planned for
(d2p_lstm_persist_set_cu_budget): a four-wave workgroup never becomes resident on a CU that holds a recurrence's
workgroup (tools/corun_probe.py), so beside a launch that fills all 256 CUs the side queue only waits; a smaller budget
leaves whole CUs to it, at the price of more 16-row phases per row domain.

    python tools/cu_budget_sweep.py [--preset karel] [--budgets 0 224 192 160] [--rounds 3] [--steps 100]

One process, one trainer, the same batches; blocks alternate over the budgets."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--preset', default='karel')
    ap.add_argument('--budgets', type=int, nargs='*', default=[0, 224, 192, 160])
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--steps', type=int, default=100)
    args = ap.parse_args()
    from demo2program_amd import build, kernels as K
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

    def block(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            trainer.train_step(feeds[i % len(feeds)])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    block(30)
    res = {b: [] for b in args.budgets}
    for r in range(args.rounds):
        for b in args.budgets:
            K.lstm_set_cu_budget(b)
            block(10)
            res[b].append(block(args.steps))
            fails = trainer.settle()
            print('round %d budget %3d: %.4f ms/step (fallbacks so far %d, err 0x%x)' %
                  (r, b, res[b][-1], fails, K.lstm_persist_error() & 0xffffffff), flush=True)
    K.lstm_set_cu_budget(0)
    for b in args.budgets:
        print('budget %3d: mean %.4f ms/step  min %.4f  (%s)' % (b, sum(res[b]) / len(res[b]), min(res[b]),
                                                               ' '.join('%.4f' % v for v in res[b])))


if __name__ == '__main__':
    main()
