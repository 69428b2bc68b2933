#!/usr/bin/env python
"""Same-process A/B of a process-global C-ABI switch on the headline training step:

    python tools/step_ab.py d2p_lstm_persist_set_bwd_pubdz 0 1 [--rounds 4] [--steps 200] [--preset karel]

builds the bench's trainer once, then alternates `setter(A)` / `setter(B)` blocks of timed steps (one box, one
process, same batches: the only thing that changes between the blocks is the switch) and prints ms per step of each
block and the means.  The setter is left at its LAST value (B)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('setter')
    ap.add_argument('a', type=int)
    ap.add_argument('b', type=int)
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--preset', default='karel')
    ap.add_argument('--extra', type=int, nargs='*', default=[], help='further constant arguments of the setter')
    args = ap.parse_args()
    from demo2program_amd import build
    from demo2program_amd.config import make_config
    from demo2program_amd.lib import load
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    build.build_library()
    lib = load()
    setter = getattr(lib, args.setter)
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

    block(40)
    res = {args.a: [], args.b: []}
    for r in range(args.rounds):
        for v in (args.a, args.b):
            setter(v, *args.extra)
            block(10)
            ms = block(args.steps)
            res[v].append(ms)
            print('%s(%d): %.4f ms/step' % (args.setter, v, ms), flush=True)
    for v in (args.a, args.b):
        print('mean %s(%d): %.4f ms/step over %d blocks' % (args.setter, v, sum(res[v]) / len(res[v]), len(res[v])))
    print('persistent fallbacks: %d' % trainer.settle())


if __name__ == '__main__':
    main()
