#!/usr/bin/env python
"""Same-process A/B of an ATTRIBUTE of the trainer's Model object on the training step (the A sides of the model-level
test A/Bs: small_stream, fused_rn, decoder_skip_past_len, fold_bn, ...):

    python tools/step_ab_attr.py small_stream 0 1 [--rounds 4] [--steps 200] [--preset karel] [--marks]

alternates blocks of timed steps with model.<attr> = bool(A) / bool(B); --marks prints the main-queue timeline
(tools/step_marks.py) of the last block of each setting."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('attr')
    ap.add_argument('a', type=int)
    ap.add_argument('b', type=int)
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--preset', default='karel')
    ap.add_argument('--marks', action='store_true')
    args = ap.parse_args()
    from demo2program_amd import build
    from demo2program_amd.config import make_config
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    build.build_library()
    config = make_config(args.preset)
    trainer = Trainer(config, make_train_dir=False)
    m = trainer.model
    assert hasattr(m, args.attr), args.attr
    batches = [make_batch(config, seed=123 + i) for i in range(4)]
    for b in batches:
        b['s_h'] = b['s_h'].astype(np.uint8)
    feeds = [m.get_feed_dict(b) for b in batches]

    def block(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            trainer.train_step(feeds[i % len(feeds)])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def marks(n):
        rows, order = {}, []
        for i in range(n):
            m._marks = []
            m.mark('step:start')
            trainer.train_step(feeds[i % len(feeds)])
            m.mark('step:adam')
            mk, m._marks = m._marks, None
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(mk[:-1], mk[1:]):
                key = '%s -> %s' % (n0, n1)
                if key not in rows:
                    rows[key] = []
                    order.append(key)
                rows[key].append(e0.elapsed_time(e1) * 1e3)
        for key in order:
            print('      %-40s median %8.1f us' % (key, float(np.median(rows[key]))))

    for v in (args.a, args.b):          # both settings warm (streams probed, buffers sized)
        setattr(m, args.attr, bool(v))
        block(20)
    res = {args.a: [], args.b: []}
    for r in range(args.rounds):
        for v in (args.a, args.b):
            setattr(m, args.attr, bool(v))
            block(10)
            res[v].append(block(args.steps))
            print('%s = %d: %.4f ms/step' % (args.attr, v, res[v][-1]), flush=True)
    for v in (args.a, args.b):
        print('mean %s = %d: %.4f ms/step over %d blocks' % (args.attr, v, sum(res[v]) / len(res[v]), len(res[v])))
        if args.marks:
            setattr(m, args.attr, bool(v))
            block(10)
            marks(30)
    print('persistent fallbacks: %d' % trainer.settle())


if __name__ == '__main__':
    main()
