#!/usr/bin/env python
"""Learning sanity check on the GPU box: trains the full model on batches of REAL Karel programs
with executed demonstrations (karel_env/generator.py) and prints the loss and the program
metrics of a held-out batch as training proceeds.  Not a benchmark.
usage: tools/train_synthetic_karel.py [steps] [n_train_batches] [batch_size] [k]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build  # noqa: E402
from demo2program_amd.config import make_config  # noqa: E402
from demo2program_amd.karel_env.generator import sample_batch  # noqa: E402
from demo2program_amd.trainer import Trainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    build.build_library()
    config = make_config('karel', batch_size=bs, k=k)
    if os.environ.get('D2P_SCHEDULED_SAMPLING', '0') == '1':
        config.scheduled_sampling, config.scheduled_sampling_decay_steps = True, max(steps, 1)
    t0 = time.time()
    train = [sample_batch(config, seed=1000 + i) for i in range(n_batches)]
    held = sample_batch(config, seed=7)
    print('generated %d train batches + 1 held-out in %.1f s (%d programs)' %
          (n_batches, time.time() - t0, n_batches * bs), flush=True)
    trainer = Trainer(config, make_train_dir=False)
    m = trainer.model
    feeds = [m.get_feed_dict(b) for b in train]
    held_feed = m.get_feed_dict(held)

    def evaluate(tag):
        track = m.track_moving
        m.track_moving = False
        loss = float(m.forward(held_feed).item())
        m.track_moving = track
        _, acc = m.report(with_greedy=True)
        h = m.report_hist
        print('%s held-out loss %.4f | tok %.3f seq %.3f syntax %.3f | greedy: tok %.3f syntax %.3f exact %.3f '
              'exec(all seen demos) %.3f exec(all test demos) %.3f' %
              (tag, loss, acc['program_token_acc'], acc['program_seq_acc'], acc['program_syntax_acc'],
               acc['greedy_program_token_acc'], acc['greedy_program_syntax_acc'],
               acc['greedy_exact_program_accuracy'], h['greedy_program_execution_acc_hist'][-1],
               h['test_greedy_program_execution_acc_hist'][-1]), flush=True)

    evaluate('step %4d' % 0)
    torch.cuda.synchronize()
    t0 = time.time()
    for s in range(steps):
        loss = trainer.train_step(feeds[s % len(feeds)])
        if (s + 1) % 100 == 0:
            torch.cuda.synchronize()
            print('step %4d train loss %.4f (%.1f ms/step)' % (s + 1, float(loss.item()), (time.time() - t0) * 1e3 / (s + 1)),
                  flush=True)
            evaluate('step %4d' % (s + 1))


if __name__ == '__main__':
    main()
