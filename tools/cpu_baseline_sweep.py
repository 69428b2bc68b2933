#!/usr/bin/env python
"""The CPU restatement (oracle, torch-CPU fp32, forward + backward of BASELINE config 2) timed at several thread counts
on this host: the sweep behind bench.py's choice of 16 threads for its cpu_baseline leg.  Writes JSON to stdout.
usage: tools/cpu_baseline_sweep.py [samples=3]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from demo2program_amd.config import make_config  # noqa: E402
from demo2program_amd.params import init_params  # noqa: E402
from demo2program_amd.synthetic import make_batch, to_torch  # noqa: E402


def main():
    samples = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cfg = make_config('karel')
    ocfg = oracle.OracleConfig(
        batch_size=cfg.batch_size, k=cfg.k, max_demo_len=cfg.max_demo_len, max_program_len=cfg.max_program_len, h=cfg.h,
        w=cfg.w, depth=cfg.depth, dim_program_token=cfg.dim_program_token, action_space=cfg.action_space,
        per_dim=cfg.per_dim, num_lstm_cell_units=cfg.num_lstm_cell_units, dataset_type=cfg.dataset_type)
    tb = to_torch(make_batch(cfg, seed=123))
    tp = {n: torch.from_numpy(v) for n, v in init_params(cfg, 123).items()}
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    out = {'host_threads': os.cpu_count(), 'cpu_model': model, 'samples_per_setting': samples,
           'workload': 'oracle.loss_and_grads, Karel full model, B=32, k=10, fp32 (forward + backward, no optimizer)',
           'sweep': []}
    for nt in (4, 8, 16, 32, 64, 128):         # (256 threads: > 100 s per step on the round-3 box -- not swept)
        if nt > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt)
        oracle.loss_and_grads(tp, tb, ocfg, dtype=torch.float32)
        ts = []
        for _ in range(samples):
            t0 = time.time()
            oracle.loss_and_grads(tp, tb, ocfg, dtype=torch.float32)
            ts.append(time.time() - t0)
        best = min(ts)
        out['sweep'].append({'threads': nt, 's_per_step': [round(t, 3) for t in ts],
                             'instances_per_s_best': round(cfg.batch_size / best, 2)})
        print('threads %3d: %s s/step' % (nt, ['%.2f' % t for t in ts]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
