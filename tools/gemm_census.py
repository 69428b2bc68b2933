#!/usr/bin/env python
"""Per-shape census of the dense GEMM launches of one training step (run on the GPU box):
every d2p_gemm_f32_* call of an eager forward+backward is bracketed with events (one stream,
synchronised), grouped by (kind, M, N, K, accumulate, bias/act) and printed by summed time."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['D2P_SIDE_STREAM'] = '0'
from demo2program_amd import kernels as K  # noqa: E402
from demo2program_amd.config import make_config  # noqa: E402
from demo2program_amd.synthetic import make_batch  # noqa: E402
from demo2program_amd.trainer import Trainer  # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else 'karel'
    cfg = make_config(preset)
    tr = Trainer(cfg, make_train_dir=False, use_graph=False)
    feed = tr.model.get_feed_dict(make_batch(cfg, seed=1))
    for _ in range(3):
        tr.train_step(feed)
    torch.cuda.synchronize()
    rec = collections.OrderedDict()
    origs = {n: getattr(K, n) for n in ('gemm_raw', 'gemm_rows', 'gemm_tn_rows', 'gemm_batched')}

    def bracket(key, fn, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*a, **kw)
        e1.record()
        rec.setdefault(key, []).append((e0, e1))

    def raw(kind, M, N, Kd, A, lda, B, ldb, C, ldc, bias=None, act=0, accumulate=False, allow_split=True):
        bracket((kind, M, N, Kd, bool(accumulate), bias is not None, act), origs['gemm_raw'], kind, M, N, Kd, A, lda, B, ldb,
                C, ldc, bias, act, accumulate, allow_split)

    def rows_(kind, n_rows, N, Kd, A, lda, B, ldb, C, ldc, rows, bias=None):
        bracket((kind + '/rows', n_rows, N, Kd, False, bias is not None, 0), origs['gemm_rows'], kind, n_rows, N, Kd, A, lda,
                B, ldb, C, ldc, rows, bias)

    def tn_rows(M, N, Kd, A, lda, rowsA, B, ldb, rowsB, C, ldc, accumulate=False):
        bracket(('tn/rows', M, N, Kd, bool(accumulate), False, 0), origs['gemm_tn_rows'], M, N, Kd, A, lda, rowsA, B, ldb,
                rowsB, C, ldc, accumulate)

    def batched(kind, nb1, nb0, M, N, Kd, A, lda, sA, B, ldb, sB, C, ldc, sC, bias=None, sbias=(0, 0), act=0, accumulate=False):
        # (flops of the whole batch: M stands for nb1 * nb0 * M)
        bracket((kind + '/x%d' % (nb1 * nb0), nb1 * nb0 * M, N, Kd, bool(accumulate), bias is not None, act), origs['gemm_batched'],
                kind, nb1, nb0, M, N, Kd, A, lda, sA, B, ldb, sB, C, ldc, sC, bias, sbias, act, accumulate)

    K.gemm_raw, K.gemm_rows, K.gemm_tn_rows, K.gemm_batched = raw, rows_, tn_rows, batched
    reps = 5
    for _ in range(reps):
        tr.model.forward(feed)
        tr.model.backward()
    torch.cuda.synchronize()
    for n, f in origs.items():
        setattr(K, n, f)
    rows = []
    for key, evs in rec.items():
        us = sum(a.elapsed_time(b) for a, b in evs) * 1e3 / reps
        n = len(evs) / reps
        kind, M, N, Kd = key[:4]
        fl = 2.0 * M * N * Kd * n
        rows.append((us, n, key, fl / (us * 1e-6) / 1e12 if us > 0 else 0.0))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print('%-8s %6s %6s %6s %5s %5s | %5s %9s %8s %6s' % ('kind', 'M', 'N', 'K', 'acc', 'epi', 'n', 'us/step', 'TFLOP/s', 'share'))
    for us, n, key, tf in rows:
        kind, M, N, Kd, acc, hasb, act = key
        print('%-8s %6d %6d %6d %5s %5s | %5.1f %9.1f %8.1f %5.1f%%' % (kind, M, N, Kd, 'y' if acc else '-',
              ('b' if hasb else '-') + str(act), n, us, tf, 100 * us / tot))
    fl = sum(2.0 * k[1] * k[2] * k[3] * len(v) / reps for k, v in rec.items())
    print('total %.1f us/step, %.1f GFLOP, %.1f TFLOP/s' % (tot, fl / 1e9, fl / tot / 1e6))


if __name__ == '__main__':
    main()
