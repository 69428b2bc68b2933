#!/usr/bin/env python
"""Wide-tile persistent forward kernel (lstm_persist.hip: 16 units per column tile, up to three sequences per launch,
length-sorted domains, XCD-local hand-offs) against the one-launch-per-step kernels (lstm_step.hip) on the GPU box:
every output must be BIT-IDENTICAL (same K split, same summation order) in every mode -- a stale hand-off shows up as a
mismatch -- then microseconds per launch of the wide and the narrow kernel for the shapes of the training step.

  python tools/check_lstm_wide.py [--quick] [--time-only]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import build, kernels as K  # noqa: E402
from demo2program_amd.lib import call, load  # noqa: E402


def mk(M, T, U=512, masked=False, init=True, seed=0, lo=0):
    g = torch.Generator().manual_seed(seed * 1000 + M + T)
    q = dict(M=M, U=U, n_steps=T,
             z0=((torch.rand(T * M, 4 * U, generator=g) - 0.5) * 2).cuda(),
             Wh=((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.2).cuda(),
             hout=torch.empty(T, M, U, device='cuda'), cs=torch.empty(T, M, U, device='cuda'),
             h_final=torch.empty(M, U, device='cuda'), c_final=torch.empty(M, U, device='cuda'))
    q['z'] = q['z0'].clone()
    if init:
        q['h0'] = (torch.rand(M, U, generator=g) - 0.5).cuda()
        q['c0'] = (torch.rand(M, U, generator=g) - 0.5).cuda()
    if masked:
        lens = torch.randint(lo, T + 1, (M,), generator=g)
        lens[0] = T
        q['lens_host'] = lens.numpy().astype(np.int64)
        q['lens'] = lens.to(torch.int32).cuda()
    return q


def outputs(q):
    return [q[n].clone() for n in ('z', 'hout', 'cs', 'h_final', 'c_final')]


def poison(q):
    q['z'].copy_(q['z0'])
    for n in ('hout', 'cs', 'h_final', 'c_final'):
        q[n].fill_(float('nan'))


def run(seqs, sort=False):
    for q in seqs:
        poison(q)
        q.pop('row_order', None)
        if sort and q.get('lens') is not None:
            q['row_order'] = K.lstm_row_order(q['lens_host'])
    K.lstm_seq_fwd_multi(seqs)


def check(name, seqs, sort=False, reps=3, may_fall_back=False):
    K.set_lstm_persistent(False)
    refs = []
    for q in seqs:
        run([q])
        refs.append(outputs(q))
    torch.cuda.synchronize()
    K.set_lstm_persistent(True)
    bad = 0
    n0 = [load().d2p_lstm_persist_wide_launches(i) for i in range(4)]
    for rep in range(reps):
        run(seqs, sort)
        torch.cuda.synchronize()
        err = K.lstm_persist_error()
        if err:
            print('   !! hand-off error word 0x%08x' % (err & 0xffffffff))
            K.lstm_persist_error(True)
            bad += 1
        for i, q in enumerate(seqs):
            for nm, a, b in zip(('z', 'hout', 'cs', 'h_final', 'c_final'), outputs(q), refs[i]):
                if not torch.equal(a, b):
                    d = (a - b).abs().nan_to_num(nan=1e30)
                    print('   !! seq %d %s differs in %d elements (max |d| %.3e) rep %d'
                          % (i, nm, (a != b).sum().item(), d.max().item(), rep))
                    bad += 1
    n1 = [load().d2p_lstm_persist_wide_launches(i) for i in range(4)]
    took = n1[len(seqs)] - n0[len(seqs)] == reps and (not sort or n1[0] - n0[0] == reps)
    if not took and not may_fall_back:
        print('   !! the wide kernel did not take these launches', n0, n1)
        bad += 1
    print('%-58s %s' % (name, 'OK' if not bad else 'FAILED'))
    return bad


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def bench_lens(M, T, seed=5):
    """demonstration lengths as bench.py's synthetic batches draw them: uniform in 8..T"""
    rs = np.random.RandomState(seed)
    return rs.randint(8, T + 1, size=M).astype(np.int64)


def timing():
    K.set_lstm_persistent(True)
    enc = mk(320, 20, masked=True, init=False, seed=11)
    lens = bench_lens(320, 20)
    enc['lens_host'] = lens
    enc['lens'] = torch.from_numpy(lens.astype(np.int32)).cuda()
    enc2 = mk(320, 20, masked=True, init=True, seed=12)
    enc2['lens_host'], enc2['lens'] = enc['lens_host'], enc['lens']
    act, per, prog = mk(320, 20, seed=13), mk(320, 20, seed=14), mk(32, 50, seed=15)
    order = K.lstm_row_order(lens)
    res = {}

    def go(seqs, sort):
        for q in seqs:
            q.pop('row_order', None)
            if sort and q.get('lens') is not None:
                q['row_order'] = order
        return timed(lambda: K.lstm_seq_fwd_multi(seqs))
    for wide in (0, 1):
        call.d2p_lstm_persist_set_fwd_wide(wide, 0, 0, -1)
        tag = 'wide' if wide else 'narrow'
        res[tag + ' enc1 (no h0)'] = go([enc], False)
        res[tag + ' enc2 (h0)'] = go([enc2], False)
        if wide:
            res[tag + ' enc1 sorted'] = go([enc], True)
            res[tag + ' enc2 sorted'] = go([enc2], True)
            call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, 0)
            res[tag + ' enc2 sorted, write-through only'] = go([enc2], True)
            res[tag + ' triple, write-through only'] = go([act, per, prog], False)
            call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, 1)
        res[tag + ' act alone'] = go([act], False)
        res[tag + ' prog alone'] = go([prog], False)
        res[tag + ' act + prog'] = go([act, prog], False)
        if wide:
            res[tag + ' act + per + prog'] = go([act, per, prog], False)
    call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, -1)
    for k, v in res.items():
        print('   %-42s %7.1f us' % (k, v))
    print('   error word: 0x%x' % K.lstm_persist_error())
    return res


def sweep():
    """us per launch over the hand-off knobs: la_from x defer_from x request position (quarters of the chain)"""
    K.set_lstm_persistent(True)
    enc = mk(320, 20, masked=True, init=True, seed=12)
    lens = bench_lens(320, 20)
    enc['lens_host'], enc['lens'] = lens, torch.from_numpy(lens.astype(np.int32)).cuda()
    act, per, prog = mk(320, 20, seed=13), mk(320, 20, seed=14), mk(32, 50, seed=15)
    order = K.lstm_row_order(lens)
    print('la_from defer_from quarters | enc unsorted | enc sorted | triple | act alone')
    for la in (2, 3, 4):
        for df in (4, 5, 6, 8):
            for q in (2, 3):
                call.d2p_lstm_persist_set_fwd_wide(1, la, df, 1 | (2 if q == 2 else 4))
                enc.pop('row_order', None)
                t0 = timed(lambda: K.lstm_seq_fwd_multi([enc]))
                enc['row_order'] = order
                t1 = timed(lambda: K.lstm_seq_fwd_multi([enc]))
                t2 = timed(lambda: K.lstm_seq_fwd_multi([act, per, prog]))
                t3 = timed(lambda: K.lstm_seq_fwd_multi([act]))
                print('   %d %d %d | %6.1f | %6.1f | %6.1f | %6.1f' % (la, df, q, t0, t1, t2, t3))
    call.d2p_lstm_persist_set_fwd_wide(1, 3, 5, 1 | 4)
    print('   error word: 0x%x' % K.lstm_persist_error())


if __name__ == '__main__':
    build.build_library()
    if '--sweep' in sys.argv:
        sweep()
        sys.exit(0)
    quick = '--quick' in sys.argv
    bad = 0
    if '--time-only' not in sys.argv:
        reps = 2 if quick else 4
        for xl in (1, 0):
            call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, xl)
            tag = 'L2-local hand-offs' if xl else 'write-through'
            # one sequence: ragged row counts, every width, with / without lengths and initial state
            for (M, U, T, masked, init) in [(12, 64, 6, 1, 1), (35, 128, 4, 1, 1), (80, 256, 5, 0, 1), (48, 512, 4, 1, 0),
                                            (320, 512, 20, 1, 1), (320, 512, 20, 0, 0), (32, 512, 40, 0, 1),
                                            (400, 512, 12, 1, 1), (333, 512, 9, 1, 1), (512, 512, 8, 0, 1),
                                            (1000, 512, 5, 1, 1)]:
                bad += check('%s: M=%d U=%d T=%d masked=%d init=%d' % (tag, M, U, T, masked, init),
                             [mk(M, T, U, masked, init, seed=1)], reps=reps)
                if masked:
                    bad += check('%s: ... sorted by length' % tag, [mk(M, T, U, masked, init, seed=1)], sort=True, reps=reps)
            # lengths as the training batches have them (8..20), zero-length rows, all rows full
            q = mk(320, 20, masked=True, init=True, seed=2, lo=8)
            bad += check('%s: 320 x 20, lengths 8..20, sorted' % tag, [q], sort=True, reps=reps)
            q = mk(320, 20, masked=True, init=False, seed=3, lo=0)
            q['lens_host'][5:40] = 0
            q['lens'] = torch.from_numpy(q['lens_host'].astype(np.int32)).cuda()
            bad += check('%s: 320 x 20, zero-length rows, sorted' % tag, [q], sort=True, reps=reps)
            # two and three sequences per launch
            bad += check('%s: pair 320x20 + 32x50' % tag, [mk(320, 20, seed=4), mk(32, 50, seed=5)], reps=reps)
            bad += check('%s: triple 320x20 + 320x20 + 32x50' % tag,
                         [mk(320, 20, seed=4), mk(320, 20, seed=6), mk(32, 50, seed=5)], reps=reps)
            bad += check('%s: triple 400x20 + 400x20 + 16x32' % tag,
                         [mk(400, 20, seed=4), mk(400, 20, seed=6), mk(16, 32, seed=5)], reps=reps, may_fall_back=True)
            bad += check('%s: triple, one masked + sorted' % tag,
                         [mk(320, 20, masked=True, seed=7, lo=8), mk(100, 7, seed=6), mk(32, 50, seed=5)], sort=True,
                         reps=reps)
        # every hand-off form on every domain size
        for (la, df) in ((2, 3), (2, 4), (3, 5), (4, 8)):
            call.d2p_lstm_persist_set_fwd_wide(1, la, df, 1)
            bad += check('la_from %d, defer_from %d: 320 x 20 masked' % (la, df), [mk(320, 20, masked=True, seed=8)],
                         reps=reps)
            bad += check('la_from %d, defer_from %d: triple' % (la, df),
                         [mk(320, 20, seed=4), mk(320, 20, seed=6), mk(32, 50, seed=5)], reps=reps)
        call.d2p_lstm_persist_set_fwd_wide(1, 3, 5, 1)
        if not quick:
            bad += check('long sequence: 320 x 200 masked', [mk(320, 200, masked=True, seed=9)], reps=3)
            bad += check('long sequence: 320 x 200 masked, sorted', [mk(320, 200, masked=True, seed=9)], sort=True, reps=3)
    print('=== timing (eager launches, us per launch)')
    timing()
    err = K.lstm_persist_error()
    print('RESULT: %s' % ('PASS' if not bad and not err else 'FAIL'))
    sys.exit(1 if bad or err else 0)
