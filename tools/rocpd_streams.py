#!/usr/bin/env python
"""Two-stream view of a rocprofv3 kernel trace (rocpd sqlite) of the default training schedule: for the
last N steps (delimited by adam_clip_kernel) the wall span per step, per queue the busy time and its
largest kernels, and the time the main queue ran alone / both queues ran.
usage: tools/rocpd_streams.py <results.db> [steps=5] [--seq]     (--seq: also the last step's launches in
start order: queue, start offset us, duration us, gap to the previous launch of the same queue, name)"""
import sqlite3
import sys


def main():
    seq = '--seq' in sys.argv
    argv = [a for a in sys.argv if a != '--seq']
    db = argv[1]
    nsteps = int(argv[2]) if len(argv) > 2 else 5
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    q = 'queue_id' if 'queue_id' in cols else 'stream_id'
    rows = con.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    adam = [i for i, r in enumerate(rows) if 'adam_clip_kernel' in r[0]]
    lo, hi = adam[-nsteps - 1] + 1, adam[-1] + 1
    ks = rows[lo:hi]
    t0, t1 = ks[0][1], max(k[2] for k in ks)
    print('%d steps: span %.3f ms/step, %d launches/step' % (nsteps, (t1 - t0) / 1e6 / nsteps, len(ks) / nsteps))
    queues = {}
    for name, s, e, qq in ks:
        queues.setdefault(qq, []).append((name, s, e))
    for qq, lst in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for _, s, e in lst)
        fam = {}
        for name, s, e in lst:
            key = name.split('(')[0].replace('void ', '')[:48]
            f = fam.setdefault(key, [0, 0])
            f[0] += 1
            f[1] += e - s
        top = sorted(fam.items(), key=lambda kv: -kv[1][1])[:6]
        print('queue %s: %.3f ms/step busy over %.1f launches/step; top: %s' % (
            qq, busy / 1e6 / nsteps, len(lst) / nsteps,
            ', '.join('%s %.2f' % (k, v[1] / 1e6 / nsteps) for k, v in top)))
    # overlap accounting
    ev = []
    for name, s, e, qq in ks:
        ev.append((s, 1, qq))
        ev.append((e, -1, qq))
    ev.sort()
    depth = {}
    last = t0
    alone = {}
    both = idle = 0
    for t, d, qq in ev:
        active = [k for k, v in depth.items() if v > 0]
        dt = t - last
        if len(active) == 0:
            idle += dt
        elif len(active) == 1:
            alone[active[0]] = alone.get(active[0], 0) + dt
        else:
            both += dt
        depth[qq] = depth.get(qq, 0) + d
        last = t
    print('idle %.3f ms/step, two queues active %.3f, one queue alone: %s' % (
        idle / 1e6 / nsteps, both / 1e6 / nsteps,
        ', '.join('q%s %.3f' % (k, v / 1e6 / nsteps) for k, v in alone.items())))
    if seq:
        one = rows[adam[-2] + 1:adam[-1] + 1]
        base = one[0][1]
        qids = {}
        last_end = {}
        for name, st, en, qq in one:
            qi = qids.setdefault(qq, len(qids))
            gap = (st - last_end[qq]) / 1e3 if qq in last_end else 0.0
            last_end[qq] = en
            print('q%d %9.1f %8.1f %7.1f  %s' % (qi, (st - base) / 1e3, (en - st) / 1e3, gap,
                                                name.split('(')[0].replace('void ', '')[:70]))


if __name__ == '__main__':
    main()
