#!/usr/bin/env python
"""Timeline view of a rocprofv3 kernel trace (rocpd sqlite): over the last N training steps
(delimited by adam_clip_kernel launches) report the wall span, the time at least one kernel was
running, the idle remainder, and the time two or more kernels overlapped; then the largest idle
gaps with the kernels on either side.  usage: tools/rocpd_gaps.py <results.db> [steps=5]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if 'adam_clip_kernel' in r[0]]
    if len(adam) < nsteps + 1:
        raise SystemExit('not enough steps in the trace')
    lo, hi = adam[-nsteps - 1] + 1, adam[-1] + 1
    ks = rows[lo:hi]
    t0, t1 = ks[0][1], max(k[2] for k in ks)
    ev = []
    for _, s, e in ks:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy = over = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    span = t1 - t0
    print('%d steps: span %.3f ms/step, busy %.3f, idle %.3f, >=2 kernels %.3f, kernel time %.3f, launches %.0f' % (
        nsteps, span / 1e6 / nsteps, busy / 1e6 / nsteps, (span - busy) / 1e6 / nsteps, over / 1e6 / nsteps,
        sum(e - s for _, s, e in ks) / 1e6 / nsteps, len(ks) / nsteps))
    gaps = []
    end = ks[0][2]
    prev = ks[0][0]
    for name, s, e in ks[1:]:
        if s > end:
            gaps.append((s - end, prev, name))
        if e > end:
            end, prev = e, name
    gaps.sort(reverse=True)
    hist = {}
    for g, a, b in gaps:
        key = (a.split('(')[0][-40:], b.split('(')[0][-40:])
        h = hist.setdefault(key, [0, 0])
        h[0] += 1
        h[1] += g
    print('idle by (kernel before -> kernel after), us/step:')
    for key, (n, g) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:14]:
        print('  %7.1f  x%-5.1f %s -> %s' % (g / 1e3 / nsteps, n / nsteps, key[0], key[1]))




def head(db, n=45):
    """first n kernels of the last step: start offset, duration, queue, name"""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    q = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else '0')
    rows = con.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    adam = [i for i, r in enumerate(rows) if 'adam_clip_kernel' in r[0]]
    lo = adam[-2] - 3
    t0 = rows[lo][1]
    for name, s, e, qq in rows[lo:lo + n]:
        print('%9.1f %7.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, qq, name.split('(')[0][-60:]))


if __name__ == '__main__':
    if len(sys.argv) > 3 and sys.argv[3] == 'head':
        head(sys.argv[1])
    else:
        main()
