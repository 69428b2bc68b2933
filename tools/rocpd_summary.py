#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a markdown table:
per kernel name -> calls, total / average / min / max duration (us), share of GPU time.
usage: tools/rocpd_summary.py <results.db> [steps] > profiles/<name>.md"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for name, calls, tot, avg, mn, mx, vg, ag, lds in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        print('| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f | %s | %s | %s |' %
              (short, calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, lds))
    print()
    print('total kernel time: %.3f ms over %d launches' % (total / 1e6, sum(r[1] for r in rows)))
    # the number of optimizer steps in the trace is what the trace says, not what the caller remembers: one
    # adam_clip_kernel launch per step (round 3's ViZDoom summary divided 21 traced steps by an argument of 8)
    adam = sum(r[1] for r in rows if r[0].startswith('adam_clip'))
    if adam:
        if steps and abs(steps - adam) > 0.5:
            print('(argument says %g steps, the trace holds %d optimizer steps: using the trace)' % (steps, adam))
        steps = float(adam)
    if steps:
        print('per training step (%g steps incl. warm-up and extra legs): %.3f ms of kernel time, %.0f launches' %
              (steps, total / 1e6 / steps, sum(r[1] for r in rows) / steps))


if __name__ == '__main__':
    main()
