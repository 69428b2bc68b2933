#!/usr/bin/env python
"""Like rocpd_summary.py but grouped by (kernel name, grid size): one row per launch shape.
usage: tools/rocpd_by_grid.py <results.db> [name-substring]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    gx = 'grid_size_x' if 'grid_size_x' in cols else ('grid_x' if 'grid_x' in cols else None)
    wx = 'workgroup_size_x' if 'workgroup_size_x' in cols else None
    if gx is None:
        print('columns:', cols)
        return
    q = ("select name, %s, %s, count(*), avg(duration), min(duration), max(duration) from kernels "
         "where name like ? group by name, %s order by name, %s" % (gx, wx or '0', gx, gx))
    print('| kernel | grid (threads) | wg | calls | avg us | min us | max us |')
    print('|---|---|---|---|---|---|---|')
    for name, g, w, calls, avg, mn, mx in con.execute(q, ('%' + sub + '%',)):
        short = name if len(name) < 100 else name[:97] + '...'
        print('| `%s` | %s | %s | %d | %.2f | %.2f | %.2f |' % (short, g, w, calls, avg / 1e3, mn / 1e3, mx / 1e3))


if __name__ == '__main__':
    main()
