#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 (rocpd sqlite) kernel trace: calls, total/avg/min/max duration, share.
usage: summarize_rocpd.py results.db [--steps N] [--skip-first-frac F]  > summary.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*\)$', '', name)          # drop the argument list
    return name if len(name) <= 110 else name[:107] + '...'


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else None
    cur = db.cursor()
    rows = cur.execute('select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d '
                       'join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start').fetchall()
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1] if rows else 0
    print('# kernels: %d dispatches, %.3f ms busy, %.3f ms first-start..last-end' % (len(rows), total / 1e6, span / 1e6))
    if steps:
        print('# per step (%d steps incl. warm-up): %.3f ms of kernel time' % (steps, total / 1e6 / steps))
    print('%-112s %8s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-112s %8d %12.3f %10.2f %10.2f %10.2f %6.2f' % (name, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3,
                                                                 a[3] / 1e3, 100.0 * a[1] / total))


if __name__ == '__main__':
    main()
