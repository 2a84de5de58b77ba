#!/usr/bin/env python
"""Ordered kernel list of one steady training step in a rocprofv3 (rocpd) trace of bench.py: start offset, duration, gap to the previous
kernel's end, name.  A step ends with k_adam_segments (or torch's fused optimiser group); the step printed is the second from the end.
usage: step_sequence.py results.db [launches_per_step: take the last N launches instead of looking for the optimiser kernel]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s '
                  'on d.kernel_id = s.id order by d.start').fetchall()
if len(sys.argv) > 2:
    seg, t0 = rows[-int(sys.argv[2]):], rows[-int(sys.argv[2]) - 1][2]
else:
    marks = [i for i, r in enumerate(rows) if 'FusedOptimizerTensorListMetadata' in r[0] or 'k_adam_segments' in r[0]]
    ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] - i > 5]
    a, b = ends[-3], ends[-2]
    seg, t0 = rows[a + 1:b + 1], rows[a][2]
print('# %d launches, %.3f ms from the previous step\'s last kernel to this step\'s' % (len(seg), (seg[-1][2] - t0) / 1e6))
prev = t0
for name, st, en in seg:
    name = re.sub(r'\.kd$', '', name)
    print('+%9.1f us  %8.2f us  gap %7.2f  %s' % ((st - t0) / 1e3, (en - st) / 1e3, max(0, st - prev) / 1e3, name[:110]))
    prev = max(prev, en)
