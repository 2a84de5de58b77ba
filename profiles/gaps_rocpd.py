#!/usr/bin/env python
"""GPU idle time inside steady-state steps of a rocprofv3 (rocpd sqlite) kernel trace of bench.py: steps are delimited by the
fused-Adam kernel; for the last N steps prints wall, busy (union of kernel intervals) and the largest gaps with the kernels
around them.  usage: gaps_rocpd.py results.db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = db.execute('select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d '
                  'join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start').fetchall()
marks = [i for i, r in enumerate(rows) if 'FusedOptimizerTensorListMetadata' in r[0] or 'k_adam_segments' in r[0]]
# a step ends with the LAST optimiser kernel of its group (torch: consecutive multi_tensor launches; the library: one k_adam_segments)
ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] - i > 5]
print('steps found:', len(ends))
for a, b in list(zip(ends[:-1], ends[1:]))[-N:]:
    seg = rows[a + 1:b + 1]
    wall = seg[-1][2] - rows[a][2]
    busy, cur_end, gaps = 0, rows[a][2], []
    for name, st, en in seg:
        if st > cur_end:
            gaps.append((st - cur_end, name))
        busy += max(0, en - max(st, cur_end))
        cur_end = max(cur_end, en)
    gaps.sort(reverse=True)
    print('step: wall %.3f ms, busy %.3f ms, idle %.3f ms in %d gaps; largest: %s' % (
        wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(gaps),
        ', '.join('%.0f us before %s' % (g / 1e3, n[:40]) for g, n in gaps[:4])))
