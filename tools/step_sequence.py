#!/usr/bin/env python
"""Ordered kernel list of the LAST training step in a rocprofv3 (rocpd) trace: name, duration, gap to the previous kernel.
usage: step_sequence.py results.db launches_per_step"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2])
rows = db.execute('select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s '
                  'on d.kernel_id = s.id order by d.start').fetchall()
rows = rows[-per:]
prev = None
for name, st, en in rows:
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*\)$', '', name)[:90]
    print('%8.2f us  gap %7.2f  %s' % ((en - st) / 1e3, 0.0 if prev is None else (st - prev) / 1e3, name))
    prev = en
