#!/usr/bin/env python3
"""Individual launches (start offset, duration, gap to the previous kernel) of the kernels whose name contains
PATTERN, in launch order (rocpd SQLite database of rocprofv3 --kernel-trace).
    python tools/rocpd_launches.py DB PATTERN [max] [skip]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
pat = sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 64
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
t0 = rows[0][1] if rows else 0
prev_end = t0
n = 0
for name, s, e in rows:
    gap = s - prev_end
    prev_end = e
    if pat in name:
        n += 1
        if n <= skip:
            continue
        short = re.sub(r"^void ", "", name).replace("ac::", "")
        short = re.sub(r"rocprim::ROCPRIM_\w+::detail::", "rocprim::", short)
        print(f"{n:5d} t={(s - t0) / 1e3:10.1f} us  dur={(e - s) / 1e3:8.1f}  gap={gap / 1e3:7.1f}  {short[:70]}")
        if n >= mx + skip:
            break
