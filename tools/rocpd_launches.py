#!/usr/bin/env python3
"""Individual launch durations of the kernels whose name contains PATTERN, in launch order (rocpd SQLite database).
    python tools/rocpd_launches.py DB PATTERN [max]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
pat = sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = 0
for name, s, e in rows:
    if pat in name:
        print(f"{n:4d} {(e - s) / 1e3:10.1f} us  {name[:80]}")
        n += 1
        if n >= mx:
            break
