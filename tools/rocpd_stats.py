#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (the `--kernel-trace --stats` output of
ROCm 7.2): calls, total / average / min / max duration, share of GPU kernel time.  Writes a CSV summary.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [BUILDS] > profiles/r01_kernel_stats.csv

BUILDS (optional): the number of graph builds the profiled command ran — adds calls and milliseconds per build."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("ac::", "")
    m = re.match(r"rocprim::ROCPRIM_\w+::detail::(\w+)<rocprim::ROCPRIM_\w+::detail::wrapped_(\w+?)_config<", name)
    if m:   # rocPRIM kernels carry their whole instantiation in the name: keep the primitive only
        return f"rocprim::{m.group(2)} ({m.group(1)})"
    m = re.match(r"rocprim::ROCPRIM_\w+::detail::(\w+)", name)
    if m:
        return f"rocprim::{m.group(1)}"
    return re.sub(r"\(.*$", "", name)


def main(path, builds=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, start, end in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = end - start
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    extra = ",CallsPerBuild,MsPerBuild" if builds else ""
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs" + extra)
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tail = f",{a[0] / builds:.1f},{a[1] / 1e6 / builds:.4f}" if builds else ""
        print(f"\"{name}\",{a[0]},{a[1]},{a[1] / a[0]:.0f},{100.0 * a[1] / total:.2f},{a[2]},{a[3]}" + tail)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
