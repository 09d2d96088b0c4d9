#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average duration) out of a rocprofv3 `--kernel-trace` result database (rocpd sqlite, the
default output format of rocprofv3 7.x): what `--stats` printed as a CSV in earlier versions.
    python tools/rocpd_stats.py RESULTS.db [--builds N] [--top 40]"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:ac::)?functor_kernel(?:_full)?<(?:ac::)?(.*)>\(", name)
    if m:
        name = m.group(1)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--builds", type=int, default=1, help="divide totals by this many builds")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.start), max(d.end) from {kd} d join {ks} s on d.kernel_id = s.id "
                     f"group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f"# {n} dispatches, {total / 1e6:.3f} ms of kernel time; per build (/{args.builds}): {n / args.builds:.0f} launches, {total / 1e6 / args.builds:.3f} ms")
    print("kernel,calls_per_build,total_ms_per_build,avg_us,share")
    for name, calls, dur, _, _ in rows[:args.top]:
        print(f"{short(name)},{calls / args.builds:.1f},{dur / 1e6 / args.builds:.4f},{dur / 1e3 / calls:.2f},{dur / total:.3f}")


if __name__ == "__main__":
    main()
