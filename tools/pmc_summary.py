#!/usr/bin/env python3
"""Per-kernel sums of one rocprofv3 PMC counter from a `*_counter_collection.csv` (rocprofv3 --pmc X -f csv).

    python tools/pmc_summary.py counter_collection.csv BUILDS > summary.csv

BUILDS = number of graph builds the profiled command ran (warmup + steps); the per-build column divides by it.
Counter values are printed raw (FETCH_SIZE / WRITE_SIZE are in KiB on gfx950); corrections are applied by the
reader (profiles/README.md), calibrated on PackFunctor whose byte counts are known exactly."""
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_stats import short


def main(path, builds):
    agg = {}
    ctr = None
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = short(row["Kernel_Name"])
            ctr = row["Counter_Name"]
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    print(f"Name,Dispatches,{ctr}_sum,{ctr}_per_build")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{name}\",{a[0]},{a[1]:.1f},{a[1] / builds:.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
