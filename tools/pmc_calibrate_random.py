#!/usr/bin/env python3
"""The two halves of tools/pmc_calibrate_random.sh: `run` drives the library's random-access kernels over three table sizes (under
rocprofv3), `summarise` turns the two counter CSVs into bytes per access for each (kernel, table size)."""
import csv
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SIZES = [1 << 24, 1 << 27, 1 << 30]      # slots of 8 bytes: 128 MB, 1 GB, 8 GB
REPS = 4                                   # repetitions inside ac_random_access_ceilings_at
N_READ = 48_000_000


def run():
    os.environ.setdefault("AC_NO_TORCH", "1")
    from autocycler_amd import _capi
    lib = _capi.load_library()
    for slots in SIZES:
        cas, rd = C.c_double(), C.c_double()
        rc = lib.ac_random_access_ceilings_at(C.c_int(0), C.c_uint64(slots), C.byref(cas), C.byref(rd))
        print(json.dumps({"slots": slots, "rc": rc, "cas_gops": cas.value, "read_gops": rd.value}), flush=True)


def summarise(fetch_csv, write_csv, run_jsonl):
    sys.path.insert(0, str(ROOT / "tools"))
    from source_hash import source_hash
    rates = [json.loads(l) for l in open(run_jsonl) if l.startswith("{")]

    def per_kernel(path):
        out = {"RbReadFunctor": [], "RbClaimFunctor": []}
        for row in csv.DictReader(open(path, newline="")):
            for k in out:
                if k in row["Kernel_Name"]:
                    out[k].append(float(row["Counter_Value"]) * 1024.0)      # FETCH_SIZE / WRITE_SIZE are in KiB on gfx950
        return out
    f, w = per_kernel(fetch_csv), per_kernel(write_csv)
    rows = []
    for i, slots in enumerate(SIZES):
        n_cas = min(slots // 2, 48_000_000)
        for kern, n_ops in (("RbReadFunctor", N_READ), ("RbClaimFunctor", n_cas)):
            fr, wr = f[kern][i * REPS:(i + 1) * REPS], w[kern][i * REPS:(i + 1) * REPS]
            if not fr or not wr:
                continue
            # (the first repetition of a size touches a cold table: the steady ones are what a build sees)
            fr, wr = fr[1:] or fr, wr[1:] or wr
            rows.append({"kernel": kern, "table_bytes": slots * 8, "accesses": n_ops,
                         "fetch_raw_bytes_per_access": sum(fr) / len(fr) / n_ops, "write_raw_bytes_per_access": sum(wr) / len(wr) / n_ops,
                         "rate_gops": (rates[i]["read_gops"] if kern == "RbReadFunctor" else rates[i]["cas_gops"]) if i < len(rates) else None})
    reads = [r for r in rows if r["kernel"] == "RbReadFunctor"]
    big = [r for r in reads if r["table_bytes"] >= (1 << 30)] or reads
    per_read = sum(r["fetch_raw_bytes_per_access"] for r in big) / max(len(big), 1)
    print(json.dumps({
        "source_hash": source_hash(),
        "what": "FETCH_SIZE / WRITE_SIZE (raw, KiB x 1024) per random 8-byte access of the library's own RbReadFunctor / RbClaimFunctor (one access per lane, 64 "
                "different lines per wavefront instruction) over tables of 128 MB, 1 GB and 8 GB; rocprofv3 --pmc, one counter per pass, --kernel-trace only",
        "rows": rows,
        "random_read_raw_bytes_per_access_beyond_the_caches": per_read,
        "reading": "a random 8-byte read that misses the L2 moves ONE 64-byte sector from the memory side: if the raw counter shows ~64 B per access the raw "
                   "figure IS the traffic of a random-gather kernel (factor 1); ~32 B per access would mean the x2 of the streaming calibration applies here too",
        "factor_for_random_gathers": (64.0 / per_read) if per_read > 0 else None}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarise(*sys.argv[2:5])
