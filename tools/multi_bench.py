#!/usr/bin/env python3
"""ac_compress_build_multi on the one-GPU box: config C (or --workload) from host views through the multi-device entry with 1 rank (RCCL in a
world of one), and 2 / 4 / 8 ranks sharing device 0 (host-staged exchanges) next to the single-device host entry — what the protocol costs
and moves, not a scaling measurement (the ranks time-slice one device).  One JSON line per variant.
    AC_NO_TORCH=1 python tools/multi_bench.py [--workload configC_k51] [--steps 4] [--worlds 1,2,4]"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("AC_NO_TORCH", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="configC_k51")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--worlds", default="1,2,4")
    args = ap.parse_args()
    import bench
    from autocycler_amd import _capi, synth
    lib = _capi.load_library()
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    k, n_asm, gen = synth.WORKLOADS[args.workload]
    seqs, fn, hd = synth.flatten(gen())
    h = bench.prepare(lib, k, seqs, fn, hd, n_asm, threads=os.cpu_count() or 1, repair=1)
    n = lib.ac_seqs_count(h)
    views = lib.ac_seqs_views(h)
    bases = sum(views[i].length for i in range(n))

    def run(devices):
        g = C.c_void_p()
        t0 = time.perf_counter()
        if devices is None:
            rc = lib.ac_compress_build(C.c_uint32(k), C.c_uint32(n_asm), views, C.c_uint32(n), C.c_int(0), C.byref(g))
        else:
            dv = (C.c_int * len(devices))(*devices)
            rc = lib.ac_compress_build_multi(C.c_uint32(k), C.c_uint32(n_asm), views, C.c_uint32(n), dv, C.c_int(len(devices)), C.byref(g))
        dt = time.perf_counter() - t0
        if rc:
            raise RuntimeError(lib.ac_last_error().decode())
        return _capi.Graph(lib, g, n), dt

    for label, devices in [("single-device host entry", None)] + [(f"multi, {w} rank(s) on device 0", [0] * w) for w in map(int, args.worlds.split(","))]:
        times = []
        for i in range(args.steps + 2):
            g, dt = run(devices)
            if i >= 2:
                times.append(dt)
            if i < args.steps + 1:
                g.close()
        info = None
        if devices is not None:
            mi = _capi.MultiInfo()
            lib.ac_multi_info_get(g._h, C.byref(mi))
            info = mi.as_dict()
        print(json.dumps({"variant": label, "workload": args.workload, "bases": bases, "ms_median": sorted(times)[len(times) // 2] * 1e3, "ms_min": min(times) * 1e3,
                          "gfa_md5": hashlib.md5(g.gfa(fn, hd).encode()).hexdigest(), "unitigs": g.stats_post["unitigs"], "multi": info}), flush=True)
        g.close()
    lib.ac_seqs_free(h)


if __name__ == "__main__":
    main()
