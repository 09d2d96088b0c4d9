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
os.environ.setdefault("AC_TUNING_FOLLOW_ENV", "1")      # the variants change AC_* knobs between builds of this process


def bench_job(args):
    """The N-species job of bench.py's sharded mode as ONE call per step from ONE process over N devices."""
    import argparse as _a
    import bench
    from autocycler_amd import _capi
    lib = _capi.load_library()
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    N, k = args.bench_job, 51
    if lib.ac_device_count() < N:
        print(json.dumps({"error": f"{lib.ac_device_count()} devices visible, {N} needed"})); return
    ns = _a.Namespace(assemblies=args.assemblies, genome=args.genome, plasmid=100_000, sub=1e-4, indel=1e-5, species="per-gpu")
    seqs, fn, hd = [], [], []
    t0 = time.time()
    for r in range(N):
        s, f, h_ = bench.make_inputs(ns, r, True)
        seqs += s; fn += f; hd += h_
    h = bench.prepare(lib, k, seqs, fn, hd, N * args.assemblies, threads=os.cpu_count() or 1, repair=1)
    n = lib.ac_seqs_count(h)
    views = lib.ac_seqs_views(h)
    bases = sum(views[i].length for i in range(n))
    t_prep = time.time() - t0
    dv = (C.c_int * N)(*range(N))
    times, g = [], None
    for i in range(args.warmup + args.steps):
        if g is not None:
            g.close()
        hg = C.c_void_p()
        t1 = time.perf_counter()
        if lib.ac_compress_build_multi(C.c_uint32(k), C.c_uint32(N * args.assemblies), views, C.c_uint32(n), dv, C.c_int(N), C.byref(hg)):
            print(json.dumps({"error": lib.ac_last_error().decode()})); return
        dt = time.perf_counter() - t1
        g = _capi.Graph(lib, hg, n)
        if i >= args.warmup:
            times.append(dt)
    mi = _capi.MultiInfo()
    lib.ac_multi_info_get(g._h, C.byref(mi))
    med = sorted(times)[len(times) // 2]
    print(json.dumps({"what": f"ONE job of {N} species x {args.assemblies} assemblies through ac_compress_build_multi: one process, one host thread per device, "
                              "sequences in host RAM -> final graph in host RAM (upload included), exchanges inside the library",
                      "n_devices": N, "bases": bases, "value": bases / 1e6 / med, "unit": "Mbp/s", "ms_per_step": med * 1e3, "steps": len(times),
                      "step_ms": [round(t * 1e3, 2) for t in times], "prep_s": t_prep, "unitigs": g.stats_post["unitigs"],
                      "gfa_md5": hashlib.md5(g.gfa(fn, hd).encode()).hexdigest() if N <= 2 else None, "multi": mi.as_dict()}), flush=True)
    g.close()
    lib.ac_seqs_free(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="configC_k51")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--worlds", default="1,2,4")
    ap.add_argument("--bench-job", type=int, default=0, help="N > 0: bench.py's one-job workload at N GPUs (species r = 96 assemblies of the config C model, "
                    "r < N) through ac_compress_build_multi on devices 0 .. N-1 (one rank per device, RCCL): what bench.py --gpus N reports as `in_library_multi`")
    ap.add_argument("--assemblies", type=int, default=96)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    if args.bench_job:
        return bench_job(args)
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
