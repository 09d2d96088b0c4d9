#!/usr/bin/env python3
"""A/B of the library's tuning knobs on one MI355X, without torch (starts in seconds): config C of bench.py, the same
device-resident text for every variant, wall-clock per build, stage table, and a digest of the finished graph (all variants
must produce the same graph).

    AC_NO_TORCH=1 python tools/ab_knobs.py [--assemblies 96] [--steps 8] [--variants "base;AC_DEGREE_VARIANT=1;AC_TABLE_SHIFT=1"]
"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assemblies", type=int, default=96)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--kmer", type=int, default=51)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--host-entry", action="store_true", help="build through ac_compress_build from host views (the T_hot bracket: upload included)")
    ap.add_argument("--workload", type=str, default=None, help="a named workload of autocycler_amd.synth.WORKLOADS (overrides --assemblies/--genome/--kmer)")
    ap.add_argument("--lib", type=str, default=None, help="another build of the library (e.g. one made with -DAC_MEASUREMENT_KNOBS)")
    ap.add_argument("--emu", action="store_true", help="dry run of this script on the CPU emulation (tests/_emu), small sizes only")
    ap.add_argument("--variants", type=str, default="base;AC_TABLE_SHIFT=0,AC_MINKEY_VARIANT=0;AC_TABLE_SHIFT=2;base")
    args = ap.parse_args()
    import numpy as np
    from autocycler_amd import _capi, synth
    import bench
    if args.emu:
        sys.path.insert(0, str(ROOT / "tests"))
        import emu_lib
        lib = _capi.load_library(emu_lib.emu_path())
        hip = None
    else:
        lib = _capi.load_library(args.lib)
        hip = C.CDLL("libamdhip64.so.7")
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    k = args.kmer
    ns = argparse.Namespace(assemblies=args.assemblies, genome=args.genome, plasmid=100_000, sub=1e-4, indel=1e-5, species="per-gpu")
    t0 = time.time()
    if args.workload:
        k, args.assemblies, gen = synth.WORKLOADS[args.workload]
        seqs, fn, hd = synth.flatten(gen())
    else:
        seqs, fn, hd = bench.make_inputs(ns, 0, False)
    h_seqs = bench.prepare(lib, k, seqs, fn, hd, args.assemblies, threads=os.cpu_count() or 1, repair=0)
    del seqs
    n = lib.ac_seqs_count(h_seqs)
    views = lib.ac_seqs_views(h_seqs)
    n_text = lib.ac_text_size(C.c_uint32(k), views, C.c_uint32(n))
    text = np.empty(n_text, dtype=np.uint8)
    off = (C.c_uint64 * n)(); d1 = (C.c_uint16 * n)(); d2 = (C.c_uint16 * n)()
    assert lib.ac_layout_text(C.c_uint32(k), views, C.c_uint32(n), text.ctypes.data_as(C.c_void_p), off, d1, d2) == 0
    lens = (C.c_uint32 * n)(*[views[i].length for i in range(n)])
    ids = (C.c_uint16 * n)(*[views[i].id for i in range(n)])
    bases = sum(lens)
    if hip is None:
        d_text = C.c_void_p(text.ctypes.data)
    else:
        d_text = C.c_void_p()
        assert hip.hipSetDevice(0) == 0
        assert hip.hipMalloc(C.byref(d_text), C.c_size_t(n_text)) == 0
        assert hip.hipMemcpy(d_text, text.ctypes.data_as(C.c_void_p), C.c_size_t(n_text), C.c_int(1)) == 0
    secs, nm = C.c_double(), C.c_uint64()
    if lib.ac_end_repair_device(C.c_uint32(k), d_text, C.c_uint64(n_text), off, lens, d1, d2, C.c_uint32(n), C.c_int(0), C.byref(secs), C.byref(nm)):
        raise RuntimeError(lib.ac_last_error().decode())
    print(json.dumps({"prep_s": time.time() - t0, "bases": bases, "sequences": n, "repair_matches": nm.value}), flush=True)

    hviews = None
    if args.host_entry:      # the repaired text back in pageable host memory, one view per sequence (what the Rust caller holds)
        if hip is not None:
            assert hip.hipMemcpy(text.ctypes.data_as(C.c_void_p), d_text, C.c_size_t(n_text), C.c_int(2)) == 0
        hviews = (_capi.SeqView * n)()
        for i in range(n):
            hviews[i].fwd = C.cast(C.c_void_p(text.ctypes.data + off[i]), C.c_char_p)
            hviews[i].length = lens[i]; hviews[i].id = ids[i]

    def build():
        h = C.c_void_p()
        if hviews is not None:
            if lib.ac_compress_build(C.c_uint32(k), C.c_uint32(args.assemblies), hviews, C.c_uint32(n), C.c_int(0), C.byref(h)):
                raise RuntimeError(lib.ac_last_error().decode())
            return _capi.Graph(lib, h, n)
        if lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(args.assemblies), d_text, C.c_uint64(n_text), off, lens, ids, d1, d2,
                                        C.c_uint32(n), C.c_int(0), C.byref(h)):
            raise RuntimeError(lib.ac_last_error().decode())
        return _capi.Graph(lib, h, n)

    def digest(g):
        m = hashlib.md5()
        m.update(g.gfa(fn, hd).encode())
        return m.hexdigest()

    for _ in range(2):
        build().close()
    for variant in args.variants.split(";"):
        for kn in [kn for kn in os.environ if kn.startswith("AC_") and kn != "AC_NO_TORCH"]:
            del os.environ[kn]
        if variant != "base":
            for kv in variant.split(","):
                a, b = kv.split("=")
                os.environ[a] = b
        try:
            for _ in range(2):
                build().close()
            if hip is not None:
                hip.hipDeviceSynchronize()
            ts = []
            ins = []
            upl = []
            for _ in range(args.steps):
                t1 = time.perf_counter()
                g = build()
                ts.append((time.perf_counter() - t1) * 1e3)
                ins.append(g.timings()["insert_kernel_ms"])
                upl.append((g.timings()["upload_device_ms"], g.timings()["h2d"] * 1e3, g.timings()["total_device"] * 1e3))
                g.close()
            lib.ac_set_stage_timing(C.c_int(1))
            g = build()
            tm = g.timings()
            lib.ac_set_stage_timing(C.c_int(0))
            dg = digest(g)
            st = g.stats_post
            g.close()
            ts_sorted = sorted(ts)
            print(json.dumps({"variant": variant, "ms_median": ts_sorted[len(ts) // 2], "ms_min": ts_sorted[0], "ms_mean": sum(ts) / len(ts),
                              "Mbp_s_median": bases / 1e3 / ts_sorted[len(ts) // 2], "insert_kernel_ms": sum(ins) / len(ins),
                              **({"upload_device_ms": sorted(u[0] for u in upl)[len(upl) // 2], "upload_host_side_ms": sorted(u[1] for u in upl)[len(upl) // 2],
                                  "build_ms": sorted(u[2] for u in upl)[len(upl) // 2]} if hviews is not None else {}),
                              "stages_ms": {a: round(b * 1e3, 3) for a, b in tm.items() if isinstance(b, float) and b > 2e-5 and a not in ("insert_kernel_ms", "insert_rest_known", "insert_rest_sampled")},
                              "table_capacity": tm["table_capacity"], "insert_launches": tm["insert_launches"], "launches": tm.get("launches"), "readbacks": tm.get("readbacks"), "unitigs": st["unitigs"], "gfa_md5": dg,
                              "path_runs_copied": tm["path_runs_copied"], "path_entries_walked": tm["path_entries_walked"], "position_retries": tm["position_retries"],
                              "path_entries": tm.get("n_path_entries"), "path_stretches": tm.get("path_stretches"),
                              "expand": {q: tm[q] for q in ("n_candidates", "n_levels", "simplify_passes", "expand_sparse_sweeps", "expand_sparse_start") if q in tm}}), flush=True)
        except Exception as e:      # a variant that fails must not take the others with it
            print(json.dumps({"variant": variant, "error": str(e)}), flush=True)
    lib.ac_seqs_free(h_seqs)


if __name__ == "__main__":
    main()
