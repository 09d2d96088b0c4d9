#!/usr/bin/env python3
"""BASELINE.json configs[4] on one MI355X: SPECIES species x STRAINS strains (1 % apart) x ~GENOME bp, one compress job, k = 51
(`synth.make_mixed_species`): end repair on the device, BUILDS builds, one more with the stage table, and (--check) the
size-independent properties of tests/fullsize_e.py evaluated on the device.  One JSON line on stdout.
    python tools/config_e.py [--species 25] [--strains 40] [--genome 5000000] [--builds 2] [--check]"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--species", type=int, default=25)
    ap.add_argument("--strains", type=int, default=40)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--builds", type=int, default=2)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--no-stages", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import fullsize_e
    from autocycler_amd import _capi
    log = lambda *a: print(*a, file=sys.stderr, flush=True)
    lib = _capi.load_library()
    mem_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30
    log(f"host: {os.cpu_count()} cpus, {mem_gb:.0f} GB")
    job = fullsize_e.make_job(args.species, args.strains, genome=args.genome, plasmid=args.genome // 50)
    log(f"job: {job['n']} sequences, {job['bases']} bases, generate {job['generate_s']:.1f}s layout {job['layout_s']:.1f}s")
    t0 = time.time()
    d_text = torch.from_numpy(job["text"]).to("cuda:0")
    torch.cuda.synchronize()
    log(f"text on the device {time.time() - t0:.1f}s")
    g, times, repair_s = fullsize_e.build_device(lib, job, d_text.data_ptr(), repair=True, builds=args.builds)
    log(f"builds: {times}")
    tm = g.timings()
    stages = None
    if not args.no_stages:
        lib.ac_set_stage_timing(1)
        g.close()
        g, t_st, _ = fullsize_e.build_device(lib, job, d_text.data_ptr(), repair=False, builds=1)
        lib.ac_set_stage_timing(0)
        tm = g.timings()
        stages = {kk: round(v * 1e3, 3) for kk, v in tm.items() if isinstance(v, float) and kk not in ("insert_kernel_ms", "upload_device_ms") and v}
        log(f"stage build: {t_st[0]:.4f}s {stages}")
    out = {"workload": f"{args.species} species x {args.strains} strains x ~{args.genome} bp (1 % strain divergence), k=51, one MI355X",
           "bases": job["bases"], "sequences": job["n"], "host_cpus": os.cpu_count(), "host_mem_gb": round(mem_gb),
           "generate_s": job["generate_s"], "end_repair_device_s": repair_s, "build_s": times,
           "Mbp_per_s": job["bases"] / 1e6 / min(times), "stages_ms": stages,
           "graph": {**g.stats_post, "pre_total_length": g.stats_pre["total_length"], "kmers": g.kmer_count, "path_entries": tm["n_path_entries"],
                     "table_capacity": tm["table_capacity"], "n_distinct": tm["n_distinct"], "passes": tm["simplify_passes"],
                     "levels": tm["n_levels"], "candidates": tm["n_candidates"], "insert_kernel_ms": tm["insert_kernel_ms"],
                     "insert_real": tm["insert_real"], "insert_launches": tm["insert_launches"]}}
    if args.check:
        lib.ac_release_memory()
        out["check"] = fullsize_e.check_on_device(g, job, d_text, log=log)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
