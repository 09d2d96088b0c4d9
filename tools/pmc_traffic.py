#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two per-kernel PMC summaries of tools/pmc_round.sh (FETCH_SIZE and WRITE_SIZE passes,
values in KiB per build): HBM-side bytes per build of the kernel bench.py prices the roofline on, calibrated on PackFunctor
whose bytes are known exactly (reads n_text bytes, writes 0.375 n_text).

    python tools/pmc_traffic.py FETCH.csv WRITE.csv N_TEXT TAG [BUILDS_IN_THE_PASS=7] > profiles/pmc_traffic.json"""
import csv
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from source_hash import source_hash


def load(path):
    return {r["Name"]: float(r[[c for c in r if c.endswith("_per_build")][0]]) * 1024 for r in csv.DictReader(open(path))}


def per_pack(path, pat, builds):
    """Bytes per WHOLE-TEXT pack of the kernels matching pat.  The device entry packs the text in two dispatches per build (head on
    stream 0, tail under the first insert phase) and the driver's end repair packs it once more in one: `builds` builds in the
    pass => builds + 1 whole-text packs, however many dispatches that took."""
    tot = 0.0
    for r in csv.DictReader(open(path)):
        if pat in r["Name"]:
            tot += float(r[[c for c in r if c.endswith("_sum")][0]]) * 1024
    return tot / (builds + 1)


# Kernels whose reads are narrow random gathers (one line per lane): their FETCH_SIZE is corrected with the factor calibrated on that very
# access pattern (tools/pmc_calibrate_random.sh -> profiles/pmc_calibration_random.json), not with the streaming x2.
RANDOM_GATHER = ("insert_wave_kernel", "DegreeProbeFunctor", "DegreeFunctor", "LinksFunctor", "PathWalkFunctor", "expand_wave_kernel", "RunOutFunctor",
                 "RunRangeFunctor", "AnswerFunctor", "FirstFunctor", "LevelRelaxFunctor", "LevelPredsFunctor", "RemapFunctor")


def main(fetch_csv, write_csv, n_text, tag, builds=7, workload="configC_k51", calibration=None):
    f, w = load(fetch_csv), load(write_csv)
    pick = lambda d, pat: sum(v for k, v in d.items() if pat in k)
    n_text = int(n_text)
    pf, pw = per_pack(fetch_csv, "PackFunctor", int(builds)), per_pack(write_csv, "PackFunctor", int(builds))
    fcal, wcal = pf / n_text, pw / (0.375 * n_text)
    fcorr = 2.0 if 0.4 < fcal < 0.6 else 1.0     # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md, HBM section) — wide streaming reads
    cal = None
    if calibration is None:
        import os
        cand = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_calibration_random.json")
        calibration = cand if os.path.exists(cand) else None
    if calibration:
        cal = json.load(open(calibration))
    frand = cal.get("factor_for_random_gathers") if cal else None
    if frand is not None:
        frand = max(1.0, min(2.0, float(frand)))      # (between "the raw figure is the traffic" and the streaming correction)
    is_random = lambda name: any(p in name for p in RANDOM_GATHER)
    factor = lambda name: (frand if (frand is not None and is_random(name)) else fcorr)
    kf, kw = pick(f, "insert_wave_kernel"), pick(w, "insert_wave_kernel")
    ins_factor = frand if frand is not None else fcorr
    table = {k: {"fetch_raw": f.get(k, 0.0), "write_raw": w.get(k, 0.0), "fetch_factor": factor(k), "access": "random gather" if is_random(k) else "streaming",
                 "traffic_raw": f.get(k, 0.0) + w.get(k, 0.0), "traffic_streaming_x2": f.get(k, 0.0) * fcorr + w.get(k, 0.0),
                 "hbm_side_bytes": f.get(k, 0.0) * factor(k) + w.get(k, 0.0)}
             for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, 0.0) * fcorr + w.get(k, 0.0)))[:64]}
    print(json.dumps({
        "source_hash": source_hash(),      # the sources the profiled library was built from (bench.py refuses the file when its library differs)
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), tools/pmc_lean.sh (torch-free driver, "
                  f"{builds} builds of {workload} per pass) on MI355X; profiles/{tag}_pmc_*",
        "workload": workload,
        "kernel": "insert_wave_kernel<W> (all phase launches of a build, summed)",
        "fetch_size_raw_bytes": kf, "write_size_raw_bytes": kw,
        "calibration": {"kernel": "functor_kernel<PackFunctor>: reads n_text bytes with 16 B/lane loads, writes 0.375*n_text bytes",
                        "n_text": n_text, "fetch_raw_over_known": fcal, "write_raw_over_known": wcal},
        "calibration_random": ({"file": "profiles/pmc_calibration_random.json", "source_hash": cal.get("source_hash"),
                                "raw_bytes_per_random_read": cal.get("random_read_raw_bytes_per_access_beyond_the_caches"),
                                "factor_for_random_gathers": frand} if cal else None),
        "fetch_correction": ins_factor, "fetch_correction_streaming": fcorr, "write_correction": 1.0,
        "traffic_raw": kf + kw, "traffic_streaming_x2": kf * fcorr + kw,
        "traffic_bytes_per_build": kf * ins_factor + kw,
        "per_kernel_per_build": table,
        "note": "FETCH_SIZE on gfx950 tallies 128-B requests at 64 B (confirmed here on PackFunctor), so STREAMING kernels' fetches are doubled; kernels whose "
                "reads are narrow random gathers (the insert, the probing stages, the walks) use the factor calibrated on that pattern "
                "(pmc_calibration_random.json: raw bytes per random 8-byte read beyond the caches; 64 B per access = factor 1) when that file exists, "
                "else the streaming factor as an upper bound (lower bound = traffic_raw)."}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:8])
