"""BASELINE configs[4] at full size on ONE MI355X, timed: 25 species x 40 strains x ~5 Mbp = 1000 assemblies (5.08 G bp, k = 51), text
resident in HBM, `--builds` builds (the first one grows the arena: not counted), stage table of the last one, ac_verify_graph_device at
the end.  tests/test_gpu_fullsize.py::test_config_e_full_size_k51 is the same job as a test (one build, checks only).

    AC_NO_TORCH= python tools/fullsize_e_time.py [--builds 3] [--species 25 --strains 40]      (needs torch for the device copy of the text)
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--builds", type=int, default=3)
    ap.add_argument("--species", type=int, default=25)
    ap.add_argument("--strains", type=int, default=40)
    args = ap.parse_args()
    import torch
    import fullsize_e
    from autocycler_amd import _capi
    lib = _capi.load_library()
    t0 = time.time()
    job = fullsize_e.make_job(args.species, args.strains)
    t_gen = time.time() - t0
    d_text = torch.from_numpy(job["text"]).to("cuda:0")
    lib.ac_set_stage_timing(1)
    g, times, repair_s = fullsize_e.build_device(lib, job, d_text.data_ptr(), repair=True, builds=args.builds + 1)
    tm = g.timings()
    stages = {k: round(v * 1e3, 1) for k, v in tm.items() if isinstance(v, float) and 0 < v < 100 and not k.endswith("_ms") and k not in ("insert_rest_known", "insert_rest_sampled")}
    lib.ac_release_memory()
    r = g.verify_device(d_text.data_ptr(), job["n_text"], job["off"], job["lens"])
    print(json.dumps({"what": f"{args.species} species x {args.strains} strains x ~5 Mbp, k = 51, one device, text resident in HBM",
                      "bases": job["bases"], "n_text": job["n_text"], "generate_s": round(t_gen, 1), "end_repair_s": repair_s,
                      "build_s": [round(t, 4) for t in times], "first_build_s_not_counted": round(times[0], 4),
                      "Gbp_per_s_best": round(job["bases"] / 1e9 / min(times[1:]), 2), "stages_ms_last_build": stages,
                      "insert_kernel_ms": tm.get("insert_kernel_ms"), "launches": tm.get("launches"), "round_trips": tm.get("readbacks"), "path_stretches": tm.get("path_stretches"),
                      "unitigs": r["unitigs"], "path_entries": r["path_entries"], "verify_failed": r["failed"], "kmers": g.kmer_count}))
    g.close()


if __name__ == "__main__":
    main()
