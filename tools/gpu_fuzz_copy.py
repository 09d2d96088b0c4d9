#!/usr/bin/env python3
"""Randomised device parity sweep for the opt-in copying path walk (AC_PATH_COPY=1) and the position bound (AC_POS_CAP): random
redundant assembly sets — where the insert follows long runs and the copying walk engages — against the oracle, under random
settings of the run piece length and the bound.  Time-limited:   python tools/gpu_fuzz_copy.py [SECONDS]"""
import os, random, sys, time
os.environ.setdefault("AC_TUNING_FOLLOW_ENV", "1")      # (the knobs are read once per process otherwise)
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import parity_util
from autocycler_amd import synth

limit = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
t0 = time.time(); n = 0; engaged = 0; retries = 0
for seed in range(5000, 9000):
    r = random.Random(seed)
    k = r.choice([21, 31, 51, 51, 51, 77, 101])
    na = r.randint(4, 14); genome = r.choice([20000, 50000, 120000, 250000]); plasmid = r.choice([0, 1500, 4000])
    sub = r.choice([0, 1e-5, 1e-4, 1e-3, 5e-3]); indel = r.choice([0, 1e-5, 1e-4])
    os.environ["AC_PATH_COPY"] = r.choice(["1", "1", "1", "0"])
    os.environ["AC_RUN_PIECE"] = r.choice(["4096", "150", "700", "100000"])
    os.environ["AC_POS_CAP"] = r.choice(["65536", "0", "3", "50", "1000"])
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(na, genome=genome, plasmid=plasmid, sub=sub, indel=indel, seed=seed)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    g, _, _ = parity_util.check_case(k, seqs, fn, hd)
    tm = g.timings()
    engaged += tm["path_runs_copied"] > 0; retries += tm["position_retries"]; n += 1
    if time.time() - t0 > limit:
        break
print(f"copying walk / position bound fuzz OK: {n} sets, the copying walk engaged on {engaged}, {retries} builds repeated with exact positions", flush=True)
