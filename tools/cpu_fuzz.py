#!/usr/bin/env python3
"""One-off randomised parity sweep on the CPU emulation of the kernels (tests/_emu) against the oracle — the CPU counterpart of
tools/gpu_fuzz.py: adversarial generator cases beyond the seeds the test-suite uses, random synthetic assembly sets, and the
sharded entry points at world size 1.  ~20 minutes.     python tools/cpu_fuzz.py"""
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import emu_lib, parity_util, seqgen
from autocycler_amd import synth
emu = emu_lib.emu_path()
t0=time.time(); n=0
for seed in range(24, 700):
    for k in (5, 11, 21, 31, 51):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd, lib_path=emu, repair=(seed % 2 == 0)); n+=1
    if time.time()-t0 > 420: break
print('adversarial cases OK', n, 'up to seed', seed, flush=True)
t0=time.time(); m=0
for seed in range(1000, 1400):
    r = random.Random(seed)
    k = r.choice([21, 31, 51, 51, 77, 101])
    na = r.randint(2, 10); genome = r.choice([8000, 20000, 50000, 120000]); plasmid = r.choice([0, 1500, 4000])
    sub = r.choice([1e-4, 1e-3, 5e-3, 2e-2]); indel = r.choice([0, 1e-5, 1e-4, 1e-3])
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(na, genome=genome, plasmid=plasmid, sub=sub, indel=indel, seed=seed)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    parity_util.check_case(k, seqs, fn, hd, lib_path=emu); m+=1
    if time.time()-t0 > 420: break
print('synthetic sets OK', m, flush=True)

import torch
import sharded_util
from autocycler_amd import sharded
comm = sharded.Comm(torch.device("cpu"))
t0=time.time(); n=0
for seed in range(24, 400):
    for k in (5, 11, 31, 51):
        seqs, fn, hd = seqgen.make_case(seed, k)
        sharded_util.run_case(emu, k, seqs, fn, hd, comm, torch.device("cpu"), repair=(seed % 2 == 0)); n+=1
    if time.time()-t0 > 300: break
print('sharded world-1 cases OK', n, 'up to seed', seed)
