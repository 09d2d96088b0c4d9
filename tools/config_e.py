#!/usr/bin/env python3
"""BASELINE config E in miniature on one MI355X: SPECIES species x STRAINS assemblies of ~5 Mbp (strains 1 % diverged from their
species root), one compress job, k = 51: end repair on the device, graph build, internal consistency checks, timing.
    python tools/config_e.py [SPECIES=5] [STRAINS=40]"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import bench
from autocycler_amd import _capi, synth

species = int(sys.argv[1]) if len(sys.argv) > 1 else 5
strains = int(sys.argv[2]) if len(sys.argv) > 2 else 40
k = 51
lib = _capi.load_library()
lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
lib.ac_seqs_count.restype = C.c_uint32
lib.ac_seqs_free.argtypes = [C.c_void_p]
t0 = time.time()
seqs, fn, hd = [], [], []
for sp in range(species):
    for i, contigs in enumerate(synth.make_assemblies(strains, sub=1e-2, indel=1e-4, seed=900_000 + 1000 * sp)):
        for header, s in contigs:
            seqs.append(np.ascontiguousarray(s)); fn.append(f"species{sp:02d}_strain{i:03d}.fasta"); hd.append(header)
t_gen = time.time() - t0
n_asm = species * strains
h = bench.prepare(lib, k, seqs, fn, hd, n_asm, threads=32, repair=0)
n = lib.ac_seqs_count(h)
views = lib.ac_seqs_views(h)
n_text = lib.ac_text_size(C.c_uint32(k), views, C.c_uint32(n))
text = np.empty(n_text, dtype=np.uint8)
off = (C.c_uint64 * n)(); d1 = (C.c_uint16 * n)(); d2 = (C.c_uint16 * n)()
assert lib.ac_layout_text(C.c_uint32(k), views, C.c_uint32(n), text.ctypes.data_as(C.c_void_p), off, d1, d2) == 0
lens = (C.c_uint32 * n)(*[views[i].length for i in range(n)])
ids = (C.c_uint16 * n)(*[views[i].id for i in range(n)])
bases = sum(lens)
d_text = torch.from_numpy(text).to("cuda:0")
secs = C.c_double()
assert lib.ac_end_repair_device(C.c_uint32(k), C.c_void_p(d_text.data_ptr()), C.c_uint64(n_text), off, lens, d1, d2, C.c_uint32(n), C.c_int(0),
                                C.byref(secs), None) == 0, lib.ac_last_error()
times = []
g = None
for it in range(3):
    if g is not None:
        g.close()
    hg = C.c_void_p()
    t1 = time.perf_counter()
    rc = lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(n_asm), C.c_void_p(d_text.data_ptr()), C.c_uint64(n_text), off, lens, ids, d1, d2,
                                      C.c_uint32(n), C.c_int(0), C.byref(hg))
    assert rc == 0, lib.ac_last_error()
    times.append(time.perf_counter() - t1)
    g = _capi.Graph(lib, hg, n)
tm = g.timings()
print(json.dumps({"workload": f"{species} species x {strains} strains x ~5 Mbp, 1 % strain divergence, k={k}, one MI355X", "bases": bases,
                  "sequences": n, "generate_s": t_gen, "end_repair_device_s": secs.value, "build_s": times, "Mbp_per_s": bases / 1e6 / min(times),
                  "graph": {**g.stats_post, "kmers": g.kmer_count, "path_entries": tm["n_path_entries"], "table_capacity": tm["table_capacity"]}}))
