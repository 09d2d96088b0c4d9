"""Shared driver of the sharded-build tests: slices one job's sequences over `world` ranks, runs this rank's part
through the library (product .so on the GPU box, the serial emulation on CPU) and, on the root, checks the result
against the oracle byte-for-byte."""
import ctypes as C

import numpy as np
import torch

import oracle_lib as O
import parity_util
from autocycler_amd import _capi, sharded


def slice_bounds(n, world):
    """Contiguous, near-equal slices of n sequences (every rank gets at least one: n >= world)."""
    return [(r * n) // world for r in range(world + 1)]


def local_shard(lib, k, loaded, lo, hi, assembly_count, device):
    """Lays sequences [lo, hi) out as this rank's device text (ids stay the job-wide ones)."""
    part = loaded[lo:hi]
    n = len(part)
    views = (_capi.SeqView * n)()
    keep = []
    for i, q in enumerate(part):
        b = bytes(q["fwd"]); keep.append(b)
        views[i].fwd, views[i].length, views[i].id = b, q["length"], q["id"]
    n_text = lib.ac_text_size(C.c_uint32(k), views, C.c_uint32(n))
    text = np.empty(n_text, dtype=np.uint8)
    off = (C.c_uint64 * n)(); d1 = (C.c_uint16 * n)(); d2 = (C.c_uint16 * n)()
    assert lib.ac_layout_text(C.c_uint32(k), views, C.c_uint32(n), text.ctypes.data_as(C.c_void_p), off, d1, d2) == 0
    d_text = torch.from_numpy(text).to(device)
    return sharded.LocalShard(k, assembly_count, d_text, n_text, list(off), [q["length"] for q in part],
                              [q["id"] for q in part], list(d1), list(d2))


def run_case(lib_path, k, seqs, filenames, headers, comm, device, repair=True, device_index=0, gather_paths=True, direct_when_alone=False):
    """All ranks call this with the same inputs.  Returns the whole GFA text on the root, None elsewhere.
    gather_paths=False: every rank keeps the P lines of its own sequences; the test stitches the parts together."""
    lib = _capi.load_library(lib_path)
    s = O.Seqs.from_raw(k, seqs, filenames=filenames, headers=headers, repair=repair)
    loaded = s.all()
    b = slice_bounds(len(loaded), comm.world)
    lo, hi = b[comm.rank], b[comm.rank + 1]
    assert hi > lo, "fewer sequences than ranks"
    local_assemblies = max(1, len({q["filename"] for q in loaded[lo:hi]}))
    shard = local_shard(lib, k, loaded, lo, hi, local_assemblies, device)
    # (direct_when_alone=False: a world of one rank still runs every phase of the protocol — that is what these tests are for)
    g, info = sharded.sharded_build(lib, shard, comm, device_index=device_index, root=0, gather_paths=gather_paths, direct_when_alone=direct_when_alone)
    assert g.stats_post["unitigs"] == info["unitigs"]
    # the tail is partitioned: a rank runs the junctions of its own conflict components only (about 1 / world of them)
    # (whole components: a small graph whose largest component is a good part of it balances less well)
    assert info["candidates_owned"] <= info["candidates"]
    if comm.world > 1 and info["candidates"] >= 400:
        assert info["candidates_owned"] * comm.world <= 2.0 * info["candidates"] + 128, (info["candidates_owned"], info["candidates"], comm.world)
    if comm.world == 1:
        assert info["candidates_owned"] == info["candidates"]
    if gather_paths or comm.world == 1:
        if comm.rank != 0:
            return None
        gfa_g = g.gfa([q["filename"] for q in loaded], [q["header"] for q in loaded])
    else:
        mine = loaded[lo:hi]
        part = g.gfa([q["filename"] for q in mine], [q["header"] for q in mine], parts=3 if comm.rank == 0 else 2)
        parts = [None] * comm.world
        comm.dist.all_gather_object(parts, part)
        if comm.rank != 0:
            return None
        gfa_g = "".join(parts)
    gfa_o, st, _ = s.compress(k)
    assert g.kmer_count == st["kmers"]
    assert g.stats_pre == dict(unitigs=st["unitigs_pre"], links=st["links_pre"], total_length=st["length_pre"])
    assert g.stats_post == dict(unitigs=st["unitigs_post"], links=st["links_post"], total_length=st["length_post"])
    assert gfa_o == gfa_g, parity_util.first_diff(gfa_o, gfa_g)
    return gfa_g
