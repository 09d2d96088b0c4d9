"""BASELINE.json configs[4] (mixed species: SPECIES x STRAINS assemblies of ~5 Mbp, strains 1 % apart, k = 51) on ONE MI355X:
generation, text layout, end repair + build through the C ABI, and the size-independent properties of test_gpu_fullsize.py
evaluated in chunks with torch ON THE DEVICE (a 5 G bp job has ~8 x 10^7 unitigs and > 10^9 path entries: per-unitig ctypes calls
and whole-array numpy temporaries do not scale to it).  torch is only the checker's array engine here — nothing below is product
code.  Used by tests/test_gpu_fullsize.py::test_config_e_full_size_k51 and tools/config_e.py.

Properties (same list as test_gpu_fullsize.check_properties):
  * every path spells its input sequence byte for byte (decompress identity: tests.rs:114-127, unitig_graph.rs:362-388);
  * every step of every path is a link; links are unique and come in reverse-complement pairs (check_links, unitig_graph.rs:752-793);
  * depth == number of path occurrences (unitig.rs:149-156); unitigs are in renumber_unitigs order (unitig_graph.rs:295-315);
  * the printed statistics are consistent (link_count, total_length, kmers.len() == 2 x pre-simplification total length)."""
import ctypes as C
import time

import numpy as np


def make_job(species, strains, k=51, genome=5_000_000, plasmid=100_000, workers=None, seed=77_000):
    """-> dict: text (uint8, the layout of include/autocycler_hip.h: '$' + padded sequences each followed by '$'), off / lens / ids /
    d1 / d2 (ctypes arrays), seq_start (index of each sequence's first BASE in text), n_assemblies, bases."""
    from autocycler_amd import synth
    t0 = time.time()
    asm = synth.make_mixed_species_parallel(species, strains, workers=workers, genome=genome, plasmid=plasmid, strain_div=1e-2, sub=1e-4,
                                            indel=1e-5, seed=seed)
    seqs = [s for contigs in asm for _, s in contigs]
    t_gen = time.time() - t0
    n = len(seqs)
    h = k // 2
    n_text = 1 + sum(len(s) + 2 * h + 1 for s in seqs)
    text = np.empty(n_text, dtype=np.uint8)
    text[0] = ord("$")
    off = (C.c_uint64 * n)(); lens = (C.c_uint32 * n)(); ids = (C.c_uint16 * n)(); d1 = (C.c_uint16 * n)(); d2 = (C.c_uint16 * n)()
    seq_start = np.empty(n, dtype=np.int64)
    p = 1
    for i, s in enumerate(seqs):
        L = len(s)
        off[i] = p; lens[i] = L; ids[i] = i + 1; d1[i] = h; d2[i] = h
        text[p:p + h] = ord("."); text[p + h:p + h + L] = s; text[p + h + L:p + 2 * h + L] = ord("."); text[p + 2 * h + L] = ord("$")
        seq_start[i] = p + h
        p += L + 2 * h + 1
    assert p == n_text
    return dict(k=k, text=text, n_text=n_text, off=off, lens=lens, ids=ids, d1=d1, d2=d2, n=n, seq_start=seq_start, n_assemblies=species * strains,
                bases=int(sum(len(s) for s in seqs)), generate_s=t_gen, layout_s=time.time() - t0 - t_gen)


def build_device(lib, job, d_text, repair=True, builds=1):
    """End repair (once) + `builds` builds of the device-resident text; returns (Graph of the last build, seconds per build, repair s)."""
    from autocycler_amd import _capi
    k, n = job["k"], job["n"]
    secs = C.c_double(0)
    if repair:
        rc = lib.ac_end_repair_device(C.c_uint32(k), C.c_void_p(d_text), C.c_uint64(job["n_text"]), job["off"], job["lens"], job["d1"], job["d2"],
                                      C.c_uint32(n), C.c_int(0), C.byref(secs), None)
        assert rc == 0, lib.ac_last_error()
    times, g = [], None
    for _ in range(builds):
        if g is not None:
            g.close()
        hg = C.c_void_p()
        t1 = time.perf_counter()
        rc = lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(job["n_assemblies"]), C.c_void_p(d_text), C.c_uint64(job["n_text"]), job["off"],
                                          job["lens"], job["ids"], job["d1"], job["d2"], C.c_uint32(n), C.c_int(0), C.byref(hg))
        assert rc == 0, lib.ac_last_error()
        times.append(time.perf_counter() - t1)
        g = _capi.Graph(lib, hg, n)
    return g, times, secs.value


def check_on_device(g, job, d_text_t, chunk=1 << 26, log=lambda *a: None):
    """The property list of the module docstring; d_text_t = the job's text as a torch uint8 tensor on the device (the end-repaired
    text: repair only rewrites padding dots, the bases between them are the input sequences)."""
    import torch
    dev = d_text_t.device
    t0 = time.time()
    b = g.bulk()
    U = g.unitig_count
    S = job["n"]
    ulen = torch.from_numpy(b["seq_len"].astype(np.int64)).to(dev)
    ubeg = torch.from_numpy(b["seq_begin"].astype(np.int64)).to(dev)
    depth = torch.from_numpy(np.ascontiguousarray(b["depth"])).to(dev)
    useq = torch.from_numpy(b["seq_bytes"]).to(dev)
    assert bool((ulen > 0).all())
    total_len = int(ulen.sum())
    assert g.stats_post["total_length"] == total_len
    assert g.stats_pre["unitigs"] == g.stats_post["unitigs"] == U
    assert g.kmer_count == 2 * g.stats_pre["total_length"]          # trimmed length == number of k-mers (unitig.rs:158-166)
    log(f"arrays on the device {time.time() - t0:.1f}s: U={U} total_length={total_len}")

    # --- renumber_unitigs order (unitig_graph.rs:295-315): length desc, then sequence asc, then depth desc — every adjacent pair
    assert bool((ulen[1:] <= ulen[:-1]).all())
    pair = torch.nonzero(ulen[1:] == ulen[:-1]).flatten()          # i: unitigs i and i + 1 have the same length
    n_pairs = int(pair.numel())
    undecided = pair
    o = 0
    while undecided.numel():
        L = ulen[undecided]
        live = o < L
        # (pairs whose sequences are identical over the whole length: the depth decides)
        done_eq = undecided[~live]
        if done_eq.numel():
            assert bool((depth[done_eq] >= depth[done_eq + 1]).all()), "equal sequences out of depth order"
        undecided = undecided[live]
        if not undecided.numel():
            break
        L = ulen[undecided]
        va = torch.zeros(undecided.numel(), dtype=torch.int64, device=dev); vb = torch.zeros_like(va)
        for j in range(7):                                          # 7 bytes per round, big-endian, zero beyond the end
            ok = (o + j) < L
            ia = torch.where(ok, ubeg[undecided] + o + j, torch.zeros_like(L))
            ib = torch.where(ok, ubeg[undecided + 1] + o + j, torch.zeros_like(L))
            va = va * 256 + torch.where(ok, useq[ia].to(torch.int64), torch.zeros_like(L))
            vb = vb * 256 + torch.where(ok, useq[ib].to(torch.int64), torch.zeros_like(L))
        assert bool((va <= vb).all()), "unitigs of equal length out of sequence order"
        undecided = undecided[va == vb]
        o += 7
    log(f"renumber order: {n_pairs} equal-length neighbours checked {time.time() - t0:.1f}s")

    # --- links: unique, reverse-complement pairs, one-way count as link_count() defines it (unitig_graph.rs:478-507)
    lk = b["links"]
    n_links = len(lk)
    la = torch.from_numpy(lk["a"].astype(np.int64)).to(dev)      # (signed unitig numbers: ABI 7)
    lb = torch.from_numpy(lk["b"].astype(np.int64)).to(dev)
    assert bool(((la.abs() >= 1) & (la.abs() <= U) & (lb.abs() >= 1) & (lb.abs() <= U)).all())
    code = lambda x, y: (x + (1 << 31)) * (1 << 32) + (y + (1 << 31))
    lset = torch.sort(code(la, lb)).values
    assert bool((lset[1:] != lset[:-1]).all()), "duplicate links"
    mirror = code(-lb, -la)
    pos = torch.searchsorted(lset, mirror).clamp_(max=n_links - 1)
    assert bool((lset[pos] == mirror).all()), "a link without its reverse-complement mirror"
    self_mirror = int((la == -lb).sum())
    assert g.stats_post["links"] == (n_links + self_mirror) // 2
    del mirror, pos
    log(f"links: {n_links} unique, mirrored {time.time() - t0:.1f}s")

    # --- paths: follow links, spell the inputs, define the depths
    pe = b["path_entries"]; po = b["path_off"].astype(np.int64)
    n_ent = len(pe)
    assert po[0] == 0 and po[-1] == n_ent and len(po) == S + 1
    occ = torch.zeros(U, dtype=torch.int64, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, c in zip(b"ACGT", b"TGCA"):
        comp[a] = c
    seq_start = job["seq_start"]
    lens = np.asarray(job["lens"], dtype=np.int64)
    checked_bases = 0
    s = 0
    while s < S:                                                    # a batch of whole sequences, <= chunk bases
        e = s + 1
        while e < S and int(lens[s:e + 1].sum()) <= chunk:
            e += 1
        p = torch.from_numpy(pe[po[s]:po[e]].astype(np.int64)).to(dev)
        idx = p.abs() - 1
        assert bool(((idx >= 0) & (idx < U)).all())
        occ += torch.bincount(idx, minlength=U)
        first = torch.zeros(p.numel(), dtype=torch.bool, device=dev)      # entries that start a sequence's path: no step into them
        first[torch.from_numpy(po[s:e] - po[s]).to(dev)] = True
        step_code = code(p[:-1], p[1:])[~first[1:]]
        at = torch.searchsorted(lset, step_code).clamp_(max=n_links - 1)
        assert bool((lset[at] == step_code).all()), f"a path of sequences {s}..{e - 1} leaves the links"
        ln = ulen[idx]
        seq_of_entry = torch.cumsum(first.to(torch.int64), 0) - 1         # 0-based within the batch
        per_seq = torch.zeros(e - s, dtype=torch.int64, device=dev).index_add_(0, seq_of_entry, ln)
        assert bool((per_seq == torch.from_numpy(lens[s:e]).to(dev)).all()), "a path does not add up to its sequence's length"
        # expected text position of every entry's first base: sequence start + bases spelled before it within the sequence
        csum = torch.cumsum(ln, 0) - ln
        seq_base = torch.from_numpy(seq_start[s:e]).to(dev)
        seq_csum0 = csum[torch.from_numpy(po[s:e] - po[s]).to(dev)]
        tpos = seq_base[seq_of_entry] + (csum - seq_csum0[seq_of_entry])
        n_b = int(ln.sum())
        ent = torch.repeat_interleave(torch.arange(p.numel(), device=dev), ln)
        within = torch.arange(n_b, device=dev) - csum[ent]
        fwd = (p > 0)[ent]
        src = ubeg[idx][ent] + torch.where(fwd, within, ln[ent] - 1 - within)
        out = useq[src]
        out = torch.where(fwd, out, comp[out.to(torch.int64)])
        want = d_text_t[tpos[ent] + within]
        assert bool((out == want).all()), f"sequences {s}..{e - 1} are not reproduced by their paths"
        checked_bases += n_b
        del ent, within, fwd, src, out, want
        s = e
    assert checked_bases == job["bases"]
    assert bool((occ.to(torch.float64) == depth).all()), "depth != number of path occurrences"
    log(f"paths: {n_ent} entries over {S} sequences spell {checked_bases} bases {time.time() - t0:.1f}s")
    return dict(unitigs=U, links=n_links, path_entries=n_ent, equal_length_neighbours=n_pairs, check_s=time.time() - t0)
