"""BASELINE.json's full-size configurations on the device, checked through size-independent properties (the oracle, like
the reference, needs minutes to hours at these sizes): config B = 12 x 5 Mbp and config C = 96 x 5 Mbp, k = 51.

  * decompress property (tests.rs:114-127, unitig_graph.rs:362-388): the unitig strand sequences along every path spell
    the input sequence, byte for byte;
  * every step of every path is a link, links come in reverse-complement pairs (check_links, unitig_graph.rs:752-793);
  * depth == number of path occurrences (unitig.rs:149-156), unitigs are in renumber_unitigs order (unitig_graph.rs:295-315);
  * the printed statistics are consistent (link_count, total_length, kmers.len() == 2 x sum of pre-simplification lengths);
  * config B only: GFA save -> load -> save is the identity (tests.rs:108-112) through the oracle's loader/writer, and a
    scaled-down replica of the same generator is byte-identical to the oracle (test_gpu_parity)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def build(n_assemblies, k=51, genome=5_000_000, pieces=1, device_repair=False, plasmid=None, seed=51_000):
    """pieces > 1: every replicon is cut into that many contigs (fragmented assemblies: thousands of sequences)."""
    import torch
    import bench
    from autocycler_amd import _capi, synth
    lib = _capi.load_library()
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(n_assemblies, genome=genome, plasmid=genome // 50 if plasmid is None else plasmid, seed=seed)):
        for header, s in contigs:
            cuts = np.linspace(0, len(s), pieces + 1).astype(int) if len(s) >= 200 * pieces else np.array([0, len(s)])
            for j in range(len(cuts) - 1):
                seqs.append(np.ascontiguousarray(s[cuts[j]:cuts[j + 1]])); fn.append(f"assembly_{i:04d}.fasta"); hd.append(f"{header} part={j}")
    h = bench.prepare(lib, k, seqs, fn, hd, n_assemblies, threads=32, repair=0 if device_repair else 1)
    n = lib.ac_seqs_count(h)
    views = lib.ac_seqs_views(h)
    n_text = lib.ac_text_size(C.c_uint32(k), views, C.c_uint32(n))
    text = np.empty(n_text, dtype=np.uint8)
    off = (C.c_uint64 * n)(); d1 = (C.c_uint16 * n)(); d2 = (C.c_uint16 * n)()
    assert lib.ac_layout_text(C.c_uint32(k), views, C.c_uint32(n), text.ctypes.data_as(C.c_void_p), off, d1, d2) == 0
    lens = (C.c_uint32 * n)(*[views[i].length for i in range(n)])
    ids = (C.c_uint16 * n)(*[views[i].id for i in range(n)])
    d_text = torch.from_numpy(text).to("cuda:0")
    if device_repair:
        assert lib.ac_end_repair_device(C.c_uint32(k), C.c_void_p(d_text.data_ptr()), C.c_uint64(n_text), off, lens, d1, d2, C.c_uint32(n),
                                        C.c_int(0), None, None) == 0, lib.ac_last_error()
    g = C.c_void_p()
    rc = lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(n_assemblies), C.c_void_p(d_text.data_ptr()), C.c_uint64(n_text),
                                      off, lens, ids, d1, d2, C.c_uint32(n), C.c_int(0), C.byref(g))
    assert rc == 0, lib.ac_last_error()
    lib.ac_seqs_free(h)
    build.last = dict(d_text=d_text, n_text=n_text, off=off, lens=lens)      # (the job's text stays on the device for verify_on_device)
    return _capi.Graph(lib, g, n), seqs, fn, hd


def verify_on_device(g, seqs):
    """ac_verify_graph_device (round 5): the property list of check_properties as device kernels behind the ABI, against the
    text the graph was built from (end repair only rewrites padding dots; the bases between them are the inputs)."""
    import time
    j = build.last
    t0 = time.time()
    rep = g.verify_device(j["d_text"].data_ptr(), j["n_text"], j["off"], j["lens"])
    assert rep["failed"] == 0, rep
    # round 6: the order-sensitive guarantees all ran — L-line order with seed numbers, maximality, expand_repeats' fixed point
    # (unitig_graph.rs:192-223, 333-350; graph_simplification.rs:26-86): "the reference's graph", not only "lossless and consistent"
    assert rep["checks"] == 15, rep
    assert rep["bases_checked"] == sum(len(s) for s in seqs) and rep["unitigs"] == g.unitig_count
    print(f"ac_verify_graph_device: {rep['unitigs']} unitigs, {rep['links']} links, {rep['path_entries']} path entries, {rep['bases_checked']} bases in "
          f"{rep['seconds'] * 1e3:.1f} ms on the device ({(time.time() - t0) * 1e3:.1f} ms with the call)")
    return rep["unitigs"]


def check_properties(g, seqs, k):
    U = g.unitig_count
    ulen = np.empty(U, dtype=np.int64); depth = np.empty(U, dtype=np.float64)
    pieces = []
    for i in range(U):
        s, d = g.unitig(i)
        pieces.append(s); ulen[i] = len(s); depth[i] = d
    useq = np.frombuffer(b"".join(pieces), dtype=np.uint8)
    uoff = np.cumsum(ulen) - ulen
    assert (ulen > 0).all()
    # renumber_unitigs order: length descending, then sequence ascending (then depth descending)
    assert (np.diff(ulen) <= 0).all()
    for i in np.nonzero(np.diff(ulen) == 0)[0][:20000]:
        a, b = pieces[i], pieces[i + 1]
        assert a < b or (a == b and depth[i] >= depth[i + 1]), i
    # links: reverse-complement pairs; one-way count as link_count() defines it
    links = g.links()
    la = np.array([(a if af else -a, b if bf else -b) for a, af, b, bf in links], dtype=np.int64)
    code = lambda x, y: (x + (1 << 31)) * (1 << 32) + (y + (1 << 31))
    lset = np.unique(code(la[:, 0], la[:, 1]))
    assert len(lset) == len(la)                                     # no duplicate links
    assert np.isin(code(-la[:, 1], -la[:, 0]), lset).all()          # a -> b  <=>  -b -> -a
    self_mirror = int((la[:, 0] == -la[:, 1]).sum())
    assert g.stats_post["links"] == (len(la) + self_mirror) // 2
    assert g.stats_post["total_length"] == int(ulen.sum())
    assert g.stats_pre["unitigs"] == g.stats_post["unitigs"] == U
    assert g.kmer_count == 2 * g.stats_pre["total_length"]          # trimmed length == number of k-mers (unitig.rs:158-166)
    # paths: spell the inputs, follow links, define the depths
    occ = np.zeros(U, dtype=np.int64)
    for s, orig in enumerate(seqs):
        p = np.asarray(g.path(s), dtype=np.int64)
        idx = np.abs(p) - 1
        fwd = p > 0
        occ += np.bincount(idx, minlength=U)
        assert np.isin(code(p[:-1], p[1:]), lset).all(), f"path {s} leaves the links"
        ln = ulen[idx]
        assert int(ln.sum()) == len(orig)
        starts = np.cumsum(ln) - ln
        within = np.arange(len(orig), dtype=np.int64) - np.repeat(starts, ln)
        f = np.repeat(fwd, ln)
        src = np.repeat(uoff[idx], ln) + np.where(f, within, np.repeat(ln, ln) - 1 - within)
        out = useq[src]
        out = np.where(f, out, COMP[out])
        assert np.array_equal(out, orig), f"sequence {s} is not reproduced by its path"
    assert np.array_equal(occ.astype(np.float64), depth)
    return U


def test_config_b_12_assemblies():
    import oracle_lib as O
    g, seqs, fn, hd = build(12)
    U = check_properties(g, seqs, 51)
    assert U > 1000 and verify_on_device(g, seqs) == U      # the numpy restatement of the properties and the device verifier agree
    gfa = g.gfa(fn, hd)
    assert O.gfa_resave(gfa) == gfa          # save -> load -> save identity (tests.rs:108-112)
    dec = O.decompress(gfa)                  # decompress.rs through the oracle: (filename, header, sequence) per contig
    assert len(dec) == len(seqs)
    assert [d[2].encode() for d in dec] == [s.tobytes() for s in seqs]


def test_config_c_96_assemblies():
    import time
    g, seqs, fn, hd = build(96)
    U = check_properties(g, seqs, 51)
    assert U > 10000 and verify_on_device(g, seqs) == U
    assert g.stats_post["total_length"] < g.stats_pre["total_length"]
    # pairwise_contig_distances (cluster.rs:132-157) on the device against a numpy restatement, every 7th row
    t0 = time.time()
    d = g.pairwise_distances()
    dt = time.time() - t0
    print(f"pairwise distances of {len(seqs)} sequences on the device: {dt * 1e3:.1f} ms")
    ulen = np.array([len(g.unitig(i)[0]) for i in range(U)], dtype=np.int64)
    sets = [np.unique(np.abs(np.asarray(g.path(s), dtype=np.int64)) - 1) for s in range(len(seqs))]
    for a in range(0, len(seqs), 7):
        a_len = float(ulen[sets[a]].sum())
        for b in range(len(seqs)):
            ab = float(ulen[np.intersect1d(sets[a], sets[b], assume_unique=True)].sum())
            assert d[a][b] == 1.0 - ab / a_len, (a, b)
        assert d[a][a] == 0.0


def test_fragmented_assemblies_many_sequences():
    # 30 assemblies of a 1 Mbp genome cut into 400 contigs each (+ plasmid pieces): > 12 000 sequences, every one with two end
    # repair patterns on the device, through the build and the size-independent properties
    g, seqs, fn, hd = build(30, genome=1_000_000, pieces=400, device_repair=True)
    assert len(seqs) > 12_000
    U = check_properties(g, seqs, 51)
    assert verify_on_device(g, seqs) == U
    assert U > 1000


def test_config_d_prime_k101():
    # scaled replica of BASELINE config D (24 x 100 Mbp, k = 101): 24 x 10 Mbp, four-word keys, device end repair
    g, seqs, fn, hd = build(24, k=101, genome=10_000_000, device_repair=True)
    U = check_properties(g, seqs, 101)
    assert U > 10000


def test_config_d_full_size_k101():
    # BASELINE configs[3] at FULL size on one MI355X: 24 x 100 Mbp, k = 101 (2.4 G bases, ~126 M distinct canonical k-mers).  The
    # oracle cannot hold it, so the device result is checked through the size-independent properties.
    g, seqs, fn, hd = build(24, k=101, genome=100_000_000, device_repair=True, plasmid=0, seed=101_000)
    assert sum(len(s) for s in seqs) > 2_390_000_000
    U = verify_on_device(g, seqs)      # (round 5: the device verifier behind the ABI instead of the numpy restatement — 2.4 G bases spelled back on the device)
    assert U > 100_000
    assert g.kmer_count > 240_000_000


def test_config_e_full_size_k51():
    """BASELINE configs[4] at FULL size on one MI355X: 25 species x 40 strains (1 % apart) x ~5 Mbp = 1000 assemblies, 5.08 G bp, k = 51
    (`synth.make_mixed_species(25, 40)`): ~2.1 G distinct canonical k-mers, ~8 x 10^7 unitigs, > 10^9 path entries — more text
    positions than a 32-bit index holds and more threads than one launch may have.  Neither the oracle nor the reference can hold
    it (8 KB of heap per distinct k-mer at N = 1000), so the device result is checked through the size-independent
    properties, evaluated chunk by chunk on the device (tests/fullsize_e.py; the E' and E2 replicas of the same generator are compared
    with the oracle's md5 below)."""
    import torch
    import fullsize_e
    from autocycler_amd import _capi
    lib = _capi.load_library()
    job = fullsize_e.make_job(25, 40)
    assert job["bases"] > 5_000_000_000 and job["n_assemblies"] == 1000
    d_text = torch.from_numpy(job["text"]).to("cuda:0")
    g, times, _ = fullsize_e.build_device(lib, job, d_text.data_ptr(), repair=True, builds=1)
    lib.ac_release_memory()
    # round 5: ac_verify_graph_device — the decompress identity, check_links, depth, renumber order and the statistics as device kernels
    # behind the ABI (tests/fullsize_e.py::check_on_device is the torch restatement rounds 3-4 used; AC_TEST_TORCH_CHECK=1 runs it as well)
    r = g.verify_device(d_text.data_ptr(), job["n_text"], job["off"], job["lens"])
    assert r["failed"] == 0 and r["bases_checked"] == job["bases"], r
    assert r["checks"] == 15, r      # round 6: L-line order (with seed numbers), maximality and expand_repeats' fixed point ran on all 81.9 M unitigs
    tm = g.timings()
    # round 6: above 8 M unitigs the path entries reach the host as stretches of consecutive text-order numbers — at least 4x fewer records than entries
    assert 0 < tm["path_stretches"] * 4 <= tm["n_path_entries"], (tm["path_stretches"], tm["n_path_entries"])
    if os.environ.get("AC_TEST_TORCH_CHECK"):
        lib.ac_release_memory()
        fullsize_e.check_on_device(g, job, d_text, log=print)
    print("config E full size:", times, g.stats_post, "kmers", g.kmer_count, r)
    assert r["unitigs"] > 50_000_000 and r["path_entries"] > 1_000_000_000
    assert g.kmer_count > 4_000_000_000
    g.close()
    lib.ac_release_memory()


@pytest.mark.parametrize("golden_name", ["configC_k51", "configDprime_k101", "configB_k51", "configEprime_k51", "configE2_k51", "configDprime_k201"])
def test_gfa_digest_equals_the_oracle(golden_name):
    """Bit-exact parity at full size: the GFA built on the device (end repair on the device text, build, GFA text — the flow of
    tools/ab_knobs.py, run here as the same torch-free process) has the md5 the ORACLE produced for the same FASTA files on the CPU
    (tests/golden/*.json, made by tests/golden/make_configC_golden.sh: 26 and 12 minutes of the restated reference path), and the
    same unitig count.  configC_k51 = BASELINE configs[2] (96 x ~5 Mbp, k = 51: the benchmark workload); configDprime_k101 = the
    scaled replica of configs[3] (24 x ~10 Mbp, k = 101: four-word keys); configB_k51 = BASELINE configs[1] (12 x ~5 Mbp, k = 51);
    configEprime_k51 = the scaled replica of configs[4] (mixed species: 5 species x 20 strains 1 % apart x ~1 Mbp = 100 assemblies,
    1.73 M unitigs, a 310 MB GFA; tests/golden/make_golden.py, 32 minutes of the oracle); configE2_k51 = the same model with ~2 Mbp
    genomes (203 M bp, 3.46 M unitigs; 69 minutes and 38 GB of the oracle: about the largest mixed-species input it can hold here);
    configDprime_k201 = D' again at k = 201 (round 6: SEVEN-word keys, i.e. the 8-word instantiation of every kernel — the reference's
    --kmer range goes to 501, compress.rs:56-60)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    golden = json.loads((root / "tests" / "golden" / (golden_name + ".json")).read_text())
    out = subprocess.run([sys.executable, str(root / "tools" / "ab_knobs.py"), "--variants", "base", "--steps", "1", "--workload", golden_name],
                         env={**os.environ, "AC_NO_TORCH": "1"}, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    base = [r for r in rows if r.get("variant") == "base"]
    assert base and "error" not in base[0], rows
    assert base[0]["unitigs"] == golden["post"]["unitigs"]
    assert base[0]["gfa_md5"] == golden["gfa_md5"]
    # the library's cost model picks the copying path walk for the 96 similar assemblies of config C and for nothing else here
    assert (base[0]["path_runs_copied"] > 0) == (golden_name == "configC_k51")


def test_config_c_digest_with_the_plain_path_walk():
    """Config C with the copying walk switched off (AC_PATH_COPY=0; the cost model switches it on for this input): the same digest."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    golden = json.loads((root / "tests" / "golden" / "configC_k51.json").read_text())
    out = subprocess.run([sys.executable, str(root / "tools" / "ab_knobs.py"), "--variants", "AC_PATH_COPY=0;AC_PATH_COPY=1,AC_POS_CAP=0", "--steps", "1", "--workload", "configC_k51"],
                         env={**os.environ, "AC_NO_TORCH": "1"}, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{") and '"variant"' in l]
    assert len(rows) == 2 and all("error" not in r for r in rows), rows
    assert all(r["gfa_md5"] == golden["gfa_md5"] for r in rows)
    assert rows[0]["path_runs_copied"] == 0 and rows[1]["path_runs_copied"] > 0


@pytest.mark.parametrize("variants", ["base", "AC_PATH_COPY=0", "AC_UPLOAD_OVERLAP=0", "AC_HOST_PACK=0", "AC_UPLOAD_THREADS=5", "AC_UPLOAD_DIRECT=0", "AC_UPLOAD_DIRECT=0,AC_UPLOAD_THREADS=5",
                                      "AC_UPLOAD_DIRECT=0,AC_UPLOAD_SLOTS=2", "AC_UPLOAD_DIRECT=0,AC_UPLOAD_SLOTS=1,AC_UPLOAD_OVERLAP=0", "AC_UPLOAD_DIRECT=1,AC_UPLOAD_THREADS=64"])
def test_host_entry_full_size_digest(variants):
    """The HOST entry (ac_compress_build from pageable per-sequence buffers: host-side 2-bit pack by background threads — straight into
    device memory where it is host-visible, AC_UPLOAD_DIRECT=0: through the pinned ring in 16 MB copies —, the insert issued piece by
    piece as the chunks land) on the whole config C — 487 MB of text, eight 64 MB chunks — gives the oracle's GFA digest: also with the
    overlap off, with the byte upload + device pack, with an odd number of packing threads, and, through the ring, with so few staging
    slots that every chunk has to wait for an earlier one's slot."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    golden = json.loads((root / "tests" / "golden" / "configC_k51.json").read_text())
    out = subprocess.run([sys.executable, str(root / "tools" / "ab_knobs.py"), "--variants", variants, "--steps", "1", "--workload", "configC_k51", "--host-entry"],
                         env={**os.environ, "AC_NO_TORCH": "1"}, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{") and '"variant"' in l]
    assert rows and "error" not in rows[0], rows
    assert rows[0]["gfa_md5"] == golden["gfa_md5"] and rows[0]["unitigs"] == golden["post"]["unitigs"]
    assert rows[0]["upload_device_ms"] > 0
