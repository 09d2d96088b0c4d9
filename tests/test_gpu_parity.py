"""GPU parity tests proper: libautocycler_hip.so (through the C ABI) against the oracle on the same seeded
inputs — GFA text byte-for-byte plus every printed statistic."""
import pytest

import parity_util
import seqgen
from test_oracle_kats import FIXED

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_lib():
    import autocycler_amd
    lib = autocycler_amd.load_library()       # raises HipLibraryMissing: the product has no fallback
    assert lib.ac_device_count() >= 1, "no HIP device visible"


@pytest.mark.parametrize("k", [3, 5, 9, 13, 51])
def test_fixed_seqs(k):   # tests.rs:131-148 inputs
    parity_util.check_case(k, [FIXED[c] for c in "abcde"], ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz"], list("abcde"))


@pytest.mark.parametrize("k", [5, 11, 31, 51])
@pytest.mark.parametrize("seed", range(24))
def test_adversarial_cases(k, seed):
    seqs, fn, hd = seqgen.make_case(seed, k)
    parity_util.check_case(k, seqs, fn, hd, repair=True)
    parity_util.check_case(k, seqs, fn, hd, repair=False)


@pytest.mark.parametrize("k", [27, 29, 59, 61, 91, 93, 123])
def test_key_word_boundaries(k):
    for seed in (1, 2, 6, 7):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd)


def _synth_case(n, genome, plasmid, sub, indel, seed):
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(n, genome=genome, plasmid=plasmid, sub=sub, indel=indel, seed=seed)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    return seqs, fn, hd


@pytest.mark.parametrize("k", [51, 101])
def test_synthetic_assemblies_medium(k):
    # scaled-down replica of configs B / D': 8 assemblies of a 200 kbp genome with planted repeats
    seqs, fn, hd = _synth_case(8, 200_000, 8_000, 1e-3, 1e-4, 4242)
    g, gfa, _ = parity_util.check_case(k, seqs, fn, hd)
    assert g.stats_post["unitigs"] > 100


@pytest.mark.parametrize("n", [20, 70, 150])
@pytest.mark.parametrize("k", [51, 101])
def test_renumber_tie_groups(n, k):
    from test_emu_parity import shared_prefix_case
    seqs, fn, hd = shared_prefix_case(n, k)
    parity_util.check_case(k, seqs, fn, hd, repair=False)
    parity_util.check_case(k, seqs, fn, hd, repair=True)


def test_synthetic_many_path_entries():
    # > 262 144 path entries: several renumbering chunks, each copied to the host while the next is renumbered
    seqs, fn, hd = _synth_case(12, 300_000, 10_000, 5e-3, 2e-4, 777)
    g, gfa, _ = parity_util.check_case(51, seqs, fn, hd)
    assert g.timings()["n_path_entries"] > 262_144


@pytest.mark.parametrize("k", [5, 21, 51, 101])
def test_end_repair_device(k):
    # sequence_end_repair on the device text against the oracle's restatement of compress.rs:202-270
    import autocycler_amd
    import repair_util
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        repair_util.check_repair(autocycler_amd.LIB_PATH, k, seqs, fn, hd, device="cuda:0")


def test_end_repair_then_build_on_the_same_device_text():
    # the whole device flow: padded text -> ac_end_repair_device (in place) -> ac_compress_build_device -> GFA == oracle
    import ctypes as C
    import autocycler_amd
    import oracle_lib as O
    import repair_util
    from autocycler_amd import _capi
    k = 51
    seqs, fn, hd = _synth_case(8, 200_000, 8_000, 1e-3, 1e-4, 4242)
    d_text, off, lens, d1, d2, n_matches, _ = repair_util.check_repair(autocycler_amd.LIB_PATH, k, seqs, fn, hd, device="cuda:0")
    assert n_matches >= 2 * len(seqs)
    lib = _capi.load_library()
    n = len(seqs)
    ids = (C.c_uint16 * n)(*range(1, n + 1))
    g = C.c_void_p()
    rc = lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(8), C.c_void_p(d_text.data_ptr()), C.c_uint64(d_text.numel()), off, lens, ids,
                                      d1, d2, C.c_uint32(n), C.c_int(0), C.byref(g))
    assert rc == 0, lib.ac_last_error()
    graph = _capi.Graph(lib, g, n)
    s = O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd, repair=True)
    gfa_o, _, _ = s.compress(k)
    loaded = s.all()
    assert graph.gfa([q["filename"] for q in loaded], [q["header"] for q in loaded]) == gfa_o


@pytest.mark.parametrize("k", [125, 201, 251, 253, 501])
def test_wide_keys(k):   # keys of 8 and 16 words (--kmer up to 501, compress.rs:56-60)
    for seed in (1, 2, 6, 7):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd)


@pytest.mark.parametrize("k", [11, 51])
def test_pairwise_distances(k):   # cluster.rs:132-157 on the graph just built
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd, distances=True)
    seqs, fn, hd = _synth_case(8, 200_000, 8_000, 1e-3, 1e-4, 4242)
    parity_util.check_case(k, seqs, fn, hd, distances=True)


def test_random_synthetic_sets():
    # randomised assembly sets (2-14 assemblies, 8-50 kbp, substitution rates up to 2 %, indels, several k) through the build,
    # the sharded path at world size 1 and the pairwise distances
    import random
    import torch
    import autocycler_amd
    import sharded_util
    from autocycler_amd import sharded, synth
    dev = torch.device("cuda", 0)
    for seed in range(24):
        r = random.Random(seed)
        k = r.choice([21, 31, 51, 51, 51, 77, 101, 151])
        na = r.randint(2, 14); genome = r.choice([8000, 20000, 50000]); plasmid = r.choice([0, 1500, 4000])
        sub = r.choice([0, 1e-4, 1e-3, 5e-3, 2e-2]); indel = r.choice([0, 1e-4, 2e-3])
        seqs, fn, hd = [], [], []
        for i, contigs in enumerate(synth.make_assemblies(na, genome=genome, plasmid=plasmid, sub=sub, indel=indel, seed=1000 + seed)):
            for header, s in contigs:
                seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
        parity_util.check_case(k, seqs, fn, hd, distances=(seed % 2 == 0))
        if seed % 3 == 0:
            sharded_util.run_case(autocycler_amd.LIB_PATH, k, seqs, fn, hd, sharded.Comm(dev), dev)


def test_high_diversity_table_growth():
    # see tests/test_emu_parity.py: the k-mer table outgrows its capacity hint, twice
    import time
    seqs, fn, hd = [], [], []
    from autocycler_amd import synth
    for i, contigs in enumerate(synth.make_assemblies(30, genome=100_000, plasmid=0, sub=1e-2, indel=1e-4, seed=900_000)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"strain_{i:03d}.fasta"); hd.append(header)
    t0 = time.time()
    g, _, _ = parity_util.check_case(51, seqs, fn, hd)
    assert g.timings()["n_distinct"] > 1_000_000
    assert time.time() - t0 < 120


@pytest.mark.parametrize("knobs", [{"AC_TABLE_SHIFT": "0"}, {"AC_TABLE_SHIFT": "2"}, {"AC_MINKEY_VARIANT": "0"}, {"AC_MINKEY_VARIANT": "1"}, {"AC_MINKEY_VARIANT": "2"}, {"AC_MINKEY_VARIANT": "2", "AC_MINKEY_PREFIX_BASES": "2"}, {"AC_PATH_CHUNK": "64"}, {"AC_PATH_FILTER": "0"}, {"AC_EXPAND_REWRITE_ALWAYS": "1"}, {"AC_EXPAND_LEVEL_TABLE": "1"}, {"AC_SEED_RADIX_LIMIT": "0"}, {"AC_SEQ_WRITER": "0"}, {"AC_SEQ_WRITER": "1"}, {"AC_HOST_PACK": "0"}, {"AC_UPLOAD_OVERLAP": "0"}, {"AC_SEED_PREFIX_SORT": "0"}, {"AC_SEED_PREFIX_BITS": "3"}, {"AC_SEED_PREFIX_BITS": "12"}, {"AC_POS_CAP": "0"}, {"AC_POS_CAP": "3"}, {"AC_POS_CAP": "200", "AC_PATH_COPY": "1"}, {"AC_PATH_COPY": "1"}, {"AC_PATH_COPY": "1", "AC_RUN_PIECE": "150"}, {"AC_PATH_COPY": "1", "AC_PATH_FILTER": "0"}, {"AC_SEED_PREFIX_BITS": "3", "AC_SEED_MAX_GROUP": "2"}, {"AC_SEED_PREFIX_BITS": "6", "AC_SEED_MAX_GROUP": "1", "AC_SEED_RADIX_LIMIT": "1000000000"}, {"AC_SEED_PREFIX_SORT": "0", "AC_SEED_RADIX_LIMIT": "0"}, {"AC_RENUM_TWO_PASS": "1"}, {"AC_RENUM_MAX_GROUP": "1"}, {"AC_SORT_CHECKS": "1"}, {"AC_SORT_CHECKS": "1", "AC_RENUM_MAX_GROUP": "1"}, {"AC_DEGREE_FLAGS": "0"}, {"AC_DEGREE_REGION_CAP": "1"}, {"AC_DEGREE_REGION_CAP": "3"}, {"AC_UPLOAD_THREADS": "3"}, {"AC_UPLOAD_DIRECT": "0"}, {"AC_UPLOAD_DIRECT": "0", "AC_UPLOAD_THREADS": "3"}, {"AC_UPLOAD_DIRECT": "1"},
                                   {"AC_REMAP_BLOCK": "128"}, {"AC_HOST_REMAP": "1"}, {"AC_HOST_REMAP": "1", "AC_UPLOAD_THREADS": "3"}, {"AC_HOST_REMAP": "0"}, {"AC_HOST_REMAP": "2"}, {"AC_HOST_REMAP": "2", "AC_UPLOAD_THREADS": "2"}, {"AC_HOST_REMAP": "2", "AC_REMAP_BLOCK": "128", "AC_STRETCH_DEVICE_SHARE": "50"}, {"AC_HOST_REMAP": "2", "AC_STRETCH_DEVICE_SHARE": "0"}, {"AC_SEQ_CODES": "2"}, {"AC_SEQ_CODES": "0"}, {"AC_LATE_COPIES": "2"}, {"AC_LATE_COPIES": "2", "AC_HOST_REMAP": "2", "AC_REMAP_BLOCK": "128", "AC_SEQ_CODES": "2"}, {"AC_LATE_COPIES": "2", "AC_HOST_REMAP": "0"}, {"AC_SEQ_CODES": "2", "AC_PACK_SCALAR": "1"}, {"AC_INSERT_ADAPT": "0", "AC_INSERT_GROWTH": "4"},
                                   {"AC_INSERT_CHUNK": "256", "AC_INSERT_WAVES": "1024"}],
                         ids=lambda d: ",".join(f"{a}={b}" for a, b in d.items()))
def test_tuning_knobs_do_not_change_the_result(monkeypatch, knobs):
    # the knobs of graph_build.hip (read on every build) only move work around: same GFA as the oracle under each of them
    for a, b in knobs.items():
        monkeypatch.setenv(a, b)
    for k, seed in ((11, 7), (51, 13)):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd)
    seqs, fn, hd = _synth_case(6, 60_000, 3_000, 1e-3, 1e-4, 99)
    parity_util.check_case(51, seqs, fn, hd)


@pytest.mark.parametrize("copy", ["0", "1"])
def test_position_bound_and_the_repeat_with_exact_positions(monkeypatch, copy):
    # smallest positions beyond AC_POS_CAP are a lower bound; a common sequence longer than the bound repeats the build with exact ones
    monkeypatch.setenv("AC_PATH_COPY", copy)
    retried = 0
    for cap in ("1", "7", "40"):
        monkeypatch.setenv("AC_POS_CAP", cap)
        for k, seed in ((11, 7), (51, 13), (51, 21)):
            seqs, fn, hd = seqgen.make_case(seed, k)
            g, _, _ = parity_util.check_case(k, seqs, fn, hd)
            retried += g.timings()["position_retries"]
        seqs, fn, hd = _synth_case(6, 30_000, 1_500, 1e-3, 1e-4, 77)
        g, _, _ = parity_util.check_case(51, seqs, fn, hd)
        retried += g.timings()["position_retries"]
    assert retried > 0
    monkeypatch.delenv("AC_POS_CAP")
    g, _, _ = parity_util.check_case(51, seqs, fn, hd)
    assert g.timings()["position_retries"] == 0


@pytest.mark.parametrize("piece", ["0", "700"])
def test_copying_path_walk_on_a_redundant_text(monkeypatch, piece):
    # K10c (AC_PATH_COPY=1, opt-in): most of ten similar assemblies lies in followed runs, whose path entries are copied, not walked
    monkeypatch.setenv("AC_PATH_COPY", "1")
    if piece != "0":
        monkeypatch.setenv("AC_RUN_PIECE", piece)
    for rep in range(3):      # (which runs the insert follows depends on the order its wavefronts ran in)
        seqs, fn, hd = _synth_case(10, 80_000, 2_000, 2e-4, 2e-5, 2024 + rep)
        g, _, _ = parity_util.check_case(51, seqs, fn, hd)
        tm = g.timings()
        assert tm["path_runs_copied"] > 0 and 0 < tm["path_entries_walked"] < tm["n_path_entries"]      # (the first two assemblies are walked whatever happens)


@pytest.mark.parametrize("adapt", ["1", "0"])
def test_insert_phase_schedule_on_a_redundant_text(monkeypatch, adapt):
    # third phase reached: the claim counters of the first two send the rest of a redundant text in one launch
    monkeypatch.setenv("AC_INSERT_ADAPT", adapt)
    seqs, fn, hd = _synth_case(10, 80_000, 2_000, 2e-4, 2e-5, 2024)
    g, _, _ = parity_util.check_case(51, seqs, fn, hd)
    assert g.timings()["insert_launches"] == (3 if adapt == "1" else 5)
