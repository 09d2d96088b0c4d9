"""CPU parity of the kernels' logic and the host tail: the serial emulation of the HIP sources
(-DAC_EMU, built by the test-suite only) against the oracle, byte-for-byte on the GFA text and on every
statistic the reference prints (compress.rs:152,165,177).  The same cases run on the real device in
test_gpu_parity.py."""
import pytest

import emu_lib
import parity_util
import seqgen
from test_oracle_kats import FIXED


@pytest.fixture(scope="module")
def emu():
    return emu_lib.emu_path()


@pytest.mark.parametrize("k", [3, 5, 9, 13, 51])
def test_fixed_seqs(emu, k):   # the reference's own five sequences (tests.rs:133-142)
    parity_util.check_case(k, [FIXED[c] for c in "abcde"], ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz"],
                           list("abcde"), lib_path=emu)


@pytest.mark.parametrize("k", [3, 5, 7, 11, 21, 31, 51])
@pytest.mark.parametrize("seed", range(24))
def test_adversarial_cases(emu, k, seed):
    seqs, fn, hd = seqgen.make_case(seed, k)
    parity_util.check_case(k, seqs, fn, hd, lib_path=emu, repair=True)
    parity_util.check_case(k, seqs, fn, hd, lib_path=emu, repair=False)   # every end keeps its dots


@pytest.mark.parametrize("k", [27, 29, 59, 61, 91, 93, 123])
def test_key_word_boundaries(emu, k):   # k around the 1/2/3/4-word key boundaries
    for seed in (1, 2, 6, 7):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd, lib_path=emu)


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("k", [5, 11, 31, 51])
def test_scheduling_independence(emu, monkeypatch, order, k):
    # The emulation visits the logical threads of every launch in descending / pseudo-random order: the
    # run-following insert, the atomics and the two-pass path walk must not depend on scheduling.
    monkeypatch.setenv("AC_EMU_ORDER", str(order))
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd, lib_path=emu, repair=(seed % 2 == 0))


@pytest.mark.parametrize("order", [0, 2])
def test_synthetic_assemblies_emu(emu, monkeypatch, order):
    # long shared runs with sparse variants: the regime the run-following insert and the link-walking path
    # kernel are built for (scaled-down BASELINE config B)
    from autocycler_amd import synth
    monkeypatch.setenv("AC_EMU_ORDER", str(order))
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(6, genome=60_000, plasmid=3_000, sub=1e-3, indel=1e-4, seed=99)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    tm = g.timings()
    assert tm["insert_real"] < tm["insert_positions"] // 2     # most positions were run-followed, not inserted


def test_accessors(emu):
    seqs, fn, hd = seqgen.make_case(2, 9)
    g, gfa, loaded = parity_util.check_case(9, seqs, fn, hd, lib_path=emu)
    lines = gfa.splitlines()
    s_lines = [l.split("\t") for l in lines if l[0] == "S"]
    assert g.unitig_count == len(s_lines)
    for i, parts in enumerate(s_lines):
        seq, depth = g.unitig(i)
        assert seq.decode() == parts[2] and f"DP:f:{depth:.2f}" == parts[3]
    l_lines = [tuple(l.split("\t")[1:5]) for l in lines if l[0] == "L"]
    assert [(str(a), "+" if af else "-", str(b), "+" if bf else "-") for a, af, b, bf in g.links()] == l_lines
    p_lines = [l.split("\t") for l in lines if l[0] == "P"]
    for i, parts in enumerate(p_lines):
        assert ",".join(f"{abs(v)}{'+' if v > 0 else '-'}" for v in g.path(i)) == parts[2]
    # positions as from_gfa_lines rebuilds them: every sequence start appears exactly once at pos 0, forward strand
    starts = 0
    for i in range(g.unitig_count):
        for fwd in (True, False):
            for sid, strand, pos in g.positions(i, fwd):
                starts += (strand and pos == 0)
    assert starts == len(loaded)


def test_errors(emu):
    from autocycler_amd import AutocyclerError, compress_build
    with pytest.raises(AutocyclerError, match="odd"):
        compress_build(10, 1, [(b"." * 5 + b"ACGTACGTACGT" + b"." * 4, 12, 1)], lib_path=emu)
    with pytest.raises(AutocyclerError, match="no sequences"):
        compress_build(9, 1, [], lib_path=emu)
    with pytest.raises(AutocyclerError, match="not supported"):      # above the reference's own limit (compress.rs:56-60)
        compress_build(503, 1, [(b"." * 251 + b"A" * 600 + b"." * 251, 600, 1)], lib_path=emu)


def shared_prefix_case(n, k, seed=3):
    """n contigs that all start with the same 40 bases and are otherwise random: every contig is one unitig, all of the
    same length and with the same 32-base prefix — the tie groups of the radix-key renumbering (insertion sort up to
    64 members, comparator merge sort beyond)."""
    import random
    r = random.Random(seed)
    z = seqgen.rand_seq(r, 40)
    seqs = [z + seqgen.rand_seq(r, 2 * k + 8) for _ in range(n)]
    seqs += [seqs[0][:k + 30] + seqgen.rand_seq(r, 5), seqs[1]]      # shared k-mers and an exact duplicate as well
    return seqs, [f"a{i % 7}.fasta" for i in range(len(seqs))], [f"c{i}" for i in range(len(seqs))]


@pytest.mark.parametrize("n", [3, 20, 70, 150])
@pytest.mark.parametrize("k", [51, 101])
def test_renumber_tie_groups(emu, n, k):
    seqs, fn, hd = shared_prefix_case(n, k)
    parity_util.check_case(k, seqs, fn, hd, lib_path=emu, repair=False)
    parity_util.check_case(k, seqs, fn, hd, lib_path=emu, repair=True)


@pytest.mark.parametrize("k", [125, 151, 201, 251, 253, 301, 501])
def test_wide_keys(emu, k):   # keys of 8 and 16 words: the reference allows --kmer up to 501 (compress.rs:56-60)
    for seed in (1, 2, 6, 7, 13):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd, lib_path=emu)
    seqs, fn, hd = shared_prefix_case(12, k)
    parity_util.check_case(k, seqs, fn, hd, lib_path=emu, repair=False)


@pytest.mark.parametrize("k", [5, 11, 51])
def test_pairwise_distances(emu, k):   # cluster.rs:132-157 on the graph just built
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd, lib_path=emu, distances=True)


def test_high_diversity_table_growth(emu):
    # 30 strains 1 % apart: far more distinct k-mers than the capacity hint (assembly count) suggests.  The insert must notice
    # the full table early, retry with a larger one (this used to take minutes: every insert scanned 16 K slots) and still
    # give the oracle's graph.
    import time
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(30, genome=20_000, plasmid=0, sub=1e-2, indel=1e-4, seed=900_000)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"strain_{i:03d}.fasta"); hd.append(header)
    t0 = time.time()
    g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    assert g.timings()["table_capacity"] >= 4 * 65536 and g.timings()["n_distinct"] > 200_000
    assert time.time() - t0 < 60


KNOB_SETTINGS = [{"AC_TABLE_SHIFT": "0"}, {"AC_TABLE_SHIFT": "2"}, {"AC_MINKEY_VARIANT": "0"}, {"AC_MINKEY_VARIANT": "1"}, {"AC_MINKEY_VARIANT": "2"}, {"AC_MINKEY_VARIANT": "2", "AC_MINKEY_PREFIX_BASES": "2"}, {"AC_PATH_CHUNK": "64"}, {"AC_PATH_FILTER": "0"}, {"AC_EXPAND_REWRITE_ALWAYS": "1"}, {"AC_EXPAND_LEVEL_TABLE": "1"}, {"AC_SEED_RADIX_LIMIT": "0"}, {"AC_HOST_PACK": "0"}, {"AC_SEED_PREFIX_SORT": "0"}, {"AC_SEED_PREFIX_BITS": "3"}, {"AC_SEED_PREFIX_BITS": "12"}, {"AC_POS_CAP": "0"}, {"AC_POS_CAP": "3"}, {"AC_POS_CAP": "200", "AC_PATH_COPY": "1"}, {"AC_PATH_COPY": "1"}, {"AC_PATH_COPY": "1", "AC_RUN_PIECE": "150"}, {"AC_PATH_COPY": "1", "AC_PATH_FILTER": "0"}, {"AC_SEED_PREFIX_BITS": "3", "AC_SEED_MAX_GROUP": "2"}, {"AC_SEED_PREFIX_BITS": "6", "AC_SEED_MAX_GROUP": "1", "AC_SEED_RADIX_LIMIT": "1000000000"}, {"AC_SEED_PREFIX_SORT": "0", "AC_SEED_RADIX_LIMIT": "0"}, {"AC_RENUM_TWO_PASS": "1"}, {"AC_RENUM_MAX_GROUP": "1"}, {"AC_SORT_CHECKS": "1"}, {"AC_SORT_CHECKS": "1", "AC_RENUM_MAX_GROUP": "1"}, {"AC_RENUM_MAX_GROUP": "2"}, {"AC_DEGREE_FLAGS": "0"}, {"AC_DEGREE_REGION_CAP": "1"}, {"AC_DEGREE_REGION_CAP": "3"}, {"AC_UPLOAD_THREADS": "3"},
                 {"AC_REMAP_BLOCK": "128"}, {"AC_HOST_REMAP": "1"}, {"AC_HOST_REMAP": "1", "AC_UPLOAD_THREADS": "3"}, {"AC_HOST_REMAP": "0"}, {"AC_HOST_REMAP": "2"}, {"AC_HOST_REMAP": "2", "AC_UPLOAD_THREADS": "2"}, {"AC_HOST_REMAP": "2", "AC_REMAP_BLOCK": "128", "AC_STRETCH_DEVICE_SHARE": "50"}, {"AC_HOST_REMAP": "2", "AC_STRETCH_DEVICE_SHARE": "0"}, {"AC_SEQ_CODES": "2"}, {"AC_SEQ_CODES": "0"}, {"AC_LATE_COPIES": "2"}, {"AC_LATE_COPIES": "2", "AC_HOST_REMAP": "2", "AC_REMAP_BLOCK": "128", "AC_SEQ_CODES": "2"}, {"AC_LATE_COPIES": "2", "AC_HOST_REMAP": "0"}, {"AC_SEQ_CODES": "2", "AC_PACK_SCALAR": "1"}, {"AC_INSERT_ADAPT": "0", "AC_INSERT_GROWTH": "4"}, {"AC_INSERT_CHUNK": "256", "AC_INSERT_WAVES": "1024"}]


@pytest.mark.parametrize("knobs", KNOB_SETTINGS, ids=lambda d: ",".join(f"{a}={b}" for a, b in d.items()))
def test_tuning_knobs_do_not_change_the_result(emu, monkeypatch, knobs):
    # every tuning knob of graph_build.hip (table sizing, seed-kernel form, walker span, insert phases) only moves work around
    for a, b in knobs.items():
        monkeypatch.setenv(a, b)
    for k, seed in ((5, 3), (11, 7), (31, 11), (51, 13), (51, 21)):
        seqs, fn, hd = seqgen.make_case(seed, k)
        parity_util.check_case(k, seqs, fn, hd, lib_path=emu)


def test_paths_cross_as_stretches(emu, monkeypatch):
    """Round 6: AC_HOST_REMAP=2 = the path entries reach the host as stretches of consecutive text-order numbers (what a build with more than
    8 M unitigs does by itself) and are written out there: the same GFA, fewer records than entries."""
    monkeypatch.setenv("AC_HOST_REMAP", "2")
    seqs, fn, hd = _redundant_set(6, 40_000, 5)
    g, gfa, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    tm = g.timings()
    assert 0 < tm["path_stretches"] < tm["n_path_entries"], tm
    g.close()
    # ... with the last share of the entries renumbered on the device instead (the default is 30 %; small blocks so that this input has several)
    monkeypatch.setenv("AC_REMAP_BLOCK", "128")
    for share in ("0", "30", "55", "100"):
        monkeypatch.setenv("AC_STRETCH_DEVICE_SHARE", share)
        g, gfa_s, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
        assert gfa_s == gfa and g.timings()["path_stretches"] == tm["path_stretches"], share
        g.close()
    monkeypatch.delenv("AC_REMAP_BLOCK"); monkeypatch.delenv("AC_STRETCH_DEVICE_SHARE")
    monkeypatch.setenv("AC_HOST_REMAP", "0")
    g, gfa0, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    assert gfa0 == gfa and g.timings()["path_stretches"] == 0
    g.close()


def _redundant_set(n_asm, genome, seed):
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(n_asm, genome=genome, plasmid=genome // 40, sub=2e-4, indel=2e-5, seed=seed)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    return seqs, fn, hd


@pytest.mark.parametrize("adapt", ["1", "0"])
def test_insert_phase_schedule_on_a_redundant_text(emu, monkeypatch, adapt):
    # 10 similar assemblies of 80 kbp: long enough for the insert to reach its third phase, where the claim counters of
    # the first two decide that the rest of the text goes in one launch (3 launches) instead of doubling on (5 launches)
    monkeypatch.setenv("AC_INSERT_ADAPT", adapt)
    seqs, fn, hd = _redundant_set(10, 80_000, 2024)
    g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    assert g.timings()["insert_launches"] == (3 if adapt == "1" else 5)


def test_copying_walk_on_a_job_of_two_species(emu, monkeypatch):
    # Six assemblies of one genome, then six of another: the second species' FIRST assembly goes in with the one-launch rest of the
    # insert, so its copies repeat text whose novel bits the same launch is still writing.  The insert takes those bits on trust
    # (wave_match's nov_from; checking them ended every run after a few positions: benchjob8's insert 36 ms instead of 18) and the
    # piece-by-piece filter of the copying walk settles them: the graph is the oracle's, and runs of BOTH species are copied.
    from autocycler_amd import synth
    monkeypatch.setenv("AC_PATH_COPY", "1")
    seqs, fn, hd = [], [], []
    for sp, seed in enumerate((4711, 4712)):
        for i, contigs in enumerate(synth.make_assemblies(6, genome=70_000, plasmid=2_000, sub=3e-4, indel=3e-5, seed=seed)):
            for header, s_ in contigs:
                seqs.append(s_.tobytes().decode()); fn.append(f"species{sp}_{i:02d}.fasta"); hd.append(header)
    n_a = sum(1 for f in fn if f.startswith("species0"))
    for piece in ("4096", "300"):
        monkeypatch.setenv("AC_RUN_PIECE", piece)
        g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
        tm = g.timings()
        assert tm["insert_launches"] == 3 and tm["path_runs_copied"] > 0
        # the sample of the rest taken behind the second stretch: four of the ten assemblies still to come repeat the first species
        assert 0.25 < tm["insert_rest_known"] < 0.6
        one, _, _ = parity_util.check_case(51, seqs[:n_a], fn[:n_a], hd[:n_a], lib_path=emu)      # the first species alone
        assert tm["path_runs_copied"] > 1.6 * one.timings()["path_runs_copied"] > 0
        assert one.timings()["insert_rest_known"] > 0.9
    monkeypatch.delenv("AC_RUN_PIECE")


@pytest.mark.parametrize("copy", ["0", "1"])
def test_position_bound_and_the_repeat_with_exact_positions(emu, monkeypatch, copy):
    # The walk only notes smallest positions within AC_POS_CAP of a sequence end; beyond it expand_repeats sees a lower bound, and a
    # common sequence longer than the bound repeats the build with exact positions (exp_avoid_start_of_path).  With a bound of 1..40
    # positions the repeat must happen on some of these inputs and never change the result; with the shipped bound it never happens here.
    monkeypatch.setenv("AC_PATH_COPY", copy)
    retried = 0
    for cap in ("1", "7", "40"):
        monkeypatch.setenv("AC_POS_CAP", cap)
        for k, seed in ((11, 7), (31, 11), (51, 13), (51, 21)):
            seqs, fn, hd = seqgen.make_case(seed, k)
            g, _, _ = parity_util.check_case(k, seqs, fn, hd, lib_path=emu)
            retried += g.timings()["position_retries"]
        seqs, fn, hd = _redundant_set(6, 30_000, 77)
        g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
        retried += g.timings()["position_retries"]
    assert retried > 0
    monkeypatch.delenv("AC_POS_CAP")
    g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    assert g.timings()["position_retries"] == 0


@pytest.mark.parametrize("piece", ["0", "700"])
def test_copying_path_walk_on_a_redundant_text(emu, monkeypatch, piece):
    # K10c (AC_PATH_COPY=1, opt-in): the insert's followed runs are copied from the stretch they repeat instead of being walked; on ten
    # similar assemblies most of the text lies in runs — the knob must engage (runs copied, fewer entries walked than there are) and
    # change nothing.  piece: the runs cut into pieces of 700 positions
    monkeypatch.setenv("AC_PATH_COPY", "1")
    if piece != "0":
        monkeypatch.setenv("AC_RUN_PIECE", piece)
    seqs, fn, hd = _redundant_set(10, 80_000, 2024)
    g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    tm = g.timings()
    assert tm["path_runs_copied"] > 0 and 0 < tm["path_entries_walked"] < tm["n_path_entries"]      # (the first two assemblies are walked whatever happens)
    monkeypatch.delenv("AC_PATH_COPY")
    g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    assert g.timings()["path_runs_copied"] == 0


def test_seed_kernel_long_unitigs_across_wavefronts(emu, monkeypatch):
    # two identical 30 kbp sequences + one with a single substitution: unitigs of thousands of k-mers, i.e. dozens of
    # 64-entry wavefront pieces per unitig for the segmented-min seed kernel, in both forms
    import numpy as np
    rng = np.random.default_rng(5)
    a = "".join("ACGT"[i] for i in rng.integers(0, 4, size=30_000))
    b = a[:17_321] + ("A" if a[17_321] != "A" else "C") + a[17_322:]
    for variant in ("1", "0"):
        monkeypatch.setenv("AC_MINKEY_VARIANT", variant)
        for k in (21, 51, 101):
            parity_util.check_case(k, [a, a, b], ["x.fasta", "y.fasta", "z.fasta"], ["a", "a2", "b"], lib_path=emu)


def test_deferred_sort_flags_repeat_the_build(emu, monkeypatch):
    """Round 5: a single-device build reads the "group too large" flags of the seed sort and of the two renumberings with its LAST read-back
    (three host round trips less); a build in which one was set is repeated with checked sorts — forced here with tiny group limits."""
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []      # (SNP alleles: unitigs that tie on length and their first bases)
    for i, contigs in enumerate(synth.make_assemblies(4, genome=20_000, plasmid=2_000, sub=2e-3, indel=2e-4, seed=7)):
        for header, sq in contigs:
            seqs.append(sq.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
    assert g.timings()["sort_retries"] == 0
    g.close()
    for knobs in ({"AC_RENUM_MAX_GROUP": "1"}, {"AC_SEED_PREFIX_BITS": "3", "AC_SEED_MAX_GROUP": "1"}):
        for kk, vv in knobs.items():
            monkeypatch.setenv(kk, vv)
        g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)      # (== the oracle, byte for byte)
        assert g.timings()["sort_retries"] == 1, knobs
        g.close()
        monkeypatch.setenv("AC_SORT_CHECKS", "1")
        g, _, _ = parity_util.check_case(51, seqs, fn, hd, lib_path=emu)
        assert g.timings()["sort_retries"] == 0
        g.close()
        monkeypatch.delenv("AC_SORT_CHECKS")
        for kk in knobs:
            monkeypatch.delenv(kk)



_SPARSE_CASE = {}


def _sparse_case():
    """One mixed-species input and the oracle's answer for it (computed once for the six settings below)."""
    if not _SPARSE_CASE:
        import oracle_lib as O
        from autocycler_amd import synth
        seqs, fn, hd = synth.flatten(synth.make_mixed_species(2, 5, genome=40_000, plasmid=2_000, strain_div=2e-2, sub=2e-3, indel=2e-4, seed=7))
        s = O.Seqs.from_raw(51, [bytes(q) for q in seqs], filenames=fn, headers=hd, repair=True)
        gfa_o, st, _ = s.compress(51)
        _SPARSE_CASE.update(s=s, gfa=gfa_o, st=st, loaded=s.all())
    return _SPARSE_CASE


@pytest.mark.parametrize("knobs,ran", [({"AC_EXPAND_SPARSE_MAX": "0"}, False), ({}, True), ({"AC_EXPAND_SPARSE_MAX": "100000"}, True),
                                       ({"AC_EXPAND_SPARSE_MAX": "100000", "AC_EXPAND_SPARSE_BATCH": "3"}, True),
                                       ({"AC_EXPAND_SPARSE_MAX": "100000", "AC_EXPAND_SPARSE_LIST": "1"}, True),
                                       ({"AC_EXPAND_SPARSE_MAX": "100000", "AC_EXPAND_REWRITE_ALWAYS": "1"}, True)],
                         ids=["off", "default", "always", "tiny_lds_batch", "hands_back_after_a_sweep", "rewrite_every_check"])
def test_expand_sparse_tail(emu, monkeypatch, knobs, ran):
    """Round 6: behind the first two passes of expand_repeats ONE workgroup runs the remaining passes from a list of the dirty junctions
    (kernels_tail.inc expand_mopup_kernel).  The same graph whether it runs, stages a level in LDS or reads it from the list, or stops at a
    sweep boundary and hands back to the level launches — and the same number of passes as the level launches count."""
    from autocycler_amd import compress_build
    for a, b in knobs.items():
        monkeypatch.setenv(a, b)
    c = _sparse_case()
    loaded, st = c["loaded"], c["st"]
    g = compress_build(51, c["s"].assembly_count, [(q["fwd"], q["length"], q["id"]) for q in loaded], lib_path=emu)
    gfa_g = g.gfa([q["filename"] for q in loaded], [q["header"] for q in loaded])
    assert g.stats_post == dict(unitigs=st["unitigs_post"], links=st["links_post"], total_length=st["length_post"])
    assert gfa_g == c["gfa"], parity_util.first_diff(c["gfa"], gfa_g)
    tm = g.timings()
    assert tm["simplify_passes"] == 4, tm
    assert (tm["expand_sparse_sweeps"] > 0) == ran and (tm["expand_sparse_start"] > 0) == ran, tm
    if knobs.get("AC_EXPAND_SPARSE_LIST"):
        assert tm["expand_sparse_sweeps"] == 1, tm      # (it moved something, the list was longer than 1: back to the level launches)
    g.close()
