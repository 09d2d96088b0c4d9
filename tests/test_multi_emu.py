"""ac_compress_build_multi on the CPU emulation: one process, a host thread per rank, exchanges staged through host memory — the world
sizes the scaling benchmark runs at (2, 4, 8) and an odd one, byte for byte against the oracle; the walk-start keys routed by owner."""
import pytest

import emu_lib
import multi_cases as M
from test_oracle_kats import FIXED


@pytest.fixture(scope="module")
def emu():
    return emu_lib.emu_path()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_adversarial_cases(emu, world):
    assert M.adversarial(emu, [0] * world) == 4 * 24 * 2


@pytest.mark.parametrize("world", [4, 8])
def test_adversarial_cases_many_ranks(emu, world):
    assert M.adversarial(emu, [0] * world, ks=(11, 51)) == 2 * 24 * 2


@pytest.mark.parametrize("k", [3, 5, 13, 51])
def test_fixed_seqs_more_ranks_than_sequences(emu, k):
    # five sequences, eight devices asked for: five ranks run
    _, info = M.run_case(emu, k, [FIXED[c] for c in "abcde"], ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz"], list("abcde"), [0] * 8)
    assert info["n_ranks"] == 5


@pytest.mark.parametrize("world", [2, 4, 8])
def test_synthetic_and_mixed_species_jobs(emu, world, monkeypatch):
    seqs, fn, hd = M.synth_case(9, 30_000, 2_000, 1e-3, 1e-4, 11)
    gfa1, _ = M.run_case(emu, 51, seqs, fn, hd, [0])
    gfa, info = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
    assert gfa == gfa1
    assert info["transport"] == 1 and info["queries_total"] > 0
    seqs, fn, hd = M.mixed_case(world, 3, 12_000)
    gfa, info = M.run_case(emu, 21, seqs, fn, hd, [0] * world)
    # the table is partitioned: every rank's share is well below the whole job's table, and the shares add up to about one table
    monkeypatch.setenv("AC_MULTI_TRANSPORT", "host")      # (a named transport: one rank runs the protocol too — its table is sized the way the ranks' shares are)
    _, one = M.run_case(emu, 21, seqs, fn, hd, [0])
    monkeypatch.delenv("AC_MULTI_TRANSPORT")
    assert one["transport"] == 1
    assert info["table_capacity_max"] * world <= 4 * one["table_capacity_max"] and info["table_capacity_max"] < one["table_capacity_max"] or world == 1
    # owner routing: a rank's queries go to the rank that owns them — about (world - 1) / world of them leave, none is broadcast
    assert info["queries_sent_away"] <= info["queries_total"]
    assert info["queries_sent_away"] >= info["queries_total"] * (world - 1) // (2 * world)
    assert info["bytes_queries"] == info["queries_sent_away"] * 8 * 1      # k = 21: one-word keys, each sent to exactly ONE rank


@pytest.mark.parametrize("world", [2, 4, 8])
def test_expand_repeats_is_partitioned_by_conflict_component(emu, world, monkeypatch):
    """the order-sensitive tail is no longer replicated in full — a rank runs only the junctions of the conflict
    components it owns (1 / world of them, within the imbalance of whole components), the results are merged by all-reduces, and the
    graph is still the oracle's (run_case compares).  Also with a rank-local rewrite in between (the merge then separates folded
    pieces), and against the replicated tail."""
    for name, (seqs, fn, hd) in {"one species": M.synth_case(8, 60_000, 3_000, 1e-3, 1e-4, 7), "mixed": M.mixed_case(4, 4, 40_000)}.items():
        gfa, info = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
        total, most = info["candidates_total"], info["candidates_owned_max"]
        assert total > 500 and most * world <= 1.5 * total + 64, (name, world, total, most)
        monkeypatch.setenv("AC_EXPAND_REWRITE_ALWAYS", "1")
        gfa_rw, _ = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
        monkeypatch.delenv("AC_EXPAND_REWRITE_ALWAYS")
        monkeypatch.setenv("AC_MULTI_TAIL", "replicated")
        gfa_rep, rep = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
        monkeypatch.delenv("AC_MULTI_TAIL")
        assert gfa == gfa_rw == gfa_rep and rep["candidates_owned_max"] == rep["candidates_total"] == total


@pytest.mark.parametrize("world", [2, 8])
def test_fragments_travel_as_two_bit_codes(emu, world, monkeypatch):
    """the union text stays 2-bit packed — the ranks exchange code words on the union text's word grid (a quarter of
    the bytes of the text exchange, AC_MULTI_FRAGMENTS=bytes), the receivers derive the mask plane from the fragment records."""
    seqs, fn, hd = M.synth_case(8, 60_000, 3_000, 1e-3, 1e-4, 7)
    gfa, info = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
    monkeypatch.setenv("AC_MULTI_FRAGMENTS", "bytes")
    gfa_b, info_b = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
    assert gfa == gfa_b and info["fragments"] == info_b["fragments"] and info["union_text_bytes"] == info_b["union_text_bytes"]
    records = 8 * info["fragments"] * (world - 1)            # the 8-byte records travel either way
    assert (info["bytes_fragments"] - records) * 3.5 <= (info_b["bytes_fragments"] - records) + 64 * world * world


def test_errors_come_back_from_the_rank_threads(emu):
    import ctypes as C
    from autocycler_amd import AutocyclerError
    lib = _capi_lib(emu)
    seqs = [("." * 5 + "ACGTNACGTACGTTGCA" + "." * 5).encode(), ("." * 5 + "ACGTTACGTACGTTGCA" + "." * 5).encode()]
    with pytest.raises(AutocyclerError, match="input sequence 1 contains non-ACGT characters"):
        from autocycler_amd import _capi
        _capi.compress_build_multi(11, 2, [(seqs[0], 17, 1), (seqs[1], 17, 2)], [0, 0], lib_path=emu)
    # and the library is usable afterwards
    M.run_case(emu, 11, ["ACGTAACGTACGTTGCAAGGT", "ACGTTACGTACGTTGCATTGA"], ["a.fa", "b.fa"], ["x", "y"], [0, 0])
    assert lib.ac_release_memory() == 0


def _capi_lib(emu):
    from autocycler_amd import _capi
    return _capi.load_library(emu)


@pytest.mark.parametrize("world", [1, 3])
def test_whole_command_over_several_ranks(emu, tmp_path, world):
    import boundary_cases as B
    B.compress_dir_multi_matches_the_oracle(_capi_lib(emu), tmp_path, 13, [0] * world)


def test_one_device_is_a_single_device_build(emu, monkeypatch):
    """Round 5: ac_compress_build_multi over ONE device has nobody to exchange with — it takes the single-device build (transport 3 =
    direct); naming a transport keeps every phase of the protocol (what the device suite's RCCL-in-a-world-of-one tests rely on)."""
    seqs, fn, hd = M.synth_case(6, 30_000, 2_000, 1e-3, 1e-4, 3)
    gfa_d, direct = M.run_case(emu, 51, seqs, fn, hd, [0])
    assert direct["transport"] == 3 and direct["n_ranks"] == 1 and direct["fragments"] == 0 and direct["bytes_links"] == 0
    monkeypatch.setenv("AC_MULTI_TRANSPORT", "host")
    gfa_p, proto = M.run_case(emu, 51, seqs, fn, hd, [0])
    assert gfa_p == gfa_d and proto["transport"] == 1 and proto["fragments"] > 0


@pytest.mark.parametrize("world", [2, 3])
def test_round5_protocol_knobs(emu, world, monkeypatch):
    """The things round 5 gave the N-rank protocol, each switched off again: the sibling bits (exchanged by novel index) with the probe-free
    degree step and its compact exchange (AC_SHARD_DEGREE_FLAGS=0: every degree by probing, a full byte per k-mer), the host-side
    renumbering of a rank's own paths (AC_SHARD_HOST_REMAP=0).  Same graph either way."""
    for seqs, fn, hd, k in (M.synth_case(8, 40_000, 2_000, 1e-3, 1e-4, 7) + (51,), M.mixed_case(3, 3, 20_000) + (21,)):
        gfa, info = M.run_case(emu, k, seqs, fn, hd, [0] * world)
        monkeypatch.setenv("AC_SHARD_DEGREE_FLAGS", "0")
        gfa_probe, info_probe = M.run_case(emu, k, seqs, fn, hd, [0] * world)
        monkeypatch.delenv("AC_SHARD_DEGREE_FLAGS")
        monkeypatch.setenv("AC_SHARD_HOST_REMAP", "0")
        gfa_dev, _ = M.run_case(emu, k, seqs, fn, hd, [0] * world)
        monkeypatch.delenv("AC_SHARD_HOST_REMAP")
        assert gfa == gfa_probe == gfa_dev
        # the bitmap exchange is the same; the sibling bits are 2 per DISTINCT k-mer; the degree exchange is a byte per k-mer the light step
        # left open (+ 4 per sequence end) instead of a byte per distinct k-mer
        ring = lambda nbytes: nbytes * (world - 1) // world * 2 * world      # an all-reduce of nbytes as every rank's received bytes, summed
        assert info["bytes_bitmap"] == info_probe["bytes_bitmap"] and info_probe["bytes_sibling"] == 0
        assert 0 < info["bytes_sibling"] <= ring(2 * (info["distinct"] // 64 + 2) * 8) + 64
        assert info_probe["bytes_degrees"] >= ring(info["distinct"]) - 64 and info_probe["degrees_open"] == 0
        assert 0 < info["degrees_open"] < info["distinct"] // 3
        assert info["bytes_degrees"] <= ring(info["degrees_open"] + 8 * len(seqs)) + 64 < info_probe["bytes_degrees"]
        assert info["bytes_links"] == info_probe["bytes_links"]      # (40 B per unitig either way since the walk words stay at home)
        assert 0 < info["bytes_received_max"] * world <= 2 * sum(info[x] for x in info if x.startswith("bytes_") and x != "bytes_received_max")
    M.adversarial(emu, [0] * world, ks=(11, 51), seeds=range(6))


@pytest.mark.parametrize("world", [1, 2, 3])
def test_copying_walk_of_the_ranks(emu, world, monkeypatch):
    """Round 5: a rank's LOCAL insert notes the runs it follows, the rank checks them against its OWN novel bitmap and walks only the
    gaps (the walk-start keys it sends to the owners are the gap walkers'); what it copies instead of walking shows in
    ac_multi_info.path_runs_copied.  Same graph as with every rank walking all of its text (AC_SHARD_PATH_COPY=0) and as one device."""
    monkeypatch.setenv("AC_MULTI_TRANSPORT", "host")      # (one rank: every phase of the protocol, not the direct dispatch)
    monkeypatch.setenv("AC_PATH_COPY", "1")
    # every rank's slice must be redundant in itself: 8 assemblies of 70 kbp per rank (the insert's one-launch rest needs > 4 x 64 K positions)
    seqs, fn, hd = M.synth_case(8 * world, 70_000, 2_000, 3e-4, 3e-5, 11)
    for piece in (None, "300"):
        if piece: monkeypatch.setenv("AC_RUN_PIECE", piece)
        gfa, info = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
        assert info["path_runs_copied"] > 0
        monkeypatch.setenv("AC_SHARD_PATH_COPY", "0")
        gfa_walk, info_walk = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
        monkeypatch.delenv("AC_SHARD_PATH_COPY")
        assert gfa == gfa_walk and info_walk["path_runs_copied"] == 0
        assert info["queries_total"] < info_walk["queries_total"]      # fewer walkers: only the gaps are walked
    monkeypatch.delenv("AC_RUN_PIECE")
    monkeypatch.delenv("AC_MULTI_TRANSPORT")
    gfa_one, _ = M.run_case(emu, 51, seqs, fn, hd, [0])
    assert gfa_one == gfa
