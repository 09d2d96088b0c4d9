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
def test_synthetic_and_mixed_species_jobs(emu, world):
    seqs, fn, hd = M.synth_case(9, 30_000, 2_000, 1e-3, 1e-4, 11)
    gfa1, _ = M.run_case(emu, 51, seqs, fn, hd, [0])
    gfa, info = M.run_case(emu, 51, seqs, fn, hd, [0] * world)
    assert gfa == gfa1
    assert info["transport"] == 1 and info["queries_total"] > 0
    seqs, fn, hd = M.mixed_case(world, 3, 12_000)
    gfa, info = M.run_case(emu, 21, seqs, fn, hd, [0] * world)
    # the table is partitioned: every rank's share is well below the whole job's table, and the shares add up to about one table
    _, one = M.run_case(emu, 21, seqs, fn, hd, [0])
    assert info["table_capacity_max"] * world <= 4 * one["table_capacity_max"] and info["table_capacity_max"] < one["table_capacity_max"] or world == 1
    # owner routing: a rank's queries go to the rank that owns them — about (world - 1) / world of them leave, none is broadcast
    assert info["queries_sent_away"] <= info["queries_total"]
    assert info["queries_sent_away"] >= info["queries_total"] * (world - 1) // (2 * world)
    assert info["bytes_queries"] == info["queries_sent_away"] * 8 * 1      # k = 21: one-word keys, each sent to exactly ONE rank


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
