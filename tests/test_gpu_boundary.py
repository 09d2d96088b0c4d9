"""tests/boundary_cases.py on the MI355X: the boundary pieces the round-1 device suite did not reach."""
import pytest

import boundary_cases as B
import seqgen
from test_oracle_kats import FIXED

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import autocycler_amd
    lib = autocycler_amd.load_library()       # raises HipLibraryMissing: the product has no fallback
    assert lib.ac_device_count() >= 1, "no HIP device visible"
    return lib


def test_positions(lib):
    from test_gpu_parity import _synth_case
    cases = [(k, *seqgen.make_case(seed, k)) for k, seed in ((5, 1), (9, 2), (21, 5), (51, 7), (51, 13), (101, 3))]
    cases.append((13, [FIXED[c] for c in "abcde"], [f for f, _ in B.FIVE], list("abcde")))
    cases.append((51, *_synth_case(6, 60_000, 3_000, 1e-3, 1e-4, 99)))
    B.positions_match_the_oracle(None, cases)


def test_fixed_seqs_k1(lib):      # tests.rs:131-148 runs k = 1 as well
    import parity_util
    parity_util.check_case(1, [FIXED[c] for c in "abcde"], [f for f, _ in B.FIVE], list("abcde"))


@pytest.mark.parametrize("k", [13, 51])
def test_compress_dir_five_file_fixture(lib, tmp_path, k):
    B.compress_dir_matches_the_oracle(lib, tmp_path, k)


def test_compress_dir_written_in_many_pieces(lib, tmp_path):
    B.compress_dir_many_pieces(lib, tmp_path)


@pytest.mark.parametrize("k", [13, 51])
def test_cli_five_file_fixture(lib, tmp_path, k):
    B.cli_matches_the_oracle(tmp_path, k)


def test_two_device_ordinals(lib):
    B.two_device_ordinals(None)


def test_decompress_a_device_built_graph(lib, tmp_path):
    """ac_decompress_seq on the handle ac_compress_build returned (no GFA in between) and ac_decompress on the GFA file written
    from it: both reproduce every input sequence (unitig_graph.rs:362-388, decompress.rs:83-105)."""
    import ctypes as C
    import parity_util
    from test_gpu_parity import _synth_case
    seqs, fn, hd = _synth_case(6, 60_000, 3_000, 1e-3, 1e-4, 99)
    g, gfa, loaded = parity_util.check_case(51, seqs, fn, hd)
    from autocycler_amd import graph_from_gfa
    g2, fns, hds = graph_from_gfa(gfa)
    assert [g2.decompress(i).decode() for i in range(len(seqs))] == seqs
    p = tmp_path / "input_assemblies.gfa"
    p.write_text(gfa)
    one = tmp_path / "all.fasta"
    assert lib.ac_decompress(str(p).encode(), None, str(one).encode(), C.c_int(4)) == 0, lib.ac_last_error()
    want = "".join(f">{f}__{h}\n{s}\n" for f, h, s in sorted(zip(fn, hd, seqs), key=lambda x: x[0]))
    assert one.read_text() == want


@pytest.mark.parametrize("k", [9, 51])
def test_c_client(lib, tmp_path, k):
    """A plain C99 program bound to include/autocycler_hip.h (tests/c_client/client.c) — the view a Rust / cgo / JNI binding has of
    the library — builds the same GFA as the oracle on the device."""
    from test_gpu_parity import _synth_case
    seqs, fn, hd = seqgen.make_case(7, k)
    B.c_client_matches_the_oracle(tmp_path, k, seqs, len(set(fn)))
    if k == 51:
        seqs, fn, hd = _synth_case(4, 30_000, 2_000, 1e-3, 1e-4, 5)
        B.c_client_matches_the_oracle(tmp_path, k, seqs, 4)


def test_foreign_bytes_are_rejected(lib):
    import autocycler_amd
    B.foreign_bytes_are_rejected(autocycler_amd.LIB_PATH)


def test_bulk_accessors_equal_the_per_item_ones(lib):
    """ac_unitigs_bulk / ac_paths_bulk (zero-copy views for graphs of 10^8 unitigs) hold what ac_unitig / ac_path / ac_links return."""
    import numpy as np
    import parity_util
    from test_gpu_parity import _synth_case
    seqs, fn, hd = _synth_case(6, 60_000, 3_000, 1e-3, 1e-4, 99)
    g, gfa, _ = parity_util.check_case(51, seqs, fn, hd)
    b = g.bulk()
    for i in range(g.unitig_count):
        s, d = g.unitig(i)
        assert b["seq_bytes"][int(b["seq_begin"][i]):int(b["seq_begin"][i]) + int(b["seq_len"][i])].tobytes() == s and b["depth"][i] == d
    assert [(abs(int(l["a"])), int(l["a"]) > 0, abs(int(l["b"])), int(l["b"]) > 0) for l in b["links"]] == g.links()
    for sidx in range(len(seqs)):
        assert b["path_entries"][int(b["path_off"][sidx]):int(b["path_off"][sidx + 1])].tolist() == list(g.path(sidx))


def test_library_is_built_from_this_tree():
    """bench.py only quotes PMC traffic counted on a library of exactly the sources in the tree: the digest the loaded product library
    carries (csrc/Makefile SRC_HASH) is the digest of the tree (tools/source_hash.py)."""
    import importlib.util
    from pathlib import Path
    import autocycler_amd
    spec = importlib.util.spec_from_file_location("source_hash", Path(__file__).resolve().parent.parent / "tools" / "source_hash.py")
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert autocycler_amd.load_library().ac_source_hash().decode() == m.source_hash()


def test_verify_graph_accepts_oracle_equal_graphs(lib):
    # ac_verify_graph: every oracle-equal graph holds, built or reloaded from its GFA
    import verify_cases
    assert verify_cases.accepts_oracle_equal_graphs(None, ks=(5, 11, 51), seeds=range(12)) == 36


def test_verify_graph_names_the_damage(lib):
    # one flipped base, one dropped link, one swapped path entry, ... : each reported with its class; the graph holds again once restored
    import verify_cases
    verify_cases.names_the_damage(None)


def test_verify_graph_names_order_sensitive_damage(lib):
    # round 6: a unitig cut in two (maximality, unitig_graph.rs:192-223), a shift expand_repeats did not apply (graph_simplification.rs:26-86),
    # L lines out of get_links_for_gfa order (unitig_graph.rs:333-350) — each named
    import verify_cases
    verify_cases.names_order_sensitive_damage(None)


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_hand_written_primitives_equal_std(lib, kind):
    # csrc/device_prims.hpp on the device: thousands of tiles in flight (the look-back chains), every key kind, odd end bits
    import ctypes as C
    for n in (0, 1, 257, 2048, 2049, 100_003, 1_000_000, 6_000_011):
        for end_bit in ((64, 33, 5) if n < 200_000 else (64, 27)):
            assert lib.ac_selftest_primitives(C.c_int(0), C.c_uint64(n), C.c_uint64(11 * n + kind), C.c_int(end_bit), C.c_int(kind)) == 0, (n, end_bit, lib.ac_last_error())
