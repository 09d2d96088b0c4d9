"""tests/boundary_cases.py on the serial CPU emulation of the kernels (the product library needs the GPU)."""
import pytest

import boundary_cases as B
import emu_lib
import seqgen
from autocycler_amd import _capi
from test_oracle_kats import FIXED


@pytest.fixture(scope="module")
def emu():
    return emu_lib.emu_path()


def test_positions(emu):
    cases = [(k, *seqgen.make_case(seed, k)) for k, seed in ((5, 1), (9, 2), (21, 5), (51, 7), (51, 13))]
    cases.append((13, [FIXED[c] for c in "abcde"], [f for f, _ in B.FIVE], list("abcde")))
    B.positions_match_the_oracle(emu, cases)


@pytest.mark.parametrize("k", [1])
def test_fixed_seqs_k1(emu, k):      # tests.rs:131-148 runs k = 1 as well: one-base k-mers, no padding at all
    import parity_util
    parity_util.check_case(k, [FIXED[c] for c in "abcde"], [f for f, _ in B.FIVE], list("abcde"), lib_path=emu)


@pytest.mark.parametrize("k", [13, 51])
def test_compress_dir_five_file_fixture(emu, tmp_path, k):
    B.compress_dir_matches_the_oracle(_capi.load_library(emu), tmp_path, k)


def test_compress_dir_written_in_many_pieces(emu, tmp_path):
    B.compress_dir_many_pieces(_capi.load_library(emu), tmp_path)


def test_two_device_ordinals(emu):
    B.two_device_ordinals(emu)


def test_host_pack_simd_equals_scalar(emu):
    """K1 on the host (the packed upload of ac_compress_build): the widest SIMD loop this machine has (AVX-512, else AVX2 / BMI2) ==
    the portable loop == the definition, on bytes of every kind (bases, dots, separators, lower case, arbitrary), at lengths around
    the 32-byte groups."""
    B.host_pack_simd_equals_scalar(emu)


def test_host_pack_avx2_equals_scalar(emu):
    """The same with the AVX2 / BMI2 loop forced (the choice is made once per process: a fresh one)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    here = Path(__file__).resolve().parent
    code = (f"import sys; sys.path.insert(0, {str(here.parent)!r}); sys.path.insert(0, {str(here)!r}); "
            f"import boundary_cases as B; B.host_pack_simd_equals_scalar({str(emu)!r}); print('pack ok')")
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "AC_PACK_AVX2": "1", "AC_NO_TORCH": "1"}, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "pack ok" in out.stdout, out.stderr[-2000:]


def test_host_path_renumbering_portable_loop(emu):
    """PathRemapJob with the portable loop forced (AC_PACK_SCALAR switches the AVX-512 gather off; read once per process: a fresh one):
    paths renumbered on the host == the oracle's, positions included."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    here = Path(__file__).resolve().parent
    code = (f"import sys; sys.path.insert(0, {str(here.parent)!r}); sys.path.insert(0, {str(here)!r}); "
            "import boundary_cases as B, seqgen; "
            "cases = [(k, *seqgen.make_case(seed, k)) for k, seed in ((9, 2), (21, 5), (51, 13))]; "
            f"B.positions_match_the_oracle({str(emu)!r}, cases); print('remap ok')")
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "AC_PACK_SCALAR": "1", "AC_HOST_REMAP": "1", "AC_NO_TORCH": "1"},
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "remap ok" in out.stdout, out.stderr[-2000:]


def test_device_entry_rejects_a_bad_sequence_table(emu):
    """ac_compress_build_device / ac_shard_begin / ac_end_repair_device index the text with the caller's table — a table
    that does not describe the text must come back as an error (never a fault or a silent wrong graph)."""
    import ctypes as C
    import numpy as np
    lib = _capi.load_library(emu)
    k = 11
    seqs = ["ACGTTGCATGCATGGCATCGATCGGCTA", "GGCATCGATCGGCTAACGTTGCATGCAT"]
    text = ("$" + "$".join("." * 5 + s + "." * 5 for s in seqs) + "$").encode()
    buf = np.frombuffer(text, dtype=np.uint8).copy()
    good_off = [1, 1 + len(seqs[0]) + 10 + 1]
    lens = [len(s) for s in seqs]

    def run(off, ln, d1=(5, 5), d2=(5, 5), n_text=len(text)):
        g = C.c_void_p()
        rc = lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(2), buf.ctypes.data_as(C.c_void_p), C.c_uint64(n_text), (C.c_uint64 * 2)(*off),
                                          (C.c_uint32 * 2)(*ln), (C.c_uint16 * 2)(1, 2), (C.c_uint16 * 2)(*d1), (C.c_uint16 * 2)(*d2), C.c_uint32(2),
                                          C.c_int(0), C.byref(g))
        if rc == 0:
            lib.ac_free(g)
        return rc, lib.ac_last_error().decode()
    assert run(good_off, lens)[0] == 0
    assert "shorter than k" in run(good_off, [5, lens[1]])[1]
    assert "right behind the separator" in run([1, 20], lens)[1]
    assert "right behind the separator" in run([0, good_off[1]], lens)[1]
    assert "right behind the separator" in run([1, good_off[1] + 1], lens)[1]      # a gap between two sequences
    assert "does not end with the separator" in run(good_off, lens, n_text=len(text) + 1)[1]
    assert "past the end" in run(good_off, [lens[0], lens[1] + 3])[1]
    assert "past the end" in run(good_off, lens, n_text=len(text) - 2)[1]
    assert "padding dots" in run(good_off, lens, d1=(11, 5))[1]


def test_foreign_bytes_are_rejected(emu):
    B.foreign_bytes_are_rejected(emu)


def test_bulk_accessors_equal_the_per_item_ones(emu):
    import parity_util
    k, (seqs, fn, hd) = 21, seqgen.make_case(5, 21)
    g, gfa, _ = parity_util.check_case(k, seqs, fn, hd, lib_path=emu)
    b = g.bulk()
    for i in range(g.unitig_count):
        s, d = g.unitig(i)
        assert b["seq_bytes"][int(b["seq_begin"][i]):int(b["seq_begin"][i]) + int(b["seq_len"][i])].tobytes() == s and b["depth"][i] == d
    assert [(abs(int(l["a"])), int(l["a"]) > 0, abs(int(l["b"])), int(l["b"]) > 0) for l in b["links"]] == g.links()
    for sidx in range(len(seqs)):
        assert b["path_entries"][int(b["path_off"][sidx]):int(b["path_off"][sidx + 1])].tolist() == list(g.path(sidx))


def test_config_e_checker_on_a_small_mixed_species_job(emu):
    """tests/fullsize_e.py (the property checker of test_gpu_fullsize.py::test_config_e_full_size_k51, torch on the device there) on a
    12-assembly mixed-species job built by the emulation: it passes on the real graph and it FAILS on a flipped path entry, a changed
    unitig base and a changed depth."""
    import torch
    import fullsize_e
    lib = _capi.load_library(emu)
    job = fullsize_e.make_job(3, 4, genome=20_000, plasmid=1_500, workers=1)
    text = job["text"]
    g, _, _ = fullsize_e.build_device(lib, job, text.ctypes.data)
    t = torch.from_numpy(text)
    r = fullsize_e.check_on_device(g, job, t, chunk=70_000)
    assert r["unitigs"] == g.unitig_count and r["path_entries"] > 0
    b = g.bulk()
    def broken(arr, i, v):
        old = arr[i]; arr[i] = v
        try:
            fullsize_e.check_on_device(g, job, t, chunk=70_000)
        except AssertionError:
            return True
        finally:
            arr[i] = old
        return False
    assert broken(b["path_entries"], 5, -b["path_entries"][5])
    assert broken(b["seq_bytes"], 100, ord("A") if b["seq_bytes"][100] != ord("A") else ord("C"))
    assert broken(b["depth"], 3, b["depth"][3] + 1)


def test_verify_graph_accepts_oracle_equal_graphs(emu):
    import verify_cases
    assert verify_cases.accepts_oracle_equal_graphs(emu, ks=(5, 11, 51), seeds=range(8)) == 24


def test_verify_graph_names_the_damage(emu):
    import verify_cases
    verify_cases.names_the_damage(emu)


def test_verify_graph_names_order_sensitive_damage(emu):
    """maximality, expand_repeats' fixed point, L-line order (unitig_graph.rs:192-223, 333-350; graph_simplification.rs:26-86)"""
    import verify_cases
    verify_cases.names_order_sensitive_damage(emu)


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_hand_written_primitives_equal_std(emu, kind):
    # csrc/device_prims.hpp (scan with decoupled look-back, onesweep radix sort, merge sort by ranks) under the lockstep emulation: tile
    # boundaries (2048 items per tile), end bits that are not a multiple of eight, duplicate-heavy keys (stability)
    import ctypes as C
    lib = _capi.load_library(emu)
    for n in (0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4096, 4097, 9999, 40_000):
        for end_bit in ((64, 33, 8, 5) if n < 5000 else (64, 21)):
            assert lib.ac_selftest_primitives(C.c_int(0), C.c_uint64(n), C.c_uint64(7 * n + kind), C.c_int(end_bit), C.c_int(kind)) == 0, (n, end_bit, lib.ac_last_error())
