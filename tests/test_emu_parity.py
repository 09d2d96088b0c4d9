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
    with pytest.raises(AutocyclerError, match="not supported"):
        compress_build(201, 1, [(b"." * 100 + b"A" * 300 + b"." * 100, 300, 1)], lib_path=emu)
