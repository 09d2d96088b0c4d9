"""Device end repair on CPU: the serial emulation of the same sources behind ac_end_repair_device,
against the oracle's restatement of sequence_end_repair + find_best_match (compress.rs:202-270), byte for byte."""
import pytest

import emu_lib
import repair_util
import seqgen
from test_oracle_kats import FIXED


@pytest.fixture(scope="module")
def emu():
    return emu_lib.emu_path()


@pytest.mark.parametrize("k", [3, 5, 9, 13, 51])
def test_fixed_seqs(emu, k):
    repair_util.check_repair(emu, k, [FIXED[c] for c in "abcde"], ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz"], list("abcde"))


@pytest.mark.parametrize("k", [3, 5, 7, 11, 21, 31, 51, 101, 123, 129, 131, 251, 501])
def test_adversarial_cases(emu, k):
    total = 0
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        total += repair_util.check_repair(emu, k, seqs, fn, hd)[5]
    assert total > 0


def test_synthetic_assemblies(emu):
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(10, genome=30_000, plasmid=2_000, sub=2e-3, indel=2e-4, seed=5)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    for k in (51, 21):
        repair_util.check_repair(emu, k, seqs, fn, hd)
