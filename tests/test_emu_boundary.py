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


def test_two_device_ordinals(emu):
    B.two_device_ordinals(emu)
