"""The sharded build (one compress job over several ranks) on the real device, through the C ABI
(ac_shard_*) and autocycler_amd/sharded.py: world_size 1 in-process, and two ranks as two processes sharing the test
box's single MI355X (gloo moves the buffers through the host there; on a multi-GPU node the same code runs over
RCCL).  GFA byte-for-byte against the oracle."""
import pytest
import torch

import seqgen
import sharded_util
from autocycler_amd import sharded
from test_oracle_kats import FIXED
from test_sharded_emu import launch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib_path():
    import autocycler_amd
    lib = autocycler_amd.load_library()       # raises HipLibraryMissing: the product has no fallback
    assert lib.ac_device_count() >= 1, "no HIP device visible"
    return autocycler_amd.LIB_PATH


@pytest.mark.parametrize("k", [3, 9, 51])
def test_single_rank_fixed_seqs(lib_path, k):
    dev = torch.device("cuda", 0)
    sharded_util.run_case(lib_path, k, [FIXED[c] for c in "abcde"], ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz"],
                          list("abcde"), sharded.Comm(dev), dev)


@pytest.mark.parametrize("k", [11, 51])
def test_single_rank_adversarial(lib_path, k):
    dev = torch.device("cuda", 0)
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        sharded_util.run_case(lib_path, k, seqs, fn, hd, sharded.Comm(dev), dev, repair=(seed % 2 == 0))


def test_single_rank_synthetic_medium(lib_path):
    from test_gpu_parity import _synth_case
    dev = torch.device("cuda", 0)
    seqs, fn, hd = _synth_case(8, 200_000, 8_000, 1e-3, 1e-4, 4242)
    sharded_util.run_case(lib_path, 51, seqs, fn, hd, sharded.Comm(dev), dev)


def test_two_ranks_one_gpu(lib_path):
    cases = ",".join(f"{k}:{seed}" for k in (11, 51) for seed in range(12)) + ",synth:51"
    launch(2, lib_path, "cuda:0", cases, timeout=900)


def test_two_ranks_realistic_size(lib_path):
    # 12 assemblies of a 2 Mbp genome split over two ranks == the single-device build of the same 12
    outs = launch(2, lib_path, "cuda:0", "big:12:2000000", timeout=900)
    assert "big case" in outs[0]


def test_rccl_collectives_world_of_one(lib_path):
    """Every collective autocycler_amd/sharded.py issues — all_gather_into_tensor (uint8 fragments, int32 degree slices, int64 sizes),
    all_reduce SUM and MIN (int32), gather (int32 paths, int64 records) — through the nccl backend (= RCCL) on device tensors, in a
    world of one rank on the test box's single MI355X, with the result checked against the oracle: RCCL is not first exercised on
    the multi-GPU node.  Multi-rank RCCL itself stays unmeasured until a node with several GPUs runs bench.py."""
    cases = ",".join(f"{k}:{seed}" for k in (11, 51) for seed in range(6)) + ",synth:51,mixed:51"
    outs = launch(1, lib_path, "cuda:0", cases, timeout=900, backend="nccl")
    assert "(nccl)" in outs[0]


def test_distinct_devices_over_rccl(lib_path):
    """where the box has several GPUs, one PROCESS per distinct device over the nccl backend (= RCCL) — the layout
    bench.py --gpus N runs under torch.distributed.run — on the adversarial set, the synthetic and the mixed-species job; every
    collective of autocycler_amd/sharded.py then moves device buffers between two GPUs.  Skipped on a one-GPU box."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU(s) visible: the inter-device RCCL path needs at least two")
    cases = ",".join(f"{k}:{seed}" for k in (11, 51) for seed in range(8)) + ",synth:51,mixed:51,partition"
    for world in sorted({2, min(n, 8)}):
        outs = launch(world, lib_path, "cuda:rank", cases, timeout=900, backend="nccl")
        assert "(nccl)" in outs[0]
