"""One compress job sharded by sequence over several ranks, on CPU: the serial emulation of the HIP
sources behind the very same ac_shard_* entry points and the very same torch.distributed plumbing
(autocycler_amd/sharded.py), gloo backend, world_size 1, 2 and 3 — GFA byte-for-byte against the oracle."""
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

import emu_lib
import seqgen
import sharded_util
from autocycler_amd import sharded
from test_oracle_kats import FIXED

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def emu():
    return emu_lib.emu_path()


@pytest.mark.parametrize("k", [3, 5, 9, 13, 51])
def test_single_rank_fixed_seqs(emu, k):
    # world == 1: the fragment / union-text / two-context walk machinery alone, no collectives
    comm = sharded.Comm(torch.device("cpu"))
    sharded_util.run_case(emu, k, [FIXED[c] for c in "abcde"], ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz"], list("abcde"),
                          comm, torch.device("cpu"))


@pytest.mark.parametrize("k", [5, 11, 31, 51])
def test_single_rank_adversarial(emu, k):
    comm = sharded.Comm(torch.device("cpu"))
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        for repair in (True, False):
            sharded_util.run_case(emu, k, seqs, fn, hd, comm, torch.device("cpu"), repair=repair)


def test_single_rank_takes_the_single_device_build(emu):
    # round 5: a world of one rank has nobody to exchange with — sharded_build() hands the job to the single-device entry
    comm = sharded.Comm(torch.device("cpu"))
    for seed in (1, 5, 9):
        seqs, fn, hd = seqgen.make_case(seed, 11)
        assert sharded_util.run_case(emu, 11, seqs, fn, hd, comm, torch.device("cpu"), direct_when_alone=True) == \
            sharded_util.run_case(emu, 11, seqs, fn, hd, comm, torch.device("cpu"), direct_when_alone=False)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return str(p)


def launch(world, lib_path, device, cases, timeout=600, backend="gloo"):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "sharded_worker.py"), str(r), str(world), port, str(lib_path),
                               device, cases, backend], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "cases OK" in out
    return outs


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_gloo(emu, world):
    cases = ",".join(f"{k}:{seed}" for k in (5, 11, 31, 51) for seed in range(24)) + ",synth:51,synth:21,mixed:51"
    launch(world, emu, "cpu", cases)


@pytest.mark.parametrize("world", [4, 8])
def test_many_ranks_gloo(emu, world):
    # the world sizes the scaling benchmark runs at (bench.py --gpus 4 / 8): a mixed-species job with one species per rank, the
    # single-species synthetic job, and adversarial cases that have at least `world` sequences
    cases = ",".join(f"{k}:{seed}" for k in (11, 51) for seed in range(24)) + ",synth:51,mixed:51"
    outs = launch(world, emu, "cpu", cases, timeout=900)
    done = int(outs[0].split("rank 0:")[1].split()[0])
    assert done >= 3, outs[0][-500:]


def test_copying_walk_over_gloo(emu, monkeypatch):
    # round 5: every rank copies the followed runs of its own slice instead of walking them (its local insert notes them, its own novel
    # bitmap settles them); two processes over gloo, 8 redundant assemblies of 70 kbp each, result = the single-device build's
    monkeypatch.setenv("AC_PATH_COPY", "1")
    outs = launch(2, emu, "cpu", "big:16:70000", timeout=900)
    assert "big case" in outs[0]


@pytest.mark.parametrize("world", [2, 8])
def test_table_is_partitioned_over_the_ranks(emu, world):
    # per-rank table capacity ~ 1/N of the single-device one, result still byte-identical to the oracle
    outs = launch(world, emu, "cpu", "partition", timeout=900)
    assert f"partition case: world {world}" in outs[0]


def test_phase_order_is_enforced(emu):
    """The ac_shard_* calls only work in protocol order, and no other build may start while a sharded one is in flight."""
    import ctypes as C
    from autocycler_amd import AutocyclerError, _capi, compress_build
    lib = _capi.load_library(emu)
    seqs, fn, hd = seqgen.make_case(1, 11)
    import oracle_lib as O
    loaded = O.Seqs.from_raw(11, seqs, filenames=fn, headers=hd).all()
    shard = sharded_util.local_shard(lib, 11, loaded, 0, len(loaded), 1, torch.device("cpu"))
    h = C.c_void_p()
    assert lib.ac_shard_begin(C.c_uint32(11), C.c_uint32(1), C.c_void_p(shard.d_text.data_ptr()), C.c_uint64(shard.n_text), shard.off, shard.lens,
                              shard.ids, shard.d1, shard.d2, C.c_uint32(shard.n_seqs), C.c_int(0), C.byref(h)) == 0
    try:
        g = C.c_void_p()
        assert lib.ac_shard_build_graph(h, None) != 0 and b"wrong phase" in lib.ac_last_error()
        assert lib.ac_shard_finish(h, C.c_int(3), C.byref(g)) != 0 and b"wrong phase" in lib.ac_last_error()
        assert lib.ac_shard_reduce_import(h, None, None) != 0
        with pytest.raises(AutocyclerError, match="sharded build is in flight"):
            compress_build(11, 1, [(q["fwd"], q["length"], q["id"]) for q in loaded], lib_path=emu)
        assert lib.ac_release_memory() != 0
    finally:
        lib.ac_shard_free(h)
    assert lib.ac_release_memory() == 0
    compress_build(11, 1, [(q["fwd"], q["length"], q["id"]) for q in loaded], lib_path=emu).close()      # and builds work again


@pytest.mark.parametrize("world", [2, 3])
def test_a_refused_slice_fails_every_rank(emu, world):
    # a rank that fails before the first collective must not leave its peers inside it
    launch(world, emu, "cpu", "badinput", timeout=300)


def test_three_ranks_medium_size_vs_single(emu):
    # the same comparison on the CPU emulation at a size it can do: 9 assemblies of 40 kbp over three ranks
    outs = launch(3, emu, "cpu", "big:9:40000")
    assert "big case" in outs[0]
