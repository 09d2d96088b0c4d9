"""ac_compress_build_multi on the MI355X box (one GPU): the in-library RCCL path in a world of one rank (ncclCommInitAll, ncclAllReduce on
uint8 / int32 / int64 / uint64, grouped ncclSend / ncclRecv to itself: every call the N-rank build issues), and 2, 3 and 4 ranks sharing
the one device over the host-staged transport (a thread, a device context, an arena and a k-mer table share per rank) — byte for byte
against the oracle; BASELINE configs[1] over two ranks against its golden digest."""
import hashlib
import json
import os
from pathlib import Path

import pytest

import multi_cases as M
from test_oracle_kats import FIXED

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import autocycler_amd
    lib = autocycler_amd.load_library()
    assert lib.ac_device_count() >= 1, "no HIP device visible"
    return lib


def test_rccl_world_of_one(lib, monkeypatch):
    monkeypatch.setenv("AC_MULTI_TRANSPORT", "rccl")
    for k in (5, 11, 51):
        for seed in range(6):
            import seqgen
            seqs, fn, hd = seqgen.make_case(seed, k)
            _, info = M.run_case(None, k, seqs, fn, hd, [0])
            assert info["transport"] == 2 and info["n_ranks"] == 1
    seqs, fn, hd = M.synth_case(6, 60_000, 3_000, 1e-3, 1e-4, 99)
    _, info = M.run_case(None, 51, seqs, fn, hd, [0])
    assert info["transport"] == 2 and info["queries_total"] > 0 and info["queries_sent_away"] == 0


@pytest.mark.parametrize("world", [1, 2, 4])
def test_copying_walk_of_the_ranks(lib, world, monkeypatch):
    """Round 5 (tests/test_multi_emu.py has the same on the emulation): a rank's local insert notes its followed runs, the rank copies
    their path entries instead of walking them — forced here (AC_PATH_COPY=1: the cost model only picks it for slices of config C's size,
    where `bench.py --mode sharded --protocol-always` shows it).  Same graph as with AC_SHARD_PATH_COPY=0 and as one device."""
    monkeypatch.setenv("AC_MULTI_TRANSPORT", "host")
    seqs, fn, hd = M.synth_case(8 * world, 70_000, 2_000, 3e-4, 3e-5, 11)
    monkeypatch.setenv("AC_PATH_COPY", "1")
    gfa, info = M.run_case(None, 51, seqs, fn, hd, [0] * world)
    assert info["path_runs_copied"] > 0
    monkeypatch.setenv("AC_SHARD_PATH_COPY", "0")
    gfa_walk, info_walk = M.run_case(None, 51, seqs, fn, hd, [0] * world)
    monkeypatch.delenv("AC_SHARD_PATH_COPY")
    assert gfa == gfa_walk and info_walk["path_runs_copied"] == 0 and info["queries_total"] < info_walk["queries_total"]
    monkeypatch.delenv("AC_PATH_COPY")
    monkeypatch.delenv("AC_MULTI_TRANSPORT")
    gfa_one, _ = M.run_case(None, 51, seqs, fn, hd, [0])
    assert gfa_one == gfa


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_sharing_the_device(lib, world):
    assert M.adversarial(None, [0] * world, ks=(11, 51), seeds=range(8)) == 2 * 8 * 2
    seqs, fn, hd = M.synth_case(8, 60_000, 3_000, 1e-3, 1e-4, 7)
    gfa1, _ = M.run_case(None, 51, seqs, fn, hd, [0])
    gfa, info = M.run_case(None, 51, seqs, fn, hd, [0] * world)
    assert gfa == gfa1 and info["transport"] == 1 and info["n_ranks"] == world
    seqs, fn, hd = M.mixed_case(world, 4, 40_000)
    _, info = M.run_case(None, 51, seqs, fn, hd, [0] * world)
    assert 0 < info["queries_sent_away"] < info["queries_total"]
    M.run_case(None, 13, [FIXED[c] for c in "abcde"], ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz"], list("abcde"), [0] * world)


def _golden_job_over(lib, workload, device_lists):
    """A golden workload (tests/golden/<workload>.json: the oracle's md5) as ONE job through ac_compress_build_multi over each of the device
    lists; returns the multi info of each build."""
    import ctypes as C
    import bench
    from autocycler_amd import _capi, synth
    golden = json.loads((ROOT / "tests" / "golden" / f"{workload}.json").read_text())
    k, n_asm, gen = synth.WORKLOADS[workload]
    seqs, fn, hd = synth.flatten(gen())
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    h = bench.prepare(lib, k, seqs, fn, hd, n_asm, threads=32, repair=1)
    n = lib.ac_seqs_count(h)
    views = lib.ac_seqs_views(h)
    infos = []
    for devices in device_lists:
        g = C.c_void_p()
        dv = (C.c_int * len(devices))(*devices)
        assert lib.ac_compress_build_multi(C.c_uint32(k), C.c_uint32(n_asm), views, C.c_uint32(n), dv, C.c_int(len(devices)), C.byref(g)) == 0, lib.ac_last_error()
        gr = _capi.Graph(lib, g, n)
        mi = _capi.MultiInfo()
        lib.ac_multi_info_get(g, C.byref(mi))
        assert gr.stats_post["unitigs"] == golden["post"]["unitigs"]
        assert hashlib.md5(gr.gfa(fn, hd).encode()).hexdigest() == golden["gfa_md5"]
        infos.append(mi)
        gr.close()
    lib.ac_seqs_free(h)
    assert lib.ac_release_memory() == 0
    return infos


def test_config_b_over_two_ranks_digest(lib):
    """BASELINE configs[1] (12 x ~5 Mbp, k = 51) as one job over two ranks: the GFA has the oracle's md5 (tests/golden/configB_k51.json)."""
    _golden_job_over(lib, "configB_k51", ([0, 0], [0]))


def test_config_d_prime_k101_over_two_and_four_ranks_digest(lib):
    """D' (24 x ~10 Mbp, k = 101 — FOUR-word keys through every exchange: the routed walk-start keys are 32 bytes each) as
    one job over 2 and over 4 ranks sharing the device: the oracle's md5 (tests/golden/configDprime_k101.json)."""
    for mi in _golden_job_over(lib, "configDprime_k101", ([0, 0], [0, 0, 0, 0])):
        assert mi.n_ranks in (2, 4) and mi.queries_total > 0


def test_config_c_over_eight_ranks_digest(lib):
    """BASELINE configs[2] (96 x ~5 Mbp, k = 51) as one job over EIGHT ranks sharing the device (12 assemblies per rank: every rank's slice is
    new to it, the hardest case for the fragments): the oracle's whole-workload md5 (tests/golden/configC_k51.json)."""
    mi, = _golden_job_over(lib, "configC_k51", ([0] * 8,))
    assert mi.n_ranks == 8 and mi.bytes_received_max > 0


def _distinct_devices(lib):
    n = lib.ac_device_count()
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: the inter-device RCCL path needs at least two (it runs on the driver's multi-GPU node)")
    return list(range(min(n, 8)))


def test_distinct_devices_adversarial_over_rccl(lib, monkeypatch):
    """where the box has several GPUs, one rank per DISTINCT device — every exchange of the build is then an RCCL call
    between two devices (ncclAllReduce, grouped ncclSend / ncclRecv over xGMI), not the host-staged transport ranks sharing a device use."""
    devices = _distinct_devices(lib)
    monkeypatch.delenv("AC_MULTI_TRANSPORT", raising=False)
    for world in sorted({2, len(devices)}):
        dv = devices[:world]
        assert M.adversarial(None, dv, ks=(11, 51), seeds=range(8)) == 2 * 8 * 2
        seqs, fn, hd = M.synth_case(8, 60_000, 3_000, 1e-3, 1e-4, 7)
        gfa1, _ = M.run_case(None, 51, seqs, fn, hd, [0])
        gfa, info = M.run_case(None, 51, seqs, fn, hd, dv)
        assert gfa == gfa1 and info["transport"] == 2 and info["n_ranks"] == world      # 2 = RCCL
        seqs, fn, hd = M.mixed_case(world, 4, 40_000)
        _, info = M.run_case(None, 51, seqs, fn, hd, dv)
        assert info["transport"] == 2 and 0 < info["queries_sent_away"] < info["queries_total"]


def test_distinct_devices_golden_digests(lib, monkeypatch):
    """configs[1] and the mixed-species replica E' as one job over all the box's devices (RCCL): the oracle's md5."""
    devices = _distinct_devices(lib)
    monkeypatch.delenv("AC_MULTI_TRANSPORT", raising=False)
    for workload in ("configB_k51", "configEprime_k51"):
        for mi in _golden_job_over(lib, workload, (devices[:2], devices)):
            assert mi.transport == 2 and mi.n_ranks >= 2


def test_whole_command_over_several_ranks(lib, tmp_path):
    import subprocess
    import boundary_cases as B
    B.compress_dir_multi_matches_the_oracle(lib, tmp_path, 51, [0, 0])
    # and the CLI's --devices (an extension of the reference's flag surface)
    import oracle_lib as O
    cli = ROOT / "autocycler_amd" / "autocycler-compress"
    src = tmp_path / "asm_cli_multi"
    B.write_five_file_fixture(src)
    out_o, out_c = tmp_path / "o_cli_multi", tmp_path / "c_cli_multi"
    O.compress_dir(src, out_o, k=13)
    pr = subprocess.run([str(cli), "compress", "-i", str(src), "-a", str(out_c), "--kmer", "13", "--devices", "0,0,0"], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert (out_c / "input_assemblies.gfa").read_bytes() == (out_o / "input_assemblies.gfa").read_bytes()
    assert (out_c / "input_assemblies.yaml").read_text() == (out_o / "input_assemblies.yaml").read_text()


def test_mixed_species_replica_over_four_ranks_digest(lib):
    """E' (5 species x 20 strains x ~1 Mbp, 45 M distinct k-mers, 1.7 M unitigs) as one job over four ranks sharing the device: the owner-routed key
    exchange under load (each rank routes ~10^5 walk-start keys to three others) — the GFA has the oracle's md5."""
    mi, = _golden_job_over(lib, "configEprime_k51", ([0, 0, 0, 0],))
    assert mi.n_ranks == 4 and mi.queries_sent_away > mi.queries_total // 2
