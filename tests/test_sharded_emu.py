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


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return str(p)


def launch(world, lib_path, device, cases, timeout=600):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "sharded_worker.py"), str(r), str(world), port, str(lib_path),
                               device, cases], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
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
    cases = ",".join(f"{k}:{seed}" for k in (5, 11, 31, 51) for seed in range(24)) + ",synth:51,synth:21"
    launch(world, emu, "cpu", cases)
