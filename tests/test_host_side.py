"""Host logic around the hot path (CPU): loader, end repair, whole-command driver, YAML, C-ABI surface."""
import ctypes as C
import gzip
import re
from pathlib import Path

import pytest

import emu_lib
import oracle_lib as O
import seqgen
from autocycler_amd import _capi

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def emu():
    lib = _capi.load_library(emu_lib.emu_path())
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    return lib


def test_source_hash_matches_the_tree():
    # bench.py quotes PMC traffic only when it was counted on a library built from exactly these sources
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("source_hash", Path(__file__).resolve().parent.parent / "tools" / "source_hash.py")
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    # (the emulation is rebuilt by emu_lib whenever a source is newer than it; the product library is whatever build() left,
    # so only its format is checked here — tests/test_gpu_boundary.py compares it with the tree on the device box)
    assert _capi.load_library(emu_lib.emu_path()).ac_source_hash().decode() == m.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", _capi.load_library().ac_source_hash().decode())


def test_lockstep_emulation_of_the_wave_primitives(tmp_path):
    """autocycler_amd/csrc/wave_rt.hpp under AC_EMU — what lets the CPU suite run the SAME wave kernels the device runs — checked on its own:
    ballots, shuffles, a barrier over LDS, diverging lane groups, lanes that return early, a reported deadlock."""
    import subprocess
    # (twice: the hand-written x86-64 context switch, and the portable <ucontext.h> fallback other hosts get: -DAC_EMU_UCONTEXT)
    for extra in ([], ["-DAC_EMU_UCONTEXT"]):
        exe = tmp_path / ("wave_rt_check" + ("_uc" if extra else ""))
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", *extra, "-I", str(ROOT / "autocycler_amd" / "csrc"), str(ROOT / "tests" / "c_client" / "wave_rt_check.cpp"), "-o", str(exe)])
        out = subprocess.run([str(exe)], capture_output=True, text=True)
        assert out.returncode == 0 and "wave_rt_check: OK" in out.stdout, out.stdout[-2000:]
        assert "overran its stack" in out.stdout


def test_header_symbols_exported_by_product_library():
    """The C-ABI library loads (no GPU needed) and exports every function include/autocycler_hip.h declares."""
    header = (ROOT / "include" / "autocycler_hip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)   # prototypes only
    declared = set(re.findall(r"\b(ac_[a-z0-9_]+)\s*\(", header))
    assert {"ac_compress_build", "ac_compress_build_device", "ac_links", "ac_path", "ac_compress_dir"} <= declared
    lib = _capi.load_library()      # raises HipLibraryMissing if the extension was not built
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(_capi.EXPORTS) <= declared
    assert b"gfx950" in lib.ac_version()


def test_header_is_plain_c_and_links(tmp_path):
    """include/autocycler_hip.h compiles as pedantic C99 and a C client links against the product library (no GPU needed to link);
    run without a device it fails with the library's error text and exit code 1, the way the Rust shim would call quit_with_error."""
    import subprocess
    import boundary_cases as B
    import torch
    exe = B.build_c_client(tmp_path)
    if not torch.cuda.is_available():
        pr = subprocess.run([str(exe), "9", "1", "0"], input="1 14 ....ACGTACGTACGTAC....\n", capture_output=True, text=True, timeout=120)
        assert pr.returncode == 1 and "Error:" in pr.stderr and ("no HIP device" in pr.stderr or "HIP error" in pr.stderr or "no CPU fallback" in pr.stderr), pr.stderr


def test_product_library_has_no_cpu_fallback():
    """Without a GPU the build call must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from autocycler_amd import AutocyclerError, compress_build
    with pytest.raises(AutocyclerError, match="no HIP device|no CPU fallback|HIP error"):
        compress_build(9, 1, [(b"....ACGTACGTACGTAC....", 14, 1)])


def _prepare(lib, k, seqs, fn, hd, repair):
    n = len(seqs)
    bufs = [s.encode() for s in seqs]
    out = C.c_void_p()
    rc = lib.ac_seqs_from_raw(C.c_uint32(k), C.c_uint32(n), (C.c_char_p * n)(*bufs), (C.c_uint32 * n)(*[len(b) for b in bufs]),
                              (C.c_char_p * n)(*[x.encode() for x in fn]), (C.c_char_p * n)(*[x.encode() for x in hd]),
                              C.c_uint32(len(set(fn))), C.c_int(repair), C.c_int(3), C.byref(out))
    assert rc == 0, lib.ac_last_error()
    v = lib.ac_seqs_views(out)
    res = [C.string_at(v[i].fwd, v[i].length + k - 1) for i in range(n)]
    lib.ac_seqs_free(out)
    return res


@pytest.mark.parametrize("k", [3, 5, 11, 21, 51])
def test_end_repair_matches_reference_semantics(emu, k):
    """Indexed one-pass end repair == the literal regex scan of compress.rs:202-270 (oracle)."""
    for seed in range(40):
        seqs, fn, hd = seqgen.make_case(seed, k)
        want = [q["fwd"] for q in O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd, repair=True).all()]
        assert _prepare(emu, k, seqs, fn, hd, 1) == want


def _write_dir(d, k):
    seqs, fn, hd = seqgen.make_case(10, k)
    files = {}
    for s, f, h in zip(seqs, fn, hd):
        files.setdefault(f, []).append((h, s))
    names = []
    for i, (f, recs) in enumerate(sorted(files.items())):
        text = "".join(f">{h}\t extra  words\n{s[:len(s)//2]}\n{s[len(s)//2:].lower()}\n" for h, s in recs)
        if i == 0:
            text += ">tiny\nACG\n>skipme autocycler_ignore\n" + "ACGT" * (k) + "\n"
        name = f if i % 2 == 0 else f + ".gz"
        if name.endswith(".gz"):
            with gzip.open(d / name, "wt") as fh:
                fh.write(text)
        else:
            (d / name).write_text(text)
        names.append(name)
    (d / "notes.txt").write_text("not an assembly\n")
    return names


@pytest.mark.parametrize("k", [11, 51])
def test_compress_dir_matches_oracle(emu, tmp_path, k):
    """compress.rs:32-50 end to end: same GFA bytes and same YAML as the oracle's whole-command restatement."""
    src = tmp_path / "asm"
    src.mkdir()
    _write_dir(src, k)
    out_o, out_p = tmp_path / "o", tmp_path / "p"
    O.compress_dir(src, out_o, k=k)
    times = (C.c_double * 4)()
    rc = emu.ac_compress_dir(str(src).encode(), str(out_p).encode(), C.c_uint32(k), C.c_uint32(25), C.c_int(4), C.c_int(0), None, times)
    assert rc == 0, emu.ac_last_error()
    assert (out_p / "input_assemblies.gfa").read_bytes() == (out_o / "input_assemblies.gfa").read_bytes()
    assert (out_p / "input_assemblies.yaml").read_text() == (out_o / "input_assemblies.yaml").read_text()
    # the reference's own properties (tests.rs:108-127) on the product's file
    g = (out_p / "input_assemblies.gfa").read_text()
    assert O.gfa_resave(g) == g
    assert len(O.decompress(g)) == len([l for l in g.splitlines() if l.startswith("P")])


def test_compress_dir_user_errors(emu, tmp_path):
    def run(src, out, k=51, threads=8):
        rc = emu.ac_compress_dir(str(src).encode(), str(out).encode(), C.c_uint32(k), C.c_uint32(25), C.c_int(threads), C.c_int(0), None, None)
        return rc, emu.ac_last_error().decode()
    src = tmp_path / "asm"
    src.mkdir()
    assert run(tmp_path / "missing", tmp_path / "o")[1].startswith("directory does not exist")
    assert "no assemblies found" in run(src, tmp_path / "o")[1]
    (src / "a.fasta").write_text(">a\n" + "ACGT" * 30 + "\n")
    assert run(src, tmp_path / "o", k=9)[1] == "--kmer cannot be less than 11"
    assert run(src, tmp_path / "o", k=503)[1] == "--kmer cannot be greater than 501"
    assert run(src, tmp_path / "o", k=52)[1] == "--kmer must be odd"
    assert run(src, tmp_path / "o", threads=0)[1] == "--threads cannot be less than 1"
    (src / "b.fasta").write_text(">b\nACGTNNACGT" * 20 + "\n")
    assert "contains non-ACGT characters" in run(src, tmp_path / "o", k=11)[1]
    (src / "b.fasta").write_text("")
    assert "is an empty file" in run(src, tmp_path / "o", k=11)[1]


@pytest.mark.parametrize("k", [5, 11, 51])
def test_gfa_load_save_identity_and_decompress(k):
    """The reference's round-trip properties (tests.rs:108-127) through the library's own loader: save -> load -> save is the
    identity, every path spells its input sequence, and the loaded graph gives the oracle's pairwise distances."""
    from autocycler_amd import graph_from_gfa
    path = emu_lib.emu_path()
    for seed in range(24):
        seqs, fn, hd = seqgen.make_case(seed, k)
        s = O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd)
        gfa, _, _ = s.compress(k)
        g, fns, hds = graph_from_gfa(gfa, lib_path=path)
        assert g.gfa(fns, hds) == gfa
        assert [g.decompress(i).decode() for i in range(len(seqs))] == seqs
        assert g.pairwise_distances() == O.pairwise_distances(gfa)
        g.close()


def test_gfa_loader_errors():
    from autocycler_amd import AutocyclerError, graph_from_gfa
    path = emu_lib.emu_path()
    with pytest.raises(AutocyclerError, match="depth tag"):
        graph_from_gfa("H\tVN:Z:1.0\tKM:i:9\nS\t1\tACGT\n", lib_path=path)
    with pytest.raises(AutocyclerError, match="non-zero overlap"):
        graph_from_gfa("H\tVN:Z:1.0\tKM:i:9\nS\t1\tACGT\tDP:f:1.00\nL\t1\t+\t1\t+\t5M\n", lib_path=path)
    with pytest.raises(AutocyclerError, match="nonexistent unitig"):
        graph_from_gfa("H\tVN:Z:1.0\tKM:i:9\nS\t1\tACGT\tDP:f:1.00\nL\t1\t+\t2\t+\t0M\n", lib_path=path)
    with pytest.raises(AutocyclerError, match="missing required tag"):
        graph_from_gfa("H\tVN:Z:1.0\tKM:i:9\nS\t1\tACGT\tDP:f:1.00\nP\t1\t1+\t*\tLN:i:4\n", lib_path=path)
    with pytest.raises(AutocyclerError, match="mismatch"):
        graph_from_gfa("H\tVN:Z:1.0\tKM:i:9\nS\t1\tACGT\tDP:f:1.00\nP\t1\t1+\t*\tLN:i:5\tFN:Z:a.fasta\tHD:Z:c\n", lib_path=path)


def test_decompress_command_reproduces_the_input_files(tmp_path):
    """tests.rs:114-127 through the library: compress a directory (oracle), decompress the GFA (library) -> the same files
    (plain ones byte for byte, gzipped ones after gunzip); plus the single-file output format of decompress.rs:117-137."""
    import gzip
    lib = _capi.load_library(emu_lib.emu_path())
    src = tmp_path / "in"; src.mkdir()
    files = {"a.fasta": [("c1 some text", "ACGTTGCAAGGCTTACGATCGATCGGATCGATTAGC"), ("c2", "TTGACCGATGCATGCATGGGATCAAC")],
             "b with space.fna": [("x", "GGATCGATTAGCACGTTGCAAGGCTTACGATCGATC")],
             "c.fa.gz": [("z 1", "ACGTTGCAAGGCTTACGATCGATCGGATCGATTAGCAAA")]}
    for name, recs in files.items():
        body = "".join(f">{h}\n{s}\n" for h, s in recs).encode()
        (src / name).write_bytes(gzip.compress(body) if name.endswith(".gz") else body)
    out = tmp_path / "ac"
    O.compress_dir(src, out, k=11)
    dec = tmp_path / "dec"
    one = tmp_path / "all.fasta"
    assert lib.ac_decompress(str(out / "input_assemblies.gfa").encode(), str(dec).encode(), str(one).encode(), C.c_int(3)) == 0, lib.ac_last_error()
    for name, recs in files.items():
        body = "".join(f">{h}\n{s}\n" for h, s in recs).encode()
        got = (dec / name).read_bytes()
        assert (gzip.decompress(got) if name.endswith(".gz") else got) == body
    want = "".join(f">{name.replace(' ', '_')}__{h}\n{s}\n" for name, recs in sorted(files.items()) for h, s in recs)
    assert one.read_text() == want
    assert lib.ac_decompress(str(out / "missing.gfa").encode(), str(dec).encode(), None, C.c_int(1)) != 0


def test_ab_knobs_tool_on_the_emulation():
    # tools/ab_knobs.py (the torch-free A/B driver of the tuning knobs) stays runnable: every variant reports the same graph digest
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "tools" / "ab_knobs.py"), "--emu", "--assemblies", "4", "--genome", "30000", "--steps", "1",
                          "--variants", "base;AC_TABLE_SHIFT=0,AC_MINKEY_VARIANT=0;AC_PATH_FILTER=0;AC_INSERT_ADAPT=0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    variants = [r for r in rows if "variant" in r]
    assert len(variants) == 4 and all("error" not in r for r in variants), variants
    assert len({r["gfa_md5"] for r in variants}) == 1
    assert variants[0]["table_capacity"] == 2 * variants[1]["table_capacity"]


def _bench_dry_run(extra_args, launcher=(), env_extra=None, timeout=900):
    """bench.py's own plumbing (rank layout, gloo collectives, the JSON line) on the CPU emulation of the kernels
    (BENCH_EMU_LIB): the numbers mean nothing, the line's shape is what the driver parses."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    import emu_lib
    root = Path(__file__).resolve().parent.parent
    env = {**os.environ, "BENCH_EMU_LIB": str(emu_lib.emu_path()), **(env_extra or {})}
    cmd = [sys.executable, *launcher, str(root / "bench.py"), "--assemblies", "3", "--genome", "30000", "--plasmid", "1500",
           "--steps", "2", "--warmup", "1", *extra_args]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(root))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


CONTRACT_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline"]


def test_bench_line_contract_single_rank_dry_run():
    j = _bench_dry_run(["--cpu-sample", "2x20000"], env_extra={"BENCH_EMU_ASSUME_DEFAULT": "1"})
    for key in CONTRACT_KEYS + ["cpu_baseline"]:
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["unit"] == "Mbp/s" and j["value"] > 0 and j["vs_baseline"] is None and j["dtype"] == "u64"
    assert "workload" in j["config"] and "model" not in j["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in j["roofline"], key
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["peak"] == 8000.0 and j["roofline"]["traffic"] > 0
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in j["cpu_baseline"], key
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1
    assert [o["kernel"].split("<")[0] for o in j["roofline_other"]] == ["PathWalkFunctor", "expand_wave_kernel"]
    # a fraction of the HBM peak from MEASURED bytes, a stated lower-bound model next to it, the binding ceiling by name
    for r in [j["roofline"]] + j["roofline_other"]:
        for key in ("needed_bytes", "needed_model", "frac_needed", "waste", "ceiling", "frac_source", "kernel_ms"):
            assert key in r, key
        assert r["needed_bytes"] > 0 and r["waste"] > 0 and r["traffic"] > 0 and "PMC" in r["frac_source"]
        assert abs(r["frac"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 8e12) < 1e-9 * max(1.0, r["frac"])
    assert j["roofline"]["ceiling"] is None or j["roofline"]["ceiling"]["name"] == "cas"
    assert j["roofline"]["whole_path_equiv"]["B_per_bp"] == 49
    # cpu_baseline.value is what was TIMED IN THIS RUN (a bounded sample, this host's cores); the whole workload's figure, recorded
    # once elsewhere, is context under its own key, with the golden digest it belongs to; the GPU path's host thread count stands beside it
    assert j["cpu_baseline"]["value"] > 0 and "sample" in j["cpu_baseline"] and j["cpu_baseline"]["host_threads_in_t_hot"] >= 1
    rec = j["cpu_baseline"]["recorded_whole_workload"]
    assert rec["gfa_md5"] == "c28d41ea9e4784f5f1dd06da6b3eb587" and 0.5 < rec["value"] < 1.0 and rec["cores"] == 1
    # the median and the slowest step beside the bracket's mean; launches and host round trips of a build
    assert j["ms_per_step_median"] > 0 and j["ms_per_step_max"] >= j["ms_per_step_median"] and j["value_from_median_step"] > 0
    assert "launches_per_build" in j and "host_round_trips_per_build" in j
    assert "dry run" in j["data"]
    # the headline is SURVEY.md 8(d)'s T_hot bracket (host RAM -> host RAM); the device-resident one is an extra key
    assert "T_hot" in j["config"]["timed_region"] and j["value"] == j["t_hot"]["value"] and j["ms_per_step"] == j["t_hot"]["ms_per_step"]
    assert j["hbm_resident"]["value"] > 0 and j["hbm_resident"]["steps"] == 2 and "RESIDENT IN HBM" in j["hbm_resident"]["timed_region"]
    assert j["untimed_builds_before_the_timed_steps"]["init"] == j["init_builds_run"] >= 2
    assert len(j["roofline"]["library_source_hash"]) == 16
    # the host-RAM -> host-RAM bracket of the same region (SURVEY.md 8d T_hot), timed in the same run through ac_compress_build
    for key in ("value", "ms_per_step", "timed_region", "upload_ms", "gfa_md5"):
        assert key in j["t_hot"], key
    assert j["t_hot"]["same_graph_as_device_entry"] is True and j["t_hot"]["steps"] == 2
    assert j["cold_first_build_ms"] > 0 and "configs[2]" not in j["config"]["workload"]      # 3 x 30 kbp is not a BASELINE configuration


def test_bench_line_two_ranks_dry_run():
    # the N > 1 path of bench.py as the driver launches it (torch.distributed.run, one process per rank): ONE mixed-species job
    # sharded over the ranks, then the same ranks as independent jobs; collectives over gloo here, RCCL on the GPU box
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    j = _bench_dry_run(["--gpus", "2"], launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                   "--master-port", str(port)))
    for key in CONTRACT_KEYS + ["sharded", "independent_jobs"]:
        assert key in j, key
    assert j["n_gpus"] == 2 and j["config"]["mode"] == "sharded" and "2 species" in j["config"]["workload"]
    assert j["value"] > 0 and j["independent_jobs"]["value"] > 0
    assert j["sharded"]["unitigs"] > 0 and j["sharded"]["fragments"] >= j["sharded"]["fragments_rank0"]
    assert "cpu_baseline" not in j          # rank 0 at N = 1 only
