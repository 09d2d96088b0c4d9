"""Boundary pieces of the C ABI that are not the graph build itself, as test bodies shared by the CPU suite (the serial emulation of
the kernels, tests/test_emu_boundary.py) and the device suite (libautocycler_hip.so on the MI355X, tests/test_gpu_boundary.py):
`Position` accessors (position.rs:18-46), the whole command `ac_compress_dir` / `autocycler-compress` on the reference's five-file
fixture set (tests.rs:131-148; flags main.rs:140-160), and `ac_decompress` on a GFA the library has just built (decompress.rs:83-105)."""
import ctypes as C
import gzip
import subprocess
from pathlib import Path

import oracle_lib as O
import parity_util
import seqgen
from autocycler_amd import _capi
from test_oracle_kats import FIXED

ROOT = Path(__file__).resolve().parent.parent
FIVE = [("a.fasta", "a"), ("b.fna", "b"), ("c.fa", "c"), ("d.fasta.gz", "d"), ("e.fna.gz", "e")]      # tests.rs:133-142


def positions_match_the_oracle(lib_path, cases):
    """ac_unitig_positions == the forward / reverse positions from_gfa_lines rebuilds (unitig_graph.rs:151-174), vector order included."""
    checked = 0
    for k, seqs, fn, hd in cases:
        g, gfa, loaded = parity_util.check_case(k, seqs, fn, hd, lib_path=lib_path)
        want = O.gfa_positions(gfa)
        assert len(want) == g.unitig_count
        for i in range(g.unitig_count):
            for fwd in (True, False):
                got = g.positions(i, fwd)
                assert got == want[i + 1][0 if fwd else 1], (k, i, fwd)
                checked += len(got)
        # Position invariants (position.rs:24-46): every occurrence on one strand implies the mirrored one on the other strand
        lens = {q["id"]: q["length"] for q in loaded}
        for i in range(g.unitig_count):
            n = len(g.unitig(i)[0])
            mirrored = sorted((sid, not strand, lens[sid] - n - pos) for sid, strand, pos in g.positions(i, True))
            assert mirrored == sorted(g.positions(i, False)), i
    assert checked > 0


def write_five_file_fixture(d):
    """The reference's five fixed sequences (tests.rs:133-142) as the five kinds of assembly file find_all_assemblies accepts."""
    d.mkdir(parents=True, exist_ok=True)
    for name, key in FIVE:
        body = f">{key}\n{FIXED[key]}\n".encode()
        (d / name).write_bytes(gzip.compress(body) if name.endswith(".gz") else body)


def compress_dir_matches_the_oracle(lib, tmp_path, k, device=0):
    """compress.rs:32-50 on the five-file fixture: same GFA bytes and the same YAML as the oracle's whole-command restatement;
    then `ac_decompress` of that very GFA gives the input files back (tests.rs:114-127)."""
    src = tmp_path / "asm"
    write_five_file_fixture(src)
    out_o, out_p = tmp_path / "o", tmp_path / "p"
    O.compress_dir(src, out_o, k=k)
    times = (C.c_double * 4)()
    rc = lib.ac_compress_dir(str(src).encode(), str(out_p).encode(), C.c_uint32(k), C.c_uint32(25), C.c_int(4), C.c_int(device), None, times)
    assert rc == 0, lib.ac_last_error()
    assert (out_p / "input_assemblies.gfa").read_bytes() == (out_o / "input_assemblies.gfa").read_bytes()
    assert (out_p / "input_assemblies.yaml").read_text() == (out_o / "input_assemblies.yaml").read_text()
    dec = tmp_path / "dec"
    assert lib.ac_decompress(str(out_p / "input_assemblies.gfa").encode(), str(dec).encode(), None, C.c_int(3)) == 0, lib.ac_last_error()
    for name, key in FIVE:
        got = (dec / name).read_bytes()
        assert (gzip.decompress(got) if name.endswith(".gz") else got) == f">{key}\n{FIXED[key]}\n".encode()
    return out_o


def compress_dir_many_pieces(lib, tmp_path, k=31, device=0, threads=48):
    """The whole command on a synthetic directory whose graph has hundreds of unitigs and links, written with many threads: the
    GFA is formatted in pieces (header, S lines by sequence bytes, L lines, P lines by path entries: gfa_chunks) that land in the
    file through parallel pwrite — byte for byte the oracle's file."""
    from autocycler_amd import synth
    src = tmp_path / "asm_many"
    synth.write_fasta_dir(synth.make_assemblies(5, genome=30_000, plasmid=2_000, sub=2e-3, indel=2e-4, seed=4242), str(src))
    out_o, out_p = tmp_path / "o_many", tmp_path / "p_many"
    O.compress_dir(src, out_o, k=k)
    rc = lib.ac_compress_dir(str(src).encode(), str(out_p).encode(), C.c_uint32(k), C.c_uint32(25), C.c_int(threads), C.c_int(device), None, None)
    assert rc == 0, lib.ac_last_error()
    want = (out_o / "input_assemblies.gfa").read_bytes()
    assert want.count(b"\nS\t") > 200 and want.count(b"\nL\t") > 200      # (enough lines for several S and L pieces)
    assert (out_p / "input_assemblies.gfa").read_bytes() == want
    assert (out_p / "input_assemblies.yaml").read_text() == (out_o / "input_assemblies.yaml").read_text()


def cli_matches_the_oracle(tmp_path, k):
    """The `autocycler-compress` binary (flag surface of main.rs:140-160) on the same fixture, as a fresh process."""
    cli = ROOT / "autocycler_amd" / "autocycler-compress"
    assert cli.exists(), "build the product first (make -C autocycler_amd/csrc)"
    src = tmp_path / "asm_cli"
    write_five_file_fixture(src)
    out_o, out_c = tmp_path / "o_cli", tmp_path / "c_cli"
    O.compress_dir(src, out_o, k=k)
    pr = subprocess.run([str(cli), "compress", "--assemblies_dir", str(src), "--autocycler_dir", str(out_c), "--kmer", str(k), "--threads", "2"],
                        capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert (out_c / "input_assemblies.gfa").read_bytes() == (out_o / "input_assemblies.gfa").read_bytes()
    assert (out_c / "input_assemblies.yaml").read_text() == (out_o / "input_assemblies.yaml").read_text()
    assert "unitigs" in pr.stderr and "Graph contains" in pr.stderr          # the statistics compress.rs:152,165,177 prints
    dec = tmp_path / "dec_cli"
    pr = subprocess.run([str(cli), "decompress", "-i", str(out_c / "input_assemblies.gfa"), "-o", str(dec)], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-2000:]
    for name, key in FIVE:
        got = (dec / name).read_bytes()
        assert (gzip.decompress(got) if name.endswith(".gz") else got) == f">{key}\n{FIXED[key]}\n".encode()
    # user errors leave through exit code 1 with the reference's message shape (misc.rs:131-137)
    pr = subprocess.run([str(cli), "compress", "-i", str(tmp_path / "nowhere"), "-a", str(out_c)], capture_output=True, text=True, timeout=60)
    assert pr.returncode == 1 and "Error: directory does not exist" in pr.stderr


def compress_dir_multi_matches_the_oracle(lib, tmp_path, k, devices):
    """The whole command over several ranks (ac_compress_dir_multi: host loader + host end repair + ac_compress_build_multi): the oracle's
    GFA and YAML for the five-file fixture and for a synthetic directory."""
    from autocycler_amd import synth
    src = tmp_path / "asm_multi"
    write_five_file_fixture(src)
    src2 = tmp_path / "asm_multi2"
    synth.write_fasta_dir(synth.make_assemblies(6, genome=30_000, plasmid=2_000, sub=2e-3, indel=2e-4, seed=99), str(src2))
    dv = (C.c_int * len(devices))(*devices)
    for n, d in enumerate((src, src2)):
        out_o, out_p = tmp_path / f"o_multi{n}", tmp_path / f"p_multi{n}"
        O.compress_dir(d, out_o, k=k)
        rc = lib.ac_compress_dir_multi(str(d).encode(), str(out_p).encode(), C.c_uint32(k), C.c_uint32(25), C.c_int(4), dv, C.c_int(len(devices)), None, None)
        assert rc == 0, lib.ac_last_error()
        assert (out_p / "input_assemblies.gfa").read_bytes() == (out_o / "input_assemblies.gfa").read_bytes()
        assert (out_p / "input_assemblies.yaml").read_text() == (out_o / "input_assemblies.yaml").read_text()


def two_device_ordinals(lib_path):
    """the process-wide device arena belongs to the device of the previous call; a call that names another ordinal
    must not run on its blocks.  (The emulation ignores the ordinal itself but runs the same release path.)"""
    k = 21
    seqs, fn, hd = seqgen.make_case(4, k)
    g0, gfa0, _ = parity_util.check_case(k, seqs, fn, hd, lib_path=lib_path, device=0)
    lib = _capi.load_library(lib_path)
    n_dev = lib.ac_device_count()
    other = 1 if (n_dev == 0 or n_dev > 1) else None      # emulation (0 devices): any ordinal; one GPU: nothing else to name
    if other is not None:
        g1, gfa1, _ = parity_util.check_case(k, seqs, fn, hd, lib_path=lib_path, device=other)
        assert gfa1 == gfa0
    g2, gfa2, _ = parity_util.check_case(k, seqs, fn, hd, lib_path=lib_path, device=0)
    assert gfa2 == gfa0 and g0.stats_post == g2.stats_post


def build_c_client(out_dir, lib_dir=None):
    """Compiles tests/c_client/client.c — a pedantic C99 translation unit including only include/autocycler_hip.h — against the
    product library: the header is what a foreign-function binding sees, so it must be valid C and every symbol must link."""
    lib_dir = Path(lib_dir) if lib_dir else ROOT / "autocycler_amd"
    exe = Path(out_dir) / "c_client"
    subprocess.check_call(["gcc", "-std=c99", "-D_GNU_SOURCE", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", str(ROOT / "include"),
                           str(ROOT / "tests" / "c_client" / "client.c"), "-o", str(exe), "-L", str(lib_dir), "-lautocycler_hip",
                           f"-Wl,-rpath,{lib_dir}"])
    return exe


def c_client_matches_the_oracle(tmp_path, k, seqs, assembly_count):
    """The C client on the device: same GFA as the oracle for the same padded, end-repaired sequences."""
    exe = build_c_client(tmp_path)
    fn = [f"f{i}" for i in range(len(seqs))]; hd = [f"h{i}" for i in range(len(seqs))]
    s = O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd, repair=True, assembly_count=assembly_count)
    gfa_o, st, _ = s.compress(k)
    text = "".join(f"{q['id']} {q['length']} {q['fwd'].decode()}\n" for q in s.all())
    pr = subprocess.run([str(exe), str(k), str(assembly_count), "0"], input=text, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert pr.stdout == gfa_o
    assert f"kmers {st['kmers']} pre {st['unitigs_pre']} {st['links_pre']} {st['length_pre']} post {st['unitigs_post']} {st['links_post']} {st['length_post']}" in pr.stderr


def host_pack_simd_equals_scalar(lib_path):
    """K1 on the host (the packed upload of ac_compress_build): the AVX2 / BMI2 loop == the portable loop == the definition, on
    bytes of every kind (bases, dots, separators, lower case, arbitrary), at lengths around the 32-byte groups."""
    import ctypes as C
    import numpy as np
    from autocycler_amd import _capi
    lib = _capi.load_library(lib_path)
    rng = np.random.default_rng(5)
    for n in (1, 31, 32, 33, 64, 1000, 4096 + 17, 100_003):
        pool = np.frombuffer(b"ACGTACGTACGTACGT.$acgtN\x00\xff", dtype=np.uint8)
        text = pool[rng.integers(0, len(pool), size=n)].copy()
        g = (n + 31) // 32
        outs = []
        for scalar in (0, 1, 0):      # the third output starts 8 bytes off a 16-byte boundary (the streaming stores' prologue)
            bits = np.zeros(g + 3, dtype=np.uint64); mask = np.zeros(g, dtype=np.uint32)
            shift = (bits.ctypes.data // 8 + len(outs) // 2) % 2
            bits = bits[shift:shift + g]
            assert lib.ac_pack_text(text.ctypes.data_as(C.c_void_p), C.c_uint64(n), bits.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                                    C.c_int(scalar)) == 0
            outs.append((bits, mask))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1])
        padded = np.concatenate([text, np.full(g * 32 - n, ord("$"), dtype=np.uint8)]).reshape(g, 32)
        good = np.isin(padded, np.frombuffer(b"ACGT", dtype=np.uint8))
        code = np.where(good, ((padded >> 1) ^ (padded >> 2)) & 3, 0).astype(np.uint64)
        want_bits = (code << (62 - 2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
        want_mask = ((~good).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
        assert np.array_equal(outs[0][0], want_bits) and np.array_equal(outs[0][1], want_mask)


def foreign_bytes_are_rejected(lib_path, device=0):
    """sequence.rs:39-41: a Sequence holds A, C, G, T only (quit_with_error "... contains non-ACGT characters"), and its padding dots
    sit at its two ends (sequence.rs:44-46).  Every entry that takes sequences or a text from the caller must refuse anything else —
    an N, a lower-case base, a dot or a separator inside a sequence, a sequence table whose dot counts do not match the text — with
    that message instead of building a graph from it: the host entry (its packers classify every byte), the device entry and the first
    phase of a sharded build (K1 checks every non-base byte against the sequence table)."""
    import numpy as np
    lib = _capi.load_library(lib_path)
    k = 11
    rng = np.random.default_rng(5)
    base = ["".join("ACGT"[c] for c in rng.integers(0, 4, size=n)) for n in (700, 333, 90)]

    def host(seqs):
        views = (_capi.SeqView * len(seqs))()
        keep = [("." * 5 + s + "." * 5).encode() for s in seqs]
        for i, b in enumerate(keep):
            views[i].fwd, views[i].length, views[i].id = b, len(seqs[i]), i + 1
        g = C.c_void_p()
        rc = lib.ac_compress_build(C.c_uint32(k), C.c_uint32(len(seqs)), views, C.c_uint32(len(seqs)), C.c_int(device), C.byref(g))
        if rc == 0:
            lib.ac_free(g)
        return rc, lib.ac_last_error().decode()

    def dev(seqs, d1=None, d2=None, shard=False):
        text = ("$" + "$".join("." * 5 + s + "." * 5 for s in seqs) + "$").encode()
        n = len(seqs)
        off, p = [], 1
        for s in seqs:
            off.append(p); p += len(s) + 10 + 1
        if str(lib_path).endswith("libautocycler_emu.so"):
            buf = np.frombuffer(text, dtype=np.uint8).copy(); ptr = buf.ctypes.data
        else:
            import torch
            buf = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(f"cuda:{device}"); ptr = buf.data_ptr()
        args = [C.c_uint32(k), C.c_uint32(n), C.c_void_p(ptr), C.c_uint64(len(text)), (C.c_uint64 * n)(*off), (C.c_uint32 * n)(*[len(s) for s in seqs]),
                (C.c_uint16 * n)(*range(1, n + 1)), (C.c_uint16 * n)(*(d1 or [5] * n)), (C.c_uint16 * n)(*(d2 or [5] * n)), C.c_uint32(n), C.c_int(device)]
        g = C.c_void_p()
        if shard:
            rc = lib.ac_shard_begin(*args, C.byref(g))
            if rc == 0:
                lib.ac_shard_free(g)
        else:
            rc = lib.ac_compress_build_device(*args, C.byref(g))
            if rc == 0:
                lib.ac_free(g)
        return rc, lib.ac_last_error().decode()

    assert host(base)[0] == 0 and dev(base)[0] == 0 and dev(base, shard=True)[0] == 0
    for which, at, ch in ((0, 350, "N"), (1, 0, "N"), (2, 89, "n"), (1, 100, "a"), (0, 5, "."), (2, 40, "$"), (1, 332, "-"), (0, 699, "\x00")):
        bad = list(base)
        bad[which] = bad[which][:at] + ch + bad[which][at + 1:]
        want = f"input sequence {which + 1} contains non-ACGT characters"
        for rc, msg in (host(bad), dev(bad), dev(bad, shard=True)):
            assert rc != 0 and msg == want, (which, at, ch, msg)
    # a table that claims fewer / more padding dots than the text holds
    for d1, d2 in (([4, 5, 5], None), (None, [5, 5, 3])):
        rc, msg = dev(base, d1=d1, d2=d2)
        assert rc != 0 and "non-ACGT" in msg, msg
    # end-repaired sequences (bases where padding was) are fine when the table says so
    rep = list(base)
    text_ok = dev(rep)[0]
    assert text_ok == 0
