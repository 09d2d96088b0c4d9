"""Pins the CPU oracle against every known-answer test the reference's own test-suite holds for the
compress path.  Each test names the reference test it reproduces
(file:line relative to /root/reference/src).  CPU only."""
import gzip
from pathlib import Path

import pytest

import oracle_lib as O

GOLDEN = Path(__file__).parent / "golden"
SEQ20 = "ACGACTGACATCAGCACTGC"

FIXED = {  # tests.rs:133-142
    "a": "CTTATGAGCAGTCCTTAACGTAGCGGTGTGTGGCTTTGAGAAGTTAGCGGTGGCGAGCTACATCCTGGCTCCAAT",
    "b": "ACCGTTACGTTAAGGACTGCTCATAAGATTGGAGCCAGGATGTAGCTCGCCACGGCTAACTTCTCAAAGCGGCAC",
    "c": "CATCCTGGCTCCAATCTTATGAGCAGTCCTTAACGTAACGGTGTGTGGCTTTGAGAAGTTAGCCGTGGCGAGATA",
    "d": "GGACTGCTCATAAGATTGGAGCCAGGATGTAGCTCGCCACGGCTAACTTCTCAAAGCCACACACCGTTACGTTAA",
    "e": "TTGAGAAGTTAGCCGTGGCGAGCTACATCCTGGCTCCAATCTTATGAGCAGTCCTTAACGTAACGGTGTGTGGCC",
}


def gfa(n):
    return (GOLDEN / f"test_gfa_{n}.gfa").read_text()


# kmer_graph.rs:189-197 test_kmer
def test_kmer_display():
    assert O.kat_kmer_display() == "ACGA:1+123,2-456"


# kmer_graph.rs:199-212 test_kmer_graph + :266-282 test_iterate_kmers
def test_kmer_graph_40_kmers_sorted():
    expected = ["..ACG", "..GCA", ".ACGA", ".GCAG", "ACATC", "ACGAC", "ACTGA", "ACTGC", "AGCAC", "AGTCG",
                "AGTGC", "ATCAG", "ATGTC", "CACTG", "CAGCA", "CAGTC", "CAGTG", "CATCA", "CGACT", "CGT..",
                "CTGAC", "CTGAT", "CTGC.", "GACAT", "GACTG", "GATGT", "GCACT", "GCAGT", "GCTGA", "GTCAG",
                "GTCGT", "GTGCT", "TCAGC", "TCAGT", "TCGT.", "TGACA", "TGATG", "TGC..", "TGCTG", "TGTCA"]
    got = O.kat_kmers(SEQ20, 5)
    assert len(got) == 40
    assert [g.split(":")[0] for g in got] == expected
    # Position bookkeeping (kmer_graph.rs:103-132): first forward k-mer at +0, its RC at reverse pos L-1.
    d = dict(g.split(":") for g in got)
    assert d["..ACG"] == "1+0" and d["CGT.."] == "1-19" and d["..GCA"] == "1-0" and d["TGC.."] == "1+19"


# kmer_graph.rs:214-238 test_next_kmers
def test_next_kmers():
    assert O.kat_neighbours(SEQ20, 5, "ACATC", True) == ["CATCA"]
    assert O.kat_neighbours(SEQ20, 5, "CACTG", True) == ["ACTGA", "ACTGC"]
    assert O.kat_neighbours(SEQ20, 5, "ACTGA", True) == ["CTGAC", "CTGAT"]
    assert O.kat_neighbours(SEQ20, 5, "AAAAA", True) == []


# kmer_graph.rs:240-263 test_prev_kmers
def test_prev_kmers():
    assert O.kat_neighbours(SEQ20, 5, "CATCA", False) == ["ACATC"]
    assert O.kat_neighbours(SEQ20, 5, "CTGAC", False) == ["ACTGA", "GCTGA"]
    assert O.kat_neighbours(SEQ20, 5, "ACTGC", False) == ["CACTG", "GACTG"]
    assert O.kat_neighbours(SEQ20, 5, "AAAAA", False) == []


# position.rs:64-72 test_position
def test_position():
    assert O.kat_position(1, True, 123) == "1+123"
    assert O.kat_position(2, False, 456) == "2-456"
    assert O.kat_position(32767, True, 4294967295) == "32767+4294967295"


# unitig.rs:411-441 test_from_kmers
def test_unitig_from_kmers_trim():
    assert O.kat_unitig_from_kmers() == ["7", "GCATAGC", "GCTATGC", "3", "ATA", "TAT"]


# unitig.rs:458-556 shift primitives
def test_shift_primitives():
    assert O.kat_shift("remove_start", 2) == ["TGAAGGGC", "GCCCTTCA", "102", "890", "202", "790"]
    assert O.kat_shift("remove_end", 2) == ["GCTGAAGG", "CCTTCAGC", "100", "892", "200", "792"]
    assert O.kat_shift("add_start", "AC") == ["ACGCTGAAGGGC", "GCCCTTCAGCGT", "98", "890", "198", "790"]
    assert O.kat_shift("add_end", "AC") == ["GCTGAAGGGCAC", "GTGCCCTTCAGC", "100", "888", "200", "788"]


# compress.rs:281-344 test_find_best_match_1/2
def test_find_best_match():
    f = O.find_best_match
    assert f(["...ACGT"]) == "...ACGT"
    assert f(["...ACGT", "..GACGT"]) == "..GACGT"
    assert f(["..GACGT", "...ACGT"]) == "..GACGT"
    assert f(["...GAAA", "...CAAA", "...TAAA"]) == "...CAAA"
    assert f(["...ACGT", "..GACGT", "..CACGT", "..GACGT", "..CACGT"]) == "..CACGT"
    assert f(["...ACGT", "..GACGT", "..GACGT", ".AGACGT", ".CGACGT"]) == ".AGACGT"
    assert f(["...ACGT", ".CGACGT", "..GACGT", ".AGACGT", ".CGACGT"]) == ".CGACGT"
    assert f(["ACGT..."]) == "ACGT..."
    assert f(["ACGT...", "ACGTT.."]) == "ACGTT.."
    assert f(["GAAA...", "CAAA...", "TAAA..."]) == "CAAA..."
    assert f(["CACG...", "GACGT..", "CACGT..", "GACGT..", "CACGT.."]) == "CACGT.."
    assert f(["AGAC...", "AGACG..", "AGACG..", "AGACGT.", "CGACGT."]) == "AGACGT."


# compress.rs:346-369 test_load_sequences_1/2
def test_load_sequences_counts_and_dup_name(tmp_path):
    d = tmp_path / "ok"
    d.mkdir()
    (d / "a.fasta").write_text(">a1\nACGT\n")
    (d / "b.fasta").write_text(">b1\nACGT\n>b2\nACGT\n")
    (d / "c.fasta").write_text(">c1\nACGT\n>c2\nACGT\n>c3\nACGT\n")
    s = O.Seqs.from_dir(d, 3)
    assert len(s) == 6 and s.assembly_count == 3
    (d / "c.fasta").write_text(">c1\nACGT\n>c1\nACGT\n>c3\nACGT\n")
    with pytest.raises(O.OracleError, match="duplicate name"):
        O.Seqs.from_dir(d, 3)


# tests.rs:170-188 test_whitespace
def test_whitespace_and_padding(tmp_path):
    (tmp_path / "assembly.fasta").write_text(">name abc  def\tghi\nCTTATGAGCAGTCCTTAACGTAGCGGT\n")
    s = O.Seqs.from_dir(tmp_path, 11)
    assert s.assembly_count == 1
    q = s.get(0)
    assert q["filename"] == "assembly.fasta"
    assert q["header"] == "name abc def ghi"
    assert q["fwd"] == b".....CTTATGAGCAGTCCTTAACGTAGCGGT....."


# misc.rs:589-593
def test_reverse_complement():
    assert O.reverse_complement("GGTATCACTCAGGAAGC") == "GCTTCCTGAGTGATACC"
    assert O.reverse_complement("XYZ") == "NNN"
    assert O.reverse_complement("..AC") == "GT.."


# misc.rs:780-826 (FASTA loader: multi-line records, gz, uppercase, empty file fatal)
def test_load_fasta(tmp_path):
    p = tmp_path / "t.fasta"
    p.write_text(">A info\nACGT\nacgt\n\n>B\nTT\r\nGG\n")
    assert O.load_fasta(p) == [("A", "A info", "ACGTACGT"), ("B", "B", "TTGG")]
    g = tmp_path / "t.fasta.gz"
    with gzip.open(g, "wt") as f:
        f.write(">A info\nACGT\nacgt\n")
    assert O.load_fasta(g) == [("A", "A info", "ACGTACGT")]
    e = tmp_path / "e.fasta"
    e.write_text("")
    with pytest.raises(O.OracleError, match="empty file"):
        O.load_fasta(e)


# misc.rs:65-96 incl. the operator-precedence quirk
def test_find_all_assemblies(tmp_path):
    for n in ["b.fna", "a.fasta", "c.fa", "d.fasta.gz", "e.fna.gz", "e.xyz", "weird.fna.txt", "x.txt"]:
        (tmp_path / n).write_text(">x\nA\n")
    names = [Path(p).name for p in O.find_all_assemblies(tmp_path)]
    assert names == ["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz", "weird.fna.txt"]


# unitig_graph.rs:993-1043 test_graph_stats
@pytest.mark.parametrize("n,expected", [(1, (9, 10, 92, 21, 11)), (2, (9, 3, 31, 8, 4)), (3, (9, 7, 85, 15, 8)),
                                        (4, (3, 5, 43, 10, 5)), (5, (3, 6, 60, 8, 4)), (6, (3, 2, 34, 2, 1)),
                                        (7, (3, 2, 34, 2, 1))])
def test_graph_stats(n, expected):
    assert O.gfa_stats(gfa(n)) == expected


# graph_simplification.rs:540-580
def test_common_start_end_seq():
    a, b, c = "ACGATCAGC", "ACTATCAGC", "ACTACGACT"
    assert O.common_seq([a + "+", b + "+", c + "+"], True) == "AC"
    assert O.common_seq([a + "+", b + "+", c + "-"], True) == "A"
    assert O.common_seq([a + "+", b + "-", c + "-"], True) == ""
    assert O.common_seq([a + "+", b + "+", c + "+"], False) == ""
    assert O.common_seq([a + "-", b + "-", c + "+"], False) == "T"
    assert O.common_seq([a + "-", b + "-", c + "-"], False) == "GT"


# graph_simplification.rs:582-625
def test_exclusive_inputs_outputs():
    exp = [("2+,3-", ""), ("", ""), ("", ""), ("", "7-,8+"), ("", ""), ("", ""), ("9-,9+", ""), ("", "10-"),
           ("", ""), ("", "8-")]
    assert O.gfa_exclusive(gfa(1)) == exp


# graph_simplification.rs:627-671 test_simplify_structure_1/2 — exact post-state and order
def test_simplify_structure_fixtures():
    assert O.gfa_simplify(gfa(1)) == ["GCATTCGCTGCGCTCGCTTCGCTTT", "TGCCGTCGTCGCTGT", "CTGAATCGCCTA", "GCTCGGCTCGA",
                                      "CGAACCAT", "TACTTGT", "GCCT", "TCT", "GC", "T"]
    assert O.gfa_simplify(gfa(2)) == ["CACCGCTGCGCTCGCTTCGCTCTAT", "CG", "G"]


# graph_simplification.rs:673-684
def test_check_for_duplicates():
    assert not O.check_duplicates([1, 2, 3])
    assert O.check_duplicates([1, 2, 1])


# unitig_graph.rs:1253-1270-style: path -> sequence on the fixture with paths, and save->load->save identity
def test_fixture_14_roundtrip():
    g = gfa(14)
    once = O.gfa_resave(g)
    assert O.gfa_resave(once) == once
    seqs = O.decompress(g)
    assert len(seqs) == len([l for l in g.splitlines() if l.startswith("P")])
    for line, (fn, hd, s) in zip([l for l in g.splitlines() if l.startswith("P")], seqs):
        ln = int([t for t in line.split("\t") if t.startswith("LN:i:")][0][5:])
        assert len(s) == ln


def _write_fixed(d, gz=True):
    (d / "a.fasta").write_text(f">a\n{FIXED['a']}\n")
    (d / "b.fna").write_text(f">b\n{FIXED['b']}\n")
    (d / "c.fa").write_text(f">c\n{FIXED['c']}\n")
    for name, key in (("d.fasta.gz", "d"), ("e.fna.gz", "e")):
        with gzip.open(d / name, "wt") as f:
            f.write(f">{key}\n{FIXED[key]}\n")
    (d / "e.xyz").write_text(f">a\n{FIXED['a']}\n")  # bad extension, not included


# tests.rs:131-148 test_fixed_seqs — (a) save->load->save byte identity, (b) decompress == input
@pytest.mark.parametrize("k", [1, 5, 9, 13, 51])
def test_fixed_seqs_roundtrip(tmp_path, k):
    _write_fixed(tmp_path)
    s = O.Seqs.from_dir(tmp_path, k)
    assert s.assembly_count == 5 and len(s) == 5
    g1, stats, _ = s.compress(k)
    assert O.gfa_resave(g1) == g1
    rec = O.decompress(g1)
    assert [(f, h, q) for f, h, q in rec] == [("a.fasta", "a", FIXED["a"]), ("b.fna", "b", FIXED["b"]),
                                             ("c.fa", "c", FIXED["c"]), ("d.fasta.gz", "d", FIXED["d"]),
                                             ("e.fna.gz", "e", FIXED["e"])]
    assert stats["kmers"] % 2 == 0 or k == 1


# tests.rs:151-167 test_random_seqs — same properties on random sequences (own PRNG: Rust's StdRng
# stream is not reproducible here, and the reference asserts properties, not bytes).
@pytest.mark.parametrize("length", [10, 20, 50, 100])
@pytest.mark.parametrize("seed", [0, 5, 10, 15, 20])
def test_random_seqs_roundtrip(length, seed):
    import random
    for k in (3, 5, 7, 9):
        seqs = []
        for i in range(5):
            r = random.Random(seed + i)
            seqs.append("".join(r.choice("ACGT") for _ in range(length)))
        s = O.Seqs.from_raw(k, seqs, filenames=["a.fasta", "b.fasta", "c.fasta", "d.fasta", "e.fasta"],
                            headers=list("abcde"))
        g1, _, _ = s.compress(k)
        assert O.gfa_resave(g1) == g1
        assert [q for _, _, q in O.decompress(g1)] == seqs


@pytest.mark.parametrize("name,k", [("configB_k51", 51), ("configC_k51", 51), ("configDprime_k101", 101)])
def test_full_size_goldens_are_consistent(name, k):
    # tests/golden/*.json (the oracle on whole configurations, made by tests/golden/make_configC_golden.sh): what the device test
    # compares its GFA digest with.  Internal consistency of the recorded statistics (unitig.rs:158-166: trimmed length = number
    # of k-mers; kmers.len() counts both strands).
    import json
    from pathlib import Path
    g = json.loads((Path(__file__).resolve().parent / "golden" / (name + ".json")).read_text())
    assert g["k"] == k and len(g["gfa_md5"]) == 32
    assert g["kmers"] == 2 * g["pre"]["total_length"]
    assert g["pre"]["unitigs"] == g["post"]["unitigs"] and g["pre"]["links"] == g["post"]["links"]
    assert g["post"]["total_length"] < g["pre"]["total_length"]


def test_position_reserve_is_only_a_hint():
    """ORACLE_NO_POSITION_RESERVE=1 (used to record tests/golden/configEprime_k51.json within the container's memory) skips the
    reference's Vec::with_capacity(assembly_count) per k-mer (kmer_graph.rs:40): a capacity hint — the whole compress output must be
    byte-identical with and without it (run in a fresh process each way: the switch is read once)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = ("import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import oracle_lib as O, seqgen\n"
            "from autocycler_amd import synth\n"
            "h = hashlib.md5()\n"
            "for k, seed in ((9, 3), (21, 7), (51, 13)):\n"
            "    seqs, fn, hd = seqgen.make_case(seed, k)\n"
            "    gfa, st, _ = O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd).compress(k)\n"
            "    h.update(gfa.encode()); h.update(repr(sorted(st.items())).encode())\n"
            "seqs, fn, hd = synth.flatten(synth.make_mixed_species(2, 3, genome=20000, plasmid=1500, seed=5))\n"
            "gfa, st, _ = O.Seqs.from_raw(51, [s.tobytes().decode() for s in seqs], filenames=fn, headers=hd, assembly_count=6).compress(51)\n"
            "h.update(gfa.encode()); print(h.hexdigest())\n") % (str(root), str(root / "tests"))
    import os
    outs = []
    for env in ({}, {"ORACLE_NO_POSITION_RESERVE": "1"}):
        e = {k: v for k, v in os.environ.items() if k != "ORACLE_NO_POSITION_RESERVE"}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) == 32


def test_input_assemblies_yaml_follows_the_struct_declarations(tmp_path):
    """metrics.rs:65-107: serde_yaml writes a struct's fields in declaration order — InputAssemblyMetrics { input_assemblies_count,
    input_assemblies_total_contigs, input_assemblies_total_length, compressed_unitig_count, compressed_unitig_total_length,
    input_assembly_details: Vec<InputAssemblyDetails { filename, contigs: Vec<InputContigDetails { name, description, length }> }> } —
    as block-style YAML (metrics.rs:256-260, serde_yaml 0.9: `- ` items at the parent key's indentation, nested mappings indented by
    two).  The oracle's file for the five-file fixture is that, line for line; name / description are the contig header up to / after its
    first space (sequence.rs:77-83).  Where the reference sources are at hand the field lists are re-read from them."""
    import re
    from pathlib import Path
    import boundary_cases as B
    src = tmp_path / "asm"
    B.write_five_file_fixture(src)
    (src / "f.fasta").write_text(">contig_7 length=75 circular=true\n" + FIXED["a"] + "\n>plain\n" + FIXED["b"] + "\n")
    out = tmp_path / "o"
    O.compress_dir(src, out, k=13)
    lines = (out / "input_assemblies.yaml").read_text().splitlines()
    top = ["input_assemblies_count", "input_assemblies_total_contigs", "input_assemblies_total_length", "compressed_unitig_count",
           "compressed_unitig_total_length", "input_assembly_details"]
    detail, contig = ["filename", "contigs"], ["name", "description", "length"]
    ref = Path("/root/reference/src/metrics.rs")
    if ref.exists():      # the declarations themselves (build container only)
        text = ref.read_text()
        fields = lambda name: re.findall(r"pub (\w+):", re.search(r"pub struct %s \{(.*?)\n\}" % name, text, re.S).group(1))
        assert fields("InputAssemblyMetrics") == top and fields("InputAssemblyDetails") == detail and fields("InputContigDetails") == contig
    assert [l.split(":")[0] for l in lines if not l.startswith((" ", "-"))] == top
    assert lines[0] == "input_assemblies_count: 6" and lines[1] == "input_assemblies_total_contigs: 7"
    assert lines[2] == "input_assemblies_total_length: %d" % (75 * 7) and lines[5] == "input_assembly_details:"
    body = lines[6:]
    files = [i for i, l in enumerate(body) if l.startswith("- filename: ")]
    # (the path as find_all_assemblies produced it, misc.rs:65-96 -> InputAssemblyDetails::new, metrics.rs:82-87: directory included)
    assert [body[i][len("- filename: "):] for i in files] == [str(src / f) for f in sorted(["a.fasta", "b.fna", "c.fa", "d.fasta.gz", "e.fna.gz", "f.fasta"])]
    for i in files:
        assert body[i + 1] == "  contigs:"
        assert body[i + 2].startswith("  - name: ") and body[i + 3].startswith("    description: ") and body[i + 4] == "    length: 75"
    j = files[-1]
    assert body[j + 2] == "  - name: contig_7" and body[j + 3] == "    description: length=75 circular=true"
    assert body[j + 5] == "  - name: plain" and body[j + 6] == "    description: ''"      # serde_yaml writes the empty string quoted


def test_reference_binary_hook_skips_cleanly_without_cargo():
    """tests/golden/check_against_rust.sh: with no Rust toolchain it says so and exits 0; with one it builds the
    reference and diffs GFA and YAML of oracle and product against it."""
    import shutil
    import subprocess
    from pathlib import Path
    script = Path(__file__).resolve().parent / "golden" / "check_against_rust.sh"
    out = subprocess.run(["bash", str(script)], capture_output=True, text=True, timeout=600)
    if shutil.which("cargo") is None:
        assert out.returncode == 0 and "no cargo" in out.stdout
    else:
        assert out.returncode == 0, out.stdout[-2000:]
