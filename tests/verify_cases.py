"""ac_verify_graph (the round-trip verifier behind the ABI) — test bodies shared by the CPU suite (the emulation of the
same kernels) and the device suite: it accepts every oracle-equal graph, it accepts graphs reloaded from their GFA, and it names the
damage in graphs that were corrupted after the build (the result arrays of a handle are plain host memory: the tests write into them)."""
import numpy as np

import oracle_lib as O
import parity_util
import seqgen
from autocycler_amd import _capi, compress_build, graph_from_gfa

F_UNITIG, F_ORDER, F_LINK_RANGE, F_LINK_DUP, F_LINK_MIRROR, F_PATH_RANGE, F_PATH_STEP, F_PATH_LEN, F_SPELL, F_DEPTH, F_STATS = (1 << i for i in range(11))


def build(lib_path, k, seqs, fn, hd, repair=True):
    s = O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd, repair=repair)
    loaded = s.all()
    triples = [(q["fwd"], q["length"], q["id"]) for q in loaded]
    g = compress_build(k, max(1, len({q["filename"] for q in loaded})), triples, lib_path=lib_path)
    return g, triples, loaded


def accepts_oracle_equal_graphs(lib_path, ks=(5, 11, 51), seeds=range(12)):
    done = 0
    for k in ks:
        for seed in seeds:
            seqs, fn, hd = seqgen.make_case(seed, k)
            parity_util.check_case(k, seqs, fn, hd, lib_path=lib_path).__getitem__(0).close()      # == the oracle, byte for byte
            g, triples, loaded = build(lib_path, k, seqs, fn, hd)
            rep = g.verify(triples)
            assert rep["failed"] == 0, (k, seed, rep)
            assert rep["bases_checked"] == sum(t[1] for t in triples) and rep["unitigs"] == g.unitig_count
            # ... and the writing twin of the spelling check: every sequence decompressed on the device == the host's per-sequence form == the input
            dec = g.decompress_all()
            assert dec == [g.decompress(i) for i in range(len(triples))]
            assert dec == [bytes(t[0])[k // 2:k // 2 + t[1]] for t in triples]
            assert rep["path_entries"] == sum(g.path_counts())
            # the same graph reloaded from its GFA text (ac_graph_from_gfa: what `autocycler cluster` / `decompress` start from)
            gfa = g.gfa([q["filename"] for q in loaded], [q["header"] for q in loaded])
            g2, _, _ = graph_from_gfa(gfa, lib_path=lib_path)
            rep2 = g2.verify(triples)
            assert rep2["failed"] == 0, (k, seed, rep2)
            g.close(); g2.close()
            done += 1
    return done


def names_the_damage(lib_path, k=21):
    """One flipped base, one dropped link, one swapped path entry and a few more: every corruption is reported with the
    right class, and the graph verifies again once the bytes are restored."""
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(5, genome=30_000, plasmid=2_000, sub=2e-3, indel=2e-4, seed=23)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    g, triples, _ = build(lib_path, k, seqs, fn, hd)
    b = g.bulk()
    assert g.verify(triples)["failed"] == 0
    U = g.unitig_count
    assert U > 50 and len(b["links"]) > 50 and len(b["path_entries"]) > 200
    for a in (b["seq_bytes"], b["seq_len"], b["depth"], b["links"], b["path_entries"]):
        a.setflags(write=True)

    def check(expect, must_not=0):
        rep = g.verify(triples)
        assert rep["failed"] & expect == expect, (bin(rep["failed"]), bin(expect))
        assert rep["failed"] & must_not == 0, (bin(rep["failed"]), bin(must_not))
        return rep

    # one flipped base in the longest unitig: the paths through it no longer spell their sequences
    pos = int(b["seq_begin"][0]) + int(b["seq_len"][0]) // 2
    old = int(b["seq_bytes"][pos])
    b["seq_bytes"][pos] = ord("A") if old != ord("A") else ord("C")
    rep = check(F_SPELL, must_not=F_LINK_MIRROR | F_PATH_STEP | F_DEPTH | F_PATH_LEN)
    assert rep["first_bad_base"] < sum(t[1] for t in triples)
    b["seq_bytes"][pos] = old
    assert g.verify(triples)["failed"] == 0
    # one dropped link (overwritten by its neighbour): a duplicate, a mirror without partner, and path steps that are no links any more
    used = {(int(x), int(y)) for p0, p1 in zip(b["path_off"][:-1], b["path_off"][1:]) for x, y in zip(b["path_entries"][int(p0):int(p1) - 1], b["path_entries"][int(p0) + 1:int(p1)])}
    sgn = lambda n, f: int(n) if f else -int(n)
    li = next(i for i in range(len(b["links"]) - 1) if (sgn(b["links"][i]["a"], b["links"][i]["a_fwd"]), sgn(b["links"][i]["b"], b["links"][i]["b_fwd"])) in used)
    keep = b["links"][li].copy()
    b["links"][li] = b["links"][li + 1]
    rep = check(F_LINK_DUP | F_LINK_MIRROR | F_PATH_STEP, must_not=F_SPELL | F_DEPTH)
    assert rep["first_bad_link"] <= li + 1
    b["links"][li] = keep
    # a link that points beyond the graph
    keep_b = int(b["links"][3]["b"])
    b["links"][3]["b"] = U + 7
    check(F_LINK_RANGE)
    b["links"][3]["b"] = keep_b
    assert g.verify(triples)["failed"] == 0
    # two neighbouring path entries swapped (different unitigs): steps leave the links, the path spells something else
    po = b["path_off"]
    j = next(j for j in range(int(po[0]), int(po[1]) - 1) if abs(int(b["path_entries"][j])) != abs(int(b["path_entries"][j + 1])))
    x, y = int(b["path_entries"][j]), int(b["path_entries"][j + 1])
    b["path_entries"][j], b["path_entries"][j + 1] = y, x
    rep = check(F_PATH_STEP)
    assert rep["failed"] & (F_SPELL | F_PATH_LEN) and rep["first_bad_path_entry"] <= j + 1
    b["path_entries"][j], b["path_entries"][j + 1] = x, y
    # a path entry that names no unitig; a strand flipped
    b["path_entries"][j] = U + 1
    check(F_PATH_RANGE)
    b["path_entries"][j] = -x
    check(F_PATH_STEP | F_SPELL) if int(b["seq_len"][abs(x) - 1]) > 1 else check(F_PATH_STEP)
    b["path_entries"][j] = x
    # depth off by one; a unitig shortened (path lengths and the statistics no longer add up)
    b["depth"][5] += 1.0
    check(F_DEPTH, must_not=F_SPELL | F_PATH_STEP)
    b["depth"][5] -= 1.0
    b["seq_len"][0] -= 1
    check(F_PATH_LEN | F_STATS)
    b["seq_len"][0] += 1
    # renumber order: two unitigs of different length exchanged in the length table only
    if int(b["seq_len"][0]) != int(b["seq_len"][U - 1]):
        l0, l1 = int(b["seq_len"][0]), int(b["seq_len"][U - 1])
        b["seq_len"][0], b["seq_len"][U - 1] = l1, l0
        check(F_ORDER)
        b["seq_len"][0], b["seq_len"][U - 1] = l0, l1
    rep = g.verify(triples)
    assert rep["failed"] == 0, rep
    # the wrong sequences: same count and lengths, one base changed in the INPUT
    fwd = bytearray(triples[0][0]); mid = len(fwd) // 2
    fwd[mid] = ord("A") if fwd[mid] != ord("A") else ord("C")
    rep = g.verify([(bytes(fwd),) + triples[0][1:]] + triples[1:])
    assert rep["failed"] == F_SPELL and rep["first_bad_base"] == mid - k // 2
    g.close()
