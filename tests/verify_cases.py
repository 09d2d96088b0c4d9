"""ac_verify_graph (the round-trip verifier behind the ABI) — test bodies shared by the CPU suite (the emulation of the
same kernels) and the device suite: it accepts every oracle-equal graph, it accepts graphs reloaded from their GFA, and it names the
damage in graphs that were corrupted after the build (the result arrays of a handle are plain host memory: the tests write into them)."""
import numpy as np

import oracle_lib as O
import parity_util
import seqgen
from autocycler_amd import _capi, compress_build, graph_from_gfa

F_UNITIG, F_ORDER, F_LINK_RANGE, F_LINK_DUP, F_LINK_MIRROR, F_PATH_RANGE, F_PATH_STEP, F_PATH_LEN, F_SPELL, F_DEPTH, F_STATS = (1 << i for i in range(11))
F_LINK_ORDER, F_MAXIMAL, F_EXPAND = 2048, 4096, 8192                    # round 6: the order-sensitive guarantees
C_LINK_ORDER, C_LINK_ORDER_SEEDS, C_MAXIMAL, C_EXPAND = 1, 2, 4, 8      # report["checks"]: which of them ran


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
            assert rep["checks"] == C_LINK_ORDER | C_LINK_ORDER_SEEDS | C_MAXIMAL | C_EXPAND, rep      # a built graph: all of them, with seed numbers
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
            assert rep2["checks"] == C_LINK_ORDER | C_MAXIMAL | C_EXPAND, rep2      # a GFA holds no seed numbers
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
    li = next(i for i in range(len(b["links"]) - 1) if (int(b["links"][i]["a"]), int(b["links"][i]["b"])) in used)
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


# ---- round 6: a unitig cut in two, a shift expand_repeats did not apply, L lines out of order -------------------------------------------
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


class _Gfa:
    """A compress-written GFA as lists the tests can edit (S: [sequence, tags], L: (a, a_strand, b, b_strand) signed pairs, P: [id, entries, tags])."""

    def __init__(self, text):
        self.header, self.segs, self.links, self.paths = "", [], [], []
        for line in text.split("\n"):
            f = line.split("\t")
            if f[0] == "H": self.header = line
            elif f[0] == "S": assert int(f[1]) == len(self.segs) + 1; self.segs.append([f[2], f[3:]])
            elif f[0] == "L": self.links.append(((int(f[1]) if f[2] == "+" else -int(f[1])), (int(f[3]) if f[4] == "+" else -int(f[3]))))
            elif f[0] == "P": self.paths.append([f[1], [(int(e[:-1]) if e[-1] == "+" else -int(e[:-1])) for e in f[2].split(",")], f[3:]])

    def text(self):
        sg = lambda v: f"{abs(v)}\t{'+' if v > 0 else '-'}"
        out = [self.header]
        out += [f"S\t{i + 1}\t{q}\t" + "\t".join(t) for i, (q, t) in enumerate(self.segs)]
        out += [f"L\t{sg(a)}\t{sg(b)}\t0M" for a, b in self.links]
        out += ["P\t" + pid + "\t" + ",".join(f"{abs(e)}{'+' if e > 0 else '-'}" for e in ents) + "\t" + "\t".join(t) for pid, ents, t in self.paths]
        return "\n".join(out) + "\n"

    def succ(self):
        nx = {}
        for a, b in self.links: nx.setdefault(a, []).append(b)
        return nx

    def strand_seq(self, v):
        q = self.segs[abs(v) - 1][0]
        return q if v > 0 else "".join(_COMP[c] for c in reversed(q))

    def candidates(self):
        """(unitig number, side) that pass expand_repeats' static test (graph_simplification.rs:43-86, 190-280), restated on the GFA."""
        nx = self.succ()
        fixed_start, fixed_end = set(), set()
        for _, ents, _ in self.paths:
            (fixed_start if ents[0] > 0 else fixed_end).add(abs(ents[0]))
            (fixed_end if ents[-1] > 0 else fixed_start).add(abs(ents[-1]))
        fs0, fe0 = set(fixed_start), set(fixed_end)
        for u in fs0:
            for t in nx.get(-u, []): (fixed_end if -t > 0 else fixed_start).add(abs(t))
        for u in fe0:
            for t in nx.get(u, []): (fixed_start if t > 0 else fixed_end).add(abs(t))
        out = []
        for u in range(1, len(self.segs) + 1):
            ins = [-t for t in nx.get(-u, [])]
            if len(ins) >= 2 and u not in fixed_start and all(nx.get(p, []) == [u] and abs(p) != u and not ((p > 0 and abs(p) in fixed_end) or (p < 0 and abs(p) in fixed_start)) for p in ins):
                out.append((u, 0, ins))
            outs = list(nx.get(u, []))
            if len(outs) >= 2 and u not in fixed_end and all(nx.get(-q, []) == [-u] and abs(q) != u and not ((q > 0 and abs(q) in fixed_start) or (q < 0 and abs(q) in fixed_end)) for q in outs):
                out.append((u, 1, outs))
        return out


def names_order_sensitive_damage(lib_path, k=21):
    """one split unitig, one un-applied shift and one swapped L line are each reported with their own class."""
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(5, genome=30_000, plasmid=2_000, sub=2e-3, indel=2e-4, seed=23)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    g, triples, loaded = build(lib_path, k, seqs, fn, hd)
    fns, hds = [q["filename"] for q in loaded], [q["header"] for q in loaded]
    ALL = C_LINK_ORDER | C_LINK_ORDER_SEEDS | C_MAXIMAL | C_EXPAND
    rep = g.verify(triples)
    assert rep["failed"] == 0 and rep["checks"] == ALL, rep
    text = g.gfa(fns, hds)
    U = g.unitig_count

    def verify_text(t):
        g2, _, _ = graph_from_gfa(t, lib_path=lib_path)
        r = g2.verify(triples)
        g2.close()
        return r

    base = _Gfa(text)
    assert base.text() == text                                     # the editor round-trips the writer's bytes
    assert verify_text(text)["failed"] == 0
    # --- (iii) L lines: two neighbours of different groups, then two of one group, swapped in the handle's own array (seed numbers at hand)
    b = g.bulk()
    L = b["links"]; L.setflags(write=True)
    grp = lambda l: (abs(int(l["a"])), 0 if int(l["a"]) > 0 else 1)
    i_diff = next(i for i in range(len(L) - 1) if grp(L[i]) != grp(L[i + 1]))
    i_same = next(i for i in range(len(L) - 1) if grp(L[i]) == grp(L[i + 1]))
    for i in (i_diff, i_same):
        x, y = L[i].copy(), L[i + 1].copy()
        L[i], L[i + 1] = y, x
        rep = g.verify(triples)
        assert rep["failed"] == F_LINK_ORDER and rep["first_bad_link"] in (i, i + 1), (i, rep)      # the link SET is untouched: nothing else fires
        L[i], L[i + 1] = x, y
    assert g.verify(triples)["failed"] == 0
    # ... and in a GFA (no seed numbers): a forward_next list whose b- comes before its b+ (unitig_graph.rs:248-263)
    e = _Gfa(text)
    j = next(j for j in range(len(e.links) - 1) if e.links[j][0] == e.links[j + 1][0] and e.links[j][0] > 0 and e.links[j][1] > 0 > e.links[j + 1][1])
    e.links[j], e.links[j + 1] = e.links[j + 1], e.links[j]
    rep = verify_text(e.text())
    assert rep["failed"] == F_LINK_ORDER and not rep["checks"] & C_LINK_ORDER_SEEDS, rep
    # --- (i) one unitig cut in two: u keeps its first half, a new unitig U + 1 takes the rest, the links out of u's end and the paths follow
    e = _Gfa(text)
    depth_of = lambda tags: next(t for t in tags if t.startswith("DP:f:"))
    u = next(n for n in range(1, U + 1) if len(e.segs[n - 1][0]) >= 4 and all(abs(a) != n or abs(bb) != n for a, bb in e.links))
    q = e.segs[u - 1][0]; h = len(q) // 2; V = U + 1
    e.segs[u - 1][0] = q[:h]
    e.segs.append([q[h:], [depth_of(e.segs[u - 1][1])]])
    e.links = [((V if a == u else a), (-V if bb == -u else bb)) for a, bb in e.links] + [(u, V), (-V, -u)]
    for pth in e.paths:
        pth[1] = [x for v in pth[1] for x in ((u, V) if v == u else (-V, -u) if v == -u else (v,))]
    rep = verify_text(e.text())
    assert rep["failed"] & F_MAXIMAL and rep["checks"] & C_MAXIMAL, rep
    assert rep["failed"] & ~(F_MAXIMAL | F_ORDER | F_STATS) == 0, bin(rep["failed"])      # lossless and consistent all the same: what round 5 could not tell apart
    assert rep["failed"] & (F_SPELL | F_PATH_STEP | F_LINK_MIRROR | F_DEPTH | F_EXPAND) == 0
    # --- (ii) one shift un-applied: the first base of a junction unitig goes back to the end of each of its exclusive inputs (or the last one to the outputs)
    done = 0
    for want_side in (0, 1):
        e = _Gfa(text)
        cands = [(n, side, src) for n, side, src in e.candidates() if side == want_side and len(e.segs[n - 1][0]) >= 3 and len({abs(v) for v in src}) == len(src)]
        if not cands: continue
        n, side, src = cands[0]
        q = e.segs[n - 1][0]
        if side == 0:
            c, e.segs[n - 1][0] = q[0], q[1:]
            for p in src:
                if p > 0: e.segs[p - 1][0] += c
                else: e.segs[-p - 1][0] = _COMP[c] + e.segs[-p - 1][0]
        else:
            c, e.segs[n - 1][0] = q[-1], q[:-1]
            for p in src:
                if p > 0: e.segs[p - 1][0] = c + e.segs[p - 1][0]
                else: e.segs[-p - 1][0] += _COMP[c]
        rep = verify_text(e.text())
        assert rep["failed"] & F_EXPAND and rep["first_bad_junction"] == 2 * (n - 1) + side, (n, side, rep)
        assert rep["failed"] & ~(F_EXPAND | F_ORDER) == 0, bin(rep["failed"])      # every path still spells its sequence
        done += 1
    assert done == 2, "the test graph has no junction of one of the two kinds"
    g.close()
