"""Seeded generators of small, structurally nasty assembly sets for parity tests (CPU and GPU)."""
import random

COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


def rand_seq(r, n, alphabet="ACGT"):
    return "".join(r.choice(alphabet) for _ in range(n))


def mutate(r, s, sub=0.01, indel=0.002):
    out = []
    i = 0
    while i < len(s):
        x = r.random()
        if x < sub:
            out.append(r.choice([c for c in "ACGT" if c != s[i]]))
        elif x < sub + indel:
            if r.random() < 0.5:
                out.append(rand_seq(r, r.randint(1, 3)))
                out.append(s[i])
            # else: deletion
        else:
            out.append(s[i])
        i += 1
    return "".join(out)


def rotate(r, s):
    i = r.randrange(len(s))
    return s[i:] + s[:i]


def make_case(seed, k):
    """Returns (seqs, filenames, headers): a list of contigs spread over a few 'assemblies'."""
    r = random.Random(seed)
    kind = seed % 12
    L = r.choice([k + 1, k + 7, 2 * k + 3, 5 * k, 12 * k, 300, 800])
    L = max(L, k + 1)
    seqs = []
    if kind == 0:      # independent random contigs (tests.rs:151-167 style)
        seqs = [rand_seq(r, max(k, r.randint(k, L))) for _ in range(r.randint(1, 6))]
    elif kind == 1:    # mutated, rotated, strand-flipped copies of one circular genome
        g = rand_seq(r, L)
        for _ in range(r.randint(2, 7)):
            s = mutate(r, rotate(r, g), sub=r.choice([0, 0.005, 0.02]))
            seqs.append(rc(s) if r.random() < 0.5 else s)
    elif kind == 2:    # genome with planted direct + inverted repeats
        rep = rand_seq(r, r.randint(k - 1, 3 * k))
        g = rand_seq(r, L) + rep + rand_seq(r, L // 2 + 1) + rc(rep) + rand_seq(r, L // 3 + 1) + rep + rand_seq(r, k)
        for _ in range(r.randint(1, 5)):
            s = mutate(r, g, sub=r.choice([0, 0.01]), indel=0)
            seqs.append(rc(s) if r.random() < 0.3 else s)
    elif kind == 3:    # tandem repeats and homopolymers (low complexity)
        unit = rand_seq(r, r.randint(1, 7))
        g = rand_seq(r, k) + unit * (r.randint(2, 3 * k) // len(unit) + 2) + rand_seq(r, k) + "A" * r.randint(k - 2, 2 * k) + rand_seq(r, k)
        seqs = [g, mutate(r, g, 0.01, 0), rc(g)]
    elif kind == 4:    # identical contigs, and a contig that is the exact RC of another
        g = rand_seq(r, L)
        seqs = [g, g, rc(g), rand_seq(r, L)]
    elif kind == 5:    # palindromic (self-RC) sequences: unitig midpoint Y == rc(X)
        half = rand_seq(r, r.randint(k, L))
        seqs = [half + rc(half), rand_seq(r, k + 2) + half[-k:] + rc(half[-k:]) + rand_seq(r, k + 2)]
    elif kind == 6:    # linear fragments: ends NOT covered by anything else -> dots survive end repair
        g = rand_seq(r, 3 * L)
        seqs = [g[: 2 * L], g[L // 2:], rc(g[L: L + max(k, L)]), rand_seq(r, k)]
    elif kind == 7:    # shared prefixes/suffixes between contigs (dot k-mers shared or branching)
        p = rand_seq(r, k + 3)
        q = rand_seq(r, k + 3)
        seqs = [p + rand_seq(r, L) + q, p + rand_seq(r, L) + q, p + rand_seq(r, L), rc(q) + rand_seq(r, L)]
    elif kind == 8:    # two-letter alphabet: dense collisions, many branches
        seqs = [rand_seq(r, max(k, L // 2), "AC") for _ in range(r.randint(2, 5))]
    elif kind == 9:    # exactly-k contigs and k+1 contigs
        seqs = [rand_seq(r, k), rand_seq(r, k + 1), rand_seq(r, k)]
        seqs.append(seqs[0])
    elif kind == 10:   # plasmid present in some assemblies only + chromosome variants (bubbles)
        c = rand_seq(r, L)
        pl = rand_seq(r, max(k, L // 4))
        for _ in range(r.randint(2, 6)):
            seqs.append(mutate(r, rotate(r, c), 0.01, 0.002))
            if r.random() < 0.6:
                seqs.append(rotate(r, pl))
    else:              # overlapping windows of one genome, both strands
        g = rand_seq(r, 2 * L + k)
        for _ in range(r.randint(2, 6)):
            a = r.randrange(0, len(g) - k)
            b = min(len(g), a + r.randint(k, L + k))
            s = g[a:b]
            seqs.append(rc(s) if r.random() < 0.5 else s)
    seqs = [s for s in seqs if len(s) >= k]
    if not seqs:
        seqs = [rand_seq(r, k + 5)]
    n_asm = r.randint(1, min(4, len(seqs)))
    filenames = [f"asm_{i % n_asm}.fasta" for i in range(len(seqs))]
    order = sorted(range(len(seqs)), key=lambda i: filenames[i])
    seqs = [seqs[i] for i in order]
    filenames = [filenames[i] for i in order]
    headers = [f"contig_{i + 1} length={len(s)}" for i, s in enumerate(seqs)]
    return seqs, filenames, headers
