"""Deterministic synthetic assembly sets — numpy, all seeds fixed.

One species: a root chromosome of iid uniform bases with planted repeats (7 x 5 kb, 10 x 1.3 kb, scaled down
for small genomes) and one plasmid; per assembly: substitutions, short indels, random rotation of each
circular replicon, strand flip with p = 0.5, plasmid kept with p = 0.75."""
import numpy as np

_ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    _COMP[a] = b


def _root(rng, genome, plasmid):
    chrom = _ALPHA[rng.integers(0, 4, size=genome, dtype=np.uint8)]
    scale = min(1.0, genome / 5_000_000)
    for count, length in ((7, 5000), (10, 1300)):
        length = max(60, int(length * max(scale, 0.02)))
        if length * 2 >= genome:
            continue
        rep = _ALPHA[rng.integers(0, 4, size=length, dtype=np.uint8)]
        for _ in range(count):
            p = int(rng.integers(0, genome - length))
            chrom[p:p + length] = rep
    plas = _ALPHA[rng.integers(0, 4, size=plasmid, dtype=np.uint8)] if plasmid else None
    return chrom, plas


def _mutate(rng, seq, sub, indel):
    seq = seq.copy()
    n = len(seq)
    n_sub = rng.binomial(n, sub)
    if n_sub:
        pos = rng.integers(0, n, size=n_sub)
        seq[pos] = _ALPHA[(np.searchsorted(_ALPHA, seq[pos]) + rng.integers(1, 4, size=n_sub)) % 4]
    n_indel = rng.binomial(n, indel)
    if n_indel:
        pieces, last = [], 0
        for p in np.sort(rng.integers(1, n - 1, size=n_indel)):
            if p <= last:
                continue
            ln = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                pieces.append(seq[last:p]); pieces.append(_ALPHA[rng.integers(0, 4, size=ln, dtype=np.uint8)]); last = p
            else:
                pieces.append(seq[last:p]); last = min(n, p + ln)
        pieces.append(seq[last:])
        seq = np.concatenate(pieces)
    return seq


def _place(rng, seq):
    r = int(rng.integers(0, len(seq)))
    seq = np.concatenate([seq[r:], seq[:r]])
    if rng.random() < 0.5:
        seq = _COMP[seq][::-1]
    return np.ascontiguousarray(seq)


def make_assemblies(n_assemblies, genome=5_000_000, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000, first=0):
    """Returns a list of assemblies; each is a list of (header, uint8 array of ACGT bytes).  `first`: index of the
    first assembly to make (assemblies first .. first + n_assemblies - 1 of the same species: one job's slice)."""
    chrom, plas = _root(np.random.default_rng(seed - 1), genome, plasmid)
    out = []
    for i in range(first, first + n_assemblies):
        rng = np.random.default_rng(seed + i)
        contigs = []
        c = _place(rng, _mutate(rng, chrom, sub, indel))
        contigs.append((f"contig_1 length={len(c)} circular=true", c))
        if plas is not None and rng.random() < 0.75:
            p = _place(rng, _mutate(rng, plas, sub, indel))
            contigs.append((f"contig_2 length={len(p)} circular=true", p))
        out.append(contigs)
    return out


def write_fasta_dir(assemblies, out_dir):
    """Single-line records (so decompress reproduces the files byte-for-byte, tests.rs:122-127)."""
    from pathlib import Path
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    for i, contigs in enumerate(assemblies):
        with open(out_dir / f"assembly_{i:04d}.fasta", "wb") as f:
            for header, seq in contigs:
                f.write(b">" + header.encode() + b"\n" + seq.tobytes() + b"\n")
