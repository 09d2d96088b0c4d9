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


def make_mixed_species(n_species, n_strains, genome=5_000_000, plasmid=100_000, strain_div=1e-2, sub=1e-4, indel=1e-5, seed=77_000):
    """BASELINE.json configs[4] model ("E"): n_species unrelated roots; every assembly is one strain of its
    species = the root with substitutions at rate strain_div, then the per-assembly noise (sub, indel), rotation, strand flip and
    plasmid loss of make_assemblies.  Assembly i = species i // n_strains, strain i % n_strains (file names sort in that order)."""
    out = []
    for sp in range(n_species):
        chrom, plas = _root(np.random.default_rng(seed + 1_000_003 * (sp + 1)), genome, plasmid)
        for st in range(n_strains):
            rng = np.random.default_rng(seed + sp * n_strains + st)
            contigs = []
            c = _place(rng, _mutate(rng, _mutate(rng, chrom, strain_div, 0.0), sub, indel))
            contigs.append((f"contig_1 length={len(c)} circular=true", c))
            if plas is not None and rng.random() < 0.75:
                p = _place(rng, _mutate(rng, _mutate(rng, plas, strain_div, 0.0), sub, indel))
                contigs.append((f"contig_2 length={len(p)} circular=true", p))
            out.append(contigs)
    return out


# Named workloads: the BASELINE.json configurations (or their scaled replicas) that tests/golden/*.json hold oracle results for.
# name -> (k, number of assemblies, generator)
WORKLOADS = {
    "configB_k51": (51, 12, lambda: make_assemblies(12, genome=5_000_000, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000)),
    "configC_k51": (51, 96, lambda: make_assemblies(96, genome=5_000_000, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000)),
    "configDprime_k101": (101, 24, lambda: make_assemblies(24, genome=10_000_000, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000)),
    "configD_k101": (101, 24, lambda: make_assemblies(24, genome=100_000_000, plasmid=0, sub=1e-4, indel=1e-5, seed=101_000)),
    # E' = scaled replica of configs[4] (1000 x 5 Mbp, 25 species x 40 strains): 5 species x 20 strains = 100 assemblies of ~1 Mbp,
    # strains 1 % apart.  The genome is 1 Mbp instead of 5 because the oracle keeps the reference's per-k-mer heap records
    # (kmer_graph.rs:36-41) and nearly every k-mer of such an input is distinct: 100 x 5 Mbp would need several hundred GB.
    "configEprime_k51": (51, 100, lambda: make_mixed_species(5, 20, genome=1_000_000, plasmid=20_000, strain_div=1e-2, sub=1e-4, indel=1e-5, seed=77_000)),
    # E2: the same model with ~2 Mbp genomes (about the largest mixed-species input whose oracle run fits the 62 GB build container)
    "configE2_k51": (51, 100, lambda: make_mixed_species(5, 20, genome=2_000_000, plasmid=40_000, strain_div=1e-2, sub=1e-4, indel=1e-5, seed=78_000)),
    # bench.py's one-job workload at N GPUs (species r = 96 assemblies of the config C model on rank r) as ONE single-device job: what
    # the replicated stages of the sharded build cost a rank at that N
    "benchjob2_k51": (51, 192, lambda: [a for sp in range(2) for a in make_assemblies(96, seed=51_000 + 1000 * sp)]),
    "benchjob4_k51": (51, 384, lambda: [a for sp in range(4) for a in make_assemblies(96, seed=51_000 + 1000 * sp)]),
    "benchjob8_k51": (51, 768, lambda: [a for sp in range(8) for a in make_assemblies(96, seed=51_000 + 1000 * sp)]),
    # mini-E: 2 species x 40 strains x 5 Mbp — configs[4]'s per-GPU shape at N = 8 is 125 assemblies; no oracle golden (memory)
    "configEmini_k51": (51, 80, lambda: make_mixed_species(2, 40, genome=5_000_000, plasmid=100_000, strain_div=1e-2, sub=1e-4, indel=1e-5, seed=77_000)),
}


def flatten(assemblies):
    """-> (sequences as uint8 arrays, file names, headers) in the order `compress` numbers them."""
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(assemblies):
        for header, s in contigs:
            seqs.append(np.ascontiguousarray(s)); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    return seqs, fn, hd


def write_fasta_dir(assemblies, out_dir):
    """Single-line records (so decompress reproduces the files byte-for-byte, tests.rs:122-127)."""
    from pathlib import Path
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    for i, contigs in enumerate(assemblies):
        with open(out_dir / f"assembly_{i:04d}.fasta", "wb") as f:
            for header, seq in contigs:
                f.write(b">" + header.encode() + b"\n" + seq.tobytes() + b"\n")
