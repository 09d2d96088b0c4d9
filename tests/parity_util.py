"""Shared parity check: product library (or the CPU emulation of its kernels) vs the oracle."""
import oracle_lib as O
from autocycler_amd import compress_build


def first_diff(a, b):
    la, lb = a.splitlines(), b.splitlines()
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return f"line {i}: oracle={x[:160]!r} got={y[:160]!r}"
    return f"line counts differ: oracle={len(la)} got={len(lb)}"


def check_case(k, seqs, filenames, headers, lib_path=None, repair=True, device=0, distances=False):
    s = O.Seqs.from_raw(k, seqs, filenames=filenames, headers=headers, repair=repair)
    gfa_o, st, _ = s.compress(k)
    loaded = s.all()
    g = compress_build(k, s.assembly_count, [(q["fwd"], q["length"], q["id"]) for q in loaded], device=device,
                       lib_path=lib_path)
    gfa_g = g.gfa([q["filename"] for q in loaded], [q["header"] for q in loaded])
    assert g.kmer_count == st["kmers"]
    assert g.stats_pre == dict(unitigs=st["unitigs_pre"], links=st["links_pre"], total_length=st["length_pre"])
    assert g.stats_post == dict(unitigs=st["unitigs_post"], links=st["links_post"], total_length=st["length_post"])
    assert gfa_o == gfa_g, first_diff(gfa_o, gfa_g)
    if distances:     # cluster.rs:132-157 on the same graph: exact f64 equality (integer sums, one division)
        assert g.pairwise_distances(device) == O.pairwise_distances(gfa_o)
    return g, gfa_g, loaded
