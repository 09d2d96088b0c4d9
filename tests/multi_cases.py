"""ac_compress_build_multi — one job over several devices from ONE process, the exchanges inside the library — as test bodies shared by the
CPU suite (the serial emulation: ranks are threads, exchanges staged through host memory) and the device suite (RCCL in a world of one
rank; several ranks sharing the one MI355X over the host-staged transport)."""
import oracle_lib as O
import parity_util
import seqgen
from autocycler_amd import _capi


def run_case(lib_path, k, seqs, filenames, headers, devices, repair=True):
    """The job through ac_compress_build_multi on `devices` (one rank per entry) == the oracle, byte for byte; returns (gfa, multi info)."""
    s = O.Seqs.from_raw(k, seqs, filenames=filenames, headers=headers, repair=repair)
    loaded = s.all()
    n_asm = max(1, len({q["filename"] for q in loaded}))
    g, info = _capi.compress_build_multi(k, n_asm, [(q["fwd"], q["length"], q["id"]) for q in loaded], devices, lib_path=lib_path)
    gfa_g = g.gfa([q["filename"] for q in loaded], [q["header"] for q in loaded])
    gfa_o, st, _ = s.compress(k)
    assert g.kmer_count == st["kmers"]
    assert g.stats_pre == dict(unitigs=st["unitigs_pre"], links=st["links_pre"], total_length=st["length_pre"])
    assert g.stats_post == dict(unitigs=st["unitigs_post"], links=st["links_post"], total_length=st["length_post"])
    assert gfa_o == gfa_g, parity_util.first_diff(gfa_o, gfa_g)
    assert info["n_ranks"] == min(len(devices), len(loaded))
    g.close()
    return gfa_g, info


def synth_case(n_assemblies, genome, plasmid, sub, indel, seed):
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(n_assemblies, genome=genome, plasmid=plasmid, sub=sub, indel=indel, seed=seed)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    return seqs, fn, hd


def mixed_case(n_species, n_strains, genome, seed=5):
    from autocycler_amd import synth
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_mixed_species(n_species, n_strains, genome=genome, plasmid=genome // 20, seed=seed)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    return seqs, fn, hd


def adversarial(lib_path, devices, ks=(5, 11, 31, 51), seeds=range(24)):
    done = 0
    for k in ks:
        for seed in seeds:
            seqs, fn, hd = seqgen.make_case(seed, k)
            for repair in (True, False):
                run_case(lib_path, k, seqs, fn, hd, devices, repair=repair)
                done += 1
    return done
