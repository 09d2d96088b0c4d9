"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (never imported by autocycler_amd)."""
import ctypes as C
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(ROOT / "oracle" / "liboracle.so"))
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_free.argtypes = [C.c_void_p]
    return _lib


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError(lib().orc_last_error().decode())


def _str_call(fn, *args):
    out = C.c_void_p()
    _check(fn(*args, C.byref(out)))
    s = C.string_at(out.value).decode()
    lib().orc_free(out)
    return s


def _b(s):
    return s.encode() if isinstance(s, str) else s


def kat_kmers(seq, k):
    return _str_call(lib().orc_kat_kmers, _b(seq), C.c_uint32(k)).splitlines()


def kat_neighbours(seq, k, kmer, nxt):
    s = _str_call(lib().orc_kat_neighbours, _b(seq), C.c_uint32(k), _b(kmer), C.c_int(1 if nxt else 0))
    return s.split(",") if s else []


def kat_position(seq_id, strand, pos):
    return _str_call(lib().orc_kat_position, C.c_uint16(seq_id), C.c_int(1 if strand else 0), C.c_uint64(pos))


def kat_kmer_display():
    return _str_call(lib().orc_kat_kmer_display)


def kat_unitig_from_kmers():
    return _str_call(lib().orc_kat_unitig_from_kmers).split(",")


def kat_shift(op, arg):
    return _str_call(lib().orc_kat_shift, _b(op), _b(str(arg))).split(",")


def find_best_match(matches):
    return _str_call(lib().orc_find_best_match, _b("\n".join(matches)))


def reverse_complement(seq):
    return _str_call(lib().orc_reverse_complement, _b(seq))


def load_fasta(path):
    return [tuple(l.split("\t")) for l in _str_call(lib().orc_load_fasta, _b(str(path))).splitlines()]


def find_all_assemblies(d):
    return _str_call(lib().orc_find_all_assemblies, _b(str(d))).splitlines()


def gfa_stats(gfa):
    return tuple(int(x) for x in _str_call(lib().orc_gfa_stats, _b(gfa)).split())


def gfa_resave(gfa):
    return _str_call(lib().orc_gfa_resave, _b(gfa))


def gfa_simplify(gfa):
    return _str_call(lib().orc_gfa_simplify, _b(gfa)).splitlines()


def gfa_exclusive(gfa):
    return [tuple(l.split("|")) for l in _str_call(lib().orc_gfa_exclusive, _b(gfa)).split("\n")[:-1]]


def common_seq(segs, start):
    return _str_call(lib().orc_common_seq, _b("\n".join(segs)), C.c_int(1 if start else 0))


def check_duplicates(numbers):
    r = C.c_int()
    _check(lib().orc_check_duplicates(_b(",".join(str(n) for n in numbers)), C.byref(r)))
    return bool(r.value)


def decompress(gfa):
    return [tuple(l.split("\t")) for l in _str_call(lib().orc_decompress, _b(gfa)).splitlines()]


def gfa_positions(gfa):
    """unitig_graph.rs:151-174: {number: (forward positions, reverse positions)}, each a list of (seq_id, strand, pos) in vector order."""
    out = {}
    for line in _str_call(lib().orc_gfa_positions, _b(gfa)).splitlines():
        num, f, r = line.split("\t")
        parse = lambda x: [tuple(int(v) for v in p.split(",")) for p in x[2:].split(";")] if len(x) > 2 else []
        out[int(num)] = ([(a, bool(b), c) for a, b, c in parse(f)], [(a, bool(b), c) for a, b, c in parse(r)])
    return out


def pairwise_distances(gfa):
    """cluster.rs:132-157 on a GFA text -> S x S list of lists (row a, column b)."""
    n = C.c_uint32()
    _check(lib().orc_pairwise_distances(_b(gfa), None, C.byref(n)))
    S = n.value
    out = (C.c_double * (S * S))()
    _check(lib().orc_pairwise_distances(_b(gfa), out, C.byref(n)))
    return [[out[a * S + b] for b in range(S)] for a in range(S)]


_COMPRESS_MEMO = {}      # digest of a loaded set + k -> (gfa, stats, times) of Seqs.compress


class Seqs:
    """Loaded (padded, end-repaired) sequences as the reference has them at compress.rs:41."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_dir(cls, d, k, max_contigs=25, threads=8):
        h = C.c_void_p()
        _check(lib().orc_load_sequences(_b(str(d)), C.c_uint32(k), C.c_uint32(max_contigs), C.c_int(threads), C.byref(h)))
        return cls(h)

    @classmethod
    def from_raw(cls, k, seqs, filenames=None, headers=None, repair=True, threads=8, assembly_count=None):
        n = len(seqs)
        filenames = filenames or [f"a{i}.fasta" for i in range(n)]
        headers = headers or [f"c{i}" for i in range(n)]
        arr = lambda xs: (C.c_char_p * n)(*[_b(x) for x in xs])
        h = C.c_void_p()
        _check(lib().orc_seqs_from_raw(C.c_uint32(k), C.c_uint32(n), arr(seqs), arr(filenames), arr(headers),
                                       C.c_int(1 if repair else 0), C.c_int(threads),
                                       C.c_uint32(assembly_count or len(set(filenames))), C.byref(h)))
        return cls(h)

    def __len__(self):
        lib().orc_seqs_count.restype = C.c_uint32
        return lib().orc_seqs_count(self.h)

    @property
    def assembly_count(self):
        lib().orc_seqs_assembly_count.restype = C.c_uint32
        return lib().orc_seqs_assembly_count(self.h)

    def get(self, i):
        fwd, rev, fn, hd = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p()
        length, sid = C.c_uint32(), C.c_uint16()
        _check(lib().orc_seq_get(self.h, C.c_uint32(i), C.byref(fwd), C.byref(rev), C.byref(length), C.byref(sid),
                                 C.byref(fn), C.byref(hd)))
        return dict(fwd=fwd.value, rev=rev.value, length=length.value, id=sid.value,
                    filename=fn.value.decode(), header=hd.value.decode())

    def all(self):
        return [self.get(i) for i in range(len(self))]

    def compress(self, k):
        """compress.rs:42-47 -> (gfa_text, stats dict, times dict).  The oracle's answer for one loaded set is computed once per process: the
        multi-rank and knob tests ask for the same set under many settings (a digest of everything the oracle reads is the key)."""
        import hashlib
        hsh = hashlib.blake2b(digest_size=16)
        hsh.update(b"%d|%d|" % (k, self.assembly_count))
        for q in self.all():
            hsh.update(b"%d|%d|" % (q["id"], q["length"])); hsh.update(q["fwd"]); hsh.update(b"|"); hsh.update(q["filename"].encode()); hsh.update(b"|"); hsh.update(q["header"].encode()); hsh.update(b"\n")
        key = hsh.digest()
        hit = _COMPRESS_MEMO.get(key)
        if hit is not None:
            return hit[0], dict(hit[1]), dict(hit[2])
        res = self._compress(k)
        if len(_COMPRESS_MEMO) >= 12:
            _COMPRESS_MEMO.pop(next(iter(_COMPRESS_MEMO)))
        _COMPRESS_MEMO[key] = res
        return res[0], dict(res[1]), dict(res[2])

    def _compress(self, k):
        out = C.c_void_p()
        stats = (C.c_uint64 * 7)()
        times = (C.c_double * 6)()
        _check(lib().orc_compress(self.h, C.c_uint32(k), C.byref(out), stats, times))
        gfa = C.string_at(out.value).decode()
        lib().orc_free(out)
        names = ["kmers", "unitigs_pre", "links_pre", "length_pre", "unitigs_post", "links_post", "length_post"]
        tnames = ["load", "repair", "kmer_graph", "unitig_graph", "simplify", "save"]
        return gfa, dict(zip(names, stats)), dict(zip(tnames, times))

    def metrics_yaml(self, unitig_count, total_length):
        return _str_call(lib().orc_metrics_yaml, self.h, C.c_uint32(unitig_count), C.c_uint64(total_length))

    def __del__(self):
        try:
            lib().orc_seqs_free(self.h)
        except Exception:
            pass


def compress_dir(in_dir, out_dir, k=51, max_contigs=25, threads=8):
    stats = (C.c_uint64 * 7)()
    times = (C.c_double * 6)()
    _check(lib().orc_compress_dir(_b(str(in_dir)), _b(str(out_dir)), C.c_uint32(k), C.c_uint32(max_contigs),
                                  C.c_int(threads), stats, times))
    names = ["kmers", "unitigs_pre", "links_pre", "length_pre", "unitigs_post", "links_post", "length_post"]
    tnames = ["load", "repair", "kmer_graph", "unitig_graph", "simplify", "save"]
    return dict(zip(names, stats)), dict(zip(tnames, times))
