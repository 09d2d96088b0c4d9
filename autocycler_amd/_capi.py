"""ctypes binding of libautocycler_hip.so (include/autocycler_hip.h).

There is no Python or CPU implementation behind these calls: if the HIP library has not been built
(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C autocycler_amd/csrc`) importing the
package still works, but any graph build raises HipLibraryMissing."""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libautocycler_hip.so"


class HipLibraryMissing(RuntimeError):
    pass


class AutocyclerError(RuntimeError):
    """The text the reference would hand to quit_with_error (misc.rs:131-137)."""


class SeqView(C.Structure):
    _fields_ = [("fwd", C.c_char_p), ("length", C.c_uint32), ("id", C.c_uint16)]


class Position(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("seq_id_and_strand", C.c_uint16)]


class Link(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32)]      # signed unitig numbers: +n forward strand, -n reverse (ABI 7)


class Stats(C.Structure):
    _fields_ = [("unitigs", C.c_uint32), ("links_one_way", C.c_uint64), ("total_length", C.c_uint64)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("h2d", "pack", "insert", "collect_sort", "degree", "segment", "minkey",
                                          "rank", "paths", "links", "seqs", "d2h", "total_device", "expand",
                                          "insert_kernel_ms")] + \
               [(n, C.c_uint64) for n in ("insert_positions", "table_capacity", "n_distinct", "n_path_entries")] + \
               [("simplify_passes", C.c_uint32), ("insert_launches", C.c_uint32), ("insert_real", C.c_uint64),
                ("analysis", C.c_double), ("finalize", C.c_double), ("n_candidates", C.c_uint32), ("n_levels", C.c_uint32),
                ("fragments", C.c_double), ("union_pack", C.c_double), ("union_insert", C.c_double),
                ("n_local_distinct", C.c_uint64), ("n_fragments", C.c_uint64), ("fragment_bytes", C.c_uint64),
                ("upload_device_ms", C.c_double), ("path_runs_copied", C.c_uint64), ("path_entries_walked", C.c_uint64), ("position_retries", C.c_uint64), ("n_candidates_owned", C.c_uint32),
                ("launches", C.c_uint32), ("readbacks", C.c_uint32), ("n_degrees_open", C.c_uint64), ("sort_retries", C.c_uint64), ("insert_rest_known", C.c_double), ("insert_rest_sampled", C.c_double), ("path_stretches", C.c_uint64),
                ("expand_sparse_sweeps", C.c_uint32), ("expand_sparse_start", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int)      # ac_allreduce_fn

EXPORTS = ["ac_compress_build", "ac_compress_build_multi", "ac_multi_info_get", "ac_compress_build_device", "ac_pack_text", "ac_text_size", "ac_layout_text", "ac_kmer_count",
           "ac_stats_pre", "ac_stats_post", "ac_unitig_count", "ac_unitig", "ac_unitigs_bulk", "ac_paths_bulk", "ac_unitig_positions", "ac_links",
           "ac_path", "ac_timings_get", "ac_timings_get_sized", "ac_free", "ac_gfa_string", "ac_string_free", "ac_last_error",
           "ac_device_count", "ac_max_kmer", "ac_version", "ac_abi_version", "ac_set_host_side_device", "ac_source_hash", "ac_set_stage_timing", "ac_random_access_ceilings", "ac_random_access_ceilings_at", "ac_release_memory", "ac_end_repair_device", "ac_pairwise_distances", "ac_selftest_primitives", "ac_verify_graph", "ac_verify_graph_device", "ac_graph_from_gfa", "ac_graph_kmer_size", "ac_graph_seq_info", "ac_decompress_seq", "ac_decompress_device", "ac_decompress",
           "ac_shard_begin", "ac_shard_fragment_sizes", "ac_shard_fragments_export", "ac_shard_build_union", "ac_shard_fragment_packed_words", "ac_shard_fragments_export_packed", "ac_shard_build_union_packed",
           "ac_shard_unitig_count", "ac_shard_table_capacity", "ac_shard_bitmap_words", "ac_shard_bitmap_export", "ac_shard_build_novel", "ac_shard_sib_words", "ac_shard_sib_export", "ac_shard_degrees",
           "ac_shard_degree_bytes", "ac_multi_info_get_sized", "ac_shard_links_export", "ac_shard_links_import",
           "ac_shard_query_count", "ac_shard_query_key_words", "ac_shard_queries_export", "ac_shard_answer", "ac_shard_walk", "ac_shard_queries_route", "ac_shard_walk_routed", "ac_shard_local_distinct", "ac_shard_set_distinct_upper_bound", "ac_shard_distinct_count", "ac_shard_degrees_export", "ac_shard_build_graph", "ac_gfa_string_parts", "ac_shard_reduce_export", "ac_shard_reduce_import", "ac_shard_finish", "ac_shard_set_allreduce", "ac_device_copy",
           "ac_shard_path_entries", "ac_shard_paths_export", "ac_shard_free", "ac_graph_set_paths", "ac_graph_seq_count", "ac_path_counts",
           "ac_seqs_load", "ac_seqs_from_raw", "ac_seqs_count", "ac_seqs_assembly_count", "ac_seqs_views", "ac_seqs_get",
           "ac_seqs_repair_seconds", "ac_seqs_metrics_yaml", "ac_seqs_free", "ac_compress_seqs", "ac_compress_dir", "ac_compress_dir_multi"]

_libs = {}


def load_library(path=None):
    path = Path(path) if path else LIB_PATH
    key = str(path)
    if key in _libs:
        return _libs[key]
    if not path.exists():
        raise HipLibraryMissing(f"{path} not found: build the HIP extension first (make -C autocycler_amd/csrc). "
                                "There is no CPU fallback.")
    try:
        if os.environ.get("AC_NO_TORCH"):      # torch-free processes (tools/ab_knobs.py): the system HIP runtime is the only one
            raise ImportError
        # PyTorch bundles its own HIP runtime; when ours (/opt/rocm) is loaded first, torch later finds "no HIP GPUs".
        # Importing torch first makes the dynamic loader resolve our libamdhip64 dependency to the copy torch loaded.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(key)
    lib.ac_last_error.restype = C.c_char_p
    lib.ac_version.restype = C.c_char_p
    lib.ac_source_hash.restype = C.c_char_p
    lib.ac_kmer_count.restype = C.c_uint64
    lib.ac_kmer_count.argtypes = [C.c_void_p]
    lib.ac_stats_pre.restype = Stats
    lib.ac_stats_pre.argtypes = [C.c_void_p]
    lib.ac_stats_post.restype = Stats
    lib.ac_stats_post.argtypes = [C.c_void_p]
    lib.ac_unitig_count.restype = C.c_uint32
    lib.ac_unitig_count.argtypes = [C.c_void_p]
    lib.ac_max_kmer.restype = C.c_uint32
    lib.ac_text_size.restype = C.c_uint64
    lib.ac_free.argtypes = [C.c_void_p]
    lib.ac_string_free.argtypes = [C.c_void_p]
    lib.ac_unitig.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.ac_unitig_positions.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.POINTER(Position)), C.POINTER(C.c_uint32)]
    lib.ac_links.argtypes = [C.c_void_p, C.POINTER(C.POINTER(Link)), C.POINTER(C.c_uint64)]
    lib.ac_path.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_uint32)]
    lib.ac_timings_get.argtypes = [C.c_void_p, C.POINTER(Timings)]
    lib.ac_shard_unitig_count.restype = C.c_uint32
    lib.ac_shard_unitig_count.argtypes = [C.c_void_p]
    lib.ac_shard_fragment_packed_words.restype = C.c_uint64
    lib.ac_shard_fragment_packed_words.argtypes = [C.c_void_p, C.c_uint64]
    lib.ac_shard_local_distinct.restype = C.c_uint64
    lib.ac_shard_local_distinct.argtypes = [C.c_void_p]
    lib.ac_shard_set_distinct_upper_bound.argtypes = [C.c_void_p, C.c_uint64]
    lib.ac_shard_set_distinct_upper_bound.restype = None
    lib.ac_shard_distinct_count.restype = C.c_uint64
    lib.ac_shard_distinct_count.argtypes = [C.c_void_p]
    lib.ac_shard_path_entries.restype = C.c_uint64
    lib.ac_shard_path_entries.argtypes = [C.c_void_p]
    lib.ac_shard_free.argtypes = [C.c_void_p]
    for name in ("ac_shard_table_capacity", "ac_shard_bitmap_words", "ac_shard_query_count", "ac_shard_sib_words", "ac_shard_degree_bytes"):
        getattr(lib, name).restype = C.c_uint64
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.ac_shard_query_key_words.restype = C.c_uint32
    lib.ac_shard_query_key_words.argtypes = [C.c_void_p]
    lib.ac_graph_seq_count.restype = C.c_uint32
    lib.ac_graph_seq_count.argtypes = [C.c_void_p]
    _libs[key] = lib
    return lib


def _check(lib, rc):
    if rc != 0:
        raise AutocyclerError(lib.ac_last_error().decode(errors="replace"))


class Graph:
    """Owning handle of an ac_graph (the final UnitigGraph after simplify_structure)."""

    def __init__(self, lib, handle, n_seqs):
        self._lib, self._h, self.n_seqs = lib, handle, n_seqs

    def close(self):
        if self._h:
            self._lib.ac_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def kmer_count(self):
        return self._lib.ac_kmer_count(self._h)

    def _stats(self, fn):
        s = fn(self._h)
        return dict(unitigs=s.unitigs, links=s.links_one_way, total_length=s.total_length)

    @property
    def stats_pre(self):
        return self._stats(self._lib.ac_stats_pre)

    @property
    def stats_post(self):
        return self._stats(self._lib.ac_stats_post)

    @property
    def unitig_count(self):
        return self._lib.ac_unitig_count(self._h)

    def unitig(self, idx):
        p, n, d = C.c_void_p(), C.c_uint32(), C.c_double()
        _check(self._lib, self._lib.ac_unitig(self._h, idx, C.byref(p), C.byref(n), C.byref(d)))
        return C.string_at(p.value, n.value), d.value

    def positions(self, idx, forward):
        p, n = C.POINTER(Position)(), C.c_uint32()
        _check(self._lib, self._lib.ac_unitig_positions(self._h, idx, 1 if forward else 0, C.byref(p), C.byref(n)))
        return [(p[i].seq_id_and_strand & 0x7FFF, bool(p[i].seq_id_and_strand & 0x8000), p[i].pos) for i in range(n.value)]

    def links(self):
        p, n = C.POINTER(Link)(), C.c_uint64()
        _check(self._lib, self._lib.ac_links(self._h, C.byref(p), C.byref(n)))
        return [(abs(p[i].a), p[i].a > 0, abs(p[i].b), p[i].b > 0) for i in range(n.value)]      # (unitig, forward?) pairs like UnitigStrand

    def path(self, seq_index):
        p, n = C.POINTER(C.c_int32)(), C.c_uint32()
        _check(self._lib, self._lib.ac_path(self._h, seq_index, C.byref(p), C.byref(n)))
        return p[:n.value]

    def bulk(self):
        """Zero-copy numpy views (valid while this handle lives): seq_bytes, seq_begin, seq_len, depth, links (structured: a, b — signed
        unitig numbers, negative = reverse strand), path_entries, path_off."""
        import numpy as np
        U = self.unitig_count
        sb, bg, ln, dp = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(self._lib, self._lib.ac_unitigs_bulk(self._h, C.byref(sb), C.byref(bg), C.byref(ln), C.byref(dp)))
        view = lambda ptr, n, ct: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)) if n else np.zeros(0, dtype=ct)
        seq_begin = view(bg, U, C.c_uint64); seq_len = view(ln, U, C.c_uint32); depth = view(dp, U, C.c_double)
        total = int((seq_begin + seq_len).max()) if U else 0
        seq_bytes = view(sb, total, C.c_uint8)
        lp, n = C.POINTER(Link)(), C.c_uint64()
        _check(self._lib, self._lib.ac_links(self._h, C.byref(lp), C.byref(n)))
        ldt = np.dtype([("a", "<i4"), ("b", "<i4")])
        assert ldt.itemsize == C.sizeof(Link)
        links = np.frombuffer((C.c_uint8 * (n.value * C.sizeof(Link))).from_address(C.addressof(lp.contents)), dtype=ldt) if n.value else np.zeros(0, dtype=ldt)
        pe, po, ne = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._lib, self._lib.ac_paths_bulk(self._h, C.byref(pe), C.byref(po), C.byref(ne)))
        S = self._lib.ac_graph_seq_count(self._h)
        return dict(seq_bytes=seq_bytes, seq_begin=seq_begin, seq_len=seq_len, depth=depth, links=links,
                    path_entries=view(pe, ne.value, C.c_int32), path_off=view(po, S + 1, C.c_uint64))

    def path_counts(self):
        n = self._lib.ac_graph_seq_count(self._h)
        out = (C.c_uint64 * n)()
        _check(self._lib, self._lib.ac_path_counts(self._h, out))
        return list(out)

    def decompress(self, seq_index):
        idv, ln = C.c_uint16(), C.c_uint32()
        _check(self._lib, self._lib.ac_graph_seq_info(self._h, C.c_uint32(seq_index), C.byref(idv), C.byref(ln), None, None))
        buf = C.create_string_buffer(ln.value)
        _check(self._lib, self._lib.ac_decompress_seq(self._h, C.c_uint32(seq_index), buf))
        return buf.raw

    def decompress_all(self, device=0):
        """reconstruct_original_sequences for every sequence at once, on the device (ac_decompress_device) -> list of bytes."""
        n = self._lib.ac_graph_seq_count(self._h)
        lens = []
        for i in range(n):
            ln = C.c_uint32()
            _check(self._lib, self._lib.ac_graph_seq_info(self._h, C.c_uint32(i), None, C.byref(ln), None, None))
            lens.append(ln.value)
        buf = C.create_string_buffer(max(sum(lens), 1))
        _check(self._lib, self._lib.ac_decompress_device(self._h, C.c_int(device), buf, C.c_uint64(sum(lens))))
        out, o = [], 0
        for ln in lens:
            out.append(buf.raw[o:o + ln]); o += ln
        return out

    def pairwise_distances(self, device=0):
        """cluster.rs:132-157 -> S x S list of lists (row a, column b)."""
        S = self._lib.ac_graph_seq_count(self._h)
        out = (C.c_double * (S * S))()
        _check(self._lib, self._lib.ac_pairwise_distances(self._h, C.c_int(device), out))
        return [[out[a * S + b] for b in range(S)] for a in range(S)]

    def verify(self, seqs, device=0):
        """ac_verify_graph: the round-trip verifier on the device (decompress identity, check_links, depth, renumber order, statistics).
        seqs: [(padded forward bytes, unpadded length, id)] as for compress_build.  Returns the report as a dict; report["failed"] == 0
        means the graph holds."""
        n = len(seqs)
        views = (SeqView * n)()
        keep = []
        for i, (fwd, length, sid) in enumerate(seqs):
            b = bytes(fwd); keep.append(b)
            views[i].fwd, views[i].length, views[i].id = b, length, sid
        rep = VerifyReport()
        _check(self._lib, self._lib.ac_verify_graph(self._h, views, C.c_uint32(n), C.c_int(device), C.byref(rep)))
        return rep.as_dict()

    def verify_device(self, d_text_ptr, n_text, off, lens, device=0):
        """The same against a text resident on the device (off / lens: ctypes arrays or sequences)."""
        n = len(lens)
        rep = VerifyReport()
        _check(self._lib, self._lib.ac_verify_graph_device(self._h, C.c_void_p(d_text_ptr), C.c_uint64(n_text), (C.c_uint64 * n)(*list(off)),
                                                           (C.c_uint32 * n)(*list(lens)), C.c_uint32(n), C.c_int(device), C.byref(rep)))
        return rep.as_dict()

    def timings(self):
        t = Timings()
        _check(self._lib, self._lib.ac_timings_get(self._h, C.byref(t)))
        return t.as_dict()

    def gfa(self, filenames, headers, parts=3):
        """parts: bit 0 = H, S, L lines; bit 1 = P lines (of the sequences this handle holds paths for)."""
        n = self.n_seqs
        fn = (C.c_char_p * n)(*[f.encode() for f in filenames])
        hd = (C.c_char_p * n)(*[h.encode() for h in headers])
        out, ln = C.c_void_p(), C.c_uint64()
        _check(self._lib, self._lib.ac_gfa_string_parts(self._h, C.c_int(parts), fn, hd, C.byref(out), C.byref(ln)))
        s = C.string_at(out.value, ln.value).decode()
        self._lib.ac_string_free(out)
        return s


def graph_from_gfa(gfa_text, lib_path=None):
    """UnitigGraph::from_gfa_lines for a compress-written GFA -> (Graph, filenames, headers)."""
    lib = load_library(lib_path)
    b = gfa_text.encode() if isinstance(gfa_text, str) else gfa_text
    h = C.c_void_p()
    _check(lib, lib.ac_graph_from_gfa(b, C.c_uint64(len(b)), C.byref(h)))
    n = lib.ac_graph_seq_count(h)
    g = Graph(lib, h, n)
    fns, hds = [], []
    for i in range(n):
        fn, hd = C.c_char_p(), C.c_char_p()
        _check(lib, lib.ac_graph_seq_info(h, C.c_uint32(i), None, None, C.byref(fn), C.byref(hd)))
        fns.append(fn.value.decode()); hds.append(hd.value.decode())
    return g, fns, hds


class VerifyReport(C.Structure):
    _fields_ = [("failed", C.c_uint32)] + [(n, C.c_uint64) for n in ("first_bad_unitig", "first_bad_link", "first_bad_path_entry", "first_bad_sequence",
                                                                      "first_bad_base", "unitigs", "links", "path_entries", "bases_checked",
                                                                      "self_mirror_links")] + [("seconds", C.c_double), ("checks", C.c_uint32),
                                                                                               ("first_bad_junction", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MultiInfo(C.Structure):
    _fields_ = [("n_ranks", C.c_uint32), ("transport", C.c_int)] + \
               [(n, C.c_uint64) for n in ("bytes_fragments", "bytes_bitmap", "bytes_degrees", "bytes_links", "bytes_queries", "bytes_answers",
                                          "bytes_reduce", "queries_total", "queries_sent_away", "table_capacity_max", "table_capacity_sum",
                                          "union_text_bytes", "fragments", "distinct")] + \
               [("seconds_total", C.c_double), ("seconds_exchange_max", C.c_double), ("candidates_total", C.c_uint64), ("candidates_owned_max", C.c_uint64),
                ("bytes_sibling", C.c_uint64), ("bytes_tail", C.c_uint64), ("degrees_open", C.c_uint64), ("bytes_received_max", C.c_uint64), ("path_runs_copied", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def compress_build_multi(k, assembly_count, seqs, devices, lib_path=None):
    """One job over several devices from this one process (ac_compress_build_multi).  seqs as for compress_build; devices: HIP
    ordinals, one per rank (repeats allowed: host-staged exchanges).  Returns (Graph, multi-info dict)."""
    lib = load_library(lib_path)
    seqs = list(seqs)
    arr = (SeqView * len(seqs))()
    keep = []
    for i, (fwd, length, sid) in enumerate(seqs):
        b = bytes(fwd)
        keep.append(b)
        arr[i].fwd, arr[i].length, arr[i].id = b, length, sid
    h = C.c_void_p()
    dv = (C.c_int * len(devices))(*devices)
    _check(lib, lib.ac_compress_build_multi(C.c_uint32(k), C.c_uint32(assembly_count), arr, C.c_uint32(len(seqs)), dv, C.c_int(len(devices)),
                                            C.byref(h)))
    info = MultiInfo()
    _check(lib, lib.ac_multi_info_get(h, C.byref(info)))
    return Graph(lib, h, len(seqs)), info.as_dict()


def compress_build(k, assembly_count, seqs, device=0, lib_path=None):
    """seqs: iterable of (padded_forward_bytes, unpadded_length, seq_id).  Replaces compress.rs:42-44."""
    lib = load_library(lib_path)
    seqs = list(seqs)
    arr = (SeqView * len(seqs))()
    keep = []
    for i, (fwd, length, sid) in enumerate(seqs):
        b = bytes(fwd)
        keep.append(b)
        arr[i].fwd, arr[i].length, arr[i].id = b, length, sid
    h = C.c_void_p()
    _check(lib, lib.ac_compress_build(C.c_uint32(k), C.c_uint32(assembly_count), arr, C.c_uint32(len(seqs)),
                                      C.c_int(device), C.byref(h)))
    return Graph(lib, h, len(seqs))
