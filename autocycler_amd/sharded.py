"""One `autocycler compress` job sharded by SEQUENCE over several MI355X.

One process per GPU.  Rank r holds a contiguous slice of the job's sequences (rank order = sequence order) as a
device-resident text.  The compute is libautocycler_hip.so (ac_shard_* in include/autocycler_hip.h); this module is
only the plumbing between its phases — all-gathers and SUM / MIN all-reduces over torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU test-suite):

    ac_shard_begin          local k-mer insert -> this rank's novel runs ("fragments")
      all-gather            fragment text (2-bit codes) + 8-byte records of every rank  (∝ distinct content, not ∝ input; 0.25 B per base)
    ac_shard_build_union    this rank inserts the union-text k-mers it OWNS (owner = hash of the canonical middle mod world):
                            a table of ~1/world of the job's k-mers
      all-reduce SUM        the ranks' novel bitmaps (disjoint)                     (1 bit per union-text position)
    ac_shard_build_novel    novel list; this rank's sibling bits by novel index
      all-reduce SUM        sibling bits (disjoint)                                 (2 bits per distinct k-mer)
    ac_shard_degrees        degrees: what the sibling bits settle is settled on every rank alike (97-99 %), the rest and the first flags by
                            probing owned groups only
      all-reduce SUM        the probes' contributions, compact                      (1 B per k-mer left open + 4 B per sequence end)
    ac_shard_build_graph    unitigs (identical everywhere); links, probing owned groups only
      all-reduce SUM        link words                                              (40 B per unitig; the walk words are derived on arrival)
    ac_shard_links_import   the keys this rank's path walkers start from
    ac_shard_queries_route  ... ordered by owner rank
      all-to-all            each key to the ONE rank whose table holds it; ac_shard_answer looks them up   (~1/world of the keys per rank)
      all-to-all (reverse)  the answers, in the order the keys were sent           (8 B per 256 input positions, once)
    ac_shard_walk_routed    the paths of the local sequences
      all-reduce SUM, MIN   per-unitig depth / path-end counts, smallest positions  (5 x U int32)
    ac_shard_finish         order-sensitive tail; expand_repeats on this rank's share of the junctions (conflict components), merged by
      all-reduce SUM (x2)   field lengths (12 B per unitig) and sequence bytes, through the callback of ac_shard_set_allreduce
      [gather to root]      optional: paths of all sequences in final numbers -> one rank holds the whole GFA;
                            by default every rank keeps the P lines of its own sequences (Graph.gfa(parts=2))

Nothing here computes: without the library the calls raise HipLibraryMissing."""
import ctypes as C
import time

import torch

from . import _capi


class Comm:
    """The collectives a sharded build needs.  world == 1 (or no process group) degenerates to local copies."""

    def __init__(self, device, group=None, always_collective=False):
        """always_collective: issue the collectives even in a world of one rank (they are local copies otherwise) — how the
        device test-suite drives every RCCL call of this module on the one-GPU box (tests/test_sharded_gpu.py)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.device = torch.device(device)
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
            backend = dist.get_backend(group)
        else:
            self.world, self.rank, backend = 1, 0, None
        self.local_only = self.world == 1 and not (always_collective and backend is not None)
        self.calls = {}
        # gloo moves host memory: device tensors are staged through the host (CPU test-suite, single-GPU dry runs)
        self.stage = (backend == "gloo") and self.device.type != "cpu"
        self.seconds = 0.0

    def _out(self, t):
        return t.cpu() if self.stage else t

    def _back(self, t):
        return t.to(self.device) if self.stage else t

    def all_gather_sizes(self, values):
        """values: list of ints -> list (per rank) of lists."""
        if self.local_only:
            return [list(values)]
        t0 = time.perf_counter()
        self.calls["all_gather_into_tensor"] = self.calls.get("all_gather_into_tensor", 0) + 1
        mine = self._out(torch.tensor(values, dtype=torch.int64, device=self.device))
        out = torch.empty(self.world * len(values), dtype=torch.int64, device=mine.device)
        self.dist.all_gather_into_tensor(out, mine, group=self.group)
        res = out.cpu().view(self.world, len(values)).tolist()
        self.seconds += time.perf_counter() - t0
        return res

    def all_gather_padded(self, t, sizes):
        """t: 1-D tensor holding sizes[rank] elements (may be longer); returns the per-rank slices."""
        if self.local_only:
            return [t[:sizes[0]]]
        t0 = time.perf_counter()
        self.calls["all_gather_into_tensor"] = self.calls.get("all_gather_into_tensor", 0) + 1
        m = max(sizes)
        mine = t if t.numel() == m else torch.cat([t[:sizes[self.rank]], t.new_zeros(m - sizes[self.rank])])
        mine = self._out(mine)
        out = torch.empty(self.world * m, dtype=t.dtype, device=mine.device)
        self.dist.all_gather_into_tensor(out, mine, group=self.group)
        out = self._back(out)
        self.seconds += time.perf_counter() - t0
        return [out[r * m:r * m + sizes[r]] for r in range(self.world)]

    def gather_padded(self, t, sizes, root):
        """Like all_gather_padded, but only `root` receives (others get None)."""
        if self.local_only:
            return [t[:sizes[0]]]
        t0 = time.perf_counter()
        self.calls["gather"] = self.calls.get("gather", 0) + 1
        m = max(sizes)
        mine = t if t.numel() == m else torch.cat([t[:sizes[self.rank]], t.new_zeros(m - sizes[self.rank])])
        mine = self._out(mine)
        if self.rank == root:
            bufs = [torch.empty(m, dtype=t.dtype, device=mine.device) for _ in range(self.world)]
            self.dist.gather(mine, bufs, dst=root, group=self.group)
            res = [self._back(bufs[r])[:sizes[r]] for r in range(self.world)]
        else:
            self.dist.gather(mine, None, dst=root, group=self.group)
            res = None
        self.seconds += time.perf_counter() - t0
        return res

    def all_to_all(self, t, send_counts, recv_counts):
        """t: 1-D tensor, send_counts[r] elements for rank r (in rank order); returns the recv_counts[r] elements of every rank r, in rank order."""
        if self.local_only:
            return t[:send_counts[0]]
        t0 = time.perf_counter()
        self.calls["all_to_all_single"] = self.calls.get("all_to_all_single", 0) + 1
        x = self._out(t[:sum(send_counts)].contiguous())
        out = torch.empty(sum(recv_counts), dtype=t.dtype, device=x.device)
        self.dist.all_to_all_single(out, x, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=self.group)
        out = self._back(out)
        self.seconds += time.perf_counter() - t0
        return out

    def all_reduce(self, t, op):
        if self.local_only:
            return t
        t0 = time.perf_counter()
        self.calls["all_reduce_" + op] = self.calls.get("all_reduce_" + op, 0) + 1
        x = self._out(t)
        self.dist.all_reduce(x, op=getattr(self.dist.ReduceOp, op), group=self.group)
        if self.stage:
            t.copy_(x)
        self.seconds += time.perf_counter() - t0
        return t


class LocalShard:
    """This rank's slice of the job, resident on the device: the arguments of ac_shard_begin."""

    def __init__(self, k, local_assembly_count, d_text, n_text, off, lens, ids, d1, d2):
        self.k, self.local_assembly_count = k, local_assembly_count
        self.d_text, self.n_text = d_text, n_text          # torch uint8 tensor on the device (kept alive here)
        self.n_seqs = len(lens)
        self.off = (C.c_uint64 * self.n_seqs)(*off)
        self.lens = (C.c_uint32 * self.n_seqs)(*lens)
        self.ids = (C.c_uint16 * self.n_seqs)(*ids)
        self.d1 = (C.c_uint16 * self.n_seqs)(*d1)
        self.d2 = (C.c_uint16 * self.n_seqs)(*d2)
        self.bases = int(sum(lens))


def _check(lib, rc):
    if rc != 0:
        raise _capi.AutocyclerError(lib.ac_last_error().decode(errors="replace"))


def sharded_build(lib, shard, comm, device_index=0, root=0, gather_paths=False, partition_tail=True, direct_when_alone=True):
    """Runs one sharded compress build.  Returns (Graph, info).  On `root` the Graph holds unitigs, links and
    statistics; every rank's Graph holds the paths of its own sequences (Graph.gfa(parts=2) -> its P lines), unless
    gather_paths: then root's Graph holds the paths of ALL sequences (Graph.gfa() is the whole file) and the other
    ranks keep statistics only.

    direct_when_alone: a world of ONE rank has nobody to exchange with or to dedup against — the job is a single-device build and
    takes that entry (ac_compress_build_device: no fragments, no union text, no second insert; round 5 — the protocol at world size 1
    cost 1.7x the build it stands for).  False runs every phase of the protocol anyway (the test-suite's coverage of it)."""
    dev = comm.device
    if comm.local_only and direct_when_alone:
        g = C.c_void_p()
        _check(lib, lib.ac_compress_build_device(C.c_uint32(shard.k), C.c_uint32(shard.local_assembly_count), C.c_void_p(shard.d_text.data_ptr()),
                                                 C.c_uint64(shard.n_text), shard.off, shard.lens, shard.ids, shard.d1, shard.d2,
                                                 C.c_uint32(shard.n_seqs), C.c_int(device_index), C.byref(g)))
        graph = _capi.Graph(lib, g, shard.n_seqs)
        tmg = graph.timings()
        return graph, {"fragments": 0, "union_text_bytes": 0, "distinct": tmg["n_distinct"], "unitigs": graph.stats_post["unitigs"], "comm_s": 0.0,
                       "table_capacity": tmg["table_capacity"], "walk_queries": 0, "walk_queries_sent_away": 0,
                       "candidates": tmg["n_candidates"], "candidates_owned": tmg["n_candidates"], "direct": True}
    h = C.c_void_p()
    # A rank whose slice is refused (foreign bytes in a sequence, a layout that does not add up: the user-facing failures of this phase) must
    # not leave the others waiting in the first collective: its status travels with the fragment sizes, and every rank raises.
    begin_rc = lib.ac_shard_begin(C.c_uint32(shard.k), C.c_uint32(shard.local_assembly_count), C.c_void_p(shard.d_text.data_ptr()),
                                  C.c_uint64(shard.n_text), shard.off, shard.lens, shard.ids, shard.d1, shard.d2,
                                  C.c_uint32(shard.n_seqs), C.c_int(device_index), C.byref(h))
    begin_err = lib.ac_last_error().decode(errors="replace") if begin_rc else None
    try:
        # fragments of all ranks -> union text
        nb, nf = C.c_uint64(), C.c_uint64()
        if not begin_rc:
            lib.ac_shard_fragment_sizes(h, C.byref(nb), C.byref(nf))
        nb, nf = nb.value, nf.value
        gathered = comm.all_gather_sizes([nf, nb, 0 if begin_rc else lib.ac_shard_local_distinct(h), 1 if begin_rc else 0])
        failed = [r for r, row in enumerate(gathered) if row[3]]
        if begin_rc:
            raise _capi.AutocyclerError(begin_err)
        if failed:
            raise _capi.AutocyclerError(f"rank {failed[0]} of the sharded build failed in ac_shard_begin (its own exception says why)")
        sizes = [(f, b) for f, b, _, _ in gathered]
        lib.ac_shard_set_distinct_upper_bound(h, C.c_uint64(sum(d for _, _, d, _ in gathered)))
        # the fragment texts travel as 2-bit codes on the union text's word grid (a quarter of the bytes): [records | code words] per rank
        nf_total = sum(f for f, _ in sizes)
        nb_total = 1 + sum(b for _, b in sizes)
        offs = [1 + sum(b for _, b in sizes[:r]) for r in range(comm.world)]       # where each rank's stretch begins in the union text
        first_word = [o >> 5 for o in offs]
        n_words = [(((o + b - 1) >> 5) - (o >> 5) + 1) if b else 0 for o, (_, b) in zip(offs, sizes)]
        assert lib.ac_shard_fragment_packed_words(h, offs[comm.rank]) == n_words[comm.rank]
        mine = torch.empty(8 * (nf + n_words[comm.rank]), dtype=torch.uint8, device=dev)
        _check(lib, lib.ac_shard_fragments_export_packed(h, C.c_uint64(offs[comm.rank]), C.c_void_p(mine.data_ptr() + 8 * nf), C.c_void_p(mine.data_ptr())))
        parts = comm.all_gather_padded(mine, [8 * (f + w) for (f, _), w in zip(sizes, n_words)])
        staged = torch.cat([p[8 * f:] for p, (f, _) in zip(parts, sizes)] + [torch.zeros(8, dtype=torch.uint8, device=dev)]).contiguous()
        meta = torch.cat([p[:8 * f] for p, (f, _) in zip(parts, sizes)]).contiguous()    # own tensor: 8-byte aligned
        _check(lib, lib.ac_shard_build_union_packed(h, C.c_uint32(comm.rank), C.c_uint32(comm.world), C.c_void_p(staged.data_ptr()),
                                                    (C.c_uint64 * comm.world)(*first_word), (C.c_uint64 * comm.world)(*n_words),
                                                    C.c_uint64(nb_total), C.c_void_p(meta.data_ptr()), C.c_uint64(nf_total)))
        del staged
        del parts, mine
        table_capacity = lib.ac_shard_table_capacity(h)
        solo = comm.local_only       # one rank and no forced collectives: nothing to sum
        ptr = lambda t: C.c_void_p(t.data_ptr())
        # novel positions: every rank's share (the k-mers it owns) -> the whole bitmap
        bm = torch.empty(lib.ac_shard_bitmap_words(h), dtype=torch.int64, device=dev)
        if solo:
            _check(lib, lib.ac_shard_build_novel(h, None))
        else:
            _check(lib, lib.ac_shard_bitmap_export(h, ptr(bm)))
            comm.all_reduce(bm, "SUM")          # disjoint bits: the sum is the OR
            _check(lib, lib.ac_shard_build_novel(h, ptr(bm)))
        del bm
        # the sibling bits (2 per distinct k-mer, by novel index): with their sum every rank settles 97-99 % of the degrees without a probe
        N = lib.ac_shard_distinct_count(h)
        sw = lib.ac_shard_sib_words(h)
        if sw:
            sib = torch.empty(sw, dtype=torch.int64, device=dev)
            _check(lib, lib.ac_shard_sib_export(h, ptr(sib)))
            comm.all_reduce(sib, "SUM")         # disjoint bits again
            _check(lib, lib.ac_shard_degrees(h, ptr(sib)))
            del sib
        # degrees + first flags: what the owners' probes found for the k-mers left open (compact: a byte each), or for all of them
        if solo:
            _check(lib, lib.ac_shard_build_graph(h, None))
        else:
            nb = lib.ac_shard_degree_bytes(h)
            deg = torch.zeros(max(nb, 1), dtype=torch.uint8, device=dev)
            _check(lib, lib.ac_shard_degrees_export(h, ptr(deg)))
            comm.all_reduce(deg, "SUM")
            _check(lib, lib.ac_shard_build_graph(h, ptr(deg)))
            del deg
        # links: contributions of the owners
        U = lib.ac_shard_unitig_count(h)
        if solo:
            _check(lib, lib.ac_shard_links_import(h, None, None))
        else:
            lk = torch.empty(10 * U, dtype=torch.int32, device=dev)
            _check(lib, lib.ac_shard_links_export(h, ptr(lk), None))      # (the walk words follow from the link words on arrival: 40 B per unitig cross, not 120)
            comm.all_reduce(lk, "SUM")
            _check(lib, lib.ac_shard_links_import(h, ptr(lk), None))
            del lk
        # where this rank's walkers start: every key goes to the ONE rank that owns it (all-to-all), the answers come back the same way
        nq = lib.ac_shard_query_count(h)
        kw = lib.ac_shard_query_key_words(h)
        keys = torch.empty(max(nq * kw, 1), dtype=torch.int64, device=dev)
        to = (C.c_uint64 * comm.world)()
        _check(lib, lib.ac_shard_queries_route(h, C.c_uint32(comm.world), ptr(keys), to))
        to = [int(x) for x in to]
        frm = [row[comm.rank] for row in comm.all_gather_sizes(to)]       # frm[r]: keys rank r sends here
        in_keys = comm.all_to_all(keys, [c * kw for c in to], [c * kw for c in frm]).contiguous()
        n_in = sum(frm)
        in_ans = torch.empty(max(n_in, 1), dtype=torch.int64, device=dev)
        _check(lib, lib.ac_shard_answer(h, ptr(in_keys), C.c_uint64(n_in), ptr(in_ans)))
        my_ans = comm.all_to_all(in_ans, frm, to).contiguous()
        if my_ans.numel() == 0:
            my_ans = torch.zeros(1, dtype=torch.int64, device=dev)
        _check(lib, lib.ac_shard_walk_routed(h, ptr(my_ans)))
        queries_sent_away = nq - to[comm.rank]
        del keys, in_keys, in_ans, my_ans
        # per-unitig quantities over all sequences
        red = torch.empty(5 * U, dtype=torch.int32, device=dev)
        _check(lib, lib.ac_shard_reduce_export(h, C.c_void_p(red.data_ptr()), C.c_void_p(red.data_ptr() + 12 * U)))
        comm.all_reduce(red[:3 * U], "SUM")
        comm.all_reduce(red[3 * U:], "MIN")
        _check(lib, lib.ac_shard_reduce_import(h, C.c_void_p(red.data_ptr()), C.c_void_p(red.data_ptr() + 12 * U)))
        # the tail: expand_repeats on this rank's share of the junctions, merged by all-reduces the library asks for through this callback
        # (a device buffer of its own: staged through a torch tensor)
        def _allreduce(_user, d_buf, count, dtype, op):
            try:
                t = torch.empty(count, dtype=torch.uint8 if dtype == 0 else torch.int32, device=dev)
                nbytes = count * (1 if dtype == 0 else 4)
                if lib.ac_device_copy(C.c_void_p(t.data_ptr()), C.c_void_p(d_buf), C.c_uint64(nbytes), C.c_int(device_index)):
                    return 1
                comm.all_reduce(t, "SUM" if op == 0 else "MIN")
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                return lib.ac_device_copy(C.c_void_p(d_buf), C.c_void_p(t.data_ptr()), C.c_uint64(nbytes), C.c_int(device_index))
            except Exception:      # noqa: BLE001 — an exception must not unwind through the C frames: the library reports the failure
                import traceback
                traceback.print_exc()
                return 1
        cb = _capi.ALLREDUCE_FN(_allreduce)
        if not comm.local_only and partition_tail:
            _check(lib, lib.ac_shard_set_allreduce(h, cb, None))
        g = C.c_void_p()
        is_root = comm.rank == root
        want = (1 if is_root else 0) | (0 if (gather_paths and not comm.local_only) else 2)
        _check(lib, lib.ac_shard_finish(h, C.c_int(want), C.byref(g)))
        graph = _capi.Graph(lib, g, shard.n_seqs)
        tmg = graph.timings()
        candidates, candidates_owned = tmg["n_candidates"], tmg["n_candidates_owned"]
        if gather_paths and not comm.local_only:      # paths of all sequences -> root
            ne = lib.ac_shard_path_entries(h)
            psz = comm.all_gather_sizes([ne, shard.n_seqs])
            ent = torch.empty(max(ne, 1), dtype=torch.int32, device=dev)
            _check(lib, lib.ac_shard_paths_export(h, C.c_void_p(ent.data_ptr())))
            counts = graph.path_counts()
            info = torch.tensor([[shard.ids[i], shard.lens[i], counts[i]] for i in range(shard.n_seqs)], dtype=torch.int64,
                                device=dev).reshape(-1)
            infos = comm.gather_padded(info, [3 * s for _, s in psz], root)
            ents = comm.gather_padded(ent, [e for e, _ in psz], root)
            if is_root:
                all_info = torch.cat(infos).cpu().view(-1, 3)
                n_total = all_info.shape[0]
                all_ent = torch.cat(ents).contiguous()
                ids = (C.c_uint16 * n_total)(*all_info[:, 0].tolist())
                lens = (C.c_uint32 * n_total)(*all_info[:, 1].tolist())
                cnts = (C.c_uint64 * n_total)(*all_info[:, 2].tolist())
                _check(lib, lib.ac_graph_set_paths(g, C.c_uint32(n_total), ids, lens, cnts, C.c_void_p(all_ent.data_ptr()),
                                                   C.c_int(device_index)))
                graph.n_seqs = n_total
        return graph, {"fragments": nf_total, "union_text_bytes": nb_total, "distinct": N, "unitigs": U, "comm_s": comm.seconds,
                       "table_capacity": table_capacity, "walk_queries": nq, "walk_queries_sent_away": queries_sent_away,
                       "candidates": candidates, "candidates_owned": candidates_owned,
                       "path_runs_copied_here": tmg["path_runs_copied"]}      # (this rank's copying walk: pieces of its followed runs it copied instead of walking)
    finally:
        if h:
            lib.ac_shard_free(h)
