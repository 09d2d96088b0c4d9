// One compress job over several devices: the phases of GraphBuilder between which the caller runs its collectives (torch.distributed /
// RCCL / the host-staged transport of multi_build.cpp) — DESIGN.md §7; the interface and the protocol: graph_build.hpp.
#include "graph_impl.hpp"

namespace ac {

// ---- sharded build (one compress job over several devices; the collectives between the phases belong to the
// caller, e.g. torch.distributed over RCCL) -----------------------------------------------------------------------
void GraphBuilder::shard_begin(uint32_t local_assembly_hint) {
    BuildTimings keep = tm_;
    tm_ = BuildTimings();
    tm_.h2d = keep.h2d;
    tm_.local_hint = local_assembly_hint;
    Impl& m = *impl_;
    m.begin(&tm_);
    m.check_sizes(m.loc);
    m.pack_overlapped(local_assembly_hint);      // (round 5: the tail of the pack under the first insert phase, like a single-device build)
    m.lap(&tm_.pack);
    AC_DISPATCH_W(fragments, (*impl_))
}
uint64_t GraphBuilder::local_distinct_count() const { return tm_.n_local_distinct; }
void GraphBuilder::set_distinct_upper_bound(uint64_t n) { impl_->distinct_upper = n; }
uint64_t GraphBuilder::fragment_text_bytes() const { return impl_->frag_bytes; }
uint64_t GraphBuilder::fragment_count() const { return impl_->n_frags; }
void GraphBuilder::fragments_export(void* d_text_out, void* d_meta_out) {
    Impl& m = *impl_;
    launch((m.frag_bytes + 63) / 64, FragCopyFunctor{m.loc.bits.ptr(), m.loc.mask.ptr(), m.frag_fpos.ptr(), m.frag_boff.ptr(), m.n_frags, m.frag_bytes, (u8*)d_text_out});
    copy_d2d(d_meta_out, impl_->frag_meta.ptr(), impl_->n_frags * 8);
    stream_sync();
}
// The fragment text as 2-bit codes on the union text's word grid (FragPackFunctor): union_off = where this rank's stretch begins in
// the union text ('$' + the ranks' fragment texts in rank order).
uint64_t GraphBuilder::fragment_packed_words(uint64_t union_off) const {
    const u64 n = impl_->frag_bytes;
    return n ? ((union_off + n - 1) >> 5) - (union_off >> 5) + 1 : 0;
}
void GraphBuilder::fragments_export_packed(uint64_t union_off, void* d_words_out, void* d_meta_out) {
    Impl& m = *impl_;
    const u64 nw = fragment_packed_words(union_off);
    if (nw) launch(nw, FragPackFunctor{m.loc.bits.ptr(), m.loc.mask.ptr(), m.frag_fpos.ptr(), m.frag_boff.ptr(), m.n_frags, m.frag_bytes, union_off, (u64*)d_words_out});
    copy_d2d(d_meta_out, m.frag_meta.ptr(), m.n_frags * 8);
    stream_sync();
}
void GraphBuilder::shard_build_union_packed(uint32_t rank, uint32_t n_shards, const void* d_staged_words, const uint64_t* first_word,
                                            const uint64_t* n_words, uint64_t n_union_text, const void* d_meta, uint64_t n_frags_total) {
    build_union_impl(rank, n_shards, nullptr, d_staged_words, first_word, n_words, n_union_text, d_meta, n_frags_total);
}
void GraphBuilder::shard_build_union(uint32_t rank, uint32_t n_shards, const uint8_t* d_union_text, uint64_t n_union_text,
                                     const void* d_meta, uint64_t n_frags_total) {
    build_union_impl(rank, n_shards, d_union_text, nullptr, nullptr, nullptr, n_union_text, d_meta, n_frags_total);
}
void GraphBuilder::build_union_impl(uint32_t rank, uint32_t n_shards, const uint8_t* d_union_text, const void* d_staged_words, const uint64_t* first_word,
                                    const uint64_t* n_words, uint64_t n_union_text, const void* d_meta, uint64_t n_frags_total) {
    if (n_shards == 0 || rank >= n_shards) throw DeviceError("invalid rank / shard count");
    if (n_shards > 255) throw DeviceError("a sharded build takes at most 255 ranks (junction and field owners are bytes)");
    Impl& m = *impl_;
    m.t0 = now_s();
    if (n_frags_total == 0 || n_frags_total >= 0xFFFFFFF0ULL) throw DeviceError("invalid fragment count");
    {   // the fragment table of the union text: lengths, dots and flags from the records, offsets by a scan (UnionMetaFunctor)
        PackedText& u = m.uni;
        u.d_text = d_union_text;
        u.n_text = n_union_text;
        u.alloc_table((u32)n_frags_total);
        DBuf<u64> ext(n_frags_total + 1), ext_scan(n_frags_total + 1), sums(4);
        sums.fill_bytes(0);
        ext.fill_bytes_from(n_frags_total * 8, 0);      // [n] = 0: the exclusive scan then ends with the total
        launch_full(n_frags_total, UnionMetaFunctor{(const u64*)d_meta, n_frags_total, impl_->k, u.seq_len.ptr(), u.seq_d1.ptr(), u.seq_d2.ptr(), u.seq_flags.ptr(), ext.ptr(), sums.ptr()});
        exclusive_scan_u64(ext.ptr(), ext_scan.ptr(), n_frags_total + 1);
        launch(n_frags_total, UnionOffFunctor{ext_scan.ptr(), u.seq_off.ptr()});
        u64 h_sums[4] = {0, 0, 0, 0}, h_total = 0;
        { ReadBatch rb; rb.add(h_sums, sums.ptr(), 32); rb.add(&h_total, ext_scan.ptr() + n_frags_total, 8); rb.run(); }
        if (h_sums[3]) throw DeviceError("invalid fragment record");
        if (h_total + 1 != n_union_text) throw DeviceError("fragment records do not add up to the union text size");
        u.set_sums(h_sums[0], h_sums[1], h_sums[2]);
    }
    m.G = &m.uni;
    tm_.graph_hint = n_shards;
    if (d_union_text) m.uni.pack();
    else {      // the ranks' code words are here already: OR them into place; the mask plane follows from the fragment records
        PackedText& u = m.uni;
        u.pack_alloc();                                                // (clears / sets the slack behind the text)
        flush_fills();                                                 // (the fills of one batch run side by side: the mask's body below overlaps the slack fill's first bytes)
        const u64 groups = (n_union_text + 31) / 32;
        u.bits.fill_bytes(0);
        u.mask.fill_bytes_first(((n_union_text + 63) / 64) * 8, 0);   // (MaskTableFunctor then sets the bits of the text's own words)
        u64 staged_at = 0;
        for (uint32_t r = 0; r < n_shards; r++) {
            if (n_words[r]) {
                if (first_word[r] + n_words[r] > groups) throw DeviceError("fragment words beyond the union text");
                launch(n_words[r], OrWordsFunctor{(const u64*)d_staged_words + staged_at, n_words[r], first_word[r], u.bits.ptr()});
            }
            staged_at += n_words[r];
        }
        launch((u64)u.n_seqs + 1, MaskTableFunctor{u.seq_off.ptr(), u.seq_len.ptr(), u.seq_d1.ptr(), u.seq_d2.ptr(), u.n_seqs, (int)impl_->k, n_union_text, u.mask.ptr()});
        u.packed = true;
    }
    m.lap(&tm_.union_pack);
    m.n_owners = n_shards; m.my_owner = rank;      // this rank's table holds the k-mers whose home hash it owns
    AC_DISPATCH_W(table, (*impl_))
}
uint64_t GraphBuilder::bitmap_words() const { return impl_->uni.n_text / 64 + 2; }
void GraphBuilder::bitmap_export(void* d_out) {      // this rank's novel bits (disjoint from every other rank's: the owners partition the keys)
    copy_d2d(d_out, impl_->bm.ptr(), bitmap_words() * 8);
    stream_sync();
}
// Novel list from the summed bitmap.  With the sibling bits in use (round 5) the degree stage waits for their sum: sib_words() > 0 then,
// and the caller goes sib_export -> all-reduce SUM -> shard_degrees before degrees_export.  Otherwise the degree stage runs here.
void GraphBuilder::shard_build_novel(const void* d_bitmap_sum) {
    Impl& m = *impl_;
    m.t0 = now_s();
    if (d_bitmap_sum) copy_d2d(m.bm.ptr(), d_bitmap_sum, bitmap_words() * 8);
    else if (m.n_owners > 1) throw DeviceError("the novel bitmaps of the other ranks are missing");
    if (m.n_owners > 1) m.novel_list(0);      // (one owner: table() has made the list already)
    if (m.n_owners > 1 && m.sflags.size()) {
        // this rank's sibling bits, two per distinct k-mer, at the novel index of the position their slot ended up holding
        m.sibn.alloc(2 * (m.N / 64 + 2)); m.sibn.fill_bytes(0);
        launch(m.cap, SibByRankFunctor{m.slots.ptr(), m.sflags.ptr(), Novel{m.bm.ptr(), m.wprefix.ptr()}, m.sibn.ptr()});
        m.sib_pending = true;
        m.lap(&tm_.collect_sort);
        return;
    }
    AC_DISPATCH_W(degrees, (*impl_))
}
uint64_t GraphBuilder::sib_words() const { return impl_->sib_pending ? impl_->sibn.size() : 0; }
void GraphBuilder::sib_export(void* d_out) {
    if (!impl_->sib_pending) throw DeviceError("sib_export: no sibling bits to exchange");
    copy_d2d(d_out, impl_->sibn.ptr(), impl_->sibn.size() * 8);
    stream_sync();
}
void GraphBuilder::shard_degrees(const void* d_sib_sum) {
    Impl& m = *impl_;
    m.t0 = now_s();
    if (!m.sib_pending) throw DeviceError("shard_degrees: nothing pending (the degree stage ran in shard_build_novel)");
    if (!d_sib_sum) throw DeviceError("the sibling bits of the other ranks are missing");
    copy_d2d(m.sibn.ptr(), d_sib_sum, m.sibn.size() * 8);
    m.sib_pending = false;
    AC_DISPATCH_W(degrees, (*impl_))
}
uint64_t GraphBuilder::distinct_count() const { return impl_->N; }
// What the degree exchange moves: one byte per k-mer the light degree step left open + four per flagged fragment end (compact form), or a
// byte per distinct k-mer: [first(rc T):1][first(T):1][in:3][out:3] (every degree by probing: AC_SHARD_DEGREE_FLAGS=0, k < 3).
uint64_t GraphBuilder::degree_bytes() const {
    if (impl_->sib_pending) throw DeviceError("degree_bytes: the degree stage has not run (shard_degrees)");
    return impl_->kcontrib.size() ? impl_->n_pending + 4 * impl_->n_first : impl_->N;
}
void GraphBuilder::degrees_export(void* d_out) {
    Impl& m = *impl_;
    if (m.sib_pending) throw DeviceError("degrees_export: the degree stage has not run (shard_degrees)");
    if (m.kcontrib.size()) { const u64 nb = degree_bytes(); if (nb) launch(nb, DegPackFunctor{m.kcontrib.ptr(), m.n_pending, m.kcontrib.ptr() + m.n_pending, m.n_first, (u8*)d_out}); }
    else launch(m.N, KinfoPackFunctor{m.kinfo.ptr(), (u8*)d_out});
    stream_sync();
}
void GraphBuilder::shard_build_graph(const void* d_kinfo_sum) {
    Impl& m = *impl_;
    m.t0 = now_s();
    if (m.sib_pending) throw DeviceError("shard_build_graph: the degree stage has not run (shard_degrees)");
    if (m.kcontrib.size()) {
        if (!d_kinfo_sum) throw DeviceError("the degree contributions of the other ranks are missing");
        launch(m.N, DegUnpackFunctor{(const u8*)d_kinfo_sum, m.pend.ptr(), m.pidx.ptr(), m.kinfo.ptr(), m.counters.ptr() + 3});
        launch(m.n_first, FirstWordsApplyFunctor{(const u8*)d_kinfo_sum, m.n_pending, m.N, m.kinfo.ptr(), m.counters.ptr() + 3});
    } else if (d_kinfo_sum) launch(m.N, KinfoUnpackFunctor{(const u8*)d_kinfo_sum, m.kinfo.ptr(), m.counters.ptr() + 3});
    else if (m.n_owners > 1) throw DeviceError("the degree words of the other ranks are missing");
    m.kcontrib = DBuf<u32>();
    AC_DISPATCH_W(unitigs, (*impl_))
}
void GraphBuilder::links_export(void* d_links_i32, void* d_wlinks_i64) {
    Impl& m = *impl_;
    copy_d2d(d_links_i32, m.links.ptr(), (size_t)m.U * 10 * 4);
    if (d_wlinks_i64) copy_d2d(d_wlinks_i64, m.wlinks.ptr(), (size_t)m.U * 10 * 8);      // (optional: the walk words follow from the link words)
    stream_sync();
}
void GraphBuilder::links_import(const void* d_links_i32, const void* d_wlinks_i64) {
    Impl& m = *impl_;
    m.t0 = now_s();
    if (d_links_i32) {
        // the walk words are a function of the link words and the unitig lengths every rank holds: only the 40 bytes of link words per
        // unitig cross between the ranks, not the 80 bytes of walk words as well (round 5)
        copy_d2d(m.links.ptr(), d_links_i32, (size_t)m.U * 10 * 4);
        if (d_wlinks_i64) copy_d2d(m.wlinks.ptr(), d_wlinks_i64, (size_t)m.U * 10 * 8);
        launch((u64)m.U * 10, LinkSumCheckFunctor{m.links.ptr(), m.U, m.counters.ptr() + 3, d_wlinks_i64 ? nullptr : m.wlinks.ptr(), m.ulen.ptr()});
    } else if (m.n_owners > 1) throw DeviceError("the link words of the other ranks are missing");
    AC_DISPATCH_W(walk_queries, (*impl_))
}
uint64_t GraphBuilder::query_count() const { return impl_->n_queries; }
uint32_t GraphBuilder::query_key_words() const { return (uint32_t)key_words((int)impl_->k); }
void GraphBuilder::queries_export(void* d_out) {
    copy_d2d(d_out, impl_->qkeys.ptr(), impl_->n_queries * query_key_words() * 8);
    stream_sync();
}
void GraphBuilder::answer_queries(const void* d_keys, uint64_t n, void* d_out) {
    AC_DISPATCH_W(answer_queries, (*impl_, (const u64*)d_keys, n, (u64*)d_out))
    stream_sync();
}
void GraphBuilder::queries_route(uint32_t n_shards, void* d_routed_keys, uint64_t* counts_host) {
    if (n_shards == 0 || n_shards != impl_->n_owners) throw DeviceError("queries_route: shard count mismatch");
    AC_DISPATCH_W(route_queries, (*impl_, n_shards, (u64*)d_routed_keys, counts_host))
}
void GraphBuilder::shard_walk_routed(const void* d_routed_answers) {      // answers in the order queries_route sent the keys
    Impl& m = *impl_;
    if (!m.qidx.size() && m.n_queries) throw DeviceError("shard_walk_routed: queries_route has not run");
    m.qanswers.alloc(m.n_queries);
    launch(m.n_queries, AnswerScatterFunctor{(const u64*)d_routed_answers, m.qidx.ptr(), m.qanswers.ptr()});
    shard_walk(m.qanswers.ptr());
}
void GraphBuilder::shard_walk(const void* d_answers_mine) {
    Impl& m = *impl_;
    m.t0 = now_s();
    if (!d_answers_mine) throw DeviceError("the answers to this rank's walk queries are missing");
    m.walk_answers = (const u64*)d_answers_mine;
    AC_DISPATCH_W(walk, (*impl_))
    stream_sync();      // the answers buffer is the caller's
    m.walk_answers = nullptr;
}
uint32_t GraphBuilder::unitig_count() const { return impl_->U; }
void GraphBuilder::reduce_export(int32_t* d_sum, int32_t* d_min) {
    Impl& m = *impl_;
    launch(m.U, ReduceExportFunctor{m.depth.ptr(), m.fs0.ptr(), m.fe0.ptr(), m.minpos_fwd.ptr(), m.minpos_rev.ptr(), m.U, d_sum, d_min});
    stream_sync();
}
void GraphBuilder::reduce_import(const int32_t* d_sum, const int32_t* d_min) {
    Impl& m = *impl_;
    m.t0 = now_s();
    launch(m.U, ReduceImportFunctor{m.depth.ptr(), m.fs0.ptr(), m.fe0.ptr(), m.minpos_fwd.ptr(), m.minpos_rev.ptr(), m.U, d_sum, d_min});
    stream_sync();
}
void GraphBuilder::set_tail_exchange(std::function<void(void*, uint64_t, int, int)> all_reduce) { impl_->tail_xchg = std::move(all_reduce); }
void GraphBuilder::shard_finish(FinalGraph* out, bool want_graph, bool want_paths) {
    impl_->t0 = now_s();
    // a rank that keeps the paths of its own sequences lets the host give them their final numbers, like a single-device build (round 5:
    // PathRemapJob — the entries cross PCIe under the tail instead of behind it); the device copy then stays in seed numbers
    impl_->host_remap_allowed = want_paths && shard_host_remap();
    AC_DISPATCH_W(tail, (*impl_, out, want_graph, want_paths))
}
uint64_t GraphBuilder::path_entry_count() const { return impl_->n_ent; }
void GraphBuilder::paths_export(void* d_out) {
    if (impl_->paths_in_seed_numbers) throw DeviceError("paths_export: this rank kept its own paths (they were renumbered on the host)");
    copy_d2d(d_out, impl_->ent_val.ptr(), impl_->n_ent * 4);
    stream_sync();
}

}  // namespace ac
