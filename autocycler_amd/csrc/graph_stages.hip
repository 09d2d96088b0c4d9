// The width-dependent stages of the device graph build (GraphBuilder::Impl::insert / fragments / table / degrees / unitigs / walk* /
// tail), behind one explicitly instantiated Stages<W> per key width: compiled once per width (-DAC_W_ONLY=1,2,3,4,8,16) so that the
// widths build in parallel; without AC_W_ONLY (the CPU emulation) one unit instantiates all six.
#include "graph_impl.hpp"

namespace ac {


// renumber_unitigs (unitig_graph.rs:295-315): stable sort of `order` by (length desc, sequence asc, depth desc).
[[maybe_unused]] static bool renum_two_pass() { return knobs().renum_two_pass; }      // 1: always the two-pass renumber sort
[[maybe_unused]] static u32 renum_max_group() { return knobs().renum_max_group; }      // tests: smaller groups take the fallbacks
// deferred: do not wait for the "group too large" flag (a host round trip per renumbering) — the caller reads it with the build's last
// read-back and repeats the build with checked sorts if it was ever set (GraphBuilder::build; the flag is sticky then: never cleared here).
// sc: the sort's scratch if the caller prepared it (RadixScratch); len_bits: no unitig is 2^len_bits long or longer (the keys' leading
// field is ~length: its bits above that are ones in every key and need no pass)
[[maybe_unused]] static void renumber_sort(DBuf<u32>& order, u32 U, const u32* len, const u64* off, const u8* seq, const u32* depth, u32* flag, bool deferred = false,
                                           RadixScratch* sc = nullptr, int len_bits = 32, bool order_is_identity = false) {
    if (U <= 1) return;
    DBuf<u32> backup(deferred && !renum_two_pass() ? 0 : U);      // (the order to fall back from: only a checked sort ever does)
    if (backup.size()) copy_d2d(backup.ptr(), order.ptr(), (size_t)U * 4);
    DBuf<u64> prefix(U), key(U);
    UnitigLess less{len, off, seq, depth};
    u32 zero = 0;
    if (!renum_two_pass()) {      // one sort on (length | 16 bases), ties by the comparator
        // (one kernel where `order` is still the identity — the first renumbering: element i IS unitig i and the sequences are read front to
        // back; the second renumbering's order is by length, and a fused kernel would gather its prefixes from all over the sequences:
        // mini-E's finalize stage 4.4 -> 5.7 ms when it did, r14d)
        if (order_is_identity) launch(U, RenumKeyPassFunctor{RenumKeyFunctor{len, off, seq, prefix.ptr()}, RenumPassFunctor{order.ptr(), len, depth, prefix.ptr(), 2, key.ptr()}});
        else {
            launch(U, RenumKeyFunctor{len, off, seq, prefix.ptr()});
            launch(U, RenumPassFunctor{order.ptr(), len, depth, prefix.ptr(), 2, key.ptr()});
        }
        sort_pairs_u64_u32(key, order, U, 32 + std::min(std::max(len_bits, 1), 32), 0, 0, sc);
        launch(U, RenumTieFunctor{order.ptr(), U, len, depth, prefix.ptr(), less, flag, 0, renum_max_group()});
#ifdef AC_EMU
        if (knobs().degree_diag) {
            u64 groups = 0, members = 0, biggest = 0, cur = 1, longest = 0;
            for (u32 i = 1; i <= U; i++) {
                bool same = i < U && len[order.ptr()[i]] == len[order.ptr()[i - 1]] && (prefix.ptr()[order.ptr()[i]] >> 32) == (prefix.ptr()[order.ptr()[i - 1]] >> 32);
                if (same) cur++;
                else { if (cur > 1) { groups++; members += cur; if (cur > biggest) biggest = cur; if (len[order.ptr()[i - 1]] > longest) longest = len[order.ptr()[i - 1]]; } cur = 1; }
            }
            fprintf(stderr, "renumber diag: U %u, groups %llu, members %llu, biggest %llu, longest member %llu, flag %u\n", U, (unsigned long long)groups,
                    (unsigned long long)members, (unsigned long long)biggest, (unsigned long long)longest, *flag);
        }
#endif
        if (deferred) return;
        if (!read_scalar(flag)) return;
        copy_d2d(order.ptr(), backup.ptr(), (size_t)U * 4);      // a large group of unitigs sharing length and 16 bases: the two-pass form
        copy_h2d(flag, &zero, 4);
    }
    if (renum_two_pass()) launch(U, RenumKeyFunctor{len, off, seq, prefix.ptr()});      // (else the one-pass attempt above left the prefixes)
    for (int pass = 0; pass < 2; pass++) {
        launch(U, RenumPassFunctor{order.ptr(), len, depth, prefix.ptr(), pass, key.ptr()});
        sort_pairs_u64_u32(key, order, U, 64);
    }
    launch(U, RenumTieFunctor{order.ptr(), U, len, depth, prefix.ptr(), less, flag, 1, renum_max_group()});
    if (read_scalar(flag)) {    // a large group of long unitigs sharing length and 32-base prefix: comparator merge sort
        copy_d2d(order.ptr(), backup.ptr(), (size_t)U * 4);
        sort_keys_cmp(order, U, less);
        copy_h2d(flag, &zero, 4);
    }
}

// K2 insert.  Capacity from the reference's own capacity hint (assembly_count, kmer_graph.rs:40): similar assemblies
// share most k-mers.  Overflow -> retry with a larger table.
template <int W>
void GraphBuilder::Impl::insert(const PackedText& pt, u32 hint, DBuf<u64>* slots_out, u64* cap_out, u64* n_distinct_out, DBuf<u64>* bm_out, bool want_sib) {
    TextCtx t = pt.ctx((int)k);
    const u64 p_end_all = pt.n_text - (u64)k + 1;     // one past the last window that fits in the text
    if (hint == 0) hint = 1;
    u64 est = pt.n_bases / hint;
    u64 c = next_pow2(std::max<u64>(1024, est * 3 + 4096));
    if (&pt == &uni && distinct_upper) {      // an upper bound is known (sum of the ranks' local counts); a rank holds about 1/n_owners of the keys
        const u64 mine = distinct_upper / std::max<u32>(n_owners, 1) + distinct_upper / (8 * (u64)std::max<u32>(n_owners, 1)) + 4096;
        c = next_pow2(std::max<u64>(1024, mine * 10 / 7));
    }
    if (c > next_pow2(pt.n_bases * 2 + 1024)) c = next_pow2(pt.n_bases * 2 + 1024);
    // Twice the reference-style capacity (load ~0.23 on similar assemblies: short probe clusters) while that keeps the table around
    // the size of the Infinity Cache; a table that is far beyond it anyway (config D: 4 GB) gains nothing from being sparser and its
    // scans and claims get cheaper when it is not (config D 49.5 -> 45.0 ms per build at shift 0).
    const int shift = table_shift() >= 0 ? table_shift() : (c > (1ULL << 25) ? 0 : 1);
    c <<= shift;
    // the capacity the previous build of a text of this very size ended with (a process that builds the same job again, or a
    // stream of similar jobs, does not pay for the overflow retries twice)
    // (four texts remembered, not one: a sharded build inserts its local slice AND the union text, each with a size of its own — with one slot
    // the two evicted each other and the local insert of a mixed-species job overflowed and started over in every build: E' 5.8 instead of 4.1 ms)
    struct CapMemo { u64 n_text = 0, cap = 0; u32 k = 0; int shift = -2; u32 owners = 0; };
    static thread_local CapMemo memo[4]; static thread_local unsigned memo_next = 0;      // (a capacity is a number, not memory: valid on any device)
    const u32 memo_owners = (&pt == &uni) ? n_owners : 1u;
    for (const CapMemo& m : memo)
        if (pt.n_text == m.n_text && k == m.k && m.shift == table_shift() && m.owners == memo_owners && m.cap > c) c = m.cap;
    // (want_sib = the graph table of a single-device build; a sharded build's LOCAL insert notes its runs too: round 5)
    const int copy_mode = (&pt == &loc && (want_sib || (local_insert_of_shard && shard_path_copy()))) ? path_copy() : 0;
    bool want_runs = false;
    DBuf<InsertStats> istats(257);       // [256].real doubles as the kernel's error word: one D2H reads everything
    DBuf<u64> sl;
    DBuf<u64> nbm(pt.n_text / 64 + 2);   // K3a falls out of the insert: bit p set <=> p is the smallest occurrence of its canonical k-mer
    u64 n_distinct = 0;
    const Arena::Mark retry_mark = Arena::device().mark();      // a retry gives the table it outgrew back (configs[4]: 21 GB of them)
    std::vector<u64> phase_end;
    for (;;) {
        sl.alloc(c);
        sl.fill_bytes(0xFF);
        nbm.fill_bytes(0);
        counters.fill_bytes(0);
        istats.fill_bytes(0);
        if (want_sib) { sflags.alloc(c / 32 + 1); sflags.fill_bytes(0); }
        else sflags = DBuf<u64>();
        u32* ierr = (u32*)&istats.ptr()[256].real;
        Table tb{sl.ptr(), c - 1, nullptr, nbm.ptr(), (&pt == &uni) ? n_owners : 1u, (&pt == &uni) ? my_owner : 0u, want_sib ? sflags.ptr() : nullptr,
                 &istats.ptr()[256].claimed, nullptr, nullptr, 0};
        if (&pt == &loc) { runs = DBuf<u64>(); run_count = DBuf<u32>(); run_rows = run_rows_cap = 0; cplan = CopyPlan(); }      // (the union insert of a sharded build leaves the local insert's runs alone)
        phase_end.clear();
        stream_sync();
#ifndef AC_EMU
        // the dominant kernel's duration, live: one event pair around EVERY phase launch, summed (what sits between the launches — the
        // read-back after the second phase, the wait for the tail of the pack / upload — is not the kernel's time)
        std::vector<hipEvent_t> evs;
        flush_fills();
#endif
        // Phases over geometrically growing prefixes: [0, n/A), [n/A, 2n/A), [2n/A, 4n/A), ...  (A = assembly
        // count): what a phase streams has, for similar assemblies, mostly been inserted by the earlier ones.
        // After the second phase the claim counters say how redundant the text is: if the second stretch (one more
        // assembly's worth) brought few new k-mers, everything that follows mostly matches what is in the table already and
        // goes in ONE launch (measured on config C: 0.89 ms against 1.06 ms for the eight doubling phases); a text that keeps
        // bringing new k-mers stays on the doubling schedule, which bounds the share of a phase that cannot follow runs.
        u32 launches = 0;
        u64 rest_chunk = wave_chunk_rest();
        u64 first = std::max<u64>(p_end_all / hint, 1u << 16);
        u64 pb = 0;
        bool rest_at_once = false;
        while (pb < p_end_all) {
            // (a text that keeps bringing new k-mers — the adaptive test below said no — has little to follow: its later phases are wider,
            // x4 per phase, and cut into more wavefronts: E' 19.50 -> 19.21 ms, mini-E 77.7 -> 76.6, r08k)
            const bool diverse = launches >= 2 && !rest_at_once && insert_adaptive();
            u64 pe = (pb == 0) ? first : pb * (diverse ? std::max<u64>(insert_growth(), 4) : insert_growth());      // (a diverse text: wider phases, r08k)
            if (rest_at_once || pe > p_end_all || p_end_all - pe < (1u << 16)) pe = p_end_all;
#ifndef AC_EMU
            if (upload_pending && &pt == &loc && pe + (u64)k + 8192 > upload_avail) {      // this phase reads beyond the first uploaded chunk
                flush_fills();
                AC_HIP_CHECK(hipStreamWaitEvent(0, (hipEvent_t)upload_done, 0));
                upload_pending = false;
            }
            if (job && &pt == &loc) {
                // the one-launch rest of a redundant text goes out chunk by chunk while the upload is still running: each piece as
                // soon as the chunk it ends in has been sent
                if (rest_at_once) pe = std::min<u64>(pe, upload_rest_limit(pb));
                need_text(pe + (u64)k + 8192);
            }
#endif
            const u64 len = pe - pb;
            {      // one wavefront per chunk: >= ~16 K wavefronts when the phase is long
                u64 c = (len / (diverse ? std::max<u64>(insert_waves_target(), 65536) : insert_waves_target()) + 63) & ~63ULL;      // (... cut into more wavefronts)
                u32 chunk = (u32)std::min<u64>(std::max<u64>(c, 256), rest_at_once ? rest_chunk : wave_chunk_max());
                u64 n_waves = (len + chunk - 1) / chunk;
                if (want_runs && rest_at_once) {
                    // the one-launch rest of a redundant text (with the host entry: its few pieces) notes the runs it follows, a row per
                    // wavefront; reserved with the first piece for twice what the whole rest needs at this piece's chunk length, and a
                    // later piece that would not fit does not note (its text is walked)
                    if (!run_rows_cap) {
                        run_rows_cap = 2 * ((p_end_all - pb) / chunk + 1) + n_waves + 64;
                        // (a short first piece has a short chunk: never more than four times what the longest chunks would need)
                        run_rows_cap = std::min<u64>(run_rows_cap, 4 * (pt.n_text / rest_chunk + 1) + n_waves + 1024);
                        runs.alloc(3 * run_rows_cap * RUN_ROW); run_count.alloc(run_rows_cap + 1);
                        run_count.fill_bytes(0);
                    }
                    if (run_rows + n_waves <= run_rows_cap) { tb.runs = runs.ptr(); tb.run_count = run_count.ptr(); tb.run_row0 = run_rows; run_rows += n_waves; }
                    else tb.runs = nullptr;
                }
                const u64 blocks = (n_waves + 3) / 4;
#ifndef AC_EMU
                if (insert_profile()) {      // measurement only: per-wavefront cycle split of this launch on stderr
                    DBuf<u64> prof(16);
                    prof.fill_bytes(0);
                    launch_wave_kernel(insert_wave_kernel<W, true>, blocks, 0, t, tb, pb, pe, chunk, istats.ptr(), ierr, prof.ptr());
                    std::vector<u64> h = to_host(prof, 16);
                    fprintf(stderr, "insert launch %u: positions %llu chunk %u waves %llu | opener %llu steps avg %.0f cy | wide %llu steps avg %.0f cy | follow %llu runs avg %.0f cy | "
                            "wave avg %.0f cy, longest %llu cy\n", launches, (unsigned long long)len, chunk, (unsigned long long)h[7],
                            (unsigned long long)h[1], h[1] ? (double)h[0] / h[1] : 0.0, (unsigned long long)h[3], h[3] ? (double)h[2] / h[3] : 0.0,
                            (unsigned long long)h[5], h[5] ? (double)h[4] / h[5] : 0.0, h[7] ? (double)h[6] / h[7] : 0.0, (unsigned long long)h[8]);
                } else {
                    hipEvent_t ea, eb;
                    AC_HIP_CHECK(hipEventCreate(&ea)); AC_HIP_CHECK(hipEventCreate(&eb));
                    evs.push_back(ea); evs.push_back(eb);
                    flush_fills();
                    AC_HIP_CHECK(hipEventRecord(ea, 0));
                    launch_wave_kernel(insert_wave_kernel<W, false>, blocks, 0, t, tb, pb, pe, chunk, istats.ptr(), ierr, (u64*)nullptr);
                    AC_HIP_CHECK(hipEventRecord(eb, 0));
                }
#else
                launch_wave_kernel(insert_wave_kernel<W, false>, blocks, 0, t, tb, pb, pe, chunk, istats.ptr(), ierr, (u64*)nullptr);      // the same kernel, lanes in lockstep (wave_rt.hpp)
#endif
            }
            launches++;
            pb = pe;
            phase_end.push_back(pe);
            if (launches == 2 && insert_adaptive() && pb < p_end_all && (p_end_all - pb) > 4 * first) {
                // ... and a sample of the REST looked up in the table as it stands (the first two stretches): how much of what is still
                // to come repeats them.  A rest of copies (one species: ~all found) goes in chunks of 16 K positions; a rest that brings
                // new content of its own (more species behind the first: benchjob8 finds 1 in 8) in shorter ones — a launch keeps
                // ~8 K wavefronts x chunk of text in flight, every copy of a new stretch inside that window inserts it for real, and
                // the shorter chunk is the narrower window (benchjob8 16 K / 8 K / 4 K / 2 K: insert 18.2 / 14.7 / 12.2 / 11.9 ms; config C
                // 1.24 / 1.25 / 1.32 / 1.44, D 10.1 / 10.3 / 11.1 / 12.2: r12v).  Same read-back as the claim counters.
                // (only text that IS on the device is sampled.  With the host entry's overlapped upload — or the device entry's pack
                // beside the first phase — the text behind what stream 0 has waited for is not there yet: the sample takes what the first piece
                // of the rest is about to read anyway (one more chunk at most) and ends where the uploaded text ends; ac_timings.
                // insert_rest_sampled says how much of the rest that was.)
                u64 probe_end = p_end_all;
#ifndef AC_EMU
                if (&pt == &loc) {
                    if (job) {
                        need_text(std::min<u64>(p_end_all, pb + (1u << 20)) + (u64)k + 8192);
                        const u64 avail = job ? std::min<u64>(job->n, job->next_wait * job->CH) : pt.n_text;      // (the last chunk joins the uploaders: job is null then, all text is there)
                        probe_end = std::min<u64>(p_end_all, avail > (u64)k + 64 ? avail - (u64)k - 64 : 0);
                    } else if (upload_pending) probe_end = std::min<u64>(p_end_all, upload_avail > (u64)k + 64 ? upload_avail - (u64)k - 64 : 0);
                }
#endif
                if (probe_end < pb) probe_end = pb;
                DBuf<u32> probe(2);
                probe.fill_bytes(0);
                const u64 n_probe = std::min<u64>(32768, (probe_end - pb) / 4096 + 1);
                if (probe_end > pb) launch(n_probe, RestProbeFunctor<W>{t, tb, pb, probe_end, (probe_end - pb) / n_probe, probe.ptr()});
                tm->insert_rest_sampled = (double)(probe_end - pb) / (double)(p_end_all - pb);
                std::vector<InsertStats> st2(257); u32 h_probe[2] = {0, 0};
                { ReadBatch rb; rb.add(st2.data(), istats.ptr(), 257 * sizeof(InsertStats)); rb.add(h_probe, probe.ptr(), 8); rb.run(); }
                u64 claimed = 0;
                for (size_t q = 0; q < 256; q++) claimed += st2[q].claimed;
                if (st2[256].real == 0 && claimed * 4 <= first * 5) rest_at_once = true;      // <= 25 % of the second stretch was new
                const double known = h_probe[0] ? (double)h_probe[1] / (double)h_probe[0] : 1.0;
                rest_chunk = insert_chunk_rest_env() ? insert_chunk_rest_env() : (known >= 0.9 ? 16384 : known >= 0.6 ? 8192 : known >= 0.3 ? 4096 : 2048);
                tm->insert_rest_known = known;
                // ... and whether the path walk will copy the runs this launch follows (then it has to note them)
                const double r2 = claimed > first ? (double)(claimed - first) / (double)first : 0.0;
                want_runs = rest_at_once && (copy_mode == 1 || (copy_mode == 2 && path_copy_pays(pt.n_text, hint, k, r2)));
                if (knobs().debug_arena) fprintf(stderr, "insert: second stretch %.4f new, one-launch rest %d, copying walk %d (mode %d)\n", r2, (int)rest_at_once, (int)want_runs, copy_mode);
            }
        }
#ifndef AC_EMU
        flush_fills();
        if (!evs.empty()) AC_HIP_CHECK(hipEventSynchronize(evs.back()));
        for (size_t i = 0; i + 1 < evs.size(); i += 2) {
            float ms = 0; AC_HIP_CHECK(hipEventElapsedTime(&ms, evs[i], evs[i + 1]));
            tm->insert_kernel_ms += ms;
        }
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
#endif
        tm->insert_launches += launches;
        std::vector<InsertStats> st = to_host(istats, 257);
        const bool ins_err = st[256].real != 0;
        const u64 full_at = ~st[256].claimed;      // smallest position that found the table full (valid with ins_err)
        st.pop_back();
        n_distinct = 0;
        u64 real = 0;
        for (auto& x : st) { n_distinct += x.claimed; real += x.real; }
        bool overflow = ins_err || (n_distinct * 10 > c * 7);
        if (!overflow) { tm->insert_real += real; tm->insert_positions += pt.n_text; break; }
        const u64 c_max = next_pow2(pt.n_bases * 4 + 1024);
        if (c >= c_max) throw DeviceError("k-mer table overflow");
        // How much larger?  At least four times.  A run that got through knows its k-mer count (target load 0.5); one that filled the
        // table after a fraction of the text extrapolates from the end of the phase it filled it in (the phases run one after the
        // other) — a mixed-species job (configs[4]: 2.1 G distinct k-mers behind a capacity hint of 1000 assemblies) otherwise climbs
        // 32 M -> 128 M -> 512 M -> 2 G -> 8 G slots, re-inserting everything each time.
        u64 want = c * 4;
        if (!ins_err) want = std::max(want, next_pow2(n_distinct * 2));
        else {
            u64 pb_full = 0, pe_full = p_end_all;
            for (u64 e : phase_end) { if (full_at < e) { pe_full = e; break; } pb_full = e; }
            // (the table filled somewhere inside that phase: the geometric mean of its two ends as the text done so far)
            const double done = std::sqrt((double)std::max<u64>(pb_full, pe_full / 4) * (double)pe_full);
            const double need = (double)c * 0.7 * (double)p_end_all / std::max(done, 1.0);
            want = std::max(want, next_pow2((u64)std::min(need * 2.0, 9.0e18)));
        }
        c = std::min(want, c_max);
        tm->insert_kernel_ms = 0; tm->insert_launches = 0;
        stream_sync();
        sl = DBuf<u64>(); sflags = DBuf<u64>();
        Arena::device().rewind(retry_mark);
    }
    if (n_distinct >= 0xFFFFFFF0ULL) throw DeviceError("too many distinct k-mers for 32-bit novel indices");
    // the next build of this text: the capacity that worked — twice that if it ended more than half full (probe sequences at load 0.66
    // instead of 0.33 cost the insert 20-25 % and the probing stages after it as much: mini-E 19.3 -> 15.4 ms, E' 5.45 -> 4.45, r08k)
    {
        CapMemo* slot = nullptr;
        for (CapMemo& m : memo) if (m.n_text == pt.n_text && m.k == k && m.owners == memo_owners) slot = &m;
        if (!slot) slot = &memo[memo_next++ % 4];
        slot->n_text = pt.n_text; slot->k = k; slot->shift = table_shift(); slot->owners = memo_owners;
        slot->cap = (n_distinct * 2 > c && c * 2 <= next_pow2(pt.n_bases * 4 + 1024)) ? c * 2 : c;
    }
    *slots_out = std::move(sl);
    *cap_out = c;
    *n_distinct_out = n_distinct;
    *bm_out = std::move(nbm);
}

// Sharded phase 1 (after the local insert): novel runs of this rank -> fragment text + one meta record per fragment.
template <int W> void GraphBuilder::Impl::fragments() {
    DBuf<u64> lslots; DBuf<u64>& lbm = loc_bm; u64 lcap = 0, ln = 0;      // (the rank's novel bitmap is kept: the copying walk checks its runs against it)
    local_insert_of_shard = true;
    insert<W>(loc, tm->local_hint, &lslots, &lcap, &ln, &lbm);
    local_insert_of_shard = false;
    tm->n_local_distinct = ln;
    lap(&tm->insert);
    u64 nw = loc.n_text / 64 + 1;
    DBuf<u32> ns(nw + 1), ne(nw + 1), so(nw + 1), eo(nw + 1);
    ns.fill_bytes(0); ne.fill_bytes(0);
    launch(nw, RunEdgeCountFunctor{lbm.ptr(), nw, ns.ptr(), ne.ptr()});
    exclusive_scan_u32(ns.ptr(), so.ptr(), nw + 1);
    exclusive_scan_u32(ne.ptr(), eo.ptr(), nw + 1);
    u64 n_runs = read_scalar(so.ptr() + nw);
    if (n_runs != (u64)read_scalar(eo.ptr() + nw)) throw DeviceError("internal error: unbalanced novel runs");
    DBuf<u64> run_start(n_runs), run_end(n_runs);
    launch(nw, RunEdgeFillFunctor{lbm.ptr(), nw, so.ptr(), eo.ptr(), run_start.ptr(), run_end.ptr()});
    n_frags = n_runs + 2 * (u64)loc.n_seqs;
    DBuf<u64> blen(n_frags + 1);
    DBuf<u64>& fpos = frag_fpos; DBuf<u64>& boff = frag_boff;      // (kept: fragments_export / fragments_export_packed read them)
    fpos.alloc(n_frags); boff.alloc(n_frags + 1);
    frag_meta.alloc(n_frags);
    launch(n_frags + 1, FragMetaFunctor{loc.ctx((int)k), run_start.ptr(), run_end.ptr(), n_runs, n_frags, fpos.ptr(), frag_meta.ptr(),
                                        blen.ptr(), counters.ptr() + 6});
    exclusive_scan_u64(blen.ptr(), boff.ptr(), n_frags + 1);
    frag_bytes = read_scalar(boff.ptr() + n_frags);
    {
        u32 frag_err = 0, pack_bad[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        ReadBatch rb;
        rb.add(&frag_err, counters.ptr() + 6, 4);
        if (loc.check_alphabet && loc.pack_bad.size()) rb.add(pack_bad, loc.pack_bad.ptr(), 8);
        rb.run();
        if (loc.pack_bad.size()) loc.verify_alphabet(pack_bad);
        if (frag_err) throw DeviceError("internal error: novel run outside a sequence");
    }
    tm->n_fragments = n_frags; tm->fragment_bytes = frag_bytes;
    lap(&tm->fragments);
}

// K2, K3 on the graph text G: k-mer table and sorted novel list.
template <int W> void GraphBuilder::Impl::table() {
    PackedText& g = *G;
    check_sizes(g);
    // the degree pass's shortcut (sibling bits).  Sharded builds collect them too since round 5: all the k-mers of one middle have one owner,
    // so an owner's table sees every sibling pair; the bits cross between the ranks by novel index (sib_export)
    const bool want_sib = k >= 3 && degree_flags() && (n_owners <= 1 || shard_degree_flags());
    insert<W>(g, tm->graph_hint, &slots, &cap, &N, &bm, want_sib);      // sharded builds: only the k-mers this rank owns (N = how many)
    tm->table_capacity = cap;
    tm->n_distinct = N;
    lap(G == &loc ? &tm->insert : &tm->union_insert);
    // the scan moves the sibling bits the insert left per slot to the text positions the slots ended up holding (a sharded build moves
    // them to NOVEL INDICES once the ranks' bitmaps are summed: shard_build_novel)
    const bool sib_by_pos = want_sib && n_owners <= 1;
    if (sib_by_pos) { sib.alloc(2 * (g.n_text / 64 + 2)); sib.fill_bytes(0); }
    else sib = DBuf<u64>();
    sibn = DBuf<u64>(); sib_pending = false;
    // (round 6 ran this table scan on the side stream beside the novel list — neither reads what the other writes — and took it back: both are
    // bandwidth-bound, collect + degree took 0.51 ms together instead of 0.45 on config C and the same as before on E' / mini-E, r14c)
    occupancy_bitmap(slots, cap, &occ, sib_by_pos ? sflags.ptr() : nullptr, sib_by_pos ? sib.ptr() : nullptr);
    if (n_owners <= 1) novel_list(N);      // a sharded build first sums the ranks' (disjoint) bitmaps: bitmap_import
}

// K5 out/in degrees + K6 first flags of all novel k-mers.  Sharded builds: every rank goes over all of them but only the probes
// its table owns find anything (the known text neighbour of a group is counted by the group's owner too), so the ranks' kinfo
// words are disjoint contributions that add up.
template <int W> void GraphBuilder::Impl::degrees() {
    PackedText& g = *G;
    Table tb = graph_table();
    Novel nv{bm.ptr(), wprefix.ptr()};
    kcontrib = DBuf<u32>(); pend = DBuf<u32>(); pidx = DBuf<u32>(); n_pending = n_first = 0;
    u32* kout = kinfo.ptr();      // where probes and first flags add what they find
    const bool by_index = n_owners > 1 && sibn.size() != 0;      // sharded: the summed sibling bits, by novel index
    const u64* sib_ptr = by_index ? sibn.ptr() : (sib.size() ? sib.ptr() : nullptr);
    EndSet es{nullptr, 0, nullptr, nullptr};
    if (sib_ptr && g.any_dots) {
        // (a sharded build's "sequences" are fragments, most of them without a dot: the set is sized for those that have one)
        endset_mask = next_pow2(4 * (g.has_flags ? g.n_dotted : (u64)g.n_seqs) + 16) - 1;
        endset.alloc((endset_mask + 1) * W);
        endset.fill_bytes(0xFF);
        endset_bloom.alloc(2 * ENDSET_BLOOM_WORDS);
        endset_bloom.fill_bytes(0);
        es = EndSet{endset.ptr(), endset_mask, endset_bloom.ptr(), endset_bloom.ptr() + ENDSET_BLOOM_WORDS};
    }
    // (launched below, behind the allocations of whichever branch runs: their fills then leave in one batch with the set's own)
    auto fill_end_set = [&] { if (es.keys) launch(2 * (u64)g.n_seqs, EndSetFunctor<W>{g.ctx((int)k), es}); };
    const u8* fflags = g.has_flags ? g.seq_flags.ptr() : nullptr;
    if (sib_ptr && degree_flags() == 1 && (!g.any_dots || es.keys)) {      // settle what the sibling bits settle, queue the rest, probe the queues
        // Sharded builds (round 5): the light step is the same on every rank (it reads the text and the summed bit planes only) and its
        // results stay in kinfo; what the probes and the first-flag lookups find is a rank's CONTRIBUTION — only the owner of a probe cluster
        // finds anything in it — and goes to a COMPACT array: a byte for each of the P k-mers the light step left open (1-3 % of them), in
        // novel order (the flags the light step raises, scanned: the same on every rank), and a word for each flagged fragment end.  That
        // array is what degrees_export sends (P + 4 F bytes instead of a byte per distinct k-mer).
        DBuf<u32> fslot;
        if (by_index) {
            pend.alloc(N + 1); pidx.alloc(N + 1);
            pend.fill_bytes_from(N * 4, 0);
            fslot.alloc((u64)g.n_seqs + 1);
            DBuf<u32> fcnt((u64)g.n_seqs + 1);
            fill_end_set();
            launch((u64)g.n_seqs + 1, FirstSlotCountFunctor{fflags, g.n_seqs, fcnt.ptr()});
            exclusive_scan_u32(fcnt.ptr(), fslot.ptr(), (u64)g.n_seqs + 1);
        }
        const Arena::Mark deg_mark = Arena::device().mark();      // the queues below are the stage's own (8 B per distinct k-mer)
        DegWork wk;
        // a k-mer whose window holds dots starts within k - 1 positions of a sequence end: at most 2 (k - 1) per sequence
        const u64 max_generic = std::min<u64>(N, 2 * ((u64)k - 1) * g.n_seqs);
        wk.rcap[0] = (u32)(N / DEG_REGIONS + N / (4 * DEG_REGIONS) + 64 * DEG_BATCH); wk.ocap[0] = N;
        wk.rcap[1] = (u32)(max_generic / DEG_REGIONS + 64 * DEG_BATCH); wk.ocap[1] = max_generic;
        if (degree_region_cap()) { wk.rcap[0] = std::min(wk.rcap[0], degree_region_cap()); wk.rcap[1] = std::min(wk.rcap[1], degree_region_cap()); }      // tests: regions spill
        DBuf<u64> items(wk.words()); DBuf<u32> counts(DEG_LISTS * (DEG_REGIONS + 1));
        counts.fill_bytes(0);
        wk.items = items.ptr(); wk.counts = counts.ptr();
        if (!by_index) fill_end_set();
        const u64 n_thr = (((N + DEG_BATCH - 1) / DEG_BATCH) + 63) & ~63ULL;
        launch_full(n_thr, DegreeLightFunctor<W>{g.ctx((int)k), npos.ptr(), kinfo.ptr(), g.any_dots, bm.ptr(), sib_ptr, es, wk, N, n_thr, by_index ? 1 : 0,
                                                 by_index ? pend.ptr() : nullptr});
        DBuf<u32> kc_tmp;
        if (by_index) {
            exclusive_scan_u32(pend.ptr(), pidx.ptr(), N + 1);
            u32 hp[2];
            { ReadBatch rb; rb.add(&hp[0], pidx.ptr() + N, 4); rb.add(&hp[1], fslot.ptr() + g.n_seqs, 4); rb.run(); }
            n_pending = hp[0]; n_first = hp[1];
            kc_tmp.alloc(n_pending + n_first + 1); kc_tmp.fill_bytes(0);
            kout = kc_tmp.ptr();
        }
        launch((u64)DEG_LISTS * DEG_REGIONS * DEG_PROBE_THREADS, DegreeProbeFunctor<W>{g.ctx((int)k), tb, npos.ptr(), kout, g.any_dots, wk, es, by_index ? pidx.ptr() : nullptr});
#ifdef AC_EMU
        if (knobs().degree_diag) {
            u64 c0 = 0, c1 = 0;
            for (u32 r = 0; r <= DEG_REGIONS; r++) { c0 += wk.count(0)[r]; c1 += wk.count(1)[r]; }
            fprintf(stderr, "degree diag: N %llu, queued real %llu, generic %llu, left open %llu, any_dots %d\n", (unsigned long long)N,
                    (unsigned long long)c0, (unsigned long long)c1, (unsigned long long)n_pending, (int)g.any_dots);
        }
#endif
        if (by_index) launch(g.n_seqs, FirstFunctor<W>{g.ctx((int)k), tb, nv, kinfo.ptr(), fflags, fslot.ptr(), kc_tmp.ptr() + n_pending});
        items = DBuf<u64>(); counts = DBuf<u32>();
        const u32* kc_src = kc_tmp.ptr();
        kc_tmp = DBuf<u32>();
        Arena::device().rewind(deg_mark);
        if (by_index) {      // the compact array moves to where the queues began (it lay behind them: 4 (P + F) bytes against >= 8 N of queues)
            kcontrib.alloc(n_pending + n_first + 1);
            const u64 kc_bytes = (n_pending + n_first + 1) * 4;
            if (kcontrib.ptr() != kc_src) {
                // (nothing but sizes promised that the two do not overlap — F can reach twice the fragment count, AC_DEGREE_REGION_CAP
                // shrinks the queues — and an overlapping device-to-device copy is undefined: then through a buffer behind both)
                const u8* dst_b = (const u8*)kcontrib.ptr(); const u8* src_b = (const u8*)kc_src;
                if (dst_b + kc_bytes <= src_b || src_b + kc_bytes <= dst_b) copy_d2d(kcontrib.ptr(), kc_src, kc_bytes);
                else {
                    const Arena::Mark bounce_mark = Arena::device().mark();
                    DBuf<u32> pad((u64)(src_b + kc_bytes - dst_b) / 4 + 1), bounce(n_pending + n_first + 1);      // (pad: up to the end of the source, so that the bounce lies behind it)
                    copy_d2d(bounce.ptr(), kc_src, kc_bytes);
                    copy_d2d(kcontrib.ptr(), bounce.ptr(), kc_bytes);
                    stream_sync();
                    pad = DBuf<u32>(); bounce = DBuf<u32>();
                    Arena::device().rewind(bounce_mark);
                }
            }
            tm->n_degrees_open = n_pending;
            lap(&tm->degree);
            return;
        }
    } else {
        fill_end_set();
        launch(N, DegreeFunctor<W>{g.ctx((int)k), tb, npos.ptr(), kinfo.ptr(), g.any_dots, 0, bm.ptr(), sib.size() && n_owners <= 1 ? sib.ptr() : nullptr, es});
    }
    launch(g.n_seqs, FirstFunctor<W>{g.ctx((int)k), tb, nv, kout, fflags, nullptr, nullptr});
    lap(&tm->degree);
}

// K6..K11 on G: first flags, unitigs in seed order, links by successor symbol.
template <int W> void GraphBuilder::Impl::unitigs() {
    PackedText& g = *G;
    TextCtx t = g.ctx((int)k);
    Table tb = graph_table();
    Novel nv{bm.ptr(), wprefix.ptr()};
    // K7 heads -> unitig ids
    // (head / scan: allocated, and their tails cleared, by novel_list)
    launch(N, HeadFunctor{npos.ptr(), kinfo.ptr(), head.ptr(), N});
    inclusive_scan_u32(head.ptr(), scan.ptr(), N);
    U = read_scalar(scan.ptr() + (N - 1));
    ustart.alloc((u64)U + 1);
    // (what the seed order needs cleared — its sort's scratch, the flag of its tie-break — now, with this launch's batch)
    int seed_keep = seed_prefix_bits();
    if (seed_keep <= 0) { int lg = 1; while ((1ULL << lg) < (u64)U) lg++; seed_keep = std::min(64, ((2 * lg + 8 + 7) / 8) * 8); }
    RadixScratch seed_rs; DBuf<u32> seed_big(1, true);
    if (seed_prefix_sort()) seed_rs.prepare(U, seed_keep);
    launch(N, UnitigStartFunctor{head.ptr(), scan.ptr(), ustart.ptr(), N});
    lap(&tm->segment);

    // K8 min canonical k-mer per unitig
    DBuf<MinVal<W>> umin(U);
    // Automatic: the prefix form pays one full key and one join per UNITIG to save a full key per K-MER — it wins with long keys and
    // long unitigs (config D, k = 101, 128 k-mers per unitig: 2.13 -> 1.62 ms) and loses with short ones (config C, k = 51, 49 per
    // unitig: 0.105 -> 0.157 ms; E', 5 per unitig: 0.56 -> 0.82 ms); keys wider than four words have no register form at all.
    const int mk = minkey_variant() >= 0 ? minkey_variant() : ((W > 4 || (W >= 3 && N >= 32 * (u64)U)) ? 2 : 1);
    if (mk == 2) {      // prefix form: (f, index, mark) per piece, one full key per unitig
        const u64 n_waves = (N + 63) / 64;
        DBuf<MinPre> upre(U), wfirst(n_waves), wlast(n_waves);
        MinPreArgs a{t, npos.ptr(), scan.ptr(), N, upre.ptr(), wfirst.ptr(), wlast.ptr(), minkey_prefix_bases()};
        launch_wave_kernel(minpre_wave_kernel<W>, (N + 255) / 256, 0, a);
        launch(U, MinFinishFunctor<W>{t, npos.ptr(), ustart.ptr(), U, N, upre.ptr(), wfirst.ptr(), wlast.ptr(), umin.ptr()});
    } else if constexpr (W <= 4) {
        if (mk == 1) {      // wavefront form: keys stay in registers
            const u64 n_waves = (N + 63) / 64;
            DBuf<MinVal<W>> wfirst(n_waves), wlast(n_waves);
            MinWaveArgs<W> a{t, npos.ptr(), scan.ptr(), N, umin.ptr(), wfirst.ptr(), wlast.ptr()};
            launch_wave_kernel(minkey_wave_kernel<W>, (N + 255) / 256, 0, a);
            launch(U, MinJoinFunctor<W>{ustart.ptr(), U, N, wfirst.ptr(), wlast.ptr(), umin.ptr()});
        } else {
            DBuf<MinVal<W>> vals(N); DBuf<u32> seg(N);
            launch(N, CKeyFunctor<W>{t, npos.ptr(), scan.ptr(), vals.ptr(), seg.ptr()});
            reduce_by_segment(seg.ptr(), vals.ptr(), N, umin.ptr(), U, MinOp<W>(), counters.ptr() + 3);
        }
    } else {      // wide keys: arg-min over indices, the keys recomputed from the text inside the operator
        DBuf<u32> umin_idx(U);
        segment_argmin(scan.ptr(), N, umin_idx.ptr(), U, MinIdxOp<W>{t, npos.ptr()}, counters.ptr() + 3);      // scan[i] = unitig of novel k-mer i
        launch(U, UnitigMinFunctor<W>{t, npos.ptr(), umin_idx.ptr(), umin.ptr()});
    }
    lap(&tm->minkey);

    // K9 seed order = rank of the smallest k-mer
    order.alloc(U);
    bool seeds_ordered = false;
    if (seed_prefix_sort()) {      // one sort on a 64-bit prefix of the seed keys, ties on full keys: any key width, any number of unitigs
        DBuf<u64> wkey(U);
        // as many leading bits of the prefix as tell U seeds apart with a few ties to spare (twice log2 U, and a byte for the bias of a
        // MINIMUM towards small values): the ties are ranked on full keys anyway (SeedTieFunctor), and every digit less is a pass less
        const int keep = seed_keep;
        DBuf<u32> by_prefix(U);
        launch(U, SeedPrefixFunctor<W>{umin.ptr(), (int)k, wkey.ptr(), keep, by_prefix.ptr()});      // (... and the identity the sort permutes)
        sort_pairs_u64_u32(wkey, by_prefix, U, 64, 0, 64 - keep, &seed_rs);
        DBuf<u32> settled(U); DBuf<u32>& big = seed_big;
        // (a single-device build does not wait for the "group too large" flag: it is read with the build's last read-back, and a build in
        // which it was set is repeated with checked sorts — one host round trip less here, two in the renumberings)
        const bool defer = deferred_sort_checks();
        launch(U, SeedTieFunctor<W>{by_prefix.ptr(), wkey.ptr(), U, umin.ptr(), settled.ptr(), seed_max_group(), defer ? sort_flags.ptr() : big.ptr()});
        if (defer || read_scalar(big.ptr()) == 0) {
            order = std::move(settled);
            DBuf<MinVal<W>> sorted(U);
            launch(U, GatherMinFunctor<W>{order.ptr(), umin.ptr(), sorted.ptr()});
            umin = std::move(sorted);
            seeds_ordered = true;
        }      // else: a huge group of equal prefixes — `order` is still the identity: the full-key sorts below
    }
    if (!seeds_ordered) launch(U, IotaFunctor{order.ptr()});      // (the full-key sorts permute the identity)
    if (seeds_ordered) {
    } else if constexpr (W <= 4) {
        if ((u64)U >= seed_radix_limit() || seed_prefix_sort()) {      // many unitigs (or the prefix sort's fallback) (mixed-species graphs: millions): W stable LSD radix passes over the key words
            DBuf<u64> wkey(U);                    // (the comparator merge sort takes 2.4 ms for 3.5 M seeds, 5.3 ms for 6.5 M)
            for (int word = W - 1; word >= 0; word--) {
                launch(U, MinWordFunctor<W>{order.ptr(), umin.ptr(), word, wkey.ptr()});
                sort_pairs_u64_u32(wkey, order, U, 64);
            }
            DBuf<MinVal<W>> sorted(U);
            launch(U, GatherMinFunctor<W>{order.ptr(), umin.ptr(), sorted.ptr()});
            umin = std::move(sorted);
        } else {
            sort_by_key_cmp(umin, order, U, MinValLess<W>());
        }
    } else {      // wide keys stay where they are: sort the indices, then gather
        sort_keys_cmp(order, U, MinValIdxLess<W>{umin.ptr()});
        DBuf<MinVal<W>> sorted(U);
        launch(U, GatherMinFunctor<W>{order.ptr(), umin.ptr(), sorted.ptr()});
        umin = std::move(sorted);
    }
    rank.alloc(U); ulen.alloc(U); ustartpos.alloc(U); useq_off.alloc((u64)U + 1); uorient.alloc(U);
    DBuf<u64> ulen64((u64)U + 1);
    launch((u64)U + 1, UnitigMetaFunctor<W>{order.ptr(), ustart.ptr(), npos.ptr(), umin.ptr(), U, N, rank.ptr(), ulen.ptr(),
                                            ulen64.ptr(), ustartpos.ptr(), uorient.ptr()});
    exclusive_scan_u64(ulen64.ptr(), useq_off.ptr(), (u64)U + 1);
    UnitigCtx uc{head.ptr(), scan.ptr(), rank.ptr(), uorient.ptr(), ustart.ptr(), ulen.ptr(), U, N};
    lap(&tm->rank);

    // K11 links by successor symbol
    links.alloc((u64)U * 10); wlinks.alloc((u64)U * 10);
    DBuf<V16> ui(U);      // (16 bytes per unitig; the walk builds its own with the destination flags the links decide)
    launch(U, WalkInfoFunctor{uc, nullptr, ui.ptr()});
    launch((u64)U * 2, LinksFunctor<W>{t, tb, nv, uc, order.ptr(), npos.ptr(), g.any_dots, links.ptr(), wlinks.ptr(), counters.ptr() + 3, ui.ptr()});
    lap(&tm->links);
}

// Sharded builds: the keys of this rank's walker starts (see WalkQueryFunctor), and the owned answers to a batch of such keys.
template <int W> void GraphBuilder::Impl::walk_queries() {
    const u32 PC = path_chunk(N, U);
    u64 n_walkers = (loc.n_text + PC - 1) / PC;
    // the copying walk for this rank's sequences (round 5): the local insert noted its runs; if enough of them repeat first occurrences of
    // this rank's own text, the walkers are the gaps' walkers and only THEIR first k-mers are asked for
    cplan = CopyPlan();
    if (run_rows && loc_bm.size() && PC <= 65535) {
        const u64 nw = loc.n_text / 64 + 1;
        DBuf<u32> wcnt(nw + 1);
        loc_wprefix.alloc(nw + 1);
        launch(nw + 1, PopcFunctor{loc_bm.ptr(), wcnt.ptr(), nw});
        exclusive_scan_u32(wcnt.ptr(), loc_wprefix.ptr(), nw + 1);
        if (walk_copy_prepare<W>(PC, Novel{loc_bm.ptr(), loc_wprefix.ptr()})) n_walkers = cplan.NW;
    }
    n_queries = n_walkers + loc.n_seqs;
    qkeys.alloc(n_queries * W);
    launch(n_queries, WalkQueryFunctor<W>{loc.ctx((int)k), PC, n_walkers, qkeys.ptr(), cplan.ok ? cplan.w_begin.ptr() : nullptr});
}
// The queries in owner order (stable): d_routed_keys[i] = key of query qidx[i]; counts_host[o] = how many go to owner o.
template <int W> void GraphBuilder::Impl::route_queries(u32 n_shards, u64* d_routed_keys, u64* counts_host) {
    DBuf<u64> owner64(n_queries), first(n_shards);
    qidx.alloc(n_queries);
    first.fill_bytes(0xFF);
    launch(n_queries, QueryOwnerFunctor<W>{qkeys.ptr(), (int)k, n_shards, my_owner, owner64.ptr(), qidx.ptr()});
    int bits = 1;
    while ((1u << bits) < n_shards) bits++;
    sort_pairs_u64_u32(owner64, qidx, n_queries, bits);
    launch(n_queries, OwnerBoundsFunctor{owner64.ptr(), first.ptr()});
    launch(n_queries, QueryGatherFunctor<W>{qkeys.ptr(), qidx.ptr(), d_routed_keys});
    std::vector<u64> h_first = to_host(first, n_shards);
    u64 end = n_queries;
    for (u32 o = n_shards; o-- > 0;) {
        if (h_first[o] == ~0ULL) { counts_host[o] = 0; continue; }
        counts_host[o] = end - h_first[o];
        end = h_first[o];
    }
}
template <int W> void GraphBuilder::Impl::answer_queries(const u64* d_keys, u64 n, u64* d_out) {
    launch(n, AnswerFunctor<W>{G->ctx((int)k), graph_table(), d_keys, d_out});
}

// K10c: walk the text between the insert's followed runs, copy the runs' entries from the stretches they repeat (kernels_paths.inc).
// First half: which pieces of the runs are usable (nv_text: the novel bitmap of the TEXT THE RUNS LIE IN — the graph's on a single device,
// the rank's own in a sharded build), the gaps between them cut into walkers.  Everything it keeps is in `cplan`.
template <int W> bool GraphBuilder::Impl::walk_copy_prepare(u32 PC, const Novel& nv_text) {
    TextCtx t = loc.ctx((int)k);
    cplan = CopyPlan();
    // the runs in text order: the rows of the insert's wavefronts one behind the other
    const Arena::Mark mk = Arena::device().mark();
    DBuf<u32> rfirst(run_rows + 1);
    exclusive_scan_u32(run_count.ptr(), rfirst.ptr(), run_rows + 1);      // (run_count[run_rows] is a zero the insert never touches)
    const u64 R0 = read_scalar(rfirst.ptr() + run_rows);
    if (R0 == 0 || R0 >= 0xFFFFFFF0ULL) { Arena::device().rewind(mk); return false; }
    DBuf<RunRec> sorted(R0);
    launch(run_rows * RUN_ROW, RunGatherFunctor{runs.ptr(), run_count.ptr(), rfirst.ptr(), sorted.ptr()});
    DBuf<u32> ok(R0 + 1), at(R0 + 1); DBuf<u64> covered(1, true); DBuf<u32> overlap(1, true);
    ok.fill_bytes(0);
    DBuf<RunRec> fixed(R0); DBuf<u32> fseq(R0);
    launch(R0, RunFilterFunctor{sorted.ptr(), fixed.ptr(), R0, nv_text, loc.n_text, ok.ptr(), covered.ptr(), t, overlap.ptr(), run_piece(), fseq.ptr()});
    exclusive_scan_u32(ok.ptr(), at.ptr(), R0 + 1);
    // pieces, and the gaps between them cut into walkers — launched over a bound on the number of pieces, so that their number, the
    // positions they cover and the number of walkers reach the host in ONE read-back
    const u64 Rb = R0 + loc.n_text / run_piece() + 1;
    CopyPlan& c = cplan;
    c.rr.alloc(Rb); c.rseq.alloc(Rb);
    launch(R0, RunCompactFunctor{fixed.ptr(), ok.ptr(), at.ptr(), c.rr.ptr(), run_piece(), fseq.ptr(), c.rseq.ptr(), nv_text, loc.n_text});
    DBuf<u64> gw(Rb + 2);
    c.wfirst.alloc(Rb + 2);
    launch(Rb + 2, GapWalkersFunctor{c.rr.ptr(), at.ptr() + R0, loc.n_text, PC, gw.ptr()});
    exclusive_scan_u64(gw.ptr(), c.wfirst.ptr(), Rb + 2);
    u64 h_cov = 0, NW = 0; u32 h_R = 0, h_overlap = 0;
    { ReadBatch rb; rb.add(&h_cov, covered.ptr(), 8); rb.add(&h_R, at.ptr() + R0, 4); rb.add(&h_overlap, overlap.ptr(), 4); rb.add(&NW, c.wfirst.ptr() + (Rb + 1), 8); rb.run(); }
    const u64 R = h_R;
    if (knobs().debug_arena) fprintf(stderr, "path copy: %llu runs on the list, %llu pieces usable, covering %llu of %llu positions, %llu walkers\n", (unsigned long long)R0, (unsigned long long)R, (unsigned long long)h_cov, (unsigned long long)loc.n_text, (unsigned long long)NW);
    if (R == 0 || h_overlap || h_cov * 2 < loc.n_text || NW == 0 || NW >= 0xFFFFFFF0ULL) {      // little to copy: the plain walk
        cplan = CopyPlan();
        Arena::device().rewind(mk);
        return false;
    }
    c.w_begin.alloc(NW); c.w_end.alloc(NW); c.w_gap.alloc(NW);
    launch(NW, WalkerRangeFunctor{c.rr.ptr(), R, loc.n_text, PC, c.wfirst.ptr(), c.w_begin.ptr(), c.w_end.ptr(), c.w_gap.ptr()});
    c.R = R; c.NW = NW; c.Rb = Rb; c.ok = true;
    return true;
}
// Second half: the gap walkers (their first lookups answered by the owners beforehand in a sharded build: walk_answers), then the copies.
template <int W> void GraphBuilder::Impl::walk_copy_finish(u32 PC) {
    TextCtx t = loc.ctx((int)k), g = G->ctx((int)k);
    Table tb = graph_table();
    Novel nv{bm.ptr(), wprefix.ptr()};
    UnitigCtx uc{head.ptr(), scan.ptr(), rank.ptr(), uorient.ptr(), ustart.ptr(), ulen.ptr(), U, N};
    CopyPlan& c = cplan;
    const u64 R = c.R, NW = c.NW;
    DBuf<u64> wcount(NW + 1), woff(NW + 1);
    const u64 n_slots = ((NW + 63) / 64) * 64 * PC;
    DBuf<int32_t> stage(n_slots); DBuf<u16> stage_off(n_slots);
    DBuf<u32> seq_tid(loc.n_seqs), seq_j(loc.n_seqs);
    wcount.fill_bytes(0);
    const bool filter = maybe_dest_valid;
    walk_tables(filter);
    walked_plain = false;
    DBuf<V16> uinfo(U);
    launch(U, WalkInfoFunctor{uc, filter ? maybe_dest.ptr() : nullptr, uinfo.ptr()});
    launch(NW, PathWalkFunctor<W>{t, g, tb, nv, uc, uinfo.ptr(), wl_text.ptr(), PC, stage.ptr(), wcount.ptr(), seq_tid.ptr(), seq_j.ptr(),
                                 depth_u.ptr(), minpos_fwd_u.ptr(), minpos_rev_u.ptr(), counters.ptr() + 4, filter ? maybe_dest.ptr() : nullptr,
                                 pos_cap_now, 0, walk_answers, NW, c.w_begin.ptr(), c.w_end.ptr(), stage_off.ptr(), nullptr, nullptr});
    exclusive_scan_u64(wcount.ptr(), woff.ptr(), NW + 1);
    const u64 NE = read_scalar(woff.ptr() + NW);      // walked entries
    DBuf<int32_t> ent(NE); DBuf<u64> ent_pos(NE), ent_end(NE); DBuf<u8> ent_want(NE); DBuf<u32> ent_gap(NE);
    launch_full(NW, WalkCompactFunctor{stage.ptr(), stage_off.ptr(), wcount.ptr(), woff.ptr(), c.w_begin.ptr(), c.w_gap.ptr(), PC, NW, uinfo.ptr(),
                                       ent.ptr(), ent_pos.ptr(), ent_end.ptr(), ent_want.ptr(), ent_gap.ptr()});
    // what every run copies; entries per segment; the final array
    DBuf<u64> ra(R), rcnt(R + 1), seg(2 * R + 2), segoff(2 * R + 2); DBuf<u32> cov(NE + 1), copies(NE + 1);
    cov.fill_bytes(0);
    launch(R, RunRangeFunctor{c.rr.ptr(), ent_pos.ptr(), ent_end.ptr(), NE, ra.ptr(), rcnt.ptr(), cov.ptr()});
    launch(2 * R + 2, SegCountFunctor{c.wfirst.ptr(), woff.ptr(), rcnt.ptr(), R, seg.ptr()});
    exclusive_scan_u64(seg.ptr(), segoff.ptr(), 2 * R + 2);
    inclusive_scan_u32(cov.ptr(), copies.ptr(), NE + 1);
    n_ent = read_scalar(segoff.ptr() + (2 * R + 1));
    if (knobs().debug_arena) {
        std::vector<u64> h = to_host(rcnt, R); u64 mx = 0, sum = 0, big = 0, hist[8] = {0};
        for (u64 v : h) { mx = std::max(mx, v); sum += v; if (v > 256) big++; int b = 0; while ((32ull << b) < v && b < 7) b++; hist[b]++; }
        fprintf(stderr, "path copy: %llu pieces, entries per piece: max %llu, mean %.1f, %llu above 256; hist(<=32,64,128,..): %llu %llu %llu %llu %llu %llu %llu %llu\n", (unsigned long long)R, (unsigned long long)mx, (double)sum / (double)R, (unsigned long long)big,
                (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)hist[6], (unsigned long long)hist[7]);
    }
    // (the scratch above stays where it is until the build ends: for a text this redundant it is a fraction of the text's size)
    ent_val.alloc(n_ent);
    int32_t* const out_ptr = ent_val.ptr();
    launch(NE, GapOutFunctor{ent.ptr(), ent_gap.ptr(), woff.ptr(), c.wfirst.ptr(), segoff.ptr(), copies.ptr(), out_ptr, depth_u.ptr()});
    launch_full(R * 32, RunOutFunctor<32, 1>{c.rr.ptr(), R, ra.ptr(), rcnt.ptr(), segoff.ptr(), ent.ptr(), ent_pos.ptr(), ent_end.ptr(), uinfo.ptr(),
                                             ent_want.ptr(), t, c.rseq.ptr(), minpos_fwd_u.ptr(), minpos_rev_u.ptr(), out_ptr, pos_cap_now});
    launch(loc.n_seqs, PathOffCopyFunctor{seq_tid.ptr(), seq_j.ptr(), woff.ptr(), c.wfirst.ptr(), c.w_gap.ptr(), segoff.ptr(), path_off.ptr()});
    tm->n_path_entries = n_ent;
    tm->path_runs_copied = R; tm->path_entries_walked = NE;
    copy_h2d(path_off.ptr() + loc.n_seqs, &n_ent, 8);
    launch(loc.n_seqs, PathEndsFunctor{ent_val.ptr(), path_off.ptr(), fs0.ptr(), fe0.ptr(), rank.ptr(), uorient.ptr()});
    walk_to_seed_order();
}

// K10 paths of this rank's sequences against the graph: count, scan, write; first / last unitig of every path.
template <int W> void GraphBuilder::Impl::walk() {
    TextCtx t = loc.ctx((int)k), g = G->ctx((int)k);
    Table tb = graph_table();
    Novel nv{bm.ptr(), wprefix.ptr()};
    UnitigCtx uc{head.ptr(), scan.ptr(), rank.ptr(), uorient.ptr(), ustart.ptr(), ulen.ptr(), U, N};
    const u32 PC = path_chunk(N, U);
    u64 n_walkers = (loc.n_text + PC - 1) / PC;
    // (sharded builds keep exact positions: a repeat of the build would have to be agreed between the ranks)
    pos_cap_now = (exact_positions || walk_answers || n_owners > 1 || G != &loc || pos_cap() == 0) ? 0xFFFFFFFFu : pos_cap();
    path_off.alloc((u64)loc.n_seqs + 1);
    const bool filter = path_filter();
    maybe_dest_valid = filter;
    fs0.alloc(U, true); fe0.alloc(U, true);
    walk_arrays();      // (everything the stage needs cleared is queued before its first launch: one fill batch)
    DBuf<u64> wcount_plain(n_walkers + 1);      // (the plain walk's counts; a copying walk sizes its own)
    wcount_plain.fill_bytes(0);
    if (filter) { maybe_dest.alloc((u64)U * 2); launch((u64)U * 2, MaybeDestFunctor{links.ptr(), maybe_dest.ptr()}); }
    if (walk_answers) {      // a sharded build planned (or not) before the walk-start keys went out (walk_queries)
        if (cplan.ok) { walk_copy_finish<W>(PC); lap(&tm->paths); return; }
    } else if (run_rows && n_owners <= 1 && G == &loc && PC <= 65535 && walk_copy_prepare<W>(PC, nv)) { walk_copy_finish<W>(PC); lap(&tm->paths); return; }
    walk_tables(filter);
    walked_plain = true;
    // everything from here to the compaction is the walk's own: 4 bytes of staging per text position (configs[4]: 20 GB) go back to
    // the arena once the entries are compacted — they are compacted into the staging area's own first bytes
    const Arena::Mark walk_mark = Arena::device().mark();
    DBuf<int32_t> stage(((n_walkers + 63) / 64) * 64 * PC);
    DBuf<u64>& wcount = wcount_plain; DBuf<u64> woff(n_walkers + 1);
    DBuf<u32> seq_tid(loc.n_seqs), seq_j(loc.n_seqs);
    DBuf<V16> uinfo(U);
    launch(U, WalkInfoFunctor{uc, filter ? maybe_dest.ptr() : nullptr, uinfo.ptr()});
    launch(n_walkers, PathWalkFunctor<W>{t, g, tb, nv, uc, uinfo.ptr(), wl_text.ptr(), PC, stage.ptr(), wcount.ptr(), seq_tid.ptr(), seq_j.ptr(),
                                        depth_u.ptr(), minpos_fwd_u.ptr(), minpos_rev_u.ptr(), counters.ptr() + 4, filter ? maybe_dest.ptr() : nullptr,
                                        pos_cap_now, path_diag(), walk_answers, n_walkers, nullptr, nullptr, nullptr, run_start.ptr(), run_end.ptr()});
    exclusive_scan_u64(wcount.ptr(), woff.ptr(), n_walkers + 1);
    n_ent = read_scalar(woff.ptr() + n_walkers);
    launch(loc.n_seqs, PathOffFunctor{seq_tid.ptr(), seq_j.ptr(), woff.ptr(), path_off.ptr()});
    tm->n_path_entries = n_ent;
    copy_h2d(path_off.ptr() + loc.n_seqs, &n_ent, 8);
    {
        DBuf<int32_t> packed(n_ent);      // (beyond the staging area: the compaction reads rows that later wavefronts' outputs would overwrite)
        launch_full(((n_walkers + 63) / 64) * 64, PathCompactFunctor{stage.ptr(), wcount.ptr(), woff.ptr(), PC, n_walkers, packed.ptr()});
        stage = DBuf<int32_t>(); woff = DBuf<u64>(); seq_tid = DBuf<u32>(); seq_j = DBuf<u32>(); uinfo = DBuf<V16>();
        Arena::device().rewind(walk_mark);
        ent_val.alloc(n_ent);             // where the staging area began; `packed` lies behind the staging area's end (n_ent <= its size)
        if (ent_val.ptr() != packed.ptr()) copy_d2d(ent_val.ptr(), packed.ptr(), n_ent * 4);
    }
    launch(loc.n_seqs, PathEndsFunctor{ent_val.ptr(), path_off.ptr(), fs0.ptr(), fe0.ptr(), rank.ptr(), uorient.ptr()});
    walk_to_seed_order();
    lap(&tm->paths);
}

// K12..K17 + D2H: sequences, link push order, expand_repeats, both renumberings, final numbering.  Needs depth,
// min positions and path ends of ALL sequences (reduced over ranks first in a sharded build).
template <int W> void GraphBuilder::Impl::tail(FinalGraph* out, bool want_graph, bool want_paths) {
    PackedText& g = *G;
    const u32 n_seqs = loc.n_seqs;
    SideStream& side = SideStream::get();
    HostBlock number_block;      // (declared before the guard: it goes after the side stream has drained, whatever ends this scope)
    SideStream::Guard side_guard;
    // (see PathRemapJob) the entries go now, in seed numbers, under everything that follows
    const bool host_remap = want_paths && host_remap_allowed && n_ent > 0 &&
                            (host_remap_mode() == 1 || (host_remap_mode() < 0 && n_ent >= (1u << 18) && U <= (8u << 20) && path_remap_is_wide()));
    PathRemapJob remap_job;
    struct RemapJoin { PathRemapJob& j; ~RemapJoin() { path_remap_finish(j); } } remap_join{remap_job};      // (the threads are done before the guard and the table go)
    HostBlock seq_words_block; SeqExpandJob seq_job; bool seq_as_codes = false;
    // Late copies (round 6): a result of the last stage is copied out behind the kernel that made it — through an event of stream 0 the side
    // stream waits for (after_main), or, AC_LATE_COPIES, ISSUED BY THIS THREAD once that event has fired: it has enqueued every kernel of the
    // build long before the device gets there and only waits from then on, so it can watch the events in order and hand the copy engine work
    // whose inputs are ready (on some boxes a copy queue that has to wait for a kernel wakes up late: DESIGN.md §6.2, the 0.800 s builds).
    struct LateCopy { void* ev; std::function<void()> issue; };
    std::vector<LateCopy> late;
    bool late_by_host = false;
    auto late_copy = [&](std::function<void()> issue) {
        if (late_by_host) late.push_back(LateCopy{side.main_event(), std::move(issue)});
        else { side.after_main(); issue(); }
    };
    auto issue_late_copies = [&]() { for (auto& c : late) { SideStream::wait_event(c.ev); c.issue(); } late.clear(); };
    struct SeqJoin { SeqExpandJob& j; ~SeqJoin() { seq_expand_finish(j); } } seq_join{seq_job};
    // Round 6: where that table would be too large for the host's caches (more than 8 M unitigs: a mixed-species job) the entries cross as
    // STRETCHES of consecutive text-order numbers (kernels_paths.inc) — 8 bytes per stretch instead of 4 per entry — and the host writes
    // the final numbers out from the table front to back: configs[4]'s 4.8 GB of entries are 84 ms of the 57 GB/s the link gives device ->
    // host (tools/microbench/d2h_probe.hip), behind 3 GB of other late results.  Taken when it at least halves the bytes (one more read-back, on a build of seconds).
    HostBlock rec_val_block, rec_pos_block;
    u64 n_stretch = 0;
    bool host_stretch = false;
    if (!host_remap && want_paths && host_remap_allowed && n_ent > 0 && n_ent < 0xFFFFFFF0ULL &&
        (host_remap_mode() == 2 || (host_remap_mode() < 0 && n_ent >= ((u64)1 << 24)))) {
        const bool always = host_remap_mode() == 2;
        const Arena::Mark all_mark = Arena::device().mark();
        const u64 rec_cap = always ? n_ent : n_ent / 4 + 1;      // (taken when it at least halves the bytes: at most n_ent / 4 records)
        DBuf<int32_t> rv(rec_cap); DBuf<u32> rp(rec_cap);         // (stay until the build ends: the copies below read them)
        const Arena::Mark scratch_mark = Arena::device().mark();
        const u64 n_groups = (n_ent + 63) / 64;
        DBuf<u32> sflag(n_groups + 1), sat(n_groups + 1);      // (stretch starts per group of 64 entries; [n_groups] = 0: the scan ends with the total)
        launch_full((n_groups + 1) * 64, StretchCountFunctor{ent_val.ptr(), n_ent, sflag.ptr()});
        exclusive_scan_u32(sflag.ptr(), sat.ptr(), n_groups + 1);
        n_stretch = read_scalar(sat.ptr() + n_groups);
        if (n_stretch <= rec_cap) {
            launch_full(n_groups * 64, StretchRecordFunctor{ent_val.ptr(), n_ent, sat.ptr(), rv.ptr(), rp.ptr()});
            rec_val_block = PinnedPool::get().alloc(n_stretch * 4); rec_pos_block = PinnedPool::get().alloc(n_stretch * 4);
            out->path_block = PinnedPool::get().alloc(n_ent * 4);      // (written by the host's threads, not by a copy)
            side.after_main();
            copy_d2h_async(rec_val_block.p, rv.ptr(), n_stretch * 4, side.stream());
            copy_d2h_async(rec_pos_block.p, rp.ptr(), n_stretch * 4, side.stream());
            host_stretch = true;
        }
        sflag = DBuf<u32>(); sat = DBuf<u32>();
        Arena::device().rewind(host_stretch ? scratch_mark : all_mark);      // (stream 0 reuses the scratch only behind the kernels that read it)
        if (!host_stretch) { rv = DBuf<int32_t>(); rp = DBuf<u32>(); }
    }
    const bool host_numbers = host_remap || host_stretch;      // the device only checks the paths' length sums; the host writes the final numbers
    // (stretch mode: the waves of RemapFunctor — remap_block() entries each — from this one on are the DEVICE's share of the renumbering; see K16 below)
    const u64 stretch_split_wave = [&]() -> u64 {
        const u64 n_waves = (n_ent + remap_block() - 1) / remap_block();
        const u32 pct = std::min<u32>(stretch_device_share(), 100u);
        return host_stretch ? n_waves - (u64)((double)n_waves * pct / 100.0) : n_waves;
    }();
    paths_in_seed_numbers = host_numbers;
    if (host_remap) {
        out->path_block = PinnedPool::get().alloc(n_ent * 4);
        side.after_main();
        copy_d2h_async(out->path_block.p, ent_val.ptr(), n_ent * 4, side.stream());
    }
    // K12 sequences
    u64 total = N;   // sum of unitig lengths == number of distinct canonical k-mers
    DBuf<u8> useq(total);
    // (both sequence writers: one thread per 64 output bytes; on the device through the block index + LDS tile of seq_write_kernel)
    auto write_seqs = [&](int mode, const ExpState* es, const u64* off, u64 n_bytes, u8* dst, bool total_on_device = false) {
        // the indexed / LDS-tiled writer pays for its index (four small launches) from ~16 MB of output on: config C (7.8 MB) 6.02 vs
        // 5.97 ms per build with it, E' (45 MB) 24.9 vs 26.6, config D (126 MB): see DESIGN.md §6
        if (seq_writer_plain() || (n_bytes < ((u64)16 << 20) && !seq_writer_forced())) {
            const u32 per = 16;      // output bytes per thread (r06n: 64 left most of the chip idle on 7.8 MB)
            if (mode == 0) launch((n_bytes + per - 1) / per, SeqFunctor{g.bits.ptr(), off, ustartpos.ptr(), ulen.ptr(), uorient.ptr(), U, n_bytes, (int)(k / 2), dst, per});
            else launch((n_bytes + per - 1) / per, MaterializeFunctor{*es, off, U, n_bytes, dst, per, total_on_device});
            return;
        }
        const u64 n_blocks = (n_bytes + 63) / 64;
        if (n_blocks == 0) return;
        DBuf<u32> bmax(n_blocks), first(n_blocks);
        bmax.fill_bytes(0);
        launch(U, BlockMaxFunctor{off, U, n_blocks, bmax.ptr()});
        inclusive_max_scan_u32(bmax.ptr(), first.ptr(), n_blocks);
        SeqSrc q{g.bits.ptr(), ustartpos.ptr(), ulen.ptr(), uorient.ptr(), (int)(k / 2)};
        ExpState e0{};
        if (mode == 0) launch_wave_kernel(seq_write_kernel<0>, (n_blocks + 255) / 256, 0, q, e0, off, (const u32*)first.ptr(), U, n_bytes, dst);
        else launch_wave_kernel(seq_write_kernel<1>, (n_blocks + 255) / 256, 0, q, *es, off, (const u32*)first.ptr(), U, n_bytes, dst);
    };
    // Round 6: every buffer of the tail that starts out cleared — and the scratch of its two renumbering sorts — is allocated HERE, before the
    // tail's first launch, so that their fills leave as one batch with it (a buffer cleared right before its first use cost a fill launch
    // each: fourteen of them between here and the last read-back of a config C build).
    const u64 J = (u64)U * 2;
    DBuf<u8> fixed_start(U, true), fixed_end(U, true), cand((u64)U * 2);
    DBuf<u32> renum_flag(1, true), cflag(J + 1), pool_used(EXP_SUBPOOLS + 1);
    cflag.fill_bytes(0);       // [J] = 0: the exclusive scan then ends with the total
    pool_used.fill_bytes(0);
    static const u32 SHIFT_CHECKS = 64;      // host checks of the pass loop whose "moved something" words are cleared up front (two words per check)
    DBuf<u64> shifted2(2 * SHIFT_CHECKS), lcount((u64)U + 1), sums(n_seqs), n_links_dev(256);
    shifted2.fill_bytes(0); lcount.fill_bytes(0); sums.fill_bytes(0); n_links_dev.fill_bytes(0);
    RadixScratch sort1, sort2;
    int len_bits = 32;      // no unitig is longer than the longest sequence of a text whose sequences this build knows
    if (G == &loc && !loc.h_len.empty()) { u32 mx = 0; for (u32 l : loc.h_len) mx = std::max(mx, l); len_bits = 1; while (len_bits < 32 && (mx >> len_bits)) len_bits++; }
    if (!renum_two_pass()) { sort1.prepare(U, 32 + len_bits); sort2.prepare(U, 32 + len_bits); }
    write_seqs(0, nullptr, useq_off.ptr(), total, useq.ptr());
    lap(&tm->seqs);

    // K13 link push order, K14 static analysis for expand_repeats, K15 first renumber_unitigs
    DBuf<int32_t> lord((u64)U * 10); DBuf<u8> lcnt((u64)U * 2);
    launch_full((u64)U * 2, LinkOrderFunctor{links.ptr(), lord.ptr(), lcnt.ptr(), counters.ptr() + 5, n_links_dev.ptr()});
    OrderedLinks L{lord.ptr(), lcnt.ptr()};
    launch(U, FixedSpreadFunctor{fs0.ptr(), fe0.ptr(), L, fixed_start.ptr(), fixed_end.ptr()});
    launch((u64)U * 2, CandFunctor{L, fixed_start.ptr(), fixed_end.ptr(), cand.ptr()});
    if (maybe_dest_valid)      // the walk only collected smallest positions where maybe_dest says so: every real candidate must be covered
        launch((u64)U * 2, CandCoveredFunctor{cand.ptr(), maybe_dest.ptr(), counters.ptr() + 4});
    DBuf<u32> order1(U);
    launch(U, IotaFunctor{order1.ptr()});
    const bool defer_sorts = deferred_sort_checks();
    renumber_sort(order1, U, ulen.ptr(), useq_off.ptr(), useq.ptr(), depth.ptr(), defer_sorts ? sort_flags.ptr() + 1 : renum_flag.ptr(), defer_sorts, &sort1, len_bits, /*order_is_identity=*/true);
    lap(&tm->analysis);

    // K17 expand_repeats, level-scheduled (see the kernels)
    DBuf<u64> coff(U), len64((u64)U + 1), noff((u64)U + 1);
    DBuf<u32> clen(U); DBuf<ExpU> ev(U);      // the views of expand_repeats (one 32-byte record per unitig); coff / clen: offsets and lengths as plain arrays for what follows
    DBuf<u8> seq_alt(total), pool(std::min<u64>(8 * total + (1u << 20), 0xFFFFFFF0ULL)), dirty((u64)U * 2);
    DBuf<u64> shifted(1);
    launch(U, ExpInitFunctor{useq_off.ptr(), ulen.ptr(), cand.ptr(), ev.ptr(), coff.ptr(), clen.ptr(), dirty.ptr()});      // core views = the unitigs, dirty = the candidates (three copies, one launch)
    u8* cur = useq.ptr(); u8* alt = seq_alt.ptr();
    u64 final_total = total;
    bool final_total_pending = false;      // the last rewrite's sum is read with the build's last batch (small outputs: MaterializeFunctor takes it from the device)
    u64 n_links = 0;
    int passes = 0;
    u32 n_cand = 0, n_levels = 0, sparse_sweeps = 0, sparse_start = 0;
    const bool partitioned = n_owners > 1 && (bool)tail_xchg;      // (decided by the driver: the same on every rank)
    DBuf<u8> jowner; DBuf<u32> owned_count, gpre, gpost;
    u32 n_cand_owned = 0;
    {
        DBuf<u32> cpos(J + 1), prio(J);
        launch(J, CandFlagFunctor{order1.ptr(), cand.ptr(), cflag.ptr()});
        exclusive_scan_u32(cflag.ptr(), cpos.ptr(), J + 1);
        {      // (... and the number of links, for the buffers of K16)
            u64 part[256];
            ReadBatch rb; rb.add(&n_cand, cpos.ptr() + J, 4); rb.add(part, n_links_dev.ptr(), sizeof part); rb.run();
            n_links = 0;
            for (u64 v : part) n_links += v;
        }
        if (n_cand == 0) {
            passes = 1;   // the reference's single pass that moves nothing (the same on every rank of a sharded build: nothing to merge)
        } else {
            u64 C = n_cand;
            DBuf<u32> clist(C), level(C);
            prio.fill_bytes(0xFF);
            DBuf<u32> changed(16), preds(C * MAX_PREDS); DBuf<u8> npred(C);      // changed[9]: the sparse tail's list length, [10..12]: what it reports (MopState::out)
            changed.fill_bytes(0);      // (the first round of sweeps: cleared with this batch)
            RadixScratch sort_lv;
            sort_lv.prepare(C, 32);
            launch(J, CandListFunctor{order1.ptr(), cflag.ptr(), cpos.ptr(), clist.ptr(), prio.ptr()});
            launch(C, FillU32Functor{level.ptr(), 1u});
            DBuf<V16> touch(U);      // the candidate junctions touching each unitig: for the conflict lists here and for every junction that moves something
            launch(U, TouchFunctor{L, cand.ptr(), touch.ptr()});
            launch(C, LevelPredsFunctor{L, cand.ptr(), clist.ptr(), prio.ptr(), C, preds.ptr(), npred.ptr(), touch.ptr()});
            if (partitioned) {      // this rank's share of the junctions: the conflict components it owns
                DBuf<u32> parent(C);
                jowner.alloc(C); owned_count.alloc(1); owned_count.fill_bytes(0);
                launch(C, UfInitFunctor{parent.ptr()});
                launch(C, UfUnionFunctor{preds.ptr(), npred.ptr(), C, parent.ptr()});
                launch(C, UfOwnerFunctor{parent.ptr(), n_owners, jowner.ptr()});
                launch_full((J + 63) & ~63ULL, OwnedDirtyFunctor{cand.ptr(), prio.ptr(), jowner.ptr(), my_owner, dirty.ptr(), owned_count.ptr(), J});
                gpre.alloc(U, true); gpost.alloc(U, true);
            }
            u32 max_level = 1;
            for (int round = 0;; round++) {   // longest-path levels of the conflict DAG, settled front to back (LevelRelaxFunctor); eight sweeps per host
                if (round) changed.fill_bytes(0);       // check, done when the last of them left no candidate open (a sweep settles one more level)
                for (int it = 0; it < 8; it++)
                    launch(C, LevelRelaxFunctor{preds.ptr(), npred.ptr(), C, level.ptr(), changed.ptr() + it, it ? changed.ptr() + it - 1 : nullptr, changed.ptr() + 8});
                const std::vector<u32> hc = to_host(changed, 9);
                max_level = std::max(max_level, hc[8]);
                if (hc[7] == 0) break;
            }
            DBuf<u64> lkey(C);
            launch(C, LevelKeyFunctor{level.ptr(), lkey.ptr()});
            int level_bits = 1;
            while (level_bits < 32 && (max_level >> level_bits)) level_bits++;
            sort_pairs_u64_u32(lkey, clist, C, level_bits, 0, 0, &sort_lv);      // (the highest level came back with the convergence flags: one or two digits)
            // first index of every level; [0] = number of levels (levels beyond the table: a second, exact read)
            const u32 LV_TABLE = expand_level_table();
            DBuf<u32> bstart((u64)LV_TABLE + 2);
            launch(C, LevelBoundsFunctor{lkey.ptr(), C, bstart.ptr(), LV_TABLE});
            std::vector<u32> hb = to_host(bstart, (u64)LV_TABLE + 2);
            n_levels = hb[0];
            if (n_levels > LV_TABLE) {
                DBuf<u32> big((u64)n_levels + 2);
                launch(C, LevelBoundsFunctor{lkey.ptr(), C, big.ptr(), n_levels});
                hb = to_host(big, (u64)n_levels + 2);
            }
            hb.resize((size_t)n_levels + 2);
            hb[n_levels + 1] = (u32)C;
            ExpState e{cur, ev.ptr(), pool.ptr(), pool_used.ptr(), minpos_fwd.ptr(), minpos_rev.ptr(), dirty.ptr(), cand.ptr(), L, shifted.ptr(), touch.ptr()};
            u64 moved = 0, moved_since_rewrite = 0;
            // Rewrites the sequences contiguously (gained pieces folded into the core views) and empties the pool.  Once after the
            // last pass — and in between whenever the pool is a quarter full: a side that gains again gets a new piece holding its
            // old one as well, so without this the pool use of a many-pass input grows with the square of the passes.
            auto rewrite = [&](bool last) {
                if (partitioned) launch(U, ExpFoldFunctor{e, gpre.ptr(), gpost.ptr()});      // (what the fold makes of the gained pieces: the merge below)
                launch((u64)U + 1, ExpLenFunctor{e, len64.ptr(), U});
                exclusive_scan_u64(len64.ptr(), noff.ptr(), (u64)U + 1);
                // (expand_repeats only ever shortens the total — n >= 2 sources lose what ONE destination gains — so the last total is a bound for
                // this one: the last rewrite of a small output launches over the bound and lets the kernel read the sum, one round trip less)
                const bool defer_total = last && !partitioned && (seq_writer_plain() || (final_total < ((u64)16 << 20) && !seq_writer_forced()));
                if (defer_total) { write_seqs(1, &e, noff.ptr(), final_total, alt, /*total_on_device=*/true); final_total_pending = true; }
                else { final_total = read_scalar(noff.ptr() + U); write_seqs(1, &e, noff.ptr(), final_total, alt); }
                launch(U, ExpResetFunctor{e, noff.ptr(), coff.ptr(), clen.ptr()});
                std::swap(cur, alt);
                e.cur = cur;
                if (!last) pool_used.fill_bytes(0);      // (nothing allocates from the pool after the last rewrite)
                moved_since_rewrite = 0;
            };
            const u32 sparse_max = expand_sparse_max();
            const u32 sub_limit = (u32)(pool.size() / 2 / EXP_SUBPOOLS / 2);      // a region half full (or anything in the overflow half) asks for a rewrite
            auto run_level = [&](u32 lv) {
                const u64 cnt = (u64)(hb[lv + 1] - hb[lv]);
                // sixteen lanes per junction, four junctions per wavefront (expand_wave_kernel; the emulation runs the same kernel in
                // lockstep, wave_rt.hpp).  A thread per junction and 8 / 32 / 64 lanes were measured and retired (r06u/v: G = 16
                // wins from config C to mixed-species graphs)
                if (cnt) launch_wave_kernel(expand_wave_kernel<W, 16>, (cnt * 16 + 255) / 256, 0, e, (const u32*)clist.ptr(), (u64)hb[lv], cnt, (u32)pool.size(), counters.ptr() + 7);
            };
            for (u32 check = 0;; check++) {   // two passes per host check: if the first moved nothing the second is an (uncounted) no-op
                u64* const sh_words = shifted2.ptr() + 2 * (u64)(check % SHIFT_CHECKS);
                if (check && check % SHIFT_CHECKS == 0) shifted2.fill_bytes(0);      // (the words cleared up front are used up)
                for (int half = 0; half < 2; half++) {
                    e.shifted = sh_words + half;
                    for (u32 lv = 1; lv <= n_levels; lv++) run_level(lv);
                }
                if (sparse_max) {      // the dirty junctions listed, and — if they are few and the second pass moved something — all remaining passes by one workgroup (kernels_tail.inc)
                    launch_full((C + 63) & ~63ULL, MopCompactFunctor{clist.ptr(), lkey.ptr(), dirty.ptr(), (u64*)preds.ptr(), changed.ptr() + 9, C});
                    const MopState ms{(u64*)preds.ptr(), changed.ptr() + 9, prio.ptr(), level.ptr(), sh_words, changed.ptr() + 10, sparse_max, expand_sparse_list() ? expand_sparse_list() : 8 * sparse_max, sub_limit, std::min(MOP_BATCH, expand_sparse_batch())};
                    launch_wave_kernel_sized(expand_mopup_kernel<W>, 1, MOP_THREADS, 0, e, ms, (u32)pool.size(), counters.ptr() + 7);
                }
                u64 sh[2]; u32 used = 0; u32 mop[3] = {0, 0, 0};
                {
                    std::vector<u32> pu(EXP_SUBPOOLS + 1);
                    ReadBatch rb;
                    rb.add(sh, sh_words, 16);
                    rb.add(pu.data(), pool_used.ptr(), (EXP_SUBPOOLS + 1) * 4);
                    if (sparse_max) rb.add(mop, changed.ptr() + 10, 12);
                    rb.run();
                    for (u32 q = 0; q < EXP_SUBPOOLS; q++) used = std::max(used, pu[q]);
                    if (pu[EXP_SUBPOOLS]) used = 0xFFFFFFFFu;
                }
                moved += sh[0] + sh[1]; moved_since_rewrite += sh[0] + sh[1];
                if (sh[0] == 0) { passes += 1; break; }
                passes += 2;
                if (sh[1] == 0) break;
                if (mop[0]) {      // the one-workgroup tail ran: its sweeps are passes of the reference (the last one of a finished tail moved nothing)
                    passes += (int)mop[1];
                    sparse_sweeps += mop[1]; sparse_start = mop[2];
                    if (mop[0] == 1) break;
                }
                if (used > sub_limit || expand_rewrite_always()) rewrite(false);
            }
            if (!partitioned) { if (moved_since_rewrite) rewrite(true); }
            else {
                // every rank ran its own junctions: merge what they did to the unitigs, field by field (kernels_tail.inc), and agree on
                // the number of passes (the reference's count is that of the component that needed most)
                DBuf<u8> fown((u64)U * 3); DBuf<int32_t> lens3((u64)U * 3 + 1);
                launch(U, FieldOwnerFunctor{L, cand.ptr(), prio.ptr(), jowner.ptr(), n_owners, fown.ptr()});
                launch(U, OwnedLensFunctor{e, fown.ptr(), gpre.ptr(), gpost.ptr(), my_owner, lens3.ptr()});
                const int32_t neg_passes = -(int32_t)passes;
                copy_h2d(lens3.ptr() + (u64)U * 3, &neg_passes, 4);
                stream_sync();
                tail_xchg(lens3.ptr(), (u64)U * 3, 1, 0);
                tail_xchg(lens3.ptr() + (u64)U * 3, 1, 1, 1);      // MIN of the negated counts
                launch((u64)U + 1, Lens3SumFunctor{lens3.ptr(), len64.ptr(), U});
                exclusive_scan_u64(len64.ptr(), noff.ptr(), (u64)U + 1);
                int32_t min_neg = 0;
                {
                    ReadBatch rb;
                    rb.add(&final_total, noff.ptr() + U, 8);
                    rb.add(&min_neg, lens3.ptr() + (u64)U * 3, 4);
                    rb.add(&n_cand_owned, owned_count.ptr(), 4);
                    rb.run();
                }
                passes = -min_neg;
                if (final_total > seq_alt.size()) throw DeviceError("internal error: merged sequences longer than before expand_repeats");
                const u32 per = 16;
                launch((final_total + per - 1) / per, MergeSeqFunctor{e, fown.ptr(), gpre.ptr(), gpost.ptr(), my_owner, lens3.ptr(), noff.ptr(), U, final_total, alt, per});
                stream_sync();
                tail_xchg(alt, final_total, 0, 0);
                launch(U, ExpResetFunctor{e, noff.ptr(), coff.ptr(), clen.ptr()});
                std::swap(cur, alt);
                e.cur = cur;
            }
            (void)moved;
        }
    }
    tm->simplify_passes = (u32)passes; tm->n_candidates = n_cand; tm->n_levels = n_levels;
    tm->expand_sparse_sweeps = sparse_sweeps; tm->expand_sparse_start = sparse_start;
    tm->n_candidates_owned = partitioned && n_cand ? n_cand_owned : n_cand;
    lap(&tm->expand);

    // K15b second renumber_unitigs (graph_simplification.rs:39): a stable sort of the CURRENT order on the new
    // sequences; K16 per-unitig outputs in final order, links in get_links_for_gfa order, paths in final numbers
    // D2H on a second stream, each array as soon as it is final, straight into pinned blocks owned by the result; the
    // paths go in four chunks, each copied while the next is still being renumbered.
    DBuf<u64> seq_words; DBuf<u32> seq_bad;
    if (want_graph) {
        out->seq_block = PinnedPool::get().alloc(final_total + 64);
        // As 2-bit codes, written out by host threads (SeqExpandJob): a quarter of the bytes over the link — where the sequences are most of what
        // is final this late (one species of long genomes: config D 104 of 142 MB, build 24.9 -> 24.3 ms).  Elsewhere the bytes the host's threads
        // then write slow the device's copies into the same memory down by as much as the link saves (mini-E 55.2 = 55.2 ms, config C 3.70 -> 3.76:
        // profiles/r15o_*), so the codes are taken only when the sequences outweigh the other late results two to one — and are at least 32 MB: on
        // D' at k = 201 (10 MB of sequences) the job's fixed costs were 0.1 ms more than the link saved.
        const u64 other_late = (u64)U * 28 + n_links * sizeof(Link);
        if (seq_codes_transfer() == 2 || (seq_codes_transfer() == 1 && final_total >= ((u64)32 << 20) && final_total > 2 * other_late)) {
            const u64 nw = (final_total + 31) / 32;
            seq_words.alloc(nw); seq_bad.alloc(1); seq_bad.fill_bytes(0);
            launch(nw, SeqPack2Functor{cur, final_total, seq_words.ptr(), seq_bad.ptr()});
            seq_words_block = PinnedPool::get().alloc(nw * 8);
            side.after_main();
            copy_d2h_async(seq_words_block.p, seq_words.ptr(), nw * 8, side.stream());
            seq_job.words = (const u64*)seq_words_block.p; seq_job.out = (u8*)out->seq_block.p; seq_job.total = final_total;
            seq_job.landed = side.mark();
#ifndef AC_EMU
            AC_HIP_CHECK(hipGetDevice(&seq_job.dev));
            seq_expand_start(seq_job, 12);
#endif
            seq_as_codes = true;
        } else {
            side.after_main();     // sequences are final since the materialise step: their copy runs under the second renumbering
            copy_d2h_async(out->seq_block.p, cur, final_total, side.stream());
        }
    }
    DBuf<u32> order2(U);
    copy_d2d(order2.ptr(), order1.ptr(), (size_t)U * 4);
    renumber_sort(order2, U, clen.ptr(), coff.ptr(), cur, depth.ptr(), defer_sorts ? sort_flags.ptr() + 1 : renum_flag.ptr(), defer_sorts, &sort2, len_bits);
    DBuf<u64> number_len(U), number_len_text(U), loff((u64)U + 1);      // (_text: by text-order index, what the path entries are in)
    DBuf<u32> number_only(host_numbers ? U : 0);
    DBuf<u8> meta((size_t)U * 24);
    u64* d_seq_begin = (u64*)meta.ptr();
    double* d_depth = (double*)(meta.ptr() + (size_t)U * 8);
    u32* d_seq_len = (u32*)(meta.ptr() + (size_t)U * 16);
    u32* d_seed_index = (u32*)(meta.ptr() + (size_t)U * 20);
    out->k = k;
    out->n_kmers = 2 * (u64)N;
    out->n_unitigs = U;
    late_by_host = late_copies() == 2 || (late_copies() == 1 && final_total + (u64)U * 28 + n_links * sizeof(Link) >= ((u64)256 << 20));
    launch(U, FinalMetaFunctor{order2.ptr(), coff.ptr(), clen.ptr(), depth.ptr(), lcnt.ptr(), number_len.ptr(), d_seq_begin, d_depth,
                               d_seq_len, lcount.ptr(), host_numbers ? number_only.ptr() : nullptr, d_seed_index, order.ptr(), number_len_text.ptr(), uorient.ptr()});
    if (host_numbers) {      // the number table first: the host threads start on the entries while the rest is still crossing
        number_block = PinnedPool::get().alloc((size_t)U * 4);
        remap_job.path = (int32_t*)out->path_block.p; remap_job.n_ent = n_ent;
        if (host_stretch) { remap_job.rec_val = (const int32_t*)rec_val_block.p; remap_job.rec_pos = (const u32*)rec_pos_block.p; remap_job.n_rec = n_stretch; }
        remap_job.number = (const u32*)number_block.p; remap_job.n_unitigs = U;
        remap_job.ent_limit = host_stretch ? (u64)stretch_split_wave * remap_block() : ~0ULL;
        const u32* d_number_only = number_only.ptr();
        const int remap_threads = (int)(host_stretch ? 2 * upload_threads() : upload_threads());      // (writing the stretches out is bound by the host's memory, not by its cores' arithmetic: twice the packing threads — configs[4] writes 4.8 GB)
        late_copy([&, d_number_only, remap_threads]() {
            copy_d2h_async(number_block.p, d_number_only, (size_t)U * 4, side.stream());
            remap_job.landed = side.mark();
#ifndef AC_EMU
            AC_HIP_CHECK(hipGetDevice(&remap_job.dev));
            path_remap_start(remap_job, remap_threads);
#endif
        });
    }
    if (want_graph) {
        out->meta_block = PinnedPool::get().alloc((size_t)U * 24);
        const u8* d_meta = meta.ptr();
        late_copy([&, d_meta]() { copy_d2h_async(out->meta_block.p, d_meta, (size_t)U * 24, side.stream()); });
    }
    exclusive_scan_u64(lcount.ptr(), loff.ptr(), (u64)U + 1);
    DBuf<Link> links_out(n_links);      // (n_links: counted by LinkOrderFunctor, read with the candidate count)
    launch(U, LinkOutFunctor{order2.ptr(), L, number_len.ptr(), loff.ptr(), links_out.ptr()});
    if (want_graph) {
        out->links_block = PinnedPool::get().alloc(n_links * sizeof(Link));
        const Link* d_links_out = links_out.ptr();
        late_copy([&, d_links_out]() { copy_d2h_async(out->links_block.p, d_links_out, n_links * sizeof(Link), side.stream()); });
    }
    if (host_numbers) {      // the device only checks that every path spells its sequence's length (the sums), it stores nothing ...
        const u64 RB = remap_block();
        const u64 n_waves = (n_ent + RB - 1) / RB;
        const u64 w_split = host_stretch ? std::min<u64>(stretch_split_wave, n_waves) : n_waves;
        if (w_split) launch_full(w_split * 64, RemapFunctor{ent_val.ptr(), number_len_text.ptr(), path_off.ptr(), n_seqs, n_ent, sums.ptr(), 0, (u32)RB, nullptr, false});
        // ... except for the LAST share of a build that sends its paths as stretches (round 6).  The host's threads can only begin when the
        // number table exists and write configs[4]'s 4.8 GB at ~37 GB/s (a table row fetched per ~10-entry stretch: the host's memory latency,
        // tools/microbench/host_write_probe.hip), which ends ~65 ms AFTER the link has delivered everything else; so the entries behind
        // PathRemapJob::ent_limit are renumbered here and follow the other results over the link, and both ends finish together.
        if (w_split < n_waves) {
            const u64 n_chunks = 4, per_chunk = std::max<u64>((n_waves - w_split + n_chunks - 1) / n_chunks, 64);
            for (u64 w = w_split; w < n_waves; w += per_chunk) {
                const u64 cnt = std::min<u64>(per_chunk, n_waves - w);
                launch_full(cnt * 64, RemapFunctor{ent_val.ptr(), number_len_text.ptr(), path_off.ptr(), n_seqs, n_ent, sums.ptr(), w, (u32)RB, nullptr, true});
                const u64 b = w * RB, e2 = std::min<u64>((w + cnt) * RB, n_ent);
                const int32_t* d_ent = ent_val.ptr();
                late_copy([&, d_ent, b, e2]() { copy_d2h_async((int32_t*)out->path_block.p + b, d_ent + b, (e2 - b) * 4, side.stream()); });      // (the one copy stream: behind the unitig records and the links)
            }
        }
    } else {
        if (want_paths) out->path_block = PinnedPool::get().alloc(n_ent * 4);
        const u64 RB = remap_block();
        const u64 n_waves = (n_ent + RB - 1) / RB;
        // Four chunks, each copied while the next is renumbered (the kernel storing straight into the pinned block measured equal, r08j:
        // either way the 4 bytes per entry cross PCIe after the final numbering exists — 42 MB = 0.7 ms on config C)
        // (round 6: from 256 MB of entries on, eight chunks alternating between two copy streams — configs[4] moves 4.8 GB here, and one copy
        // queue alone ran at 33 GB/s)
        const u64 n_chunks = n_ent * 4 >= ((u64)256 << 20) ? 8 : 4;
        const u64 per_chunk = std::max<u64>((n_waves + n_chunks - 1) / n_chunks, 64);
        int turn = 0;
        for (u64 w = 0; w < n_waves; w += per_chunk) {
            u64 cnt = std::min<u64>(per_chunk, n_waves - w);
            launch_full(cnt * 64, RemapFunctor{ent_val.ptr(), number_len_text.ptr(), path_off.ptr(), n_seqs, n_ent, sums.ptr(), w, (u32)RB, nullptr, true});
            if (want_paths) {
                u64 b = w * RB, e2 = std::min<u64>((w + cnt) * RB, n_ent);
                const int which = n_chunks == 8 ? (turn++ & 1) : 0;
                side.after_main(which);
                copy_d2h_async((int32_t*)out->path_block.p + b, ent_val.ptr() + b, (e2 - b) * 4, side.stream(which));
            }
        }
    }
    issue_late_copies();      // (every kernel of the build is enqueued: from here on this thread only waits)
    lap(&tm->finalize);

    std::vector<u64> h_sums(n_seqs);
    out->path_off.resize((size_t)n_seqs + 1);
    std::vector<u32> errs(8);
    u32 pack_bad[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    u32 h_sort_flags[2] = {0, 0};
    u64 n_links_check = 0, final_total_read = 0; u32 h_seq_bad = 0;
    {
        ReadBatch rb;
        rb.add(h_sums.data(), sums.ptr(), (size_t)n_seqs * 8);
        rb.add(out->path_off.data(), path_off.ptr(), ((size_t)n_seqs + 1) * 8);
        rb.add(errs.data(), counters.ptr(), 8 * 4);
        rb.add(h_sort_flags, sort_flags.ptr(), 8);
        rb.add(&n_links_check, loff.ptr() + U, 8);
        if (final_total_pending) rb.add(&final_total_read, noff.ptr() + U, 8);
        if (loc.check_alphabet && loc.pack_bad.size()) rb.add(pack_bad, loc.pack_bad.ptr(), 8);
        if (seq_as_codes) rb.add(&h_seq_bad, seq_bad.ptr(), 4);
        rb.run();                                   // synchronises stream 0 (once)
    }
    const double t_main_done = now_s();
    side.sync();                                    // ... and the copies: everything above has landed
    const double t_copies_done = now_s();
    if (seq_as_codes) {
#ifdef AC_EMU
        for (u64 b = 0; b < final_total; b += 4096) seq_expand_range(seq_job.words, seq_job.out, b, std::min<u64>(b + 4096, final_total));
#endif
        seq_expand_finish(seq_job);
        if (seq_job.ready.load() == 3) throw DeviceError("internal error: the sequence codes did not reach the host");
        if (h_seq_bad) {      // (never, short of a bug: a byte that is no base in a trimmed unitig sequence — the bytes themselves, then)
            copy_d2h_async(out->seq_block.p, cur, final_total, side.stream());
            side.sync();
        }
    }
    if (host_numbers) {
#ifdef AC_EMU
        if (host_stretch) { for (u64 b = 0; b < n_stretch; b += 7) path_stretch_range(remap_job, b, std::min<u64>(b + 7, n_stretch), &remap_job.bad); }      // (small blocks: the partial lines where two threads' blocks meet)
        else path_remap_range(remap_job.path, n_ent, remap_job.number, U, &remap_job.bad);
#endif
        path_remap_finish(remap_job);
    }
    tm->path_stretches = host_stretch ? n_stretch : 0;
    if (knobs().debug_arena && seq_as_codes)
        fprintf(stderr, "d2h: sequence codes: job started %.3f ms before stream 0 drained, its codes had landed %.3f ms after that start, %.1f MB written out in %.3f ms more\n",
                (t_main_done - seq_job.t_start.load()) * 1e3, (seq_job.t_ready.load() - seq_job.t_start.load()) * 1e3, final_total / 1e6, (seq_job.t_last.load() - seq_job.t_ready.load()) * 1e3);
    if (knobs().debug_arena && host_numbers)
        fprintf(stderr, "d2h: the host's renumbering threads started %.3f ms %s stream 0 drained (their table had landed) and were done %.3f ms later\n",
                std::fabs(remap_job.t_ready.load() - t_main_done) * 1e3, remap_job.t_ready.load() < t_main_done ? "before" : "after", (remap_job.t_last.load() - remap_job.t_ready.load()) * 1e3);
    if (knobs().debug_arena)      // (where the d2h stage goes: stream 0 drained -> the copies landed -> the host's renumbering threads done)
        fprintf(stderr, "d2h: copies landed %.3f ms after stream 0 drained, host renumbering done %.3f ms later; late results %.1f MB (sequences %.1f, unitig records %.1f, links %.1f, number table %.1f), entries %.1f MB\n",
                (t_copies_done - t_main_done) * 1e3, (now_s() - t_copies_done) * 1e3, (final_total + (double)U * 24 + n_links * sizeof(Link) + (host_numbers ? (double)U * 4 : 0)) / 1e6,
                final_total / 1e6, (double)U * 24 / 1e6, n_links * sizeof(Link) / 1e6, host_numbers ? (double)U * 4 / 1e6 : 0.0, n_ent * 4 / 1e6);
    if (loc.pack_bad.size()) loc.verify_alphabet(pack_bad);      // before any internal check: a text with foreign bytes explains them all
    if (errs[7] & 128u) throw NeedExactPositions();      // (before anything else: a repeat of the build settles it)
    if (h_sort_flags[0] || h_sort_flags[1]) {
        if (!deferred_sort_checks()) throw DeviceError("internal error: a sort flag was left set by a checked sort");
        throw NeedCheckedSorts();      // (the order the flagged sort left is a permutation, not THE order: everything behind it is void)
    }
    if (errs[7]) throw DeviceError("internal error: expand_repeats pool overflow");
    if (n_links_check != n_links) throw DeviceError("internal error: link counts disagree");
    if (final_total_pending) { if (final_total_read > final_total) throw DeviceError("internal error: expand_repeats lengthened the sequences"); final_total = final_total_read; }
    if (errs[3] || errs[4])
        throw DeviceError("internal error: inconsistent unitig ends (codes " + std::to_string(errs[3]) + "/" + std::to_string(errs[4]) + ")");
    if (remap_job.bad.load()) throw DeviceError("internal error: path entries without a unitig");
    if (want_graph) {
        out->seq_begin = (const u64*)out->meta_block.p;
        out->depth = (const double*)((const u8*)out->meta_block.p + (size_t)U * 8);
        out->seq_len = (const u32*)((const u8*)out->meta_block.p + (size_t)U * 16);
        out->seed_index = (const u32*)((const u8*)out->meta_block.p + (size_t)U * 20);
        out->links = (const Link*)out->links_block.p;
    }
    if (want_paths) out->path = (const int32_t*)out->path_block.p;
    out->n_links = n_links;
    out->n_path = n_ent;
    u64 n_self = errs[5];
    u64 links_one_way = (n_links + n_self) / 2;   // link_count().1 (unitig_graph.rs:478-507): a link and its mirror count
                                                   // once; a link that is its own mirror (a+ -> a-, a- -> a+) counts once
    out->pre = GraphStats{U, links_one_way, total};
    out->post = GraphStats{U, links_one_way, final_total};
    out->simplify_passes = passes;
    // The path of every sequence must spell its full length (unitig_graph.rs:160-174, decompress.rs).
    for (u32 s = 0; s < n_seqs; s++)
        if (h_sums[s] != (u64)loc.h_len[s])
            throw DeviceError("internal error: path length mismatch for sequence " + std::to_string(s + 1));
    lap(&tm->d2h);
    tm->total_device = now_s() - t_begin;
    tm->launches = rt_counters().launches; tm->readbacks = rt_counters().readbacks;
    if (knobs().debug_arena)
        fprintf(stderr, "arena: used %.1f MB (peak %.1f) of %.1f MB (n_text %.1f MB), %.3f s in hipMalloc / hipFree so far\n", Arena::device().total_used() / 1e6,
                Arena::device().peak() / 1e6, Arena::device().capacity() / 1e6, loc.n_text / 1e6, Arena::device().alloc_seconds());
}

template <int W> void Stages<W>::table(GraphBuilder::Impl& m) { m.template table<W>(); }
template <int W> void Stages<W>::degrees(GraphBuilder::Impl& m) { m.template degrees<W>(); }
template <int W> void Stages<W>::walk_queries(GraphBuilder::Impl& m) { m.template walk_queries<W>(); }
template <int W> void Stages<W>::answer_queries(GraphBuilder::Impl& m, const u64* d_keys, u64 n, u64* d_out) { m.template answer_queries<W>(d_keys, n, d_out); }
template <int W> void Stages<W>::route_queries(GraphBuilder::Impl& m, u32 n_shards, u64* d_routed_keys, u64* counts_host) { m.template route_queries<W>(n_shards, d_routed_keys, counts_host); }
template <int W> void Stages<W>::unitigs(GraphBuilder::Impl& m) { m.template unitigs<W>(); }
template <int W> void Stages<W>::walk(GraphBuilder::Impl& m) { m.template walk<W>(); }
template <int W> void Stages<W>::tail(GraphBuilder::Impl& m, FinalGraph* out, bool want_graph, bool want_paths) { m.template tail<W>(out, want_graph, want_paths); }
template <int W> void Stages<W>::fragments(GraphBuilder::Impl& m) { m.template fragments<W>(); }
template <int W> void Stages<W>::warm() {
#ifndef AC_EMU
    TextCtx t{}; Table tb{};
    hipLaunchKernelGGL((insert_wave_kernel<W, false>), dim3(1), dim3(256), 0, 0, t, tb, (u64)0, (u64)0, 256u, (InsertStats*)nullptr, (u32*)nullptr, (u64*)nullptr);      // (no chunk at all: every wavefront returns at once)
    (void)hipGetLastError();
#endif
}
#if AC_W_ONLY != 0
template struct Stages<AC_W_ONLY>;
#else
template struct Stages<1>; template struct Stages<2>; template struct Stages<3>; template struct Stages<4>; template struct Stages<8>; template struct Stages<16>;
#endif

}  // namespace ac
