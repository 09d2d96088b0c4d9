// Device graph build (see graph_build.hpp).  gfx950 HIP; the same file compiles as the CPU emulation
// under -DAC_EMU for the CPU test-suite.
//
// Order-free formulation (derived from unitig_graph.rs:176-226):
//   * one canonical key per strand pair, stored in an open-addressing table whose slot holds the text
//     position of the SMALLEST occurrence ("novel" position) of that k-mer;
//   * out(X) = number of set members sharing X's (k-1)-suffix as prefix (5 probes, kmer_graph.rs:136-150),
//     in(X) = out(rc X);
//   * step X->Y between consecutive text k-mers is unitig-internal iff
//        !first(rc X) && out(X)==1 && in(Y)==1 && !first(Y)            (unitig_graph.rs:192-223)
//     (the `seen` test only ever fires for Y == rc X, which cannot be two distinct novel positions);
//   * every unitig lies contiguously inside the run of novel positions of the first sequence that
//     contains it, so unitigs = segments of the sorted novel-position list cut at non-internal steps;
//   * unitig forward strand = strand holding its smallest k-mer; seed number = rank of that k-mer.
#ifdef AC_EMU
#define AC_EMU_DEFINE_CTX_SWITCH      // (the lockstep emulation's context switch is defined by this translation unit: wave_rt.hpp)
#endif
#include "graph_build.hpp"

#include <chrono>
#include <cmath>
#include <map>
#include <string>
#include <algorithm>
#include <atomic>
#include <thread>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <fcntl.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "device_rt.hpp"

namespace ac {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}



static const int MAX_PROBES = 1 << 14;
// An insert that walks this far has met a table that is (nearly) full — the capacity hint was too small for the input's
// diversity.  It raises the error word; every wavefront polls that word and stops, and the host retries with a table four times
// the size.  (Without the early stop a full table turns every insert into a scan of MAX_PROBES slots: minutes instead of ms.)
static const int MAX_PROBES_INSERT = 1 << 10;
// This file is compiled once per key width (-DAC_W_ONLY=1,2,3,4,8,16: the kernels and the stage code of that width only) and
// once as the main unit (AC_W_ONLY=0: everything that does not depend on the width, and the dispatch), so that the widths
// build in parallel.  The CPU emulation compiles it once with everything in.
#ifndef AC_W_ONLY
#define AC_W_ONLY 0
#endif
#if AC_W_ONLY == 0
bool g_stage_timing = false;
void set_stage_timing(bool on) { g_stage_timing = on; }
bool stage_timing() { return g_stage_timing; }
#else
extern bool g_stage_timing;
#endif

// kinfo bits (per novel k-mer, relative to the text orientation T of its smallest occurrence)
static const u32 KI_OUT_MASK = 7u, KI_IN_SHIFT = 3, KI_FIRST_T = 1u << 6, KI_FIRST_RCT = 1u << 7;

struct TextCtx {
    const u64* bits;
    const u64* mask;
    u64 n_text;
    int k;
    const u64* seq_off;
    const u32* seq_len;
    const u16* seq_d1;
    const u16* seq_d2;
    u32 n_seqs;
};

struct Table {
    u64* slots;
    u64 cap_mask;
    const u64* occ;   // optional (lookups after the insert): bit s set <=> slot s is occupied.  2 MB for 16 M slots, so it
                      // stays in L2 and answers the majority of the lookups of ABSENT k-mers (their first slot is empty
                      // with probability 1 - load) without touching the table, which only lives in the Infinity Cache
    u64* novel;       // during the insert only: bit p toggles when p becomes / stops being the position a slot holds, so that at
                      // the end of the insert bit p is set <=> p is the smallest occurrence of its canonical k-mer
    u32 n_owners;     // > 1: one job over several devices (§7) — this table only holds the k-mers whose home hash maps to `my_owner`;
    u32 my_owner;     // inserts of other keys are skipped, lookups of other keys answer "not here" (their owner answers)
    u64* sflags;      // during the insert only (optional): two SIBLING bits per slot (sib_note); MarkFunctor moves them to text positions
    u64* full_at;     // during the insert only (optional): ~(smallest text position whose insert found the table full), by atomic max
    // during the one-launch rest of a redundant text only (optional): the followed runs of at least RUN_MIN positions, three words each —
    // [first position p of the run | its length n | the position q it repeats, bit 63 = in the same orientation]: position p + i repeats
    // q + i (same) or q - i (reverse complement), 0 <= i < n.  The path walk copies the unitig paths of such stretches instead of walking
    // them (K10c).  A wavefront owns a row of RUN_ROW records (its chunk of <= 16384 positions cannot hold more runs) and counts them in
    // a register: no atomics, and rows in wavefront order ARE the runs in text order (a shared list cost the insert half a million
    // atomic appends — 1.2 ms on ONE counter, 0.03 ms on 256 — and the path stage a sort).  run_row0 = the row of this launch's wavefront 0.
    u64* runs; u32* run_count; u64 run_row0;
};
static const u64 RUN_MIN = 128;
static const u32 RUN_ROW = 128;
// the run the follow from (pj, qj) verified: positions pj + 1 .. pj + n.  `noted` = the calling wavefront's count so far (one lane calls)
AC_D void run_note(const Table& tb, u64 wave, u32& noted, u64 pj, u64 qj, bool same, u64 n) {
    if (!tb.runs || n < RUN_MIN || noted >= RUN_ROW) return;      // (a run that is not on the list is walked like any other text)
    u64* rec = tb.runs + 3 * ((tb.run_row0 + wave) * RUN_ROW + noted);
    noted++;
    // (the anchor itself repeats qj: with it on board two runs that a single-lane opener joins lie back to back, and no walker has to
    // look the one position between them up — unless qj is not a first occurrence: then it stays outside)
    const bool with_anchor = ((tb.novel[qj >> 6] >> (qj & 63)) & 1) != 0;
    if (with_anchor) { rec[0] = pj; rec[1] = n + 1; rec[2] = qj | ((u64)(same ? 1 : 0) << 63); }
    else { rec[0] = pj + 1; rec[1] = n; rec[2] = (same ? qj + 1 : qj - 1) | ((u64)(same ? 1 : 0) << 63); }
}
AC_D void run_note_done(const Table& tb, u64 wave, u32 noted) { if (tb.runs) tb.run_count[tb.run_row0 + wave] = noted; }
// Sibling bits.  Two k-mers of one middle are siblings in x (same first base, read in the orientation in which the middle is
// canonical: key_place) or in y (same last base); a k-mer WITHOUT a sibling in x / y is the only successor / predecessor its text
// neighbour can have, which the degree pass (DegreeLightFunctor) uses to skip the probe.  The insert finds the siblings for free:
// the k-mers of one middle share a home slot, so of any two of them the one in the LATER slot walked over the earlier one when it
// looked for its place (the earlier slot was occupied by then, or the walker would have taken it), and the tag shows middle
// fingerprint, x and y.  It marks both slots.  A clear bit is exact; a set bit may be a 13-bit fingerprint coincidence between
// different middles in one cluster, which only costs the probe.
AC_D void sib_note(const Table& tb, u64 s, u32 fl) { if (fl) atomic_or64(&tb.sflags[s >> 5], (u64)fl << (2 * (s & 31))); }
// v: an occupied slot a walker for the real k-mer with slot word `mine` passes.  Returns the sibling bits the two share.
AC_HD u32 sib_bits(u64 v, u64 mine) {
    const u64 d = v ^ mine;
    if ((d >> TAG_MFP_SHIFT) != 0 || slot_isdot(v) || (d >> 41) == 0) return 0;      // another middle / a dot k-mer / the same tag
    return (((d >> TAG_X_SHIFT) & 3) == 0 ? 1u : 0u) | (((d >> TAG_Y_SHIFT) & 3) == 0 ? 2u : 0u);
}
// Which rank's table a key lives in: a function of the HOME hash (key_home), so a k-mer's four successors — one middle, one
// home — have one owner, and a grouped probe is answered by a single rank.
AC_HD bool table_owns(const Table& tb, u64 home_hash) { return tb.n_owners <= 1 || (u32)((home_hash >> 40) % tb.n_owners) == tb.my_owner; }

// Largest s with off[s] <= p; valid iff p is a k-mer start of that sequence.
AC_HD bool locate(const TextCtx& t, u64 p, u32* s_out, u32* f_out) {
    if (t.n_seqs == 0 || p < t.seq_off[0]) return false;
    u32 lo = 0, hi = t.n_seqs;  // invariant: off[lo] <= p, hi exclusive
    while (hi - lo > 1) {
        u32 mid = lo + ((hi - lo) >> 1);
        if (t.seq_off[mid] <= p) lo = mid; else hi = mid;
    }
    u64 f = p - t.seq_off[lo];
    if (f >= (u64)t.seq_len[lo]) return false;
    *s_out = lo; *f_out = (u32)f;
    return true;
}

// General extended k-mer at a text position (handles dots).  False if p is not a k-mer start.
template <int W> AC_HD bool xkmer_at(const TextCtx& t, u64 p, XKmer<W>* x) {
    u32 s, f;
    if (!locate(t, p, &s, &f)) return false;
    int k = t.k;
    int plen = (int)t.seq_len[s] + k - 1;
    int ld = (int)t.seq_d1[s] - (int)f;
    int td = (int)f + k - (plen - (int)t.seq_d2[s]);
    x->ld = ld > 0 ? ld : 0;
    x->td = td > 0 ? td : 0;
    x->fwd = text_extract<W>(t.bits, p, k);
    return true;
}

// Does the k-mer whose smallest occurrence is recorded in slot value v equal `ukey`?
// 0 = no, 1 = yes and its text orientation is the canonical one, 2 = yes and it is flipped.
template <int W> AC_HD int claimant_match(const TextCtx& t, u64 v, const Key<W>& ukey) {
    XKmer<W> y;
    if (!slot_isdot(v)) {
        y.fwd = text_extract<W>(t.bits, slot_pos(v), t.k);
        y.ld = 0; y.td = 0;
    } else {
        if (!xkmer_at<W>(t, slot_pos(v), &y)) return 0;
    }
    bool yf;
    Key<W> yk = xk_canonical<W>(y, t.k, &yf);
    if (!key_eq<W>(yk, ukey)) return 0;
    return yf ? 2 : 1;
}

struct FindResult { u64 pos; int claimant_flipped; bool found; };

template <int W> AC_HD FindResult table_find(const TextCtx& t, const Table& tb, const Key<W>& ukey, bool isdot) {
    const KeyPlace pl = key_place<W>(ukey, t.k, isdot, key_hash<W>(ukey));
    u64 tag = slot_make(pl.tag, isdot, 0);
    const u64 hh = pl.home;
    u64 s = hh & tb.cap_mask;
    FindResult r; r.found = false; r.pos = 0; r.claimant_flipped = 0;
    if (!table_owns(tb, hh)) return r;
    if (tb.occ && !((tb.occ[s >> 6] >> (s & 63)) & 1)) return r;
    for (int probes = 0; probes < MAX_PROBES; probes++) {
        u64 v = tb.slots[s];
        if (v == SLOT_EMPTY) return r;
        if (slot_tag_eq(v, tag)) {
            int m = claimant_match<W>(t, v, ukey);
            if (m) { r.found = true; r.pos = slot_pos(v); r.claimant_flipped = (m == 2); return r; }
        }
        s = (s + 1) & tb.cap_mask;
    }
    return r;
}

// Lookup of an extended k-mer in text orientation.  *pos = the k-mer's smallest ("novel") text position;
// rel_same: the query reads the same way as that smallest occurrence does in the text.
template <int W> AC_HD bool find_xk(const TextCtx& t, const Table& tb, const XKmer<W>& x, u64* pos, bool* rel_same) {
    bool flipped;
    Key<W> uk = xk_canonical<W>(x, t.k, &flipped);
    FindResult r = table_find<W>(t, tb, uk, x.ld > 0 || x.td > 0);
    if (!r.found) return false;
    *pos = r.pos;
    *rel_same = ((r.claimant_flipped != 0) == flipped);
    return true;
}

// Does this rank's table own the extended k-mer x (always true on a single device)?
template <int W> AC_HD bool owns_xk(const TextCtx& t, const Table& tb, const XKmer<W>& x) {
    if (tb.n_owners <= 1) return true;
    bool flipped;
    Key<W> uk = xk_canonical<W>(x, t.k, &flipped);
    return table_owns(tb, key_home<W>(uk, t.k, x.ld > 0 || x.td > 0, key_hash<W>(uk)));
}

// Rank support over the novel-position bitmap: index of a novel position in the sorted novel list.
struct Novel {
    const u64* bm;        // bit p set <=> p is the smallest occurrence of its canonical k-mer
    const u32* wprefix;   // number of set bits before word w
};
AC_HD int popc64(u64 x) {
#ifdef AC_EMU
    return __builtin_popcountll(x);
#else
    return __popcll(x);
#endif
}
AC_HD u32 novel_rank(const Novel& nv, u64 pos) {
    u64 w = pos >> 6;
    int b = (int)(pos & 63);
    u64 below = b ? (nv.bm[w] & ((1ULL << b) - 1)) : 0;
    return nv.wprefix[w] + (u32)popc64(below);
}

static const u64 NOREF = ~0ULL;

// Insert with "smallest text position wins" semantics.  Stale (cached) reads of a slot can only show
// an older state of a monotone word (EMPTY -> pos -> smaller pos of the same key), so every decision
// taken on them stays valid; claiming is decided by the CAS alone.
// Returns the position q < p of an EARLIER occurrence of the same canonical k-mer if the slot showed one
// (*same = it reads in the same orientation as the occurrence at p), else NOREF.
// *mine_now = p has just become the position its slot holds (claimed an empty slot, or lowered a larger position): the
// caller toggles bit p of the novel bitmap (a wavefront does it for its 64 lanes with one 64-bit atomic); the bit of a
// position this call displaced is toggled here.  Every position becomes the slot value at most once and is displaced at most
// once, and XOR commutes, so whatever order the atomics land in, the bitmap ends with exactly the final slot positions set.
template <int W> AC_D u64 table_insert(const TextCtx& t, const Table& tb, const Key<W>& ukey, bool isdot, bool flipped, u64 p,
                                       u32* claimed, u32* err, bool* same, bool* mine_now) {
    const KeyPlace pl = key_place<W>(ukey, t.k, isdot, key_hash<W>(ukey));
    u64 mine = slot_make(pl.tag, isdot, p);
    const u64 hh = pl.home;
    u64 s = hh & tb.cap_mask;
    *mine_now = false;
    if (!table_owns(tb, hh)) return NOREF;      // another rank's k-mer
    const bool note = tb.sflags != nullptr && !isdot;
    u32 my_fl = 0;
    for (int probes = 0; probes < MAX_PROBES_INSERT; probes++) {
        u64 v = tb.slots[s];
        if (v == SLOT_EMPTY) {
            u64 old = atomic_cas64(&tb.slots[s], SLOT_EMPTY, mine);
            if (old == SLOT_EMPTY) { (*claimed)++; *mine_now = true; if (note) sib_note(tb, s, my_fl); return NOREF; }
            v = old;
        }
        if (slot_tag_eq(v, mine)) {
            if (slot_pos(v) == p) { if (note) sib_note(tb, s, my_fl); return NOREF; }
            int m = claimant_match<W>(t, v, ukey);
            if (m) {
                if (note) sib_note(tb, s, my_fl);
                if (slot_pos(v) > p) {
                    u64 old = atomic_min64(&tb.slots[s], mine);      // the same key's word: tag and isdot agree, positions order it
                    if (old > mine) {
                        *mine_now = true;
                        u64 q = slot_pos(old);
                        atomic_xor64(&tb.novel[q >> 6], 1ULL << (q & 63));
                    }
                    return NOREF;
                }
                *same = ((m == 2) == flipped);
                return slot_pos(v);
            }
        } else if (note) {
            const u32 fl = sib_bits(v, mine);
            if (fl) { sib_note(tb, s, fl); my_fl |= fl; }
        }
        s = (s + 1) & tb.cap_mask;
    }
    atomic_or32(err, 1u);
    if (tb.full_at) atomic_max64(tb.full_at, ~p);
    return NOREF;
}

struct alignas(16) V16 { u32 a, b, c, d; };

#include "kernels_table.inc"      // K1 pack, K2 / K2w k-mer insert, K3 novel list
#include "kernels_unitigs.inc"      // K5 degrees, K6 first flags, K7 heads, K8 seed k-mers, K9 unitig metadata, K11 links
#include "kernels_paths.inc"      // K10 path walk
#include "kernels_tail.inc"      // K13 link order, K14 analysis, K15 renumbering, K17 expand_repeats, K16 finalisation, K12 sequences
#include "kernels_shard.inc"      // fragments and reduce buffers of a sharded build
// =============================================================================================================
// The paths' final numbers, applied on the host.  The path entries are final — in SEED numbers — when the walk ends, the final numbers
// exist only after expand_repeats and the second renumbering, and 4 bytes per entry over PCIe were the last thing a build waited for
// (config C: 42 MB = 0.7 ms of 4.5).  A single-device build therefore sends the entries right after the walk, under the whole tail,
// and the final number per seed index (4 bytes per unitig) as soon as it exists; host threads rewrite the entries in the pinned result
// block while the remaining results (unitig records, links) are still crossing.  (unitig_graph.rs:renumber_unitigs only permutes.)
struct PathRemapJob {
    int32_t* path = nullptr; u64 n_ent = 0;
    const u32* number = nullptr; u32 n_unitigs = 0;      // pinned: final number of seed index r at [r]
    void* landed = nullptr;                              // event: entries and number table are in host memory
    int dev = 0;
    std::atomic<u64> next{0}; std::atomic<int> ready{0};      // ready: 0 nobody waits yet, 1 one thread waits for `landed`, 2 go, 3 failed
    std::atomic<u32> bad{0};                             // entries that name no unitig (never, short of a bug: reported as an internal error)
    u64 ticket = 0; bool started = false;
};
void path_remap_range(int32_t* p, u64 n, const u32* number, u32 n_unitigs, std::atomic<u32>* bad);
bool path_remap_is_wide();      // the host has the 16-lane gather (without it a thread renumbers ~5x slower and the device keeps the job)
void path_remap_start(PathRemapJob& j, int threads);      // returns at once; the work runs on the packing threads' pool
void path_remap_finish(PathRemapJob& j) noexcept;         // until every thread is done (idempotent)
#if AC_W_ONLY == 0
std::vector<uint8_t> layout_text(const std::vector<SeqView>& seqs, uint32_t k, std::vector<uint64_t>* off,
                                 std::vector<uint32_t>* len, std::vector<uint16_t>* d1, std::vector<uint16_t>* d2) {
    u64 n = 1;
    for (auto& s : seqs) n += (u64)s.length + k - 1 + 1;
    std::vector<uint8_t> text(n);
    off->clear(); len->clear(); d1->clear(); d2->clear();
    u64 p = 0;
    text[p++] = '$';
    for (auto& s : seqs) {
        u64 plen = (u64)s.length + k - 1;
        off->push_back(p);
        len->push_back(s.length);
        memcpy(&text[p], s.fwd, plen);
        u16 a = 0, b = 0;
        while (a < plen && s.fwd[a] == '.') a++;
        while (b < plen && s.fwd[plen - 1 - b] == '.') b++;
        d1->push_back(a); d2->push_back(b);
        p += plen;
        text[p++] = '$';
    }
    return text;
}

int max_supported_k() { return 501; }   // the reference's own limit (compress.rs:56-60); keys of 1, 2, 3, 4, 8 or 16 words
#endif
[[maybe_unused]] static int key_words(int k) { int w = words_for_k(k); return w <= 4 ? w : (w <= 8 ? 8 : 16); }

// renumber_unitigs (unitig_graph.rs:295-315): stable sort of `order` by (length desc, sequence asc, depth desc).
[[maybe_unused]] static bool renum_two_pass() { const char* e = getenv("AC_RENUM_TWO_PASS"); return e && atoi(e) != 0; }      // 1: always the two-pass renumber sort
[[maybe_unused]] static u32 renum_max_group() { const char* e = getenv("AC_RENUM_MAX_GROUP"); int v = e ? atoi(e) : (int)RENUM_MAX_GROUP; return (u32)(v < 1 ? 1 : (v > (int)RENUM_MAX_GROUP ? (int)RENUM_MAX_GROUP : v)); }      // tests: smaller groups take the fallbacks
// deferred: do not wait for the "group too large" flag (a host round trip per renumbering) — the caller reads it with the build's last
// read-back and repeats the build with checked sorts if it was ever set (GraphBuilder::build; the flag is sticky then: never cleared here).
[[maybe_unused]] static void renumber_sort(DBuf<u32>& order, u32 U, const u32* len, const u64* off, const u8* seq, const u32* depth, u32* flag, bool deferred = false) {
    if (U <= 1) return;
    DBuf<u32> backup(deferred && !renum_two_pass() ? 0 : U);      // (the order to fall back from: only a checked sort ever does)
    if (backup.size()) copy_d2d(backup.ptr(), order.ptr(), (size_t)U * 4);
    DBuf<u64> prefix(U), key(U);
    launch(U, RenumKeyFunctor{len, off, seq, prefix.ptr()});
    UnitigLess less{len, off, seq, depth};
    u32 zero = 0;
    if (!renum_two_pass()) {      // one sort on (length | 16 bases), ties by the comparator
        launch(U, RenumPassFunctor{order.ptr(), len, depth, prefix.ptr(), 2, key.ptr()});
        sort_pairs_u64_u32(key, order, U, 64);
        launch(U, RenumTieFunctor{order.ptr(), U, len, depth, prefix.ptr(), less, flag, 0, renum_max_group()});
#ifdef AC_EMU
        if (getenv("AC_DEGREE_DIAG")) {
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

static u64 next_pow2(u64 x) { u64 p = 1; while (p < x) p <<= 1; return p; }
// Device memory one build of an n_text-byte text needs, roughly: packed text + bitmaps (~0.6 B/position), staging slots
// of the path walk (4 B/position), k-mer table and per-k-mer arrays (sized by distinct content), unitig-sized buffers.
[[maybe_unused]] static size_t arena_estimate(u64 n_text, bool owns_text) { return (size_t)n_text * (owns_text ? 8 : 7) + ((size_t)768 << 20); }
// Tuning knobs (environment), read on every build so that one process can compare settings (tools/ab_knobs.py; measurements in
// profiles/r03*_ab_knobs_configC.jsonl, profiles/r04*_ab_*.jsonl).  None of them changes a result (tests: *_tuning_knobs_*):
//   AC_TABLE_SHIFT    k-mer table capacity = 2^n x the reference-style sizing.  Unset = automatic: 1 (load ~0.23 on similar
//                     assemblies: short probe clusters) while the table stays about cache-sized, 0 for tables far beyond it.
//   AC_MINKEY_VARIANT seed k-mer per unitig: 2 = on a 64-bit key prefix, one full key per unitig; 1 = wavefront segmented min with the full
//                     keys in registers; 0 = key records + library reduce-by-key.  Unset = automatic (2 for long keys and unitigs, else 1).
//   AC_MINKEY_PREFIX_BASES   (tests) bases in that prefix, default 31.
//   AC_SEED_PREFIX_SORT  1 (default): seed order by one sort on a 64-bit prefix of the seed keys + full-key ranking inside the groups that
//                     agree on it (AC_SEED_PREFIX_BITS: tests; a group of more than AC_SEED_MAX_GROUP = 1024 members sends the build to the
//                     full-key sorts); 0: the full-key sorts —
//   AC_SEED_RADIX_LIMIT  unitigs from which those are W radix passes instead of the comparator merge sort (default 2^19).
//   AC_PATH_CHUNK     text positions per path walker (default: 5 x the mean unitig length, a power of two in [64, 2048]).
//   AC_PATH_FILTER    1 (default): the walk keeps smallest positions only for unitig sides that can become expand_repeats
//                     destinations.
//   AC_POS_CAP        (65536) single-device builds: occurrences further than this from both ends of their sequence do not lower a
//                     unitig's smallest positions; beyond it expand_repeats works with a lower bound and, where that cannot decide,
//                     the build is repeated with exact positions (kernels_tail.inc exp_avoid_start_of_path).  0: every occurrence counts.
//   AC_PATH_COPY      the copying path walk (K10c: followed runs are copied from the stretch they repeat, the text between them is
//                     walked): 1 whenever the insert has a one-launch rest, 0 never, unset: where the cost model says it pays
//                     (path_copy_pays).  AC_RUN_PIECE (4096): positions per copied piece of a run (tests).  AC_SHARD_PATH_COPY (1): the
//                     same for a rank's own sequences in a sharded build (round 5).
//   AC_REMAP_BLOCK    path entries per wavefront in the final renumbering (tests).
//   AC_INSERT_CHUNK / AC_INSERT_GROWTH / AC_INSERT_WAVES   insert phases: longest wavefront chunk, prefix growth factor, wavefronts per phase;
//   AC_INSERT_ADAPT (default 1)   redundant text: everything after the second phase in one launch, in chunks of 16384 positions — or
//                     shorter ones where a sample of that rest finds content of its own (round 5); AC_INSERT_CHUNK_REST fixes the chunk.
//   AC_EXPAND_REWRITE_ALWAYS  rewrite the sequences contiguously after every host check of the expand passes (tests);
//                     AC_EXPAND_LEVEL_TABLE (1024): levels the first read of the level bounds holds (tests: the exact second read).
//   AC_SORT_CHECKS    1: every "group too large" flag of a sort read where it is raised (default: with the build's last read-back, and a
//                     build that had one set is repeated).
//   AC_SHARD_DEGREE_FLAGS (1) / AC_SHARD_HOST_REMAP (1)   sharded builds: sibling bits + probe-free degrees; own paths renumbered on the host.
//   AC_SEQ_WRITER     0 / 1 = always the search-per-thread / the indexed LDS-tiled sequence writers (default: by output size).
//   AC_DEGREE_FLAGS   1 (default): degrees from the sibling bits the insert collects, probes only where they do not settle it (two
//                     passes); 0: every degree by probing (what sharded builds and k < 3 do).
//   AC_RENUM_TWO_PASS 1: renumber with two sorts (length | 32 bases | depth) instead of one (length | 16 bases); AC_RENUM_MAX_GROUP (tests).
//   AC_UPLOAD_THREADS (24) / AC_HOST_PACK (1) / AC_UPLOAD_OVERLAP (1)   host entry: packing threads, 2-bit pack on the host, the
//                     insert issued chunk by chunk while background threads still pack and send the rest (0: everything is sent
//                     before anything else is issued); AC_UPLOAD_SLOTS (tests: staging slots).
//   AC_NO_MAILBOX     (read once) small read-backs through hipMemcpyAsync + synchronise instead of the mapped mailbox page.
//   AC_INSERT_PROFILE (read once) per-wavefront cycle split of every insert launch on stderr (measurement).
//   AC_DEBUG_LAUNCH   (read once) every functor launch announced on stderr and waited for (device_rt.hpp); AC_DEBUG_ARENA: arena and copy-walk figures.
[[maybe_unused]] static int minkey_variant() { const char* e = getenv("AC_MINKEY_VARIANT"); return e ? atoi(e) : -1; }      // -1 = automatic
// AC_PATH_COPY: 0 never / 1 whenever a redundant text has a one-launch rest / unset: when the cost model below says it pays
[[maybe_unused]] static int path_copy() { const char* e = getenv("AC_PATH_COPY"); return e ? (atoi(e) != 0 ? 1 : 0) : 2; }
// The copying path walk (K10c) against the plain one, as measured on MI355X (profiles/r09e_ab_path_copy_rows.txt): the
// plain walk costs ~70 ps per path entry (its depth atomic, successor gather and staging), the copying walk ~17 ps per copied entry plus
// ~0.25 ms of launches and read-backs, and the insert ~0.3 ps per text position for ending runs where their source stops being a first
// occurrence; the first two assemblies' worth of text (the phases before the one-launch rest) is walked either way.  Path entries are
// estimated from what the insert knows when it decides: of the second assembly's worth of text a share r2 was new k-mers, ~k per variant
// site, and every site of every one of the A assemblies cuts the unitigs of the final graph about twice.  The estimate is rough (config C:
// 13.8 M for 10.6 M entries; config D, k = 101: 25 M for ~18 M, and its copying stage gains less than this model says), so the copying walk
// is only chosen where the predicted saving is half again the predicted cost: on, of the measured workloads, config C (-3.5 %) and off
// on B, D', D (where it would cost 2 %), E'.
[[maybe_unused]] static bool path_copy_pays(u64 n_text, u32 assemblies, u32 k, double r2) {
    const double A = (double)std::max<u32>(assemblies, 1);
    const double entries = (double)n_text * std::min(1.0, 2.0 * A * r2 / (double)k);
    const double saving = entries * 53e-12 * std::max(0.0, 1.0 - 2.0 / A), cost = 0.25e-3 + 0.3e-12 * (double)n_text;
    return saving > 1.5 * cost;
}
[[maybe_unused]] static u64 run_piece() { const char* e = getenv("AC_RUN_PIECE"); const long v = e ? atol(e) : 0; return v > 0 ? (u64)v : 4096; }      // positions per copied piece of a run (RunFilterFunctor)
// AC_POS_CAP: occurrences further than this from both ends of their sequence do not lower a unitig's smallest positions (0 = all do)
[[maybe_unused]] static u32 pos_cap() { const char* e = getenv("AC_POS_CAP"); const long v = e ? atol(e) : 65536; return v < 0 ? 0u : (u32)std::min<long>(v, 0x3FFFFFFF); }
[[maybe_unused]] static bool path_filter() { const char* e = getenv("AC_PATH_FILTER"); return e ? atoi(e) != 0 : true; }   // smallest positions only for possible expand_repeats destinations
// AC_PATH_DIAG (skips the walk's depth atomics / position updates to price them: the result is WRONG when set) only exists in
// builds made with -DAC_MEASUREMENT_KNOBS; the shipped library ignores the variable.
#ifdef AC_MEASUREMENT_KNOBS
[[maybe_unused]] static int path_diag() { const char* e = getenv("AC_PATH_DIAG"); return e ? (atoi(e) & 3) : 0; }
#else
[[maybe_unused]] static int path_diag() { return 0; }
#endif
// Text positions per path walker.  A walker pays one table lookup and then one dependent gather per unitig it steps through: the
// chunk is sized for ~5 unitigs per walker — 5 x the mean unitig length N / U, to the nearest power of two in [64, 2048] (config C
// 256, config D 512, E' 64; r06h: C 128 / 256 / 512 = 1.00 / 0.93 / 1.03 ms, D 256 / 512 / 2048 = 2.15 / 1.78 / 1.50 ms, E' 128 /
// 256 = 1.45 / 1.58 ms).  AC_PATH_CHUNK overrides.
[[maybe_unused]] static u32 path_chunk(u64 n_kmers, u32 n_unitigs) {
    const char* e = getenv("AC_PATH_CHUNK");
    if (e) { int v = atoi(e); return (u32)(v < 64 ? 64 : (v > 4096 ? 4096 : v)); }
    const u64 want = 5 * n_kmers / std::max<u32>(n_unitigs, 1);
    u32 pc = 64;
    while (pc < 2048 && (u64)pc * 3 / 2 < want) pc *= 2;
    return pc;
}
// Path entries leave the device in seed numbers right after the walk and get their final numbers on the host (single-device builds):
// 1 always, 0 never, otherwise when the number table (4 bytes per unitig) stays in the host's caches — up to 8 M unitigs — and there is
// enough to hide.  Measured (r10p/q): config C 4.50 -> 3.93 ms, E' 18.5 -> 17.7, mini-E (6.5 M unitigs) 69.8 -> 64.5; with 26 M unitigs
// (8 species) 277 -> 321 ms and with 82 M (configs[4]) 0.89 -> 1.29 s: random gathers from a table in DRAM are slower than the link.
[[maybe_unused]] static int host_remap_mode() { const char* e = getenv("AC_HOST_REMAP"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }
[[maybe_unused]] static u32 remap_block() { const char* e = getenv("AC_REMAP_BLOCK"); int v = e ? atoi(e) : 4096; v = v < 64 ? 64 : (v > 65536 ? 65536 : v); return (u32)(v & ~63); }
[[maybe_unused]] static int minkey_prefix_bases() { const char* e = getenv("AC_MINKEY_PREFIX_BASES"); int v = e ? atoi(e) : 31; return v < 1 ? 1 : (v > 31 ? 31 : v); }      // tests: a shorter prefix takes the full-key path often
[[maybe_unused]] static bool seed_prefix_sort() { const char* e = getenv("AC_SEED_PREFIX_SORT"); return e ? atoi(e) != 0 : true; }      // 0: seed order by the full-key sorts
[[maybe_unused]] static u32 seed_max_group() { const char* e = getenv("AC_SEED_MAX_GROUP"); int v = e ? atoi(e) : 1024; return (u32)(v < 1 ? 1 : v); }      // tests: smaller groups take the fallback
[[maybe_unused]] static int seed_prefix_bits() { const char* e = getenv("AC_SEED_PREFIX_BITS"); if (!e) return 0; int v = atoi(e); return v < 1 ? 1 : (v > 64 ? 64 : v); }      // tests; unset = 0 = automatic
[[maybe_unused]] static u32 degree_region_cap() { const char* e = getenv("AC_DEGREE_REGION_CAP"); int v = e ? atoi(e) : 0; return (u32)(v < 0 ? 0 : v); }      // tests: entries per queue region (0 = sized from N)
[[maybe_unused]] static u64 upload_chunk_bytes() { return (u64)64 << 20; }      // text bytes per upload chunk (16 MB of codes per copy; 8-32 MB chunks over 2-3 copy queues: 2.8 against 3.2 ms in tools/microbench/upload_probe.hip, nothing in the build: r10o)
// The packers write the codes straight into device memory (through the PCIe BAR, write-combined) instead of into a pinned ring a copy
// engine then reads: 1 / 0 forces / forbids, otherwise on when the device says its whole memory is host-visible (hipDeviceAttributeIsLargeBar).
[[maybe_unused]] static int upload_direct_mode() { const char* e = getenv("AC_UPLOAD_DIRECT"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }
#ifndef AC_EMU
template <int UNUSED> __global__ void __launch_bounds__(256) bar_selftest_kernel(const u64* p, u64 n, u64* out) {
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) acc += p[i] * (i + 1);
    if (acc) atomicAdd((unsigned long long*)out, (unsigned long long)acc);
}
// ONE self-test per device before the packers are allowed to store into device memory: that the device reports a large BAR
// does not promise that a hipMalloc pointer can be stored through from the host, nor that a kernel then sees what was stored.  The test
// (a) asks the kernel whether the pointer is host-writable WITHOUT touching it (read(2) into it fails with EFAULT instead of a fault),
// (b) stores a pattern through it the way the packers do (plain stores, store fence, one read back), (c) has a kernel on the device
// checksum the buffer.  Any mismatch — or any HIP error — sends every build on this device through the pinned ring.
static bool bar_selftest(int dev) {
    const u64 n = (u64)1 << 17;      // 1 MB of words
    u64* d = nullptr; u64* d_out = nullptr;
    bool ok = false;
    int fd = -1;
    do {
        if (hipSetDevice(dev) != hipSuccess) break;
        if (hipMalloc((void**)&d, n * 8) != hipSuccess || hipMalloc((void**)&d_out, 8) != hipSuccess) break;
        if (hipMemset(d, 0, n * 8) != hipSuccess || hipMemset(d_out, 0, 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
        fd = ::open("/dev/zero", O_RDONLY);
        if (fd < 0) break;
        if (::read(fd, (void*)d, 4096) != 4096 || ::read(fd, (void*)(d + n - 512), 4096) != 4096) break;      // EFAULT: not mapped for the host
        u64 expect = 0;
        for (u64 i = 0; i < n; i++) { const u64 v = (i * 0x9E3779B97F4A7C15ULL) | 1ULL; d[i] = v; expect += v * (i + 1); }
#if defined(__x86_64__)
        _mm_sfence();
#endif
        std::atomic_thread_fence(std::memory_order_seq_cst);
        const volatile u64* back = d + (n - 1);
        if (*back != (((n - 1) * 0x9E3779B97F4A7C15ULL) | 1ULL)) break;      // (a PCIe read does not pass the posted writes before it)
        hipLaunchKernelGGL(bar_selftest_kernel<0>, dim3(256), dim3(256), 0, 0, (const u64*)d, n, d_out);
        u64 got = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpy(&got, d_out, 8, hipMemcpyDeviceToHost) != hipSuccess) break;
        ok = got == expect;
    } while (false);
    if (fd >= 0) ::close(fd);
    (void)hipGetLastError();
    if (d) (void)hipFree(d);
    if (d_out) (void)hipFree(d_out);
    if (getenv("AC_DEBUG_ARENA")) fprintf(stderr, "direct upload self-test on device %d: %s\n", dev, ok ? "passed" : "FAILED (the packed upload goes through the pinned ring)");
    return ok;
}
#endif
[[maybe_unused]] static bool upload_direct_for(int dev) {
#ifndef AC_EMU
    if (upload_direct_mode() == 0) return false;
    if (upload_direct_mode() < 0) {
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (!large_bar) return false;
    }
    // (forced on or offered by the device: either way only after the self-test, once per device and process)
    static std::mutex mu; static std::map<int, bool> tested;
    std::lock_guard<std::mutex> lock(mu);
    auto it = tested.find(dev);
    if (it == tested.end()) it = tested.emplace(dev, bar_selftest(dev)).first;
    return it->second;
#else
    (void)dev; return false;
#endif
}
[[maybe_unused]] static int upload_slots() { const char* e = getenv("AC_UPLOAD_SLOTS"); int v = e ? atoi(e) : 1 << 20; return v < 1 ? 1 : v; }      // tests: fewer staging slots, so that chunks wait for one
[[maybe_unused]] static int degree_flags() { const char* e = getenv("AC_DEGREE_FLAGS"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }      // 0: every degree by probing (what sharded builds and k < 3 do)
[[maybe_unused]] static int table_shift() { const char* e = getenv("AC_TABLE_SHIFT"); if (!e) return -1; int v = atoi(e); return v < 0 ? 0 : (v > 3 ? 3 : v); }      // -1 = automatic
[[maybe_unused]] static u64 wave_chunk_max() { const char* e = getenv("AC_INSERT_CHUNK"); u64 x = e ? (u64)atoll(e) : 8192; return (std::max<u64>(x, 256) + 63) & ~63ULL; }
[[maybe_unused]] static u64 wave_chunk_rest() { return 16384; }   // longest chunk of the one-launch rest (r04c, config C: 4096 / 8192 / 16384 = 0.90 / 0.84 / 0.82 ms); shorter where the rest brings new content
[[maybe_unused]] static u64 insert_chunk_rest_env() { const char* e = getenv("AC_INSERT_CHUNK_REST"); if (!e) return 0; return (std::max<u64>((u64)atoll(e), 256) + 63) & ~63ULL; }      // measurement / tests: the chunk of the rest, whatever the sample says   // longest chunk of the one-launch rest (r04c: 4096 / 8192 / 16384 = 0.90 / 0.84 / 0.82 ms)
static thread_local int tl_upload_threads_cap = 0;      // a rank of a multi-device build: its share of the host's cores (0 = no cap)
[[maybe_unused]] static u64 upload_threads() {
    const char* e = getenv("AC_UPLOAD_THREADS");
    // 24 since round 5: 16 / 24 / 32 threads pack config C at the same median (5.9-6.0 ms per build, two 64-core sockets), but with 32 one step
    // in twenty waits 10-20 ms for a straggler (unpinned threads on a shared host): mean 6.5-6.8 ms against 5.95-6.07 (r13b)
    long x = e ? atol(e) : 24;
    if (tl_upload_threads_cap > 0 && x > tl_upload_threads_cap) x = tl_upload_threads_cap;
    return (u64)(x < 1 ? 1 : (x > 128 ? 128 : x));
}   // host threads laying out / packing the text (the byte upload uses at most 8)
[[maybe_unused]] static bool host_pack() { const char* e = getenv("AC_HOST_PACK"); return e ? atoi(e) != 0 : true; }      // 0: upload the text as bytes and pack on the device
[[maybe_unused]] static bool insert_profile() { static const bool v = getenv("AC_INSERT_PROFILE") != nullptr; return v; }      // measurement only
[[maybe_unused]] static u32 expand_level_table() { const char* e = getenv("AC_EXPAND_LEVEL_TABLE"); int v = e ? atoi(e) : 1024; return (u32)(v < 1 ? 1 : v); }      // tests: a table too small for the levels
[[maybe_unused]] static bool shard_path_copy() { const char* e = getenv("AC_SHARD_PATH_COPY"); return !(e && atoi(e) == 0); }      // 0 = a sharded build walks all of its text (rounds 3-4)
[[maybe_unused]] static bool expand_rewrite_always() { return getenv("AC_EXPAND_REWRITE_ALWAYS") != nullptr; }      // tests: compact the expand pool after every host check
[[maybe_unused]] static bool seq_writer_plain() { const char* e = getenv("AC_SEQ_WRITER"); return e && atoi(e) == 0; }      // 0 = always the search-per-thread writers
[[maybe_unused]] static bool seq_writer_forced() { const char* e = getenv("AC_SEQ_WRITER"); return e && atoi(e) == 1; }    // 1 = always the indexed / LDS-tiled writers
[[maybe_unused]] static u64 seed_radix_limit() { const char* e = getenv("AC_SEED_RADIX_LIMIT"); return e ? (u64)atoll(e) : (1u << 19); }      // unitigs from which the seed order is a radix sort
[[maybe_unused]] static bool upload_overlap() { const char* e = getenv("AC_UPLOAD_OVERLAP"); return e ? atoi(e) != 0 : true; }      // host entry: first insert phases while the upload's tail is in flight
[[maybe_unused]] static bool sort_checks_deferrable() { const char* e = getenv("AC_SORT_CHECKS"); return !(e && atoi(e) == 1); }      // 1 = every "group too large" flag read where it is raised (round 4)
[[maybe_unused]] static bool shard_host_remap() { const char* e = getenv("AC_SHARD_HOST_REMAP"); return e ? atoi(e) != 0 : true; }      // 0 = sharded builds renumber their paths on the device (round 4)
[[maybe_unused]] static bool shard_degree_flags() { const char* e = getenv("AC_SHARD_DEGREE_FLAGS"); return e ? atoi(e) != 0 : true; }      // 0 = sharded builds probe every degree (round 4)
[[maybe_unused]] static bool insert_adaptive() { const char* e = getenv("AC_INSERT_ADAPT"); return e ? atoi(e) != 0 : true; }
[[maybe_unused]] static u64 insert_growth() { const char* e = getenv("AC_INSERT_GROWTH"); long x = e ? atol(e) : 2; return (u64)(x < 2 ? 2 : x); }      // phase i+1 ends at growth x the end of phase i
[[maybe_unused]] static u64 insert_waves_target() { const char* e = getenv("AC_INSERT_WAVES"); long x = e ? atol(e) : 16384; return (u64)(x < 1024 ? 1024 : x); }   // wavefronts a long phase is cut into

// A text resident in HBM with its sequence table and its 2-bit packing.
struct PackedText {
    const u8* d_text = nullptr;
    u64 n_text = 0, n_bases = 0;
    u32 n_seqs = 0;
    int any_dots = 0; u64 n_dotted = 0;      // sequences (fragments) that kept a dot at either end
    DBuf<u64> seq_off; DBuf<u32> seq_len; DBuf<u16> seq_d1, seq_d2; DBuf<u8> seq_flags;
    bool has_flags = false;
    DBuf<u64> bits, mask;
    // alphabet check of K1 (pack_check, sequence.rs:39-41): [0] = smallest (sequence index + 1) holding an illegal byte, [1] = number of
    // non-base bytes - 1 (both start as all-ones); expected_nonbase = padding dots + separators the sequence table promises
    DBuf<u32> pack_bad; bool check_alphabet = false; int k = 0; u64 expected_nonbase = 0;
    u64 index_base = 0;      // sequences of the job in front of this text's first one (a rank of a multi-device build): for messages
    PackCheck chk() const { return PackCheck{seq_off.ptr(), seq_len.ptr(), seq_d1.ptr(), seq_d2.ptr(), n_seqs, k, check_alphabet ? const_cast<u32*>(pack_bad.ptr()) : nullptr}; }
    std::vector<u64> h_off; std::vector<u32> h_len;
    void set_table(const std::vector<uint64_t>& off, const std::vector<uint32_t>& len, const std::vector<uint16_t>& d1,
                   const std::vector<uint16_t>& d2, const std::vector<uint8_t>* flags = nullptr) {
        n_seqs = (u32)off.size();
        h_off = off; h_len = len;
        seq_off.alloc(n_seqs); seq_len.alloc(n_seqs); seq_d1.alloc(n_seqs); seq_d2.alloc(n_seqs);
        copy_h2d(seq_off.ptr(), off.data(), (size_t)n_seqs * 8);
        copy_h2d(seq_len.ptr(), len.data(), (size_t)n_seqs * 4);
        copy_h2d(seq_d1.ptr(), d1.data(), (size_t)n_seqs * 2);
        copy_h2d(seq_d2.ptr(), d2.data(), (size_t)n_seqs * 2);
        has_flags = flags != nullptr;
        if (flags) { seq_flags.alloc(n_seqs); copy_h2d(seq_flags.ptr(), flags->data(), n_seqs); }
        n_bases = 0; any_dots = 0; n_dotted = 0; expected_nonbase = (u64)n_seqs + 1;
        for (u32 i = 0; i < n_seqs; i++) { n_bases += len[i]; if (d1[i] || d2[i]) { any_dots = 1; n_dotted++; } expected_nonbase += (u64)d1[i] + d2[i]; }
        stream_sync();
    }
    // The table of a union text, built on the device (build_union_impl): the arrays are filled by the caller, the sums come with its read-back.
    void alloc_table(u32 n) { n_seqs = n; h_off.clear(); h_len.clear(); seq_off.alloc(n); seq_len.alloc(n); seq_d1.alloc(n); seq_d2.alloc(n); seq_flags.alloc(n); has_flags = true; }
    void set_sums(u64 bases, u64 dotted, u64 dots) { n_bases = bases; n_dotted = dotted; any_dots = dotted ? 1 : 0; expected_nonbase = (u64)n_seqs + 1 + dots; }
    TextCtx ctx(int k) const { return TextCtx{bits.ptr(), mask.ptr(), n_text, k, seq_off.ptr(), seq_len.ptr(), seq_d1.ptr(), seq_d2.ptr(), n_seqs}; }
    bool packed = false;      // the host entry packs chunk by chunk behind the upload (set_sequences_host)
    void pack_alloc(stream_t s = 0) {
        u64 n_bits_words = n_text / 32 + 24, n_mask_words = n_text / 64 + 12;   // slack for W <= 16 key words
        bits.alloc(n_bits_words); mask.alloc(n_mask_words);
        // every 32-position group of the text gets its code word and its half mask word written (PackFunctor, or the host packers'
        // copies): only the slack behind the last group has to be set — zero codes, all-ones mask (183 MB of fills per build on config C)
        const u64 groups = (n_text + 31) / 32;
        bits.fill_bytes_from(groups * 8, 0, s);
        mask.fill_bytes_from(groups * 4, 0xFF, s);
        if (check_alphabet) { pack_bad.alloc(2); pack_bad.fill_bytes(0xFF, s); }
    }
    // What pack_check found (read back with the build's last read-back): throws the reference's message for a text that holds
    // anything but A, C, G, T between its padding dots (sequence.rs:39-41).
    void verify_alphabet(const u32* bad) const {
        if (!check_alphabet) return;
        if (bad[0] != 0xFFFFFFFFu) throw DeviceError("input sequence " + std::to_string(index_base + bad[0]) + " contains non-ACGT characters");
        const u64 found = (u64)(u32)(bad[1] + 1u);
        if (found != (expected_nonbase & 0xFFFFFFFFULL))
            throw DeviceError("the text does not match its sequence table: " + std::to_string(expected_nonbase) + " padding dots and separators expected, " +
                              std::to_string(found) + " non-ACGT characters found");
    }
    void pack() {   // K1
        if (packed) return;
        pack_alloc();
        launch((n_text + 31) / 32, PackFunctor{d_text, n_text, bits.ptr(), (u32*)mask.ptr(), 0, chk()});
        packed = true;
    }
};

// All device state of one build.  Buffers are slices of the device arena, which the owning GraphBuilder resets
// when it is created, so the state of a sharded build survives between its phases.
struct GraphBuilder::Impl {
    u32 k = 0;
    DBuf<u8> text_owned;
    PackedText loc;            // this rank's sequences
    PackedText uni;            // sharded builds: union of all ranks' fragments
    PackedText* G = &loc;      // the text the graph is built from
    BuildTimings* tm = nullptr;
    double t0 = 0, t_begin = 0;
    // Stage timers need a stream synchronisation per stage (~20-40 us of idle GPU each, ~0.3 ms per build): they run only
    // when asked for (ac_set_stage_timing); the event-timed insert kernel and total_device are always measured.
    void lap(double* acc) { if (!g_stage_timing) return; stream_sync(); double t = now_s(); *acc += t - t0; t0 = t; }

    DBuf<u32> counters;        // [1] insert err, [3] link err, [4] path err, [5] self-mirror links, [6] fragment err, [7] pool overflow
    // k-mer table and novel list of G
    DBuf<u64> slots; u64 cap = 0; u64 N = 0;
    DBuf<u64> bm; DBuf<u32> wprefix; DBuf<u64> npos;
    // unitigs in seed order
    u32 U = 0;
    DBuf<u32> kinfo, head, scan, ustart, order, rank, ulen;
    DBuf<u64> ustartpos, useq_off; DBuf<u8> uorient;
    DBuf<int32_t> links; DBuf<u64> wlinks;
    // per-occurrence quantities from the walk over loc
    DBuf<u32> depth, minpos_fwd, minpos_rev; DBuf<u64> path_off; DBuf<int32_t> ent_val; u64 n_ent = 0;
    DBuf<u8> fs0, fe0;
    // single-device builds: smallest positions beyond it are kept as a lower bound only (kernels_tail.inc exp_avoid_start_of_path); all ones = exact
    u32 pos_cap_now = 0xFFFFFFFFu; bool exact_positions = false;
    bool host_remap_allowed = false;      // GraphBuilder::build, and a rank of a sharded build that keeps its own paths: the result block of the paths is this build's own
    bool paths_in_seed_numbers = false;   // the tail left ent_val in seed numbers (the host renumbered the copy it took)
    // single-device builds: the "group too large" flags of the seed sort and of the two renumberings are read with the build's LAST read-back
    // (sort_flags: [0] seed ties, [1] renumbering) and a build that had one set is repeated with every flag checked where it is raised
    bool checked_sorts = false; DBuf<u32> sort_flags;
    bool deferred_sort_checks() const { return host_remap_allowed_build && !checked_sorts && sort_checks_deferrable(); }
    bool host_remap_allowed_build = false;      // (GraphBuilder::build only: the one driver that can repeat a build)
    DBuf<u8> maybe_dest; bool maybe_dest_valid = false;      // (unitig, side) that may become an expand_repeats destination (walk's position filter)
    // fragments of a sharded build
    DBuf<u8> frag_text; DBuf<u64> frag_meta, frag_fpos, frag_boff; u64 frag_bytes = 0, n_frags = 0;      // (frag_text: only when someone asks for bytes)
    u64 distinct_upper = 0;    // sharded builds: sum of the ranks' local distinct counts (0 = unknown)

    void begin(BuildTimings* t) {
        tm = t; t_begin = t0 = now_s();
        rt_counters() = RtCounters();
        host_remap_allowed = false;      // (GraphBuilder::build switches it on for itself)
        counters.alloc(8); counters.fill_bytes(0);
        sort_flags.alloc(2); sort_flags.fill_bytes(0);
    }
    void check_sizes(const PackedText& t) const {
        if (t.n_text >= POS_MASK) throw DeviceError("input too large for 40-bit text positions");
        if (t.n_seqs == 0 || t.n_text < (u64)k + 2) throw DeviceError("no sequences");
    }
    template <int W> void insert(const PackedText& t, u32 hint, DBuf<u64>* slots_out, u64* cap_out, u64* n_distinct_out, DBuf<u64>* bm_out, bool want_sib = false);
    DBuf<u64> sflags;          // sibling bits per slot, written by the insert (sib_note); empty = not collected
    DBuf<u64> runs; DBuf<u32> run_count; u64 run_rows = 0, run_rows_cap = 0;      // the insert's followed runs (Table::runs: rows used / reserved), for the copying path walk
    // Sharded builds (round 5): the runs are those of the LOCAL insert, checked against the rank's own novel bitmap (a run's source must be
    // a first occurrence within this rank's text: then it lies in walked text); loc_bm / loc_wprefix = that bitmap with rank support
    DBuf<u64> loc_bm; DBuf<u32> loc_wprefix; bool local_insert_of_shard = false;
    // The plan of a copying walk (walk_copy_prepare): the usable pieces, the gaps between them cut into walkers.  A sharded build makes it
    // before the walk-start keys go to their owners (the walkers ARE the gap walkers then) and walks when the answers are back.
    struct CopyPlan { bool ok = false; u64 R = 0, NW = 0, Rb = 0; DBuf<RunRec> rr; DBuf<u32> rseq; DBuf<u64> wfirst, w_begin, w_end; DBuf<u32> w_gap; } cplan;
    void occupancy_bitmap(const DBuf<u64>& sl, u64 c, DBuf<u64>* occ_out, const u64* sflags_in, u64* sib_out);
    DBuf<u64> occ;             // slot-occupancy bitmap of the graph table
    DBuf<u64> sib;             // sibling bits of the graph table's real k-mers, two per text position (MarkFunctor); empty = not used
    DBuf<u64> sibn; bool sib_pending = false;      // sharded builds: the sibling bits by NOVEL INDEX (SibByRankFunctor) — this rank's, then the ranks' sum; pending = not summed yet
    // sharded builds with the light degree step: this rank's contributions to the k-mers that step left open, compact (degrees()):
    // [n_pending degree words | n_first first-flag words]; pend / pidx: which k-mers, and where in that array
    DBuf<u32> kcontrib, pend, pidx; u64 n_pending = 0, n_first = 0;
    DBuf<u64> endset, endset_bloom; u64 endset_mask = 0;      // sequence-end set (EndSetFunctor) and its two filters
    u32 n_owners = 1, my_owner = 0;      // sharded builds: which slice of the key space the graph table holds (§7)
    // host entry: stream 0 only waited for the FIRST chunk of the packed upload; positions below upload_avail are on the device, the
    // rest arrives while the first insert phases run (upload_done = the event behind the last chunk)
    void* upload_done = nullptr; u64 upload_avail = 0; bool upload_pending = false;
    // host entry, packed upload: the chunks are packed and sent by background threads while this thread already issues the insert
    // phases — each phase first waits (host: until the copy of the chunks it reads has been ISSUED; stream 0: until it has LANDED).
    struct UploadJob {
#ifndef AC_EMU
        const std::vector<SeqView>* seqs = nullptr;      // the caller's views: valid until the build has taken the last chunk
        std::vector<uint64_t> off;
        uint32_t k = 0; u64 n = 0, CH = 0, SUB = 0, n_chunks = 0, slot_bytes = 0; int NSLOT = 0, dev = 0;
        hipStream_t up = nullptr, pk = nullptr;
        u64* d_bits = nullptr;
        bool direct = false;                                          // the packers store into d_bits themselves (no ring, no copies, no `landed` events)
        double t_start = 0; std::atomic<u64> chunks_issued{0}; std::atomic<double> t_last{0};      // direct: host clock from the first store to the last flush
        hipEvent_t fills_done = nullptr;                              // direct: the device has set the slack behind the last group (the last work item waits for it)
        std::atomic<u64> next{0}, nonbase{0}, bar_sink{0}; u64 expected_nonbase = 0;      // alphabet check: non-base bytes the packers met / the sequence table promises
        std::vector<std::atomic<u32>> done, slot_state, issued;      // slot_state: 0 untouched, 1 someone is waiting for the slot, 2 free
        std::vector<hipEvent_t> landed;                               // per chunk: both of its copies are on the device
        std::mutex hip_mu; std::string fail; std::atomic<bool> stop{false};
        u64 ticket = 0;                                               // UploadPool: which run of the pool this job is
        void* stager = nullptr;                                       // the HostStager of the context that started the job (the pool's threads have no context of their own)
        u64 next_wait = 0;                                            // chunks stream 0 already waits for
        void run();
#endif
    };
    UploadJob* job = nullptr;
    u64 upload_rest_limit(u64 pb) const;      // where a piece of the one-launch rest that starts at pb may end so that one more chunk suffices
    void need_text(u64 upto);      // everything below text position `upto` is on the device before whatever stream 0 gets next
    void finish_upload();          // joins the uploaders (idempotent); throws what they threw
    ~Impl();
    // K1 of the device entry in two launches: the head of the text — what the first insert phase reads — on stream 0, the rest on the
    // side stream, under that first phase (a bandwidth-bound pack next to a CAS-bound insert); the insert waits for the rest before
    // its second phase, through the same hook as the host entry's chunked upload.
    void pack_overlapped(u32 hint) {
        PackedText& pt = loc;
        if (pt.packed) return;
#ifndef AC_EMU
        const u64 p_end_all = pt.n_text - (u64)k + 1;
        const u64 first = std::max<u64>(p_end_all / std::max<u32>(hint, 1), 1u << 16);
        const u64 H = (first + (u64)k + 8192 + 4095) & ~4095ULL;      // (Impl::insert: a phase that ends at pe reads below pe + k + 8192)
        // only next to a first phase that claims into a cache-sized table (config C: 5.15 -> 5.08 ms): where the table is far larger
        // (config D: 4 GB) that phase is bound by HBM lines itself and the pack beside it costs more than it hides (28.96 -> 29.13)
        const u64 cap_est = next_pow2(std::max<u64>(1024, pt.n_bases / std::max<u32>(hint, 1) * 3 + 4096));
        if (cap_est <= (1ULL << 25) && H + (1u << 22) < pt.n_text) {
            pt.pack_alloc();
            launch(H / 32, PackFunctor{pt.d_text, pt.n_text, pt.bits.ptr(), (u32*)pt.mask.ptr(), 0, pt.chk()});
            SideStream& side = SideStream::get();
            side.after_main();      // (the fills of bits / mask went out with the head's launch)
            launch((pt.n_text + 31) / 32 - H / 32, PackFunctor{pt.d_text, pt.n_text, pt.bits.ptr(), (u32*)pt.mask.ptr(), H / 32, pt.chk()}, side.stream());
            upload_done = side.mark(); upload_avail = H; upload_pending = true;
            pt.packed = true;
            return;
        }
#endif
        pt.pack();
    }
    Table graph_table() const { return Table{const_cast<u64*>(slots.ptr()), cap - 1, occ.ptr(), nullptr, n_owners, my_owner, nullptr, nullptr, nullptr, nullptr, 0}; }
    template <int W> void fragments();
    template <int W> void table();                      // K2, K3 on G
    void novel_list(u64 known_n);
    template <int W> void degrees();                    // K5, K6 for all novel k-mers
    template <int W> void walk_queries();               // sharded: the keys this rank's walkers start from
    template <int W> void answer_queries(const u64* d_keys, u64 n, u64* d_out);   // sharded: the owned ones, looked up in this rank's table
    DBuf<u64> qkeys; u64 n_queries = 0;
    DBuf<u32> qidx;                                     // routed position -> query (queries_route)
    DBuf<u64> qanswers;                                 // the answers in query order (answers_unroute)
    template <int W> void route_queries(u32 n_shards, u64* d_routed_keys, u64* counts_host);
    const u64* walk_answers = nullptr;                  // sharded: [n_walkers | n_seqs] answers (0 = not found), nullptr = look the table up
    template <int W> void unitigs();                    // K6..K11 on G
    template <int W> void walk();
    template <int W> bool walk_copy_prepare(u32 PC, const Novel& nv_text);   // K10c, first half: false = not worth it (or not possible) for this text, nothing kept
    template <int W> void walk_copy_finish(u32 PC);                          // K10c, second half: the walk over the gaps and the copies
    template <int W> void tail(FinalGraph* out, bool want_graph, bool want_paths);
    // sharded builds: in-place all-reduce of a device buffer over the ranks (dtype 0 = uint8, 1 = int32; op 0 = SUM, 1 = MIN), given by
    // whoever drives the ranks.  With it the tail runs expand_repeats on this rank's share of the junctions only (conflict components,
    // kernels_tail.inc) and merges the sequences; without it every rank runs all of them.
    std::function<void(void*, uint64_t, int, int)> tail_xchg;
};

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
                if (getenv("AC_DEBUG_ARENA")) fprintf(stderr, "insert: second stretch %.4f new, one-launch rest %d, copying walk %d (mode %d)\n", r2, (int)rest_at_once, (int)want_runs, copy_mode);
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

// Slot-occupancy bitmap of a finished table (one ballot word per wavefront of the scan; no atomics).
inline void GraphBuilder::Impl::occupancy_bitmap(const DBuf<u64>& sl, u64 c, DBuf<u64>* occ_out, const u64* sflags_in, u64* sib_out) {
    occ_out->alloc((c + 63) / 64);
#ifdef AC_EMU
    occ_out->fill_bytes(0);      // the serial emulation ORs bit by bit; the device writes whole ballot words
#endif
    launch_full(c, MarkFunctor{sl.ptr(), occ_out->ptr(), sflags_in, sib_out});
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
    occupancy_bitmap(slots, cap, &occ, sib_by_pos ? sflags.ptr() : nullptr, sib_by_pos ? sib.ptr() : nullptr);
    if (n_owners <= 1) novel_list(N);      // a sharded build first sums the ranks' (disjoint) bitmaps: bitmap_import
}
// K3: novel-position bitmap -> sorted novel list + rank support.  known_n = the number of set bits if the caller knows it (the
// single-device insert counted its claims), 0 = count them here.
inline void GraphBuilder::Impl::novel_list(u64 known_n) {
    PackedText& g = *G;
    u64 n_bm_words = g.n_text / 64 + 1;
    DBuf<u32> wcnt(n_bm_words);
    wprefix.alloc(n_bm_words);
    launch(n_bm_words, PopcFunctor{bm.ptr(), wcnt.ptr()});
    exclusive_scan_u32(wcnt.ptr(), wprefix.ptr(), n_bm_words);
    if (known_n) N = known_n;
    else {
        u32 last[2];
        ReadBatch rb;
        rb.add(&last[0], wprefix.ptr() + (n_bm_words - 1), 4);
        rb.add(&last[1], wcnt.ptr() + (n_bm_words - 1), 4);
        rb.run();
        N = (u64)last[0] + last[1];
        if (N == 0) throw DeviceError("internal error: no k-mers in the union of the shards");
        tm->n_distinct = N;
    }
    npos.alloc(N);
    launch_wave_kernel(fill_novel_wave_kernel<0>, (n_bm_words + 255) / 256, 0, (const u64*)bm.ptr(), (const u32*)wprefix.ptr(), npos.ptr(), n_bm_words);
    kinfo.alloc(N, true);
    lap(&tm->collect_sort);
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
        launch(2 * (u64)g.n_seqs, EndSetFunctor<W>{g.ctx((int)k), es});
    }
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
        if (getenv("AC_DEGREE_DIAG")) {
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
    } else
        launch(N, DegreeFunctor<W>{g.ctx((int)k), tb, npos.ptr(), kinfo.ptr(), g.any_dots, 0, bm.ptr(), sib.size() && n_owners <= 1 ? sib.ptr() : nullptr, es});
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
    head.alloc(N + 1); scan.alloc(N + 1);      // (HeadFunctor and the scan write entries 0 .. N - 1)
    head.fill_bytes_from(N * 4, 0); scan.fill_bytes_from(N * 4, 0);
    launch(N, HeadFunctor{npos.ptr(), kinfo.ptr(), head.ptr(), N});
    inclusive_scan_u32(head.ptr(), scan.ptr(), N);
    U = read_scalar(scan.ptr() + (N - 1));
    ustart.alloc((u64)U + 1);
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
    launch(U, IotaFunctor{order.ptr()});
    bool seeds_ordered = false;
    if (seed_prefix_sort()) {      // one sort on a 64-bit prefix of the seed keys, ties on full keys: any key width, any number of unitigs
        DBuf<u64> wkey(U);
        // as many leading bits of the prefix as tell U seeds apart with a few ties to spare (twice log2 U, and a byte for the bias of a
        // MINIMUM towards small values): the ties are ranked on full keys anyway (SeedTieFunctor), and every digit less is a pass less
        int keep = seed_prefix_bits();
        if (keep <= 0) { int lg = 1; while ((1ULL << lg) < (u64)U) lg++; keep = std::min(64, ((2 * lg + 8 + 7) / 8) * 8); }
        DBuf<u32> by_prefix(U);
        launch(U, SeedPrefixFunctor<W>{umin.ptr(), (int)k, wkey.ptr(), keep, by_prefix.ptr()});      // (... and the identity the sort permutes)
        sort_pairs_u64_u32(wkey, by_prefix, U, 64, 0, 64 - keep);
        DBuf<u32> settled(U), big(1, true);
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
        DBuf<u32> wcnt(nw);
        loc_wprefix.alloc(nw);
        launch(nw, PopcFunctor{loc_bm.ptr(), wcnt.ptr()});
        exclusive_scan_u32(wcnt.ptr(), loc_wprefix.ptr(), nw);
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
    if (getenv("AC_DEBUG_ARENA")) fprintf(stderr, "path copy: %llu runs on the list, %llu pieces usable, covering %llu of %llu positions, %llu walkers\n", (unsigned long long)R0, (unsigned long long)R, (unsigned long long)h_cov, (unsigned long long)loc.n_text, (unsigned long long)NW);
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
    launch((u64)U * 10, WlinkFlagFunctor{filter ? maybe_dest.ptr() : nullptr, wlinks.ptr(), counters.ptr() + 4});
    DBuf<V16> uinfo(U);
    launch(U, WalkInfoFunctor{uc, filter ? maybe_dest.ptr() : nullptr, uinfo.ptr()});
    launch(NW, PathWalkFunctor<W>{t, g, tb, nv, uc, uinfo.ptr(), wlinks.ptr(), PC, stage.ptr(), wcount.ptr(), seq_tid.ptr(), seq_j.ptr(),
                                 depth.ptr(), minpos_fwd.ptr(), minpos_rev.ptr(), counters.ptr() + 4, filter ? maybe_dest.ptr() : nullptr,
                                 pos_cap_now, 0, walk_answers, NW, c.w_begin.ptr(), c.w_end.ptr(), stage_off.ptr()});
    exclusive_scan_u64(wcount.ptr(), woff.ptr(), NW + 1);
    const u64 NE = read_scalar(woff.ptr() + NW);      // walked entries
    DBuf<int32_t> ent(NE); DBuf<u64> ent_pos(NE), ent_end(NE); DBuf<u8> ent_want(NE); DBuf<u32> ent_gap(NE);
    launch_full(NW, WalkCompactFunctor{stage.ptr(), stage_off.ptr(), wcount.ptr(), woff.ptr(), c.w_begin.ptr(), c.w_gap.ptr(), PC, NW, ulen.ptr(),
                                       filter ? maybe_dest.ptr() : nullptr, ent.ptr(), ent_pos.ptr(), ent_end.ptr(), ent_want.ptr(), ent_gap.ptr()});
    // what every run copies; entries per segment; the final array
    DBuf<u64> ra(R), rcnt(R + 1), seg(2 * R + 2), segoff(2 * R + 2); DBuf<u32> cov(NE + 1), copies(NE + 1);
    cov.fill_bytes(0);
    launch(R, RunRangeFunctor{c.rr.ptr(), ent_pos.ptr(), ent_end.ptr(), NE, ra.ptr(), rcnt.ptr(), cov.ptr()});
    launch(2 * R + 2, SegCountFunctor{c.wfirst.ptr(), woff.ptr(), rcnt.ptr(), R, seg.ptr()});
    exclusive_scan_u64(seg.ptr(), segoff.ptr(), 2 * R + 2);
    inclusive_scan_u32(cov.ptr(), copies.ptr(), NE + 1);
    n_ent = read_scalar(segoff.ptr() + (2 * R + 1));
    if (getenv("AC_DEBUG_ARENA")) {
        std::vector<u64> h = to_host(rcnt, R); u64 mx = 0, sum = 0, big = 0, hist[8] = {0};
        for (u64 v : h) { mx = std::max(mx, v); sum += v; if (v > 256) big++; int b = 0; while ((32ull << b) < v && b < 7) b++; hist[b]++; }
        fprintf(stderr, "path copy: %llu pieces, entries per piece: max %llu, mean %.1f, %llu above 256; hist(<=32,64,128,..): %llu %llu %llu %llu %llu %llu %llu %llu\n", (unsigned long long)R, (unsigned long long)mx, (double)sum / (double)R, (unsigned long long)big,
                (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)hist[6], (unsigned long long)hist[7]);
    }
    // (the scratch above stays where it is until the build ends: for a text this redundant it is a fraction of the text's size)
    ent_val.alloc(n_ent);
    int32_t* const out_ptr = ent_val.ptr();
    launch(NE, GapOutFunctor{ent.ptr(), ent_gap.ptr(), woff.ptr(), c.wfirst.ptr(), segoff.ptr(), copies.ptr(), out_ptr, depth.ptr()});
    launch_full(R * 32, RunOutFunctor<32, 1>{c.rr.ptr(), R, ra.ptr(), rcnt.ptr(), segoff.ptr(), ent.ptr(), ent_pos.ptr(), ent_end.ptr(), ulen.ptr(),
                                             ent_want.ptr(), t, c.rseq.ptr(), minpos_fwd.ptr(), minpos_rev.ptr(), out_ptr, pos_cap_now});
    launch(loc.n_seqs, PathOffCopyFunctor{seq_tid.ptr(), seq_j.ptr(), woff.ptr(), c.wfirst.ptr(), c.w_gap.ptr(), segoff.ptr(), path_off.ptr()});
    tm->n_path_entries = n_ent;
    tm->path_runs_copied = R; tm->path_entries_walked = NE;
    copy_h2d(path_off.ptr() + loc.n_seqs, &n_ent, 8);
    launch(loc.n_seqs, PathEndsFunctor{ent_val.ptr(), path_off.ptr(), fs0.ptr(), fe0.ptr()});
}

// K10 paths of this rank's sequences against the graph: count, scan, write; first / last unitig of every path.
template <int W> void GraphBuilder::Impl::walk() {
    TextCtx t = loc.ctx((int)k), g = G->ctx((int)k);
    Table tb = graph_table();
    Novel nv{bm.ptr(), wprefix.ptr()};
    UnitigCtx uc{head.ptr(), scan.ptr(), rank.ptr(), uorient.ptr(), ustart.ptr(), ulen.ptr(), U, N};
    const u32 PC = path_chunk(N, U);
    u64 n_walkers = (loc.n_text + PC - 1) / PC;
    depth.alloc(U, true); minpos_fwd.alloc(U); minpos_rev.alloc(U);
    // (sharded builds keep exact positions: a repeat of the build would have to be agreed between the ranks)
    pos_cap_now = (exact_positions || walk_answers || n_owners > 1 || G != &loc || pos_cap() == 0) ? 0xFFFFFFFFu : pos_cap();
    if (pos_cap_now == 0xFFFFFFFFu) { minpos_fwd.fill_bytes(0xFF); minpos_rev.fill_bytes(0xFF); }
    else {
        launch(U, FillU32PairFunctor{minpos_fwd.ptr(), minpos_rev.ptr(), (pos_cap_now + 1) | POS_BOUND});
    }
    path_off.alloc((u64)loc.n_seqs + 1);
    const bool filter = path_filter();
    maybe_dest_valid = filter;
    if (filter) { maybe_dest.alloc((u64)U * 2); launch((u64)U * 2, MaybeDestFunctor{links.ptr(), maybe_dest.ptr()}); }
    fs0.alloc(U, true); fe0.alloc(U, true);
    if (walk_answers) {      // a sharded build planned (or not) before the walk-start keys went out (walk_queries)
        if (cplan.ok) { walk_copy_finish<W>(PC); lap(&tm->paths); return; }
    } else if (run_rows && n_owners <= 1 && G == &loc && PC <= 65535 && walk_copy_prepare<W>(PC, nv)) { walk_copy_finish<W>(PC); lap(&tm->paths); return; }
    // everything from here to the compaction is the walk's own: 4 bytes of staging per text position (configs[4]: 20 GB) go back to
    // the arena once the entries are compacted — they are compacted into the staging area's own first bytes
    const Arena::Mark walk_mark = Arena::device().mark();
    DBuf<int32_t> stage(((n_walkers + 63) / 64) * 64 * PC);
    DBuf<u64> wcount(n_walkers + 1), woff(n_walkers + 1);
    DBuf<u32> seq_tid(loc.n_seqs), seq_j(loc.n_seqs);
    wcount.fill_bytes(0);
    launch((u64)U * 10, WlinkFlagFunctor{filter ? maybe_dest.ptr() : nullptr, wlinks.ptr(), counters.ptr() + 4});
    DBuf<V16> uinfo(U);
    launch(U, WalkInfoFunctor{uc, filter ? maybe_dest.ptr() : nullptr, uinfo.ptr()});
    launch(n_walkers, PathWalkFunctor<W>{t, g, tb, nv, uc, uinfo.ptr(), wlinks.ptr(), PC, stage.ptr(), wcount.ptr(), seq_tid.ptr(), seq_j.ptr(),
                                        depth.ptr(), minpos_fwd.ptr(), minpos_rev.ptr(), counters.ptr() + 4, filter ? maybe_dest.ptr() : nullptr,
                                        pos_cap_now, path_diag(), walk_answers, n_walkers});
    exclusive_scan_u64(wcount.ptr(), woff.ptr(), n_walkers + 1);
    n_ent = read_scalar(woff.ptr() + n_walkers);
    launch(loc.n_seqs, PathOffFunctor{seq_tid.ptr(), seq_j.ptr(), woff.ptr(), path_off.ptr()});
    tm->n_path_entries = n_ent;
    copy_h2d(path_off.ptr() + loc.n_seqs, &n_ent, 8);
    {
        DBuf<int32_t> packed(n_ent);      // (beyond the staging area: the compaction reads rows that later wavefronts' outputs would overwrite)
        launch_full(((n_walkers + 63) / 64) * 64, PathCompactFunctor{stage.ptr(), wcount.ptr(), woff.ptr(), PC, n_walkers, packed.ptr()});
        stage = DBuf<int32_t>(); wcount = DBuf<u64>(); woff = DBuf<u64>(); seq_tid = DBuf<u32>(); seq_j = DBuf<u32>(); uinfo = DBuf<V16>();
        Arena::device().rewind(walk_mark);
        ent_val.alloc(n_ent);             // where the staging area began; `packed` lies behind the staging area's end (n_ent <= its size)
        if (ent_val.ptr() != packed.ptr()) copy_d2d(ent_val.ptr(), packed.ptr(), n_ent * 4);
    }
    launch(loc.n_seqs, PathEndsFunctor{ent_val.ptr(), path_off.ptr(), fs0.ptr(), fe0.ptr()});
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
    paths_in_seed_numbers = host_remap;
    if (host_remap) {
        out->path_block = PinnedPool::get().alloc(n_ent * 4);
        side.after_main();
        copy_d2h_async(out->path_block.p, ent_val.ptr(), n_ent * 4, side.stream());
    }
    // K12 sequences
    u64 total = N;   // sum of unitig lengths == number of distinct canonical k-mers
    DBuf<u8> useq(total);
    // (both sequence writers: one thread per 64 output bytes; on the device through the block index + LDS tile of seq_write_kernel)
    auto write_seqs = [&](int mode, const ExpState* es, const u64* off, u64 n_bytes, u8* dst) {
        // the indexed / LDS-tiled writer pays for its index (four small launches) from ~16 MB of output on: config C (7.8 MB) 6.02 vs
        // 5.97 ms per build with it, E' (45 MB) 24.9 vs 26.6, config D (126 MB): see DESIGN.md §6
        if (seq_writer_plain() || (n_bytes < ((u64)16 << 20) && !seq_writer_forced())) {
            const u32 per = 16;      // output bytes per thread (r06n: 64 left most of the chip idle on 7.8 MB)
            if (mode == 0) launch((n_bytes + per - 1) / per, SeqFunctor{g.bits.ptr(), off, ustartpos.ptr(), ulen.ptr(), uorient.ptr(), U, n_bytes, (int)(k / 2), dst, per});
            else launch((n_bytes + per - 1) / per, MaterializeFunctor{*es, off, U, n_bytes, dst, per});
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
    write_seqs(0, nullptr, useq_off.ptr(), total, useq.ptr());
    lap(&tm->seqs);

    // K13 link push order, K14 static analysis for expand_repeats, K15 first renumber_unitigs
    DBuf<int32_t> lord((u64)U * 10); DBuf<u8> lcnt((u64)U * 2);
    launch((u64)U * 2, LinkOrderFunctor{links.ptr(), lord.ptr(), lcnt.ptr(), counters.ptr() + 5});
    OrderedLinks L{lord.ptr(), lcnt.ptr()};
    DBuf<u8> fixed_start(U, true), fixed_end(U, true), cand((u64)U * 2);
    launch(U, FixedSpreadFunctor{fs0.ptr(), fe0.ptr(), L, fixed_start.ptr(), fixed_end.ptr()});
    launch((u64)U * 2, CandFunctor{L, fixed_start.ptr(), fixed_end.ptr(), cand.ptr()});
    if (maybe_dest_valid)      // the walk only collected smallest positions where maybe_dest says so: every real candidate must be covered
        launch((u64)U * 2, CandCoveredFunctor{cand.ptr(), maybe_dest.ptr(), counters.ptr() + 4});
    DBuf<u32> order1(U);
    launch(U, IotaFunctor{order1.ptr()});
    DBuf<u32> renum_flag(1, true);
    const bool defer_sorts = deferred_sort_checks();
    renumber_sort(order1, U, ulen.ptr(), useq_off.ptr(), useq.ptr(), depth.ptr(), defer_sorts ? sort_flags.ptr() + 1 : renum_flag.ptr(), defer_sorts);
    lap(&tm->analysis);

    // K17 expand_repeats, level-scheduled (see the kernels)
    DBuf<u64> coff(U), len64((u64)U + 1), noff((u64)U + 1);
    DBuf<u32> clen(U); DBuf<ExpU> ev(U);      // the views of expand_repeats (one 32-byte record per unitig); coff / clen: offsets and lengths as plain arrays for what follows
    DBuf<u8> seq_alt(total), pool(std::min<u64>(8 * total + (1u << 20), 0xFFFFFFF0ULL)), dirty((u64)U * 2);
    DBuf<u64> shifted(1); DBuf<u32> pool_used(EXP_SUBPOOLS + 1);
    launch(U, ExpInitFunctor{useq_off.ptr(), ulen.ptr(), cand.ptr(), ev.ptr(), coff.ptr(), clen.ptr(), dirty.ptr()});      // core views = the unitigs, dirty = the candidates (three copies, one launch)
    u8* cur = useq.ptr(); u8* alt = seq_alt.ptr();
    u64 final_total = total;
    int passes = 0;
    u32 n_cand = 0, n_levels = 0;
    const bool partitioned = n_owners > 1 && (bool)tail_xchg;      // (decided by the driver: the same on every rank)
    DBuf<u8> jowner; DBuf<u32> owned_count, gpre, gpost;
    u32 n_cand_owned = 0;
    {
        u64 J = (u64)U * 2;
        DBuf<u32> cflag(J + 1), cpos(J + 1), prio(J);
        cflag.fill_bytes(0);       // [J] = 0: the exclusive scan then ends with the total
        launch(J, CandFlagFunctor{order1.ptr(), cand.ptr(), cflag.ptr()});
        exclusive_scan_u32(cflag.ptr(), cpos.ptr(), J + 1);
        n_cand = read_scalar(cpos.ptr() + J);
        if (n_cand == 0) {
            passes = 1;   // the reference's single pass that moves nothing (the same on every rank of a sharded build: nothing to merge)
        } else {
            u64 C = n_cand;
            DBuf<u32> clist(C), level(C);
            prio.fill_bytes(0xFF);
            launch(J, CandListFunctor{order1.ptr(), cflag.ptr(), cpos.ptr(), clist.ptr(), prio.ptr()});
            launch(C, FillU32Functor{level.ptr(), 1u});
            DBuf<u32> changed(9), preds(C * MAX_PREDS); DBuf<u8> npred(C);
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
            for (;;) {   // longest-path levels of the conflict DAG, settled front to back (LevelRelaxFunctor); eight sweeps per host
                changed.fill_bytes(0);       // check, done when the last of them left no candidate open (a sweep settles one more level)
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
            sort_pairs_u64_u32(lkey, clist, C, level_bits);      // (the highest level came back with the convergence flags: one or two digits)
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
            pool_used.fill_bytes(0);
            u64 moved = 0, moved_since_rewrite = 0;
            DBuf<u64> shifted2(2);
            // Rewrites the sequences contiguously (gained pieces folded into the core views) and empties the pool.  Once after the
            // last pass — and in between whenever the pool is a quarter full: a side that gains again gets a new piece holding its
            // old one as well, so without this the pool use of a many-pass input grows with the square of the passes.
            auto rewrite = [&] {
                if (partitioned) launch(U, ExpFoldFunctor{e, gpre.ptr(), gpost.ptr()});      // (what the fold makes of the gained pieces: the merge below)
                launch((u64)U + 1, ExpLenFunctor{e, len64.ptr(), U});
                exclusive_scan_u64(len64.ptr(), noff.ptr(), (u64)U + 1);
                final_total = read_scalar(noff.ptr() + U);
                write_seqs(1, &e, noff.ptr(), final_total, alt);
                launch(U, ExpResetFunctor{e, noff.ptr(), coff.ptr(), clen.ptr()});
                std::swap(cur, alt);
                e.cur = cur;
                pool_used.fill_bytes(0);
                moved_since_rewrite = 0;
            };
            const u32 sub_limit = (u32)(pool.size() / 2 / EXP_SUBPOOLS / 2);      // a region half full (or anything in the overflow half) asks for a rewrite
            auto run_level = [&](u32 lv) {
                const u64 cnt = (u64)(hb[lv + 1] - hb[lv]);
                // sixteen lanes per junction, four junctions per wavefront (expand_wave_kernel; the emulation runs the same kernel in
                // lockstep, wave_rt.hpp).  A thread per junction and 8 / 32 / 64 lanes were measured and retired (r06u/v: G = 16
                // wins from config C to mixed-species graphs)
                if (cnt) launch_wave_kernel(expand_wave_kernel<W, 16>, (cnt * 16 + 255) / 256, 0, e, (const u32*)clist.ptr(), (u64)hb[lv], cnt, (u32)pool.size(), counters.ptr() + 7);
            };
            for (;;) {   // two passes per host check: if the first moved nothing the second is an (uncounted) no-op
                shifted2.fill_bytes(0);
                for (int half = 0; half < 2; half++) {
                    e.shifted = shifted2.ptr() + half;
                    for (u32 lv = 1; lv <= n_levels; lv++) run_level(lv);
                }
                u64 sh[2]; u32 used = 0;
                {
                    std::vector<u32> pu(EXP_SUBPOOLS + 1);
                    ReadBatch rb;
                    rb.add(sh, shifted2.ptr(), 16);
                    rb.add(pu.data(), pool_used.ptr(), (EXP_SUBPOOLS + 1) * 4);
                    rb.run();
                    for (u32 q = 0; q < EXP_SUBPOOLS; q++) used = std::max(used, pu[q]);
                    if (pu[EXP_SUBPOOLS]) used = 0xFFFFFFFFu;
                }
                moved += sh[0] + sh[1]; moved_since_rewrite += sh[0] + sh[1];
                if (sh[0] == 0) { passes += 1; break; }
                passes += 2;
                if (sh[1] == 0) break;
                if (used > sub_limit || expand_rewrite_always()) rewrite();
            }
            if (!partitioned) { if (moved_since_rewrite) rewrite(); }
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
    tm->n_candidates_owned = partitioned && n_cand ? n_cand_owned : n_cand;
    lap(&tm->expand);

    // K15b second renumber_unitigs (graph_simplification.rs:39): a stable sort of the CURRENT order on the new
    // sequences; K16 per-unitig outputs in final order, links in get_links_for_gfa order, paths in final numbers
    // D2H on a second stream, each array as soon as it is final, straight into pinned blocks owned by the result; the
    // paths go in four chunks, each copied while the next is still being renumbered.
    if (want_graph) {
        out->seq_block = PinnedPool::get().alloc(final_total);
        side.after_main();     // sequences are final since the materialise step: their copy runs under the second renumbering
        copy_d2h_async(out->seq_block.p, cur, final_total, side.stream());
    }
    DBuf<u32> order2(U);
    copy_d2d(order2.ptr(), order1.ptr(), (size_t)U * 4);
    renumber_sort(order2, U, clen.ptr(), coff.ptr(), cur, depth.ptr(), defer_sorts ? sort_flags.ptr() + 1 : renum_flag.ptr(), defer_sorts);
    DBuf<u64> number_len(U), lcount((u64)U + 1), loff((u64)U + 1);
    DBuf<u32> number_only(host_remap ? U : 0);
    DBuf<u8> meta((size_t)U * 24);
    u64* d_seq_begin = (u64*)meta.ptr();
    double* d_depth = (double*)(meta.ptr() + (size_t)U * 8);
    u32* d_seq_len = (u32*)(meta.ptr() + (size_t)U * 16);
    u32* d_seed_index = (u32*)(meta.ptr() + (size_t)U * 20);
    lcount.fill_bytes(0);
    out->k = k;
    out->n_kmers = 2 * (u64)N;
    out->n_unitigs = U;
    launch(U, FinalMetaFunctor{order2.ptr(), coff.ptr(), clen.ptr(), depth.ptr(), lcnt.ptr(), number_len.ptr(), d_seq_begin, d_depth,
                               d_seq_len, lcount.ptr(), host_remap ? number_only.ptr() : nullptr, d_seed_index});
    if (host_remap) {      // the number table first: the host threads start on the entries while the rest is still crossing
        number_block = PinnedPool::get().alloc((size_t)U * 4);
        side.after_main();
        copy_d2h_async(number_block.p, number_only.ptr(), (size_t)U * 4, side.stream());
        remap_job.path = (int32_t*)out->path_block.p; remap_job.n_ent = n_ent;
        remap_job.number = (const u32*)number_block.p; remap_job.n_unitigs = U;
        remap_job.landed = side.mark();
#ifndef AC_EMU
        AC_HIP_CHECK(hipGetDevice(&remap_job.dev));
        path_remap_start(remap_job, (int)upload_threads());
#endif
    }
    if (want_graph) {
        out->meta_block = PinnedPool::get().alloc((size_t)U * 24);
        side.after_main();
        copy_d2h_async(out->meta_block.p, meta.ptr(), (size_t)U * 24, side.stream());
    }
    exclusive_scan_u64(lcount.ptr(), loff.ptr(), (u64)U + 1);
    u64 n_links = read_scalar(loff.ptr() + U);
    DBuf<Link> links_out(n_links);
    launch(U, LinkOutFunctor{order2.ptr(), L, number_len.ptr(), loff.ptr(), links_out.ptr()});
    if (want_graph) {
        out->links_block = PinnedPool::get().alloc(n_links * sizeof(Link));
        side.after_main();
        copy_d2h_async(out->links_block.p, links_out.ptr(), n_links * sizeof(Link), side.stream());
    }
    DBuf<u64> sums(n_seqs);
    sums.fill_bytes(0);
    if (host_remap) {      // the device only checks that every path spells its sequence's length (the sums), it stores nothing
        const u64 RB = remap_block();
        const u64 n_waves = (n_ent + RB - 1) / RB;
        launch_full(n_waves * 64, RemapFunctor{ent_val.ptr(), number_len.ptr(), path_off.ptr(), n_seqs, n_ent, sums.ptr(), 0, (u32)RB, nullptr, false});
    } else {
        if (want_paths) out->path_block = PinnedPool::get().alloc(n_ent * 4);
        const u64 RB = remap_block();
        const u64 n_waves = (n_ent + RB - 1) / RB;
        // Four chunks, each copied while the next is renumbered (the kernel storing straight into the pinned block measured equal, r08j:
        // either way the 4 bytes per entry cross PCIe after the final numbering exists — 42 MB = 0.7 ms on config C)
        const u64 per_chunk = std::max<u64>((n_waves + 3) / 4, 64);
        for (u64 w = 0; w < n_waves; w += per_chunk) {
            u64 cnt = std::min<u64>(per_chunk, n_waves - w);
            launch_full(cnt * 64, RemapFunctor{ent_val.ptr(), number_len.ptr(), path_off.ptr(), n_seqs, n_ent, sums.ptr(), w, (u32)RB, nullptr, true});
            if (want_paths) {
                u64 b = w * RB, e2 = std::min<u64>((w + cnt) * RB, n_ent);
                side.after_main();
                copy_d2h_async((int32_t*)out->path_block.p + b, ent_val.ptr() + b, (e2 - b) * 4, side.stream());
            }
        }
    }
    lap(&tm->finalize);

    std::vector<u64> h_sums(n_seqs);
    out->path_off.resize((size_t)n_seqs + 1);
    std::vector<u32> errs(8);
    u32 pack_bad[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    u32 h_sort_flags[2] = {0, 0};
    {
        ReadBatch rb;
        rb.add(h_sums.data(), sums.ptr(), (size_t)n_seqs * 8);
        rb.add(out->path_off.data(), path_off.ptr(), ((size_t)n_seqs + 1) * 8);
        rb.add(errs.data(), counters.ptr(), 8 * 4);
        rb.add(h_sort_flags, sort_flags.ptr(), 8);
        if (loc.check_alphabet && loc.pack_bad.size()) rb.add(pack_bad, loc.pack_bad.ptr(), 8);
        rb.run();                                   // synchronises stream 0 (once)
    }
    side.sync();                                    // ... and the copies: everything above has landed
    if (host_remap) {
#ifdef AC_EMU
        path_remap_range(remap_job.path, n_ent, remap_job.number, U, &remap_job.bad);
#endif
        path_remap_finish(remap_job);
    }
    if (loc.pack_bad.size()) loc.verify_alphabet(pack_bad);      // before any internal check: a text with foreign bytes explains them all
    if (errs[7] & 128u) throw NeedExactPositions();      // (before anything else: a repeat of the build settles it)
    if (h_sort_flags[0] || h_sort_flags[1]) {
        if (!deferred_sort_checks()) throw DeviceError("internal error: a sort flag was left set by a checked sort");
        throw NeedCheckedSorts();      // (the order the flagged sort left is a permutation, not THE order: everything behind it is void)
    }
    if (errs[7]) throw DeviceError("internal error: expand_repeats pool overflow");
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
    if (getenv("AC_DEBUG_ARENA"))
        fprintf(stderr, "arena: used %.1f MB (peak %.1f) of %.1f MB (n_text %.1f MB), %.3f s in hipMalloc / hipFree so far\n", Arena::device().total_used() / 1e6,
                Arena::device().peak() / 1e6, Arena::device().capacity() / 1e6, loc.n_text / 1e6, Arena::device().alloc_seconds());
}

// The width-dependent stages behind one explicitly instantiated type per width.  The main unit only sees declarations, so it
// cannot instantiate (or inline) anything width-dependent itself.
template <int W> struct Stages {
    static void table(GraphBuilder::Impl& m);
    static void degrees(GraphBuilder::Impl& m);
    static void walk_queries(GraphBuilder::Impl& m);
    static void answer_queries(GraphBuilder::Impl& m, const u64* d_keys, u64 n, u64* d_out);
    static void route_queries(GraphBuilder::Impl& m, u32 n_shards, u64* d_routed_keys, u64* counts_host);
    static void unitigs(GraphBuilder::Impl& m);
    static void walk(GraphBuilder::Impl& m);
    static void tail(GraphBuilder::Impl& m, FinalGraph* out, bool want_graph, bool want_paths);
    static void fragments(GraphBuilder::Impl& m);
    static void warm();      // loads this width's code object (an empty launch of its insert kernel)
};
#if AC_W_ONLY != 0 || defined(AC_EMU)
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
#endif
#if AC_W_ONLY != 0
template struct Stages<AC_W_ONLY>;
#endif

#if AC_W_ONLY == 0
[[maybe_unused]] static void ensure_host_stager();      // (HostStager is defined further down)
void device_warmup(int device, uint32_t k, uint64_t text_bytes_estimate) {
#ifndef AC_EMU
    const bool trace = getenv("AC_DEBUG_WARM") != nullptr;
    double t0 = now_s();
    auto lap = [&](const char* what) { if (!trace) return; const double t = now_s(); fprintf(stderr, "[warm] %-28s %7.1f ms\n", what, (t - t0) * 1e3); t0 = t; };
    AC_HIP_CHECK(hipSetDevice(device));
    void* p = nullptr;
    AC_HIP_CHECK(hipMalloc(&p, 4096));
    lap("context + first hipMalloc");
    // the pinned upload ring (192 MB of hipHostMalloc: ~47 ms) on a thread of its own, beside the code objects (~55 ms): round 4, a fresh
    // `autocycler-compress` process waited for the two one after the other (profiles/r10e_cli_fresh_process_configC.txt)
    std::thread ring;
    std::string ring_fail;
    if (text_bytes_estimate) ring = std::thread([&] { try { AC_HIP_CHECK(hipSetDevice(device)); ensure_host_stager(); } catch (const std::exception& e) { ring_fail = e.what(); } });
    struct RingJoin { std::thread& t; ~RingJoin() { if (t.joinable()) t.join(); } } ring_join{ring};
    hipLaunchKernelGGL(functor_kernel<PackFunctor>, dim3(1), dim3(256), 0, 0, (u64)1, PackFunctor{(const u8*)p, 32, (u64*)((u8*)p + 1024), (u32*)((u8*)p + 2048), 0, PackCheck{nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr}});
    (void)hipDeviceSynchronize();
    lap("main code object");
    // the code object of the key width the build will use, the device arena (hipMalloc of gigabytes: ~26 ms per GB) and the pinned upload
    // ring — everything a fresh process would otherwise pay for inside its first build, while the caller still reads its FASTA files
    if (k >= 1 && (int)k <= max_supported_k() && (k & 1)) {
        switch (key_words((int)k)) {
            case 1: Stages<1>::warm(); break; case 2: Stages<2>::warm(); break; case 3: Stages<3>::warm(); break;
            case 4: Stages<4>::warm(); break; case 8: Stages<8>::warm(); break; case 16: Stages<16>::warm(); break;
        }
        (void)hipDeviceSynchronize();
        lap("key-width code object");
    }
    if (text_bytes_estimate) {
        Arena::device().reserve(arena_estimate(text_bytes_estimate, true));
        lap("device arena");
        ring.join();
        lap("pinned upload ring (rest)");
        if (!ring_fail.empty()) throw DeviceError(ring_fail);
    }
    (void)hipDeviceSynchronize();
    (void)hipFree(p);
#else
    (void)device; (void)k; (void)text_bytes_estimate;
#endif
}

// ---- GraphBuilder ------------------------------------------------------------------------------------------------
GraphBuilder::GraphBuilder(uint32_t k) : impl_(new Impl) {
    // A builder owns the device arena for its lifetime (the C ABI serialises builds): whatever the previous build left there is
    // dead.  Nothing a caller can reach lives in it: every array of an ac_graph is a pinned block of its own (PinnedPool) or host heap.
    Arena::device().reset();
    impl_->k = k;
    impl_->loc.k = impl_->uni.k = (int)k;
    impl_->loc.check_alphabet = true;      // the caller's text; the union text of a sharded build is cut out of texts that were checked
    if (k < 1 || (k % 2) == 0) throw DeviceError("k must be odd");
    if ((int)k > max_supported_k())
        throw DeviceError("k-mer sizes above " + std::to_string(max_supported_k()) + " are not supported by this build of the HIP backend");
}
GraphBuilder::~GraphBuilder() { delete impl_; }
uint64_t GraphBuilder::n_text() const { return impl_->loc.n_text; }
void GraphBuilder::set_sequence_index_base(uint64_t n) { impl_->loc.index_base = n; }
void GraphBuilder::set_upload_threads_cap(int n) { tl_upload_threads_cap = n; }
uint64_t GraphBuilder::n_bases() const { return impl_->loc.n_bases; }

// ---- host entry: sequences in the caller's (pageable) memory -> text + packed text in HBM ----------------------------------
// What `ac_compress_build` gets is what compress.rs:41 holds: one heap buffer per Sequence.  A plain hipMemcpy from such memory
// runs at 3 GB/s the first time the runtime sees the pages (it pins them on the fly; measured 158-196 ms for the 487 MB of config
// C, tools/microbench/h2d_probe.hip), against 55-57 GB/s from pinned memory.  So the text layout ('$' + padded sequence + '$' ...)
// is written chunk by chunk into a persistent ring of pinned staging slots by a few host threads (memcpy: 25 GB/s per thread,
// 126 GB/s with eight), every filled slot goes out with one asynchronous copy on an upload stream, and K1 packs that chunk on the
// same stream right behind its copy — the PCIe link never waits, and the build that follows finds bits / mask ready.
// The packing threads of the host entry, kept between builds: starting 16-32 threads costs 0.5-0.9 ms per build (measured: the
// calling thread only gets to the build when the last one is up), waking parked ones a few microseconds.
#ifndef AC_EMU
class UploadPool {
  public:
    static UploadPool& get() { return ctx_object<UploadPool>(CTX_POOL); }
    UploadPool() {}
    // Runs fn() on n threads; returns at once.  One run at a time (the C ABI serialises builds).
    u64 start(int n, std::function<void()> fn) {
        std::unique_lock<std::mutex> lock(mu_);
        while ((int)threads_.size() < n) { const int idx = (int)threads_.size(); threads_.emplace_back([this, idx] { loop(idx); }); }
        fn_ = std::move(fn); want_ = n; active_ = n; gen_++;
        cv_.notify_all();
        return gen_;
    }
    void wait(u64 ticket) {
        std::unique_lock<std::mutex> lock(mu_);
        done_cv_.wait(lock, [&] { return gen_ != ticket || active_ == 0; });
    }
    ~UploadPool() {
        { std::unique_lock<std::mutex> lock(mu_); stop_ = true; cv_.notify_all(); }
        for (auto& t : threads_) t.join();
    }
  private:
    void loop(int idx) {
        u64 seen = 0;
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (idx >= want_) continue;
                fn = fn_;
            }
            fn();
            std::unique_lock<std::mutex> lock(mu_);
            if (--active_ == 0) done_cv_.notify_all();
        }
    }
    std::mutex mu_; std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    std::function<void()> fn_;
    u64 gen_ = 0; int want_ = 0, active_ = 0; bool stop_ = false;
};
#endif

class HostStager {
  public:
    static const size_t SLOT = (size_t)16 << 20;     // 16 MB per copy: the SDMA path reaches 55 GB/s from 16 MB up (1-4 MB: 25-37 GB/s)
    static const int NS = 12;
    static HostStager& get() { return ctx_object<HostStager>(CTX_STAGER); }
    HostStager() {}
    ~HostStager() { release(); }
    void ensure() {
#ifndef AC_EMU
        int dev = 0;
        AC_HIP_CHECK(hipGetDevice(&dev));
        if (created_ && dev == dev_) return;
        if (created_) {
            (void)hipStreamDestroy(s_); (void)hipStreamDestroy(pk_);
            for (auto& e : ev_) (void)hipEventDestroy(e);
            (void)hipEventDestroy(done_); (void)hipEventDestroy(begin_); (void)hipEventDestroy(copied_); (void)hipEventDestroy(first_);
            created_ = false;
        }
        AC_HIP_CHECK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
        AC_HIP_CHECK(hipStreamCreateWithFlags(&pk_, hipStreamNonBlocking));
        for (auto& e : ev_) AC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        AC_HIP_CHECK(hipEventCreate(&done_));
        AC_HIP_CHECK(hipEventCreate(&begin_));
        AC_HIP_CHECK(hipEventCreateWithFlags(&copied_, hipEventDisableTiming));
        AC_HIP_CHECK(hipEventCreateWithFlags(&first_, hipEventDisableTiming));
        created_ = true; dev_ = dev;
#else
        if (!ring_) ring_ = (u8*)malloc(SLOT * NS);
#endif
    }
    void release() {
#ifndef AC_EMU
        if (ring_) (void)hipHostFree(ring_);
#else
        free(ring_);
#endif
        ring_ = nullptr;
    }
    // the pinned ring: allocated when somebody stages through it (the direct upload never does)
    u8* slot(int i) {
#ifndef AC_EMU
        if (!ring_) AC_HIP_CHECK(hipHostMalloc((void**)&ring_, SLOT * NS, hipHostMallocDefault));
#endif
        return ring_ + (size_t)i * SLOT;
    }
    void ensure_ring() { (void)slot(0); }
#ifndef AC_EMU
    hipStream_t stream() { return s_; }            // the copies, back to back
    hipStream_t pack_stream() { return pk_; }      // K1 on each chunk, behind its copy (a kernel between two copies of ONE stream idles the link)
    hipEvent_t& event(int i) { return ev_[i]; }
    hipEvent_t& done() { return done_; }
    hipEvent_t& begin() { return begin_; }
    hipEvent_t& copied() { return copied_; }
    hipEvent_t& first() { return first_; }
    bool timed = false;                            // begin / done bracket an upload whose duration has not been read yet
    double direct_ms = -1;                         // ... or the packers wrote device memory themselves: host clock, first store to last flush
#else
    stream_t stream() { return 0; }
#endif
  private:
    u8* ring_ = nullptr;
    bool created_ = false;
    int dev_ = -1;
#ifndef AC_EMU
    hipStream_t s_ = nullptr, pk_ = nullptr;
    hipEvent_t ev_[NS];
    hipEvent_t done_, begin_, copied_, first_;
#endif
};
void release_host_stager() { HostStager::get().release(); }
[[maybe_unused]] static void ensure_host_stager() {      // (device_warmup: the whole-command path uploads the text as BYTES for the end repair — through the ring)
    HostStager::get().ensure();
    HostStager::get().ensure_ring();
}

// Bytes [b, e) of the text layout of `seqs` (off[i] = first padded byte of sequence i; every padded sequence is followed by '$').
static void fill_text_range(const std::vector<SeqView>& seqs, const std::vector<uint64_t>& off, uint32_t k, u64 b, u64 e, u8* dst) {
    // first sequence whose span [off, off + plen] (the '$' after it included) ends after b
    size_t lo = 0, hi = seqs.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (off[mid] + (u64)seqs[mid].length + k - 1 + 1 <= b) lo = mid + 1; else hi = mid; }
    u64 p = b;
    if (p == 0 && p < e) { dst[0] = '$'; p = 1; }
    for (size_t i = lo; i < seqs.size() && p < e; i++) {
        const u64 s0 = off[i], plen = (u64)seqs[i].length + k - 1;
        if (p < s0 + plen) {
            const u64 from = p - s0, n = std::min(e, s0 + plen) - p;
            memcpy(dst + (p - b), seqs[i].fwd + from, n);
            p += n;
        }
        if (p == s0 + plen && p < e) { dst[p - b] = '$'; p++; }
    }
}

// K1 on the host: 32 text bytes -> one word of 2-bit codes (first base most significant) + 32 mask bits, exactly what PackFunctor
// computes on the device.  AVX2 classifies 32 bytes at a time, BMI2 `pext` squeezes 8 codes out of 8 bytes; ~12 GB/s of text per
// core, so sixteen threads pack as fast as the host's memory delivers the text.
// All of them return the number of mask bits they saw set; `mask` may be null (the upload derives the mask plane on the device and
// only needs the count for the alphabet check).
static u64 pack_groups_scalar(const u8* t, u64 n_groups, u64* bits, u32* mask) {
    u64 nonbase = 0;
    for (u64 g = 0; g < n_groups; g++) {
        u64 w = 0; u32 m = 0;
        for (int i = 0; i < 32; i++) {
            u32 ch = t[g * 32 + (u64)i];
            u32 bad = !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
            u32 c = bad ? 0u : (((ch >> 1) ^ (ch >> 2)) & 3u);
            w |= (u64)c << (62 - 2 * i);
            m |= bad << i;
        }
        bits[g] = w;
        if (mask) mask[g] = m;
        nonbase += (u64)__builtin_popcount(m);
    }
    return nonbase;
}
#if defined(__x86_64__)
}  // namespace ac
#include <immintrin.h>
namespace ac {
__attribute__((target("avx2,bmi2,popcnt"))) static u64 pack_groups_avx2(const u8* t, u64 n_groups, u64* bits, u32* mask) {
    const __m256i vA = _mm256_set1_epi8('A'), vC = _mm256_set1_epi8('C'), vG = _mm256_set1_epi8('G'), vT = _mm256_set1_epi8('T');
    const __m256i three = _mm256_set1_epi8(3);
    const u64 M = 0x0303030303030303ULL;
    u64 nonbase = 0;
    for (u64 g = 0; g < n_groups; g++) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(t + g * 32));
        const __m256i good = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, vA), _mm256_cmpeq_epi8(v, vC)),
                                             _mm256_or_si256(_mm256_cmpeq_epi8(v, vG), _mm256_cmpeq_epi8(v, vT)));
        // ((ch >> 1) ^ (ch >> 2)) & 3 per byte: 16-bit shifts only move a neighbour's bit into bit 7, which the mask drops
        __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);
        c = _mm256_and_si256(c, good);
        const u32 bad = ~(u32)_mm256_movemask_epi8(good);
        if (mask) mask[g] = bad;
        nonbase += (u64)__builtin_popcount(bad);
        alignas(32) u64 q[4];
        _mm256_store_si256((__m256i*)q, c);
        bits[g] = (_pext_u64(__builtin_bswap64(q[0]), M) << 48) | (_pext_u64(__builtin_bswap64(q[1]), M) << 32) |
                  (_pext_u64(__builtin_bswap64(q[2]), M) << 16) | _pext_u64(__builtin_bswap64(q[3]), M);
    }
    return nonbase;
}
// Two groups (64 bytes) per step with AVX-512: codes ((ch >> 1) ^ (ch >> 2)) & 3 under the "is a base" mask, four of them folded into
// a byte by two multiply-adds (4 a + b per byte pair, then 16 x + y per pair of those), sixteen bytes narrowed out of the dwords and
// reversed inside each half so that the first base ends up most significant; the mask bits are the compare masks as they come.
// (Non-temporal stores for the codes — written once, read next by the copy engine — measured neutral: r10l / r10m.)
__attribute__((target("avx512f,avx512bw,avx512vl,ssse3,popcnt"))) static u64 pack_groups_avx512(const u8* t, u64 n_groups, u64* bits, u32* mask) {
    const __m512i vA = _mm512_set1_epi8('A'), vC = _mm512_set1_epi8('C'), vG = _mm512_set1_epi8('G'), vT = _mm512_set1_epi8('T');
    const __m512i three = _mm512_set1_epi8(3);
    const __m512i w1 = _mm512_set1_epi16(0x0104);      // per byte pair (first, second): 4 * first + second   (low byte = first in memory)
    const __m512i w2 = _mm512_set1_epi32(0x00010010);  // per word pair: 16 * first + second
    const __m128i rev = _mm_set_epi8(8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7);
    u64 g = 0, nonbase = 0;
    for (; g + 2 <= n_groups; g += 2) {
        const __m512i v = _mm512_loadu_si512((const void*)(t + g * 32));
        const __mmask64 good = _mm512_cmpeq_epi8_mask(v, vA) | _mm512_cmpeq_epi8_mask(v, vC) | _mm512_cmpeq_epi8_mask(v, vG) | _mm512_cmpeq_epi8_mask(v, vT);
        __m512i c = _mm512_and_si512(_mm512_xor_si512(_mm512_srli_epi16(v, 1), _mm512_srli_epi16(v, 2)), three);
        c = _mm512_maskz_mov_epi8(good, c);
        const __m512i n16 = _mm512_maddubs_epi16(c, w1);        // 16-bit lanes: 4 * b0 + b1
        const __m512i n32 = _mm512_madd_epi16(n16, w2);         // 32-bit lanes: 16 * (4 b0 + b1) + (4 b2 + b3) = four bases, first most significant
        const __m128i by = _mm_shuffle_epi8(_mm512_cvtepi32_epi8(n32), rev);
        _mm_storeu_si128((__m128i*)(bits + g), by);
        const u64 bad = ~(u64)good;
        if (mask) { mask[g] = (u32)bad; mask[g + 1] = (u32)(bad >> 32); }
        nonbase += (u64)__builtin_popcountll(bad);
    }
    if (g < n_groups) nonbase += pack_groups_avx2(t + g * 32, n_groups - g, bits + g, mask ? mask + g : nullptr);
    return nonbase;
}
#endif
static u64 pack_groups(const u8* t, u64 n_groups, u64* bits, u32* mask) {
#if defined(__x86_64__)
    static const bool simd_off = getenv("AC_PACK_SCALAR") != nullptr;
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !simd_off;
    static const bool wide = fast && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
                             getenv("AC_PACK_AVX2") == nullptr;
    if (wide) return pack_groups_avx512(t, n_groups, bits, mask);
    if (fast) return pack_groups_avx2(t, n_groups, bits, mask);
#endif
    return pack_groups_scalar(t, n_groups, bits, mask);
}

// ---- PathRemapJob: seed numbers -> final numbers in the pinned result block ---------------------------------------------------
static void path_remap_scalar(int32_t* p, u64 n, const u32* number, u32 n_unitigs, std::atomic<u32>* bad) {
    u32 wrong = 0;
    for (u64 i = 0; i < n; i++) {
        const int32_t v = p[i];
        const u32 r = (u32)(v > 0 ? v : -v) - 1u;
        if (r >= n_unitigs) { wrong++; continue; }
        const int32_t f = (int32_t)number[r], m = v >> 31;      // (the sign without a branch: strands alternate unpredictably)
        p[i] = (f ^ m) - m;
    }
    if (wrong) bad->fetch_add(wrong);
}
#if defined(__x86_64__)
// sixteen entries per step: |v| - 1 gathers the final number, the sign goes back on under a mask
__attribute__((target("avx512f"))) static void path_remap_avx512(int32_t* p, u64 n, const u32* number, u32 n_unitigs, std::atomic<u32>* bad) {
    const __m512i one = _mm512_set1_epi32(1), zero = _mm512_setzero_si512(), lim = _mm512_set1_epi32((int)n_unitigs);
    u64 i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512i v = _mm512_loadu_si512((const void*)(p + i));
        const __m512i r = _mm512_sub_epi32(_mm512_abs_epi32(v), one);
        const __mmask16 ok = _mm512_cmplt_epu32_mask(r, lim);
        if (ok != 0xFFFF) { path_remap_scalar(p + i, 16, number, n_unitigs, bad); continue; }
        __m512i f = _mm512_i32gather_epi32(r, (const void*)number, 4);
        f = _mm512_mask_sub_epi32(f, _mm512_cmplt_epi32_mask(v, zero), zero, f);
        _mm512_storeu_si512((void*)(p + i), f);
    }
    path_remap_scalar(p + i, n - i, number, n_unitigs, bad);
}
#endif
bool path_remap_is_wide() {
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx512f") && getenv("AC_PACK_SCALAR") == nullptr && getenv("AC_PACK_AVX2") == nullptr;
    return wide;
#else
    return false;
#endif
}
void path_remap_range(int32_t* p, u64 n, const u32* number, u32 n_unitigs, std::atomic<u32>* bad) {
#if defined(__x86_64__)
    if (path_remap_is_wide() && n_unitigs < 0x7FFFFFFFu) { path_remap_avx512(p, n, number, n_unitigs, bad); return; }
#endif
    path_remap_scalar(p, n, number, n_unitigs, bad);
}
#ifndef AC_EMU
void path_remap_start(PathRemapJob& j, int threads) {
    const u64 BLOCK = (u64)1 << 15;
    const int T = (int)std::max<u64>(1, std::min<u64>({(j.n_ent + BLOCK - 1) / BLOCK, (u64)std::max(threads, 1), (u64)std::max(1u, std::thread::hardware_concurrency())}));
    PathRemapJob* job = &j;
    j.started = true;
    j.ticket = UploadPool::get().start(T, [job, BLOCK] {
        int expect = 0;
        if (job->ready.compare_exchange_strong(expect, 1)) {      // one thread waits for the copies, the others watch it
            const bool ok = hipSetDevice(job->dev) == hipSuccess && hipEventSynchronize((hipEvent_t)job->landed) == hipSuccess;
            job->ready.store(ok ? 2 : 3, std::memory_order_release);
        } else {
            while (job->ready.load(std::memory_order_acquire) < 2) std::this_thread::yield();
        }
        if (job->ready.load(std::memory_order_acquire) != 2) { job->bad.fetch_add(1); return; }
        for (u64 b; (b = job->next.fetch_add(BLOCK)) < job->n_ent;)
            path_remap_range(job->path + b, std::min(BLOCK, job->n_ent - b), job->number, job->n_unitigs, &job->bad);
    });
}
void path_remap_finish(PathRemapJob& j) noexcept {
    if (!j.started) return;
    UploadPool::get().wait(j.ticket);
    j.started = false;
}
#else
void path_remap_start(PathRemapJob&, int) {}
void path_remap_finish(PathRemapJob&) noexcept {}
#endif

void pack_text_host(const uint8_t* text, uint64_t n_text, uint64_t* bits, uint32_t* mask32, bool force_scalar) {
    const u64 full = n_text / 32;
    if (force_scalar) pack_groups_scalar(text, full, bits, mask32); else pack_groups(text, full, bits, mask32);
    if (n_text % 32) {      // the last, partial group reads as if the text went on with separators
        u8 tail[32];
        for (u64 i = 0; i < 32; i++) tail[i] = (full * 32 + i < n_text) ? text[full * 32 + i] : (u8)'$';
        if (force_scalar) pack_groups_scalar(tail, 1, bits + full, mask32 + full); else pack_groups(tail, 1, bits + full, mask32 + full);
    }
}

// Packs the groups [g0, g1) of the text layout of `seqs` (group g = text bytes 32 g .. 32 g + 31; bytes beyond the text read as
// separators).  Groups that lie inside one padded sequence — all but two or three per sequence — are packed straight from the
// caller's buffer; only the groups that touch a separator are assembled in a 32-byte scratch first.
// Returns the number of non-base bytes it met (mask bits set, the separators beyond the text's end included): the host entry's
// alphabet check (sequence.rs:39-41) compares their total with what the sequence table promises.
static u64 pack_text_groups(const std::vector<SeqView>& seqs, const std::vector<uint64_t>& off, uint32_t k, u64 n_text, u64 g0, u64 g1,
                            u64* bits, u32* mask) {
    const u64 b = g0 * 32;
    // first sequence whose span [off, off + plen] (the '$' after it included) ends after b
    size_t lo = 0, hi = seqs.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (off[mid] + (u64)seqs[mid].length + k - 1 + 1 <= b) lo = mid + 1; else hi = mid; }
    size_t i = lo;
    u64 g = g0, nonbase = 0;
    auto slow = [&](u64 gg) {      // a group with a separator (or the text's end) in it
        u8 tmp[32];
        const u64 tb = gg * 32, te = std::min(n_text, tb + 32);
        if (tb < te) fill_text_range(seqs, off, k, tb, te, tmp);
        for (u64 j = te > tb ? te - tb : 0; j < 32; j++) tmp[j] = '$';
        nonbase += pack_groups(tmp, 1, bits + (gg - g0), mask ? mask + (gg - g0) : nullptr);
    };
    while (g < g1) {
        while (i < seqs.size() && off[i] + (u64)seqs[i].length + k - 1 <= g * 32) i++;      // sequence i ends at or before this group's start
        if (i >= seqs.size()) { slow(g++); continue; }
        const u64 s0 = off[i], s1 = s0 + (u64)seqs[i].length + k - 1;      // padded bytes of sequence i: [s0, s1)
        if (g * 32 < s0) { slow(g++); continue; }
        const u64 g_in = std::min(g1, s1 / 32);      // groups [g, g_in) lie wholly inside [s0, s1)
        if (g_in > g) {
            nonbase += pack_groups(seqs[i].fwd + (g * 32 - s0), g_in - g, bits + (g - g0), mask ? mask + (g - g0) : nullptr);
            g = g_in;
        } else {
            slow(g++);
        }
    }
    return nonbase;
}
// The host entry's alphabet check failed: name the first sequence that holds anything but A, C, G, T between its padding dots.
[[maybe_unused]] static void throw_bad_alphabet(const std::vector<SeqView>& seqs, uint32_t k, u64 expected, u64 found, u64 index_base = 0) {
    for (size_t i = 0; i < seqs.size(); i++) {
        const u64 plen = (u64)seqs[i].length + k - 1;
        u64 a = 0, b = 0;
        while (a < plen && seqs[i].fwd[a] == '.') a++;
        while (b < plen - a && seqs[i].fwd[plen - 1 - b] == '.') b++;
        for (u64 j = a; j < plen - b; j++) {
            const u8 c = seqs[i].fwd[j];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') throw DeviceError("input sequence " + std::to_string(index_base + i + 1) + " contains non-ACGT characters");
        }
    }
    throw DeviceError("internal error: the packed text holds " + std::to_string(found) + " non-base positions, " + std::to_string(expected) + " expected");
}

// Final (end-repaired) sequences: the text never reaches the device as bytes.  Host threads lay a piece of the text out in a
// cache-resident buffer, pack it (K1 above) straight into a pinned slot, and whoever finishes a 64 MB chunk sends its 16 MB of
// codes; the mask plane is derived on the device from the sequence table: 0.25 bytes per base cross PCIe instead of 1 (config C:
// 122 MB instead of 487 MB).
void GraphBuilder::upload_packed(const std::vector<SeqView>& seqs, const std::vector<uint64_t>& off) {
    const uint32_t k = impl_->k;
    PackedText& loc = impl_->loc;
    const u64 n = loc.n_text;
    HostStager& st = HostStager::get();
    st.ensure();
    const u64 CH = upload_chunk_bytes(), SUB = (u64)1 << 20;      // text bytes per chunk (one pair of copies) / per work item
    const u64 SLOT_BYTES = CH / 4;                         // the codes of one chunk (the mask plane is derived on the device)
    [[maybe_unused]] const int NSLOT = std::max(1, std::min((int)((HostStager::SLOT * HostStager::NS) / SLOT_BYTES), upload_slots()));
    [[maybe_unused]] const u64 n_chunks = (n + CH - 1) / CH, subs = CH / SUB;
    [[maybe_unused]] auto chunk_len = [&](u64 c) { return std::min(n, (c + 1) * CH) - c * CH; };
#ifdef AC_EMU
    loc.pack_alloc();
    u64 nonbase = 0;
    for (u64 b = 0; b < n; b += SUB) {
        const u64 e = std::min(n, b + SUB);
        nonbase += pack_text_groups(seqs, off, k, n, b / 32, (e + 31) / 32, loc.bits.ptr() + b / 32, (u32*)loc.mask.ptr() + b / 32);
    }
    const u64 expected = loc.expected_nonbase + ((n + 31) / 32 * 32 - n);
    if (nonbase != expected) throw_bad_alphabet(seqs, k, expected, nonbase, loc.index_base);
    {      // what the device does instead of receiving the mask plane (MaskTableFunctor) must give the packed one
        DBuf<u64> derived(loc.mask.size());
        derived.fill_bytes(0xFF);
        memset(derived.ptr(), 0, (size_t)((n + 63) / 64) * 8);
        launch((u64)loc.n_seqs + 1, MaskTableFunctor{loc.seq_off.ptr(), loc.seq_len.ptr(), loc.seq_d1.ptr(), loc.seq_d2.ptr(), loc.n_seqs, (int)k, n, derived.ptr()});
        if (memcmp(derived.ptr(), loc.mask.ptr(), loc.mask.size() * 8) != 0) throw DeviceError("internal error: the mask plane derived from the sequence table differs from the packed one");
    }
#else
    Impl::UploadJob* job = new Impl::UploadJob();
    impl_->job = job;
    job->seqs = &seqs; job->off = off; job->k = k; job->n = n; job->CH = CH; job->SUB = SUB; job->NSLOT = NSLOT; job->n_chunks = n_chunks;
    job->slot_bytes = SLOT_BYTES;
    job->stager = &st;
    job->expected_nonbase = loc.expected_nonbase + ((n + 31) / 32 * 32 - n);
    AC_HIP_CHECK(hipGetDevice(&job->dev));
    job->up = st.stream(); job->pk = st.pack_stream();
    flush_fills();
    AC_HIP_CHECK(hipEventRecord(st.begin(), 0));
    AC_HIP_CHECK(hipStreamWaitEvent(job->pk, st.begin(), 0));
    loc.pack_alloc(job->pk);                               // zero codes / all-ones mask beyond the text (and under it, until the copies land)
    // the mask plane from the sequence table, on the device (MaskTableFunctor): 0.25 instead of 0.375 bytes per base cross PCIe
    AC_HIP_CHECK(hipMemsetAsync(loc.mask.ptr(), 0, (size_t)((n + 63) / 64) * 8, job->pk));
    launch((u64)loc.n_seqs + 1, MaskTableFunctor{loc.seq_off.ptr(), loc.seq_len.ptr(), loc.seq_d1.ptr(), loc.seq_d2.ptr(), loc.n_seqs, (int)k, n, loc.mask.ptr()}, job->pk);
    AC_HIP_CHECK(hipEventRecord(st.copied(), job->pk));
    AC_HIP_CHECK(hipStreamWaitEvent(job->up, st.copied(), 0));
    job->d_bits = loc.bits.ptr();
    {
        job->direct = upload_direct_for(job->dev);
        if (!job->direct) st.ensure_ring();
        job->fills_done = st.copied();
        if (job->direct) AC_HIP_CHECK(hipStreamWaitEvent(0, st.copied(), 0));      // (no copies to order the insert behind the mask plane and the slack fills)
        job->t_start = now_s();
    }
    job->done = std::vector<std::atomic<u32>>(n_chunks); job->slot_state = std::vector<std::atomic<u32>>(n_chunks);
    job->issued = std::vector<std::atomic<u32>>(n_chunks);
    for (u64 c = 0; c < n_chunks; c++) { job->done[c].store(0); job->slot_state[c].store(0); job->issued[c].store(0); }
    job->landed.assign(n_chunks, nullptr);
    if (!job->direct) for (auto& e : job->landed) AC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int T = (int)std::max<u64>(1, std::min<u64>({(n + SUB - 1) / SUB, upload_threads(), (u64)std::max(1u, std::thread::hardware_concurrency())}));
    job->ticket = UploadPool::get().start(T, [job] { job->run(); });
    // This thread goes on to the build: the insert waits for the chunks as it gets to them (Impl::need_text).  Without the overlap
    // (AC_UPLOAD_OVERLAP=0) everything is on the device before anything else is issued.
    if (!upload_overlap()) { impl_->need_text(n); impl_->finish_upload(); }
#endif
    loc.packed = true;
}

#ifndef AC_EMU
void GraphBuilder::Impl::UploadJob::run() {
    HostStager& st = *(HostStager*)stager;
    const u64 subs = CH / SUB;
    auto chunk_len = [&](u64 c) { return std::min(n, (c + 1) * CH) - c * CH; };
    auto slot_bits = [&](int sl) { return (u64*)(st.slot(0) + (u64)sl * slot_bytes); };
    try {
        AC_HIP_CHECK(hipSetDevice(dev));
        for (u64 item; (item = next.fetch_add(1)) < n_chunks * subs && !stop.load();) {
            const u64 c = item / subs, sub = item % subs;
            const u64 clen = chunk_len(c);
            if (sub * SUB >= clen) continue;
            if (direct) {
                // Straight into device memory: 16-byte stores in ascending order combine into full PCIe writes, nothing is ever read
                // back from there by the packers.  A work item is on the device when its stores have left this core (sfence) and a
                // read from the device has come back behind them (a PCIe read does not pass posted writes); only then does it count.
                const u64 b = c * CH + sub * SUB, e = std::min(c * CH + clen, b + SUB);
                if (e >= n) AC_HIP_CHECK(hipEventSynchronize(fills_done));      // (the text's last words share a 16-byte unit with the slack the device zeroes)
                u64* dst = d_bits + b / 32;
                const u64 ng = (e + 31) / 32 - b / 32;
                nonbase.fetch_add(pack_text_groups(*seqs, off, k, n, b / 32, (e + 31) / 32, dst, nullptr), std::memory_order_relaxed);
#if defined(__x86_64__)
                _mm_sfence();
#endif
                std::atomic_thread_fence(std::memory_order_release);      // (hosts without sfence: at least the portable release fence)
                if (ng) { const volatile u64* back = dst + (ng - 1); bar_sink.fetch_xor(*back, std::memory_order_relaxed); }
                const u32 n_sub = (u32)((clen + SUB - 1) / SUB);
                if (done[c].fetch_add(1, std::memory_order_acq_rel) + 1 == n_sub) {
                    issued[c].store(1, std::memory_order_release);
                    if (chunks_issued.fetch_add(1) + 1 == n_chunks) t_last.store(now_s());
                }
                continue;
            }
            const int sl = (int)(c % (u64)NSLOT);
            if (c >= (u64)NSLOT) {      // the chunk that used this slot before must have left it: one thread waits, the others watch it
                u32 expect = 0;
                if (slot_state[c].compare_exchange_strong(expect, 1)) {
                    while (!issued[c - NSLOT].load(std::memory_order_acquire) && !stop.load()) std::this_thread::yield();
                    if (!stop.load()) AC_HIP_CHECK(hipEventSynchronize(landed[c - NSLOT]));
                    slot_state[c].store(2, std::memory_order_release);
                } else {
                    while (slot_state[c].load(std::memory_order_acquire) != 2 && !stop.load()) std::this_thread::yield();
                }
                if (stop.load()) break;
            }
            const u64 b = c * CH + sub * SUB, e = std::min(c * CH + clen, b + SUB);
            nonbase.fetch_add(pack_text_groups(*seqs, off, k, n, b / 32, (e + 31) / 32, slot_bits(sl) + sub * SUB / 32, nullptr),
                              std::memory_order_relaxed);
            const u32 n_sub = (u32)((clen + SUB - 1) / SUB);
            if (done[c].fetch_add(1, std::memory_order_acq_rel) + 1 == n_sub) {      // the chunk is complete: send it
                const u64 g0 = c * CH / 32, ng = (clen + 31) / 32;
                std::lock_guard<std::mutex> lock(hip_mu);
                AC_HIP_CHECK(hipMemcpyAsync(d_bits + g0, slot_bits(sl), ng * 8, hipMemcpyHostToDevice, up));
                AC_HIP_CHECK(hipEventRecord(landed[c], up));      // the chunk is on the device (and its slot free again)
                issued[c].store(1, std::memory_order_release);
            }
        }
    } catch (const std::exception& ex) {
        std::lock_guard<std::mutex> lock(hip_mu);
        if (fail.empty()) fail = ex.what();
        stop.store(true);
    }
}
void GraphBuilder::Impl::need_text(u64 upto) {
    if (!job) return;
    while (job->next_wait < job->n_chunks && job->next_wait * job->CH < upto) {
        const u64 c = job->next_wait;
        while (!job->issued[c].load(std::memory_order_acquire) && !job->stop.load()) std::this_thread::yield();
        if (job->stop.load()) finish_upload();      // throws
        if (!job->direct) {
            flush_fills();
            AC_HIP_CHECK(hipStreamWaitEvent(0, job->landed[c], 0));
        }
        job->next_wait++;
    }
    if (job->next_wait == job->n_chunks) finish_upload();
}
u64 GraphBuilder::Impl::upload_rest_limit(u64 pb) const {
    if (!job) return ~0ULL;
    for (u64 c = job->next_wait; c < job->n_chunks; c++) {      // the end of the first chunk that gives this launch something to do
        const u64 end = std::min(job->n, (c + 1) * job->CH);
        if (c + 1 == job->n_chunks) break;
        if (end > pb + (u64)k + 8192 + (1u << 20)) return end - (u64)k - 8192;
    }
    return ~0ULL;
}
void GraphBuilder::Impl::finish_upload() {
    if (!job) return;
    UploadJob* j = job;
    job = nullptr;
    UploadPool::get().wait(j->ticket);
    HostStager& st = HostStager::get();
    std::string fail = j->fail;
    if (fail.empty() && j->direct) {
        st.direct_ms = j->t_last.load() > 0 ? (j->t_last.load() - j->t_start) * 1e3 : -1.0;
    } else if (fail.empty()) {
        if (hipEventRecord(st.done(), j->up) != hipSuccess) fail = "hipEventRecord failed";
        st.timed = true;
    } else { (void)hipStreamSynchronize(j->up); (void)hipStreamSynchronize(j->pk); }
    if (fail.empty() && !j->stop.load() && j->nonbase.load() != j->expected_nonbase) {      // sequence.rs:39-41 (every chunk was packed: nobody stopped)
        try { throw_bad_alphabet(*j->seqs, j->k, j->expected_nonbase, j->nonbase.load(), loc.index_base); } catch (const std::exception& ex) { fail = ex.what(); }
    }
    for (auto& e : j->landed) if (e) (void)hipEventDestroy(e);      // (a destroyed event that a stream still waits for stays valid until then)
    delete j;
    if (!fail.empty()) throw DeviceError(fail);
}
GraphBuilder::Impl::~Impl() {
    if (job) { job->stop.store(true); try { finish_upload(); } catch (...) {} }
}
#else
void GraphBuilder::Impl::need_text(u64) {}
u64 GraphBuilder::Impl::upload_rest_limit(u64) const { return ~0ULL; }
void GraphBuilder::Impl::finish_upload() {}
GraphBuilder::Impl::~Impl() {}
#endif

void GraphBuilder::set_sequences_host(const std::vector<SeqView>& seqs, bool pack_now) {
    const double t0 = now_s();
    const uint32_t k = impl_->k;
    const size_t S = seqs.size();
    std::vector<uint64_t> off(S); std::vector<uint32_t> len(S); std::vector<uint16_t> d1(S), d2(S);
    u64 n = 1;
    for (size_t i = 0; i < S; i++) {
        const u64 plen = (u64)seqs[i].length + k - 1;
        off[i] = n; len[i] = seqs[i].length;
        u16 a = 0, b = 0;
        while (a < plen && seqs[i].fwd[a] == '.') a++;
        while (b < plen && seqs[i].fwd[plen - 1 - b] == '.') b++;
        d1[i] = a; d2[i] = b;
        n += plen + 1;
    }
    PackedText& loc = impl_->loc;
    loc.n_text = n;
    if (pack_now && host_pack()) {      // the sequences are final: pack on the host, upload 0.375 B per base
        Arena::device().reserve(arena_estimate(n, false));
        loc.d_text = nullptr;
        loc.check_alphabet = false;      // K1 runs on the host here: its packers count the non-base bytes (finish_upload)
        loc.set_table(off, len, d1, d2);
        upload_packed(seqs, off);
        tm_.h2d = now_s() - t0;
        return;
    }
    Arena::device().reserve(arena_estimate(n, true));
    impl_->text_owned.alloc(n + 64);
    loc.d_text = impl_->text_owned.ptr();
    loc.set_table(off, len, d1, d2);
    HostStager& st = HostStager::get();
    st.ensure();
    st.ensure_ring();
    const u64 C = HostStager::SLOT;
    const u64 n_chunks = (n + C - 1) / C;
    u8* const d_text = impl_->text_owned.ptr();
#ifdef AC_EMU
    if (pack_now) loc.pack_alloc();
    for (u64 c = 0; c < n_chunks; c++) {
        const u64 b = c * C, e = std::min(n, b + C);
        fill_text_range(seqs, off, k, b, e, st.slot(0));
        memcpy(d_text + b, st.slot(0), e - b);
        if (pack_now) launch((e - b + 31) / 32, PackFunctor{d_text, n, loc.bits.ptr(), (u32*)loc.mask.ptr(), b / 32, loc.chk()});
    }
#else
    int dev = 0;
    AC_HIP_CHECK(hipGetDevice(&dev));
    hipStream_t up = st.stream(), pk = st.pack_stream();
    {   // both streams start after whatever stream 0 still has in flight (the table copies above); the fills of bits / mask go
        flush_fills();
        AC_HIP_CHECK(hipEventRecord(st.begin(), 0));      // first on the pack stream
        AC_HIP_CHECK(hipStreamWaitEvent(up, st.begin(), 0));
        AC_HIP_CHECK(hipStreamWaitEvent(pk, st.begin(), 0));
    }
    if (pack_now) loc.pack_alloc(pk);
    std::atomic<u64> next{0};
    std::vector<std::atomic<u64>> issued(HostStager::NS);
    for (auto& x : issued) x.store(0);
    std::mutex hip_mu;
    std::string fail;
    std::atomic<bool> stop{false};
    auto worker = [&] {
        try {
            AC_HIP_CHECK(hipSetDevice(dev));
            for (u64 c; (c = next.fetch_add(1)) < n_chunks;) {
                const int sl = (int)(c % HostStager::NS);
                if (c >= (u64)HostStager::NS) {      // the slot's previous chunk must have left it
                    while (issued[sl].load(std::memory_order_acquire) != c - HostStager::NS + 1 && !stop.load()) std::this_thread::yield();
                    if (stop.load()) break;
                    AC_HIP_CHECK(hipEventSynchronize(st.event(sl)));
                }
                const u64 b = c * C, e = std::min(n, b + C);
                fill_text_range(seqs, off, k, b, e, st.slot(sl));
                {
                    std::lock_guard<std::mutex> lock(hip_mu);
                    AC_HIP_CHECK(hipMemcpyAsync(d_text + b, st.slot(sl), e - b, hipMemcpyHostToDevice, up));
                    AC_HIP_CHECK(hipEventRecord(st.event(sl), up));
                    if (pack_now) {
                        AC_HIP_CHECK(hipStreamWaitEvent(pk, st.event(sl), 0));
                        launch((e - b + 31) / 32, PackFunctor{d_text, n, loc.bits.ptr(), (u32*)loc.mask.ptr(), b / 32, loc.chk()}, pk);
                    }
                }
                issued[sl].store(c + 1, std::memory_order_release);
            }
        } catch (const std::exception& ex) {
            std::lock_guard<std::mutex> lock(hip_mu);
            if (fail.empty()) fail = ex.what();
            next.store(n_chunks);      // no more chunks, and nobody keeps waiting for a slot
            stop.store(true);
        }
    };
    const int T = (int)std::min<u64>(n_chunks, std::min<u64>(upload_threads(), 8));
    std::vector<std::thread> pool;
    for (int i = 1; i < T; i++) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    if (!fail.empty()) { (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(pk); throw DeviceError(fail); }
    // the build (stream 0) starts when the last chunk has landed and is packed; the caller's buffers are no longer referenced
    // from here on (every fill has been copied into the ring)
    AC_HIP_CHECK(hipEventRecord(st.copied(), up));
    AC_HIP_CHECK(hipStreamWaitEvent(pk, st.copied(), 0));
    AC_HIP_CHECK(hipEventRecord(st.done(), pk));
    AC_HIP_CHECK(hipStreamWaitEvent(0, st.done(), 0));
    st.timed = true;
#endif
    loc.packed = pack_now;
    tm_.h2d = now_s() - t0;      // host side of the pipeline (the last copies may still be in flight: the build's first sync absorbs them)
}
void GraphBuilder::repair_ends(RepairTimings* tm) {
    PackedText& loc = impl_->loc;
    if (!impl_->text_owned.ptr() || loc.packed) throw DeviceError("repair_ends: needs the unpacked text of set_sequences_host(seqs, false)");
    std::vector<uint64_t> off = loc.h_off; std::vector<uint32_t> len = loc.h_len;
    std::vector<uint16_t> d1(off.size()), d2(off.size());
    end_repair_device(impl_->k, impl_->text_owned.ptr(), loc.n_text, off, len, &d1, &d2, tm, /*reset_arena=*/false);
    loc.set_table(off, len, d1, d2);
}
void GraphBuilder::set_text_device(const uint8_t* d_text, uint64_t n_text, const std::vector<uint64_t>& off,
                                   const std::vector<uint32_t>& len, const std::vector<uint16_t>& d1,
                                   const std::vector<uint16_t>& d2) {
    impl_->loc.d_text = d_text;
    impl_->loc.n_text = n_text;
    Arena::device().reserve(arena_estimate(n_text, false));
    impl_->loc.set_table(off, len, d1, d2);
}

#define AC_DISPATCH_W(NAME, ARGS)                                        \
    switch (key_words((int)impl_->k)) {                                  \
        case 1: Stages<1>::NAME ARGS; break;                             \
        case 2: Stages<2>::NAME ARGS; break;                             \
        case 3: Stages<3>::NAME ARGS; break;                             \
        case 4: Stages<4>::NAME ARGS; break;                             \
        case 8: Stages<8>::NAME ARGS; break;                             \
        case 16: Stages<16>::NAME ARGS; break;                           \
        default: throw DeviceError("unsupported k");                     \
    }

void GraphBuilder::build(uint32_t assembly_count_hint, FinalGraph* out) {
    BuildTimings keep = tm_;
    tm_ = BuildTimings();
    tm_.h2d = keep.h2d;
    tm_.graph_hint = assembly_count_hint;
    Impl& m = *impl_;
    m.begin(&tm_);
    m.G = &m.loc;
    m.host_remap_allowed = true; m.host_remap_allowed_build = true; m.checked_sorts = false;
    m.check_sizes(m.loc);
    m.pack_overlapped(assembly_count_hint);
    m.lap(&tm_.pack);
    const Arena::Mark packed = Arena::device().mark();
    for (;;) {
        try {
            AC_DISPATCH_W(table, (*impl_))
            AC_DISPATCH_W(degrees, (*impl_))
            AC_DISPATCH_W(unitigs, (*impl_))
            AC_DISPATCH_W(walk, (*impl_))
            AC_DISPATCH_W(tail, (*impl_, out, true, true))
            break;
        } catch (const NeedCheckedSorts&) {
            // a deferred "group too large" flag was set (many unitigs sharing a key prefix): once more, every sort checked where it runs
            if (m.checked_sorts) throw DeviceError("internal error: checked sorts left a flag");
            m.checked_sorts = true;
            {
                BuildTimings again = BuildTimings();
                again.h2d = tm_.h2d; again.pack = tm_.pack; again.graph_hint = tm_.graph_hint; again.position_retries = tm_.position_retries; again.sort_retries = tm_.sort_retries + 1;
                tm_ = again;
            }
            stream_sync();
            Arena::device().rewind(packed);
            m.sort_flags.fill_bytes(0);
        } catch (const NeedExactPositions&) {
            // expand_repeats met a common sequence longer than the bound the walk kept for a destination's smallest position
            // (exp_avoid_start_of_path): everything behind the packed text again, with exact positions
            if (m.exact_positions) throw DeviceError("internal error: exact positions were not exact");
            m.exact_positions = true;
            {   // the stage times and counts are those of the attempt that delivered
                BuildTimings again = BuildTimings();
                again.h2d = tm_.h2d; again.pack = tm_.pack; again.graph_hint = tm_.graph_hint; again.position_retries = tm_.position_retries + 1;
                tm_ = again;
            }
            stream_sync();
            Arena::device().rewind(packed);
            m.sort_flags.fill_bytes(0);      // (a flag of the abandoned attempt must not repeat the next one)
        }
    }
#ifndef AC_EMU
    if (HostStager::get().timed) {      // host entry: first copy issued -> last chunk packed, on the device's clock
        float ms = 0;
        if (hipEventElapsedTime(&ms, HostStager::get().begin(), HostStager::get().done()) == hipSuccess) tm_.upload_device_ms = ms;
        HostStager::get().timed = false;
    }
    if (HostStager::get().direct_ms >= 0) { tm_.upload_device_ms = HostStager::get().direct_ms; HostStager::get().direct_ms = -1; }      // (direct stores: the host's clock)
#endif
}

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

// Self-test of the hand-written primitives of device_prims.hpp against the host's std:: algorithms on n pseudo-random items (what the
// CPU suite runs under the emulation and the device suite on the GPU: tile boundaries, duplicate-heavy keys for stability, end bits
// that are not a multiple of eight, millions of tiles' worth of look-back).  Throws on the first mismatch.
void primitives_selftest(uint64_t n, uint64_t seed, int end_bit, int key_kind) {
    Arena::device().reset();
    std::vector<u64> hk(n), hk64(n + 1); std::vector<u32> hv(n), h32(n + 1);
    u64 st = seed * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
    auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    const u64 kmask = end_bit >= 64 ? ~0ULL : ((1ULL << end_bit) - 1);
    for (u64 i = 0; i < n; i++) {
        const u64 r = rnd();
        // key kinds: 0 uniform, 1 few distinct values (long runs of equal keys: stability), 2 already sorted, 3 reverse sorted, 4 one hot digit
        hk[i] = key_kind == 0 ? r : (key_kind == 1 ? (r % 7) * 0x0101010101010101ULL : (key_kind == 2 ? i * 3 : (key_kind == 3 ? (n - i) * 5 : ((r & 0xFF00FFULL) | 0xAB00ULL))));
        hv[i] = (u32)i; h32[i] = (u32)(r >> 40) & 1023u; hk64[i] = (r >> 20) & 0x3FFFFFULL;      // (running totals below 2^46: device_prims.hpp)
    }
    h32[n] = 0; hk64[n] = 0;
    if (n) {
        DBuf<u64> dk(n); DBuf<u32> dv(n);
        copy_h2d(dk.ptr(), hk.data(), n * 8); copy_h2d(dv.ptr(), hv.data(), n * 4);
        sort_pairs_u64_u32(dk, dv, n, end_bit);
        std::vector<u64> gk = to_host(dk, n); std::vector<u32> gv = to_host(dv, n);
        std::vector<u32> idx(n);
        for (u64 i = 0; i < n; i++) idx[i] = (u32)i;
        std::stable_sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return (hk[a] & kmask) < (hk[b] & kmask); });
        for (u64 i = 0; i < n; i++)
            if (gk[i] != hk[idx[i]] || gv[i] != idx[i]) throw DeviceError("primitives self-test: radix sort differs from std::stable_sort at " + std::to_string(i) + " of " + std::to_string(n));
        // comparator sorts (fallback paths): the same order from the merge sort by ranks
        DBuf<u32> order(n);
        copy_h2d(order.ptr(), hv.data(), n * 4);
        DBuf<u64> dk2(n);
        copy_h2d(dk2.ptr(), hk.data(), n * 8);
        struct Less { const u64* k; u64 m; AC_HD bool operator()(u32 a, u32 b) const { return (k[a] & m) < (k[b] & m); } };
        if (n <= (1u << 18)) {
            sort_keys_cmp(order, n, Less{dk2.ptr(), kmask});
            std::vector<u32> go = to_host(order, n);
            for (u64 i = 0; i < n; i++) if (go[i] != idx[i]) throw DeviceError("primitives self-test: comparator sort differs at " + std::to_string(i));
        }
    }
    {   // scans over n + 1 items (the pipeline's "sentinel" form) and over n
        DBuf<u32> a(n + 1), o(n + 1); DBuf<u64> a64(n + 1), o64(n + 1);
        copy_h2d(a.ptr(), h32.data(), (n + 1) * 4); copy_h2d(a64.ptr(), hk64.data(), (n + 1) * 8);
        exclusive_scan_u32(a.ptr(), o.ptr(), n + 1);
        std::vector<u32> g = to_host(o, n + 1);
        u32 acc = 0;
        for (u64 i = 0; i <= n; i++) { if (g[i] != acc) throw DeviceError("primitives self-test: exclusive_scan_u32 differs at " + std::to_string(i)); acc += h32[i]; }
        exclusive_scan_u64(a64.ptr(), o64.ptr(), n + 1);
        std::vector<u64> g64 = to_host(o64, n + 1);
        u64 acc64 = 0;
        for (u64 i = 0; i <= n; i++) { if (g64[i] != acc64) throw DeviceError("primitives self-test: exclusive_scan_u64 differs at " + std::to_string(i)); acc64 += hk64[i]; }
        if (n) {
            inclusive_scan_u32(a.ptr(), o.ptr(), n);
            g = to_host(o, n); acc = 0;
            for (u64 i = 0; i < n; i++) { acc += h32[i]; if (g[i] != acc) throw DeviceError("primitives self-test: inclusive_scan_u32 differs at " + std::to_string(i)); }
            inclusive_max_scan_u32(a.ptr(), o.ptr(), n);
            g = to_host(o, n); acc = 0;
            for (u64 i = 0; i < n; i++) { acc = std::max(acc, h32[i]); if (g[i] != acc) throw DeviceError("primitives self-test: inclusive_max_scan_u32 differs at " + std::to_string(i)); }
            exclusive_scan_u32(a.ptr(), a.ptr(), n);      // in place
            g = to_host(a, n); acc = 0;
            for (u64 i = 0; i < n; i++) { if (g[i] != acc) throw DeviceError("primitives self-test: in-place scan differs at " + std::to_string(i)); acc += h32[i]; }
        }
    }
    stream_sync();
}

#include "neighbours.inc"      // device end repair (f-1) and pairwise contig distances (f-3)
#include "kernels_verify.inc"  // ac_verify_graph: the round-trip verifier at scale (f-4)
#endif   // AC_W_ONLY == 0

}  // namespace ac
