// Everything the translation units of the device graph build share (graph_build.hpp is the interface): the table and text types, the
// kernels (kernels_*.inc), the tuning knobs, PackedText and GraphBuilder::Impl — the device state of one build — and the per-width
// stage dispatch.  Included by graph_build.hip (builder, single-device driver), graph_stages.hip (the width-dependent stages: compiled
// once per key width), graph_upload.hip (host entry: packers, upload, path renumbering), graph_shard.hip (the phases of a build
// over several devices) and graph_extras.hip (end repair, pairwise distances, verifier).  gfx950 HIP; under -DAC_EMU the same sources
// compile as the CPU emulation for the CPU test-suite.
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
#pragma once
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
extern bool g_stage_timing;      // (graph_build.hip)

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
    // stretch mode (round 6, kernels_paths.inc): the entries did not cross the link, their stretches did — rec_val[s] = first value of
    // stretch s (a text-order number, signed), rec_pos[s] = its first entry's index; the threads WRITE path[] from the number table
    const int32_t* rec_val = nullptr; const u32* rec_pos = nullptr; u64 n_rec = 0;
    const u32* number = nullptr; u32 n_unitigs = 0;      // pinned: final number of seed index r at [r]
    u64 ent_limit = ~0ULL;                               // stretch mode: stretches that begin at or behind this entry are not the host's (the device renumbers that share and sends it over)
    void* landed = nullptr;                              // event: entries and number table are in host memory
    int dev = 0;
    std::atomic<u64> next{0}; std::atomic<int> ready{0};      // ready: 0 nobody waits yet, 1 one thread waits for `landed`, 2 go, 3 failed
    std::atomic<u32> bad{0};                             // entries that name no unitig (never, short of a bug: reported as an internal error)
    u64 ticket = 0; bool started = false;
    std::atomic<double> t_ready{0}, t_last{0};      // diagnostics (AC_DEBUG_ARENA): when the copies the job waits for had landed, when its last block was done (now_s clock)
};
void path_remap_range(int32_t* p, u64 n, const u32* number, u32 n_unitigs, std::atomic<u32>* bad);
void path_stretch_range(const PathRemapJob& j, u64 s0, u64 s1, std::atomic<u32>* bad);      // stretches [s0, s1) written out
bool path_remap_is_wide();      // the host has the 16-lane gather (without it a thread renumbers ~5x slower and the device keeps the job)
void path_remap_start(PathRemapJob& j, int threads);      // returns at once; the work runs on the packing threads' pool
void path_remap_finish(PathRemapJob& j) noexcept;         // until every thread is done (idempotent)
// SeqExpandJob (round 6): the unitig sequences are the largest result that is final only behind the last pass (config D 126 of 150 MB, mini-E
// 157 of 494) and cross the link with nothing left to hide under — as 2-bit codes (SeqPack2Functor: 32 bases per word, base i in bits 2i, 2i+1;
// A C G T = 0 1 2 3: trimmed unitig sequences hold no other byte) a quarter of the bytes do, and a few host threads write the bytes of the
// result block out while the unitig records and the links are still crossing.
struct SeqExpandJob {
    const u64* words = nullptr; u8* out = nullptr; u64 total = 0;      // pinned: the codes as they land; the result block (total bytes)
    void* landed = nullptr; int dev = 0;                                // event: the codes are in host memory
    std::atomic<u64> next{0}; std::atomic<int> ready{0};
    u64 ticket = 0; bool started = false;
    std::atomic<double> t_start{0}, t_ready{0}, t_last{0};      // diagnostics (AC_DEBUG_ARENA): job started / its codes had landed / its last block was done
};
void seq_expand_range(const u64* words, u8* out, u64 b, u64 e);      // bytes [b, e) of the sequences from their codes (b a multiple of 32)
void seq_expand_start(SeqExpandJob& j, int threads);
void seq_expand_finish(SeqExpandJob& j) noexcept;
[[maybe_unused]] static int key_words(int k) { int w = words_for_k(k); return w <= 4 ? w : (w <= 8 ? 8 : 16); }
[[maybe_unused]] static u64 next_pow2(u64 x) { u64 p = 1; while (p < x) p <<= 1; return p; }
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
// The tuning / test knobs (environment) as ONE struct, read once per process (round 6: every accessor used to call getenv on every build).
// tests and tools/ab_knobs.py, which change the variables between the builds of one process, set AC_TUNING_FOLLOW_ENV=1 before the library
// is first used: the struct is then read again whenever a build selects its device (under the C ABI's build lock).  The accessors below
// keep their names; the comments there say what each knob does.  None of them changes a result (tests: *_tuning_knobs_*).
struct Knobs {
    int minkey_variant;
    int path_copy;
    u64 run_piece;
    u32 pos_cap;
    bool path_filter;
    int host_remap_mode;
    u32 remap_block;
    int minkey_prefix_bases;
    bool seed_prefix_sort;
    u32 seed_max_group;
    int seed_prefix_bits;
    u32 degree_region_cap;
    int upload_direct_mode;
    int upload_slots;
    int degree_flags;
    int table_shift;
    u64 wave_chunk_max;
    u64 insert_chunk_rest_env;
    bool host_pack;
    u32 expand_level_table;
    bool shard_path_copy;
    bool expand_rewrite_always;
    u32 expand_sparse_max, expand_sparse_list, expand_sparse_batch;
    u32 stretch_device_share;
    int seq_codes_transfer;
    int late_copies;
    bool seq_writer_plain;
    bool seq_writer_forced;
    u64 seed_radix_limit;
    bool upload_overlap;
    bool sort_checks_deferrable;
    bool shard_host_remap;
    bool shard_degree_flags;
    bool insert_adaptive;
    u64 insert_growth;
    u64 insert_waves_target;
    bool renum_two_pass;
    u32 renum_max_group;
    u32 path_chunk_env;
    long upload_threads_env;
    bool debug_arena;
    bool degree_diag;
    int multi_transport;
    int multi_fragments;
    int multi_tail;
    static Knobs read() {
        Knobs k;
        k.minkey_variant = [&]() -> int { const char* e = getenv("AC_MINKEY_VARIANT"); return e ? atoi(e) : -1; }();
        k.path_copy = [&]() -> int { const char* e = getenv("AC_PATH_COPY"); return e ? (atoi(e) != 0 ? 1 : 0) : 2; }();
        k.run_piece = [&]() -> u64 { const char* e = getenv("AC_RUN_PIECE"); const long v = e ? atol(e) : 0; return v > 0 ? (u64)v : 4096; }();
        k.pos_cap = [&]() -> u32 { const char* e = getenv("AC_POS_CAP"); const long v = e ? atol(e) : 65536; return v < 0 ? 0u : (u32)std::min<long>(v, 0x3FFFFFFF); }();
        k.path_filter = [&]() -> bool { const char* e = getenv("AC_PATH_FILTER"); return e ? atoi(e) != 0 : true; }();
        k.host_remap_mode = [&]() -> int { const char* e = getenv("AC_HOST_REMAP"); return e ? (atoi(e) == 2 ? 2 : (atoi(e) != 0 ? 1 : 0)) : -1; }();
        k.remap_block = [&]() -> u32 { const char* e = getenv("AC_REMAP_BLOCK"); int v = e ? atoi(e) : 4096; v = v < 64 ? 64 : (v > 65536 ? 65536 : v); return (u32)(v & ~63); }();
        k.minkey_prefix_bases = [&]() -> int { const char* e = getenv("AC_MINKEY_PREFIX_BASES"); int v = e ? atoi(e) : 31; return v < 1 ? 1 : (v > 31 ? 31 : v); }();
        k.seed_prefix_sort = [&]() -> bool { const char* e = getenv("AC_SEED_PREFIX_SORT"); return e ? atoi(e) != 0 : true; }();
        k.seed_max_group = [&]() -> u32 { const char* e = getenv("AC_SEED_MAX_GROUP"); int v = e ? atoi(e) : 1024; return (u32)(v < 1 ? 1 : v); }();
        k.seed_prefix_bits = [&]() -> int { const char* e = getenv("AC_SEED_PREFIX_BITS"); if (!e) return 0; int v = atoi(e); return v < 1 ? 1 : (v > 64 ? 64 : v); }();
        k.degree_region_cap = [&]() -> u32 { const char* e = getenv("AC_DEGREE_REGION_CAP"); int v = e ? atoi(e) : 0; return (u32)(v < 0 ? 0 : v); }();
        k.upload_direct_mode = [&]() -> int { const char* e = getenv("AC_UPLOAD_DIRECT"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
        k.upload_slots = [&]() -> int { const char* e = getenv("AC_UPLOAD_SLOTS"); int v = e ? atoi(e) : 1 << 20; return v < 1 ? 1 : v; }();
        k.degree_flags = [&]() -> int { const char* e = getenv("AC_DEGREE_FLAGS"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }();
        k.table_shift = [&]() -> int { const char* e = getenv("AC_TABLE_SHIFT"); if (!e) return -1; int v = atoi(e); return v < 0 ? 0 : (v > 3 ? 3 : v); }();
        k.wave_chunk_max = [&]() -> u64 { const char* e = getenv("AC_INSERT_CHUNK"); u64 x = e ? (u64)atoll(e) : 8192; return (std::max<u64>(x, 256) + 63) & ~63ULL; }();
        k.insert_chunk_rest_env = [&]() -> u64 { const char* e = getenv("AC_INSERT_CHUNK_REST"); if (!e) return 0; return (std::max<u64>((u64)atoll(e), 256) + 63) & ~63ULL; }();
        k.host_pack = [&]() -> bool { const char* e = getenv("AC_HOST_PACK"); return e ? atoi(e) != 0 : true; }();
        k.expand_level_table = [&]() -> u32 { const char* e = getenv("AC_EXPAND_LEVEL_TABLE"); int v = e ? atoi(e) : 1024; return (u32)(v < 1 ? 1 : v); }();
        k.shard_path_copy = [&]() -> bool { const char* e = getenv("AC_SHARD_PATH_COPY"); return !(e && atoi(e) == 0); }();
        k.expand_rewrite_always = [&]() -> bool { return getenv("AC_EXPAND_REWRITE_ALWAYS") != nullptr; }();
        k.expand_sparse_max = [&]() -> u32 { const char* e = getenv("AC_EXPAND_SPARSE_MAX"); int v = e ? atoi(e) : 256; return (u32)(v < 0 ? 0 : v); }();
        k.late_copies = [&]() -> int { const char* e = getenv("AC_LATE_COPIES"); return e ? atoi(e) : 1; }();
        k.seq_codes_transfer = [&]() -> int { const char* e = getenv("AC_SEQ_CODES"); return e ? atoi(e) : 1; }();
        k.stretch_device_share = [&]() -> u32 { const char* e = getenv("AC_STRETCH_DEVICE_SHARE"); int v = e ? atoi(e) : 30; return (u32)(v < 0 ? 0 : v); }();
        k.expand_sparse_list = [&]() -> u32 { const char* e = getenv("AC_EXPAND_SPARSE_LIST"); int v = e ? atoi(e) : 0; return (u32)(v < 0 ? 0 : v); }();
        k.expand_sparse_batch = [&]() -> u32 { const char* e = getenv("AC_EXPAND_SPARSE_BATCH"); int v = e ? atoi(e) : 4096; return (u32)(v < 1 ? 1 : v); }();
        k.seq_writer_plain = [&]() -> bool { const char* e = getenv("AC_SEQ_WRITER"); return e && atoi(e) == 0; }();
        k.seq_writer_forced = [&]() -> bool { const char* e = getenv("AC_SEQ_WRITER"); return e && atoi(e) == 1; }();
        k.seed_radix_limit = [&]() -> u64 { const char* e = getenv("AC_SEED_RADIX_LIMIT"); return e ? (u64)atoll(e) : (1u << 19); }();
        k.upload_overlap = [&]() -> bool { const char* e = getenv("AC_UPLOAD_OVERLAP"); return e ? atoi(e) != 0 : true; }();
        k.sort_checks_deferrable = [&]() -> bool { const char* e = getenv("AC_SORT_CHECKS"); return !(e && atoi(e) == 1); }();
        k.shard_host_remap = [&]() -> bool { const char* e = getenv("AC_SHARD_HOST_REMAP"); return e ? atoi(e) != 0 : true; }();
        k.shard_degree_flags = [&]() -> bool { const char* e = getenv("AC_SHARD_DEGREE_FLAGS"); return e ? atoi(e) != 0 : true; }();
        k.insert_adaptive = [&]() -> bool { const char* e = getenv("AC_INSERT_ADAPT"); return e ? atoi(e) != 0 : true; }();
        k.insert_growth = [&]() -> u64 { const char* e = getenv("AC_INSERT_GROWTH"); long x = e ? atol(e) : 2; return (u64)(x < 2 ? 2 : x); }();
        k.insert_waves_target = [&]() -> u64 { const char* e = getenv("AC_INSERT_WAVES"); long x = e ? atol(e) : 16384; return (u64)(x < 1024 ? 1024 : x); }();
        k.renum_two_pass = [&]() -> bool { const char* e = getenv("AC_RENUM_TWO_PASS"); return e && atoi(e) != 0; }();
        k.renum_max_group = [&]() -> u32 { const char* e = getenv("AC_RENUM_MAX_GROUP"); int v = e ? atoi(e) : 64; return (u32)(v < 1 ? 1 : (v > 64 ? 64 : v)); }();
        k.path_chunk_env = [&]() -> u32 { const char* e = getenv("AC_PATH_CHUNK"); if (!e) return 0; int v = atoi(e); return (u32)(v < 64 ? 64 : (v > 4096 ? 4096 : v)); }();
        k.upload_threads_env = [&]() -> long { const char* e = getenv("AC_UPLOAD_THREADS"); return e ? atol(e) : 24; }();
        k.debug_arena = [&]() -> bool { return getenv("AC_DEBUG_ARENA") != nullptr; }();
        k.degree_diag = [&]() -> bool { return getenv("AC_DEGREE_DIAG") != nullptr; }();
        k.multi_transport = [&]() -> int { const char* e = getenv("AC_MULTI_TRANSPORT"); return !e ? 0 : (!strcmp(e, "host") ? 1 : (!strcmp(e, "rccl") ? 2 : 0)); }();
        k.multi_fragments = [&]() -> int { const char* e = getenv("AC_MULTI_FRAGMENTS"); return e && !strcmp(e, "bytes") ? 1 : 0; }();
        k.multi_tail = [&]() -> int { const char* e = getenv("AC_MULTI_TAIL"); return e && !strcmp(e, "replicated") ? 1 : 0; }();
        return k;
    }
};
const Knobs& knobs();      // (graph_build.hip)
void knobs_refresh();        // re-reads the environment if AC_TUNING_FOLLOW_ENV was set when the process first asked

[[maybe_unused]] static int minkey_variant() { return knobs().minkey_variant; }      // -1 = automatic
// AC_PATH_COPY: 0 never / 1 whenever a redundant text has a one-launch rest / unset: when the cost model below says it pays
[[maybe_unused]] static int path_copy() { return knobs().path_copy; }
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
[[maybe_unused]] static u64 run_piece() { return knobs().run_piece; }      // positions per copied piece of a run (RunFilterFunctor)
// AC_POS_CAP: occurrences further than this from both ends of their sequence do not lower a unitig's smallest positions (0 = all do)
[[maybe_unused]] static u32 pos_cap() { return knobs().pos_cap; }
[[maybe_unused]] static bool path_filter() { return knobs().path_filter; }      // smallest positions only for possible expand_repeats destinations
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
    if (knobs().path_chunk_env) return knobs().path_chunk_env;
    const u64 want = 5 * n_kmers / std::max<u32>(n_unitigs, 1);
    u32 pc = 64;
    while (pc < 2048 && (u64)pc * 3 / 2 < want) pc *= 2;
    return pc;
}
// Path entries leave the device in seed numbers right after the walk and get their final numbers on the host (single-device builds):
// 1 always, 0 never, 2 = always as STRETCHES (round 6; tests), otherwise when the number table (4 bytes per unitig) stays in the host's caches — up to 8 M unitigs — and there is
// enough to hide.  Measured (r10p/q): config C 4.50 -> 3.93 ms, E' 18.5 -> 17.7, mini-E (6.5 M unitigs) 69.8 -> 64.5; with 26 M unitigs
// (8 species) 277 -> 321 ms and with 82 M (configs[4]) 0.89 -> 1.29 s: random gathers from a table in DRAM are slower than the link.
[[maybe_unused]] static int host_remap_mode() { return knobs().host_remap_mode; }
[[maybe_unused]] static u32 remap_block() { return knobs().remap_block; }
[[maybe_unused]] static int minkey_prefix_bases() { return knobs().minkey_prefix_bases; }      // tests: a shorter prefix takes the full-key path often
[[maybe_unused]] static bool seed_prefix_sort() { return knobs().seed_prefix_sort; }      // 0: seed order by the full-key sorts
[[maybe_unused]] static u32 seed_max_group() { return knobs().seed_max_group; }      // tests: smaller groups take the fallback
[[maybe_unused]] static int seed_prefix_bits() { return knobs().seed_prefix_bits; }      // tests; unset = 0 = automatic
[[maybe_unused]] static u32 degree_region_cap() { return knobs().degree_region_cap; }      // tests: entries per queue region (0 = sized from N)
[[maybe_unused]] static u64 upload_chunk_bytes() { return (u64)64 << 20; }      // text bytes per upload chunk (16 MB of codes per copy; 8-32 MB chunks over 2-3 copy queues: 2.8 against 3.2 ms in tools/microbench/upload_probe.hip, nothing in the build: r10o)
// The packers write the codes straight into device memory (through the PCIe BAR, write-combined) instead of into a pinned ring a copy
// engine then reads: 1 / 0 forces / forbids, otherwise on when the device says its whole memory is host-visible (hipDeviceAttributeIsLargeBar).
[[maybe_unused]] static int upload_direct_mode() { return knobs().upload_direct_mode; }
[[maybe_unused]] static int upload_slots() { return knobs().upload_slots; }      // tests: fewer staging slots, so that chunks wait for one
[[maybe_unused]] static int degree_flags() { return knobs().degree_flags; }      // 0: every degree by probing (what sharded builds and k < 3 do)
[[maybe_unused]] static int table_shift() { return knobs().table_shift; }      // -1 = automatic
[[maybe_unused]] static u64 wave_chunk_max() { return knobs().wave_chunk_max; }
[[maybe_unused]] static u64 wave_chunk_rest() { return 16384; }   // longest chunk of the one-launch rest (r04c, config C: 4096 / 8192 / 16384 = 0.90 / 0.84 / 0.82 ms); shorter where the rest brings new content
[[maybe_unused]] static u64 insert_chunk_rest_env() { return knobs().insert_chunk_rest_env; }      // measurement / tests: the chunk of the rest, whatever the sample says   // longest chunk of the one-launch rest (r04c: 4096 / 8192 / 16384 = 0.90 / 0.84 / 0.82 ms)
static thread_local int tl_upload_threads_cap = 0;      // a rank of a multi-device build: its share of the host's cores (0 = no cap)
[[maybe_unused]] static u64 upload_threads() {

    // 24 since round 5: 16 / 24 / 32 threads pack config C at the same median (5.9-6.0 ms per build, two 64-core sockets), but with 32 one step
    // in twenty waits 10-20 ms for a straggler (unpinned threads on a shared host): mean 6.5-6.8 ms against 5.95-6.07 (r13b)
    long x = knobs().upload_threads_env;
    if (tl_upload_threads_cap > 0 && x > tl_upload_threads_cap) x = tl_upload_threads_cap;
    return (u64)(x < 1 ? 1 : (x > 128 ? 128 : x));
}   // host threads laying out / packing the text (the byte upload uses at most 8)
[[maybe_unused]] static bool host_pack() { return knobs().host_pack; }      // 0: upload the text as bytes and pack on the device
[[maybe_unused]] static bool insert_profile() { static const bool v = getenv("AC_INSERT_PROFILE") != nullptr; return v; }      // measurement only
[[maybe_unused]] static u32 expand_level_table() { return knobs().expand_level_table; }      // tests: a table too small for the levels
[[maybe_unused]] static bool shard_path_copy() { return knobs().shard_path_copy; }      // 0 = a sharded build walks all of its text (rounds 3-4)
[[maybe_unused]] static u32 expand_sparse_max() { return knobs().expand_sparse_max; }      // expand_repeats: at most this many dirty junctions for the one-workgroup tail (0: level launches to the end)
[[maybe_unused]] static int late_copies() { return knobs().late_copies; }      // the copies of the last results: 0 = queued behind an event of stream 0, 1 = issued by the host when that event has fired (from 256 MB of late results), 2 = always (tests)
[[maybe_unused]] static int seq_codes_transfer() { return knobs().seq_codes_transfer; }      // 0 = the unitig sequences cross the link as bytes (rounds 1-5), 1 = as 2-bit codes where they are >= 32 MB and most of the late results, 2 = always (tests)
[[maybe_unused]] static u32 stretch_device_share() { return knobs().stretch_device_share; }      // paths sent as stretches: per cent of the entries (the last ones) the device renumbers and sends itself (0: the host writes all of them)
[[maybe_unused]] static u32 expand_sparse_list() { return knobs().expand_sparse_list; }      // tests: the list length at which that tail hands back to the level launches (0: 8 x the start limit)
[[maybe_unused]] static u32 expand_sparse_batch() { return knobs().expand_sparse_batch; }      // tests: junctions of one level the tail stages in LDS (more: straight from the list)
[[maybe_unused]] static bool expand_rewrite_always() { return knobs().expand_rewrite_always; }      // tests: compact the expand pool after every host check
[[maybe_unused]] static bool seq_writer_plain() { return knobs().seq_writer_plain; }      // 0 = always the search-per-thread writers
[[maybe_unused]] static bool seq_writer_forced() { return knobs().seq_writer_forced; }      // 1 = always the indexed / LDS-tiled writers
[[maybe_unused]] static u64 seed_radix_limit() { return knobs().seed_radix_limit; }      // unitigs from which the seed order is a radix sort
[[maybe_unused]] static bool upload_overlap() { return knobs().upload_overlap; }      // host entry: first insert phases while the upload's tail is in flight
[[maybe_unused]] static bool sort_checks_deferrable() { return knobs().sort_checks_deferrable; }      // 1 = every "group too large" flag read where it is raised (round 4)
[[maybe_unused]] static bool shard_host_remap() { return knobs().shard_host_remap; }      // 0 = sharded builds renumber their paths on the device (round 4)
[[maybe_unused]] static bool shard_degree_flags() { return knobs().shard_degree_flags; }      // 0 = sharded builds probe every degree (round 4)
[[maybe_unused]] static bool insert_adaptive() { return knobs().insert_adaptive; }
[[maybe_unused]] static u64 insert_growth() { return knobs().insert_growth; }      // phase i+1 ends at growth x the end of phase i
[[maybe_unused]] static u64 insert_waves_target() { return knobs().insert_waves_target; }      // wavefronts a long phase is cut into

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
    // the walk's own numbering is TEXT order (kernels_paths.inc): its successor table, and its outputs before walk_to_seed_order
    DBuf<u64> wl_text; DBuf<u32> depth_u, minpos_fwd_u, minpos_rev_u, run_start, run_end;
    void walk_arrays();                             // the working arrays (allocated, their clears queued)
    void walk_tables(bool filter);                  // wl_text, the smallest positions' start value
    void walk_to_seed_order();                      // depth / smallest positions by seed-order index, for the tail
    bool walked_plain = false;                      // the plain walk ran (its stretch counters hold depth), not the copying one
    DBuf<u8> fs0, fe0;
    // single-device builds: smallest positions beyond it are kept as a lower bound only (kernels_tail.inc exp_avoid_start_of_path); all ones = exact
    u32 pos_cap_now = 0xFFFFFFFFu; bool exact_positions = false;
    bool host_remap_allowed = false;      // GraphBuilder::build, and a rank of a sharded build that keeps its own paths: the result block of the paths is this build's own
    bool paths_in_seed_numbers = false;   // the tail left ent_val as the walk wrote it — in TEXT-order numbers since round 6 — (the host renumbered the copy it took)
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
    void occupancy_bitmap(const DBuf<u64>& sl, u64 c, DBuf<u64>* occ_out, const u64* sflags_in, u64* sib_out, stream_t s = 0);
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

// Slot-occupancy bitmap of a finished table (one ballot word per wavefront of the scan; no atomics).
inline void GraphBuilder::Impl::occupancy_bitmap(const DBuf<u64>& sl, u64 c, DBuf<u64>* occ_out, const u64* sflags_in, u64* sib_out, stream_t s) {
    occ_out->alloc((c + 63) / 64);
#ifdef AC_EMU
    occ_out->fill_bytes(0);      // the serial emulation ORs bit by bit; the device writes whole ballot words
#endif
    launch_full(c, MarkFunctor{sl.ptr(), occ_out->ptr(), sflags_in, sib_out}, s);
}
// K3: novel-position bitmap -> sorted novel list + rank support.  known_n = the number of set bits if the caller knows it (the
// single-device insert counted its claims), 0 = count them here.
inline void GraphBuilder::Impl::novel_list(u64 known_n) {
    PackedText& g = *G;
    u64 n_bm_words = g.n_text / 64 + 1;
    // (round 6 measured the counts as the scan's computed SOURCE — no array, no kernel of their own — and took it back: a scan thread owns 16
    // consecutive items, so computed items are gathered with a 128-byte stride between lanes: config C 0.21 ms against 0.065 for the two kernels, r14c)
    DBuf<u32> wcnt(n_bm_words + 1);
    wprefix.alloc(n_bm_words + 1);
    launch(n_bm_words + 1, PopcFunctor{bm.ptr(), wcnt.ptr(), n_bm_words});
    exclusive_scan_u32(wcnt.ptr(), wprefix.ptr(), n_bm_words + 1);      // ([n_bm_words] = the total)
    if (known_n) N = known_n;
    else {
        N = read_scalar(wprefix.ptr() + n_bm_words);
        if (N == 0) throw DeviceError("internal error: no k-mers in the union of the shards");
        tm->n_distinct = N;
    }
    npos.alloc(N);
    launch_wave_kernel(fill_novel_wave_kernel<0>, (n_bm_words + 255) / 256, 0, (const u64*)bm.ptr(), (const u32*)wprefix.ptr(), npos.ptr(), n_bm_words);
    kinfo.alloc(N, true);
    // (K7's arrays here, so that their one-entry tails are cleared in the batch that clears kinfo: HeadFunctor and the scan write 0 .. N - 1)
    head.alloc(N + 1); scan.alloc(N + 1);
    head.fill_bytes_from(N * 4, 0); scan.fill_bytes_from(N * 4, 0);
    lap(&tm->collect_sort);
}

// The walk's tables in text-order numbering (kernels_paths.inc).  walk_arrays: the working arrays, allocated (and their clears queued) before
// the stage's first launch; walk_tables: the successor table permuted (+ the smallest positions' start value).
inline void GraphBuilder::Impl::walk_arrays() {
    wl_text.alloc((u64)U * 10);
    depth_u.alloc(U, true); minpos_fwd_u.alloc(U); minpos_rev_u.alloc(U);
    run_start.alloc((u64)U + 1, true); run_end.alloc((u64)U + 1, true);
}
inline void GraphBuilder::Impl::walk_tables(bool filter) {
    launch((u64)U * 10, WlinkFlagFunctor{filter ? maybe_dest.ptr() : nullptr, wlinks.ptr(), counters.ptr() + 4});
    const u32 pos_init = pos_cap_now == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((pos_cap_now + 1) | POS_BOUND);
    launch((u64)U * 10, WlinkTextOrderFunctor{wlinks.ptr(), rank.ptr(), order.ptr(), uorient.ptr(), wl_text.ptr(), minpos_fwd_u.ptr(), minpos_rev_u.ptr(), pos_init});
}
inline void GraphBuilder::Impl::walk_to_seed_order() {
    depth.alloc(U); minpos_fwd.alloc(U); minpos_rev.alloc(U);
    DBuf<u32> started, ended;
    if (walked_plain) {
        started.alloc((u64)U + 1); ended.alloc((u64)U + 1);
        inclusive_scan_u32(run_start.ptr(), started.ptr(), (u64)U + 1);      // stretches that began at or before u
        exclusive_scan_u32(run_end.ptr(), ended.ptr(), (u64)U + 1);          // ... and those that ended before u
    }
    launch(U, WalkToSeedOrderFunctor{order.ptr(), depth_u.ptr(), walked_plain ? started.ptr() : nullptr, walked_plain ? ended.ptr() : nullptr,
                                     minpos_fwd_u.ptr(), minpos_rev_u.ptr(), depth.ptr(), minpos_fwd.ptr(), minpos_rev.ptr()});
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

}  // namespace ac
