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
#include "graph_build.hpp"

#include <chrono>
#include <map>
#include <string>
#include <algorithm>

#include "device_rt.hpp"

namespace ac {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}



static const int MAX_PROBES = 1 << 14;
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
};

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
    u64 h = key_hash<W>(ukey);
    u64 tag = slot_make(h, isdot, 0);
    u64 s = h & tb.cap_mask;
    FindResult r; r.found = false; r.pos = 0; r.claimant_flipped = 0;
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
template <int W> AC_D u64 table_insert(const TextCtx& t, const Table& tb, const Key<W>& ukey, bool isdot, bool flipped, u64 p,
                                       u32* claimed, u32* err, bool* same) {
    u64 h = key_hash<W>(ukey);
    u64 mine = slot_make(h, isdot, p);
    u64 s = h & tb.cap_mask;
    for (int probes = 0; probes < MAX_PROBES; probes++) {
        u64 v = tb.slots[s];
        if (v == SLOT_EMPTY) {
            u64 old = atomic_cas64(&tb.slots[s], SLOT_EMPTY, mine);
            if (old == SLOT_EMPTY) { (*claimed)++; return NOREF; }
            v = old;
        }
        if (slot_tag_eq(v, mine)) {
            if (slot_pos(v) == p) return NOREF;
            int m = claimant_match<W>(t, v, ukey);
            if (m) {
                if (slot_pos(v) > p) { atomic_min64(&tb.slots[s], mine); return NOREF; }
                *same = ((m == 2) == flipped);
                return slot_pos(v);
            }
        }
        s = (s + 1) & tb.cap_mask;
    }
    atomic_or32(err, 1u);
    return NOREF;
}

struct alignas(16) V16 { u32 a, b, c, d; };

// ---- K1: ASCII text -> 2-bit words + mask ------------------------------------------------------------
// One thread per 32 text bytes (two 16-byte loads; a wavefront reads 2 KB contiguously), writing one
// 64-bit word of bases and the matching 32-bit half of a mask word.
struct PackFunctor {
    const u8* text; u64 n_text; u64* bits; u32* mask32;
    AC_HD void operator()(u64 tid) const {
        u64 base = tid * 32;
        u64 w = 0; u32 m = 0;
        alignas(16) u8 buf[32];
        if (base + 32 <= n_text && ((uintptr_t)(text + base) & 15) == 0) {
            const V16* src = (const V16*)(text + base);
            *(V16*)(buf) = src[0];
            *(V16*)(buf + 16) = src[1];
        } else {
            for (int i = 0; i < 32; i++) buf[i] = (base + (u64)i < n_text) ? text[base + (u64)i] : (u8)'$';
        }
#pragma unroll
        for (int i = 0; i < 32; i++) {
            u32 ch = buf[i];
            u32 bad = !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
            u32 c = bad ? 0u : (((ch >> 1) ^ (ch >> 2)) & 3u);
            w |= (u64)c << (62 - 2 * i);
            m |= bad << i;
        }
        bits[tid] = w; mask32[tid] = m;
    }
};

// ---- K2: run-following insert of every k-mer occurrence (kmer_graph.rs:103-133) -----------------------
// The table only has to end up holding, for every canonical k-mer, its SMALLEST text position.  A position p
// whose k-mer equals the k-mer at an earlier position q can therefore be skipped.  After a real insert at p
// has met an earlier occurrence q, position p+1+i is skippable as long as base[p+k+i] agrees with the base
// that extends q the same way (base[q+k+i], or comp(base[q-1-i]) when p reads q's reverse complement): both
// windows are then identical real k-mers and q+1+i < p+1+i.  The smallest occurrence of a k-mer never has an
// earlier one, so it is always inserted for real, which is all the final table state depends on.
// Each thread owns `chunk` consecutive positions of [p_begin, p_end); the host launches the kernel over
// geometrically growing prefixes of the text so that later phases find the earlier ones in the table.
struct InsertStats { u64 real; u64 claimed; };
template <int W> struct InsertFunctor {
    TextCtx t; Table tb; u64 p_begin, p_end; u32 chunk; InsertStats* stats; u32* err;
    AC_D void operator()(u64 tid) const {
        const int k = t.k;
        u64 p0 = p_begin + tid * (u64)chunk;
        if (p0 >= p_end) return;
        u64 p1 = p0 + (u64)chunk;
        if (p1 > p_end) p1 = p_end;
        u32 claimed = 0, real = 0;
        u64 p = p0;
        while (p < p1) {
            real++;
            if (text_mask_count(t.mask, p, k) == 0) {
                Key<W> fwd = text_extract<W>(t.bits, p, k);
                Key<W> rc = key_rc<W>(fwd, k);
                bool flipped = key_lt<W>(rc, fwd);
                Key<W> uk = flipped ? rc : fwd;
                uk.w[0] |= (u64)255 << 56;
                bool same = false;
                u64 q = table_insert<W>(t, tb, uk, false, flipped, p, &claimed, err, &same);
                if (q != NOREF) {
                    u64 maxlen = p1 - 1 - p;
                    u64 run = same ? match_run_fwd(t.bits, t.mask, p + (u64)k, q + (u64)k, maxlen)
                                   : match_run_rev(t.bits, t.mask, p + (u64)k, q - 1, maxlen);
                    p += run;
                }
            } else {
                XKmer<W> x;
                if (xkmer_at<W>(t, p, &x)) {    // else: the window crosses a separator
                    bool flipped, same;
                    Key<W> uk = xk_canonical<W>(x, k, &flipped);
                    table_insert<W>(t, tb, uk, true, flipped, p, &claimed, err, &same);
                }
            }
            p++;
        }
        InsertStats* st = stats + ((tid >> 6) & 255);
        atomic_add64(&st->real, (u64)real);
        if (claimed) atomic_add64(&st->claimed, (u64)claimed);
    }
};

// ---- K2w: the same insert, one WAVEFRONT per chunk ---------------------------------------------------------------
// The thread-per-chunk kernel above follows runs with one lane reading two private streams: every 8-byte load is its
// own memory transaction (PMC: 4.8 GB for 0.37 GB of packed text).  Here the 64 lanes of a wavefront share one chunk:
//   A. lanes insert positions p .. p+63 for real (one k-mer each: hash, probe, CAS / atomicMin);
//   B. if any lane met an EARLIER occurrence q of its k-mer, the run is followed from the last such lane: lane i
//      compares the 32-base word at offset 32 i of the text after p with the word the earlier occurrence continues
//      with (or its reverse-complement view) — 2048 positions per step from two coalesced 512-byte reads — and a
//      ballot finds the first disagreement.  Every position inside the verified run is an occurrence of a k-mer
//      that has an earlier occurrence, so it needs no table access (same argument as above).
// The table ends in the same state: every canonical k-mer's slot holds its smallest text position.
template <int W> AC_D u64 insert_one(const TextCtx& t, const Table& tb, u64 p, u32* claimed, u32* err, bool* same) {
    const int k = t.k;
    if (text_mask_count(t.mask, p, k) == 0) {
        Key<W> fwd = text_extract<W>(t.bits, p, k);
        Key<W> rc = key_rc<W>(fwd, k);
        bool flipped = key_lt<W>(rc, fwd);
        Key<W> uk = flipped ? rc : fwd;
        uk.w[0] |= (u64)255 << 56;
        return table_insert<W>(t, tb, uk, false, flipped, p, claimed, err, same);
    }
    XKmer<W> x;
    if (xkmer_at<W>(t, p, &x)) {    // else: the window crosses a separator
        bool flipped, sm;
        Key<W> uk = xk_canonical<W>(x, k, &flipped);
        table_insert<W>(t, tb, uk, true, flipped, p, claimed, err, &sm);
    }
    return NOREF;                   // dot k-mers sit at sequence ends: never worth following
}
// Lane `lane`'s view of one verification step: matching bases (0..32) of its word, 0 when its word lies beyond the run limit.
AC_HD int wave_match(const TextCtx& t, u64 a, u64 b, bool same, u64 off, u64 maxlen) {
    if (off >= maxlen) return 0;
    if (same) return match_word_fwd(t.bits, t.mask, a + off, b + off);
    if (off > b) return 0;          // ran off the start of the text (position 0 is a separator, so unreachable)
    return match_word_rev(t.bits, t.mask, a + off, b - off);
}
#ifndef AC_EMU
template <int W>
__global__ void __launch_bounds__(256) insert_wave_kernel(TextCtx t, Table tb, u64 p_begin, u64 p_end, u32 chunk, InsertStats* stats, u32* err) {
    const int lane = (int)(threadIdx.x & 63);
    const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const u64 c0 = p_begin + wave * (u64)chunk;
    if (c0 >= p_end) return;                                   // wave-uniform
    const u64 c1 = (c0 + chunk < p_end) ? c0 + chunk : p_end;
    const int k = t.k;
    u32 claimed = 0, real = 0;
    u64 p = c0;
    while (p < c1) {
        const u64 pi = p + (u64)lane;
        u64 q = NOREF; bool same = false;
        if (pi < c1) { real++; q = insert_one<W>(t, tb, pi, &claimed, err, &same); }
        const u64 hits = __ballot(q != NOREF);
        u64 next = p + 64;
        if (hits) {
            const int jl = 63 - __clzll((long long)hits);
            const u64 qj = (u64)__shfl((unsigned long long)q, jl);
            const bool sj = __shfl((int)same, jl) != 0;
            const u64 pj = p + (u64)jl;
            const u64 maxlen = c1 - 1 - pj;
            const u64 a = pj + (u64)k, b = sj ? qj + (u64)k : qj - 1;
            u64 n = 0;
            while (n < maxlen) {
                int tmatch = wave_match(t, a, b, sj, n + 32 * (u64)lane, maxlen);
                const u64 bal = __ballot(tmatch < 32);
                if (bal == 0) { n += 2048; continue; }
                const int f = __ffsll((long long)bal) - 1;
                n += 32 * (u64)f + (u64)__shfl(tmatch, f);
                break;
            }
            if (n > maxlen) n = maxlen;
            if (pj + 1 + n > next) next = pj + 1 + n;
        }
        p = next;
    }
    // one pair of atomics per wavefront (64 lanes adding to one address would serialise in L2)
    for (int o = 32; o; o >>= 1) { real += (u32)__shfl_xor((int)real, o); claimed += (u32)__shfl_xor((int)claimed, o); }
    if (lane == 0) {
        InsertStats* st = stats + (wave & 255);
        atomic_add64(&st->real, (u64)real);
        if (claimed) atomic_add64(&st->claimed, (u64)claimed);
    }
}
#endif
// The wavefront kernel's logic with the 64 lanes visited one after the other (CPU emulation: tests only; also the
// reference the device kernel is read against).
template <int W> struct InsertWaveEmuFunctor {
    TextCtx t; Table tb; u64 p_begin, p_end; u32 chunk; InsertStats* stats; u32* err;
    AC_D void operator()(u64 wave) const {
        const u64 c0 = p_begin + wave * (u64)chunk;
        if (c0 >= p_end) return;
        const u64 c1 = (c0 + chunk < p_end) ? c0 + chunk : p_end;
        const int k = t.k;
        u32 claimed = 0, real = 0;
        u64 p = c0;
        while (p < c1) {
            u64 q[64]; bool same[64];
            int jl = -1;
            for (int lane = 0; lane < 64; lane++) {
                q[lane] = NOREF; same[lane] = false;
                if (p + (u64)lane < c1) { real++; q[lane] = insert_one<W>(t, tb, p + (u64)lane, &claimed, err, &same[lane]); }
                if (q[lane] != NOREF) jl = lane;
            }
            u64 next = p + 64;
            if (jl >= 0) {
                const u64 pj = p + (u64)jl;
                const u64 maxlen = c1 - 1 - pj;
                const u64 a = pj + (u64)k, b = same[jl] ? q[jl] + (u64)k : q[jl] - 1;
                u64 n = 0;
                while (n < maxlen) {
                    int f = -1, tf = 0;
                    for (int lane = 0; lane < 64 && f < 0; lane++) {
                        int tmatch = wave_match(t, a, b, same[jl], n + 32 * (u64)lane, maxlen);
                        if (tmatch < 32) { f = lane; tf = tmatch; }
                    }
                    if (f < 0) { n += 2048; continue; }
                    n += 32 * (u64)f + (u64)tf;
                    break;
                }
                if (n > maxlen) n = maxlen;
                if (pj + 1 + n > next) next = pj + 1 + n;
            }
            p = next;
        }
        InsertStats* st = stats + (wave & 255);
        atomic_add64(&st->real, (u64)real);
        if (claimed) atomic_add64(&st->claimed, (u64)claimed);
    }
};

// ---- K3: novel-position bitmap -> sorted novel list + rank support ---------------------------------------
struct MarkFunctor {      // also writes the slot-occupancy bitmap (one ballot word per wavefront; `occ` zeroed beforehand)
    const u64* slots; u32* bm32; u64* occ;
    AC_D void operator()(u64 s, bool valid) const {
        u64 v = valid ? slots[s] : SLOT_EMPTY;
        bool full = v != SLOT_EMPTY;
        if (full) { u64 pos = slot_pos(v); atomic_or32(&bm32[pos >> 5], 1u << (pos & 31)); }
#ifdef AC_EMU
        if (full) occ[s >> 6] |= 1ULL << (s & 63);
#else
        u64 b = __ballot(full);
        if ((s & 63) == 0 && valid) occ[s >> 6] = b;
#endif
    }
};
struct PopcFunctor {
    const u64* bm; u32* cnt;
    AC_HD void operator()(u64 w) const { cnt[w] = (u32)popc64(bm[w]); }
};
struct FillNovelFunctor {
    const u64* bm; const u32* wprefix; u64* npos;
    AC_HD void operator()(u64 w) const {
        u64 x = bm[w];
        u32 i = wprefix[w];
        while (x) {
            u64 low = x & (~x + 1);
            npos[i++] = w * 64 + (u64)popc64(low - 1);
            x ^= low;
        }
    }
};

// ---- K5: out/in degrees per distinct k-mer (kmer_graph.rs:136-166) -------------------------------------
// `known` (0..4 or -1): a successor symbol already known to be in the set (the k-mer that follows / precedes
// this one in the text), counted without a probe.
template <int W> AC_D int count_successors(const TextCtx& t, const Table& tb, const XKmer<W>& x, int max_c, int known) {
    int n = 0;
    for (int c = 0; c < max_c; c++) {
        XKmer<W> y;
        if (!xk_next<W>(x, t.k, c, &y)) continue;
        if (c == known) { n++; continue; }
        u64 pos; bool rel;
        if (find_xk<W>(t, tb, y, &pos, &rel)) n++;
    }
    return n;
}
// Is the real (dot-free) k-mer with strands (f, r) in the set?
template <int W> AC_D bool real_kmer_exists(const TextCtx& t, const Table& tb, const Key<W>& f, const Key<W>& r) {
    Key<W> uk = key_lt<W>(r, f) ? r : f;
    uk.w[0] |= (u64)255 << 56;
    return table_find<W>(t, tb, uk, false).found;
}
// The four real successors of a real k-mer with strands (fwd, rc): both strands of a candidate follow from the parent's
// by one rolling step each — no reverse complement per candidate (the degree kernel was instruction-bound on those).
template <int W> AC_D int count_real_successors(const TextCtx& t, const Table& tb, const Key<W>& fwd, const Key<W>& rc, int known) {
    const Key<W> km = key_kmask<W>(t.k);
    int n = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (c == known) { n++; continue; }
        Key<W> f = fwd, r = rc;
        key_roll_fwd<W>(f, (u32)c, km);
        key_roll_rc<W>(r, (u32)c, t.k);
        if (real_kmer_exists<W>(t, tb, f, r)) n++;
    }
    return n;
}
template <int W> struct DegreeFunctor {
    TextCtx t; Table tb; const u64* npos; u32* kinfo; int any_dots; u64 first;   // handles novel indices first, first+1, ...
    AC_D void operator()(u64 i) const {
        i += first;
        XKmer<W> x;
        u64 p = npos[i];
        int known_out = -1, known_in = -1;
        int out, in;
        if (text_mask_count(t.mask, p, t.k) == 0) {
            x.fwd = text_extract<W>(t.bits, p, t.k); x.ld = 0; x.td = 0;
            // an unmasked neighbour base means the neighbouring window is a real k-mer of the same sequence
            if (!text_mask(t.mask, p + (u64)t.k)) known_out = (int)text_code(t.bits, p + (u64)t.k);
            if (!text_mask(t.mask, p - 1)) known_in = 3 - (int)text_code(t.bits, p - 1);
            Key<W> rc = key_rc<W>(x.fwd, t.k);
            out = count_real_successors<W>(t, tb, x.fwd, rc, known_out);
            in = count_real_successors<W>(t, tb, rc, x.fwd, known_in);       // in(X) = out(rc X), and rc(rc X) = X
            if (any_dots) {   // the '.' successor / predecessor (kmer_graph.rs:142,158): a dot k-mer, generic path
                XKmer<W> y; u64 pos; bool rel;
                if (xk_next<W>(x, t.k, 4, &y) && find_xk<W>(t, tb, y, &pos, &rel)) out++;
                XKmer<W> r = xk_rc<W>(x, t.k);
                if (xk_next<W>(r, t.k, 4, &y) && find_xk<W>(t, tb, y, &pos, &rel)) in++;
            }
        } else {
            if (!xkmer_at<W>(t, p, &x)) return;
            int max_c = any_dots ? 5 : 4;
            out = count_successors<W>(t, tb, x, max_c, known_out);
            XKmer<W> r = xk_rc<W>(x, t.k);
            in = count_successors<W>(t, tb, r, max_c, known_in);
        }
        kinfo[i] |= (u32)out | ((u32)in << KI_IN_SHIFT);
    }
};

// ---- K6: first_position flags (kmer_graph.rs:57-60): first forward k-mer of each sequence and the RC of
// its last forward k-mer sit at pos 0 of a strand.
// `flags` (sharded builds, where the "sequences" of the graph text are fragments): bit 0 = the fragment starts at a
// sequence start, bit 1 = it ends at a sequence end; nullptr = every entry is a whole sequence.
template <int W> struct FirstFunctor {
    TextCtx t; Table tb; Novel nv; u32* kinfo; const u8* flags;
    AC_D void operator()(u64 s) const {
        for (int which = 0; which < 2; which++) {
            if (flags && !(flags[s] & (which ? 2u : 1u))) continue;
            u64 p = t.seq_off[s] + (which ? (u64)t.seq_len[s] - 1 : 0);
            XKmer<W> x;
            if (!xkmer_at<W>(t, p, &x)) continue;
            u64 pos; bool rel_same;
            if (!find_xk<W>(t, tb, x, &pos, &rel_same)) continue;
            u32 j = novel_rank(nv, pos);
            // which==0: first(X) holds;  which==1: first(rc X) holds.
            bool flag_on_T = (which == 0) ? rel_same : !rel_same;
            atomic_or32(&kinfo[j], flag_on_T ? KI_FIRST_T : KI_FIRST_RCT);
        }
    }
};

// ---- K7: unitig heads among novel positions --------------------------------------------------------------
struct HeadFunctor {
    const u64* npos; const u32* kinfo; u32* head; u64 n;
    AC_HD void operator()(u64 i) const {
        u32 h = 1;
        if (i > 0 && npos[i] == npos[i - 1] + 1) {
            u32 a = kinfo[i - 1], b = kinfo[i];
            bool internal = !(a & KI_FIRST_RCT) && (a & KI_OUT_MASK) == 1 && ((b >> KI_IN_SHIFT) & 7u) == 1 && !(b & KI_FIRST_T);
            h = internal ? 0u : 1u;
        }
        head[i] = h;
    }
};
struct UnitigStartFunctor {
    const u32* head; const u32* scan; u32* ustart; u64 n;
    AC_HD void operator()(u64 i) const { if (head[i]) ustart[scan[i] - 1] = (u32)i; }
};

// ---- K8: smallest canonical k-mer per unitig ------------------------------------------------------------
template <int W> struct MinVal { Key<W> key; u32 flipped; u32 pad; };
template <int W> struct MinOp {
    AC_HD MinVal<W> operator()(const MinVal<W>& a, const MinVal<W>& b) const { return key_lt<W>(b.key, a.key) ? b : a; }
};
template <int W> struct MinValLess {
    AC_HD bool operator()(const MinVal<W>& a, const MinVal<W>& b) const { return key_lt<W>(a.key, b.key); }
};
// Canonical key of the k-mer at a novel position (all-ones if the position is not a k-mer start: cannot happen).
template <int W> AC_D Key<W> canonical_at(const TextCtx& t, u64 p, bool* flipped) {
    XKmer<W> x;
    bool ok = true;
    if (text_mask_count(t.mask, p, t.k) == 0) { x.fwd = text_extract<W>(t.bits, p, t.k); x.ld = 0; x.td = 0; }
    else ok = xkmer_at<W>(t, p, &x);
    *flipped = false;
    if (ok) return xk_canonical<W>(x, t.k, flipped);
    Key<W> bad;
#pragma unroll
    for (int j = 0; j < W; j++) bad.w[j] = ~0ULL;
    return bad;
}
// Which of two novel k-mers (given by their indices in the novel list) has the smaller canonical key?  The keys are
// recomputed from the packed text on every call: cheaper than writing and re-reading a 16..136-byte key per k-mer.
template <int W> struct MinIdxOp {
    TextCtx t; const u64* npos;
    AC_D u32 operator()(const u32& a, const u32& b) const {
        bool fa, fb;
        Key<W> ka = canonical_at<W>(t, npos[a], &fa), kb = canonical_at<W>(t, npos[b], &fb);
        return key_lt<W>(kb, ka) ? b : a;
    }
};
template <int W> struct CKeyFunctor {        // narrow keys (W <= 4): materialise (key, strand) per novel k-mer for a plain segmented min
    TextCtx t; const u64* npos; const u32* scan; MinVal<W>* vals; u32* seg;
    AC_D void operator()(u64 i) const {
        bool flipped;
        MinVal<W> v;
        v.key = canonical_at<W>(t, npos[i], &flipped);
        v.flipped = flipped ? 1u : 0u; v.pad = 0;
        vals[i] = v;
        seg[i] = scan[i] - 1;
    }
};
template <int W> struct UnitigMinFunctor {   // the winner's key and strand, per unitig
    TextCtx t; const u64* npos; const u32* umin_idx; MinVal<W>* umin;
    AC_D void operator()(u64 u) const {
        bool flipped;
        MinVal<W> v;
        v.key = canonical_at<W>(t, npos[umin_idx[u]], &flipped);
        v.flipped = flipped ? 1u : 0u; v.pad = 0;
        umin[u] = v;
    }
};
template <int W> struct MinValIdxLess {      // indirect comparison for keys too wide to be moved around by the sort
    const MinVal<W>* umin;
    AC_HD bool operator()(const u32& a, const u32& b) const { return key_lt<W>(umin[a].key, umin[b].key); }
};
struct IotaFunctor { u32* a; AC_HD void operator()(u64 i) const { a[i] = (u32)i; } };
template <int W> struct GatherMinFunctor { const u32* order; const MinVal<W>* in; MinVal<W>* out; AC_HD void operator()(u64 r) const { out[r] = in[order[r]]; } };

// ---- K9: per-unitig metadata in seed (rank) order ---------------------------------------------------------
template <int W> struct UnitigMetaFunctor {
    const u32* order; const u32* ustart; const u64* npos; const MinVal<W>* sorted_min; u32 n_unitigs; u64 n_novel;
    u32* rank; u32* ulen; u64* ulen64; u64* ustartpos; u8* uorient;
    AC_HD void operator()(u64 r) const {
        if (r == n_unitigs) { ulen64[r] = 0; return; }   // sentinel so the exclusive scan yields the total too
        u32 u = order[r];
        rank[u] = (u32)r;
        u32 a = ustart[u];
        u32 b = (u + 1 < n_unitigs) ? ustart[u + 1] : (u32)n_novel;
        ulen[r] = b - a;
        ulen64[r] = (u64)(b - a);
        ustartpos[r] = npos[a];
        uorient[r] = sorted_min[r].flipped ? 0 : 1;   // forward strand == text orientation?
    }
};

// Everything a kernel needs to turn a novel index into (unitig in seed order, strand, offsets).
struct UnitigCtx {
    const u32* head; const u32* scan; const u32* rank; const u8* uorient; const u32* ustart; const u32* ulen;
    u32 n_unitigs; u64 n_novel;
};

// ---- K11: links (unitig_graph.rs:234-287: a's (k-1)-suffix == b's (k-1)-prefix  <=>  b's first k-mer is
// a successor of a's last k-mer), stored BY SUCCESSOR SYMBOL so the path kernel can walk them ----------------
template <int W> struct LinksFunctor {
    TextCtx t; Table tb; Novel nv; UnitigCtx uc; const u32* order; const u64* npos; int any_dots;
    int32_t* links; u64* wlinks; u32* err;    // wlinks: [length of the target:32][signed number:32], one load per step of a walk
    AC_D void operator()(u64 idx) const {
        u32 r = (u32)(idx >> 1);
        int side = (int)(idx & 1);           // 0: forward strand's end, 1: reverse strand's end
        u32 u = order[r];
        u32 ia = uc.ustart[u];
        u32 ib = ((u + 1 < uc.n_unitigs) ? uc.ustart[u + 1] : (u32)uc.n_novel) - 1;
        bool o = uc.uorient[r] != 0;
        // forward strand's last k-mer: o ? B : rc(A);  reverse strand's last k-mer: o ? rc(A) : B
        bool use_b = (side == 0) ? o : !o;
        XKmer<W> e;
        if (!xkmer_at<W>(t, npos[use_b ? ib : ia], &e)) { atomic_or32(err, 2u); return; }
        if (!use_b) e = xk_rc<W>(e, t.k);
        int max_c = any_dots ? 5 : 4;
        for (int c = 0; c < 5; c++) {
            int32_t val = 0;
            u32 tlen = 0;
            XKmer<W> y;
            u64 pos; bool rel_same;
            if (c < max_c && xk_next<W>(e, t.k, c, &y) && find_xk<W>(t, tb, y, &pos, &rel_same)) {
                u32 j = novel_rank(nv, pos);
                u32 rv = uc.rank[uc.scan[j] - 1];
                bool strand = rel_same ? (uc.uorient[rv] != 0) : (uc.uorient[rv] == 0);
                bool is_head = uc.head[j] != 0;
                bool is_tail = (j + 1 == uc.n_novel) || uc.head[j + 1] != 0;
                if (!(rel_same ? is_head : is_tail)) atomic_or32(err, 4u);   // successor of an end must start a unitig strand
                val = strand ? (int32_t)(rv + 1) : -(int32_t)(rv + 1);
                tlen = uc.ulen[rv];
            }
            links[idx * 5 + (u64)c] = val;
            wlinks[idx * 5 + (u64)c] = ((u64)tlen << 32) | (u64)(u32)val;
        }
    }
};

// ---- K10: paths, depth and min positions (find_starting_unitig / get_next_unitig, unitig_graph.rs:407-465;
// simplify_seqs positions, unitig.rs:136-147) ------------------------------------------------------------------
// Every occurrence of a unitig's first k-mer is followed by the whole unitig, so a sequence
// path is walked unitig by unitig: ONE table lookup locates the walker inside its first unitig, after that the
// next unitig (and its length) is wlinks[(current strand end)][next text symbol] — one gather, no hashing.  Thread tid
// owns the unitig heads that fall into text positions [tid*PC, (tid+1)*PC) and writes them to its own staging slots;
// an exclusive scan of the counts and a compaction put them in text order without a sort or a second walk.
// `t` is the text being walked (this rank's sequences), `g` the text the graph was built from (the same text for a
// single-device build, the union of all ranks' novel fragments for a sharded one): table slots point into `g`.
template <int W> struct PathWalkFunctor {
    TextCtx t; TextCtx g; Table tb; Novel nv; UnitigCtx uc; const u64* wlinks; u32 pc;
    int32_t* stage; u64* cnt;         // stage[tid * pc + j]: j-th entry of walker tid; cnt[tid]: how many
    u32* seq_tid; u32* seq_j;         // where each sequence's path starts: (walker, index in its staging slots)
    u32* depth; u32* minpos_fwd; u32* minpos_rev; u32* err;
    AC_D void emit(u64 tid, u32& j, u64 p, u32 s, u32 r, bool strand) const {
        stage[tid * (u64)pc + j] = strand ? (int32_t)(r + 1) : -(int32_t)(r + 1);
        j++;
        atomic_add32(&depth[r], 1u);
        u32 f = (u32)(p - t.seq_off[s]);
        u32 other = t.seq_len[s] - uc.ulen[r] - f;   // position of the same occurrence on the opposite strand
        // the words only ever decrease, so a plain (possibly stale) read that is already <= ours makes the atomic
        // redundant: a unitig of depth d settles after a few of its d occurrences
        u32 vf = strand ? f : other, vr = strand ? other : f;
        if (vf < minpos_fwd[r]) atomic_min32(&minpos_fwd[r], vf);
        if (vr < minpos_rev[r]) atomic_min32(&minpos_rev[r], vr);
    }
    AC_D void operator()(u64 tid) const {
        const int k = t.k;
        u64 p0 = tid * (u64)pc;
        u32 j = 0;
        if (p0 >= t.n_text) { cnt[tid] = 0; return; }
        u64 p1 = p0 + (u64)pc;
        if (p1 > t.n_text) p1 = t.n_text;
        // first sequence whose k-mer starts are not all below p0
        u32 lo = 0, hi = t.n_seqs;
        while (lo < hi) { u32 mid = lo + ((hi - lo) >> 1); if (t.seq_off[mid] + (u64)t.seq_len[mid] <= p0) lo = mid + 1; else hi = mid; }
        u32 s = lo;
        u64 p = p0;
        while (s < t.n_seqs) {
            u64 s_begin = t.seq_off[s], s_end = s_begin + (u64)t.seq_len[s];
            if (p < s_begin) p = s_begin;
            if (p >= p1) break;
            if (p == s_begin) { seq_tid[s] = (u32)tid; seq_j[s] = j; }
            // locate the walker: which unitig strand covers the k-mer at p, and where does that unitig end here?
            XKmer<W> x;
            u64 pos; bool rel_same;
            if (!xkmer_at<W>(t, p, &x) || !find_xk<W>(g, tb, x, &pos, &rel_same)) { atomic_or32(err, 8u); break; }
            u32 jn = novel_rank(nv, pos);
            u32 u = uc.scan[jn] - 1;
            u32 a = uc.ustart[u];
            u32 b = (u + 1 < uc.n_unitigs) ? uc.ustart[u + 1] : (u32)uc.n_novel;
            u32 r = uc.rank[u];
            bool strand = rel_same ? (uc.uorient[r] != 0) : (uc.uorient[r] == 0);
            if (rel_same) { if (jn == a) emit(tid, j, p, s, r, strand); p += (u64)(b - jn); }
            else { if (jn == b - 1) emit(tid, j, p, s, r, strand); p += (u64)(jn - a + 1); }
            // walk the links
            bool bad = false;
            while (p < s_end && p < p1) {
                u64 e = p + (u64)k - 1;
                u32 c = text_mask(t.mask, e) ? 4u : text_code(t.bits, e);
                u64 wl = wlinks[((u64)r * 2 + (strand ? 0 : 1)) * 5 + c];
                int32_t val = (int32_t)(u32)wl;
                if (val == 0) { atomic_or32(err, 16u); bad = true; break; }
                strand = val > 0;
                r = (u32)(strand ? val : -val) - 1;
                emit(tid, j, p, s, r, strand);
                p += wl >> 32;
            }
            if (bad) break;
            if (p >= s_end) s++; else break;   // p >= p1
        }
        cnt[tid] = j;
    }
};
struct PathCompactFunctor {
    const int32_t* stage; const u64* cnt; const u64* woff; u32 pc; int32_t* ent_val;
    AC_HD void operator()(u64 tid) const {
        u64 n = cnt[tid], o = woff[tid];
        const int32_t* src = stage + tid * (u64)pc;
        for (u64 j = 0; j < n; j++) ent_val[o + j] = src[j];
    }
};
struct PathOffFunctor {
    const u32* seq_tid; const u32* seq_j; const u64* woff; u64* path_off;
    AC_HD void operator()(u64 s) const { path_off[s] = woff[seq_tid[s]] + (u64)seq_j[s]; }
};

// ---- K13: create_links' push order (unitig_graph.rs:248-286), a function of seed numbers only:
//   forward_next(a): all b+ (seed order ascending) then all b- (ascending);
//   reverse_next(a): a'- for a' <= a (ascending; a' == a is the a+ -> a+ self loop pushed in case 1 of iteration a),
//                    then b+ (case 3 of iteration a, ascending), then a'- for a' > a (ascending).
AC_HD u32 idx_of(int32_t v) { return (u32)(v < 0 ? -v : v) - 1; }
struct LinkOrderFunctor {
    const int32_t* sym; int32_t* ord; u8* cnt; u32* n_self_mirror;
    AC_D void operator()(u64 idx) const {
        int side = (int)(idx & 1);
        int32_t num = (int32_t)(idx >> 1) + 1;
        int32_t tmp[5]; u64 key[5]; int n = 0;
        for (int c = 0; c < 5; c++) {
            int32_t v = sym[idx * 5 + (u64)c];
            if (v == 0) continue;
            u32 cls = (side == 0) ? (v > 0 ? 0u : 1u) : (v > 0 ? 1u : ((-v <= num) ? 0u : 2u));
            u64 kv = ((u64)cls << 32) | (u64)(v < 0 ? -v : v);
            int j = n++;
            while (j > 0 && key[j - 1] > kv) { key[j] = key[j - 1]; tmp[j] = tmp[j - 1]; j--; }
            key[j] = kv; tmp[j] = v;
        }
        u32 self = 0;
        for (int i = 0; i < 5; i++) {
            ord[idx * 5 + (u64)i] = i < n ? tmp[i] : 0;
            if (i < n && tmp[i] == (side == 0 ? -num : num)) self++;
        }
        cnt[idx] = (u8)n;
        if (self) atomic_add32(n_self_mirror, self);
    }
};
struct OrderedLinks {
    const int32_t* ord; const u8* cnt;
    AC_HD const int32_t* next_of(int32_t x, u32* n) const {   // next links of a unitig strand
        u64 i = (u64)idx_of(x) * 2 + (x > 0 ? 0 : 1);
        *n = cnt[i];
        return ord + i * 5;
    }
};

// ---- K14: static analysis for expand_repeats -------------------------------------------------------------------
// Fixed starts/ends (graph_simplification.rs:190-230): first/last unitig of every sequence path plus their one-step
// neighbours; invariant across passes because paths and links never change.
struct PathEndsFunctor {
    const int32_t* path; const u64* path_off; u8* fs0; u8* fe0;
    AC_HD void operator()(u64 s) const {
        u64 b = path_off[s], e = path_off[s + 1];
        if (b == e) return;
        int32_t first = path[b], last = path[e - 1];
        if (first > 0) fs0[idx_of(first)] = 1; else fe0[idx_of(first)] = 1;
        if (last > 0) fe0[idx_of(last)] = 1; else fs0[idx_of(last)] = 1;
    }
};
struct FixedSpreadFunctor {
    const u8* fs0; const u8* fe0; OrderedLinks L; u8* fixed_start; u8* fixed_end;
    AC_HD void operator()(u64 u) const {
        int32_t num = (int32_t)u + 1;
        if (fs0[u]) {   // upstream of a fixed start: forward_prev(u) = { -e : e in reverse_next(u) }
            fixed_start[u] = 1;
            u32 n; const int32_t* p = L.next_of(-num, &n);
            for (u32 i = 0; i < n; i++) { int32_t up = -p[i]; if (up > 0) fixed_end[idx_of(up)] = 1; else fixed_start[idx_of(up)] = 1; }
        }
        if (fe0[u]) {   // downstream of a fixed end
            fixed_end[u] = 1;
            u32 n; const int32_t* p = L.next_of(num, &n);
            for (u32 i = 0; i < n; i++) { int32_t down = p[i]; if (down > 0) fixed_start[idx_of(down)] = 1; else fixed_end[idx_of(down)] = 1; }
        }
    }
};
// get_exclusive_inputs / get_exclusive_outputs (:233-280) and the fixed-end guards of expand_repeats (:66-83).
struct CandFunctor {
    OrderedLinks L; const u8* fixed_start; const u8* fixed_end; u8* cand;
    AC_HD void operator()(u64 idx) const {
        u32 x = (u32)(idx >> 1);
        bool inputs = (idx & 1) == 0;
        int32_t xnum = (int32_t)x + 1;
        u32 n; const int32_t* p = L.next_of(inputs ? -xnum : xnum, &n);
        bool ok = n >= 2 && !(inputs ? fixed_start[x] : fixed_end[x]);
        for (u32 i = 0; ok && i < n; i++) {
            int32_t other = inputs ? -p[i] : p[i];
            // inputs: other's next list must be exactly [x+].  outputs: other's prev list must be exactly [x+],
            // i.e. the next list of other's opposite strand must be exactly [x-].
            u32 m; const int32_t* q = L.next_of(inputs ? other : -other, &m);
            if (!(m == 1 && q[0] == (inputs ? xnum : -xnum))) ok = false;
            u32 u = idx_of(other);
            if (u == x) ok = false;
            if (inputs) { if ((other > 0 && fixed_end[u]) || (other < 0 && fixed_start[u])) ok = false; }
            else { if ((other > 0 && fixed_start[u]) || (other < 0 && fixed_end[u])) ok = false; }
        }
        cand[idx] = ok ? 1 : 0;
    }
};

// ---- K15: renumber_unitigs (unitig_graph.rs:295-315): length descending, forward sequence ascending, depth
// descending; the sort is stable on the incoming order -------------------------------------------------------------
struct UnitigLess {
    const u32* len; const u64* off; const u8* seq; const u32* depth;
    AC_HD bool operator()(const u32& a, const u32& b) const {
        u32 la = len[a], lb = len[b];
        if (la != lb) return la > lb;
        const u8* pa = seq + off[a]; const u8* pb = seq + off[b];
        for (u32 i = 0; i < la; i++) { u8 ca = pa[i], cb = pb[i]; if (ca != cb) return ca < cb; }
        return depth[a] > depth[b];
    }
};

// The same order without a comparator sort: two stable LSD radix passes on (length desc | first 32 bases asc | depth desc) —
// exact whenever the sequence is at most 32 long or differs from its neighbours within the first 32 bases — then a
// stable insertion sort with the full comparator inside the (rare, small) groups of longer unitigs that agree on
// length and on their first 32 bases.  Both sorts being stable, equal elements keep the incoming order.
struct RenumKeyFunctor {
    const u32* len; const u64* off; const u8* seq; u64* prefix;
    AC_HD void operator()(u64 u) const {
        u32 l = len[u];
        const u8* p = seq + off[u];
        u64 w = 0;
        u32 m = l < 32 ? l : 32;
        for (u32 i = 0; i < m; i++) { u32 ch = p[i]; w |= (u64)(((ch >> 1) ^ (ch >> 2)) & 3u) << (62 - 2 * i); }
        prefix[u] = w;
    }
};
struct RenumPassFunctor {   // key of the element currently at position i: pass 0 = (prefix low | ~depth), pass 1 = (~len | prefix high)
    const u32* order; const u32* len; const u32* depth; const u64* prefix; int pass; u64* key;
    AC_HD void operator()(u64 i) const {
        u32 u = order[i];
        key[i] = pass == 0 ? ((prefix[u] << 32) | (u64)(~depth[u])) : (((u64)(~len[u]) << 32) | (prefix[u] >> 32));
    }
};
static const u32 RENUM_MAX_GROUP = 64;
struct RenumTieFunctor {
    u32* order; u64 n; const u32* len; const u64* prefix; UnitigLess less; u32* too_big;
    AC_HD bool same(u32 a, u32 b) const { return len[a] == len[b] && prefix[a] == prefix[b]; }
    AC_D void operator()(u64 i) const {
        u32 u = order[i];
        if (len[u] <= 32) return;                                 // the radix key was the whole comparator
        if (i > 0 && same(order[i - 1], u)) return;               // not the first of its group
        u64 e = i + 1;
        while (e < n && e - i <= RENUM_MAX_GROUP && same(order[e], u)) e++;
        if (e - i > RENUM_MAX_GROUP) { atomic_or32(too_big, 1u); return; }
        for (u64 a = i + 1; a < e; a++) {                         // stable insertion sort of order[i, e)
            u32 v = order[a];
            u64 b = a;
            while (b > i && less(v, order[b - 1])) { order[b] = order[b - 1]; b--; }
            order[b] = v;
        }
    }
};

// ---- K17: expand_repeats on the device (graph_simplification.rs:26-142, shift primitives unitig.rs:217-249) ---------
// The reference visits junctions sequentially (unitigs in first-renumber order, inputs side then outputs side) and
// the result depends on that order only where two junctions touch a common unitig.  A junction (x, side) reads and
// writes x (as the destination that gains sequence) and its exclusive sources (which lose it); everything that
// decides WHICH junctions qualify is static (K14).  So: number the candidate junctions in visiting order, give
// every candidate the level 1 + max(level of earlier candidates sharing a unitig with it), and run each pass level
// by level — candidates of one level never share a unitig, every conflicting pair keeps the reference's order.
// The two sides of the same destination touch disjoint fields of it (prefix + min forward position vs. suffix + min
// reverse position) and are not a conflict.
// A unitig's sequence during a pass is [bytes gained at its start this pass][core view][bytes gained at its end this
// pass]; after every pass that moved something the sequences are rewritten contiguously.
struct ExpState {
    const u8* cur; u64* coff; u32* clen;                       // core: view into the current sequence buffer
    u32* pre_off; u32* pre_len; u32* post_off; u32* post_len;  // gained this pass: offsets into `pool`
    u8* pool; u32* pool_used;
    u32* minf; u32* minr;                                      // min forward / reverse position (graph_simplification.rs:164-181)
    u8* dirty; const u8* cand;
    OrderedLinks L;
    u64* shifted;
};
AC_HD u8 comp_base(u8 c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
AC_HD u32 exp_len(const ExpState& e, u32 u) { return e.pre_len[u] + e.clen[u] + e.post_len[u]; }
AC_HD u8 exp_at(const ExpState& e, u32 u, u32 i) {
    u32 pl = e.pre_len[u];
    if (i < pl) return e.pool[e.pre_off[u] + i];
    i -= pl;
    u32 cl = e.clen[u];
    if (i < cl) return e.cur[e.coff[u] + i];
    return e.pool[e.post_off[u] + (i - cl)];
}
AC_HD u8 exp_from_start(const ExpState& e, int32_t s, u32 i) { u32 u = idx_of(s); return s > 0 ? exp_at(e, u, i) : comp_base(exp_at(e, u, exp_len(e, u) - 1 - i)); }
AC_HD u8 exp_from_end(const ExpState& e, int32_t s, u32 i) { u32 u = idx_of(s); return s > 0 ? exp_at(e, u, exp_len(e, u) - 1 - i) : comp_base(exp_at(e, u, i)); }
AC_HD void exp_remove_start(const ExpState& e, u32 u, u32 n) {
    u32 d = n < e.pre_len[u] ? n : e.pre_len[u]; e.pre_off[u] += d; e.pre_len[u] -= d; n -= d;
    d = n < e.clen[u] ? n : e.clen[u]; e.coff[u] += d; e.clen[u] -= d; n -= d;
    e.post_off[u] += n; e.post_len[u] -= n;
}
AC_HD void exp_remove_end(const ExpState& e, u32 u, u32 n) {
    u32 d = n < e.post_len[u] ? n : e.post_len[u]; e.post_len[u] -= d; n -= d;
    d = n < e.clen[u] ? n : e.clen[u]; e.clen[u] -= d; n -= d;
    e.pre_len[u] -= n;
}
// The (at most four) candidate junctions that read or write unitig u: u as destination (2u, 2u+1) and u as an
// exclusive source of the junction its only forward / reverse link leads to.  0xFFFFFFFF = none.
AC_HD void touchers(const OrderedLinks& L, const u8* cand, u32 u, u32 out[4]) {
    out[0] = cand[2 * (u64)u] ? 2 * u : 0xFFFFFFFFu;
    out[1] = cand[2 * (u64)u + 1] ? 2 * u + 1 : 0xFFFFFFFFu;
    for (int side = 0; side < 2; side++) {
        u32 c = 0xFFFFFFFFu;
        if (L.cnt[2 * (u64)u + side] == 1) {
            int32_t p0 = L.ord[(2 * (u64)u + side) * 5];
            u32 cc = 2 * idx_of(p0) + (p0 > 0 ? 0u : 1u);
            if (cand[cc]) c = cc;
        }
        out[2 + side] = c;
    }
}
struct CandFlagFunctor {    // j = 2*oi + side over the visiting order: is (order1[oi], side) a candidate?
    const u32* order1; const u8* cand; u32* flag;
    AC_HD void operator()(u64 j) const { flag[j] = cand[2 * (u64)order1[j >> 1] + (j & 1)] ? 1u : 0u; }
};
struct CandListFunctor {
    const u32* order1; const u32* flag; const u32* pos; u32* clist; u32* prio;
    AC_HD void operator()(u64 j) const {
        if (!flag[j]) return;
        u32 c = 2 * order1[j >> 1] + (u32)(j & 1);
        clist[pos[j]] = c;
        prio[c] = pos[j];
    }
};
struct LevelRelaxFunctor {
    OrderedLinks L; const u8* cand; const u32* clist; const u32* prio; u32* level; u32* changed;
    AC_D void operator()(u64 ci) const {
        u32 c = clist[ci];
        u32 x = c >> 1;
        bool inputs = (c & 1) == 0;
        u32 n; const int32_t* p = L.next_of(inputs ? -((int32_t)x + 1) : (int32_t)x + 1, &n);
        u32 lv = 1;
        for (u32 i = 0; i <= n; i++) {
            u32 u = (i == n) ? x : idx_of(p[i]);
            u32 t[4];
            touchers(L, cand, u, t);
            for (int j = 0; j < 4; j++) {
                u32 c2 = t[j];
                if (c2 == 0xFFFFFFFFu || c2 == c) continue;
                if (u == x && (c2 >> 1) == x) continue;     // the other side of the same destination: disjoint fields
                u32 pr = prio[c2];
                if (pr < (u32)ci) { u32 l2 = level[pr] + 1; if (l2 > lv) lv = l2; }
            }
        }
        if (lv > level[ci]) { level[ci] = lv; atomic_or32(changed, 1u); }
    }
};
struct LevelKeyFunctor {
    const u32* level; u64* key;
    AC_HD void operator()(u64 ci) const { key[ci] = ((u64)level[ci] << 32) | ci; }
};
struct LevelBoundsFunctor {   // keys sorted by (level >= 1, visiting position): bstart[lv] = first index of level lv (lv <= cap),
    const u64* key; u64 n; u32* bstart; u32 cap;      // bstart[0] = number of levels
    AC_HD void operator()(u64 i) const {
        u32 lv = (u32)(key[i] >> 32);
        if ((i == 0 || (u32)(key[i - 1] >> 32) != lv) && lv <= cap) bstart[lv] = (u32)i;
        if (i + 1 == n) bstart[0] = lv;
    }
};
// A unitig strand's current sequence as seen by one junction: the descriptor is read once, after that every character
// is one independent byte load (the per-character exp_at chain of five dependent loads made every level of a pass
// latency-bound at ~25 characters x sources).
struct ExpView { u64 coff; u32 pre_len, clen, pre_off, post_off, len; bool fwd; };
AC_HD ExpView exp_view(const ExpState& e, int32_t s) {
    u32 u = idx_of(s);
    ExpView v;
    v.coff = e.coff[u]; v.pre_len = e.pre_len[u]; v.clen = e.clen[u]; v.pre_off = e.pre_off[u]; v.post_off = e.post_off[u];
    v.len = v.pre_len + v.clen + e.post_len[u];
    v.fwd = s > 0;
    return v;
}
// Branch-free addressing (selects, no control flow), so that a group of character loads can be issued back to back:
// with branches every load waited for the previous one and a level's kernel became one long latency chain.
AC_HD const u8* view_ptr(const ExpState& e, const ExpView& v, u32 i) {
    u32 i2 = i - v.pre_len;
    const u8* p_pre = e.pool + v.pre_off + i;
    const u8* p_core = e.cur + v.coff + i2;
    const u8* p_post = e.pool + v.post_off + (i2 - v.clen);
    return i < v.pre_len ? p_pre : (i2 < v.clen ? p_core : p_post);
}
AC_HD u8 comp_sel(u8 c) { u8 r = 'N'; r = c == 'A' ? (u8)'T' : r; r = c == 'C' ? (u8)'G' : r; r = c == 'G' ? (u8)'C' : r; r = c == 'T' ? (u8)'A' : r; return r; }
// Characters start .. start+7 counted from the strand's start (from_end = false) or backwards from its end (true);
// indices beyond the sequence are clamped (callers ignore those slots).
AC_HD void view_load8(const ExpState& e, const ExpView& v, bool from_end, u32 start, u8 out[8]) {
    const u32 last = v.len - 1;
#pragma unroll
    for (u32 b = 0; b < 8; b++) {
        u32 i = start + b;
        i = i > last ? last : i;
        u32 pos = (from_end == v.fwd) ? last - i : i;      // forward strand read from its end, or reverse strand read from its start
        out[b] = *view_ptr(e, v, pos);
    }
    if (!v.fwd) {
#pragma unroll
        for (u32 b = 0; b < 8; b++) out[b] = comp_sel(out[b]);
    }
}
// Gained sequence accumulates in the pool across passes (a side that gains again gets a new piece = new characters +
// old piece), so the sequences are rewritten contiguously only once, after the last pass.
struct ExpandFunctor {
    ExpState e; const u32* clist; u64 begin; u32 pool_cap; u32* err;
    AC_D void operator()(u64 i, bool valid) const {
        u32 c = valid ? clist[begin + i] : 0;
        bool active = valid && e.dirty[c];    // not dirty: unchanged since it last shifted nothing, shifts nothing again
        if (active) e.dirty[c] = 0;
        const u32 x = c >> 1;
        const bool inputs = (c & 1) == 0;
        u32 n = 0;
        int32_t srcs[5];
        ExpView sv[5];
        u32 amount = 0;
        if (active) {
            const int32_t* p = e.L.next_of(inputs ? -((int32_t)x + 1) : (int32_t)x + 1, &n);
            // inputs:  forward_prev(x) = { -l : l in reverse_next(x) }   (graph_simplification.rs:233-255)
            // outputs: forward_next(x)                                    (:258-280)
            u32 min_len = 0xFFFFFFFFu;
            bool dup = false;
#pragma unroll
            for (u32 j = 0; j < 5; j++) {
                if (j >= n) break;
                srcs[j] = inputs ? -p[j] : p[j];
                sv[j] = exp_view(e, srcs[j]);
                if (sv[j].len < min_len) min_len = sv[j].len;
                for (u32 q = 0; q < j; q++) if (idx_of(srcs[q]) == idx_of(srcs[j])) dup = true;
            }
            // get_common_end_seq (:298-312) / get_common_start_seq (:283-295) of the source strand sequences, eight
            // characters per step (the loads of a step are independent of each other)
            while (amount < min_len) {
                u32 blk = min_len - amount < 8 ? min_len - amount : 8;
                u8 c0[8], cj[4][8];
                view_load8(e, sv[0], inputs, amount, c0);
#pragma unroll
                for (u32 j = 1; j < 5; j++) if (j < n) view_load8(e, sv[j], inputs, amount, cj[j - 1]);
                u32 m = blk;
#pragma unroll
                for (u32 j = 1; j < 5; j++) {
                    if (j >= n) break;
#pragma unroll
                    for (u32 b = 0; b < 8; b++) if (b < m && cj[j - 1][b] != c0[b]) m = b;
                }
                amount += m;
                if (m < blk) break;
            }
            if (amount > 0) {   // avoid_zero_len_unitigs (:145-161): trim while min_source_len <= len * dup
                u32 lim = (min_len - 1) / (dup ? 2u : 1u);
                if (amount > lim) amount = lim;
            }
            if (amount > 0) {   // avoid_start_of_path (:164-181): trim while any forward / reverse position <= len
                u32 m = inputs ? e.minf[x] : e.minr[x];
                u32 lim = m > 0 ? m - 1 : 0;
                if (amount > lim) amount = lim;
            }
        }
        // the destination's piece on the gaining side: [new characters][old piece] (start) / [old piece][new characters] (end)
        const u32 old_len = amount ? (inputs ? e.pre_len[x] : e.post_len[x]) : 0;
        const u32 old_off = amount ? (inputs ? e.pre_off[x] : e.post_off[x]) : 0;
        const u32 off = wave_alloc32(e.pool_used, amount ? amount + old_len : 0);
        wave_add64(e.shifted, amount);
        if (amount == 0) return;
        if ((u64)off + amount + old_len > (u64)pool_cap) { atomic_or32(err, 64u); return; }
        if (inputs) {   // shift_sequence_1 (:89-119): the LAST `amount` characters of the common suffix move onto x's start
            for (u32 j0 = 0; j0 < amount; j0 += 8) {
                u8 ch[8];
                view_load8(e, sv[0], true, j0, ch);
#pragma unroll
                for (u32 b = 0; b < 8; b++) if (j0 + b < amount) e.pool[off + (amount - 1 - (j0 + b))] = ch[b];
            }
            for (u32 j = 0; j < old_len; j++) e.pool[off + amount + j] = e.pool[old_off + j];
            for (u32 j = 0; j < n; j++) {
                u32 u = idx_of(srcs[j]);
                if (srcs[j] > 0) { exp_remove_end(e, u, amount); e.minr[u] += amount; }       // unitig.rs:226-233
                else { exp_remove_start(e, u, amount); e.minf[u] += amount; }                  // unitig.rs:217-224
            }
            e.pre_off[x] = off; e.pre_len[x] = amount + old_len; e.minf[x] -= amount;          // unitig.rs:235-241
        } else {        // shift_sequence_2 (:122-142): the FIRST `amount` characters of the common prefix move onto x's end
            for (u32 j = 0; j < old_len; j++) e.pool[off + j] = e.pool[old_off + j];
            for (u32 j0 = 0; j0 < amount; j0 += 8) {
                u8 ch[8];
                view_load8(e, sv[0], false, j0, ch);
#pragma unroll
                for (u32 b = 0; b < 8; b++) if (j0 + b < amount) e.pool[off + old_len + j0 + b] = ch[b];
            }
            for (u32 j = 0; j < n; j++) {
                u32 u = idx_of(srcs[j]);
                if (srcs[j] > 0) { exp_remove_start(e, u, amount); e.minf[u] += amount; }
                else { exp_remove_end(e, u, amount); e.minr[u] += amount; }
            }
            e.post_off[x] = off; e.post_len[x] = amount + old_len; e.minr[x] -= amount;        // unitig.rs:243-249
        }
        for (u32 j = 0; j <= n; j++) {   // every junction that touches a changed unitig must be looked at again
            u32 u = (j == n) ? x : idx_of(srcs[j]);
            u32 t[4];
            touchers(e.L, e.cand, u, t);
            for (int q = 0; q < 4; q++) {
                if (t[q] == 0xFFFFFFFFu) continue;
                if (u == x && (t[q] >> 1) == x && t[q] != c) continue;   // the destination's other side reads nothing that changed
                e.dirty[t[q]] = 1;
            }
        }
    }
};
struct FillU32Functor { u32* a; u32 v; AC_HD void operator()(u64 i) const { a[i] = v; } };
struct ExpLenFunctor {
    ExpState e; u64* len64; u32 n_unitigs;
    AC_HD void operator()(u64 u) const { len64[u] = u < n_unitigs ? (u64)exp_len(e, (u32)u) : 0; }
};
struct MaterializeFunctor {   // 64 output bytes per thread
    ExpState e; const u64* noff; u32 n_unitigs; u64 total; u8* out;
    AC_HD void operator()(u64 tid) const {
        u64 g0 = tid * 64, g1 = g0 + 64;
        if (g1 > total) g1 = total;
        if (g0 >= total) return;
        u32 lo = 0, hi = n_unitigs;   // largest r with noff[r] <= g0
        while (hi - lo > 1) { u32 mid = lo + ((hi - lo) >> 1); if (noff[mid] <= g0) lo = mid; else hi = mid; }
        u32 r = lo;
        for (u64 g = g0; g < g1; g++) {
            while (g >= noff[r + 1]) r++;
            out[g] = exp_at(e, r, (u32)(g - noff[r]));
        }
    }
};
struct ExpResetFunctor {
    ExpState e; const u64* noff;
    AC_HD void operator()(u64 u) const {
        e.clen[u] = (u32)(noff[u + 1] - noff[u]); e.coff[u] = noff[u];
        e.pre_len[u] = 0; e.post_len[u] = 0; e.pre_off[u] = 0; e.post_off[u] = 0;
    }
};

// ---- K16: finalisation in the final numbering -----------------------------------------------------------------
struct FinalMetaFunctor {   // per final index i
    const u32* order2; const u64* foff; const u32* flen; const u32* depth; const u8* lcnt;
    u64* number_len; u64* seq_begin; double* depth_out; u32* seq_len; u64* lcount;
    AC_HD void operator()(u64 i) const {
        u32 r = order2[i];
        number_len[r] = ((u64)flen[r] << 32) | (u64)(i + 1);
        seq_begin[i] = foff[r];
        depth_out[i] = (double)depth[r];
        seq_len[i] = flen[r];
        lcount[i] = (u64)lcnt[2 * (u64)r] + (u64)lcnt[2 * (u64)r + 1];
    }
};
struct LinkOutFunctor {     // get_links_for_gfa (unitig_graph.rs:333-350): per final unitig, forward_next then reverse_next
    const u32* order2; OrderedLinks L; const u64* number_len; const u64* loff; Link* out;
    AC_HD void operator()(u64 i) const {
        u32 r = order2[i];
        u64 w = loff[i];
        for (int side = 0; side < 2; side++) {
            u32 n; const int32_t* p = L.next_of(side == 0 ? (int32_t)r + 1 : -((int32_t)r + 1), &n);
            for (u32 j = 0; j < n; j++) {
                Link l; l.a = (u32)i + 1; l.a_fwd = side == 0 ? 1 : 0;
                l.b = (u32)(number_len[idx_of(p[j])] & 0xFFFFFFFFu); l.b_fwd = p[j] > 0 ? 1 : 0;
                out[w++] = l;
            }
        }
    }
};
struct RemapFunctor {       // seed numbers -> final numbers, per-sequence length sums.  A wavefront owns 4096 consecutive
    int32_t* path; const u64* number_len; const u64* path_off; u32 n_seqs; u64 n_ent; u64* sums; u64 first_wave;   // entries; lane l takes l, l+64, ...
    AC_D void operator()(u64 tid, bool valid) const {
        if (!valid) return;   // the launch is a whole number of wavefronts, so this is wavefront-uniform
        const u64 wave = first_wave + (tid >> 6);
        const u32 lane = (u32)(tid & 63);
        const u64 w0 = wave * 4096;
        u64 w1 = w0 + 4096;
        if (w1 > n_ent) w1 = n_ent;
        u32 s = 0;
        u64 acc = 0;
        u64 i = w0 + lane;
        if (i < w1) {
            u32 lo = 0, hi = n_seqs;   // largest s with path_off[s] <= i
            while (hi - lo > 1) { u32 mid = lo + ((hi - lo) >> 1); if (path_off[mid] <= i) lo = mid; else hi = mid; }
            s = lo;
        }
        for (; i < w1; i += 64) {
            while (s + 1 < n_seqs && i >= path_off[s + 1]) { if (acc) atomic_add64(&sums[s], acc); acc = 0; s++; }
            int32_t v = path[i];
            u64 nl = number_len[idx_of(v)];
            int32_t f = (int32_t)(nl & 0xFFFFFFFFu);
            path[i] = v > 0 ? f : -f;
            acc += nl >> 32;
        }
#ifndef AC_EMU
        // usually the whole wavefront ends inside one sequence: one atomic instead of 64 to the same address
        const u32 s0 = (u32)__shfl((int)s, 0);
        if (__all(s == s0 || acc == 0)) {
            u64 t = acc;
#pragma unroll
            for (int o = 32; o; o >>= 1) t += (u64)__shfl_xor((unsigned long long)t, o);
            if (lane == 0 && t) atomic_add64(&sums[s0], t);
            return;
        }
#endif
        if (acc) atomic_add64(&sums[s], acc);
    }
};

// ---- K12: trimmed unitig sequences (unitig.rs:113-166) -----------------------------------------------------
struct SeqFunctor {
    const u64* bits; const u64* useq_off; const u64* ustartpos; const u32* ulen; const u8* uorient;
    u32 n_unitigs; u64 total; int h; u8* out;
    AC_HD void operator()(u64 tid) const {
        u64 g0 = tid * 64, g1 = g0 + 64;
        if (g1 > total) g1 = total;
        if (g0 >= total) return;
        u32 lo = 0, hi = n_unitigs;   // largest r with useq_off[r] <= g0
        while (hi - lo > 1) { u32 mid = lo + ((hi - lo) >> 1); if (useq_off[mid] <= g0) lo = mid; else hi = mid; }
        u32 r = lo;
        for (u64 g = g0; g < g1; g++) {
            while (g >= useq_off[r] + (u64)ulen[r]) r++;
            u64 i = g - useq_off[r];
            u32 n = ulen[r];
            u32 c;
            if (uorient[r]) c = text_code(bits, ustartpos[r] + (u64)h + i);
            else c = 3u - text_code(bits, ustartpos[r] + (u64)h + ((u64)n - 1 - i));
            out[g] = (u8)("ACGT"[c]);
        }
    }
};

// ---- sharded build, phase 1: this rank's novel runs ("fragments") --------------------------------------------
// One compress job sharded by sequence over several devices: a rank inserts only its own sequences; the maximal runs
// of consecutive rank-novel positions are exactly the text this rank can contribute to the global k-mer set (every
// k-mer's globally smallest occurrence is rank-novel on the rank that holds it, and a whole unitig is novel
// together).  The union of all ranks' fragments is a text with the same k-mer set as the whole input — for similar
// assemblies a small multiple of ONE assembly — from which every rank builds the identical global graph.
struct RunEdgeCountFunctor {
    const u64* bm; u64 n_words; u32* n_start; u32* n_end;
    AC_HD void operator()(u64 w) const {
        u64 x = bm[w];
        u64 prev = w ? (bm[w - 1] >> 63) : 0;
        u64 next = (w + 1 < n_words) ? (bm[w + 1] & 1) : 0;
        n_start[w] = (u32)popc64(x & ~((x << 1) | prev));
        n_end[w] = (u32)popc64(x & ~((x >> 1) | (next << 63)));
    }
};
struct RunEdgeFillFunctor {
    const u64* bm; u64 n_words; const u32* soff; const u32* eoff; u64* run_start; u64* run_end;
    AC_HD void operator()(u64 w) const {
        u64 x = bm[w];
        u64 prev = w ? (bm[w - 1] >> 63) : 0;
        u64 next = (w + 1 < n_words) ? (bm[w + 1] & 1) : 0;
        u64 st = x & ~((x << 1) | prev), en = x & ~((x >> 1) | (next << 63));
        u32 i = soff[w];
        while (st) { u64 low = st & (~st + 1); run_start[i++] = w * 64 + (u64)popc64(low - 1); st ^= low; }
        i = eoff[w];
        while (en) { u64 low = en & (~en + 1); run_end[i++] = w * 64 + (u64)popc64(low - 1); en ^= low; }
    }
};
// Fragment i < n_runs: the i-th novel run.  Fragments n_runs + 2s, n_runs + 2s + 1: the first and the last k-mer of
// sequence s on their own (1 k-mer each), flagged START / END: they carry first_position (kmer_graph.rs:57-60) to
// the global build.  They sit AFTER the rank's novel runs, so they never hold the smallest occurrence of a k-mer in
// the union text and cannot split a unitig.
// meta record: [len:32][leading dots:8][trailing dots:8][flags:8][0:8]
static const u32 FRAG_START = 1, FRAG_END = 2;
struct FragMetaFunctor {
    TextCtx t; const u64* run_start; const u64* run_end; u64 n_runs, n_frags; u64* fpos; u64* meta; u64* blen; u32* err;
    AC_D void operator()(u64 i) const {
        if (i == n_frags) { blen[i] = 0; return; }    // sentinel: the exclusive scan then ends with the total
        u64 a; u32 len; u32 flags = 0;
        if (i < n_runs) { a = run_start[i]; len = (u32)(run_end[i] - a + 1); }
        else { u64 j = i - n_runs; u32 s = (u32)(j >> 1); bool last = (j & 1) != 0;
               a = t.seq_off[s] + (last ? (u64)t.seq_len[s] - 1 : 0); len = 1; flags = last ? FRAG_END : FRAG_START; }
        u32 s, f;
        if (!locate(t, a, &s, &f) || (u64)f + len > (u64)t.seq_len[s]) { atomic_or32(err, 32u); fpos[i] = a; meta[i] = 0; blen[i] = 0; return; }
        int k = t.k;
        int plen = (int)t.seq_len[s] + k - 1;
        int ld = (int)t.seq_d1[s] - (int)f;
        int td = (int)(f + len - 1) + k - (plen - (int)t.seq_d2[s]);
        if (ld < 0) ld = 0;
        if (td < 0) td = 0;
        fpos[i] = a;
        meta[i] = (u64)len | ((u64)ld << 32) | ((u64)td << 40) | ((u64)flags << 48);
        blen[i] = (u64)len + (u64)k;                  // k-1 tail bytes + one separator
    }
};
struct FragCopyFunctor {     // 64 output bytes per thread
    const u8* text; const u64* fpos; const u64* boff; u64 n_frags, total; u8* out;
    AC_HD void operator()(u64 tid) const {
        u64 g0 = tid * 64, g1 = g0 + 64;
        if (g1 > total) g1 = total;
        if (g0 >= total) return;
        u64 lo = 0, hi = n_frags;   // largest r with boff[r] <= g0
        while (hi - lo > 1) { u64 mid = lo + ((hi - lo) >> 1); if (boff[mid] <= g0) lo = mid; else hi = mid; }
        u64 r = lo;
        for (u64 g = g0; g < g1; g++) {
            while (g >= boff[r + 1]) r++;
            u64 j = g - boff[r];
            out[g] = (g + 1 == boff[r + 1]) ? (u8)'$' : text[fpos[r] + j];
        }
    }
};
// ---- sharded build: per-unitig quantities that combine over ranks -------------------------------------------------
// sum[0..U) = occurrences (depth), sum[U..2U) / sum[2U..3U) = "a path starts / ends here" counts; min[0..U) / [U..2U) = smallest
// forward / reverse position, biased by 2^31 so that a signed 32-bit MIN orders them as unsigned.
struct ReduceExportFunctor {
    const u32* depth; const u8* fs0; const u8* fe0; const u32* mf; const u32* mr; u64 U; int32_t* sum; int32_t* mn;
    AC_HD void operator()(u64 i) const {
        sum[i] = (int32_t)depth[i]; sum[U + i] = fs0[i]; sum[2 * U + i] = fe0[i];
        mn[i] = (int32_t)(mf[i] ^ 0x80000000u); mn[U + i] = (int32_t)(mr[i] ^ 0x80000000u);
    }
};
struct ReduceImportFunctor {
    u32* depth; u8* fs0; u8* fe0; u32* mf; u32* mr; u64 U; const int32_t* sum; const int32_t* mn;
    AC_HD void operator()(u64 i) const {
        depth[i] = (u32)sum[i]; fs0[i] = sum[U + i] > 0 ? 1 : 0; fe0[i] = sum[2 * U + i] > 0 ? 1 : 0;
        mf[i] = (u32)mn[i] ^ 0x80000000u; mr[i] = (u32)mn[U + i] ^ 0x80000000u;
    }
};

// =============================================================================================================
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
static int key_words(int k) { int w = words_for_k(k); return w <= 4 ? w : (w <= 8 ? 8 : 16); }

// renumber_unitigs (unitig_graph.rs:295-315): stable sort of `order` by (length desc, sequence asc, depth desc).
static void renumber_sort(DBuf<u32>& order, u32 U, const u32* len, const u64* off, const u8* seq, const u32* depth, u32* flag) {
    if (U <= 1) return;
    DBuf<u32> backup(U);
    copy_d2d(backup.ptr(), order.ptr(), (size_t)U * 4);
    DBuf<u64> prefix(U), key(U);
    launch(U, RenumKeyFunctor{len, off, seq, prefix.ptr()});
    for (int pass = 0; pass < 2; pass++) {
        launch(U, RenumPassFunctor{order.ptr(), len, depth, prefix.ptr(), pass, key.ptr()});
        sort_pairs_u64_u32(key, order, U, 64);
    }
    UnitigLess less{len, off, seq, depth};
    launch(U, RenumTieFunctor{order.ptr(), U, len, prefix.ptr(), less, flag});
    if (read_scalar(flag)) {    // a large group of long unitigs sharing length and 32-base prefix: comparator merge sort
        copy_d2d(order.ptr(), backup.ptr(), (size_t)U * 4);
        sort_keys_cmp(order, U, less);
        u32 zero = 0;
        copy_h2d(flag, &zero, 4);
    }
}

static u64 next_pow2(u64 x) { u64 p = 1; while (p < x) p <<= 1; return p; }
// Device memory one build of an n_text-byte text needs, roughly: packed text + bitmaps (~0.6 B/position), staging slots
// of the path walk (4 B/position), k-mer table and per-k-mer arrays (sized by distinct content), unitig-sized buffers.
static size_t arena_estimate(u64 n_text, bool owns_text) { return (size_t)n_text * (owns_text ? 8 : 7) + ((size_t)768 << 20); }
// Tuning knobs of the insert (environment, read once): AC_INSERT_VARIANT=1 selects the thread-per-chunk kernel,
// AC_INSERT_CHUNK the largest wavefront chunk (positions).
static int insert_variant() { static int v = [] { const char* e = getenv("AC_INSERT_VARIANT"); return e ? atoi(e) : 0; }(); return v; }
static u64 wave_chunk_max() { static u64 v = [] { const char* e = getenv("AC_INSERT_CHUNK"); u64 x = e ? (u64)atoll(e) : 8192; return (std::max<u64>(x, 256) + 63) & ~63ULL; }(); return v; }

// A text resident in HBM with its sequence table and its 2-bit packing.
struct PackedText {
    const u8* d_text = nullptr;
    u64 n_text = 0, n_bases = 0;
    u32 n_seqs = 0;
    int any_dots = 0;
    DBuf<u64> seq_off; DBuf<u32> seq_len; DBuf<u16> seq_d1, seq_d2; DBuf<u8> seq_flags;
    bool has_flags = false;
    DBuf<u64> bits, mask;
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
        n_bases = 0; any_dots = 0;
        for (u32 i = 0; i < n_seqs; i++) { n_bases += len[i]; if (d1[i] || d2[i]) any_dots = 1; }
        stream_sync();
    }
    TextCtx ctx(int k) const { return TextCtx{bits.ptr(), mask.ptr(), n_text, k, seq_off.ptr(), seq_len.ptr(), seq_d1.ptr(), seq_d2.ptr(), n_seqs}; }
    void pack() {   // K1
        u64 n_bits_words = n_text / 32 + 24, n_mask_words = n_text / 64 + 12;   // slack for W <= 16 key words
        bits.alloc(n_bits_words); mask.alloc(n_mask_words);
        bits.fill_bytes(0);
        mask.fill_bytes(0xFF);
        launch((n_text + 31) / 32, PackFunctor{d_text, n_text, bits.ptr(), (u32*)mask.ptr()});
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
    // fragments of a sharded build
    DBuf<u8> frag_text; DBuf<u64> frag_meta; u64 frag_bytes = 0, n_frags = 0;
    u64 distinct_upper = 0;    // sharded builds: sum of the ranks' local distinct counts (0 = unknown)

    void begin(BuildTimings* t) {
        tm = t; t_begin = t0 = now_s();

        counters.alloc(8); counters.fill_bytes(0);
    }
    void check_sizes(const PackedText& t) const {
        if (t.n_text >= POS_MASK) throw DeviceError("input too large for 40-bit text positions");
        if (t.n_seqs == 0 || t.n_text < (u64)k + 2) throw DeviceError("no sequences");
    }
    template <int W> void insert(const PackedText& t, u32 hint, DBuf<u64>* slots_out, u64* cap_out, u64* n_distinct_out);
    void novel_bitmap(const PackedText& t, const DBuf<u64>& sl, u64 c, DBuf<u64>* bm_out, DBuf<u64>* occ_out);
    DBuf<u64> occ;             // slot-occupancy bitmap of the graph table
    Table graph_table() const { return Table{const_cast<u64*>(slots.ptr()), cap - 1, occ.ptr()}; }
    template <int W> void fragments();
    template <int W> void table();                      // K2, K3 on G
    template <int W> void degrees(u64 lo, u64 hi);      // K5 for novel indices [lo, hi)
    template <int W> void unitigs();                    // K6..K11 on G
    template <int W> void walk();
    template <int W> void tail(FinalGraph* out, bool want_graph, bool want_paths);
    u64 deg_lo = 0, deg_hi = 0;
};

// K2 insert.  Capacity from the reference's own capacity hint (assembly_count, kmer_graph.rs:40): similar assemblies
// share most k-mers.  Overflow -> retry with a larger table.
template <int W>
void GraphBuilder::Impl::insert(const PackedText& pt, u32 hint, DBuf<u64>* slots_out, u64* cap_out, u64* n_distinct_out) {
    TextCtx t = pt.ctx((int)k);
    const u64 p_end_all = pt.n_text - (u64)k + 1;     // one past the last window that fits in the text
    if (hint == 0) hint = 1;
    u64 est = pt.n_bases / hint;
    u64 c = next_pow2(std::max<u64>(1024, est * 3 + 4096));
    if (&pt == &uni && distinct_upper) c = next_pow2(std::max<u64>(1024, distinct_upper * 10 / 7 + 4096));   // no retry: an upper bound is known
    if (c > next_pow2(pt.n_bases * 2 + 1024)) c = next_pow2(pt.n_bases * 2 + 1024);
    DBuf<InsertStats> istats(257);       // [256].real doubles as the kernel's error word: one D2H reads everything
    DBuf<u64> sl;
    u64 n_distinct = 0;
    for (;;) {
        sl.alloc(c);
        sl.fill_bytes(0xFF);
        counters.fill_bytes(0);
        istats.fill_bytes(0);
        u32* ierr = (u32*)&istats.ptr()[256].real;
        Table tb{sl.ptr(), c - 1, nullptr};
        stream_sync();
#ifndef AC_EMU
        hipEvent_t e0, e1;
        AC_HIP_CHECK(hipEventCreate(&e0)); AC_HIP_CHECK(hipEventCreate(&e1));
        AC_HIP_CHECK(hipEventRecord(e0, 0));
#endif
        // Phases over geometrically growing prefixes: [0, n/A), [n/A, 2n/A), [2n/A, 4n/A), ...  (A = assembly
        // count): what a phase streams has, for similar assemblies, mostly been inserted by the earlier ones.
        u32 launches = 0;
        u64 first = std::max<u64>(p_end_all / hint, 1u << 16);
        u64 pb = 0;
        while (pb < p_end_all) {
            u64 pe = (pb == 0) ? first : pb * 2;
            if (pe > p_end_all || p_end_all - pe < (1u << 16)) pe = p_end_all;
            u64 len = pe - pb;
            if (insert_variant() == 0) {      // one wavefront per chunk: >= ~16 K wavefronts when the phase is long
                u64 c = (len / 16384 + 63) & ~63ULL;
                u32 chunk = (u32)std::min<u64>(std::max<u64>(c, 256), wave_chunk_max());
                u64 n_waves = (len + chunk - 1) / chunk;
#ifdef AC_EMU
                launch(n_waves, InsertWaveEmuFunctor<W>{t, tb, pb, pe, chunk, istats.ptr(), ierr});
#else
                u64 blocks = (n_waves + 3) / 4;
                if (blocks > 0x7FFFFFFFULL) throw DeviceError("grid too large");
                hipLaunchKernelGGL(insert_wave_kernel<W>, dim3((unsigned)blocks), dim3(256), 0, 0, t, tb, pb, pe, chunk, istats.ptr(), ierr);
                AC_HIP_CHECK(hipGetLastError());
#endif
            } else {                          // one thread per chunk (kept for comparison: AC_INSERT_VARIANT=1)
                u32 chunk = 64;
                while (chunk < 1024 && len / chunk > (1u << 19)) chunk *= 2;   // >= ~0.5 M threads when the phase is long
                launch((len + chunk - 1) / chunk, InsertFunctor<W>{t, tb, pb, pe, chunk, istats.ptr(), ierr});
            }
            launches++;
            pb = pe;
        }
#ifndef AC_EMU
        AC_HIP_CHECK(hipEventRecord(e1, 0));
        AC_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0; AC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        tm->insert_kernel_ms += ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#endif
        tm->insert_launches += launches;
        std::vector<InsertStats> st = to_host(istats, 257);
        const bool ins_err = st[256].real != 0;
        st.pop_back();
        n_distinct = 0;
        u64 real = 0;
        for (auto& x : st) { n_distinct += x.claimed; real += x.real; }
        bool overflow = ins_err || (n_distinct * 10 > c * 7);
        if (!overflow) { tm->insert_real += real; tm->insert_positions += pt.n_text; break; }
        if (c >= next_pow2(pt.n_bases * 4 + 1024)) throw DeviceError("k-mer table overflow");
        c *= 4;
        tm->insert_kernel_ms = 0; tm->insert_launches = 0;
    }
    if (n_distinct >= 0xFFFFFFF0ULL) throw DeviceError("too many distinct k-mers for 32-bit novel indices");
    *slots_out = std::move(sl);
    *cap_out = c;
    *n_distinct_out = n_distinct;
}

// K3a: bit p set <=> text position p is the smallest occurrence of its canonical k-mer.
inline void GraphBuilder::Impl::novel_bitmap(const PackedText& t, const DBuf<u64>& sl, u64 c, DBuf<u64>* bm_out, DBuf<u64>* occ_out) {
    u64 n_bm_words = t.n_text / 64 + 1;
    bm_out->alloc(n_bm_words);
    bm_out->fill_bytes(0);
    occ_out->alloc((c + 63) / 64);
    occ_out->fill_bytes(0);
    launch_full(c, MarkFunctor{sl.ptr(), (u32*)bm_out->ptr(), occ_out->ptr()});
}

// Sharded phase 1 (after the local insert): novel runs of this rank -> fragment text + one meta record per fragment.
template <int W> void GraphBuilder::Impl::fragments() {
    DBuf<u64> lslots, lbm; u64 lcap = 0, ln = 0;
    insert<W>(loc, tm->local_hint, &lslots, &lcap, &ln);
    tm->n_local_distinct = ln;
    lap(&tm->insert);
    DBuf<u64> locc;
    novel_bitmap(loc, lslots, lcap, &lbm, &locc);
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
    DBuf<u64> fpos(n_frags), blen(n_frags + 1), boff(n_frags + 1);
    frag_meta.alloc(n_frags);
    launch(n_frags + 1, FragMetaFunctor{loc.ctx((int)k), run_start.ptr(), run_end.ptr(), n_runs, n_frags, fpos.ptr(), frag_meta.ptr(),
                                        blen.ptr(), counters.ptr() + 6});
    exclusive_scan_u64(blen.ptr(), boff.ptr(), n_frags + 1);
    frag_bytes = read_scalar(boff.ptr() + n_frags);
    frag_text.alloc(frag_bytes);
    launch((frag_bytes + 63) / 64, FragCopyFunctor{loc.d_text, fpos.ptr(), boff.ptr(), n_frags, frag_bytes, frag_text.ptr()});
    if (read_scalar(counters.ptr() + 6)) throw DeviceError("internal error: novel run outside a sequence");
    tm->n_fragments = n_frags; tm->fragment_bytes = frag_bytes;
    lap(&tm->fragments);
}

// K2, K3 on the graph text G: k-mer table and sorted novel list.
template <int W> void GraphBuilder::Impl::table() {
    PackedText& g = *G;
    check_sizes(g);
    insert<W>(g, tm->graph_hint, &slots, &cap, &N);
    tm->table_capacity = cap;
    tm->n_distinct = N;
    lap(G == &loc ? &tm->insert : &tm->union_insert);

    // K3 novel-position bitmap -> sorted novel list + rank support
    u64 n_bm_words = g.n_text / 64 + 1;
    novel_bitmap(g, slots, cap, &bm, &occ);
    DBuf<u32> wcnt(n_bm_words);
    wprefix.alloc(n_bm_words);
    launch(n_bm_words, PopcFunctor{bm.ptr(), wcnt.ptr()});
    exclusive_scan_u32(wcnt.ptr(), wprefix.ptr(), n_bm_words);
    npos.alloc(N);
    launch(n_bm_words, FillNovelFunctor{bm.ptr(), wprefix.ptr(), npos.ptr()});
    kinfo.alloc(N, true);
    lap(&tm->collect_sort);
}

// K5 out/in degrees of the novel k-mers [lo, hi) (a sharded build computes one slice per rank and all-gathers them).
template <int W> void GraphBuilder::Impl::degrees(u64 lo, u64 hi) {
    PackedText& g = *G;
    Table tb = graph_table();
    deg_lo = lo; deg_hi = hi;
    launch(hi - lo, DegreeFunctor<W>{g.ctx((int)k), tb, npos.ptr(), kinfo.ptr(), g.any_dots, lo});
    lap(&tm->degree);
}

// K6..K11 on G: first flags, unitigs in seed order, links by successor symbol.
template <int W> void GraphBuilder::Impl::unitigs() {
    PackedText& g = *G;
    TextCtx t = g.ctx((int)k);
    Table tb = graph_table();
    Novel nv{bm.ptr(), wprefix.ptr()};
    launch(g.n_seqs, FirstFunctor<W>{t, tb, nv, kinfo.ptr(), g.has_flags ? g.seq_flags.ptr() : nullptr});
    lap(&tm->degree);

    // K7 heads -> unitig ids
    head.alloc(N + 1, true); scan.alloc(N + 1, true);
    launch(N, HeadFunctor{npos.ptr(), kinfo.ptr(), head.ptr(), N});
    inclusive_scan_u32(head.ptr(), scan.ptr(), N);
    U = read_scalar(scan.ptr() + (N - 1));
    ustart.alloc((u64)U + 1);
    launch(N, UnitigStartFunctor{head.ptr(), scan.ptr(), ustart.ptr(), N});
    lap(&tm->segment);

    // K8 min canonical k-mer per unitig
    DBuf<MinVal<W>> umin(U);
    if constexpr (W <= 4) {
        DBuf<MinVal<W>> vals(N); DBuf<u32> seg(N);
        launch(N, CKeyFunctor<W>{t, npos.ptr(), scan.ptr(), vals.ptr(), seg.ptr()});
        reduce_by_segment(seg.ptr(), vals.ptr(), N, umin.ptr(), U, MinOp<W>(), counters.ptr() + 3);
    } else {      // wide keys: arg-min over indices, the keys recomputed from the text inside the operator
        DBuf<u32> umin_idx(U);
        segment_argmin(scan.ptr(), N, umin_idx.ptr(), U, MinIdxOp<W>{t, npos.ptr()}, counters.ptr() + 3);      // scan[i] = unitig of novel k-mer i
        launch(U, UnitigMinFunctor<W>{t, npos.ptr(), umin_idx.ptr(), umin.ptr()});
    }
    lap(&tm->minkey);

    // K9 seed order = rank of the smallest k-mer
    order.alloc(U);
    launch(U, IotaFunctor{order.ptr()});
    if constexpr (W <= 4) {
        sort_by_key_cmp(umin, order, U, MinValLess<W>());
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
    launch((u64)U * 2, LinksFunctor<W>{t, tb, nv, uc, order.ptr(), npos.ptr(), g.any_dots, links.ptr(), wlinks.ptr(), counters.ptr() + 3});
    lap(&tm->links);
}

// K10 paths of this rank's sequences against the graph: count, scan, write; first / last unitig of every path.
template <int W> void GraphBuilder::Impl::walk() {
    TextCtx t = loc.ctx((int)k), g = G->ctx((int)k);
    Table tb = graph_table();
    Novel nv{bm.ptr(), wprefix.ptr()};
    UnitigCtx uc{head.ptr(), scan.ptr(), rank.ptr(), uorient.ptr(), ustart.ptr(), ulen.ptr(), U, N};
    const u32 PC = 256;
    u64 n_walkers = (loc.n_text + PC - 1) / PC;
    depth.alloc(U, true); minpos_fwd.alloc(U); minpos_rev.alloc(U);
    minpos_fwd.fill_bytes(0xFF); minpos_rev.fill_bytes(0xFF);
    DBuf<u64> wcount(n_walkers + 1), woff(n_walkers + 1);
    DBuf<int32_t> stage(n_walkers * PC);
    DBuf<u32> seq_tid(loc.n_seqs), seq_j(loc.n_seqs);
    path_off.alloc((u64)loc.n_seqs + 1);
    wcount.fill_bytes(0);
    launch(n_walkers, PathWalkFunctor<W>{t, g, tb, nv, uc, wlinks.ptr(), PC, stage.ptr(), wcount.ptr(), seq_tid.ptr(), seq_j.ptr(),
                                        depth.ptr(), minpos_fwd.ptr(), minpos_rev.ptr(), counters.ptr() + 4});
    exclusive_scan_u64(wcount.ptr(), woff.ptr(), n_walkers + 1);
    n_ent = read_scalar(woff.ptr() + n_walkers);
    ent_val.alloc(n_ent);
    launch(n_walkers, PathCompactFunctor{stage.ptr(), wcount.ptr(), woff.ptr(), PC, ent_val.ptr()});
    launch(loc.n_seqs, PathOffFunctor{seq_tid.ptr(), seq_j.ptr(), woff.ptr(), path_off.ptr()});
    tm->n_path_entries = n_ent;
    copy_h2d(path_off.ptr() + loc.n_seqs, &n_ent, 8);
    fs0.alloc(U, true); fe0.alloc(U, true);
    launch(loc.n_seqs, PathEndsFunctor{ent_val.ptr(), path_off.ptr(), fs0.ptr(), fe0.ptr()});
    lap(&tm->paths);
}

// K12..K17 + D2H: sequences, link push order, expand_repeats, both renumberings, final numbering.  Needs depth,
// min positions and path ends of ALL sequences (reduced over ranks first in a sharded build).
template <int W> void GraphBuilder::Impl::tail(FinalGraph* out, bool want_graph, bool want_paths) {
    PackedText& g = *G;
    const u32 n_seqs = loc.n_seqs;
    // K12 sequences
    u64 total = N;   // sum of unitig lengths == number of distinct canonical k-mers
    DBuf<u8> useq(total);
    launch((total + 63) / 64, SeqFunctor{g.bits.ptr(), useq_off.ptr(), ustartpos.ptr(), ulen.ptr(), uorient.ptr(), U, total,
                                         (int)(k / 2), useq.ptr()});
    lap(&tm->seqs);

    // K13 link push order, K14 static analysis for expand_repeats, K15 first renumber_unitigs
    DBuf<int32_t> lord((u64)U * 10); DBuf<u8> lcnt((u64)U * 2);
    launch((u64)U * 2, LinkOrderFunctor{links.ptr(), lord.ptr(), lcnt.ptr(), counters.ptr() + 5});
    OrderedLinks L{lord.ptr(), lcnt.ptr()};
    DBuf<u8> fixed_start(U, true), fixed_end(U, true), cand((u64)U * 2);
    launch(U, FixedSpreadFunctor{fs0.ptr(), fe0.ptr(), L, fixed_start.ptr(), fixed_end.ptr()});
    launch((u64)U * 2, CandFunctor{L, fixed_start.ptr(), fixed_end.ptr(), cand.ptr()});
    DBuf<u32> order1(U);
    launch(U, IotaFunctor{order1.ptr()});
    DBuf<u32> renum_flag(1, true);
    renumber_sort(order1, U, ulen.ptr(), useq_off.ptr(), useq.ptr(), depth.ptr(), renum_flag.ptr());
    lap(&tm->analysis);

    // K17 expand_repeats, level-scheduled (see the kernels)
    DBuf<u64> coff(U), len64((u64)U + 1), noff((u64)U + 1);
    DBuf<u32> clen(U), pre_off(U, true), pre_len(U, true), post_off(U, true), post_len(U, true);
    copy_d2d(coff.ptr(), useq_off.ptr(), (size_t)U * 8);
    copy_d2d(clen.ptr(), ulen.ptr(), (size_t)U * 4);
    DBuf<u8> seq_alt(total), pool(std::min<u64>(2 * total + 4096, 0xFFFFFFF0ULL)), dirty((u64)U * 2);
    DBuf<u64> shifted(1); DBuf<u32> pool_used(1);
    copy_d2d(dirty.ptr(), cand.ptr(), (size_t)U * 2);
    u8* cur = useq.ptr(); u8* alt = seq_alt.ptr();
    u64 final_total = total;
    int passes = 0;
    u32 n_cand = 0, n_levels = 0;
    {
        u64 J = (u64)U * 2;
        DBuf<u32> cflag(J + 1), cpos(J + 1), prio(J);
        cflag.fill_bytes(0);       // [J] = 0: the exclusive scan then ends with the total
        launch(J, CandFlagFunctor{order1.ptr(), cand.ptr(), cflag.ptr()});
        exclusive_scan_u32(cflag.ptr(), cpos.ptr(), J + 1);
        n_cand = read_scalar(cpos.ptr() + J);
        if (n_cand == 0) {
            passes = 1;   // the reference's single pass that moves nothing
        } else {
            u64 C = n_cand;
            DBuf<u32> clist(C), level(C);
            prio.fill_bytes(0xFF);
            launch(J, CandListFunctor{order1.ptr(), cflag.ptr(), cpos.ptr(), clist.ptr(), prio.ptr()});
            launch(C, FillU32Functor{level.ptr(), 1u});
            DBuf<u32> changed(8);
            for (;;) {   // longest-path levels of the conflict DAG by relaxation (monotone, so stale reads only delay); eight
                changed.fill_bytes(0);       // sweeps per host check, converged when the last of them changed nothing
                for (int it = 0; it < 8; it++)
                    launch(C, LevelRelaxFunctor{L, cand.ptr(), clist.ptr(), prio.ptr(), level.ptr(), changed.ptr() + it});
                if (to_host(changed, 8)[7] == 0) break;
            }
            DBuf<u64> lkey(C);
            launch(C, LevelKeyFunctor{level.ptr(), lkey.ptr()});
            sort_pairs_u64_u32(lkey, clist, C, 64);
            // first index of every level; [0] = number of levels (levels beyond the table: a second, exact read)
            const u32 LV_TABLE = 1024;
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
            ExpState e{cur, coff.ptr(), clen.ptr(), pre_off.ptr(), pre_len.ptr(), post_off.ptr(), post_len.ptr(), pool.ptr(),
                       pool_used.ptr(), minpos_fwd.ptr(), minpos_rev.ptr(), dirty.ptr(), cand.ptr(), L, shifted.ptr()};
            pool_used.fill_bytes(0);
            u64 moved = 0;
            DBuf<u64> shifted2(2);
            for (;;) {   // two passes per host check: if the first moved nothing the second is an (uncounted) no-op
                shifted2.fill_bytes(0);
                for (int half = 0; half < 2; half++) {
                    e.shifted = shifted2.ptr() + half;
                    for (u32 lv = 1; lv <= n_levels; lv++)
                        launch_full((u64)(hb[lv + 1] - hb[lv]), ExpandFunctor{e, clist.ptr(), (u64)hb[lv], (u32)pool.size(), counters.ptr() + 7});
                }
                std::vector<u64> sh = to_host(shifted2, 2);
                moved += sh[0] + sh[1];
                if (sh[0] == 0) { passes += 1; break; }
                passes += 2;
                if (sh[1] == 0) break;
            }
            if (moved) {   // rewrite the sequences contiguously, once
                launch((u64)U + 1, ExpLenFunctor{e, len64.ptr(), U});
                exclusive_scan_u64(len64.ptr(), noff.ptr(), (u64)U + 1);
                final_total = read_scalar(noff.ptr() + U);
                launch((final_total + 63) / 64, MaterializeFunctor{e, noff.ptr(), U, final_total, alt});
                launch(U, ExpResetFunctor{e, noff.ptr()});
                std::swap(cur, alt);
                e.cur = cur;
            }
        }
    }
    tm->simplify_passes = (u32)passes; tm->n_candidates = n_cand; tm->n_levels = n_levels;
    lap(&tm->expand);

    // K15b second renumber_unitigs (graph_simplification.rs:39): a stable sort of the CURRENT order on the new
    // sequences; K16 per-unitig outputs in final order, links in get_links_for_gfa order, paths in final numbers
    DBuf<u32> order2(U);
    copy_d2d(order2.ptr(), order1.ptr(), (size_t)U * 4);
    renumber_sort(order2, U, clen.ptr(), coff.ptr(), cur, depth.ptr(), renum_flag.ptr());
    DBuf<u64> number_len(U), lcount((u64)U + 1), loff((u64)U + 1);
    DBuf<u8> meta((size_t)U * 20);
    u64* d_seq_begin = (u64*)meta.ptr();
    double* d_depth = (double*)(meta.ptr() + (size_t)U * 8);
    u32* d_seq_len = (u32*)(meta.ptr() + (size_t)U * 16);
    lcount.fill_bytes(0);
    // D2H on a second stream, each array as soon as it is final, straight into pinned blocks owned by the result; the
    // paths go in four chunks, each copied while the next is still being renumbered.
    SideStream& side = SideStream::get();
    SideStream::Guard side_guard;
    out->k = k;
    out->n_kmers = 2 * (u64)N;
    out->n_unitigs = U;
    if (want_graph) {
        out->seq_block = PinnedPool::get().alloc(final_total);
        side.after_main();     // sequences are final since the materialise step
        copy_d2h_async(out->seq_block.p, cur, final_total, side.stream());
    }
    launch(U, FinalMetaFunctor{order2.ptr(), coff.ptr(), clen.ptr(), depth.ptr(), lcnt.ptr(), number_len.ptr(), d_seq_begin, d_depth,
                               d_seq_len, lcount.ptr()});
    if (want_graph) {
        out->meta_block = PinnedPool::get().alloc((size_t)U * 20);
        side.after_main();
        copy_d2h_async(out->meta_block.p, meta.ptr(), (size_t)U * 20, side.stream());
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
    if (want_paths) out->path_block = PinnedPool::get().alloc(n_ent * 4);
    {
        const u64 n_waves = (n_ent + 4095) / 4096;
        const u64 per_chunk = std::max<u64>((n_waves + 3) / 4, 64);
        for (u64 w = 0; w < n_waves; w += per_chunk) {
            u64 cnt = std::min<u64>(per_chunk, n_waves - w);
            launch_full(cnt * 64, RemapFunctor{ent_val.ptr(), number_len.ptr(), path_off.ptr(), n_seqs, n_ent, sums.ptr(), w});
            if (want_paths) {
                u64 b = w * 4096, e2 = std::min<u64>((w + cnt) * 4096, n_ent);
                side.after_main();
                copy_d2h_async((int32_t*)out->path_block.p + b, ent_val.ptr() + b, (e2 - b) * 4, side.stream());
            }
        }
    }
    lap(&tm->finalize);

    std::vector<u64> h_sums = to_host(sums, n_seqs);
    out->path_off = to_host(path_off, (size_t)n_seqs + 1);
    std::vector<u32> errs = to_host(counters, 8);   // synchronises stream 0
    side.sync();                                    // ... and the copies: everything above has landed
    if (errs[7]) throw DeviceError("internal error: expand_repeats pool overflow");
    if (errs[3] || errs[4])
        throw DeviceError("internal error: inconsistent unitig ends (codes " + std::to_string(errs[3]) + "/" + std::to_string(errs[4]) + ")");
    if (want_graph) {
        out->seq_begin = (const u64*)out->meta_block.p;
        out->depth = (const double*)((const u8*)out->meta_block.p + (size_t)U * 8);
        out->seq_len = (const u32*)((const u8*)out->meta_block.p + (size_t)U * 16);
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
    if (getenv("AC_DEBUG_ARENA"))
        fprintf(stderr, "arena: used %.1f MB of %.1f MB (n_text %.1f MB)\n", Arena::device().total_used() / 1e6, Arena::device().capacity() / 1e6, loc.n_text / 1e6);
}

// The width-dependent stages behind one explicitly instantiated type per width.  The main unit only sees declarations, so it
// cannot instantiate (or inline) anything width-dependent itself.
template <int W> struct Stages {
    static void table(GraphBuilder::Impl& m);
    static void degrees(GraphBuilder::Impl& m, u64 lo, u64 hi);
    static void unitigs(GraphBuilder::Impl& m);
    static void walk(GraphBuilder::Impl& m);
    static void tail(GraphBuilder::Impl& m, FinalGraph* out, bool want_graph, bool want_paths);
    static void fragments(GraphBuilder::Impl& m);
};
#if AC_W_ONLY != 0 || defined(AC_EMU)
template <int W> void Stages<W>::table(GraphBuilder::Impl& m) { m.template table<W>(); }
template <int W> void Stages<W>::degrees(GraphBuilder::Impl& m, u64 lo, u64 hi) { m.template degrees<W>(lo, hi); }
template <int W> void Stages<W>::unitigs(GraphBuilder::Impl& m) { m.template unitigs<W>(); }
template <int W> void Stages<W>::walk(GraphBuilder::Impl& m) { m.template walk<W>(); }
template <int W> void Stages<W>::tail(GraphBuilder::Impl& m, FinalGraph* out, bool want_graph, bool want_paths) { m.template tail<W>(out, want_graph, want_paths); }
template <int W> void Stages<W>::fragments(GraphBuilder::Impl& m) { m.template fragments<W>(); }
#endif
#if AC_W_ONLY != 0
template struct Stages<AC_W_ONLY>;
#endif

#if AC_W_ONLY == 0
void device_warmup(int device) {
#ifndef AC_EMU
    AC_HIP_CHECK(hipSetDevice(device));
    void* p = nullptr;
    AC_HIP_CHECK(hipMalloc(&p, 4096));
    hipLaunchKernelGGL(functor_kernel<PackFunctor>, dim3(1), dim3(256), 0, 0, (u64)1, PackFunctor{(const u8*)p, 32, (u64*)((u8*)p + 1024), (u32*)((u8*)p + 2048)});
    (void)hipDeviceSynchronize();
    (void)hipFree(p);
#else
    (void)device;
#endif
}

// ---- GraphBuilder ------------------------------------------------------------------------------------------------
GraphBuilder::GraphBuilder(uint32_t k) : impl_(new Impl) {
    // A builder owns the arenas for its lifetime (the C ABI serialises builds): whatever the previous build
    // left there — device buffers and the pinned RawGraph its host tail has already consumed — is dead.
    Arena::device().reset();
    Arena::pinned_host().reset();
    impl_->k = k;
    if (k < 1 || (k % 2) == 0) throw DeviceError("k must be odd");
    if ((int)k > max_supported_k())
        throw DeviceError("k-mer sizes above " + std::to_string(max_supported_k()) + " are not supported by this build of the HIP backend");
}
GraphBuilder::~GraphBuilder() { delete impl_; }
uint64_t GraphBuilder::n_text() const { return impl_->loc.n_text; }
uint64_t GraphBuilder::n_bases() const { return impl_->loc.n_bases; }

void GraphBuilder::set_sequences_host(const std::vector<SeqView>& seqs) {
    double t0 = now_s();
    std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint16_t> d1, d2;
    std::vector<uint8_t> text = layout_text(seqs, impl_->k, &off, &len, &d1, &d2);
    impl_->loc.n_text = text.size();
    Arena::device().reserve(arena_estimate(text.size(), true));
    impl_->text_owned.alloc(text.size());
    copy_h2d(impl_->text_owned.ptr(), text.data(), text.size());
    impl_->loc.d_text = impl_->text_owned.ptr();
    impl_->loc.set_table(off, len, d1, d2);
    tm_.h2d = now_s() - t0;
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
    m.check_sizes(m.loc);
    m.loc.pack();
    m.lap(&tm_.pack);
    AC_DISPATCH_W(table, (*impl_))
    AC_DISPATCH_W(degrees, (*impl_, 0, m.N))
    AC_DISPATCH_W(unitigs, (*impl_))
    AC_DISPATCH_W(walk, (*impl_))
    AC_DISPATCH_W(tail, (*impl_, out, true, true))
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
    m.loc.pack();
    m.lap(&tm_.pack);
    AC_DISPATCH_W(fragments, (*impl_))
}
uint64_t GraphBuilder::local_distinct_count() const { return tm_.n_local_distinct; }
void GraphBuilder::set_distinct_upper_bound(uint64_t n) { impl_->distinct_upper = n; }
uint64_t GraphBuilder::fragment_text_bytes() const { return impl_->frag_bytes; }
uint64_t GraphBuilder::fragment_count() const { return impl_->n_frags; }
void GraphBuilder::fragments_export(void* d_text_out, void* d_meta_out) {
    copy_d2d(d_text_out, impl_->frag_text.ptr(), impl_->frag_bytes);
    copy_d2d(d_meta_out, impl_->frag_meta.ptr(), impl_->n_frags * 8);
    stream_sync();
}
void GraphBuilder::shard_build_union(uint32_t rank, uint32_t n_shards, const uint8_t* d_union_text, uint64_t n_union_text,
                                     const void* d_meta, uint64_t n_frags_total) {
    if (n_shards == 0 || rank >= n_shards) throw DeviceError("invalid rank / shard count");
    Impl& m = *impl_;
    m.t0 = now_s();
    if (n_frags_total == 0 || n_frags_total >= 0xFFFFFFF0ULL) throw DeviceError("invalid fragment count");
    std::vector<u64> meta(n_frags_total);
    copy_d2h(meta.data(), d_meta, n_frags_total * 8);
    std::vector<uint64_t> off(n_frags_total); std::vector<uint32_t> len(n_frags_total);
    std::vector<uint16_t> d1(n_frags_total), d2(n_frags_total); std::vector<uint8_t> flags(n_frags_total);
    u64 p = 1;
    for (u64 i = 0; i < n_frags_total; i++) {
        u64 r = meta[i];
        len[i] = (u32)r; d1[i] = (u16)((r >> 32) & 0xFF); d2[i] = (u16)((r >> 40) & 0xFF); flags[i] = (u8)((r >> 48) & 0xFF);
        if (len[i] == 0) throw DeviceError("invalid fragment record");
        off[i] = p;
        p += (u64)len[i] + impl_->k;
    }
    if (p != n_union_text) throw DeviceError("fragment records do not add up to the union text size");
    m.uni.d_text = d_union_text;
    m.uni.n_text = n_union_text;
    m.uni.set_table(off, len, d1, d2, &flags);
    m.G = &m.uni;
    tm_.graph_hint = n_shards;
    m.uni.pack();
    m.lap(&tm_.union_pack);
    AC_DISPATCH_W(table, (*impl_))
    // this rank's slice of the degree computation (the one kernel of the graph stage that is both heavy and
    // embarrassingly parallel over distinct k-mers); the caller all-gathers the slices
    u64 lo = m.N * rank / n_shards, hi = m.N * (rank + 1) / n_shards;
    AC_DISPATCH_W(degrees, (*impl_, lo, hi))
}
uint64_t GraphBuilder::distinct_count() const { return impl_->N; }
void GraphBuilder::degrees_export(void* d_out) {
    Impl& m = *impl_;
    copy_d2d(d_out, m.kinfo.ptr() + m.deg_lo, (m.deg_hi - m.deg_lo) * 4);
    stream_sync();
}
void GraphBuilder::shard_build_graph(const void* d_kinfo_all) {
    Impl& m = *impl_;
    m.t0 = now_s();
    if (d_kinfo_all) copy_d2d(m.kinfo.ptr(), d_kinfo_all, m.N * 4);
    else if (!(m.deg_lo == 0 && m.deg_hi == m.N)) throw DeviceError("degree slices of the other ranks are missing");
    AC_DISPATCH_W(unitigs, (*impl_))
    AC_DISPATCH_W(walk, (*impl_))
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
void GraphBuilder::shard_finish(FinalGraph* out, bool want_graph, bool want_paths) {
    impl_->t0 = now_s();
    AC_DISPATCH_W(tail, (*impl_, out, want_graph, want_paths))
}
uint64_t GraphBuilder::path_entry_count() const { return impl_->n_ent; }
void GraphBuilder::paths_export(void* d_out) {
    copy_d2d(d_out, impl_->ent_val.ptr(), impl_->n_ent * 4);
    stream_sync();
}

// =============================================================================================================
// sequence_end_repair on the device (compress.rs:202-270; SURVEY.md §8 "next" row f-1).
// The reference runs 2S regexes (one per sequence end: k/2 wildcards next to k/2 literal bases) over all 2S forward and
// reverse sequences: 4·S·B regex bytes.  Here the S·2 literals (and their reverse complements, which stand for the
// reverse haystacks) go into one small hash table and ONE pass over the packed forward text finds every occurrence;
// the few thousand hits are turned into leftmost non-overlapping matches, tallied and chosen (find_best_match: fewest
// dots, most frequent, first alphabetically) on the host, and the winners are patched into the device text in place.
struct WindowGatherFunctor {     // m bytes from each listed text position
    const u8* text; u64 n_text; const u64* pos; u32 m; u64 n; u8* out;
    AC_HD void operator()(u64 idx) const {
        u64 w = idx / m, t = idx % m;
        u64 p = pos[w] + t;
        out[idx] = p < n_text ? text[p] : (u8)'$';
    }
};
struct WindowPatchFunctor {      // the reverse: m bytes into each listed text position
    u8* text; const u64* pos; u32 m; const u8* src;
    AC_HD void operator()(u64 idx) const { text[pos[idx / m] + idx % m] = src[idx]; }
};
template <int WL> struct EndScanFunctor {   // a thread owns the literal-length windows starting in 256 consecutive text positions
    const u64* bits; const u64* mask; u64 n_text; int lit;
    const u64* filter; const u64* tkeys; const u32* tent; u64 tmask;
    u64* hits; u32 cap; u32* n_hits;
    AC_D void operator()(u64 tid) const {
        u64 p0 = tid * 256, p1 = p0 + 256;
        if (p1 > n_text) p1 = n_text;
        u64 bend = p1 + (u64)lit - 1;
        if (bend > n_text) bend = n_text;
        const Key<WL> km = key_kmask<WL>(lit);
        Key<WL> key;
#pragma unroll
        for (int i = 0; i < WL; i++) key.w[i] = 0;
        int run = 0;
        u64 wbits = 0; u32 wmask = 0;       // the current 32-base word of the packed text and its mask bits, in registers
        for (u64 b = p0; b < bend; b++) {
            const u32 o = (u32)(b & 31);
            if (o == 0 || b == p0) {
                wbits = bits[b >> 5];
                wmask = (u32)(mask[b >> 6] >> (32 * ((b >> 5) & 1)));
            }
            if ((wmask >> o) & 1) { run = 0; continue; }
            key_roll_fwd<WL>(key, (u32)(wbits >> (62 - 2 * o)) & 3u, km);
            if (++run < lit) continue;
            u64 h = key_hash<WL>(key);
            u64 fb = h >> 44;                                       // 2^20-bit filter
            if (!((filter[fb >> 6] >> (fb & 63)) & 1)) continue;
            for (u64 s = h & tmask;; s = (s + 1) & tmask) {
                u32 e = tent[s];
                if (e == 0xFFFFFFFFu) break;
                bool eq = true;
#pragma unroll
                for (int i = 0; i < WL; i++) eq = eq && tkeys[WL * s + i] == key.w[i];
                if (eq) {
                    u32 i = atomic_add32(n_hits, 1u);
                    if (i < cap) hits[i] = ((u64)e << 40) | (b + 1 - (u64)lit);
                    break;
                }
            }
        }
    }
};

static inline char repair_comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == '.' ? '.' : 'N'; }   // misc.rs:358-376

template <int WL>
static void end_repair_impl(uint32_t k, uint8_t* d_text, uint64_t n_text, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len,
                            std::vector<uint16_t>* d1, std::vector<uint16_t>* d2, RepairTimings* tm) {
    double t_begin = now_s();
    const u32 S = (u32)off.size();
    const u32 m = k - 1, h = k / 2, lit = m - h;
    if (tm) *tm = RepairTimings();
    if (m == 0 || S == 0) return;
    if (lit > 32u * WL || lit == 0) throw DeviceError("end repair: unsupported k");
    if (n_text >= POS_MASK) throw DeviceError("input too large for 40-bit text positions");
    Arena::device().reset();     // nothing of an earlier build is alive while the repair runs
    Arena::device().reserve(arena_estimate(n_text, false));
    auto plen = [&](u32 s) { return (u64)len[s] + k - 1; };

    // the 2S pattern windows: first / last m bytes of every padded sequence (pattern 2s = start, 2s+1 = end)
    std::vector<u64> wpos(2 * (size_t)S);
    for (u32 s = 0; s < S; s++) { wpos[2 * s] = off[s]; wpos[2 * s + 1] = off[s] + plen(s) - m; }
    DBuf<u64> d_wpos(2 * (size_t)S);
    DBuf<u8> d_win((size_t)2 * S * m);
    copy_h2d(d_wpos.ptr(), wpos.data(), wpos.size() * 8);
    launch((u64)2 * S * m, WindowGatherFunctor{d_text, n_text, d_wpos.ptr(), m, (u64)2 * S, d_win.ptr()});
    std::vector<u8> win((size_t)2 * S * m);
    copy_d2h(win.data(), d_win.ptr(), win.size());

    // literal table: key -> entry; entry -> the (pattern, orientation) pairs with that literal
    struct Use { u32 pid; u32 rev; };
    std::vector<std::vector<Use>> uses;
    std::vector<Key<WL>> ekeys;
    auto code_of = [](u8 ch, bool* ok) -> u32 { if (!(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T')) *ok = false; return ((ch >> 1) ^ (ch >> 2)) & 3u; };
    const Key<WL> km = key_kmask<WL>((int)lit);
    u64 tcap = next_pow2((u64)8 * S + 16);
    std::vector<u64> tkeys((size_t)WL * tcap, 0); std::vector<u32> tent(tcap, 0xFFFFFFFFu);
    std::vector<u64> filter((1u << 20) / 64, 0);
    auto add = [&](const Key<WL>& key, u32 pid, u32 rev) {
        u64 hh = key_hash<WL>(key);
        for (u64 s = hh & (tcap - 1);; s = (s + 1) & (tcap - 1)) {
            if (tent[s] == 0xFFFFFFFFu) {
                tent[s] = (u32)ekeys.size();
                for (int i = 0; i < WL; i++) tkeys[(size_t)WL * s + i] = key.w[i];
                ekeys.push_back(key); uses.push_back({});
                u64 fb = hh >> 44; filter[fb >> 6] |= 1ULL << (fb & 63);
            }
            if (key_eq<WL>(ekeys[tent[s]], key)) { uses[tent[s]].push_back(Use{pid, rev}); return; }
        }
    };
    for (u32 pid = 0; pid < 2 * S; pid++) {
        const u8* w = &win[(size_t)pid * m];
        const u8* L = (pid & 1) ? w : w + h;          // start pattern: h wildcards then the literal; end pattern: literal first
        Key<WL> key;
        for (int i = 0; i < WL; i++) key.w[i] = 0;
        bool ok = true;
        for (u32 i = 0; i < lit; i++) key_roll_fwd<WL>(key, code_of(L[i], &ok), km);
        if (!ok) throw DeviceError("end repair: a sequence end holds something else than bases");
        add(key, pid, 0);
        add(key_rc<WL>(key, (int)lit), pid, 1);         // an occurrence of rc(L) in a forward sequence = an occurrence of L in its reverse
    }
    if (ekeys.size() >= (1u << 24)) throw DeviceError("end repair: too many patterns");

    // one pass over the packed text
    PackedText pt;
    pt.d_text = d_text; pt.n_text = n_text;
    pt.pack();
    DBuf<u64> d_filter(filter.size()), d_tkeys(tkeys.size()); DBuf<u32> d_tent(tent.size()), d_nhits(1);
    copy_h2d(d_filter.ptr(), filter.data(), filter.size() * 8);
    copy_h2d(d_tkeys.ptr(), tkeys.data(), tkeys.size() * 8);
    copy_h2d(d_tent.ptr(), tent.data(), tent.size() * 4);
    u32 cap = 1u << 20;
    std::vector<u64> hits;
    double t_scan = now_s();
    for (;;) {
        DBuf<u64> d_hits(cap);
        d_nhits.fill_bytes(0);
        launch((n_text + 255) / 256, EndScanFunctor<WL>{pt.bits.ptr(), pt.mask.ptr(), n_text, (int)lit, d_filter.ptr(), d_tkeys.ptr(), d_tent.ptr(),
                                                   tcap - 1, d_hits.ptr(), cap, d_nhits.ptr()});
        u32 n = read_scalar(d_nhits.ptr());
        if (n > cap) { if (n >= 0xFFFFFFF0u) throw DeviceError("end repair: too many literal occurrences"); cap = n; continue; }
        hits.resize(n);
        copy_d2h(hits.data(), d_hits.ptr(), (size_t)n * 8);
        break;
    }
    if (tm) { tm->scan_ms = (now_s() - t_scan) * 1e3; tm->hits = hits.size(); tm->patterns = 2 * S; }

    // hits -> candidate matches (pattern, haystack, start) in haystack coordinates (regex semantics: the h wildcards may
    // cover dots, the literal only bases; the match must lie inside the haystack)
    struct Cand { u32 pid; u32 hay; u64 i; u64 fpos; };      // fpos: text position of the forward window the match spells
    std::vector<Cand> cands;
    for (u64 hv : hits) {
        u32 e = (u32)(hv >> 40);
        u64 jt = hv & POS_MASK;
        u32 s = (u32)(std::upper_bound(off.begin(), off.end(), jt) - off.begin()) - 1;
        u64 jj = jt - off[s], pl = plen(s);
        if (jj + lit > pl) continue;      // cannot happen: masked separators end every run
        for (const Use& u : uses[e]) {
            u64 j = u.rev ? pl - jj - lit : jj;                 // literal start in the haystack's own coordinates
            bool start_pat = (u.pid & 1) == 0;
            if (start_pat && j < h) continue;
            u64 i = start_pat ? j - h : j;
            if (i + m > pl) continue;
            u64 f = u.rev ? pl - i - m : i;                     // where the forward sequence spells this match
            cands.push_back(Cand{u.pid, 2 * s + u.rev, i, off[s] + f});
        }
    }
    std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) {
        if (a.pid != b.pid) return a.pid < b.pid;
        if (a.hay != b.hay) return a.hay < b.hay;
        return a.i < b.i;
    });
    std::vector<Cand> acc;       // find_iter: leftmost, non-overlapping, per (regex, haystack)
    for (size_t a = 0; a < cands.size();) {
        size_t b = a;
        u64 next_free = 0;
        while (b < cands.size() && cands[b].pid == cands[a].pid && cands[b].hay == cands[a].hay) {
            if (cands[b].i >= next_free) { acc.push_back(cands[b]); next_free = cands[b].i + m; }
            b++;
        }
        a = b;
    }
    if (tm) tm->matches = acc.size();

    // the matched strings
    std::vector<u64> apos(acc.size());
    for (size_t i = 0; i < acc.size(); i++) apos[i] = acc[i].fpos;
    std::vector<u8> astr(acc.size() * (size_t)m);
    if (!acc.empty()) {
        DBuf<u64> d_apos(acc.size()); DBuf<u8> d_astr(astr.size());
        copy_h2d(d_apos.ptr(), apos.data(), apos.size() * 8);
        launch((u64)astr.size(), WindowGatherFunctor{d_text, n_text, d_apos.ptr(), m, (u64)acc.size(), d_astr.ptr()});
        copy_d2h(astr.data(), d_astr.ptr(), astr.size());
    }
    // find_best_match (compress.rs:239-270) per pattern, then the splices (compress.rs:225,234)
    std::vector<u8> patch((size_t)2 * S * m);
    size_t a = 0;
    for (u32 pid = 0; pid < 2 * S; pid++) {
        std::map<std::string, u32> tally;
        while (a < acc.size() && acc[a].pid == pid) {
            std::string str((const char*)&astr[a * (size_t)m], m);
            if (acc[a].hay & 1) { std::reverse(str.begin(), str.end()); for (char& c : str) c = repair_comp(c); }
            tally[str]++;
            a++;
        }
        if (tally.empty()) throw DeviceError("internal error: an end-repair pattern does not match its own sequence");
        const std::string* best = nullptr; size_t best_dots = 0; u32 best_cnt = 0;
        for (auto& kv : tally) {   // std::map iterates alphabetically: the first of equals wins the tie
            size_t dots = (size_t)std::count(kv.first.begin(), kv.first.end(), '.');
            if (!best || dots < best_dots || (dots == best_dots && kv.second > best_cnt)) { best = &kv.first; best_dots = dots; best_cnt = kv.second; }
        }
        memcpy(&patch[(size_t)pid * m], best->data(), m);
    }
    // starts first, then ends (the reference splices in that order)
    DBuf<u8> d_patch(patch.size());
    copy_h2d(d_patch.ptr(), patch.data(), patch.size());
    for (int which = 0; which < 2; which++) {
        std::vector<u64> ppos(S); std::vector<u8> psrc((size_t)S * m);
        for (u32 s = 0; s < S; s++) { ppos[s] = wpos[2 * s + which]; memcpy(&psrc[(size_t)s * m], &patch[(size_t)(2 * s + which) * m], m); }
        DBuf<u64> d_ppos(S); DBuf<u8> d_psrc(psrc.size());
        copy_h2d(d_ppos.ptr(), ppos.data(), ppos.size() * 8);
        copy_h2d(d_psrc.ptr(), psrc.data(), psrc.size());
        launch((u64)S * m, WindowPatchFunctor{d_text, d_ppos.ptr(), m, d_psrc.ptr()});
        stream_sync();
    }
    // surviving dots (what layout_text would count on the repaired sequences)
    for (u32 s = 0; s < S; s++) {
        const u8* st = &patch[(size_t)(2 * s) * m]; const u8* en = &patch[(size_t)(2 * s + 1) * m];
        u16 a1 = 0, b1 = 0;
        while (a1 < m && st[a1] == '.') a1++;
        while (b1 < m && en[m - 1 - b1] == '.') b1++;
        (*d1)[s] = a1; (*d2)[s] = b1;
    }
    if (tm) tm->total = now_s() - t_begin;
}

void end_repair_device(uint32_t k, uint8_t* d_text, uint64_t n_text, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len,
                       std::vector<uint16_t>* d1, std::vector<uint16_t>* d2, RepairTimings* tm) {
    if (k < 1 || (int)k > max_supported_k()) throw DeviceError("end repair: unsupported k");
    if ((k - 1) - k / 2 <= 64) end_repair_impl<2>(k, d_text, n_text, off, len, d1, d2, tm);
    else end_repair_impl<8>(k, d_text, n_text, off, len, d1, d2, tm);      // literals of up to 250 bases (k <= 501)
}

// =============================================================================================================
// pairwise_contig_distances (cluster.rs:132-157; SURVEY.md §8 "next" row f-3): the first step of `autocycler cluster` on the
// graph `compress` has just built.  distance(a, b) = 1 - len(U_a ∩ U_b) / len(U_a), U_s = set of unitigs on the path of s
// (strand and multiplicity ignored).  One bitset of U bits per sequence; one wavefront per pair ANDs the two bitsets and sums
// the lengths of the common unitigs (integers: exact, any summation order).
struct PathBitsFunctor {      // one thread per path entry
    const int32_t* path; const u64* path_off; u32 n_seqs; u64 n_ent; u64 words; u32* bits32;
    AC_D void operator()(u64 i) const {
        u32 lo = 0, hi = n_seqs;   // largest s with path_off[s] <= i
        while (hi - lo > 1) { u32 mid = lo + ((hi - lo) >> 1); if (path_off[mid] <= i) lo = mid; else hi = mid; }
        u32 u = idx_of(path[i]);
        atomic_or32(&bits32[(u64)lo * words * 2 + (u >> 5)], 1u << (u & 31));
    }
};
struct PairLenFunctor {       // thread t of pair p = t / 64 takes the words lane, lane + 64, ... of the two bitsets
    const u64* bits; u64 words; const u32* ulen; u32 n_seqs; u64* ab;
    AC_D void operator()(u64 tid, bool valid) const {
        if (!valid) return;
        u64 pair = tid >> 6;
        u32 lane = (u32)(tid & 63);
        u32 a = (u32)(pair / n_seqs), b = (u32)(pair % n_seqs);
        const u64* A = bits + (u64)a * words; const u64* B = bits + (u64)b * words;
        u64 sum = 0;
        for (u64 w = lane; w < words; w += 64) {
            u64 x = A[w] & B[w];
            while (x) { u64 low = x & (~x + 1); sum += ulen[w * 64 + (u64)popc64(low - 1)]; x ^= low; }
        }
#ifndef AC_EMU
#pragma unroll
        for (int o = 32; o; o >>= 1) sum += (u64)__shfl_xor((unsigned long long)sum, o);
        if (lane == 0) ab[pair] = sum;
#else
        atomic_add64(&ab[pair], sum);
#endif
    }
};
void pairwise_distances_device(const FinalGraph& g, uint32_t n_seqs, double* out) {
    if (n_seqs == 0 || g.path_off.size() != (size_t)n_seqs + 1) throw DeviceError("pairwise distances: the graph holds no paths");
    const u32 U = g.n_unitigs;
    const u64 n_ent = g.n_path, words = ((u64)U + 63) / 64;
    Arena::device().reset();
    DBuf<int32_t> d_path(n_ent); DBuf<u64> d_off((size_t)n_seqs + 1); DBuf<u32> d_len(words * 64);
    DBuf<u64> bits((u64)n_seqs * words), ab((u64)n_seqs * n_seqs);
    copy_h2d(d_path.ptr(), g.path, n_ent * 4);
    copy_h2d(d_off.ptr(), g.path_off.data(), ((size_t)n_seqs + 1) * 8);
    d_len.fill_bytes(0);
    copy_h2d(d_len.ptr(), g.seq_len, (size_t)U * 4);
    bits.fill_bytes(0); ab.fill_bytes(0);
    launch(n_ent, PathBitsFunctor{d_path.ptr(), d_off.ptr(), n_seqs, n_ent, words, (u32*)bits.ptr()});
    launch_full((u64)n_seqs * n_seqs * 64, PairLenFunctor{bits.ptr(), words, d_len.ptr(), n_seqs, ab.ptr()});
    std::vector<u64> h = to_host(ab, (size_t)n_seqs * n_seqs);
    for (u32 a = 0; a < n_seqs; a++) {
        double a_len = (double)(uint32_t)h[(size_t)a * n_seqs + a];        // |U_a ∩ U_a|; the reference sums a_len in u32
        for (u32 b = 0; b < n_seqs; b++) out[(size_t)a * n_seqs + b] = 1.0 - ((double)h[(size_t)a * n_seqs + b] / a_len);
    }
}

#endif   // AC_W_ONLY == 0

}  // namespace ac
