// K-mer primitives shared by all kernels of the compress hot path (gfx950).
//
// Reference semantics being replaced (file:line relative to /root/reference/src):
//   * k-mers are byte strings over the 5-symbol alphabet ".ACGT" (kmer_graph.rs:23); '.' only occurs as
//     sequence-end padding (sequence.rs:44-46) and sorts before 'A' (kmer_graph.rs:266-282).
//   * every k-mer is stored on both strands (kmer_graph.rs:110-132); k is odd, so X != rc(X).
// MI355X representation: 2-bit packed bases (A,C,G,T = 0..3, numeric order == byte order), a 1-bit
// "not a base" mask, and one *canonical* key per strand pair.  The canonical key of a k-mer is the
// smaller of (X, rc X) in the reference's byte order.  Dot k-mers ("...ACGT" / "ACGT...") are kept
// exact through a leading-dot count folded into the top byte of the key:
//     key.w[0] bits 63..56 = 255 - ld     (ld = number of leading dots of the canonical form)
// so that integer comparison of keys == byte-lexicographic comparison of the canonical strings
// (more leading dots sort first; real k-mers, ld = 0, sort after every dot k-mer).
//
// AC_EMU: the same code compiled by g++ as a serial CPU emulation.  It exists only so the CPU test
// suite (tests/, -m "not gpu") can exercise the kernels' logic; the product library never contains it.
#pragma once
#include <cstdint>
#include <cstddef>

#ifdef AC_EMU
#define AC_HD inline
#define AC_D inline
#else
#include <hip/hip_runtime.h>
#define AC_HD __host__ __device__ inline
#define AC_D __device__ inline
#endif

// Loops over the W words of a key: fully unrolled for every width, so that a key lives in registers with constant indices.  (Until round 6
// the wide keys — k > 123: 8 or 16 words — kept them as loops, because straight-line code for every operation inlined into every kernel of the
// ONE translation unit made the compile take tens of minutes; the looped keys lived in scratch memory: 736 bytes per lane in the insert,
// 27x its needed traffic on D' at k = 201.  With the stages in a unit of their own the unrolled 8- and 16-word builds take 50-60 s.)
#define AC_UNROLL_W _Pragma("unroll (W <= 16 ? 16 : 1)")
#define AC_UNROLL_FULL _Pragma("unroll")

namespace ac {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

// Words needed for a k-mer key: 2k bits + the 8-bit dot field.
constexpr int words_for_k(int k) { return (2 * k + 8 + 63) / 64; }

template <int W>
struct Key {
    u64 w[W];  // big-endian multiword integer: w[0] most significant
};

template <int W> AC_HD bool key_eq(const Key<W>& a, const Key<W>& b) {
    bool e = true;
AC_UNROLL_W
    for (int i = 0; i < W; i++) e = e && (a.w[i] == b.w[i]);
    return e;
}
template <int W> AC_HD bool key_lt(const Key<W>& a, const Key<W>& b) {
AC_UNROLL_W
    for (int i = 0; i < W; i++) {
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    }
    return false;
}

// Reverse the order of the 32 2-bit groups of a word.
AC_HD u64 rev2_64(u64 x) {
#ifdef AC_EMU
    x = __builtin_bswap64(x);
#else
    x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x >> 8) & 0x00FF00FF00FF00FFULL);
    x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x >> 16) & 0x0000FFFF0000FFFFULL);
    x = (x << 32) | (x >> 32);
#endif
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    return x;
}

// Multiword logical shift right by s bits (0 <= s < 64*W).  Constant word indices only: a run-time index into the key
// makes the compiler keep every key that flows through here in scratch (private) memory — 24 bytes per lane, written and
// re-read for every candidate k-mer, which showed up as ~1.4 GB of HBM writes per build in the degree kernel alone.
template <int W> AC_HD Key<W> key_shr(const Key<W>& a, int s) {
    Key<W> r;
    const int ws = s >> 6, bs = s & 63;
    if constexpr (W <= 4) {
AC_UNROLL_W
        for (int i = 0; i < W; i++) {
            u64 lo = 0, hi = 0;
AC_UNROLL_W
            for (int j = 0; j < W; j++) {
                lo = (j == i - ws) ? a.w[j] : lo;
                hi = (j == i - ws - 1) ? a.w[j] : hi;
            }
            r.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
        }
    } else {   // wide keys (k > 123, rare): W^2 selects per shift would blow the code up; a run-time index (scratch memory) is fine here
        for (int i = 0; i < W; i++) {
            int src = i - ws;
            u64 lo = (src >= 0) ? a.w[src] : 0;
            u64 hi = (src - 1 >= 0) ? a.w[src - 1] : 0;
            r.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
        }
    }
    return r;
}

// All-ones in the low 2k bits.
template <int W> AC_HD Key<W> key_kmask(int k) {
    Key<W> m;
    int bits = 2 * k;
AC_UNROLL_W
    for (int i = 0; i < W; i++) {
        int lo_bit = 64 * (W - 1 - i);  // bit index of this word's LSB
        int n = bits - lo_bit;
        m.w[i] = n <= 0 ? 0 : (n >= 64 ? ~0ULL : ((1ULL << n) - 1));
    }
    return m;
}

// Mask clearing `nlead` leading and `ntrail` trailing bases of a right-aligned k-mer value.
template <int W> AC_HD Key<W> key_inner_mask(int k, int nlead, int ntrail) {
    Key<W> hi = key_kmask<W>(k - nlead);     // ones for the low 2(k-nlead) bits
    Key<W> lo = key_kmask<W>(ntrail);        // ones for the low 2*ntrail bits
    Key<W> m;
AC_UNROLL_W
    for (int i = 0; i < W; i++) m.w[i] = hi.w[i] & ~lo.w[i];
    return m;
}

// Reverse complement of a right-aligned 2k-bit value (no dots).
template <int W> AC_HD Key<W> key_rc(const Key<W>& a, int k) {
    Key<W> t;
AC_UNROLL_W
    for (int i = 0; i < W; i++) t.w[i] = rev2_64(a.w[W - 1 - i]);
    Key<W> r = key_shr<W>(t, 64 * W - 2 * k);
    Key<W> m = key_kmask<W>(k);
AC_UNROLL_W
    for (int i = 0; i < W; i++) r.w[i] = (~r.w[i]) & m.w[i];
    return r;
}

// Rolling updates.  fwd <- (fwd << 2 | c) & mask ;  rc <- (rc >> 2) | ((3-c) << 2(k-1)).
template <int W> AC_HD void key_roll_fwd(Key<W>& a, u32 c, const Key<W>& kmask) {
AC_UNROLL_W
    for (int i = 0; i < W - 1; i++) a.w[i] = ((a.w[i] << 2) | (a.w[i + 1] >> 62)) & kmask.w[i];
    a.w[W - 1] = ((a.w[W - 1] << 2) | (u64)c) & kmask.w[W - 1];
}
template <int W> AC_HD void key_roll_rc(Key<W>& a, u32 c, int k) {
AC_UNROLL_W
    for (int i = W - 1; i > 0; i--) a.w[i] = (a.w[i] >> 2) | (a.w[i - 1] << 62);
    a.w[0] >>= 2;
    int bit = 2 * (k - 1);
    int wi = W - 1 - (bit >> 6);
    u64 v = (u64)(3 - c) << (bit & 63);
AC_UNROLL_W
    for (int i = 0; i < W; i++) a.w[i] |= (i == wi) ? v : 0;   // constant indices only (see key_shr)
}

// ---- packed text --------------------------------------------------------------------------------
// bits: base i lives in word i/32 at shift 62-2*(i%32) (first base most significant).
// mask: position i lives in word i/64 at bit i%64; 1 = '.' padding or a sequence separator.
AC_HD u32 text_code(const u64* bits, u64 i) { return (u32)(bits[i >> 5] >> (62 - 2 * (int)(i & 31))) & 3u; }
AC_HD u32 text_mask(const u64* mask, u64 i) { return (u32)(mask[i >> 6] >> (i & 63)) & 1u; }

// k bases starting at text position p, right-aligned.  Reads words p/32 .. p/32+W (the packed buffer
// carries W+1 words of slack).
template <int W> AC_HD Key<W> text_extract(const u64* bits, u64 p, int k) {
    u64 j0 = p >> 5;
    int o = 2 * (int)(p & 31);
    Key<W> l;
AC_UNROLL_W
    for (int i = 0; i < W; i++) {
        u64 a = bits[j0 + i], b = bits[j0 + i + 1];
        l.w[i] = o ? ((a << o) | (b >> (64 - o))) : a;
    }
    return key_shr<W>(l, 64 * W - 2 * k);
}
// Number of mask bits in [p, p+n), n <= 64*8.
AC_HD int text_mask_count(const u64* mask, u64 p, int n) {
    int cnt = 0;
    u64 i = p, end = p + (u64)n;
    while (i < end) {
        u64 wi = i >> 6;
        int b = (int)(i & 63);
        int take = (int)((end - i) < (u64)(64 - b) ? (end - i) : (u64)(64 - b));
        u64 m = mask[wi] >> b;
        if (take < 64) m &= ((1ULL << take) - 1);
#ifdef AC_EMU
        cnt += __builtin_popcountll(m);
#else
        cnt += __popcll(m);
#endif
        i += (u64)take;
    }
    return cnt;
}

// ---- word-at-a-time run matching over the packed text ---------------------------------------------
// Used by the run-following insert (graph_build.hip): once a k-mer occurrence at p is known to equal an
// EARLIER occurrence at q, every following position whose next base also agrees is an occurrence of a
// k-mer that already has an earlier occurrence, so it needs no table access at all.
AC_HD int clz64(u64 x) {
#ifdef AC_EMU
    return x ? __builtin_clzll(x) : 64;
#else
    return x ? __clzll((long long)x) : 64;
#endif
}
AC_HD int ctz64(u64 x) {
#ifdef AC_EMU
    return x ? __builtin_ctzll(x) : 64;
#else
    return x ? (__ffsll((long long)x) - 1) : 64;
#endif
}
AC_HD int ctz32(u32 x) {
#ifdef AC_EMU
    return x ? __builtin_ctz(x) : 32;
#else
    return x ? (__ffs((int)x) - 1) : 32;
#endif
}
AC_HD u32 brev32(u32 x) {
#ifdef AC_EMU
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
#else
    return __brev(x);
#endif
}
// 32 bases starting at p (first base most significant).
AC_HD u64 text_word(const u64* bits, u64 p) {
    u64 j = p >> 5;
    int o = 2 * (int)(p & 31);
    u64 a = bits[j];
    return o ? ((a << o) | (bits[j + 1] >> (64 - o))) : a;
}
// Mask bits of positions p .. p+31 (bit i = position p+i).
AC_HD u32 mask_word(const u64* mask, u64 p) {
    u64 j = p >> 6;
    int o = (int)(p & 63);
    u64 a = mask[j] >> o;
    if (o > 32) a |= mask[j + 1] << (64 - o);
    return (u32)a;
}
// Reverse-complemented view: complement of bases p, p-1, ..., p-31 (comp(base p) most significant), and
// the mask bits in the same order (bit i = position p-i).  Positions below 0 read as masked.
AC_HD u64 text_word_rc(const u64* bits, u64 p) {
    u64 w = (p >= 31) ? text_word(bits, p - 31) : (text_word(bits, 0) >> (2 * (31 - (int)p)));
    return ~rev2_64(w);
}
AC_HD u32 mask_word_rev(const u64* mask, u64 p) {
    u32 m = (p >= 31) ? mask_word(mask, p - 31)
                      : ((mask_word(mask, 0) << (31 - (int)p)) | ((1u << (31 - (int)p)) - 1u));
    return brev32(m);
}
// Number of i in [0, maxlen) with base[a+j] == base[b+j] and neither masked, for all j <= i.
AC_HD u64 match_run_fwd(const u64* bits, const u64* mask, u64 a, u64 b, u64 maxlen) {
    u64 n = 0;
    while (n < maxlen) {
        u64 x = text_word(bits, a + n) ^ text_word(bits, b + n);
        u32 m = mask_word(mask, a + n) | mask_word(mask, b + n);
        int d = clz64(x) >> 1, e = ctz32(m);
        int t = d < e ? d : e;
        n += (u64)t;
        if (t < 32) break;
    }
    return n < maxlen ? n : maxlen;
}
// Number of i in [0, maxlen) with comp(base[a+j]) == base[b-j] and neither masked, for all j <= i.
AC_HD u64 match_run_rev(const u64* bits, const u64* mask, u64 a, u64 b, u64 maxlen) {
    u64 n = 0;
    while (n < maxlen) {
        if (b < n) break;   // ran off the start of the text (position 0 is a separator, so unreachable)
        u64 x = text_word(bits, a + n) ^ text_word_rc(bits, b - n);
        u32 m = mask_word(mask, a + n) | mask_word_rev(mask, b - n);
        int d = clz64(x) >> 1, e = ctz32(m);
        int t = d < e ? d : e;
        n += (u64)t;
        if (t < 32) break;
    }
    return n < maxlen ? n : maxlen;
}

// Number of leading bases (0..32) on which the 32-base word at a agrees with the word at b (fwd) / with the reverse-
// complemented view ending at b (rev), stopping at the first masked position of either side.
AC_HD int match_word_fwd(const u64* bits, const u64* mask, u64 a, u64 b) {
    u64 x = text_word(bits, a) ^ text_word(bits, b);
    u32 m = mask_word(mask, a) | mask_word(mask, b);
    int d = clz64(x) >> 1, e = ctz32(m);
    return d < e ? d : e;
}
AC_HD int match_word_rev(const u64* bits, const u64* mask, u64 a, u64 b) {
    u64 x = text_word(bits, a) ^ text_word_rc(bits, b);
    u32 m = mask_word(mask, a) | mask_word_rev(mask, b);
    int d = clz64(x) >> 1, e = ctz32(m);
    return d < e ? d : e;
}

// ---- extended k-mers (with dots) ----------------------------------------------------------------
template <int W>
struct XKmer {
    Key<W> fwd;  // text orientation, dots coded 0
    int ld, td;  // leading / trailing dot counts (never both > 0: L >= k)
};

// Canonical unified key.  *flipped = canonical form is rc(text orientation).
template <int W> AC_HD Key<W> xk_canonical(const XKmer<W>& x, int k, bool* flipped) {
    Key<W> key;
    int ld;
    if (x.ld > 0) {  // "..ACG" < its rc "CGT.."
        key = x.fwd; ld = x.ld; *flipped = false;
    } else if (x.td > 0) {
        key = key_rc<W>(x.fwd, k);
        Key<W> m = key_inner_mask<W>(k, x.td, 0);  // the td former-dot bases became 'T' under rc: clear them
AC_UNROLL_W
        for (int i = 0; i < W; i++) key.w[i] &= m.w[i];
        ld = x.td; *flipped = true;
    } else {
        Key<W> r = key_rc<W>(x.fwd, k);
        if (key_lt<W>(r, x.fwd)) { key = r; *flipped = true; } else { key = x.fwd; *flipped = false; }
        ld = 0;
    }
    key.w[0] |= (u64)(255 - ld) << 56;
    return key;
}
template <int W> AC_HD int key_ld(const Key<W>& key) { return 255 - (int)(key.w[0] >> 56); }

// rc of an extended k-mer (text orientation of the opposite strand).
template <int W> AC_HD XKmer<W> xk_rc(const XKmer<W>& x, int k) {
    XKmer<W> r;
    r.fwd = key_rc<W>(x.fwd, k);
    r.ld = x.td; r.td = x.ld;
    Key<W> m = key_inner_mask<W>(k, r.ld, r.td);
AC_UNROLL_W
    for (int i = 0; i < W; i++) r.fwd.w[i] &= m.w[i];
    return r;
}

// Successor candidate: drop the first symbol, append symbol c (0..3 = ACGT, 4 = '.').
// Returns false when the result cannot be a k-mer of any padded sequence (dots not at an end).
template <int W> AC_HD bool xk_next(const XKmer<W>& x, int k, int c, XKmer<W>* out) {
    XKmer<W> y;
    Key<W> km = key_kmask<W>(k);
    y.fwd = x.fwd;
    key_roll_fwd<W>(y.fwd, c < 4 ? (u32)c : 0u, km);
    y.ld = x.ld > 0 ? x.ld - 1 : 0;
    if (c == 4) {
        y.td = x.td + 1;
        if (y.ld > 0) return false;       // dots at both ends
        if (y.td >= k) return false;
    } else {
        if (x.td > 0) return false;       // a base after a trailing dot
        y.td = 0;
    }
    *out = y;
    return true;
}

// 64-bit mix of a key.
template <int W> AC_HD u64 key_hash(const Key<W>& key) {
    u64 h = 0x9E3779B97F4A7C15ULL;
AC_UNROLL_W
    for (int i = 0; i < W; i++) {
        h ^= key.w[i];
        h *= 0xff51afd7ed558ccdULL;
        h ^= h >> 32;
    }
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 29;
    return h;
}

// Home slot of a k-mer in the table: a hash of its canonical MIDDLE (the k-2 bases without the first and the last one), not of
// the whole key.  The four successors X[1..k)+c of a k-mer X share the middle X[2..k), its four predecessors the middle
// X[0..k-2): all the neighbours a degree / link computation asks about (kmer_graph.rs:136-166, three of four of them absent) sit
// in ONE probe cluster, i.e. one cache line of the table instead of one random line each — the degree kernel was bound by
// exactly those line fetches.  At most 16 k-mers share a middle, members of a cluster are told apart by the fingerprint of the
// whole key and verified against the text as before, so only the placement changes, never a result.  Dot k-mers (sequence ends)
// and k < 3 hash the whole key.  `full_hash` = key_hash(key).
template <int W> AC_HD u64 key_home(const Key<W>& key, int k, bool isdot, u64 full_hash) {
    if (isdot || k < 3) return full_hash;
    Key<W> v = key;
    v.w[0] &= ~((u64)255 << 56);
    const Key<W> mm = key_kmask<W>(k - 2);
    Key<W> m = key_shr<W>(v, 2);
AC_UNROLL_W
    for (int i = 0; i < W; i++) m.w[i] &= mm.w[i];
    Key<W> r = key_rc<W>(m, k - 2);
    const Key<W> c = key_lt<W>(r, m) ? r : m;
    return key_hash<W>(c) ^ 0x5851F42D4C957F2DULL;
}
// The 2-bit code at bit position `bit` (even) of a key.
template <int W> AC_HD u32 key_code_at(const Key<W>& key, int bit) {
    const int wi = W - 1 - (bit >> 6);
    u64 x = 0;
AC_UNROLL_W
    for (int i = 0; i < W; i++) x = (i == wi) ? key.w[i] : x;   // constant indices only (see key_shr)
    return (u32)(x >> (bit & 63)) & 3u;
}
// Home hash AND slot tag of a k-mer.  The 23-bit tag of a REAL k-mer (k >= 3) is [mfp:13][x:2][y:2][h:6]: mfp = 13 bits of the home
// hash that do not take part in the slot index (equal for all the k-mers of one middle), (x, y) = the k-mer's first and last base
// read in the orientation in which its MIDDLE is canonical, h = 6 bits of the whole key's hash.  Two k-mers of one middle differ in
// x or y, so the tag still tells the members of a group apart; and two slots of one probe cluster whose mfp agree are (up to a
// 2^-13 coincidence) k-mers of one middle — which lets a sequential scan of the finished table decide, slot by slot, whether a
// k-mer has a "sibling" with the same x or the same y, i.e. whether the k-mer before / after it in the text can branch at all
// (SiblingFunctor, DegreeFunctor).  Dot k-mers and k < 3: 23 bits of the whole key's hash.  The tag is returned in place (bits 63..41).
struct KeyPlace { u64 home, tag; };
static const int TAG_X_SHIFT = 41 + 8, TAG_Y_SHIFT = 41 + 6, TAG_MFP_SHIFT = 41 + 10;
AC_HD u64 tag_compose(u64 home_hash, u32 x, u32 y, u64 full_hash) {
    return (((home_hash >> 51) << 10) | ((u64)x << 8) | ((u64)y << 6) | (full_hash >> 58)) << 41;
}
template <int W> AC_HD KeyPlace key_place(const Key<W>& key, int k, bool isdot, u64 full_hash) {
    KeyPlace pl;
    if (isdot || k < 3) { pl.home = full_hash; pl.tag = (full_hash >> 41) << 41; return pl; }
    Key<W> v = key;
    v.w[0] &= ~((u64)255 << 56);
    const Key<W> mm = key_kmask<W>(k - 2);
    Key<W> m = key_shr<W>(v, 2);
AC_UNROLL_W
    for (int i = 0; i < W; i++) m.w[i] &= mm.w[i];
    Key<W> r = key_rc<W>(m, k - 2);
    const bool mflip = key_lt<W>(r, m);
    const Key<W> c = mflip ? r : m;
    pl.home = key_hash<W>(c) ^ 0x5851F42D4C957F2DULL;
    const u32 first = key_code_at<W>(v, 2 * (k - 1)), last = (u32)v.w[W - 1] & 3u;
    pl.tag = tag_compose(pl.home, mflip ? 3u - last : first, mflip ? 3u - first : last, full_hash);
    return pl;
}

// ---- hash-table slot word ------------------------------------------------------------------------
// [tag:23][isdot:1][pos:40] (tag: key_place); EMPTY = all ones.  The slot stores the *text position* of the smallest
// occurrence of its canonical k-mer (the reference stores a raw pointer into the sequence,
// kmer_graph.rs:26-33); equal keys share fp and isdot, so a 64-bit atomicMin orders by position.
static const u64 SLOT_EMPTY = ~0ULL;
static const u64 POS_MASK = (1ULL << 40) - 1;
AC_HD u64 slot_make(u64 tag_in_place, bool isdot, u64 pos) { return tag_in_place | ((u64)(isdot ? 1 : 0) << 40) | pos; }
AC_HD u64 slot_pos(u64 v) { return v & POS_MASK; }
AC_HD bool slot_isdot(u64 v) { return (v >> 40) & 1; }
AC_HD bool slot_tag_eq(u64 a, u64 b) { return (a >> 40) == (b >> 40); }

}  // namespace ac
