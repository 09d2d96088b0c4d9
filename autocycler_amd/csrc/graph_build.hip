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

#include "device_rt.hpp"

namespace ac {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static const int CH = 64;          // text positions per thread in the streaming kernels

static const int MAX_PROBES = 1 << 14;

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

struct FindResult { u64 slot; int claimant_flipped; bool found; };

template <int W> AC_HD FindResult table_find(const TextCtx& t, const Table& tb, const Key<W>& ukey, bool isdot) {
    u64 h = key_hash<W>(ukey);
    u64 tag = slot_make(h, isdot, 0);
    u64 s = h & tb.cap_mask;
    FindResult r; r.found = false; r.slot = 0; r.claimant_flipped = 0;
    for (int probes = 0; probes < MAX_PROBES; probes++) {
        u64 v = tb.slots[s];
        if (v == SLOT_EMPTY) return r;
        if (slot_tag_eq(v, tag)) {
            int m = claimant_match<W>(t, v, ukey);
            if (m) { r.found = true; r.slot = s; r.claimant_flipped = (m == 2); return r; }
        }
        s = (s + 1) & tb.cap_mask;
    }
    return r;
}

// Lookup of an extended k-mer in text orientation.  rel_same: the query reads the same way as the
// stored smallest occurrence does in the text.
template <int W> AC_HD bool find_xk(const TextCtx& t, const Table& tb, const XKmer<W>& x, u64* slot, bool* rel_same) {
    bool flipped;
    Key<W> uk = xk_canonical<W>(x, t.k, &flipped);
    FindResult r = table_find<W>(t, tb, uk, x.ld > 0 || x.td > 0);
    if (!r.found) return false;
    *slot = r.slot;
    *rel_same = ((r.claimant_flipped != 0) == flipped);
    return true;
}

// Insert with "smallest text position wins" semantics.  Stale (cached) reads of a slot can only show
// an older state of a monotone word (EMPTY -> pos -> smaller pos of the same key), so every decision
// taken on them stays valid; claiming is decided by the CAS alone.
template <int W> AC_D void table_insert(const TextCtx& t, const Table& tb, const Key<W>& ukey, bool isdot, u64 p,
                                        u32* n_claimed, u32* err) {
    u64 h = key_hash<W>(ukey);
    u64 mine = slot_make(h, isdot, p);
    u64 s = h & tb.cap_mask;
    for (int probes = 0; probes < MAX_PROBES; probes++) {
        u64 v = tb.slots[s];
        if (v == SLOT_EMPTY) {
            u64 old = atomic_cas64(&tb.slots[s], SLOT_EMPTY, mine);
            if (old == SLOT_EMPTY) { atomic_add32(n_claimed, 1u); return; }
            v = old;
        }
        if (slot_tag_eq(v, mine)) {
            if (slot_pos(v) == p) return;
            if (claimant_match<W>(t, v, ukey)) {
                if (slot_pos(v) > p) atomic_min64(&tb.slots[s], mine);
                return;
            }
        }
        s = (s + 1) & tb.cap_mask;
    }
    atomic_or32(err, 1u);
}

// ---- K1: ASCII text -> 2-bit words + mask ------------------------------------------------------------
struct PackFunctor {
    const u8* text; u64 n_text; u64* bits; u64* mask;
    AC_HD void operator()(u64 tid) const {
        u64 base = tid * 64;
        u64 w0 = 0, w1 = 0, m = 0;
        for (int i = 0; i < 64; i++) {
            u64 p = base + (u64)i;
            u32 c = 0, bad = 1;
            if (p < n_text) {
                u32 ch = text[p];
                bad = !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
                c = bad ? 0u : (((ch >> 1) ^ (ch >> 2)) & 3u);
            }
            if (i < 32) w0 |= (u64)c << (62 - 2 * i); else w1 |= (u64)c << (62 - 2 * (i - 32));
            m |= (u64)bad << i;
        }
        bits[2 * tid] = w0; bits[2 * tid + 1] = w1; mask[tid] = m;
    }
};

// Shared streaming skeleton: thread handles CH consecutive text positions with rolling fwd/rc words.
// Visitor is called for every *valid* k-mer start p with either a real k-mer (isdot=false: canonical
// key + flipped) or a dot k-mer (isdot=true).
template <int W, class Visitor> AC_D void stream_chunk(const TextCtx& t, u64 tid, Visitor& vis) {
    const int k = t.k;
    u64 p0 = tid * (u64)CH;
    if (t.n_text < (u64)k || p0 > t.n_text - (u64)k) return;
    u64 p1 = p0 + (u64)CH;
    if (p1 > t.n_text - (u64)k + 1) p1 = t.n_text - (u64)k + 1;
    Key<W> km = key_kmask<W>(k);
    Key<W> fwd = text_extract<W>(t.bits, p0, k);
    Key<W> rc = key_rc<W>(fwd, k);
    int bad = 0;
    if (text_mask_count(t.mask, p0, k) > 0) {
        for (int j = k - 1; j >= 0; j--) if (text_mask(t.mask, p0 + (u64)j)) { bad = j + 1; break; }
    }
    for (u64 p = p0; p < p1; p++) {
        if (p != p0) {
            u64 e = p + (u64)k - 1;
            u32 c = text_code(t.bits, e);
            u32 m = text_mask(t.mask, e);
            key_roll_fwd<W>(fwd, c, km);
            key_roll_rc<W>(rc, c, k);
            bad = m ? k : (bad > 0 ? bad - 1 : 0);
        }
        if (bad == 0) {
            bool flipped = key_lt<W>(rc, fwd);
            Key<W> uk = flipped ? rc : fwd;
            uk.w[0] |= (u64)255 << 56;
            vis.kmer(p, uk, false, flipped);
        } else {
            XKmer<W> x;
            if (!xkmer_at<W>(t, p, &x)) continue;   // window crosses a separator
            bool flipped;
            Key<W> uk = xk_canonical<W>(x, k, &flipped);
            vis.kmer(p, uk, true, flipped);
        }
    }
}

// ---- K2: insert every k-mer occurrence (kmer_graph.rs:103-133) ---------------------------------------
template <int W> struct InsertVisitor {
    TextCtx t; Table tb; u32* n_claimed; u32* err;
    AC_D void kmer(u64 p, const Key<W>& uk, bool isdot, bool) { table_insert<W>(t, tb, uk, isdot, p, n_claimed, err); }
};
template <int W> struct InsertFunctor {
    TextCtx t; Table tb; u32* n_claimed; u32* err;
    AC_D void operator()(u64 tid) const {
        InsertVisitor<W> v{t, tb, n_claimed, err};
        stream_chunk<W>(t, tid, v);
    }
};

// ---- K3: table scan -> (novel position, slot) ----------------------------------------------------------
struct CollectFunctor {
    const u64* slots; u64* out_pos; u32* out_slot; u32* counter;
    AC_D void operator()(u64 s) const {
        u64 v = slots[s];
        if (v == SLOT_EMPTY) return;
        u32 i = atomic_add32(counter, 1u);
        out_pos[i] = slot_pos(v);
        out_slot[i] = (u32)s;
    }
};
struct Slot2NFunctor {
    const u32* nslot; u32* slot2n;
    AC_HD void operator()(u64 i) const { slot2n[nslot[i]] = (u32)i; }
};

// ---- K5: out/in degrees per distinct k-mer (kmer_graph.rs:136-166) -------------------------------------
template <int W> AC_D int count_successors(const TextCtx& t, const Table& tb, const XKmer<W>& x, int max_c) {
    int n = 0;
    for (int c = 0; c < max_c; c++) {
        XKmer<W> y;
        if (!xk_next<W>(x, t.k, c, &y)) continue;
        u64 slot; bool rel;
        if (find_xk<W>(t, tb, y, &slot, &rel)) n++;
    }
    return n;
}
template <int W> struct DegreeFunctor {
    TextCtx t; Table tb; const u64* npos; u32* kinfo; int any_dots;
    AC_D void operator()(u64 i) const {
        XKmer<W> x;
        u64 p = npos[i];
        if (text_mask_count(t.mask, p, t.k) == 0) { x.fwd = text_extract<W>(t.bits, p, t.k); x.ld = 0; x.td = 0; }
        else if (!xkmer_at<W>(t, p, &x)) return;
        int max_c = any_dots ? 5 : 4;
        int out = count_successors<W>(t, tb, x, max_c);
        XKmer<W> r = xk_rc<W>(x, t.k);
        int in = count_successors<W>(t, tb, r, max_c);
        kinfo[i] = (u32)out | ((u32)in << KI_IN_SHIFT);
    }
};

// ---- K6: first_position flags (kmer_graph.rs:57-60): first forward k-mer of each sequence and the RC of
// its last forward k-mer sit at pos 0 of a strand.
template <int W> struct FirstFunctor {
    TextCtx t; Table tb; const u32* slot2n; u32* kinfo;
    AC_D void operator()(u64 s) const {
        for (int which = 0; which < 2; which++) {
            u64 p = t.seq_off[s] + (which ? (u64)t.seq_len[s] - 1 : 0);
            XKmer<W> x;
            if (!xkmer_at<W>(t, p, &x)) continue;
            u64 slot; bool rel_same;
            if (!find_xk<W>(t, tb, x, &slot, &rel_same)) continue;
            u32 j = slot2n[slot];
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
template <int W> struct CKeyFunctor {
    TextCtx t; const u64* npos; const u32* scan; MinVal<W>* vals; u32* seg;
    AC_D void operator()(u64 i) const {
        XKmer<W> x;
        u64 p = npos[i];
        bool ok = true;
        if (text_mask_count(t.mask, p, t.k) == 0) { x.fwd = text_extract<W>(t.bits, p, t.k); x.ld = 0; x.td = 0; }
        else ok = xkmer_at<W>(t, p, &x);
        MinVal<W> v;
        bool flipped = false;
        if (ok) v.key = xk_canonical<W>(x, t.k, &flipped);
        else { for (int j = 0; j < W; j++) v.key.w[j] = ~0ULL; }
        v.flipped = flipped ? 1u : 0u; v.pad = 0;
        vals[i] = v;
        seg[i] = scan[i] - 1;
    }
};
struct IotaFunctor { u32* a; AC_HD void operator()(u64 i) const { a[i] = (u32)i; } };

// ---- K9: per-unitig metadata in seed (rank) order ---------------------------------------------------------
template <int W> struct UnitigMetaFunctor {
    const u32* order; const u32* ustart; const u64* npos; const MinVal<W>* sorted_min; u32 n_unitigs; u64 n_novel;
    u32* rank; u32* ulen; u64* ulen64; u64* ustartpos; u8* uorient;
    AC_HD void operator()(u64 r) const {
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

// ---- K10: paths, depth and min positions -----------------------------------------------------------------
template <int W> struct PathVisitor {
    TextCtx t; Table tb; const u32* slot2n; const u32* head; const u32* scan; const u32* rank; const u8* uorient;
    const u32* ulen; u64 n_novel;
    u64* ent_pos; int32_t* ent_val; u64 ent_cap; u32* ent_count; u32* depth; u32* minpos_fwd; u32* minpos_rev;
    AC_D void kmer(u64 p, const Key<W>& uk, bool isdot, bool flipped) {
        FindResult fr = table_find<W>(t, tb, uk, isdot);
        if (!fr.found) return;
        bool rel_same = ((fr.claimant_flipped != 0) == flipped);
        u32 j = slot2n[fr.slot];
        bool is_head = head[j] != 0;
        bool is_tail = (j + 1 == n_novel) || head[j + 1] != 0;
        if (!((rel_same && is_head) || (!rel_same && is_tail))) return;
        u32 r = rank[scan[j] - 1];
        bool strand = rel_same ? (uorient[r] != 0) : (uorient[r] == 0);
        u32 i = atomic_add32(ent_count, 1u);
        if ((u64)i < ent_cap) { ent_pos[i] = p; ent_val[i] = strand ? (int32_t)(r + 1) : -(int32_t)(r + 1); }
        atomic_add32(&depth[r], 1u);
        u32 s, f;
        if (locate(t, p, &s, &f)) {
            u32 other = t.seq_len[s] - ulen[r] - f;   // position of the same occurrence on the opposite strand
            if (strand) { atomic_min32(&minpos_fwd[r], f); atomic_min32(&minpos_rev[r], other); }
            else { atomic_min32(&minpos_rev[r], f); atomic_min32(&minpos_fwd[r], other); }
        }
    }
};
template <int W> struct PathFunctor {
    PathVisitor<W> v;
    AC_D void operator()(u64 tid) const {
        PathVisitor<W> vis = v;
        stream_chunk<W>(v.t, tid, vis);
    }
};
struct PathOffFunctor {   // first entry index of each sequence in the position-sorted entry list
    const u64* ent_pos; u64 n_ent; const u64* seq_off; u32 n_seqs; u64 n_text; u64* path_off;
    AC_HD void operator()(u64 s) const {
        u64 target = (s < n_seqs) ? seq_off[s] : n_text;
        u64 lo = 0, hi = n_ent;
        while (lo < hi) { u64 mid = (lo + hi) >> 1; if (ent_pos[mid] < target) lo = mid + 1; else hi = mid; }
        path_off[s] = lo;
    }
};

// ---- K11: links (unitig_graph.rs:234-287: a's (k-1)-suffix == b's (k-1)-prefix  <=>  b's first k-mer is
// a successor of a's last k-mer) --------------------------------------------------------------------------
template <int W> struct LinksFunctor {
    TextCtx t; Table tb; const u32* slot2n; const u32* head; const u32* scan; const u32* rank; const u8* uorient;
    const u32* order; const u32* ustart; const u64* npos; u32 n_unitigs; u64 n_novel; int any_dots;
    u8* link_cnt; int32_t* links; u32* err;
    AC_D void operator()(u64 idx) const {
        u32 r = (u32)(idx >> 1);
        int side = (int)(idx & 1);           // 0: forward strand's end, 1: reverse strand's end
        u32 u = order[r];
        u32 ia = ustart[u];
        u32 ib = ((u + 1 < n_unitigs) ? ustart[u + 1] : (u32)n_novel) - 1;
        bool o = uorient[r] != 0;
        // forward strand's last k-mer: o ? B : rc(A);  reverse strand's last k-mer: o ? rc(A) : B
        bool use_b = (side == 0) ? o : !o;
        XKmer<W> e;
        if (!xkmer_at<W>(t, npos[use_b ? ib : ia], &e)) { atomic_or32(err, 2u); return; }
        if (!use_b) e = xk_rc<W>(e, t.k);
        int n = 0;
        int max_c = any_dots ? 5 : 4;
        for (int c = 0; c < max_c; c++) {
            XKmer<W> y;
            if (!xk_next<W>(e, t.k, c, &y)) continue;
            u64 slot; bool rel_same;
            if (!find_xk<W>(t, tb, y, &slot, &rel_same)) continue;
            u32 j = slot2n[slot];
            u32 rv = rank[scan[j] - 1];
            bool strand = rel_same ? (uorient[rv] != 0) : (uorient[rv] == 0);
            bool is_head = head[j] != 0;
            bool is_tail = (j + 1 == n_novel) || head[j + 1] != 0;
            if (!(rel_same ? is_head : is_tail)) atomic_or32(err, 4u);   // successor of an end must start a unitig strand
            links[idx * 5 + n] = strand ? (int32_t)(rv + 1) : -(int32_t)(rv + 1);
            n++;
        }
        link_cnt[idx] = (u8)n;
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

// =============================================================================================================
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

int max_supported_k() { return (64 * 4 - 8) / 2; }   // W <= 4 key words in this build

struct GraphBuilder::Impl {
    u32 k = 0;
    u64 n_text = 0, n_bases = 0;
    u32 n_seqs = 0;
    int any_dots = 0;
    DBuf<u8> text_owned;
    const u8* d_text = nullptr;
    DBuf<u64> seq_off; DBuf<u32> seq_len; DBuf<u16> seq_d1, seq_d2;
    std::vector<u64> h_off; std::vector<u32> h_len;
    void set_table(const std::vector<uint64_t>& off, const std::vector<uint32_t>& len, const std::vector<uint16_t>& d1,
                   const std::vector<uint16_t>& d2) {
        n_seqs = (u32)off.size();
        h_off = off; h_len = len;
        seq_off.alloc(n_seqs); seq_len.alloc(n_seqs); seq_d1.alloc(n_seqs); seq_d2.alloc(n_seqs);
        copy_h2d(seq_off.ptr(), off.data(), n_seqs * 8);
        copy_h2d(seq_len.ptr(), len.data(), n_seqs * 4);
        copy_h2d(seq_d1.ptr(), d1.data(), n_seqs * 2);
        copy_h2d(seq_d2.ptr(), d2.data(), n_seqs * 2);
        n_bases = 0; any_dots = 0;
        for (u32 i = 0; i < n_seqs; i++) { n_bases += len[i]; if (d1[i] || d2[i]) any_dots = 1; }
        stream_sync();
    }
    template <int W> void build_impl(u32 assembly_count_hint, RawGraph* out, BuildTimings* tm);
};

GraphBuilder::GraphBuilder(uint32_t k) : impl_(new Impl) {
    impl_->k = k;
    if (k < 1 || (k % 2) == 0) throw DeviceError("k must be odd");
    if ((int)k > max_supported_k())
        throw DeviceError("k-mer sizes above " + std::to_string(max_supported_k()) + " are not supported by this build of the HIP backend");
}
GraphBuilder::~GraphBuilder() { delete impl_; }
uint64_t GraphBuilder::n_text() const { return impl_->n_text; }
uint64_t GraphBuilder::n_bases() const { return impl_->n_bases; }

void GraphBuilder::set_sequences_host(const std::vector<SeqView>& seqs) {
    double t0 = now_s();
    std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint16_t> d1, d2;
    std::vector<uint8_t> text = layout_text(seqs, impl_->k, &off, &len, &d1, &d2);
    impl_->n_text = text.size();
    impl_->text_owned.alloc(text.size());
    copy_h2d(impl_->text_owned.ptr(), text.data(), text.size());
    impl_->d_text = impl_->text_owned.ptr();
    impl_->set_table(off, len, d1, d2);
    tm_.h2d = now_s() - t0;
}
void GraphBuilder::set_text_device(const uint8_t* d_text, uint64_t n_text, const std::vector<uint64_t>& off,
                                   const std::vector<uint32_t>& len, const std::vector<uint16_t>& d1,
                                   const std::vector<uint16_t>& d2) {
    impl_->d_text = d_text;
    impl_->n_text = n_text;
    impl_->set_table(off, len, d1, d2);
}

static u64 next_pow2(u64 x) { u64 p = 1; while (p < x) p <<= 1; return p; }

template <int W>
void GraphBuilder::Impl::build_impl(u32 assembly_count_hint, RawGraph* out, BuildTimings* tm) {
    double t_begin = now_s(), t0 = t_begin;
    auto lap = [&](double* acc) { stream_sync(); double t = now_s(); *acc += t - t0; t0 = t; };
    if (n_text >= POS_MASK) throw DeviceError("input too large for 40-bit text positions");
    if (n_seqs == 0) throw DeviceError("no sequences");

    // K1 pack
    u64 n_bits_words = n_text / 32 + W + 4, n_mask_words = n_text / 64 + 4;
    DBuf<u64> bits(n_bits_words, true), mask(n_mask_words);
    mask.fill_bytes(0xFF);
    launch((n_text + 63) / 64, PackFunctor{d_text, n_text, bits.ptr(), mask.ptr()});
    lap(&tm->pack);

    TextCtx t{bits.ptr(), mask.ptr(), n_text, (int)k, seq_off.ptr(), seq_len.ptr(), seq_d1.ptr(), seq_d2.ptr(), n_seqs};
    u64 n_chunks = (n_text + CH - 1) / CH;

    // K2 insert.  Capacity from the reference's own capacity hint (assembly_count, kmer_graph.rs:40):
    // similar assemblies share most k-mers.  Overflow -> retry with a larger table.
    u64 est = n_bases / (assembly_count_hint ? assembly_count_hint : 1);
    u64 cap = next_pow2(std::max<u64>(1024, est * 3 + 4096));
    if (cap > next_pow2(n_bases * 2 + 1024)) cap = next_pow2(n_bases * 2 + 1024);
    DBuf<u64> slots;
    DBuf<u32> counters(8, true);   // [0] claimed, [1] err, [2] path entries, [3] link err
    u32 n_distinct = 0;
    for (;;) {
        slots.alloc(cap);
        slots.fill_bytes(0xFF);
        counters.fill_bytes(0);
        Table tb{slots.ptr(), cap - 1};
        stream_sync();
#ifndef AC_EMU
        hipEvent_t e0, e1;
        AC_HIP_CHECK(hipEventCreate(&e0)); AC_HIP_CHECK(hipEventCreate(&e1));
        AC_HIP_CHECK(hipEventRecord(e0, 0));
#endif
        launch(n_chunks, InsertFunctor<W>{t, tb, counters.ptr(), counters.ptr() + 1});
#ifndef AC_EMU
        AC_HIP_CHECK(hipEventRecord(e1, 0));
        AC_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0; AC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        tm->insert_kernel_ms = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#endif
        std::vector<u32> c = to_host(counters, 2);
        n_distinct = c[0];
        bool overflow = (c[1] != 0) || ((u64)n_distinct * 10 > cap * 7);
        if (!overflow) break;
        if (cap >= next_pow2(n_bases * 4 + 1024)) throw DeviceError("k-mer table overflow");
        cap *= 4;
    }
    tm->insert_positions = n_text;
    tm->table_capacity = cap;
    tm->n_distinct = n_distinct;
    Table tb{slots.ptr(), cap - 1};
    lap(&tm->insert);

    // K3 collect + sort by position -> novel list
    u64 N = n_distinct;
    DBuf<u64> npos(N); DBuf<u32> nslot(N);
    counters.fill_bytes(0);
    launch(cap, CollectFunctor{slots.ptr(), npos.ptr(), nslot.ptr(), counters.ptr()});
    sort_pairs_u64_u32(npos, nslot, N, 40);
    DBuf<u32> slot2n(cap);
    launch(N, Slot2NFunctor{nslot.ptr(), slot2n.ptr()});
    lap(&tm->collect_sort);

    // K5/K6 degrees + first flags
    DBuf<u32> kinfo(N, true);
    launch(N, DegreeFunctor<W>{t, tb, npos.ptr(), kinfo.ptr(), any_dots});
    launch(n_seqs, FirstFunctor<W>{t, tb, slot2n.ptr(), kinfo.ptr()});
    lap(&tm->degree);

    // K7 heads -> unitig ids
    DBuf<u32> head(N + 1, true), scan(N + 1, true);
    launch(N, HeadFunctor{npos.ptr(), kinfo.ptr(), head.ptr(), N});
    inclusive_scan_u32(head.ptr(), scan.ptr(), N);
    u32 U = read_scalar(scan.ptr() + (N - 1));
    DBuf<u32> ustart(U + 1);
    launch(N, UnitigStartFunctor{head.ptr(), scan.ptr(), ustart.ptr(), N});
    lap(&tm->segment);

    // K8 min canonical k-mer per unitig
    DBuf<MinVal<W>> umin(U);
    {
        DBuf<MinVal<W>> vals(N); DBuf<u32> seg(N);
        launch(N, CKeyFunctor<W>{t, npos.ptr(), scan.ptr(), vals.ptr(), seg.ptr()});
        reduce_by_segment(seg.ptr(), vals.ptr(), N, umin.ptr(), U, MinOp<W>());
    }
    lap(&tm->minkey);

    // K9 seed order = rank of the smallest k-mer
    DBuf<u32> order(U);
    launch(U, IotaFunctor{order.ptr()});
    sort_by_key_cmp(umin, order, U, MinValLess<W>());
    DBuf<u32> rank(U), ulen(U); DBuf<u64> ulen64(U), ustartpos(U), useq_off(U + 1); DBuf<u8> uorient(U);
    launch(U, UnitigMetaFunctor<W>{order.ptr(), ustart.ptr(), npos.ptr(), umin.ptr(), U, N, rank.ptr(), ulen.ptr(),
                                   ulen64.ptr(), ustartpos.ptr(), uorient.ptr()});
    exclusive_scan_u64(ulen64.ptr(), useq_off.ptr(), U);
    lap(&tm->rank);

    // K10 paths
    DBuf<u32> depth(U, true), minpos_fwd(U), minpos_rev(U);
    minpos_fwd.fill_bytes(0xFF); minpos_rev.fill_bytes(0xFF);
    u64 ent_cap = std::min<u64>(n_bases, std::max<u64>(1u << 20, n_bases / 4));
    DBuf<u64> ent_pos; DBuf<int32_t> ent_val;
    u64 n_ent = 0;
    for (;;) {
        ent_pos.alloc(ent_cap); ent_val.alloc(ent_cap);
        counters.fill_bytes(0);
        depth.fill_bytes(0); minpos_fwd.fill_bytes(0xFF); minpos_rev.fill_bytes(0xFF);
        PathVisitor<W> pv{t, tb, slot2n.ptr(), head.ptr(), scan.ptr(), rank.ptr(), uorient.ptr(), ulen.ptr(), N,
                          ent_pos.ptr(), ent_val.ptr(), ent_cap, counters.ptr() + 2, depth.ptr(), minpos_fwd.ptr(), minpos_rev.ptr()};
        launch(n_chunks, PathFunctor<W>{pv});
        n_ent = read_scalar(counters.ptr() + 2);
        if (n_ent <= ent_cap) break;
        ent_cap = n_ent;
    }
    sort_pairs_u64_i32(ent_pos, ent_val, n_ent, 40);
    DBuf<u64> path_off(n_seqs + 1);
    launch(n_seqs + 1, PathOffFunctor{ent_pos.ptr(), n_ent, seq_off.ptr(), n_seqs, n_text, path_off.ptr()});
    tm->n_path_entries = n_ent;
    lap(&tm->paths);

    // K11 links
    DBuf<u8> link_cnt((u64)U * 2, true); DBuf<int32_t> links((u64)U * 10, true);
    launch((u64)U * 2, LinksFunctor<W>{t, tb, slot2n.ptr(), head.ptr(), scan.ptr(), rank.ptr(), uorient.ptr(), order.ptr(),
                                      ustart.ptr(), npos.ptr(), U, N, any_dots, link_cnt.ptr(), links.ptr(), counters.ptr() + 3});
    lap(&tm->links);

    // K12 sequences
    u64 total = N;   // sum of unitig lengths == number of distinct canonical k-mers
    DBuf<u8> useq(total);
    launch((total + 63) / 64, SeqFunctor{bits.ptr(), useq_off.ptr(), ustartpos.ptr(), ulen.ptr(), uorient.ptr(), U, total,
                                         (int)(k / 2), useq.ptr()});
    lap(&tm->seqs);

    // D2H
    out->k = k;
    out->n_kmers = 2 * (u64)N;
    out->n_unitigs = U;
    out->len = to_host(ulen, U);
    out->depth = to_host(depth, U);
    out->minpos_fwd = to_host(minpos_fwd, U);
    out->minpos_rev = to_host(minpos_rev, U);
    out->seq_off = to_host(useq_off, U);
    out->seq_off.push_back(total);
    out->seqs.resize(total);
    copy_d2h(&out->seqs[0], useq.ptr(), total);
    out->link_cnt = to_host(link_cnt, (u64)U * 2);
    out->links = to_host(links, (u64)U * 10);
    out->path_off = to_host(path_off, n_seqs + 1);
    out->path = to_host(ent_val, n_ent);
    u32 lerr = read_scalar(counters.ptr() + 3);
    if (lerr) throw DeviceError("internal error: inconsistent unitig ends (code " + std::to_string(lerr) + ")");
    lap(&tm->d2h);
    tm->total_device = now_s() - t_begin;
}

void GraphBuilder::build(uint32_t assembly_count_hint, RawGraph* out) {
    BuildTimings keep = tm_;
    tm_ = BuildTimings();
    tm_.h2d = keep.h2d;
    int W = words_for_k((int)impl_->k);
    switch (W) {
        case 1: impl_->build_impl<1>(assembly_count_hint, out, &tm_); break;
        case 2: impl_->build_impl<2>(assembly_count_hint, out, &tm_); break;
        case 3: impl_->build_impl<3>(assembly_count_hint, out, &tm_); break;
        case 4: impl_->build_impl<4>(assembly_count_hint, out, &tm_); break;
        default: throw DeviceError("unsupported k");
    }
}

}  // namespace ac
