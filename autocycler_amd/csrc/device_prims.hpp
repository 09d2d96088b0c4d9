// The plain device primitives of the pipeline, hand-written for gfx950 (round 5: rocPRIM is gone from the library — its radix sort,
// merge sort, scans and reduce-by-key were ~8 % of config C's GPU time, dozens of its launches, and 850 of the code object's 880 kernels):
//
//   scan            ONE launch: a chained scan with decoupled look-back.  A workgroup takes its tile by ticket (so a tile's predecessors have
//                   all started), publishes its aggregate, and wavefront 0 looks back over the predecessors' words 64 at a time — a tile's
//                   state is ONE 64-bit word [flag:2][value:62] written and read with device-scope atomics, so there is nothing to order.
//   radix sort      stable LSD over 8-bit digits, 1 + ceil(bits / 8) launches: one histogram kernel for all digits, then one "onesweep" pass
//                   per digit — a tile ranks its keys (a wavefront owns 512 consecutive keys and takes them 64 at a time: peers of a digit by
//                   eight ballots, running counts per (wavefront, digit) in LDS), thread d looks back for digit d's prefix over the earlier
//                   tiles, and every key goes straight to its place.  One state word per (tile, digit), its flag tagged with the pass so
//                   that the array is cleared once per sort.
//   comparator sort, segmented reduce    fallback paths only (knob variants, keys wider than four words): a merge sort by ranks (one binary
//                   search per element and pass) and a thread per segment.
// The same kernel source runs under the lockstep emulation (AC_EMU): its workgroups run one after the other in ticket order, so a look-back
// always finds its predecessors complete.
#pragma once

namespace ac {

// ---- one state word per tile ---------------------------------------------------------------------------------------------------
#ifdef AC_EMU
inline u64 state_load(const u64* p) { return *p; }
inline void state_store(u64* p, u64 v) { *p = v; }
#else
__device__ inline u64 state_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void state_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

AC_HD int prim_popc64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
struct alignas(16) PrimV16 { u32 a, b, c, d; };
enum ScanOp { SCAN_ADD = 0, SCAN_MAX = 1 };
template <int OP> AC_HD u64 scan_op(u64 a, u64 b) { return OP == SCAN_ADD ? a + b : (a > b ? a : b); }
AC_D u64 wave_shfl_up64(u64 v, int d) { const int l = wv::lane(); const u64 t = wv::shfl64(v, l >= d ? l - d : l); return t; }

static const u32 SCAN_ITEMS = 16, SCAN_TILE = 256 * SCAN_ITEMS;
// One state word per tile: [epoch:16][flag:2][value:46].  The words live in a POOL that outlives the builds and is never cleared between
// scans: every scan gets the next epoch, a word of another epoch reads as "not there yet".  (A scan used to pay a fill launch for its state
// words — as rocPRIM paid its init kernel.)  Tickets come from one ever-growing counter; the host knows where a scan's first ticket is.
static const u64 SC_VAL = (1ULL << 46) - 1;
AC_HD u64 sc_word(u64 epoch, u64 flag, u64 v) { return (epoch << 48) | (flag << 46) | (v & SC_VAL); }
struct ScanPool {
    u64* words = nullptr;      // [0] = ticket counter, [1 + t] = tile t
    u64 cap = 0, epoch = 0, tickets = 0;
    int dev = -1;
    ~ScanPool() { release(); }
    void release() {
#ifdef AC_EMU
        free(words);
#else
        if (words) (void)hipFree(words);
#endif
        words = nullptr; cap = 0;
    }
    // After anything failed between a take() and the end of its kernel (a launch that threw, a build that was abandoned): the device's ticket
    // counter and the host's mirror of it may no longer agree — the next take() starts from a cleared pool.
    void invalidate() { epoch = 1ULL << 16; }
    // the pool for a scan of `tiles` tiles: returns the epoch to tag with and the value the ticket counter starts this scan at
    void take(u64 tiles, u64* epoch_out, u64* ticket_base) {
#ifndef AC_EMU
        int d = 0;
        AC_HIP_CHECK(hipGetDevice(&d));
        if (d != dev) { release(); dev = d; }
#endif
        if (tiles + 1 > cap || epoch + 1 >= (1ULL << 16)) {
            const u64 want = std::max<u64>(tiles + 1, std::max<u64>(cap, 4096));
#ifdef AC_EMU
            free(words);
            words = (u64*)calloc(want, 8);
            if (!words) throw DeviceError("out of memory for the scan state pool");
#else
            flush_fills();
            AC_HIP_CHECK(hipStreamSynchronize(0));      // (a scan in flight still reads the old words)
            if (want != cap) { if (words) (void)hipFree(words); AC_HIP_CHECK(hipMalloc((void**)&words, want * 8)); }
            AC_HIP_CHECK(hipMemset(words, 0, want * 8));
#endif
            cap = want; epoch = 0; tickets = 0;
        }
        *epoch_out = ++epoch;
        *ticket_base = tickets;
        tickets += tiles;
    }
};
enum { CTX_SCANPOOL = 7 };
inline ScanPool& scan_pool() { return ctx_object<ScanPool>(CTX_SCANPOOL); }

// out[i] = op over in[0 .. i] (INCL) or in[0 .. i) (exclusive; identity 0).  Values (and their running totals) stay below 2^46 (the pipeline's are counts and byte offsets of a text of < 2^40 positions).
template <class T, int OP, bool INCL>
AC_KERNEL void __launch_bounds__(256) scan_kernel(const T* in, T* out, u64 n, u64* pool, u64 epoch, u64 ticket_base) {
    AC_SHARED u64 s_wave[4];
    AC_SHARED u64 s_prefix;
    AC_SHARED u64 s_tile;
    u64* state = pool + 1;
    const unsigned tid = wv::tid();
    const int lane = wv::lane(), wave = (int)(tid >> 6);
    if (tid == 0) s_tile = atomic_add64(&pool[0], 1) - ticket_base;
    wv::block_sync();
    const u64 tile = s_tile;
    const u64 base = tile * SCAN_TILE + (u64)tid * SCAN_ITEMS;
    alignas(16) T v[SCAN_ITEMS];
    if (base + SCAN_ITEMS <= n && (((uintptr_t)(in + base)) & 15u) == 0) {      // whole and aligned: 16-byte loads
        const PrimV16* p = (const PrimV16*)(in + base);
        PrimV16* q = (PrimV16*)v;
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS * sizeof(T) / 16; j++) q[j] = p[j];
    } else {
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; j++) v[j] = base + j < n ? in[base + j] : (T)0;
    }
    u64 tsum = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; j++) tsum = scan_op<OP>(tsum, (u64)v[j]);
    // inclusive scan of the threads' totals over the wavefront, the wavefronts' totals through LDS
    u64 incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u64 t = wave_shfl_up64(incl, o); if (lane >= o) incl = scan_op<OP>(incl, t); }
    if (lane == 63) s_wave[wave] = incl;
    wv::block_sync();
    u64 wave_off = 0, tile_agg = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) wave_off = scan_op<OP>(wave_off, s_wave[w]); tile_agg = scan_op<OP>(tile_agg, s_wave[w]); }
    if (wave == 0) {
        // publish the aggregate, look back 64 predecessors at a time until one of them has its inclusive prefix
        const u64 T_AGG = (epoch << 2) | 1, T_INCL = (epoch << 2) | 2;
        if (lane == 0) state_store(&state[tile], sc_word(epoch, tile == 0 ? 2 : 1, tile_agg));
        u64 excl = 0;
        if (tile != 0) {      // (wave-uniform)
            int64_t p0 = (int64_t)tile - 1;
            for (;;) {
                const int64_t p = p0 - lane;
                u64 st = p >= 0 ? state_load(&state[p]) : sc_word(epoch, 2, 0);
                while (wv::ballot(p >= 0 && (st >> 46) != T_AGG && (st >> 46) != T_INCL) != 0) { if (p >= 0 && (st >> 46) != T_AGG && (st >> 46) != T_INCL) st = state_load(&state[p]); }
                const u64 incl_mask = wv::ballot((st >> 46) == T_INCL);
                int first = 64;
                if (incl_mask) first = __builtin_ctzll(incl_mask);
                u64 c = lane <= first ? (st & SC_VAL) : 0;
#pragma unroll
                for (int o = 32; o; o >>= 1) c = scan_op<OP>(c, wv::shfl_xor64(c, o));
                excl = scan_op<OP>(excl, c);
                if (incl_mask) break;
                p0 -= 64;
            }
            if (lane == 0) state_store(&state[tile], sc_word(epoch, 2, scan_op<OP>(excl, tile_agg)));
        }
        if (lane == 0) s_prefix = excl;
    }
    wv::block_sync();
    u64 run = scan_op<OP>(s_prefix, wave_off);
    {   // exclusive prefix of this thread within its wavefront
        const u64 before = wave_shfl_up64(incl, 1);
        if (lane > 0) run = scan_op<OP>(run, before);
    }
    alignas(16) T o[SCAN_ITEMS];
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; j++) {
        const u64 nxt = scan_op<OP>(run, (u64)v[j]);
        o[j] = (T)(INCL ? nxt : run);
        run = nxt;
    }
    if (base + SCAN_ITEMS <= n && (((uintptr_t)(out + base)) & 15u) == 0) {
        PrimV16* q = (PrimV16*)(out + base);
        const PrimV16* p = (const PrimV16*)o;
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS * sizeof(T) / 16; j++) q[j] = p[j];
    } else {
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; j++) if (base + j < n) out[base + j] = o[j];
    }
}
template <class T, int OP, bool INCL> inline void scan_launch(const T* in, T* out, size_t n, stream_t s) {
    if (!n) return;
    if (s != 0) throw DeviceError("scan: stream 0 only (the state pool's tickets are stream-ordered)");
    const u64 tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    u64 epoch = 0, ticket_base = 0;
    ScanPool& pool = scan_pool();
    pool.take(tiles, &epoch, &ticket_base);
    launch_wave_kernel(scan_kernel<T, OP, INCL>, tiles, s, in, out, (u64)n, pool.words, epoch, ticket_base);
}
inline void inclusive_scan_u32(const u32* in, u32* out, size_t n, stream_t s = 0) { scan_launch<u32, SCAN_ADD, true>(in, out, n, s); }
inline void inclusive_max_scan_u32(const u32* in, u32* out, size_t n, stream_t s = 0) { scan_launch<u32, SCAN_MAX, true>(in, out, n, s); }
inline void exclusive_scan_u32(const u32* in, u32* out, size_t n, stream_t s = 0) { scan_launch<u32, SCAN_ADD, false>(in, out, n, s); }
inline void exclusive_scan_u64(const u64* in, u64* out, size_t n, stream_t s = 0) { scan_launch<u64, SCAN_ADD, false>(in, out, n, s); }

// ---- stable LSD radix sort of (u64 key, 32-bit value) pairs on key bits [0, end_bit) -------------------------------------------
static const u32 RS_ITEMS = 8, RS_TILE = 256 * RS_ITEMS, RS_MAX_PASSES = 8;
static const int RS_LB = 16;
// hist[p * 256 + d] = keys whose digit p is d
template <int UNUSED> AC_KERNEL void __launch_bounds__(256) radix_hist_kernel(const u64* keys, u64 n, int passes, int begin_bit, int end_bit, u32* hist) {
    AC_SHARED u32 s_hist[RS_MAX_PASSES * 256];
    const unsigned tid = wv::tid();
    for (int i = (int)tid; i < passes * 256; i += 256) s_hist[i] = 0;
    wv::block_sync();
    const u64 base = (u64)wv::bid() * RS_TILE;
    for (u32 j = 0; j < RS_ITEMS; j++) {
        const u64 i = base + (u64)j * 256 + tid;
        if (i >= n) break;
        u64 k = keys[i];
        if (end_bit < 64) k &= (1ULL << end_bit) - 1ULL;      // (the last digit may be narrower than eight bits)
        k >>= begin_bit;
        for (int p = 0; p < passes; p++) atomic_add32(&s_hist[p * 256 + (int)((k >> (8 * p)) & 255)], 1u);
    }
    wv::block_sync();
    for (int i = (int)tid; i < passes * 256; i += 256) if (s_hist[i]) atomic_add32(&hist[i], s_hist[i]);
}
// One pass.  state[(t * 256) + d]: [pass tag + flag:8][count:56] — the flag of pass p is 2 p + 1 (aggregate) / 2 p + 2 (inclusive prefix),
// anything below is a leftover of an earlier pass, i.e. "not there yet".  ticket[pass] hands out the tiles.
template <class V>
AC_KERNEL void __launch_bounds__(256) radix_pass_kernel(const u64* kin, const V* vin, u64* kout, V* vout, u64 n, int pass, int bits_here, int begin_bit,
                                                         const u32* hist, u64* state, u64* ticket) {
    AC_SHARED u32 s_cnt[4 * 256];
    AC_SHARED u32 s_base[4 * 256];
    AC_SHARED u32 s_scan[256];
    AC_SHARED u64 s_tile;
    AC_SHARED u32 s_trivial;
    const unsigned tid = wv::tid();
    const int lane = wv::lane(), wave = (int)(tid >> 6);
    if (tid == 0) { s_tile = ticket ? atomic_add64(&ticket[pass], 1) : (u64)wv::bid(); s_trivial = 0; }
    for (int i = (int)tid; i < 4 * 256; i += 256) s_cnt[i] = 0;
    wv::block_sync();
    // a digit that is the same in every key (the high bytes of a length, of a level): the pass is the identity — copy the tile, no ranking, no look-back
    if (hist[pass * 256 + (int)tid] == (u32)n) s_trivial = 1;
    wv::block_sync();
    const u64 tile = s_tile;
    if (s_trivial) {      // (workgroup-uniform)
        for (u32 j = 0; j < RS_ITEMS; j++) {
            const u64 i = tile * RS_TILE + (u64)j * 256 + tid;
            if (i < n) { kout[i] = kin[i]; vout[i] = vin[i]; }
        }
        return;
    }
    const u32 dmask = (1u << bits_here) - 1u;
    const int shift = begin_bit + 8 * pass;
    // ranking: wavefront w owns keys [tile base + 512 w, + 512), 64 at a time
    const u64 wbase = tile * RS_TILE + (u64)wave * (64 * RS_ITEMS);
    u64 key[RS_ITEMS]; V val[RS_ITEMS]; u32 rank[RS_ITEMS]; u32 dig[RS_ITEMS];
#pragma unroll
    for (u32 r = 0; r < RS_ITEMS; r++) {      // all the loads first: one memory latency for the tile, not one per round
        const u64 i = wbase + (u64)r * 64 + (u64)lane;
        const bool valid = i < n;
        key[r] = valid ? kin[i] : 0; val[r] = valid ? vin[i] : V();
    }
#pragma unroll
    for (u32 r = 0; r < RS_ITEMS; r++) {
        const u64 i = wbase + (u64)r * 64 + (u64)lane;
        const bool valid = i < n;
        const u32 d = (u32)(key[r] >> shift) & dmask;
        dig[r] = d;
        u64 peers = wv::ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) { const u64 bal = wv::ballot(valid && ((d >> b) & 1u)); peers &= ((d >> b) & 1u) ? bal : ~bal; }
        // (an invalid lane's peers are meaningless: it does not take part below)
        int leader = 0;
        u32 before = 0, count = 0;
        if (valid) {
            leader = __builtin_ctzll(peers);
            before = (u32)prim_popc64(peers & ((1ULL << lane) - 1ULL));
            count = (u32)prim_popc64(peers);
        }
        u32 old = 0;
        if (valid && lane == leader) { old = s_cnt[wave * 256 + (int)d]; s_cnt[wave * 256 + (int)d] = old + count; }
        old = (u32)wv::shfl((int)old, valid ? leader : lane);
        rank[r] = old + before;
    }
    wv::block_sync();
    {   // thread d: digit d of this tile — the wavefronts' offsets, the digit's global start, its prefix over the earlier tiles
        const int d = (int)tid;
        const u32 c0 = s_cnt[d], c1 = s_cnt[256 + d], c2 = s_cnt[512 + d], c3 = s_cnt[768 + d];
        const u32 total = c0 + c1 + c2 + c3;
        const u64 F_AGG = (u64)(2 * pass + 1) << 56, F_INCL = (u64)(2 * pass + 2) << 56, VAL = (1ULL << 56) - 1;
        state_store(&state[tile * 256 + (u64)d], (tile == 0 ? F_INCL : F_AGG) | (u64)total);
        // exclusive scan of the digit totals of the whole input (hist) over the 256 digits
        u32 h = hist[pass * 256 + d];
        u32 incl = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)wv::shfl_up((int)incl, o); if (lane >= o) incl += t; }
        if (lane == 63) s_scan[wave] = incl;
        wv::block_sync();
        u32 gbase = incl - h;
        for (int w = 0; w < wave; w++) gbase += s_scan[w];
        u64 excl = 0;
        if (tile != 0) {
            // Look-back, RS_LB predecessors per round trip: the tiles of a small sort all start together and publish their aggregates at
            // about the same time, so tile t would otherwise walk back t dependent loads (78 tiles: 26 us per pass against 8 us of work);
            // in a large sort the nearest predecessor already has its inclusive prefix and the first load ends the walk.
            int64_t p = (int64_t)tile - 1;
            u64 st0 = state_load(&state[(u64)p * 256 + (u64)d]);
            while ((st0 >> 56) < (u64)(2 * pass + 1)) st0 = state_load(&state[(u64)p * 256 + (u64)d]);
            excl = st0 & VAL;
            bool done = (st0 >> 56) == (u64)(2 * pass + 2);
            p--;
            while (!done) {
                u64 st[RS_LB];
#pragma unroll
                for (int j = 0; j < RS_LB; j++) st[j] = p - j >= 0 ? state_load(&state[(u64)(p - j) * 256 + (u64)d]) : F_INCL;
#pragma unroll
                for (int j = 0; j < RS_LB; j++) {
                    if (done) continue;
                    while ((st[j] >> 56) < (u64)(2 * pass + 1)) st[j] = state_load(&state[(u64)(p - j) * 256 + (u64)d]);
                    excl += st[j] & VAL;
                    if ((st[j] >> 56) == (u64)(2 * pass + 2)) done = true;
                }
                p -= RS_LB;
            }
            state_store(&state[tile * 256 + (u64)d], F_INCL | (excl + (u64)total));
        }
        const u32 g = gbase + (u32)excl;
        s_base[d] = g; s_base[256 + d] = g + c0; s_base[512 + d] = g + c0 + c1; s_base[768 + d] = g + c0 + c1 + c2;
    }
    wv::block_sync();
#pragma unroll
    for (u32 r = 0; r < RS_ITEMS; r++) {
        const u64 i = wbase + (u64)r * 64 + (u64)lane;
        if (i < n) { const u32 dst = s_base[wave * 256 + (int)dig[r]] + rank[r]; kout[dst] = key[r]; vout[dst] = val[r]; }
    }
}
// What a sort needs cleared before it starts (digit histograms, tile states, tile tickets).  A caller that knows its sorts ahead of time
// prepare()s their scratch before its next launch, so that the clears leave with the fills of that launch's batch instead of as a launch of
// their own right before every sort (round 6: a build ran four such launches).  One use per prepare().
struct RadixScratch {
    DBuf<u32> hist; DBuf<u64> state, ticket; size_t n = 0; int passes = 0; bool ready = false;
    void prepare(size_t n_items, int key_bits, stream_t s = 0) {
        n = n_items; passes = std::min<int>((key_bits + 7) / 8, (int)8); ready = n > 1 && passes > 0;
        if (!ready) return;
        hist.alloc((size_t)passes * 256); state.alloc(((n + 2047) / 2048) * 256); ticket.alloc(8);
        hist.fill_bytes(0, s); state.fill_bytes(0, s); ticket.fill_bytes(0, s);
    }
};
template <class V> inline void radix_sort_pairs_impl(DBuf<u64>& keys, DBuf<V>& vals, size_t n, int begin_bit, int end_bit, stream_t s, RadixScratch* pre = nullptr) {
    if (n <= 1 || end_bit <= begin_bit) return;
    if (n >= 0xFFFFFFF0ULL) throw DeviceError("radix sort: more than 2^32 items");
    const int passes = (end_bit - begin_bit + 7) / 8;
    const u64 tiles = (n + RS_TILE - 1) / RS_TILE;
    DBuf<u64> k2(n); DBuf<V> v2(n);
    RadixScratch own;
    if (!(pre && pre->ready && pre->n >= n && pre->passes >= passes)) { own.prepare(n, end_bit - begin_bit, s); pre = &own; }
    pre->ready = false;
    DBuf<u32>& hist = pre->hist; DBuf<u64>& state = pre->state; DBuf<u64>& ticket = pre->ticket;
    launch_wave_kernel(radix_hist_kernel<0>, tiles, s, (const u64*)keys.ptr(), (u64)n, passes, begin_bit, end_bit, hist.ptr());
    u64* ka = keys.ptr(); u64* kb = k2.ptr(); V* va = vals.ptr(); V* vb = v2.ptr();
    for (int p = 0; p < passes; p++) {
        const int bits_here = std::min(8, end_bit - begin_bit - 8 * p);
        launch_wave_kernel(radix_pass_kernel<V>, tiles, s, (const u64*)ka, (const V*)va, kb, vb, (u64)n, p, bits_here, begin_bit, (const u32*)hist.ptr(), state.ptr(),
                           ticket.ptr());      // tiles always by ticket: a look-back only ever waits for tiles that have STARTED (round 5 let grids of <= 1024
                                               // tiles use blockIdx, which needs the whole grid resident — not promised when ranks share a device: ADVICE r5)
        std::swap(ka, kb); std::swap(va, vb);
    }
    if (passes & 1) { keys = std::move(k2); vals = std::move(v2); }
}
// (bits below begin_bit and from end_bit up do not take part: the order among keys that agree on [begin_bit, end_bit) is the input's)
inline void sort_pairs_u64_u32(DBuf<u64>& keys, DBuf<u32>& vals, size_t n, int end_bit, stream_t s = 0, int begin_bit = 0, RadixScratch* pre = nullptr) { radix_sort_pairs_impl<u32>(keys, vals, n, begin_bit, end_bit, s, pre); }
inline void sort_pairs_u64_i32(DBuf<u64>& keys, DBuf<int32_t>& vals, size_t n, int end_bit, stream_t s = 0, int begin_bit = 0) { radix_sort_pairs_impl<int32_t>(keys, vals, n, begin_bit, end_bit, s); }

// ---- fallback paths ---------------------------------------------------------------------------------------------------------------
// Segmented reduction of `vals` over runs of equal consecutive `seg` ids (ids are 0, 1, 2, ... in order, so run r reduces into out[r]):
// a thread per element that STARTS a run walks its run.  Knob variants and keys wider than four words only.
struct CountCheckFunctor {    // sets an error bit instead of making the host wait for the segment count
    const u32* cnt; u32 expected; u32* err; u32 bit;
    AC_D void operator()(u64) const { if (*cnt != expected) atomic_or32(err, bit); }
};
template <class V, class Op> struct SegReduceFunctor {
    const u32* seg; const V* vals; u64 n; V* out; u64 n_segments; Op op; u32* cnt;
    AC_D void operator()(u64 i) const {
        if (i > 0 && seg[i - 1] == seg[i]) return;
        V acc = vals[i];
        for (u64 j = i + 1; j < n && seg[j] == seg[i]; j++) acc = op(acc, vals[j]);
        const u64 r = (u64)seg[i] - (u64)seg[0];
        if (r < n_segments) out[r] = acc;
        atomic_add32(cnt, 1u);
    }
};
template <class V, class Op>
inline void reduce_by_segment(const u32* seg, const V* vals, size_t n, V* out, size_t n_segments, Op op, u32* err = nullptr, stream_t s = 0) {
    if (!n) return;
    DBuf<u32> cnt(1);
    cnt.fill_bytes(0, s);
    launch(n, SegReduceFunctor<V, Op>{seg, vals, (u64)n, out, (u64)n_segments, op, cnt.ptr()}, s);
    if (err) launch(1, CountCheckFunctor{cnt.ptr(), (u32)n_segments, err, 128u}, s);
    else if (read_scalar(cnt.ptr(), s) != n_segments) throw DeviceError("reduce_by_segment: segment count mismatch");
}
// Arg-min per segment: `seg` holds non-decreasing segment ids (a new id starts a new segment, run r -> out[r] counted from the first id),
// the candidate of position i is the index i itself, `op(a, b)` returns whichever of two indices wins.
template <class Op> struct SegArgminFunctor {
    const u32* seg; u64 n; u32* out; u64 n_segments; Op op; u32* cnt;
    AC_D void operator()(u64 i) const {
        if (i > 0 && seg[i - 1] == seg[i]) return;
        u32 acc = (u32)i;
        for (u64 j = i + 1; j < n && seg[j] == seg[i]; j++) acc = op(acc, (u32)j);
        const u64 r = (u64)seg[i] - (u64)seg[0];
        if (r < n_segments) out[r] = acc;
        atomic_add32(cnt, 1u);
    }
};
template <class Op>
inline void segment_argmin(const u32* seg, size_t n, u32* out, size_t n_segments, Op op, u32* err = nullptr, stream_t s = 0) {
    if (!n) return;
    DBuf<u32> cnt(1);
    cnt.fill_bytes(0, s);
    launch(n, SegArgminFunctor<Op>{seg, (u64)n, out, (u64)n_segments, op, cnt.ptr()}, s);
    if (err) launch(1, CountCheckFunctor{cnt.ptr(), (u32)n_segments, err, 128u}, s);
    else if (read_scalar(cnt.ptr(), s) != n_segments) throw DeviceError("segment_argmin: segment count mismatch");
}

// Stable comparator sorts: a bottom-up merge sort by ranks.  Runs of `run` sorted items are merged pairwise; an item of the left run
// goes to (its index in the run) + (items of the right run that are smaller), an item of the right run to (its index) + (items of the left
// run that are not larger) — one binary search per item and pass, log2(n) passes.  Fallback paths only.
template <class K, class Cmp> struct MergeRankFunctor {
    const K* kin; const u32* vin; K* kout; u32* vout; u64 n, run; Cmp cmp;
    AC_D void operator()(u64 i) const {
        const u64 pair0 = i / (2 * run) * (2 * run);
        const u64 mid = pair0 + run < n ? pair0 + run : n, end = pair0 + 2 * run < n ? pair0 + 2 * run : n;
        const K me = kin[i];
        u64 lo, hi, dst;
        if (i < mid) {      // left run: right items strictly smaller than me come first
            lo = mid; hi = end;
            while (lo < hi) { const u64 m = lo + ((hi - lo) >> 1); if (cmp(kin[m], me)) lo = m + 1; else hi = m; }
            dst = pair0 + (i - pair0) + (lo - mid);
        } else {            // right run: left items not larger than me come first
            lo = pair0; hi = mid;
            while (lo < hi) { const u64 m = lo + ((hi - lo) >> 1); if (!cmp(me, kin[m])) lo = m + 1; else hi = m; }
            dst = pair0 + (i - mid) + (lo - pair0);
        }
        kout[dst] = me;
        if (vin) vout[dst] = vin[i];
    }
};
template <class K, class Cmp>
inline void merge_sort_impl(DBuf<K>& keys, DBuf<u32>* vals, size_t n, Cmp cmp, stream_t s) {
    if (n <= 1) return;
    DBuf<K> k2(n); DBuf<u32> v2(vals ? n : 0);
    K* ka = keys.ptr(); K* kb = k2.ptr(); u32* va = vals ? vals->ptr() : nullptr; u32* vb = vals ? v2.ptr() : nullptr;
    bool flipped = false;
    for (u64 run = 1; run < n; run *= 2) {
        launch(n, MergeRankFunctor<K, Cmp>{ka, va, kb, vb, (u64)n, run, cmp}, s);
        std::swap(ka, kb); std::swap(va, vb);
        flipped = !flipped;
    }
    if (flipped) { keys = std::move(k2); if (vals) *vals = std::move(v2); }
}
// Sort (key struct, u32 value) pairs with a comparator.
template <class K, class Cmp>
inline void sort_by_key_cmp(DBuf<K>& keys, DBuf<u32>& vals, size_t n, Cmp cmp, stream_t s = 0) { merge_sort_impl<K, Cmp>(keys, &vals, n, cmp, s); }
// Stable sort of u32 keys with a comparator (the comparator usually dereferences per-key device arrays).
template <class Cmp>
inline void sort_keys_cmp(DBuf<u32>& keys, size_t n, Cmp cmp, stream_t s = 0) { merge_sort_impl<u32, Cmp>(keys, nullptr, n, cmp, s); }

}  // namespace ac
