// Device graph build (see graph_build.hpp).  gfx950 HIP; the same file compiles as the CPU emulation
// under -DAC_EMU for the CPU test-suite.
//
// (The shared types, kernels and GraphBuilder::Impl: graph_impl.hpp; the stages per key width: graph_stages.hip; the host entry:
// graph_upload.hip; the phases of a build over several devices: graph_shard.hip; end repair, distances, verifier: graph_extras.hip.)
#ifdef AC_EMU
#define AC_EMU_DEFINE_CTX_SWITCH      // (the lockstep emulation's context switch is defined by this translation unit: wave_rt.hpp)
#endif
#include "upload_rt.hpp"

namespace ac {

bool g_stage_timing = false;
void set_stage_timing(bool on) { g_stage_timing = on; }
bool stage_timing() { return g_stage_timing; }

// The knobs: one struct, read once (graph_impl.hpp).  AC_TUNING_FOLLOW_ENV=1, itself read once, makes every refresh read them again.
static Knobs& knobs_storage() { static Knobs k = Knobs::read(); return k; }
const Knobs& knobs() { return knobs_storage(); }
void knobs_refresh() {
    static const bool follow = getenv("AC_TUNING_FOLLOW_ENV") != nullptr;
    Knobs& k = knobs_storage();
    if (follow) k = Knobs::read();
}
void tuning_refresh() { knobs_refresh(); }
int tuning_multi_transport() { return knobs().multi_transport; }
bool tuning_multi_fragments_as_bytes() { return knobs().multi_fragments != 0; }
bool tuning_multi_tail_replicated() { return knobs().multi_tail != 0; }

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

}  // namespace ac
