// Host entry of the device graph build: sequences in the caller's (pageable) memory -> packed text in HBM (2-bit pack on the host,
// pinned ring or direct stores through the BAR, the insert fed chunk by chunk), and the paths' final numbers applied by the same
// thread pool (PathRemapJob).  Replaces what Sequence::new_with_seq leaves in host memory at compress.rs:41 reaching the device.
#include "upload_rt.hpp"

namespace ac {

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
    if (knobs().debug_arena) fprintf(stderr, "direct upload self-test on device %d: %s\n", dev, ok ? "passed" : "FAILED (the packed upload goes through the pinned ring)");
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

void release_host_stager() { HostStager::get().release(); }
void ensure_host_stager() {      // (device_warmup: the whole-command path uploads the text as BYTES for the end repair — through the ring)
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
// Stretch s covers entries [rec_pos[s], rec_pos[s + 1]) (the last one: up to n_ent) with the values v0, v0 + 1, ...: positive ones are the
// final numbers of consecutive table entries, front to back; negative ones walk the table backwards, negated.
void path_stretch_range(const PathRemapJob& j, u64 s0, u64 s1, std::atomic<u32>* bad) {
    u32 wrong = 0;
    // (round 6: a stretch is ~10 entries — the copy loop's exit mispredicts and the out-of-order window does not reach the next stretch's table
    // row by itself, so a thread waited out one memory latency per stretch; the rows of the stretches AHEAD are prefetched.  Streaming
    // stores for the output were measured too: no gain — the job is bound by those row fetches, tools/microbench/host_write_probe.hip)
    const u64 AHEAD = 16;
    for (u64 s = s0; s < s1; s++) {
        if (s + AHEAD < s1) { const int64_t va = j.rec_val[s + AHEAD]; const u64 fa = (u64)(va > 0 ? va : -va) - 1; if (fa < (u64)j.n_unitigs) __builtin_prefetch(j.number + fa); }
        const int64_t v0 = j.rec_val[s];
        const u64 b = j.rec_pos[s], e = s + 1 < j.n_rec ? (u64)j.rec_pos[s + 1] : j.n_ent, len = e - b;
        if (b >= j.ent_limit) break;          // (the entries from there on are renumbered on the device: PathRemapJob::ent_limit)
        int32_t* out = j.path + b;
        if (e > j.n_ent || e <= b) { wrong++; continue; }
        if (v0 > 0) {
            const u64 first = (u64)v0 - 1;
            if (first + len > (u64)j.n_unitigs) { wrong++; continue; }
            const u32* src = j.number + first;
            for (u64 i = 0; i < len; i++) out[i] = (int32_t)src[i];
        } else {
            const u64 first = (u64)(-v0) - 1;      // values v0, v0 + 1, ... = -(first + 1), -first, ...
            if (v0 == 0 || first >= (u64)j.n_unitigs || len > first + 1) { wrong++; continue; }
            const u32* src = j.number + first;
            for (u64 i = 0; i < len; i++) out[i] = -(int32_t)*(src - i);
        }
    }
    if (wrong) bad->fetch_add(wrong);
}
// ---- SeqExpandJob: 2-bit codes -> the bytes of the unitig sequences ----------------------------------------------------------------------
static void seq_expand_scalar(const u64* words, u8* out, u64 b, u64 e) {
    for (u64 i = b; i < e; i++) out[i] = (u8)"ACGT"[(words[i >> 5] >> (2 * (i & 31))) & 3];
}
#if defined(__x86_64__)
// sixteen bytes of codes (64 bases) per step: the four 2-bit fields of every byte isolated, interleaved back into base order, looked up in "ACGT"
__attribute__((target("ssse3"))) static void seq_expand_ssse3(const u64* words, u8* out, u64 b, u64 e) {
    const __m128i m3 = _mm_set1_epi8(3), lut = _mm_setr_epi8('A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T');
    const u8* src = (const u8*)words;
    u64 i = b;
    for (; i + 64 <= e; i += 64) {
        const __m128i v = _mm_loadu_si128((const __m128i*)(src + i / 4));
        const __m128i x0 = _mm_and_si128(v, m3), x1 = _mm_and_si128(_mm_srli_epi16(v, 2), m3), x2 = _mm_and_si128(_mm_srli_epi16(v, 4), m3), x3 = _mm_and_si128(_mm_srli_epi16(v, 6), m3);
        const __m128i a = _mm_unpacklo_epi8(x0, x1), c = _mm_unpacklo_epi8(x2, x3), d = _mm_unpackhi_epi8(x0, x1), f = _mm_unpackhi_epi8(x2, x3);
        _mm_storeu_si128((__m128i*)(out + i), _mm_shuffle_epi8(lut, _mm_unpacklo_epi16(a, c)));
        _mm_storeu_si128((__m128i*)(out + i + 16), _mm_shuffle_epi8(lut, _mm_unpackhi_epi16(a, c)));
        _mm_storeu_si128((__m128i*)(out + i + 32), _mm_shuffle_epi8(lut, _mm_unpacklo_epi16(d, f)));
        _mm_storeu_si128((__m128i*)(out + i + 48), _mm_shuffle_epi8(lut, _mm_unpackhi_epi16(d, f)));
    }
    seq_expand_scalar(words, out, i, e);
}
#endif
void seq_expand_range(const u64* words, u8* out, u64 b, u64 e) {
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("ssse3") && getenv("AC_PACK_SCALAR") == nullptr;
    if (fast) { seq_expand_ssse3(words, out, b, e); return; }
#endif
    seq_expand_scalar(words, out, b, e);
}
#ifndef AC_EMU
void seq_expand_start(SeqExpandJob& j, int threads) {
    const u64 BLOCK = (u64)1 << 18;      // bytes of sequence per work item
    const int T = (int)std::max<u64>(1, std::min<u64>({(j.total + BLOCK - 1) / BLOCK, (u64)std::max(threads, 1), (u64)std::max(1u, std::thread::hardware_concurrency())}));
    SeqExpandJob* job = &j;
    j.started = true;
    j.t_start.store(now_s());
    j.ticket = UploadPool::second().start(T, [job, BLOCK] {
        int expect = 0;
        if (job->ready.compare_exchange_strong(expect, 1)) {      // one thread waits for the copy, the others watch it
            const bool ok = hipSetDevice(job->dev) == hipSuccess && hipEventSynchronize((hipEvent_t)job->landed) == hipSuccess;
            job->t_ready.store(now_s());
            job->ready.store(ok ? 2 : 3, std::memory_order_release);
        } else {
            while (job->ready.load(std::memory_order_acquire) < 2) std::this_thread::yield();
        }
        if (job->ready.load(std::memory_order_acquire) != 2) return;      // (seq_expand_finish's caller sees ready == 3)
        for (u64 b; (b = job->next.fetch_add(BLOCK)) < job->total;) seq_expand_range(job->words, job->out, b, std::min(b + BLOCK, job->total));
        job->t_last.store(now_s());
    });
}
void seq_expand_finish(SeqExpandJob& j) noexcept {
    if (!j.started) return;
    UploadPool::second().wait(j.ticket);
    j.started = false;
}
#else
void seq_expand_start(SeqExpandJob&, int) {}
void seq_expand_finish(SeqExpandJob&) noexcept {}
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
    const u64 work_items = j.rec_val ? (j.n_rec + 4095) / 4096 : (j.n_ent + BLOCK - 1) / BLOCK;
    const int T = (int)std::max<u64>(1, std::min<u64>({work_items, (u64)std::max(threads, 1), (u64)std::max(1u, std::thread::hardware_concurrency())}));
    PathRemapJob* job = &j;
    j.started = true;
    j.ticket = UploadPool::get().start(T, [job, BLOCK] {
        int expect = 0;
        if (job->ready.compare_exchange_strong(expect, 1)) {      // one thread waits for the copies, the others watch it
            const bool ok = hipSetDevice(job->dev) == hipSuccess && hipEventSynchronize((hipEvent_t)job->landed) == hipSuccess;
            job->t_ready.store(now_s());
            job->ready.store(ok ? 2 : 3, std::memory_order_release);
        } else {
            while (job->ready.load(std::memory_order_acquire) < 2) std::this_thread::yield();
        }
        if (job->ready.load(std::memory_order_acquire) != 2) { job->bad.fetch_add(1); return; }
        if (job->rec_val) {      // stretch mode: blocks of stretches
            const u64 SB = 4096;
            for (u64 b; (b = job->next.fetch_add(SB)) < job->n_rec;) path_stretch_range(*job, b, std::min(b + SB, job->n_rec), &job->bad);
            job->t_last.store(now_s());
            return;
        }
        for (u64 b; (b = job->next.fetch_add(BLOCK)) < job->n_ent;)
            path_remap_range(job->path + b, std::min(BLOCK, job->n_ent - b), job->number, job->n_unitigs, &job->bad);
        job->t_last.store(now_s());
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

}  // namespace ac
