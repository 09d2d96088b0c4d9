// Thin device runtime used by the graph-build pipeline: buffers, copies, a functor launcher and (device_prims.hpp) the plain device
// primitives — radix sort, scans, comparator sort, segmented reduce — hand-written since round 5.
// HIP build: hipMalloc / hipLaunchKernelGGL on one stream of one gfx950 device.
// AC_EMU build (tests only): malloc / the same kernels under the lockstep emulation of wave_rt.hpp.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>

#include <mutex>
#include <time.h>
#include <unistd.h>

#include "kmer_ops.hpp"
#include "graph_types.hpp"
#include "wave_rt.hpp"

#ifndef AC_EMU
#include <hip/hip_runtime.h>
#endif

namespace ac {

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct NeedExactPositions {};      // thrown by the tail, caught by GraphBuilder::build (graph_build.hip)
struct NeedCheckedSorts {};        // ... a sort whose "group too large" flag was only looked at with the build's last read-back had it set: repeat, checking at once

#ifndef AC_EMU
#define AC_HIP_CHECK(expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            throw ::ac::DeviceError(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                                    std::to_string(__LINE__) + " (" #expr ")");                        \
    } while (0)
typedef hipStream_t stream_t;
#else
typedef int stream_t;
#endif

// ---- atomics --------------------------------------------------------------------------------------
#ifdef AC_EMU
inline u64 atomic_cas64(u64* p, u64 expected, u64 desired) { u64 old = *p; if (old == expected) *p = desired; return old; }
inline u64 atomic_min64(u64* p, u64 v) { u64 o = *p; if (v < o) *p = v; return o; }
inline void atomic_max64(u64* p, u64 v) { if (v > *p) *p = v; }
inline void atomic_xor64(u64* p, u64 v) { *p ^= v; }
inline u32 atomic_add32(u32* p, u32 v) { u32 o = *p; *p += v; return o; }
inline u64 atomic_add64(u64* p, u64 v) { u64 o = *p; *p += v; return o; }
inline void atomic_or32(u32* p, u32 v) { *p |= v; }
inline void atomic_or64(u64* p, u64 v) { *p |= v; }
inline void atomic_min32(u32* p, u32 v) { if (v < *p) *p = v; }
inline void atomic_max32(u32* p, u32 v) { if (v > *p) *p = v; }
inline u32 atomic_cas32(u32* p, u32 expected, u32 desired) { u32 old = *p; if (old == expected) *p = desired; return old; }
inline u32 atomic_load32(const u32* p) { return *p; }
inline u32 atomic_fetch_or32(u32* p, u32 v) { u32 o = *p; *p |= v; return o; }
inline u32 atomic_fetch_and32(u32* p, u32 v) { u32 o = *p; *p &= v; return o; }
#else
__device__ inline u64 atomic_cas64(u64* p, u64 expected, u64 desired) {
    return (u64)atomicCAS((unsigned long long*)p, (unsigned long long)expected, (unsigned long long)desired);
}
__device__ inline u64 atomic_min64(u64* p, u64 v) { return (u64)atomicMin((unsigned long long*)p, (unsigned long long)v); }
__device__ inline void atomic_max64(u64* p, u64 v) { atomicMax((unsigned long long*)p, (unsigned long long)v); }
__device__ inline void atomic_xor64(u64* p, u64 v) { atomicXor((unsigned long long*)p, (unsigned long long)v); }
__device__ inline u32 atomic_add32(u32* p, u32 v) { return atomicAdd(p, v); }
__device__ inline u64 atomic_add64(u64* p, u64 v) { return (u64)atomicAdd((unsigned long long*)p, (unsigned long long)v); }
__device__ inline void atomic_or32(u32* p, u32 v) { atomicOr(p, v); }
__device__ inline void atomic_or64(u64* p, u64 v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
__device__ inline void atomic_min32(u32* p, u32 v) { atomicMin(p, v); }
__device__ inline void atomic_max32(u32* p, u32 v) { atomicMax(p, v); }
__device__ inline u32 atomic_cas32(u32* p, u32 expected, u32 desired) { return atomicCAS(p, expected, desired); }
__device__ inline u32 atomic_load32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // (past this CU's L1: another CU's store is seen)
__device__ inline u32 atomic_fetch_or32(u32* p, u32 v) { return atomicOr(p, v); }
__device__ inline u32 atomic_fetch_and32(u32* p, u32 v) { return atomicAnd(p, v); }
#endif

// ---- per-thread counters of what a build asks of the runtime (ac_timings.launches / .readbacks) ---------------------------------------
struct RtCounters { u32 launches = 0, readbacks = 0; };
inline RtCounters& rt_counters() { static thread_local RtCounters c; return c; }

// ---- device context ------------------------------------------------------------------------------------------------------------
// Everything a build keeps between its stages and between builds — the device arena, the fill queue, the read-back mailbox, the side
// stream, the upload ring and its packing threads — lives in a DeviceCtx.  A thread that has not been given one uses the process-wide
// context (the single-device entries: one build at a time, whatever thread calls).  ac_compress_build_multi runs one host thread per
// device and gives each of them a context of its own (set_device_ctx), so that N builds drive N devices side by side.
struct DeviceCtx {
    static const int SLOTS = 10;
    void* obj[SLOTS] = {}; void (*del[SLOTS])(void*) = {};
    int arena_device = -1;      // the device the arena's blocks live on
    DeviceCtx() {}
    DeviceCtx(const DeviceCtx&) = delete;
    DeviceCtx& operator=(const DeviceCtx&) = delete;
    ~DeviceCtx() { for (int i = SLOTS; i-- > 0;) if (obj[i]) del[i](obj[i]); }
};
inline DeviceCtx*& tl_device_ctx() { static thread_local DeviceCtx* p = nullptr; return p; }
inline DeviceCtx& device_ctx() {      // (the process-wide one is never destroyed: at exit the HIP runtime may be gone before the statics)
    static DeviceCtx* const process_wide = new DeviceCtx;
    DeviceCtx* p = tl_device_ctx();
    return p ? *p : *process_wide;
}
inline void set_device_ctx(DeviceCtx* c) { tl_device_ctx() = c; }      // nullptr: back to the process-wide context
enum { CTX_ARENA = 0, CTX_FILLS, CTX_MAILBOX, CTX_SIDE, CTX_SCRATCH, CTX_STAGER, CTX_POOL, CTX_SCANPOOL_RESERVED, CTX_POOL2 };      // (7: device_prims.hpp's scan pool)
template <class T> T& ctx_object(int slot) {
    DeviceCtx& c = device_ctx();
    if (!c.obj[slot]) { c.obj[slot] = new T(); c.del[slot] = [](void* p) { delete (T*)p; }; }
    return *(T*)c.obj[slot];
}

// ---- buffers --------------------------------------------------------------------------------------
// Device memory comes from a persistent bump arena (one per process, on the device selected by the C ABI):
// a build makes ~40 allocations, and hipMalloc/hipFree (the latter synchronises the device) would cost more
// than the kernels.  The arena is reset at the start of every build, grows by whole blocks, and coalesces to
// one block on the next reset; MI355X has 288 GB of HBM, so the arena simply stays resident between builds
// (ac_release_memory() in the C ABI frees it).
class Arena {
  public:
    static Arena& device() { return ctx_object<Arena>(CTX_ARENA); }
    Arena() : host_(false) {}
    ~Arena() { for (auto& b : blocks_) raw_free(b.p); }
    static const size_t COALESCE_LIMIT = (size_t)16 << 30;
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        // first fit, forwards only: blocks behind the current one are empty (a rewind emptied them) or do not exist yet
        size_t i = cur_;
        while (i < blocks_.size() && blocks_[i].used + bytes > blocks_[i].cap) i++;
        if (i == blocks_.size()) {
            Block b; b.cap = std::max(bytes, grow_); b.used = 0; b.p = raw_alloc(b.cap);
            blocks_.push_back(b);
        }
        cur_ = i;
        Block& b = blocks_[i];
        void* r = (char*)b.p + b.used;
        b.used += bytes;
        peak_ = std::max(peak_, total_used());
        return r;
    }
    // Start of a build: everything handed out so far is dead.
    void reset() {
        on_reset();
        const size_t last_peak = peak_;
        peak_ = 0; cur_ = 0;
        size_t total = 0;
        for (auto& b : blocks_) total += b.cap;
        if (blocks_.size() > 1 && total <= COALESCE_LIMIT) {   // coalesce: next build gets one block big enough for the last one (what it used
            for (auto& b : blocks_) raw_free(b.p);              // at its peak plus an eighth; the capacities add up to more: tails a large request skipped)
            if (last_peak) total = std::min(total, last_peak + last_peak / 8 + ((size_t)256 << 20));
            blocks_.clear();
            Block b; b.cap = total; b.used = 0; b.p = raw_alloc(total);
            blocks_.push_back(b);
        } else {   // a large arena keeps its blocks: the next build of the same job makes the same requests in the same order and fits them the same
            for (auto& b : blocks_) b.used = 0;      // way, and giving 150 GB back and asking for it again costs seconds (configs[4]: 8 s)
        }
    }
    // First build of a process: one block of about the size the build will need, so that neither this build grows the
    // arena block by block nor the next one pays for coalescing (hipFree + hipMalloc of gigabytes).
    void reserve(size_t bytes) {
        if (capacity() >= bytes || total_used() != 0) return;
        release_all();
        Block b; b.cap = bytes; b.used = 0; b.p = raw_alloc(bytes);
        blocks_.push_back(b);
    }
    void release_all() {
        on_reset();
        for (auto& b : blocks_) raw_free(b.p);
        blocks_.clear(); cur_ = 0;
    }
    // Stage-local temporaries of a build that is larger than usual (BASELINE configs[4]: 270 GB without this): take a mark, allocate, and
    // rewind when the stage is done — everything allocated since the mark is dead.  All work runs in order on stream 0, so memory
    // handed out again after a rewind cannot be touched by what used it before; blocks that were added since the mark go back to the
    // arena's later allocations.
    struct Mark { size_t block, used; bool empty; };
    Mark mark() const { return blocks_.empty() ? Mark{0, 0, true} : Mark{cur_, blocks_[cur_].used, false}; }
    void rewind(const Mark& m) {
        on_rewind();
        if (blocks_.empty()) return;
        cur_ = m.empty ? 0 : m.block;
        blocks_[cur_].used = m.empty ? 0 : m.used;
        for (size_t i = cur_ + 1; i < blocks_.size(); i++) blocks_[i].used = 0;      // kept for the allocations to come (no hipFree: it waits for the device)
    }
    size_t peak() const { return peak_; }
    double alloc_seconds() const { return alloc_s_; }
    size_t total_used() const { size_t t = 0; for (auto& b : blocks_) t += b.used; return t; }
    size_t capacity() const { size_t t = 0; for (auto& b : blocks_) t += b.cap; return t; }
    void set_grow(size_t g) { grow_ = g; }

  private:
    struct Block { void* p; size_t cap, used; };
    void on_reset();      // (defined after FillQueue)
    void on_rewind();
    static double wall() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    void* raw_alloc(size_t bytes) {
        const double t0 = wall();
        struct Acc { double& a; double t0; ~Acc() { a += wall() - t0; } } acc{alloc_s_, t0};
#ifdef AC_EMU
        void* p = malloc(bytes);
        if (!p) throw DeviceError("emu malloc failed");
        return p;
#else
        void* p = nullptr;
        if (host_) AC_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        else AC_HIP_CHECK(hipMalloc(&p, bytes));
        return p;
#endif
    }
    void raw_free(void* p) {
        const double t0 = wall();
#ifdef AC_EMU
        free(p);
#else
        if (host_) (void)hipHostFree(p); else (void)hipFree(p);
#endif
        alloc_s_ += wall() - t0;
    }
    double alloc_s_ = 0;      // seconds spent in the runtime's allocator so far (AC_DEBUG_ARENA)
    bool host_;
    std::vector<Block> blocks_;
    size_t grow_ = (size_t)64 << 20;
    size_t peak_ = 0, cur_ = 0;
};

// Recycling pool of pinned host blocks for results that outlive a build (they are owned by the caller's graph
// handle).  hipHostMalloc costs milliseconds for tens of MB; in steady state (free the previous graph, build the
// next) every block is reused.
class PinnedPool {
  public:
    static PinnedPool& get() { static PinnedPool p; return p; }
    HostBlock alloc(size_t bytes) {
        if (bytes == 0) bytes = 1;
        HostBlock b;
        {
            std::lock_guard<std::mutex> lock(mu_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].cap >= bytes && free_[i].cap <= 2 * bytes + 4096 && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
            if (best != free_.size()) {
                b.p = free_[best].p; b.bytes = free_[best].cap;
                held_ -= free_[best].cap;
                free_.erase(free_.begin() + (long)best);
            }
        }
        if (!b.p) {
            size_t cap = bytes + bytes / 8 + 4096;
            if (getenv("AC_DEBUG_ARENA")) fprintf(stderr, "pinned pool: new block of %zu bytes (%zu free entries)\n", cap, free_.size());
#ifdef AC_EMU
            cap = (cap + 4095) & ~(size_t)4095;
            b.p = aligned_alloc(4096, cap);      // (page-aligned like hipHostMalloc's blocks: the streaming stretch writer wants whole 64-byte lines)
            if (!b.p) throw DeviceError("out of host memory");
#else
            AC_HIP_CHECK(hipHostMalloc(&b.p, cap, hipHostMallocDefault));
#endif
            b.bytes = cap;
        }
        b.release = &PinnedPool::give_back;
        return b;
    }
    void trim() {
        std::lock_guard<std::mutex> lock(mu_);
        for (auto& e : free_) raw_free(e.p);
        free_.clear(); held_ = 0;
    }

  private:
    struct Entry { void* p; size_t cap; };
    static void raw_free(void* p) {
#ifdef AC_EMU
        free(p);
#else
        (void)hipHostFree(p);
#endif
    }
    static void give_back(void* p, size_t cap) {
        PinnedPool& self = get();
        std::lock_guard<std::mutex> lock(self.mu_);
        if (self.free_.size() >= 32 || self.held_ + cap > keep_limit()) { raw_free(p); return; }
        self.free_.push_back(Entry{p, cap});
        self.held_ += cap;
    }
    // Pinned bytes kept for the next build: 8 GB, or a sixteenth of the host's memory up to 64 GB (the results of a BASELINE configs[4]
    // job are 14 GB, and pinning that much anew costs seconds per build); AC_PINNED_POOL_GB overrides.
    static size_t keep_limit() {
        static const size_t v = [] {
            if (const char* e = getenv("AC_PINNED_POOL_GB")) return (size_t)std::max(0, atoi(e)) << 30;
            size_t phys = 0;
#if defined(_SC_PHYS_PAGES) && defined(_SC_PAGE_SIZE)
            const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
            if (pages > 0 && psz > 0) phys = (size_t)pages * (size_t)psz;
#endif
            return std::min<size_t>(std::max<size_t>((size_t)8 << 30, phys / 16), (size_t)64 << 30);
        }();
        return v;
    }
    std::mutex mu_;
    std::vector<Entry> free_;
    size_t held_ = 0;
};

// ---- deferred, fused fills ---------------------------------------------------------------------------------------------------
// A build zeroes / 0xFF-fills ~50 buffers, most of them a few hundred KB: as hipMemsetAsync each is its own 4-6 us launch and the
// host cannot issue them faster.  Fills on stream 0 are queued instead and go out as ONE kernel right before the next operation on
// that stream (every launch, copy, sort, scan and synchronisation of this file calls flush_fills() first; the few direct HIP calls
// of graph_build.hip do the same).  Fills on any other stream are issued at once.
#ifndef AC_EMU
static const int FILL_MAX = 16;
struct FillArgs { void* p[FILL_MAX]; u64 bytes[FILL_MAX]; u64 tile0[FILL_MAX + 1]; u32 word[FILL_MAX]; int n; };
static const u64 FILL_MAX_TILES = (u64)1 << 23;
template <int UNUSED> __global__ void __launch_bounds__(256) fill_many_kernel(FillArgs a, u64 tile0) {      // a tile = 16 KB of one region
    const u64 tile = tile0 + blockIdx.x;
    int r = 0;
    while (r + 1 < a.n && a.tile0[r + 1] <= tile) r++;
    const u64 base = (tile - a.tile0[r]) * 16384;
    u8* p = (u8*)a.p[r];
    const u64 nb = a.bytes[r];
    const u32 w = a.word[r];
    const uint4 v = make_uint4(w, w, w, w);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u64 o = base + (u64)j * 4096 + (u64)threadIdx.x * 16;
        if (o + 16 <= nb) *(uint4*)(p + o) = v;
        else if (o < nb) { for (u64 b = o; b < nb; b++) p[b] = (u8)w; }
    }
}
class FillQueue {
  public:
    static FillQueue& get() { return ctx_object<FillQueue>(CTX_FILLS); }
    FillQueue() {}
    void add(void* p, size_t bytes, int byte) {
        if (n_ == FILL_MAX) flush();
        const u32 b = (u32)(byte & 0xFF);
        a_.p[n_] = p; a_.bytes[n_] = bytes; a_.word[n_] = b | (b << 8) | (b << 16) | (b << 24);
        n_++;
    }
    void flush() {
        if (!n_) return;
        u64 tiles = 0;
        for (int i = 0; i < n_; i++) { a_.tile0[i] = tiles; tiles += (a_.bytes[i] + 16383) / 16384; }
        a_.tile0[n_] = tiles;
        a_.n = n_;
        n_ = 0;
        for (u64 t0 = 0; t0 < tiles; t0 += FILL_MAX_TILES) {      // (2^32 threads per launch at most: 64 GB of fills)
            rt_counters().launches++;
            hipLaunchKernelGGL(fill_many_kernel<0>, dim3((unsigned)std::min(FILL_MAX_TILES, tiles - t0)), dim3(256), 0, 0, a_, t0);
            AC_HIP_CHECK(hipGetLastError());
        }
    }
    void drop() { n_ = 0; }      // the arena was reset: whatever was queued points at dead buffers
  private:
    FillArgs a_;
    int n_ = 0;
};
inline void flush_fills() { FillQueue::get().flush(); }
inline void Arena::on_reset() { if (!host_) FillQueue::get().drop(); }
inline void Arena::on_rewind() { if (!host_) FillQueue::get().flush(); }      // queued fills of buffers that stay alive must not be lost
#else
inline void flush_fills() {}
inline void Arena::on_reset() {}
inline void Arena::on_rewind() {}
#endif

template <class T>
class DBuf {   // a typed slice of the device arena (no ownership: the arena reset frees everything)
  public:
    DBuf() {}
    explicit DBuf(size_t n, bool zero = false) { alloc(n, zero); }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DBuf& operator=(DBuf&& o) noexcept { if (this != &o) { p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
    void alloc(size_t n, bool zero = false) {
        n_ = n;
        size_t bytes = (n ? n : 1) * sizeof(T);
        p_ = (T*)Arena::device().alloc(bytes);
        if (zero) fill_bytes(0);
    }
    void fill_bytes(int byte, stream_t s = 0) {
        size_t bytes = n_ * sizeof(T);
        if (!bytes) return;
#ifdef AC_EMU
        memset(p_, byte, bytes);
#else
        if (s == 0) FillQueue::get().add(p_, bytes, byte);      // deferred: fused with the other fills of this stage
        else AC_HIP_CHECK(hipMemsetAsync(p_, byte, bytes, s));
#endif
    }
    // The same from byte `from` (rounded down to 16) to the end: for a buffer whose front a kernel is about to overwrite in full.
    void fill_bytes_from(size_t from, int byte, stream_t s = 0) {
        const size_t bytes = n_ * sizeof(T);
        from &= ~(size_t)15;
        if (from >= bytes) return;
#ifdef AC_EMU
        memset((u8*)p_ + from, byte, bytes - from);
#else
        if (s == 0) FillQueue::get().add((u8*)p_ + from, bytes - from, byte);
        else AC_HIP_CHECK(hipMemsetAsync((u8*)p_ + from, byte, bytes - from, s));
#endif
    }
    void fill_bytes_first(size_t upto, int byte) {      // bytes [0, upto)
        const size_t bytes = std::min(n_ * sizeof(T), upto);
        if (!bytes) return;
#ifdef AC_EMU
        memset(p_, byte, bytes);
#else
        FillQueue::get().add(p_, bytes, byte);
#endif
    }
    T* ptr() { return p_; }
    const T* ptr() const { return p_; }
    size_t size() const { return n_; }
    size_t bytes() const { return n_ * sizeof(T); }

  private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

inline void copy_h2d(void* d, const void* h, size_t bytes, stream_t s = 0) {
    if (!bytes) return;
#ifdef AC_EMU
    memcpy(d, h, bytes);
#else
    if (s == 0) flush_fills();
    AC_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
#endif
}
inline void copy_d2h_async(void* h, const void* d, size_t bytes, stream_t s = 0) {
    if (!bytes) return;
#ifdef AC_EMU
    memcpy(h, d, bytes);
#else
    if (s == 0) flush_fills();
    AC_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
#endif
}
// Small read-backs (counts, flags, level tables: the ~20 host decisions of a build) land in a pinned scratch page first: a
// copy into pageable memory takes the runtime's slow staging path and costs tens of microseconds of idle GPU each time.
inline void* pinned_scratch(size_t bytes) {
#ifdef AC_EMU
    (void)bytes; return nullptr;
#else
    struct Scratch { void* p = nullptr; size_t cap = 0; ~Scratch() { if (p) (void)hipHostFree(p); } };
    Scratch& sc = ctx_object<Scratch>(CTX_SCRATCH);
    void*& p = sc.p;
    size_t& cap = sc.cap;
    if (bytes > cap) {
        if (p) (void)hipHostFree(p);
        cap = std::max<size_t>(bytes, 64 << 10);
        AC_HIP_CHECK(hipHostMalloc(&p, cap, hipHostMallocDefault));
    }
    return p;
#endif
}
// ---- mailbox read-back -------------------------------------------------------------------------------------------------------
// A build takes ~20 small host decisions (counts, flags, level tables).  As hipMemcpyAsync + hipStreamSynchronize each of them
// costs 17-25 us of idle GPU (a blit kernel, its completion signal, the runtime's wake-up).  Instead a tiny kernel on the same
// stream copies the words into a page of coherent pinned host memory that is mapped into the device's address space and then
// stores a sequence number behind a system-scope fence; the host spins on that number.  Nothing but the PCIe write latency
// (~2 us) stands between the producing kernel's end and the host seeing the value.
#ifndef AC_EMU
struct MailItem { const void* src; u32 bytes; u32 dst_off; };
static const int MAIL_MAX_ITEMS = 6;
struct MailArgs { MailItem it[MAIL_MAX_ITEMS]; int n; u8* box; u64 seq; };
template <int UNUSED> __global__ void __launch_bounds__(256) mailbox_publish_kernel(MailArgs a) {      // a template only so that every translation unit may hold a copy
    for (int i = 0; i < a.n; i++) {
        const u8* s = (const u8*)a.it[i].src;
        u8* d = a.box + 64 + a.it[i].dst_off;
        const u32 nb = a.it[i].bytes;
        if ((((uintptr_t)s | (uintptr_t)d | nb) & 7u) == 0) { for (u32 o = threadIdx.x * 8; o < nb; o += 256 * 8) *(volatile u64*)(d + o) = *(const u64*)(s + o); }
        else if ((((uintptr_t)s | (uintptr_t)d | nb) & 3u) == 0) { for (u32 o = threadIdx.x * 4; o < nb; o += 256 * 4) *(volatile u32*)(d + o) = *(const u32*)(s + o); }
        else { for (u32 o = threadIdx.x; o < nb; o += 256) *(volatile u8*)(d + o) = s[o]; }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); *(volatile u64*)a.box = a.seq; }
}
class Mailbox {
  public:
    static const size_t CAP = (size_t)64 << 10;      // payload bytes
    static Mailbox& get() { return ctx_object<Mailbox>(CTX_MAILBOX); }
    Mailbox() {}
    ~Mailbox() { release(); }
    // Copies the items (device -> host) behind everything enqueued on stream `s` so far and waits for them.
    void fetch(const MailItem* items, void* const* host_dst, int n, stream_t s) {
        ensure();
        if (s == 0) flush_fills();
        MailArgs a;
        a.n = n; a.box = d_; a.seq = ++seq_;
        for (int i = 0; i < n; i++) a.it[i] = items[i];
        rt_counters().launches++; rt_counters().readbacks++;
        hipLaunchKernelGGL(mailbox_publish_kernel<0>, dim3(1), dim3(256), 0, s, a);
        AC_HIP_CHECK(hipGetLastError());
        volatile u64* flag = (volatile u64*)h_;
        for (u64 spins = 0; *flag != a.seq; spins++) {
            if ((spins & 0xFFFFF) == 0xFFFFF) {      // ~every few ms: has the stream died under us?
                hipError_t e = hipStreamQuery(s);
                if (e != hipSuccess && e != hipErrorNotReady) AC_HIP_CHECK(e);
                if (e == hipSuccess && *flag != a.seq) {      // the kernel is done but its store is not visible: leave through the slow path
                    AC_HIP_CHECK(hipStreamSynchronize(s));
                    if (*flag != a.seq) throw DeviceError("mailbox: the device finished without delivering its read-back");
                }
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int i = 0; i < n; i++) memcpy(host_dst[i], h_ + 64 + items[i].dst_off, items[i].bytes);
    }
    void release() {
        if (h_) (void)hipHostFree(h_);
        h_ = nullptr; d_ = nullptr;
    }
  private:
    void ensure() {      // the page and its device-side address belong to the device that was current when it was mapped
        int dev = 0;
        AC_HIP_CHECK(hipGetDevice(&dev));
        if (h_ && dev == dev_) return;
        release();
        AC_HIP_CHECK(hipHostMalloc((void**)&h_, CAP + 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(h_, 0, CAP + 64);
        AC_HIP_CHECK(hipHostGetDevicePointer((void**)&d_, h_, 0));
        dev_ = dev; seq_ = 0;
    }
    u8* h_ = nullptr; u8* d_ = nullptr;
    u64 seq_ = 0;
    int dev_ = -1;
};
inline bool use_mailbox() { static const bool v = getenv("AC_NO_MAILBOX") == nullptr; return v; }
#endif
inline void copy_d2h(void* h, const void* d, size_t bytes, stream_t s = 0) {
    if (!bytes) return;
#ifndef AC_EMU
    if (s == 0) flush_fills();
    if (use_mailbox() && bytes <= Mailbox::CAP) {
        MailItem it{d, (u32)bytes, 0};
        Mailbox::get().fetch(&it, &h, 1, s);
        return;
    }
    static const bool use_scratch = getenv("AC_NO_PINNED_SCRATCH") == nullptr;
    if (use_scratch && bytes <= (64 << 10)) {
        void* p = pinned_scratch(bytes);
        rt_counters().readbacks++;
        AC_HIP_CHECK(hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s));
        AC_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(h, p, bytes);
        return;
    }
#endif
    copy_d2h_async(h, d, bytes, s);
#ifndef AC_EMU
    AC_HIP_CHECK(hipStreamSynchronize(s));
#endif
}
// Several small device arrays -> host with ONE synchronisation (one mailbox publication, or the pinned scratch page).
class ReadBatch {
  public:
    void add(void* h, const void* d, size_t bytes) { if (bytes) items_.push_back(Item{h, d, bytes}); }
    void run(stream_t s = 0) {
#ifdef AC_EMU
        for (auto& it : items_) memcpy(it.h, it.d, it.bytes);
        (void)s;
#else
        size_t total = 0;
        for (auto& it : items_) total += (it.bytes + 63) & ~(size_t)63;
        if (total == 0) return;
        if (s == 0) flush_fills();
        if (use_mailbox() && total <= Mailbox::CAP && items_.size() <= (size_t)MAIL_MAX_ITEMS) {
            MailItem mi[MAIL_MAX_ITEMS]; void* dst[MAIL_MAX_ITEMS];
            size_t o = 0;
            for (size_t i = 0; i < items_.size(); i++) { mi[i] = MailItem{items_[i].d, (u32)items_[i].bytes, (u32)o}; dst[i] = items_[i].h; o += (items_[i].bytes + 63) & ~(size_t)63; }
            Mailbox::get().fetch(mi, dst, (int)items_.size(), s);
            items_.clear();
            return;
        }
        char* p = (char*)pinned_scratch(total);
        rt_counters().readbacks++;
        size_t o = 0;
        for (auto& it : items_) { AC_HIP_CHECK(hipMemcpyAsync(p + o, it.d, it.bytes, hipMemcpyDeviceToHost, s)); o += (it.bytes + 63) & ~(size_t)63; }
        AC_HIP_CHECK(hipStreamSynchronize(s));
        o = 0;
        for (auto& it : items_) { memcpy(it.h, p + o, it.bytes); o += (it.bytes + 63) & ~(size_t)63; }
#endif
        items_.clear();
    }
  private:
    struct Item { void* h; const void* d; size_t bytes; };
    std::vector<Item> items_;
};

inline void copy_d2d(void* dst, const void* src, size_t bytes, stream_t s = 0) {
    if (!bytes) return;
#ifdef AC_EMU
    memmove(dst, src, bytes);
#else
    if (s == 0) flush_fills();
    AC_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
#endif
}
inline void stream_sync(stream_t s = 0) {
#ifndef AC_EMU
    if (s == 0) flush_fills();
    AC_HIP_CHECK(hipStreamSynchronize(s));
#endif
}
template <class T> std::vector<T> to_host(const DBuf<T>& b, size_t n, stream_t s = 0) {
    std::vector<T> v(n);
    copy_d2h(v.data(), b.ptr(), n * sizeof(T), s);
    return v;
}
template <class T> std::vector<T> to_host_ptr(const T* d, size_t n, stream_t s = 0) {
    std::vector<T> v(n);
    copy_d2h(v.data(), d, n * sizeof(T), s);
    return v;
}
template <class T> T read_scalar(const T* dptr, stream_t s = 0) {
    T v;
    copy_d2h(&v, dptr, sizeof(T), s);
    return v;
}

// A second stream for device -> host copies that overlap the kernels of stream 0 (the final D2H is PCIe-bound: the
// sooner each result array starts to move, the less of the copy is exposed).  Created without the implicit
// synchronisation with stream 0; ordering is by events.
class SideStream {
  public:
    static const unsigned N_EV = 64;
    static SideStream& get() { return ctx_object<SideStream>(CTX_SIDE); }
    SideStream() {}
    ~SideStream() { destroy(); }
    // which: 0 = the side stream; 1 = a second one (round 6: large device -> host copies alternate between the two — a copy queue of its own each)
    stream_t stream(int which = 0) {
#ifndef AC_EMU
        int dev = 0;
        AC_HIP_CHECK(hipGetDevice(&dev));
        if (!created_ || dev != dev_) {
            destroy();
            AC_HIP_CHECK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
            AC_HIP_CHECK(hipStreamCreateWithFlags(&s2_, hipStreamNonBlocking));
            for (auto& e : ev_) AC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            created_ = true; dev_ = dev;
        }
        return which ? s2_ : s_;
#else
        (void)which;
        return 0;
#endif
    }
    // Everything enqueued on stream 0 so far happens before whatever is enqueued on the side stream from now on.
    void after_main(int which = 0) {
#ifndef AC_EMU
        stream_t s = stream(which);
        hipEvent_t e = ev_[next_++ % N_EV];
        flush_fills();
        AC_HIP_CHECK(hipEventRecord(e, 0));
        AC_HIP_CHECK(hipStreamWaitEvent(s, e, 0));
#else
        (void)which;
#endif
    }
    // An event that fires when everything enqueued on STREAM 0 so far is done, and the host's wait for it (a poll: the caller is about to issue
    // the copy that needed it — see GraphBuilder::Impl::tail, "late copies").  Valid until N_EV more events were taken.
    void* main_event() {
#ifndef AC_EMU
        (void)stream();
        hipEvent_t e = ev_[next_++ % N_EV];
        flush_fills();
        AC_HIP_CHECK(hipEventRecord(e, 0));
        return (void*)e;
#else
        return nullptr;
#endif
    }
    static void wait_event(void* ev) {
#ifndef AC_EMU
        for (;;) {
            const hipError_t e = hipEventQuery((hipEvent_t)ev);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) AC_HIP_CHECK(e);
#if defined(__x86_64__)
            for (int i = 0; i < 32; i++) __builtin_ia32_pause();
#endif
        }
#else
        (void)ev;
#endif
    }
    // An event that fires when everything enqueued on the side stream so far is done (valid until N_EV more events were taken).
    void* mark() {
#ifndef AC_EMU
        stream_t s = stream();
        hipEvent_t e = ev_[next_++ % N_EV];
        AC_HIP_CHECK(hipEventRecord(e, s));
        return (void*)e;
#else
        return nullptr;
#endif
    }
    void sync() noexcept {
#ifndef AC_EMU
        if (created_) { (void)hipStreamSynchronize(s_); (void)hipStreamSynchronize(s2_); }
#endif
    }
    struct Guard { ~Guard() { SideStream::get().sync(); } };   // no copy may outlive the scope that owns its destination

  private:
    void destroy() {
#ifndef AC_EMU
        if (created_) { (void)hipStreamDestroy(s_); (void)hipStreamDestroy(s2_); for (auto& e : ev_) (void)hipEventDestroy(e); created_ = false; }
#endif
    }
#ifndef AC_EMU
    hipStream_t s_ = nullptr, s2_ = nullptr;
    hipEvent_t ev_[64];
#endif
    bool created_ = false;
    int dev_ = -1;
    unsigned next_ = 0;
};

// ---- functor launcher -----------------------------------------------------------------------------
// One logical thread per index, 256-thread workgroups (4 wavefronts); every launch in the pipeline
// has >> 256 workgroups at the benchmark sizes, so the 256 CUs / 8 XCDs fill from the grid alone.
#ifndef AC_EMU
template <class F>
__global__ void __launch_bounds__(256) functor_kernel(u64 n, F f, u64 base = 0) {
    u64 tid = base + (u64)blockIdx.x * 256 + threadIdx.x;
    if (tid < n) f(tid);
}
// A launch may not hold more than 2^32 - 1 threads (hipErrorInvalidConfiguration beyond that): a text of more than 4 G positions
// (BASELINE configs[4]: 5 G bp) takes several launches of at most this many 256-thread workgroups.
static const u64 MAX_LAUNCH_BLOCKS = (u64)1 << 23;
#endif
#ifdef AC_EMU
inline int emu_order() { const char* e = getenv("AC_EMU_ORDER"); return e ? atoi(e) : 0; }
#endif
#ifndef AC_EMU
// AC_DEBUG_LAUNCH=1: every functor launch is announced on stderr and waited for — the last line before a device fault names the kernel.
inline bool debug_launch() { static const bool v = [] { const char* e = getenv("AC_DEBUG_LAUNCH"); return e && *e == '1'; }(); return v; }
template <class F> void debug_launch_note(u64 n, bool after) {
    if (!after) { fprintf(stderr, "[launch] n=%llu %s\n", (unsigned long long)n, __PRETTY_FUNCTION__); fflush(stderr); }
    else AC_HIP_CHECK(hipDeviceSynchronize());
}
#endif
template <class F> void launch(u64 n, const F& f, stream_t s = 0) {
    if (n == 0) return;
#ifdef AC_EMU
    // The serial emulation can visit the logical threads in three orders (AC_EMU_ORDER=0/1/2: ascending,
    // descending, pseudo-random) so the tests can check that no result depends on scheduling.
    int order = emu_order();
    if (order == 1) { for (u64 i = n; i-- > 0;) f(i); }
    else if (order == 2) {
        std::vector<u64> perm(n);
        for (u64 i = 0; i < n; i++) perm[i] = i;
        u64 st = 0x9E3779B97F4A7C15ULL ^ n;
        for (u64 i = n; i > 1; i--) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; std::swap(perm[i - 1], perm[(st >> 33) % i]); }
        for (u64 i = 0; i < n; i++) f(perm[i]);
    } else { for (u64 i = 0; i < n; i++) f(i); }
#else
    const u64 blocks = (n + 255) / 256;
    if (s == 0) flush_fills();
    if (debug_launch()) debug_launch_note<F>(n, false);
    for (u64 b0 = 0; b0 < blocks; b0 += MAX_LAUNCH_BLOCKS) {
        rt_counters().launches++;
        hipLaunchKernelGGL(functor_kernel<F>, dim3((unsigned)std::min(MAX_LAUNCH_BLOCKS, blocks - b0)), dim3(256), 0, s, n, f, b0 * 256);
        AC_HIP_CHECK(hipGetLastError());
    }
    if (debug_launch()) debug_launch_note<F>(n, true);
#endif
}

// Same, but every lane of every launched wavefront runs the functor (with valid = false beyond n), so that the functor
// may use the wavefront-wide helpers below, which need all 64 lanes to arrive together.
#ifndef AC_EMU
template <class F>
__global__ void __launch_bounds__(256) functor_kernel_full(u64 n, F f, u64 base = 0) {
    u64 tid = base + (u64)blockIdx.x * 256 + threadIdx.x;
    f(tid, tid < n);
}
#endif
template <class F> void launch_full(u64 n, const F& f, stream_t s = 0) {
    if (n == 0) return;
#ifdef AC_EMU
    // whole wavefronts in lockstep (wave_rt.hpp): the functor's ballots and shuffles are the ones the device executes
    (void)s;
    const u64 blocks = (n + 255) / 256;
    const F* fp = &f;
    for (u64 b0 = 0; b0 < blocks; b0 += 0xFFFFFFu)
        wv::launch_kernel(+[](const F* g, u64 nn, u64 base) { const u64 tid = base + (u64)wv::bid() * 256 + wv::tid(); (*g)(tid, tid < nn); },
                          (unsigned)std::min<u64>(0xFFFFFFu, blocks - b0), 256u, fp, n, b0 * 256);
#else
    const u64 blocks = (n + 255) / 256;
    if (s == 0) flush_fills();
    if (debug_launch()) debug_launch_note<F>(n, false);
    for (u64 b0 = 0; b0 < blocks; b0 += MAX_LAUNCH_BLOCKS) {
        rt_counters().launches++;
        hipLaunchKernelGGL(functor_kernel_full<F>, dim3((unsigned)std::min(MAX_LAUNCH_BLOCKS, blocks - b0)), dim3(256), 0, s, n, f, b0 * 256);
        AC_HIP_CHECK(hipGetLastError());
    }
    if (debug_launch()) debug_launch_note<F>(n, true);
#endif
}
// A kernel of 256-thread workgroups that uses the wavefront primitives of wave_rt.hpp: hipLaunchKernelGGL on the device, the lockstep
// emulation under AC_EMU — the same kernel source either way.
template <class K, class... A> void launch_wave_kernel_sized(K kernel, u64 blocks, unsigned threads, stream_t s, A... args) {
    if (blocks == 0) return;
    if (blocks > 0xFFFFFFULL) throw DeviceError("grid too large");
#ifdef AC_EMU
    (void)s;
    wv::launch_kernel(kernel, (unsigned)blocks, threads, args...);
#else
    if (s == 0) flush_fills();
    rt_counters().launches++;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), 0, s, args...);
    AC_HIP_CHECK(hipGetLastError());
#endif
}
template <class K, class... A> void launch_wave_kernel(K kernel, u64 blocks, stream_t s, A... args) { launch_wave_kernel_sized(kernel, blocks, 256u, s, args...); }
// Bump allocation from a device counter with ONE atomic per wavefront (a counter hit by every lane serialises in L2).
// All 64 lanes must call it (amount may be 0).  Returns this lane's offset.
AC_D u32 wave_alloc32(u32* counter, u32 amount) {
    const int lane = wv::lane();
    u32 incl = amount;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 t = (u32)wv::shfl_up((int)incl, o); if (lane >= o) incl += t; }
    u32 total = (u32)wv::shfl((int)incl, 63);
    u32 base = 0;
    if (lane == 0 && total) base = atomic_add32(counter, total);
    base = (u32)wv::shfl((int)base, 0);
    return base + incl - amount;
}
// Adds the wavefront's total of `v` to a device counter with one atomic.  All 64 lanes must call it.
AC_D void wave_add64(u64* counter, u32 v) {
    u32 t = v;
#pragma unroll
    for (int o = 32; o; o >>= 1) t += (u32)wv::shfl_xor((int)t, o);
    if (wv::lane() == 0 && t) atomic_add64(counter, (u64)t);
}

}  // namespace ac

#include "device_prims.hpp"      // scan, radix sort, comparator sort, segmented reduce: hand-written (round 5)
