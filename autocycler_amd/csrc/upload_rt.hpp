// The host entry's runtime objects, shared by graph_upload.hip (which drives them) and graph_build.hip (warm-up, upload timing).
#pragma once
#include "graph_impl.hpp"

namespace ac {

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
    static UploadPool& second() { return ctx_object<UploadPool>(CTX_POOL2); }      // a few threads for a job that runs beside one of the first pool's (SeqExpandJob)
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
void ensure_host_stager();      // (device_warmup: the whole-command path uploads the text as BYTES for the end repair — through the ring)

}  // namespace ac
