// What bounds the T_hot upload: the host's packing threads, the link, or the way the two are pipelined?
// Config C's text (487.5 MB, pageable, first touched by ONE thread like a caller's buffers) is packed to 2-bit codes by the library's
// own packer (ac::pack_text_host) into a pinned ring and copied in 16 MB pieces, the parts first alone and then together.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/upload_probe.hip -o /tmp/upload_probe -Lautocycler_amd -lautocycler_hip \
//         -Wl,-rpath,$PWD/autocycler_amd -pthread && /tmp/upload_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
namespace ac { void pack_text_host(const uint8_t* text, uint64_t n_text, uint64_t* bits, uint32_t* mask32, bool force_scalar); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
#include <functional>
#include <sched.h>
#include <string>
// run the calling thread on the cores of one NUMA node (what it allocates and first touches then lives there)
static bool pin_to_node(int node) {
    char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r"); if (!f) return false;
    char buf[512] = {0}; if (!fgets(buf, sizeof buf, f)) { fclose(f); return false; } fclose(f);
    cpu_set_t set; CPU_ZERO(&set);
    for (char* p = buf; *p && *p != '\n';) { long a = strtol(p, &p, 10), b = a; if (*p == '-') b = strtol(p + 1, &p, 10); for (long c = a; c <= b; c++) CPU_SET((int)c, &set); if (*p == ',') p++; }
    return sched_setaffinity(0, sizeof set, &set) == 0;
}
static void unpin() { cpu_set_t set; CPU_ZERO(&set); for (int c = 0; c < 1024 && c < CPU_SETSIZE; c++) CPU_SET(c, &set); sched_setaffinity(0, sizeof set, &set); }
// T threads that already exist when the clock starts (the library keeps its packing threads in a pool too)
struct Pool {
    std::vector<std::thread> th; std::atomic<int> gen{0}, left{0}; std::atomic<bool> quit{false}; std::function<void(int)> fn;
    explicit Pool(int T) { for (int i = 0; i < T; i++) th.emplace_back([this, i] { int seen = 0; for (;;) { while (gen.load(std::memory_order_acquire) == seen) { if (quit.load()) return; std::this_thread::yield(); } seen++; fn(i); left.fetch_sub(1, std::memory_order_acq_rel); } }); }
    void run(std::function<void(int)> f) { fn = std::move(f); left.store((int)th.size()); gen.fetch_add(1, std::memory_order_release); while (left.load(std::memory_order_acquire)) std::this_thread::yield(); }
    ~Pool() { quit.store(true); for (auto& t : th) t.join(); }
};

int main(int argc, char** argv) {
    const size_t N = (size_t)487500000 / 64 * 64;
    const size_t SUB = 1 << 20;
    const int REPS = 7;
    CK(hipSetDevice(0));
    uint8_t* text = (uint8_t*)malloc(N);
    { uint64_t x = 88172645463325252ULL; for (size_t i = 0; i < N; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; text[i] = "ACGT"[x & 3]; } }
    const size_t RING = (size_t)192 << 20;
    uint8_t* ring; CK(hipHostMalloc((void**)&ring, RING, hipHostMallocDefault));
    memset(ring, 0, RING);
    uint32_t* mask_sink_all = (uint32_t*)malloc(128 * (SUB / 8));      // per-thread scratch for the mask words (not sent)
    void* d; CK(hipMalloc(&d, N / 4 + 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("{\"hardware_concurrency\": %u, \"text_mb\": %.1f, \"AC_PACK_NT\": \"%s\"}\n", std::thread::hardware_concurrency(), N / 1e6, getenv("AC_PACK_NT") ? getenv("AC_PACK_NT") : "");

    // A. pack only: T threads, 1 MB work items, codes into the ring (wrapping), nothing copied
    for (int T : {16, 32, 64}) {
        std::vector<double> ts;
        Pool pool(T);
        for (int rep = 0; rep < REPS; rep++) {
            std::atomic<size_t> next{0};
            const double t0 = now();
            pool.run([&](int i) {
                uint32_t* msink = mask_sink_all + (size_t)i * (SUB / 32);
                for (size_t o; (o = next.fetch_add(SUB)) < N;) {
                    const size_t len = std::min(SUB, N - o);
                    ac::pack_text_host(text + o, len, (uint64_t*)(ring + (o / 4) % RING), msink, false);
                }
            });
            ts.push_back(now() - t0);
        }
        const double m = median(ts);
        printf("{\"part\": \"pack_only\", \"threads\": %d, \"ms\": %.3f, \"text_gb_s\": %.1f}\n", T, m * 1e3, N / m / 1e9);
    }
    // A2. read only (sum of 8-byte words): what the host's memory gives these threads
    for (int T : {16, 32, 64}) {
        std::vector<double> ts; std::atomic<uint64_t> sink{0};
        Pool pool(T);
        for (int rep = 0; rep < REPS; rep++) {
            std::atomic<size_t> next{0};
            const double t0 = now();
            pool.run([&](int) {
                uint64_t acc = 0;
                for (size_t o; (o = next.fetch_add(SUB)) < N;) { const uint64_t* p = (const uint64_t*)(text + o); const size_t w = std::min(SUB, N - o) / 8; for (size_t j = 0; j < w; j++) acc += p[j]; }
                sink += acc;
            });
            ts.push_back(now() - t0);
        }
        const double m = median(ts);
        printf("{\"part\": \"read_only\", \"threads\": %d, \"ms\": %.3f, \"text_gb_s\": %.1f}\n", T, m * 1e3, N / m / 1e9);
    }
    // B. copy only: the 122 MB of codes from the ring in pieces
    for (size_t mb : {4, 8, 16, 32}) {
        std::vector<double> ts;
        const size_t c = mb << 20, total = N / 4;
        for (int rep = 0; rep < REPS; rep++) {
            const double t0 = now();
            for (size_t o = 0; o < total; o += c) CK(hipMemcpyAsync((char*)d + o, ring + o % RING, std::min(c, total - o), hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            ts.push_back(now() - t0);
        }
        const double m = median(ts);
        printf("{\"part\": \"copy_only\", \"piece_mb\": %zu, \"ms\": %.3f, \"link_gb_s\": %.1f}\n", mb, m * 1e3, total / m / 1e9);
    }
    // C. the pipeline: whoever completes a chunk issues its copy (the library's scheme, ring large enough that nobody waits for a slot)
    for (size_t ch_mb : {64}) for (int T : {32}) for (int ramp : {0}) {
        // chunk boundaries in text bytes; with `ramp` the first chunks are small so that the link starts early
        std::vector<size_t> edge{0};
        { size_t c = ramp ? std::min<size_t>(8, ch_mb) << 20 : ch_mb << 20; while (edge.back() < N) { edge.push_back(std::min(N, edge.back() + c)); if (c < (ch_mb << 20)) c *= 2; } }
        const size_t n_chunks = edge.size() - 1;
        std::vector<double> ts;
        Pool pool(T);
        for (int rep = 0; rep < REPS; rep++) {
            std::vector<std::atomic<uint32_t>> done(n_chunks);
            for (auto& x : done) x.store(0);
            std::atomic<size_t> next{0};
            std::atomic_flag mu = ATOMIC_FLAG_INIT;
            // work items: (chunk, sub) in order
            std::vector<std::pair<uint32_t, uint32_t>> items;
            for (size_t c = 0; c < n_chunks; c++) for (size_t o = edge[c]; o < edge[c + 1]; o += SUB) items.push_back({(uint32_t)c, (uint32_t)((o - edge[c]) / SUB)});
            const double t0 = now();
            pool.run([&](int i) {
                uint32_t* msink = mask_sink_all + (size_t)i * (SUB / 32);
                for (size_t it; (it = next.fetch_add(1)) < items.size();) {
                    const size_t c = items[it].first, o = edge[c] + (size_t)items[it].second * SUB;
                    const size_t len = std::min(SUB, edge[c + 1] - o);
                    ac::pack_text_host(text + o, len, (uint64_t*)(ring + o / 4), msink, false);      // ring >= the whole 122 MB here
                    const uint32_t n_sub = (uint32_t)((edge[c + 1] - edge[c] + SUB - 1) / SUB);
                    if (done[c].fetch_add(1) + 1 == n_sub) {
                        while (mu.test_and_set(std::memory_order_acquire)) {}
                        CK(hipMemcpyAsync((char*)d + edge[c] / 4, ring + edge[c] / 4, (edge[c + 1] - edge[c]) / 4, hipMemcpyHostToDevice, s));
                        mu.clear(std::memory_order_release);
                    }
                }
            });
            CK(hipStreamSynchronize(s));
            ts.push_back(now() - t0);
        }
        const double m = median(ts);
        printf("{\"part\": \"pipeline\", \"chunk_text_mb\": %zu, \"threads\": %d, \"ramp\": %d, \"chunks\": %zu, \"ms\": %.3f, \"text_gb_s\": %.1f}\n", ch_mb, T, ramp, n_chunks, m * 1e3, N / m / 1e9);
    }
    // D. the copies while 32 threads stream the text through their cores (no stores): does the link suffer from the host's memory load?
    {
        Pool pool(33);
        std::vector<double> ts; std::atomic<uint64_t> sink{0};
        const size_t c = (size_t)16 << 20, total = N / 4;
        for (int rep = 0; rep < REPS; rep++) {
            std::atomic<bool> stop{false}; std::atomic<double> copy_s{0};
            pool.run([&](int i) {
                if (i == 0) {
                    const double t0 = now();
                    for (size_t o = 0; o < total; o += c) CK(hipMemcpyAsync((char*)d + o, ring + o % RING, std::min(c, total - o), hipMemcpyHostToDevice, s));
                    CK(hipStreamSynchronize(s));
                    copy_s.store(now() - t0); stop.store(true);
                } else {
                    uint64_t acc = 0; size_t o = (size_t)i * SUB;
                    while (!stop.load(std::memory_order_relaxed)) { const uint64_t* p = (const uint64_t*)(text + o % (N - SUB)); for (size_t j = 0; j < SUB / 8; j++) acc += p[j]; o += 32 * SUB; }
                    sink += acc;
                }
            });
            ts.push_back(copy_s.load());
        }
        const double m = median(ts);
        printf("{\"part\": \"copy_under_read_load\", \"readers\": 32, \"piece_mb\": 16, \"ms\": %.3f, \"link_gb_s\": %.1f}\n", m * 1e3, total / m / 1e9);
    }
    // E. what kind of pinned memory the ring is, and where the time of the pipeline goes: per variant the copies alone, the copies
    //    right after the packers wrote the ring (lines still in the cores' caches), and the pipeline with the time the last packer
    //    finished and the time the last copy was issued next to the time everything had landed
    struct Kind { const char* name; unsigned flags; int node; };
    const bool all_kinds = argc > 1 && std::string(argv[1]) == "kinds";      // r10m: every kind behaves like the default one
    std::vector<Kind> kinds = {{"default", hipHostMallocDefault, -1}};
    if (all_kinds) for (const Kind& kd : {Kind{"default_node0", hipHostMallocDefault, 0}, Kind{"default_node1", hipHostMallocDefault, 1}, Kind{"write_combined", hipHostMallocWriteCombined, -1},
                                          Kind{"non_coherent", hipHostMallocNonCoherent, -1}, Kind{"coherent", hipHostMallocCoherent, -1}}) kinds.push_back(kd);
    hipStream_t ss[4] = {s, nullptr, nullptr, nullptr};
    for (int i = 1; i < 4; i++) CK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
    for (const Kind& kd : kinds) {
        if (kd.node >= 0 && !pin_to_node(kd.node)) continue;
        uint8_t* r2 = nullptr;
        if (hipHostMalloc((void**)&r2, RING, kd.flags) != hipSuccess) { (void)hipGetLastError(); unpin(); printf("{\"part\": \"ring_kind\", \"kind\": \"%s\", \"error\": \"hipHostMalloc failed\"}\n", kd.name); continue; }
        memset(r2, 0, RING);
        unpin();
        const size_t c = (size_t)16 << 20, total = N / 4;
        auto copy_all = [&] { const double t0 = now(); for (size_t o = 0; o < total; o += c) CK(hipMemcpyAsync((char*)d + o, r2 + o, std::min(c, total - o), hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); return now() - t0; };
        { std::vector<double> ts; for (int rep = 0; rep < REPS; rep++) ts.push_back(copy_all());
          const double m = median(ts); printf("{\"part\": \"ring_kind_copy_only\", \"kind\": \"%s\", \"ms\": %.3f, \"link_gb_s\": %.1f}\n", kd.name, m * 1e3, total / m / 1e9); }
        for (int T : {32, 48}) {
            Pool pool(T);
            { std::vector<double> tp, tc;      // pack everything, THEN copy everything
              for (int rep = 0; rep < REPS; rep++) {
                std::atomic<size_t> next{0};
                const double t0 = now();
                pool.run([&](int i) { uint32_t* msink = mask_sink_all + (size_t)i * (SUB / 32); for (size_t o; (o = next.fetch_add(SUB)) < N;) ac::pack_text_host(text + o, std::min(SUB, N - o), (uint64_t*)(r2 + o / 4), msink, false); });
                tp.push_back(now() - t0);
                tc.push_back(copy_all());
              }
              printf("{\"part\": \"ring_kind_pack_then_copy\", \"kind\": \"%s\", \"threads\": %d, \"pack_ms\": %.3f, \"copy_ms\": %.3f}\n", kd.name, T, median(tp) * 1e3, median(tc) * 1e3); }
            for (size_t chk_mb : {8, 16, 32, 64}) for (int NSTR : {1, 2, 3}) {
                const size_t CHK = chk_mb << 20, n_chunks = (N + CHK - 1) / CHK;
                std::vector<double> ts, tpk, tis, tfirst;
                for (int rep = 0; rep < REPS; rep++) {
                    std::vector<std::atomic<uint32_t>> done(n_chunks);
                    for (auto& x : done) x.store(0);
                    std::atomic<size_t> next{0}; std::atomic_flag mu = ATOMIC_FLAG_INIT;
                    std::atomic<double> last_issue{0}, first_issue{1e9};
                    const double t0 = now();
                    pool.run([&](int i) {
                        uint32_t* msink = mask_sink_all + (size_t)i * (SUB / 32);
                        for (size_t o; (o = next.fetch_add(SUB)) < N;) {
                            const size_t ch = o / CHK, ce = std::min(N, (ch + 1) * CHK), len = std::min(SUB, N - o);
                            ac::pack_text_host(text + o, len, (uint64_t*)(r2 + o / 4), msink, false);
                            if (done[ch].fetch_add(1) + 1 == (uint32_t)((ce - ch * CHK + SUB - 1) / SUB)) {
                                while (mu.test_and_set(std::memory_order_acquire)) {}
                                CK(hipMemcpyAsync((char*)d + ch * CHK / 4, r2 + ch * CHK / 4, (ce - ch * CHK) / 4, hipMemcpyHostToDevice, ss[ch % (size_t)NSTR]));
                                const double t = now() - t0; last_issue.store(t); if (t < first_issue.load()) first_issue.store(t);
                                mu.clear(std::memory_order_release);
                            }
                        }
                    });
                    tpk.push_back(now() - t0);
                    for (int q = 0; q < NSTR; q++) CK(hipStreamSynchronize(ss[q]));
                    ts.push_back(now() - t0); tis.push_back(last_issue.load()); tfirst.push_back(first_issue.load());
                }
                printf("{\"part\": \"ring_kind_pipeline\", \"kind\": \"%s\", \"threads\": %d, \"chunk_text_mb\": %zu, \"copy_streams\": %d, \"ms\": %.3f, \"packers_done_ms\": %.3f, \"first_copy_issued_ms\": %.3f, \"last_copy_issued_ms\": %.3f}\n",
                       kd.name, T, chk_mb, NSTR, median(ts) * 1e3, median(tpk) * 1e3, median(tfirst) * 1e3, median(tis) * 1e3);
            }
        }
        CK(hipHostFree(r2));
    }
    return 0;
}
