// How fast do T host threads fill a result block — hipHostMalloc'd (what PinnedPool hands out) against plain malloc'd memory — with the
// access pattern of path_stretch_range: sequential 4-byte stores of values read from a table at ~10-entry runs?  (round 6: configs[4]'s
// host job writes 4.8 GB at 37 GB/s whatever its thread count.)
//   hipcc -O3 -std=c++17 tools/microbench/host_write_probe.hip -o /tmp/host_write_probe -pthread && /tmp/host_write_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t n_ent = (size_t)(argc > 1 ? atof(argv[1]) : 1.2e9), U = 82u << 20;
    std::vector<int> thread_counts = {12, 24, 32};
    const size_t n_runs = n_ent / 10;
    std::vector<uint32_t> starts(n_runs);      // table row of every run (read front to back, like rec_val)
    { uint64_t x = 88172645463325252ULL; for (size_t i = 0; i < n_runs; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; starts[i] = (uint32_t)(x % (U - 16)); } }
    for (int kind = 0; kind < 2; kind++) {      // 0: pinned default, 1: malloc, 2: pinned non-coherent
        int32_t* out = nullptr; uint32_t* table = nullptr;
        if (kind == 1) { out = (int32_t*)aligned_alloc(4096, n_ent * 4); table = (uint32_t*)aligned_alloc(4096, (size_t)U * 4); }
        else {
            unsigned flags = kind == 0 ? hipHostMallocDefault : hipHostMallocNonCoherent;
            if (hipHostMalloc((void**)&out, n_ent * 4, flags) != hipSuccess || hipHostMalloc((void**)&table, (size_t)U * 4, flags) != hipSuccess) { printf("{\"kind\": %d, \"error\": \"alloc\"}\n", kind); continue; }
        }
        memset(out, 1, n_ent * 4);      // (touched: no first-touch faults in the timed part)
        for (size_t i = 0; i < U; i++) table[i] = (uint32_t)i * 2654435761u;
        for (int T : thread_counts) {
            for (int mode = 0; mode < 5; mode++) {      // 0: memset-like fill, 1: runs of 10 copied from random table rows, 2-4: the same with the rows of the runs 8 / 16 / 32 ahead prefetched
                const size_t ahead = mode == 2 ? 8 : mode == 3 ? 16 : mode == 4 ? 32 : 0;
                std::atomic<size_t> next{0};
                const size_t BLOCK = 40960;
                const double t0 = now_s();
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                                        for (size_t b; (b = next.fetch_add(BLOCK)) < n_ent;) {
                        const size_t e = std::min(b + BLOCK, n_ent);
                        if (mode == 0) { for (size_t i = b; i < e; i++) out[i] = (int32_t)i; }
                        else for (size_t r = b / 10, r1 = e / 10; r < r1; r++) {
                            if (ahead && r + ahead < r1) __builtin_prefetch(table + starts[r + ahead]);
                            const uint32_t* src = table + starts[r];
                            int32_t* o = out + r * 10;
                            for (size_t k = 0; k < 10; k++) o[k] = (int32_t)src[k];
                        }
                    }
                });
                for (auto& t : th) t.join();
                const double dt = now_s() - t0;
                printf("{\"memory\": \"%s\", \"threads\": %d, \"pattern\": \"%s\", \"GBps\": %.1f, \"ms\": %.1f}\n", kind == 0 ? "hipHostMalloc default" : kind == 1 ? "malloc" : "hipHostMalloc non-coherent",
                       T, mode == 0 ? "sequential fill" : mode == 1 ? "runs of 10 from random table rows" : mode == 2 ? "... rows 8 runs ahead prefetched" : mode == 3 ? "... 16 ahead" : "... 32 ahead", n_ent * 4 / dt / 1e9, dt * 1e3);
                fflush(stdout);
            }
        }
        if (kind == 1) { free(out); free(table); } else { (void)hipHostFree(out); (void)hipHostFree(table); }
    }
    return 0;
}
