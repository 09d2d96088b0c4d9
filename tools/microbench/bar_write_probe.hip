// Can the host's packing threads write the 2-bit codes straight into device memory (through the PCIe BAR, write-combined stores)
// instead of into a pinned ring that a copy engine then reads?  DESIGN.md §6: packers alone 2.1 ms, copies alone 2.2 ms, together
// 3.2-3.4 ms — the copies slow down while the packers write host memory.  This probe checks whether device memory is host-writable
// at all here (in a child process: a fault must not take the probe down) and, if so, what 32-64 threads packing directly into it reach.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/bar_write_probe.hip -o /tmp/bar_write_probe -Lautocycler_amd -lautocycler_hip \
//         -Wl,-rpath,$PWD/autocycler_amd -pthread && /tmp/bar_write_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>
namespace ac { void pack_text_host(const uint8_t* text, uint64_t n_text, uint64_t* bits, uint32_t* mask32, bool force_scalar); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
struct Pool {
    std::vector<std::thread> th; std::atomic<int> gen{0}, left{0}; std::atomic<bool> quit{false}; std::function<void(int)> fn;
    explicit Pool(int T) { for (int i = 0; i < T; i++) th.emplace_back([this, i] { int seen = 0; for (;;) { while (gen.load(std::memory_order_acquire) == seen) { if (quit.load()) return; std::this_thread::yield(); } seen++; fn(i); left.fetch_sub(1, std::memory_order_acq_rel); } }); }
    void run(std::function<void(int)> f) { fn = std::move(f); left.store((int)th.size()); gen.fetch_add(1, std::memory_order_release); while (left.load(std::memory_order_acquire)) std::this_thread::yield(); }
    ~Pool() { quit.store(true); for (auto& t : th) t.join(); }
};
__global__ void sum_kernel(const unsigned long long* p, size_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    atomicAdd(out, acc);
}
static int variant(const char* name, int kind) {
    // each variant in a child of its own: a store to memory the host cannot reach is a SIGSEGV / SIGBUS
    fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) {
        const size_t N = (size_t)487500000 / 64 * 64, SUB = 1 << 20;
        if (hipSetDevice(0) != hipSuccess) _exit(3);
        void* d = nullptr;
        hipError_t e = kind == 0 ? hipMalloc(&d, N / 4 + 4096)
                     : kind == 1 ? hipExtMallocWithFlags(&d, N / 4 + 4096, hipDeviceMallocFinegrained)
                     : hipExtMallocWithFlags(&d, N / 4 + 4096, hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("{\"variant\": \"%s\", \"error\": \"alloc: %s\"}\n", name, hipGetErrorString(e)); fflush(stdout); _exit(0); }
        volatile uint64_t* w = (volatile uint64_t*)d;
        w[0] = 0x1122334455667788ULL; w[511] = 42;      // faults here if the host cannot reach it
        uint8_t* text = (uint8_t*)malloc(N);
        { uint64_t x = 88172645463325252ULL; for (size_t i = 0; i < N; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; text[i] = "ACGT"[x & 3]; } }
        uint32_t* msink = (uint32_t*)malloc(128 * (SUB / 8));
        uint64_t* ref = (uint64_t*)malloc(N / 4);
        ac::pack_text_host(text, N, ref, (uint32_t*)malloc(N / 8 + 64), false);
        unsigned long long want = 0; for (size_t i = 0; i < N / 32; i++) want += ref[i];
        unsigned long long* d_sum; if (hipMalloc(&d_sum, 8) != hipSuccess) _exit(3);
        for (int T : {16, 32, 64}) {
            Pool pool(T);
            std::vector<double> ts;
            for (int rep = 0; rep < 5; rep++) {
                std::atomic<size_t> next{0};
                const double t0 = now();
                pool.run([&](int i) {
                    uint32_t* ms = msink + (size_t)i * (SUB / 32);
                    for (size_t o; (o = next.fetch_add(SUB)) < N;) ac::pack_text_host(text + o, std::min(SUB, N - o), (uint64_t*)((uint8_t*)d + o / 4), ms, false);
                });
                __sync_synchronize();
                ts.push_back(now() - t0);
            }
            // the device sees what the host wrote
            (void)hipMemset(d_sum, 0, 8);
            hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned long long*)d, N / 32, d_sum);
            unsigned long long got = 0; (void)hipMemcpy(&got, d_sum, 8, hipMemcpyDeviceToHost);
            const double m = median(ts);
            printf("{\"variant\": \"%s\", \"threads\": %d, \"ms\": %.3f, \"text_gb_s\": %.1f, \"codes_gb_s\": %.1f, \"device_sees_it\": %s}\n", name, T, m * 1e3, N / m / 1e9, N / 4 / m / 1e9,
                   got == want ? "true" : "false");
            fflush(stdout);
        }
        _exit(0);
    }
    int st = 0; waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("{\"variant\": \"%s\", \"error\": \"the host cannot write this memory (signal %d)\"}\n", name, WTERMSIG(st));
    return 0;
}
int main() {
    variant("hipMalloc", 0);
    variant("hipExtMallocWithFlags(Finegrained)", 1);
    variant("hipExtMallocWithFlags(Uncached)", 2);
    return 0;
}
