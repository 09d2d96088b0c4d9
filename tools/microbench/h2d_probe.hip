// Host -> device upload probe for the T_hot bracket: what the GPU box's host can feed the device.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/h2d_probe.hip -o /tmp/h2d_probe -pthread && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)488 << 20;
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    CK(hipSetDevice(0));
    void* d; CK(hipMalloc(&d, N));
    char* pageable = (char*)malloc(N);
    memset(pageable, 'A', N);
    double t = now(); CK(hipMemcpy(d, pageable, N, hipMemcpyHostToDevice)); printf("pageable hipMemcpy 488 MB: %.1f ms (first)\n", (now() - t) * 1e3);
    t = now(); CK(hipMemcpy(d, pageable, N, hipMemcpyHostToDevice)); double dt = now() - t; printf("pageable hipMemcpy 488 MB: %.1f ms = %.1f GB/s\n", dt * 1e3, N / dt / 1e9);
    for (size_t mb : {16, 64, 128, 512}) { void* p; t = now(); CK(hipHostMalloc(&p, mb << 20, hipHostMallocDefault)); double a = now() - t; t = now(); CK(hipHostFree(p)); printf("hipHostMalloc %zu MB: %.1f ms (free %.1f ms)\n", mb, a * 1e3, (now() - t) * 1e3); }
    t = now(); CK(hipHostRegister(pageable, N, hipHostRegisterDefault)); printf("hipHostRegister 488 MB: %.1f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipMemcpy(d, pageable, N, hipMemcpyHostToDevice)); dt = now() - t; printf("registered hipMemcpy 488 MB: %.1f ms = %.1f GB/s\n", dt * 1e3, N / dt / 1e9);
    t = now(); CK(hipHostUnregister(pageable)); printf("hipHostUnregister: %.1f ms\n", (now() - t) * 1e3);
    char* pin; CK(hipHostMalloc((void**)&pin, N, hipHostMallocDefault));
    memset(pin, 'C', N);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t chunk_mb : {1, 4, 16, 64, 488}) {
        size_t c = chunk_mb << 20;
        t = now();
        for (size_t o = 0; o < N; o += c) CK(hipMemcpyAsync((char*)d + o, pin + o, std::min(c, N - o), hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        dt = now() - t; printf("pinned async H2D in %zu MB chunks: %.2f ms = %.1f GB/s\n", chunk_mb, dt * 1e3, N / dt / 1e9);
    }
    // two streams
    { hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); size_t c = 8 << 20; t = now(); int i = 0;
      for (size_t o = 0; o < N; o += c, i++) CK(hipMemcpyAsync((char*)d + o, pin + o, std::min(c, N - o), hipMemcpyHostToDevice, (i & 1) ? s2 : s));
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); dt = now() - t; printf("pinned async H2D, 2 streams, 8 MB chunks: %.2f ms = %.1f GB/s\n", dt * 1e3, N / dt / 1e9); }
    // multi-threaded host memcpy pageable -> pinned
    for (int T : {1, 2, 4, 8, 16, 32, 64}) {
        if (T > (int)std::thread::hardware_concurrency()) break;
        for (int rep = 0; rep < 2; rep++) {
            std::atomic<size_t> next{0}; const size_t c = 1 << 20;
            t = now();
            std::vector<std::thread> th;
            for (int i = 0; i < T; i++) th.emplace_back([&] { for (size_t o; (o = next.fetch_add(c)) < N;) memcpy(pin + o, pageable + o, std::min(c, N - o)); });
            for (auto& x : th) x.join();
            dt = now() - t;
            if (rep) printf("host memcpy pageable->pinned, %d threads: %.2f ms = %.1f GB/s\n", T, dt * 1e3, N / dt / 1e9);
        }
    }
    // multi-threaded scalar 2-bit pack pageable -> pinned (32 bytes -> 8 bytes + 4 mask bytes)
    for (int T : {8, 16, 32, 64}) {
        if (T > (int)std::thread::hardware_concurrency()) break;
        for (int rep = 0; rep < 2; rep++) {
            std::atomic<size_t> next{0}; const size_t c = 1 << 20;
            uint64_t* bits = (uint64_t*)pin; uint32_t* mask = (uint32_t*)(pin + N / 4 + 4096);
            t = now();
            std::vector<std::thread> th;
            for (int i = 0; i < T; i++) th.emplace_back([&] {
                for (size_t o; (o = next.fetch_add(c)) < N;) {
                    size_t e = std::min(o + c, N);
                    for (size_t g = o; g + 32 <= e; g += 32) {
                        uint64_t w = 0; uint32_t m = 0;
                        for (int j = 0; j < 32; j++) { unsigned ch = (unsigned char)pageable[g + j]; unsigned bad = !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); unsigned cc = bad ? 0u : (((ch >> 1) ^ (ch >> 2)) & 3u); w |= (uint64_t)cc << (62 - 2 * j); m |= bad << j; }
                        bits[g / 32] = w; mask[g / 32] = m;
                    }
                }
            });
            for (auto& x : th) x.join();
            dt = now() - t;
            if (rep) printf("host 2-bit pack pageable->pinned, %d threads: %.2f ms = %.1f GB/s of text\n", T, dt * 1e3, N / dt / 1e9);
        }
    }
    // D2H pinned
    t = now(); CK(hipMemcpyAsync(pin, d, 53 << 20, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); dt = now() - t; printf("pinned D2H 53 MB: %.2f ms = %.1f GB/s\n", dt * 1e3, (53 << 20) / dt / 1e9);
    return 0;
}
