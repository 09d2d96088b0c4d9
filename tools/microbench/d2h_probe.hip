// Device -> host probe (round 6): what the link gives on the way OUT, by transfer size, number of copy streams, and by who moves the bytes
// (copy engine / a kernel storing into mapped pinned memory).  configs[4] moves 4.8 GB of path entries there; the build measured 29-33 GB/s.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/d2h_probe.hip -o /tmp/d2h_probe && /tmp/d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void store_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t N = (size_t)2 << 30;
    CK(hipSetDevice(0));
    void* d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 7, N));
    char* pin; CK(hipHostMalloc((void**)&pin, N, hipHostMallocDefault));
    memset(pin, 1, N);      // (touched: no first-touch faults inside the timings)
    hipStream_t s[4];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++)
    for (size_t chunk_mb : {16, 64, 256, 2048}) {
        for (int ns : {1, 2, 4}) {
            const size_t c = chunk_mb << 20;
            double t = now(); int i = 0;
            for (size_t o = 0; o < N; o += c, i++) CK(hipMemcpyAsync(pin + o, (char*)d + o, std::min(c, N - o), hipMemcpyDeviceToHost, s[i % ns]));
            for (int q = 0; q < ns; q++) CK(hipStreamSynchronize(s[q]));
            double dt = now() - t;
            if (rep) printf("{\"what\": \"copy engine D2H 2 GB\", \"chunk_MB\": %zu, \"streams\": %d, \"ms\": %.2f, \"GBps\": %.1f}\n", chunk_mb, ns, dt * 1e3, N / dt / 1e9);
        }
    }
    {   // H2D for comparison
        double t = now(); CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s[0])); CK(hipStreamSynchronize(s[0])); double dt = now() - t;
        printf("{\"what\": \"copy engine H2D 2 GB\", \"ms\": %.2f, \"GBps\": %.1f}\n", dt * 1e3, N / dt / 1e9);
    }
    for (int blocks : {256, 1024, 4096}) {   // a kernel storing into the mapped pinned buffer
        void* dp = nullptr; CK(hipHostGetDevicePointer(&dp, pin, 0));
        hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(256), 0, s[0], (const uint4*)d, (uint4*)dp, N / 16); CK(hipStreamSynchronize(s[0]));
        double t = now();
        hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(256), 0, s[0], (const uint4*)d, (uint4*)dp, N / 16); CK(hipStreamSynchronize(s[0]));
        double dt = now() - t;
        printf("{\"what\": \"kernel stores into mapped pinned memory, 2 GB\", \"blocks\": %d, \"ms\": %.2f, \"GBps\": %.1f}\n", blocks, dt * 1e3, N / dt / 1e9);
    }
    return 0;
}
