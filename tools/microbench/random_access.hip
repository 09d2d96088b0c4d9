// Random-access ceilings of one MI355X for the two access patterns the graph-build kernels are bound by:
//   * claims:  one atomicCAS per item on a random slot of a 2^24-slot (134 MB) u64 table  (k-mer insert, first phase)
//   * lookups: one 8-byte read per item from a random slot of the same table             (degree kernel, absent k-mers)
// hipcc --offload-arch=gfx950 -O3 tools/microbench/random_access.hip -o gpurun_out/random_access && gpurun_out/random_access
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t x) {
    x ^= 0x9E3779B97F4A7C15ULL; x *= 0xff51afd7ed558ccdULL; x ^= x >> 32; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 29;
    return x;
}
__global__ void claim(unsigned long long* slots, uint64_t mask, uint64_t n, uint64_t salt) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t s = mix(i + salt) & mask;
    atomicCAS(&slots[s], ~0ULL, (unsigned long long)i);
}
__global__ void lookup(const unsigned long long* slots, uint64_t mask, uint64_t n, uint64_t salt, unsigned long long* sink) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long v = slots[mix(i + salt) & mask];
    if (v == 0x1234567ULL) *sink = v;      // keeps the load alive
}
__global__ void lookup6(const unsigned long long* slots, uint64_t mask, uint64_t n, uint64_t salt, unsigned long long* sink) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;      // six dependent-free lookups per thread, like one k-mer of the degree kernel
    if (i >= n) return;
    unsigned long long acc = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) acc ^= slots[mix(i * 6 + j + salt) & mask];
    if (acc == 0x1234567ULL) *sink = acc;
}

int main() {
    const uint64_t cap = 1ULL << 24;
    unsigned long long *slots, *sink;
    CHECK(hipMalloc(&slots, cap * 8)); CHECK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, int kind, uint64_t n, uint64_t per_thread) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipMemset(slots, 0xFF, cap * 8);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            unsigned blocks = (unsigned)((n + 255) / 256);
            if (kind == 0) hipLaunchKernelGGL(claim, dim3(blocks), dim3(256), 0, 0, slots, cap - 1, n, (uint64_t)rep * 77);
            else if (kind == 1) hipLaunchKernelGGL(lookup, dim3(blocks), dim3(256), 0, 0, slots, cap - 1, n, (uint64_t)rep * 77, sink);
            else hipLaunchKernelGGL(lookup6, dim3(blocks), dim3(256), 0, 0, slots, cap - 1, n, (uint64_t)rep * 77, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-34s n = %9llu  %.3f ms  %.1f G/s\n", name, (unsigned long long)(n * per_thread), best, n * per_thread / (best * 1e-3) / 1e9);
    };
    run("atomicCAS, random slot", 0, 5000000, 1);
    run("atomicCAS, random slot", 0, 7773239, 1);
    run("atomicCAS, random slot", 0, 50000000, 1);
    run("8-byte read, random slot", 1, 7773239, 1);
    run("8-byte read, random slot", 1, 46600000, 1);
    run("6 x 8-byte read per thread", 2, 7773239, 6);
    return 0;
}
