// Are per-XCD copies of a counter array with workgroup-scope atomics (executed in the XCD's own L2) faster than one array with
// device-scope atomics (executed at the memory side)?  The pattern of the path walk's depth counters: N increments spread over
// U counters at random.  Also checks that the per-XCD copies add up exactly (every block only touches the copy of the XCD it
// runs on, read from HW_REG_XCC_ID, so one L2 is the only cache that ever holds a copy's lines).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/xcd_atomics.hip -o build/xcd_atomics && build/xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64; typedef unsigned u32;
__device__ inline u64 mix(u64 x) { x ^= 0x9E3779B97F4A7C15ULL; x *= 0xff51afd7ed558ccdULL; x ^= x >> 32; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 29; return x; }
__device__ inline u32 xcc_id() { u32 x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15u; }
__global__ void k_device(u32* c, u32 U, u64 n) { u64 i = (u64)blockIdx.x * 256 + threadIdx.x; if (i < n) atomicAdd(&c[mix(i) % U], 1u); }
__global__ void k_xcd(u32* c, u32 U, u64 n, u32 stride) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    u32* mine = c + (u64)xcc_id() * stride;
    if (i < n) __hip_atomic_fetch_add(&mine[mix(i) % U], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_xcd_min(u32* c, u32 U, u64 n, u32 stride) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    u32* mine = c + (u64)xcc_id() * stride;
    if (i < n) __hip_atomic_fetch_min(&mine[mix(i) % U], (u32)(mix(i * 7) >> 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_device_min(u32* c, u32 U, u64 n) { u64 i = (u64)blockIdx.x * 256 + threadIdx.x; if (i < n) atomicMin(&c[mix(i) % U], (u32)(mix(i * 7) >> 40)); }
__global__ void k_xcd_census(u32* out) { if (threadIdx.x == 0) atomicAdd(&out[xcc_id()], 1u); }
__global__ void k_sum(const u32* c, u32 U, u32 stride, u32* out) { u32 u = blockIdx.x * 256 + threadIdx.x; if (u < U) { u32 s = 0; for (int x = 0; x < 8; x++) s += c[(u64)x * stride + u]; out[u] = s; } }
__global__ void k_min8(const u32* c, u32 U, u32 stride, u32* out) { u32 u = blockIdx.x * 256 + threadIdx.x; if (u < U) { u32 s = ~0u; for (int x = 0; x < 8; x++) { u32 v = c[(u64)x * stride + u]; s = v < s ? v : s; } out[u] = s; } }
int main() {
    const u64 N = 10630307;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (u32 U : {158639u, 1732018u, 6550582u}) {
        const u32 stride = (U + 63) & ~63u;
        u32 *one, *eight, *sum, *cen; CK(hipMalloc(&one, (size_t)stride * 4)); CK(hipMalloc(&eight, (size_t)stride * 32)); CK(hipMalloc(&sum, (size_t)stride * 4)); CK(hipMalloc(&cen, 64));
        const unsigned blocks = (unsigned)((N + 255) / 256);
        float best_d = 1e9, best_x = 1e9, best_dm = 1e9, best_xm = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            float ms;
            CK(hipMemset(one, 0, (size_t)stride * 4)); CK(hipMemset(eight, 0, (size_t)stride * 32)); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); k_device<<<blocks, 256>>>(one, U, N); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_d) best_d = ms;
            CK(hipEventRecord(e0)); k_xcd<<<blocks, 256>>>(eight, U, N, stride); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_x) best_x = ms;
            if (rep == 3) {
                k_sum<<<(U + 255) / 256, 256>>>(eight, U, stride, sum); CK(hipDeviceSynchronize());
                std::vector<u32> a(U), b(U); CK(hipMemcpy(a.data(), one, (size_t)U * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), sum, (size_t)U * 4, hipMemcpyDeviceToHost));
                u64 bad = 0, tot = 0; for (u32 u = 0; u < U; u++) { bad += a[u] != b[u]; tot += b[u]; }
                printf("U=%u add: per-XCD copies summed vs device-scope: %llu counters differ, total %llu (expected %llu)\n", U, bad, tot, N);
            }
            CK(hipMemset(one, 0xFF, (size_t)stride * 4)); CK(hipMemset(eight, 0xFF, (size_t)stride * 32)); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); k_device_min<<<blocks, 256>>>(one, U, N); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_dm) best_dm = ms;
            CK(hipEventRecord(e0)); k_xcd_min<<<blocks, 256>>>(eight, U, N, stride); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_xm) best_xm = ms;
            if (rep == 3) {
                k_min8<<<(U + 255) / 256, 256>>>(eight, U, stride, sum); CK(hipDeviceSynchronize());
                std::vector<u32> a(U), b(U); CK(hipMemcpy(a.data(), one, (size_t)U * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), sum, (size_t)U * 4, hipMemcpyDeviceToHost));
                u64 bad = 0; for (u32 u = 0; u < U; u++) bad += a[u] != b[u];
                printf("U=%u min: per-XCD copies reduced vs device-scope: %llu counters differ\n", U, bad);
            }
        }
        printf("U=%u  N=%llu  atomicAdd device-scope %.3f ms (%.1f G/s) | workgroup-scope on the XCD's copy %.3f ms (%.1f G/s) | atomicMin %.3f ms vs %.3f ms\n",
               U, N, best_d, N / best_d / 1e6, best_x, N / best_x / 1e6, best_dm, best_xm);
        CK(hipMemset(cen, 0, 64)); k_xcd_census<<<4096, 256>>>(cen); CK(hipDeviceSynchronize());
        u32 h[16]; CK(hipMemcpy(h, cen, 64, hipMemcpyDeviceToHost)); printf("blocks per XCC id:"); for (int i = 0; i < 16; i++) printf(" %u", h[i]); printf("\n");
        CK(hipFree(one)); CK(hipFree(eight)); CK(hipFree(sum)); CK(hipFree(cen));
    }
    return 0;
}
