// What would locality buy the k-mer insert on diverse inputs?  The shipped table places a k-mer by a hash of its canonical middle, so
// the ~45 M claims of E'  are 45 M compare-and-swaps on random 128-byte lines: 26.7 G/s is this device's ceiling for
// that (ac_random_access_ceilings).  A minimiser-bucketed placement would put the R consecutive k-mers that share a minimiser into one
// line.  This probe measures the claim rate for R = 1 (random) ... 16 (a whole line per group), one claim per thread, consecutive
// threads = consecutive claims — the rate a locality-preserving table could be built on.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/locality_cas.hip -o /tmp/locality_cas && /tmp/locality_cas
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// claim i goes to line hash(i / R), slot (i % R) of that line's 16 (R <= 16): a group of R consecutive claims shares a line
template <int R>
__global__ void __launch_bounds__(256) claim_kernel(unsigned long long* table, uint64_t line_mask, uint64_t n, unsigned* lost) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool failed = false;
    if (i < n) {
        const uint64_t line = mix(i / R) & line_mask;
        const uint64_t slot = line * 16 + (mix(i / R + 0x9E3779B97F4A7C15ULL) + i % R) % 16;
        failed = atomicCAS(table + slot, ~0ULL, (unsigned long long)(i + 1)) != ~0ULL;
    }
    // (collisions between groups: counted per wavefront, not retried — the rate is what is measured)
    const unsigned long long b = __ballot(failed);
    if ((threadIdx.x & 63) == 0 && b) lost[1 + (i >> 6) % 65536] += (unsigned)__popcll(b);      // (no shared counter: one address takes ~0.1 G atomics/s)
}
template <int R> static int run(unsigned long long* table, uint64_t slots, uint64_t n, unsigned* lost, hipEvent_t a, hipEvent_t b) {
    std::vector<float> ms;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(table, 0xFF, slots * 8)); CK(hipMemset(lost, 0, 4 * 65537));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(claim_kernel<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, table, slots / 16 - 1, n, lost);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned> hl(65537); CK(hipMemcpy(hl.data(), lost, 4 * 65537, hipMemcpyDeviceToHost)); unsigned h_lost = 0; for (unsigned v : hl) h_lost += v;      // (racy adds: approximate)
    printf("{\"claims_per_line_group\": %d, \"claims\": %llu, \"table_slots\": %llu, \"ms\": %.3f, \"G_claims_per_s\": %.1f, \"lost_to_collisions\": %u}\n",
           R, (unsigned long long)n, (unsigned long long)slots, ms[2], n / ms[2] / 1e6, h_lost);
    return 0;
}
int main() {
    // E': 45.3 M claims; its table has 2^27 slots (1 GB, load 0.34).  The device's "random CAS ceiling" (ac_random_access_ceilings) is
    // measured on a 2^24-slot table (128 MB: inside the 256 MB Infinity Cache) — the sweep shows what a larger table does to it.
    CK(hipSetDevice(0));
    unsigned long long* table; unsigned* lost;
    const uint64_t max_slots = (uint64_t)1 << 30;
    CK(hipMalloc(&table, max_slots * 8)); CK(hipMalloc(&lost, 4 * 65537));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int lg = 24; lg <= 30; lg++) {
        const uint64_t slots = (uint64_t)1 << lg;
        const uint64_t n = std::min<uint64_t>(45300000, slots / 3);
        if (run<1>(table, slots, n, lost, a, b)) return 1;
        if (run<4>(table, slots, n, lost, a, b)) return 1;
        if (run<16>(table, slots, n, lost, a, b)) return 1;
    }
    return 0;
}
