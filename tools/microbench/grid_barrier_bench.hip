// What a grid-wide barrier costs on gfx950 (8 XCDs, one L2 each), by how it is built and how many workgroups take part.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/grid_barrier_bench.hip -o /tmp/grid_barrier_bench && /tmp/grid_barrier_bench
// Variants: 0 = agent-scope release / acquire fences around the counter (the cooperative-groups barrier: every pass writes back and
//               invalidates the XCD's L2),
//           1 = no fences: the counter alone (only right when every shared access is itself an agent-scope atomic),
//           2 = as 1, two-level: one counter per XCD-sized group of workgroups, their last arrival counts into the global one.
// Each round every thread also stores one word and (next round) reads a word another workgroup stored: `bad` counts stale reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline u32 ld(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VARIANT> __device__ void barrier(u32* ctl, u32 epoch, u32 nb) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (VARIANT == 0) __threadfence();
        if (VARIANT == 2) {
            const u32 grp = blockIdx.x / 32, ngrp = (nb + 31) / 32, in_grp = (grp + 1 == ngrp) ? nb - grp * 32 : 32;
            if (atomicAdd(ctl + 64 + grp * 16, 1u) + 1 == epoch * in_grp) atomicAdd(ctl, 1u);
            { const long long t0 = clock64(); while (ld(ctl) < epoch * ngrp) { __builtin_amdgcn_s_sleep(1); if (clock64() - t0 > (1LL << 30)) break; } }
        } else {
            atomicAdd(ctl, 1u);
            { const long long t0 = clock64(); while (ld(ctl) < epoch * nb) { __builtin_amdgcn_s_sleep(1); if (clock64() - t0 > (1LL << 30)) break; } }
        }
        if (VARIANT == 0) __threadfence();
    }
    __syncthreads();
}
template <int VARIANT> __global__ void __launch_bounds__(256) rounds_kernel(u32* ctl, u32* data, u32 nb, u32 rounds, u32* bad) {
    const u32 me = blockIdx.x * 256 + threadIdx.x, n = nb * 256;
    u32 wrong = 0;
    for (u32 r = 1; r <= rounds; r++) {
        if (VARIANT == 0) data[me] = r; else st(data + me, r);
        barrier<VARIANT>(ctl, r, nb);
        const u32 other = (me + 256 * 37 + 5) % n;      // a word of another workgroup (likely another XCD)
        const u32 v = VARIANT == 0 ? data[other] : ld(data + other);
        if (v != r && v != r + 1) wrong++;
    }
    if (wrong) atomicAdd(bad, wrong);
}
template <int VARIANT> void run(u32 nb, u32 rounds) {
    u32 *ctl, *data, *bad;
    CHECK(hipMalloc(&ctl, 4096 * 4)); CHECK(hipMalloc(&data, (size_t)nb * 256 * 4)); CHECK(hipMalloc(&bad, 4));
    float best = 1e30f; u32 hbad = 0;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipMemset(ctl, 0, 4096 * 4)); CHECK(hipMemset(data, 0, (size_t)nb * 256 * 4)); CHECK(hipMemset(bad, 0, 4));
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(rounds_kernel<VARIANT>, dim3(nb), dim3(256), 0, 0, ctl, data, nb, rounds, bad);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        u32 h = 0; CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); hbad += h;
    }
    printf("{\"variant\": %d, \"workgroups\": %u, \"rounds\": %u, \"us_per_barrier\": %.2f, \"stale_reads\": %u}\n", VARIANT, nb, rounds, best * 1000.0f / rounds, hbad);
    CHECK(hipFree(ctl)); CHECK(hipFree(data)); CHECK(hipFree(bad));
}
int main() {
    const u32 rounds = 200;
    for (u32 nb : {1u, 8u, 32u, 64u, 128u, 256u, 512u, 1024u}) { run<0>(nb, rounds); run<1>(nb, rounds); run<2>(nb, rounds); }
    return 0;
}
