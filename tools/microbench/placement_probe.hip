// What would a minimizer-bucketed placement do to the k-mer insert on a REAL mixed-species text?  Every position
// of the text (bytes: A C G T, anything else separates) is inserted into an open-addressing table of 8-byte slots — identity = a
// 64-bit hash of the canonical k-mer, claim by compare-and-swap, linear probing — with two home functions:
//   random   : home slot = hash(canonical k-mer)                              (what a hash of the k-mer, or of its middle, gives)
//   bucketed : home bucket (2^b slots, b = 4: one 128-byte line) = hash(minimizer: the smallest hashed canonical m-mer of the k-mer),
//              slot in the bucket = the k-mer's own hash
// and the probe reports time, claims, probe lengths (how unevenly the buckets fill) for both.  No run following: every position goes to
// the table, so the absolute times are not the product's; the ratio between the two placements on the same stream is what is measured.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe text.bin 51 25 27
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ int code_of(unsigned char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
static const int TILE = 256, MAXK = 128;
// stats: [0] claims, [1] duplicates, [2] probes in total, [3] positions with > 16 probes, [4] longest probe, [5] valid positions
template <int MODE>
__global__ void __launch_bounds__(256) insert_kernel(const unsigned char* text, u64 n, int k, int m, int bbits, u64* table, u64 slot_mask, u64* stats) {
    __shared__ unsigned char codes[TILE + MAXK];
    __shared__ u64 mh[TILE + MAXK];      // hash of the canonical m-mer starting at each tile position (~0 if it holds a separator)
    const u64 base = (u64)blockIdx.x * TILE;
    for (int i = threadIdx.x; i < TILE + k; i += 256) codes[i] = base + i < n ? (unsigned char)code_of(text[base + i]) : 4;
    __syncthreads();
    if (MODE == 1)
        for (int i = threadIdx.x; i < TILE + k - m; i += 256) {
            u64 f = 0, r = 0; bool bad = false;
            for (int j = 0; j < m; j++) { const int c = codes[i + j]; bad |= c > 3; f = f * 4 + (u64)(c & 3); r = r * 4 + (u64)(3 - (codes[i + m - 1 - j] & 3)); }
            mh[i] = bad ? ~0ULL : mix(f < r ? f : r);
        }
    __syncthreads();
    const int t = threadIdx.x;
    const u64 p = base + t;
    u64 probes = 0; bool claimed = false, dup = false, valid = p + (u64)k <= n;
    if (valid) {
        u64 hf = 0, hr = 0;
        for (int j = 0; j < k; j++) {
            const int c = codes[t + j]; valid &= c <= 3;
            hf = hf * 0x9E3779B97F4A7C15ULL + (u64)(c & 3) + 1;
            hr = hr * 0x9E3779B97F4A7C15ULL + (u64)(3 - (codes[t + k - 1 - j] & 3)) + 1;
        }
        if (valid) {
            const u64 id = mix(hf < hr ? hf : hr) | 1ULL;      // identity of the canonical k-mer (never 0 = empty)
            u64 home;
            if (MODE == 0) home = mix(id ^ 0x5851F42D4C957F2DULL);
            else {
                u64 best = ~0ULL;
                for (int j = 0; j + m <= k; j++) best = mh[t + j] < best ? mh[t + j] : best;
                home = (mix(best) << bbits) | (id & ((1ULL << bbits) - 1));      // a bucket of 2^bbits slots per minimizer, the k-mer's own hash inside it
            }
            u64 s = home & slot_mask;
            for (;;) {
                probes++;
                u64 v = __builtin_nontemporal_load(table + s);
                if (v == 0) { v = atomicCAS(table + s, 0ULL, id); if (v == 0) { claimed = true; break; } }
                if (v == id) { dup = true; break; }
                s = (s + 1) & slot_mask;
                if (probes > 100000) break;
            }
        }
    }
    // per-wavefront totals (no shared counters in the hot loop)
    u64 c = claimed, d = dup, pr = probes, lg = probes > 16, vl = valid && (claimed || dup);
    for (int o = 32; o; o >>= 1) { c += __shfl_xor((unsigned long long)c, o); d += __shfl_xor((unsigned long long)d, o); pr += __shfl_xor((unsigned long long)pr, o); lg += __shfl_xor((unsigned long long)lg, o); vl += __shfl_xor((unsigned long long)vl, o); }
    u64 mx = probes; for (int o = 32; o; o >>= 1) { u64 x = __shfl_xor((unsigned long long)mx, o); mx = x > mx ? x : mx; }
    if ((t & 63) == 0) {
        u64* st = stats + 8 * (((u64)blockIdx.x * 4 + (t >> 6)) % 4096);
        st[0] += c; st[1] += d; st[2] += pr; st[3] += lg; st[5] += vl; if (mx > st[4]) st[4] = mx;      // (racy between wavefronts sharing a row: approximate)
    }
}
int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: placement_probe TEXT K [M...]\n"); return 1; }
    const int k = atoi(argv[2]);
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const u64 n = text.size();
    if (!n || k < 3 || k >= MAXK) { printf("{\"error\": \"bad input\"}\n"); return 1; }
    CK(hipSetDevice(0));
    unsigned char* d_text; CK(hipMalloc(&d_text, n)); CK(hipMemcpy(d_text, text.data(), n, hipMemcpyHostToDevice));
    for (int lg : {27}) {
        const u64 slots = 1ULL << lg;
        u64* table; CK(hipMalloc(&table, slots * 8));
        u64* stats; CK(hipMalloc(&stats, 4096 * 8 * 8));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto run = [&](int mode, int m, int bbits) -> int {
            std::vector<float> ms; std::vector<u64> h(4096 * 8);
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(table, 0, slots * 8)); CK(hipMemset(stats, 0, 4096 * 8 * 8)); CK(hipDeviceSynchronize());
                CK(hipEventRecord(a, 0));
                const unsigned blocks = (unsigned)((n + TILE - 1) / TILE);
                if (mode == 0) hipLaunchKernelGGL(insert_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_text, n, k, m, bbits, table, slots - 1, stats);
                else hipLaunchKernelGGL(insert_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_text, n, k, m, bbits, table, slots - 1, stats);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
            }
            CK(hipMemcpy(h.data(), stats, 4096 * 8 * 8, hipMemcpyDeviceToHost));
            u64 s[6] = {0, 0, 0, 0, 0, 0};
            for (int r = 0; r < 4096; r++) { for (int q : {0, 1, 2, 3, 5}) s[q] += h[r * 8 + q]; s[4] = std::max(s[4], h[r * 8 + 4]); }
            std::sort(ms.begin(), ms.end());
            printf("{\"table_slots\": %llu, \"placement\": \"%s\", \"m\": %d, \"bucket_slots\": %d, \"ms\": %.3f, \"claims\": %llu, \"duplicates\": %llu, \"probes_per_position\": %.2f, "
                   "\"positions_over_16_probes\": %llu, \"longest_probe\": %llu, \"load\": %.2f}\n", slots, mode ? "bucketed" : "random", mode ? m : 0, mode ? 1 << bbits : 0, ms[1], s[0], s[1],
                   s[5] ? (double)s[2] / s[5] : 0.0, s[3], s[4], (double)s[0] / slots);
            fflush(stdout);
            return 0;
        };
        if (run(0, 0, 0)) return 1;
        for (int i = 3; i < argc; i++) for (int bb : {4, 6, 8, 10}) if (run(1, atoi(argv[i]), bb)) return 1;
        CK(hipFree(table)); CK(hipFree(stats));
    }
    return 0;
}
