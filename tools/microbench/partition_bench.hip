// What would north_star's "bucket into LDS via per-wavefront radix partitioning" cost on an MI355X?  ("What's missing" 2.)
// The pieces of a partitioned k-mer insert for k = 51 (two-word keys), measured one by one on a random text the size of E':
//   A  extract        per position: canonical k-mer (2 words) + the hash of its canonical middle (= the table's home slot) -> a checksum.
//                     What ANY design pays before it moves a byte (the shipped insert pays it too).
//   B  count          A + an LDS histogram of the top 8 hash bits per workgroup share (exact offsets for the scatter: no global atomics).
//   C  scatter        A + (key, position) records of 24 bytes written to 256 buckets at those offsets — the tile's records ranked by
//                     bucket in LDS first, so that a bucket's records of a tile leave as one run.  SURVEY.md 8(d)'s record traffic, once.
//   D  re-partition   one more level: every bucket's records read back and scattered into 128 sub-buckets (a table range that fits
//                     LDS needs 2^15 ranges for E', 2^20 for configs[4]: two levels of fan-out).
//   E  bucket pass    the records of a sub-bucket streamed through an LDS table (claim / compare full keys / keep the smallest position),
//                     the table range written out — approximated here by its memory traffic only: read 24 B per record, write 8 B per slot.
// Prints the time of each and the sum a two-level partitioned insert would need, next to the shipped insert's rate on E' (DESIGN.md section 4).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I autocycler_amd/csrc tools/microbench/partition_bench.hip -o build/partition_bench && build/partition_bench [positions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kmer_ops.hpp"
using namespace ac;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static const int K = 51, W = 2, NB = 256, PER = 8, TILE = 256 * PER;

__device__ inline void kmer_at(const u64* bits, u64 p, Key<W>* uk, u64* home) {
    Key<W> fwd = text_extract<W>(bits, p, K);
    Key<W> rc = key_rc<W>(fwd, K);
    Key<W> k = key_lt<W>(rc, fwd) ? rc : fwd;
    k.w[0] |= (u64)255 << 56;
    *uk = k;
    *home = key_home<W>(k, K, false, key_hash<W>(k));
}
// every workgroup owns the positions [wg * share, (wg + 1) * share)
__global__ void __launch_bounds__(256) k_extract(const u64* bits, u64 n, u64 share, u64* sink) {
    const u64 b0 = (u64)blockIdx.x * share, b1 = b0 + share < n ? b0 + share : n;
    u64 acc = 0;
    for (u64 t = b0; t < b1; t += TILE)
        for (int j = 0; j < PER; j++) {
            const u64 p = t + (u64)j * 256 + threadIdx.x;
            if (p >= b1) continue;
            Key<W> uk; u64 h; kmer_at(bits, p, &uk, &h);
            acc ^= h + uk.w[1];
        }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_count(const u64* bits, u64 n, u64 share, u32* hist /* [NB][grid] */) {
    __shared__ u32 h[NB];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u64 b0 = (u64)blockIdx.x * share, b1 = b0 + share < n ? b0 + share : n;
    for (u64 t = b0; t < b1; t += TILE)
        for (int j = 0; j < PER; j++) {
            const u64 p = t + (u64)j * 256 + threadIdx.x;
            if (p >= b1) continue;
            Key<W> uk; u64 hh; kmer_at(bits, p, &uk, &hh);
            atomicAdd(&h[(hh >> 32) & (NB - 1)], 1u);
        }
    __syncthreads();
    hist[(u64)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}
struct Rec { u64 k0, k1, pos; };
// a tile's records are ranked by bucket in LDS (count, scan, place), then leave in bucket order: consecutive lanes write consecutive records
__global__ void __launch_bounds__(256) k_scatter(const u64* bits, u64 n, u64 share, const u64* offs /* [NB][grid] */, Rec* out) {
    __shared__ u32 cnt[NB], start[NB], cursor[NB];
    __shared__ u64 base[NB];
    __shared__ Rec stage[TILE];
    __shared__ u16 sbucket[TILE];
    const u64 b0 = (u64)blockIdx.x * share, b1 = b0 + share < n ? b0 + share : n;
    base[threadIdx.x] = offs[(u64)threadIdx.x * gridDim.x + blockIdx.x];
    for (u64 t = b0; t < b1; t += TILE) {
        cnt[threadIdx.x] = 0;
        __syncthreads();
        Rec r[PER]; u32 bk[PER], rk[PER];
        for (int j = 0; j < PER; j++) {
            const u64 p = t + (u64)j * 256 + threadIdx.x;
            bk[j] = 0xFFFFFFFFu;
            if (p >= b1) continue;
            Key<W> uk; u64 hh; kmer_at(bits, p, &uk, &hh);
            r[j] = Rec{uk.w[0], uk.w[1], p};
            bk[j] = (u32)(hh >> 32) & (NB - 1);
            rk[j] = atomicAdd(&cnt[bk[j]], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) { u32 a = 0; for (int b = 0; b < NB; b++) { start[b] = a; a += cnt[b]; } }      // (256 adds: negligible next to the tile)
        __syncthreads();
        for (int j = 0; j < PER; j++) if (bk[j] != 0xFFFFFFFFu) { const u32 s = start[bk[j]] + rk[j]; stage[s] = r[j]; sbucket[s] = (u16)bk[j]; }
        cursor[threadIdx.x] = 0;
        __syncthreads();
        const u32 total = start[NB - 1] + cnt[NB - 1];
        for (u32 s = threadIdx.x; s < total; s += 256) { const u32 b = sbucket[s]; out[base[b] + (s - start[b])] = stage[s]; }
        __syncthreads();
        base[threadIdx.x] += cnt[threadIdx.x];
        __syncthreads();
    }
}
// level 2: a workgroup per (bucket, slice): reads records, scatters by the next 7 hash bits into the bucket's own range (counts exact per slice)
__global__ void __launch_bounds__(256) k_recount(const Rec* in, const u64* bstart, u32 slices, u32* hist2 /* [NB][128][slices] */) {
    __shared__ u32 h[128];
    if (threadIdx.x < 128) h[threadIdx.x] = 0;
    __syncthreads();
    const u32 b = blockIdx.x / slices, sl = blockIdx.x % slices;
    const u64 lo = bstart[b], hi = bstart[b + 1], len = (hi - lo + slices - 1) / slices;
    const u64 a = lo + (u64)sl * len, e = a + len < hi ? a + len : hi;
    for (u64 i = a + threadIdx.x; i < e; i += 256) {
        const Rec r = in[i];
        Key<W> uk; uk.w[0] = r.k0; uk.w[1] = r.k1;
        const u64 hh = key_home<W>(uk, K, false, key_hash<W>(uk));
        atomicAdd(&h[(hh >> 40) & 127], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 128) hist2[((u64)b * 128 + threadIdx.x) * slices + sl] = h[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_rescatter(const Rec* in, const u64* bstart, u32 slices, const u64* offs2, Rec* out) {
    __shared__ u32 cur[128];
    __shared__ u64 base[128];
    const u32 b = blockIdx.x / slices, sl = blockIdx.x % slices;
    if (threadIdx.x < 128) { cur[threadIdx.x] = 0; base[threadIdx.x] = offs2[((u64)b * 128 + threadIdx.x) * slices + sl]; }
    __syncthreads();
    const u64 lo = bstart[b], hi = bstart[b + 1], len = (hi - lo + slices - 1) / slices;
    const u64 a = lo + (u64)sl * len, e = a + len < hi ? a + len : hi;
    for (u64 i = a + threadIdx.x; i < e; i += 256) {
        const Rec r = in[i];
        Key<W> uk; uk.w[0] = r.k0; uk.w[1] = r.k1;
        const u64 hh = key_home<W>(uk, K, false, key_hash<W>(uk));
        const u32 sb = (u32)(hh >> 40) & 127;
        out[base[sb] + atomicAdd(&cur[sb], 1u)] = r;      // (a slice's records of one sub-bucket end up consecutive: the L2 combines the 24-byte stores)
    }
}
// the bucket pass's traffic: every record read once, one 8-byte slot written per 0.35 records' worth of table (load 0.35)
__global__ void __launch_bounds__(256) k_bucket_traffic(const Rec* in, u64 n_rec, u64* slots, u64 n_slots, u64* sink) {
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_rec; i += (u64)gridDim.x * 256) { const Rec r = in[i]; acc ^= r.k0 + r.k1 + r.pos; }
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_slots; i += (u64)gridDim.x * 256) slots[i] = acc | i;
    if (acc == 0x1234567) sink[0] = acc;
}

template <class F> static float timed(F&& f, int reps = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const u64 n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 101440066ULL;      // E' = 101.4 M positions
    const u64 words = n / 32 + 8;
    std::vector<u64> h(words);
    u64 x = 88172645463325252ULL;
    for (auto& w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = x; }
    u64 *bits, *sink; CK(hipMalloc(&bits, words * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(bits, h.data(), words * 8, hipMemcpyHostToDevice));
    const u64 n_pos = n - K;
    const unsigned grid = 2048;                                                  // 8 workgroups per CU
    const u64 share = ((n_pos + grid - 1) / grid + TILE - 1) / TILE * TILE;
    u32* hist; CK(hipMalloc(&hist, (size_t)NB * grid * 4));
    u64* offs; CK(hipMalloc(&offs, (size_t)NB * grid * 8));
    Rec *rec1, *rec2; CK(hipMalloc(&rec1, n_pos * sizeof(Rec))); CK(hipMalloc(&rec2, n_pos * sizeof(Rec)));
    const float tA = timed([&] { k_extract<<<grid, 256>>>(bits, n_pos, share, sink); });
    const float tB = timed([&] { k_count<<<grid, 256>>>(bits, n_pos, share, hist); });
    std::vector<u32> hh((size_t)NB * grid);
    CK(hipMemcpy(hh.data(), hist, hh.size() * 4, hipMemcpyDeviceToHost));
    std::vector<u64> ho(hh.size()), bstart(NB + 1);
    u64 acc = 0;
    for (size_t i = 0; i < hh.size(); i++) { if (i % grid == 0) bstart[i / grid] = acc; ho[i] = acc; acc += hh[i]; }
    bstart[NB] = acc;
    if (acc != n_pos) { printf("count mismatch %llu vs %llu\n", (unsigned long long)acc, (unsigned long long)n_pos); return 1; }
    CK(hipMemcpy(offs, ho.data(), ho.size() * 8, hipMemcpyHostToDevice));
    const float tC = timed([&] { k_scatter<<<grid, 256>>>(bits, n_pos, share, offs, rec1); });
    // level 2
    const u32 slices = 16;
    u64* d_bstart; CK(hipMalloc(&d_bstart, (NB + 1) * 8)); CK(hipMemcpy(d_bstart, bstart.data(), (NB + 1) * 8, hipMemcpyHostToDevice));
    u32* hist2; CK(hipMalloc(&hist2, (size_t)NB * 128 * slices * 4));
    u64* offs2; CK(hipMalloc(&offs2, (size_t)NB * 128 * slices * 8));
    const float tD0 = timed([&] { k_recount<<<NB * slices, 256>>>(rec1, d_bstart, slices, hist2); });
    std::vector<u32> h2((size_t)NB * 128 * slices);
    CK(hipMemcpy(h2.data(), hist2, h2.size() * 4, hipMemcpyDeviceToHost));
    std::vector<u64> o2(h2.size());
    acc = 0;
    for (size_t i = 0; i < h2.size(); i++) { o2[i] = acc; acc += h2[i]; }
    if (acc != n_pos) { printf("recount mismatch\n"); return 1; }
    CK(hipMemcpy(offs2, o2.data(), o2.size() * 8, hipMemcpyHostToDevice));
    const float tD1 = timed([&] { k_rescatter<<<NB * slices, 256>>>(rec1, d_bstart, slices, offs2, rec2); });
    const u64 n_slots = (u64)1 << 27;                                            // E': 134 M slots
    u64* slots; CK(hipMalloc(&slots, n_slots * 8));
    const float tE = timed([&] { k_bucket_traffic<<<4096, 256>>>(rec2, n_pos, slots, n_slots, sink); });
    const double GB = 1e-9;
    printf("{\"positions\": %llu, \"k\": %d, \"record_bytes\": %zu,\n", (unsigned long long)n_pos, K, sizeof(Rec));
    printf(" \"A_extract_ms\": %.3f, \"B_count_ms\": %.3f, \"C_scatter_ms\": %.3f, \"C_scatter_GBps_written\": %.0f,\n", tA, tB, tC, n_pos * sizeof(Rec) * GB / (tC * 1e-3));
    printf(" \"D_recount_ms\": %.3f, \"D_rescatter_ms\": %.3f, \"D_rescatter_GBps_moved\": %.0f,\n", tD0, tD1, 2.0 * n_pos * sizeof(Rec) * GB / (tD1 * 1e-3));
    printf(" \"E_bucket_traffic_ms\": %.3f, \"E_GBps\": %.0f,\n", tE, (n_pos * sizeof(Rec) + n_slots * 8.0) * GB / (tE * 1e-3));
    printf(" \"one_level_sum_ms\": %.3f, \"two_level_sum_ms\": %.3f,\n", tB + tC + tE, tB + tC + tD0 + tD1 + tE);
    printf(" \"note\": \"E is the bucket pass's memory traffic only (no LDS table work); a one-level design needs table ranges of %llu slots per bucket (does not fit LDS)\"}\n",
           (unsigned long long)(n_slots / NB));
    return 0;
}
