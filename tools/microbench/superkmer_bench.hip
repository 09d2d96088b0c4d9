// What would a SUPER-K-MER partitioned insert cost on an MI355X?  ("Next round" 2: the round-4 pricing used 24-byte (key, position)
// records per position and concluded "the record is too fat"; a minimizer-run record carries ~2 bits per base plus a header.)
// The three pieces north_star's "bucket into LDS via per-wavefront radix partitioning" needs, on a REAL text (bytes: A C G T, anything else
// separates sequences), k odd, one canonical minimizer of m bases per k-mer (smallest hashed canonical m-mer of the window):
//   cut      per tile of 4096 positions: hashed canonical m-mers in LDS, the arg-min of every k-mer window, a super-k-mer = a run of
//            consecutive k-mers with the same minimizer OCCURRENCE (cut at 46 k-mers and at tile ends).  Record = 32 bytes:
//            [position:40 | k-mers:8 | 0:16][96 bases at 2 bits].  Two launches: COUNT (bytes per bucket, one global atomic per record) and
//            SCATTER (an atomic cursor per bucket hands the record its place; one 32-byte store).  Bucket = hash(minimizer) >> (64 - B).
//   dedup    one workgroup per bucket: every k-mer of every record of the bucket -> canonical form -> a table in LDS (128 KiB: 8192 slots of
//            {64-bit identity, smallest position}; claim by LDS compare-and-swap, lower by LDS atomic-min), 16 lanes per record.  Then the
//            table leaves: a novel flag (one byte store) at every surviving position, and the bucket's slot words (8 B each) as the lookup
//            table of the stages downstream.
// Identical k-mers have identical canonical minimizers, so a bucket sees every occurrence of its k-mers: the LDS table dedups exactly.
// (Identity = a 64-bit hash here; a product build would compare the full key against the record it sits in — one more LDS read per hit.)
// Prints one JSON line: records, bytes, the time of each piece, survivors, bucket fill statistics.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench/superkmer_bench.hip -o /tmp/superkmer_bench && /tmp/superkmer_bench text.bin 51 25 14
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
typedef unsigned int u32;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ int code_of(unsigned char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
static const int TILE = 4096, MAXK = 128, MAXRUN = 46, REC_WORDS = 4;      // 32-byte records: header + 3 words = 96 bases >= k + MAXRUN - 1 (k = 51)

// MODE 0: count records per bucket; MODE 1: scatter.  stats: [0] records, [1] k-mers in records (MODE 0 only, one atomic pair per workgroup)
// Second version (the first recomputed every m-mer and every window minimum from scratch: 150 + 80 operations per position, 4 ms per
// launch on E'): a thread owns SEG consecutive positions — the m-mer values ROLL (forward and reverse complement, two shifts each), and the
// window minimum is kept while it stays inside the window and rescanned (w LDS reads) only when it leaves: ~2 reads per position amortised.
static const int SEG = 16;      // TILE = 256 threads x SEG
template <int MODE>
__global__ void __launch_bounds__(256) cut_kernel(const unsigned char* text, u64 n, int k, int m, int bbits, u32* bucket_recs, u64* cursor, const u64* bucket_off,
                                                   u64* recs, u64* stats) {
    __shared__ unsigned char codes[TILE + MAXK];
    __shared__ u64 mh[TILE + MAXK];            // hash of the canonical m-mer starting at each tile position (~0: holds a separator)
    __shared__ unsigned short amin[TILE + 1];  // tile-relative position of the window's minimizer; 0xFFFF: not a k-mer
    __shared__ u32 s_recs, s_kmers;
    const u64 base = (u64)blockIdx.x * TILE;
    if (threadIdx.x == 0) { s_recs = 0; s_kmers = 0; }
    for (int i = threadIdx.x; i < TILE + k; i += 256) codes[i] = base + i < n ? (unsigned char)code_of(text[base + i]) : 4;
    __syncthreads();
    {   // rolling m-mers: this thread's positions [p0, p1) of the TILE + k - m + 1 m-mer starts
        const int total = TILE + k - m + 1, per = (total + 255) / 256;
        const int p0 = threadIdx.x * per, p1 = p0 + per < total ? p0 + per : total;
        const u64 mmask = m < 32 ? ((1ULL << (2 * m)) - 1ULL) : ~0ULL;
        u64 f = 0, r = 0; int good = 0;      // good = bases since the last separator
        for (int q = p0; q < p1 + m - 1 && q < TILE + k; q++) {
            const int c = codes[q];
            if (c > 3) { good = 0; f = 0; r = 0; }
            else { good++; f = ((f << 2) | (u64)c) & mmask; r = (r >> 2) | ((u64)(3 - c) << (2 * (m - 1))); }
            const int start = q - (m - 1);
            if (start >= p0 && start < p1) mh[start] = good >= m ? mix(f < r ? f : r) : ~0ULL;
        }
    }
    __syncthreads();
    const int w = k - m + 1;
    {   // sliding-window minimum over this thread's SEG positions
        const int i0 = threadIdx.x * SEG;
        u64 best = ~0ULL; int at = -1; int bad_until = -1;      // bad_until: last window start that still holds a separator m-mer
        for (int i = i0; i < i0 + SEG; i++) {
            if (at < i) {      // (first position, or the minimum has left the window): rescan
                best = ~0ULL; at = i; bad_until = -1;
                for (int j = 0; j < w; j++) { const u64 h = mh[i + j]; if (h == ~0ULL) bad_until = i + j; else if (h < best) { best = h; at = i + j; } }
                if (best == ~0ULL) at = i;
            } else {
                const u64 h = mh[i + w - 1];
                if (h == ~0ULL) bad_until = i + w - 1; else if (h < best) { best = h; at = i + w - 1; }
            }
            const bool valid = base + (u64)i + (u64)k <= n && bad_until < i && best != ~0ULL;
            amin[i] = valid ? (unsigned short)at : 0xFFFF;
        }
    }
    if (threadIdx.x == 0) amin[TILE] = 0xFFFF;
    __syncthreads();
    u32 my_recs = 0, my_kmers = 0;
    for (int i = threadIdx.x; i < TILE; i += 256) {
        const unsigned short a = amin[i];
        if (a == 0xFFFF) continue;
        // a record starts where the minimizer occurrence changes — and every MAXRUN k-mers of a long run
        int run_begin = i;
        while (run_begin > 0 && amin[run_begin - 1] == a) run_begin--;
        if ((i - run_begin) % MAXRUN != 0) continue;
        int len = 1;
        while (len < MAXRUN && amin[i + len] == a) len++;
        my_recs++; my_kmers += (u32)len;
        const u32 bucket = (u32)(mix(mh[a]) >> (64 - bbits));
        if (MODE == 0) { atomicAdd(&bucket_recs[bucket], 1u); continue; }
        const u64 slot = atomicAdd(&cursor[bucket], 1ULL);
        u64* out = recs + (bucket_off[bucket] + slot) * REC_WORDS;
        u64 wd[3] = {0, 0, 0};
        for (int j = 0; j < k + len - 1; j++) wd[j >> 5] |= (u64)(codes[i + j] & 3) << (62 - 2 * (j & 31));
        out[0] = ((base + (u64)i) << 24) | ((u64)len << 16);
        out[1] = wd[0]; out[2] = wd[1]; out[3] = wd[2];
    }
    if (MODE == 0) {
        if (my_recs) { atomicAdd(&s_recs, my_recs); atomicAdd(&s_kmers, my_kmers); }
        __syncthreads();
        if (threadIdx.x == 0 && s_recs) { atomicAdd(&stats[0], (u64)s_recs); atomicAdd(&stats[1], (u64)s_kmers); }
    }
}

static const int LDS_SLOTS = 8192;
__device__ __forceinline__ u64 revpairs(u64 x) { const u64 y = __brevll(x); return ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1); }
// one workgroup per bucket.  stats: [2] distinct k-mers (claims), [3] table overflow (buckets), [4] k-mers inserted
__global__ void __launch_bounds__(1024) dedup_kernel(const u64* recs, const u64* bucket_off, int k, unsigned char* novel, u64* table_out, u64* stats) {
    __shared__ u64 s_id[LDS_SLOTS];
    __shared__ u64 s_pos[LDS_SLOTS];
    const u32 b = blockIdx.x;
    for (int i = threadIdx.x; i < LDS_SLOTS; i += 1024) { s_id[i] = 0; s_pos[i] = ~0ULL; }
    __syncthreads();
    const u64 r0 = bucket_off[b], r1 = bucket_off[b + 1];
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;      // 16 lanes per record, 64 records per sweep of the workgroup (1024 threads: the
                                                                   // 128 KiB table leaves room for one workgroup per CU, so it brings its own 16 wavefronts)
    u32 claims = 0, inserted = 0; bool overflow = false;
    for (u64 r = r0 + (u64)grp; r < r1; r += 64) {
        const u64* rec = recs + r * REC_WORDS;
        const u64 hd = rec[0], w0 = rec[1], w1 = rec[2], w2 = rec[3];
        const int len = (int)((hd >> 16) & 255);
        const u64 pos0 = hd >> 24;
        for (int off = gl; off < len; off += 16) {
            // the k-mer at `off` (33 <= k <= 63): forward value F = 2k bits, as (hi = first 32 bases, lo = the other k - 32, right-aligned)
            const int s2 = 2 * (k - 32);
            const int bo = 2 * off, sh = bo & 63;
            const bool first = bo < 64;      // (selects, not an indexed array: the words stay in registers)
            const u64 a0 = first ? w0 : w1, a1 = first ? w1 : w2, a2 = first ? w2 : 0ULL;
            const u64 hi = sh ? ((a0 << sh) | (a1 >> (64 - sh))) : a0;
            const u64 lo = (sh ? ((a1 << sh) | (a2 >> (64 - sh))) : a1) >> (64 - s2);
            // reverse complement = the 2-bit groups of ~F in reverse order: over 128 bits, then down by 128 - 2k
            const u64 Fl = (hi << s2) | lo, Fh = hi >> (64 - s2);
            const u64 Rh = revpairs(~Fl), Rl = revpairs(~Fh);                  // rev128(~F) = (revpairs(~Fl) : revpairs(~Fh))
            const int dn = 128 - 2 * k;                                         // (0 < dn < 64)
            const u64 Cl = (Rl >> dn) | (Rh << (64 - dn)), Ch = Rh >> dn;       // RC as 128 bits (high word Ch holds 2k - 64 bits)
            const u64 chi = (Ch << (64 - s2)) | (Cl >> s2), clo = Cl & ((1ULL << s2) - 1ULL);
            const bool fwd_small = hi < chi || (hi == chi && lo <= clo);
            const u64 id = mix((fwd_small ? hi : chi) ^ mix((fwd_small ? lo : clo) + 0x9E3779B97F4A7C15ULL)) | 1ULL;
            const u64 gpos = pos0 + (u64)off;
            u32 s = (u32)(id >> 20) & (LDS_SLOTS - 1);
            inserted++;
            for (int probes = 0;; probes++) {
                if (probes == LDS_SLOTS) { overflow = true; break; }
                u64 v = s_id[s];
                if (v == 0) { v = atomicCAS(&s_id[s], 0ULL, id); if (v == 0) { claims++; v = id; } }
                if (v == id) { atomicMin(&s_pos[s], gpos); break; }
                s = (s + 1) & (LDS_SLOTS - 1);
            }
        }
    }
    __syncthreads();
    // the table leaves: novel flags at the surviving positions, the slot words as the downstream lookup table
    for (int i = threadIdx.x; i < LDS_SLOTS; i += 1024) {
        const u64 p = s_pos[i];
        table_out[(u64)b * LDS_SLOTS + i] = p == ~0ULL ? ~0ULL : ((s_id[i] << 40) | p);
        if (p != ~0ULL) novel[p] = 1;
    }
    if (claims) atomicAdd(&stats[2], (u64)claims);
    if (inserted) atomicAdd(&stats[4], (u64)inserted);
    if (overflow) atomicAdd(&stats[3], 1ULL);
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: superkmer_bench text.bin [k=51] [m=25] [bucket bits=14]\n"); return 1; }
    const int k = argc > 2 ? atoi(argv[2]) : 51, m = argc > 3 ? atoi(argv[3]) : 25, bbits = argc > 4 ? atoi(argv[4]) : 14;
    if (k + MAXRUN - 1 > 96 || k < 33 || k > 63 || m >= k || m > 31) { printf("{\"error\": \"unsupported k / m\"}\n"); return 1; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<unsigned char> text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const u64 n = text.size();
    if (!n) { printf("{\"error\": \"empty text\"}\n"); return 1; }
    const u64 NBK = 1ULL << bbits;
    unsigned char* d_text; u32* d_cnt; u64 *d_cursor, *d_off, *d_stats, *d_recs, *d_table; unsigned char* d_novel;
    CK(hipMalloc(&d_text, n + 256)); CK(hipMemcpy(d_text, text.data(), n, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_cnt, NBK * 4)); CK(hipMalloc(&d_cursor, NBK * 8)); CK(hipMalloc(&d_off, (NBK + 1) * 8)); CK(hipMalloc(&d_stats, 64));
    CK(hipMalloc(&d_novel, n + 256)); CK(hipMalloc(&d_table, NBK * LDS_SLOTS * 8));
    const u64 tiles = (n + TILE - 1) / TILE;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best_count = 1e9f, best_scatter = 1e9f, best_dedup = 1e9f, best_clear = 1e9f;
    std::vector<u64> st(8), off(NBK + 1);
    std::vector<u32> cnt(NBK);
    u64 n_recs = 0;
    d_recs = nullptr;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(d_cnt, 0, NBK * 4)); CK(hipMemset(d_cursor, 0, NBK * 8)); CK(hipMemset(d_stats, 0, 64));
        float ms;
        CK(hipEventRecord(e0)); cut_kernel<0><<<tiles, 256>>>(d_text, n, k, m, bbits, d_cnt, d_cursor, d_off, nullptr, d_stats); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); best_count = std::min(best_count, ms);
        CK(hipMemcpy(cnt.data(), d_cnt, NBK * 4, hipMemcpyDeviceToHost));
        off[0] = 0; for (u64 b = 0; b < NBK; b++) off[b + 1] = off[b] + cnt[b];      // (16 K values: a one-workgroup scan in a product build)
        n_recs = off[NBK];
        if (!d_recs) CK(hipMalloc(&d_recs, n_recs * REC_WORDS * 8 + 256));
        CK(hipMemcpy(d_off, off.data(), (NBK + 1) * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0)); cut_kernel<1><<<tiles, 256>>>(d_text, n, k, m, bbits, d_cnt, d_cursor, d_off, d_recs, d_stats); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); best_scatter = std::min(best_scatter, ms);
        CK(hipEventRecord(e0)); CK(hipMemsetAsync(d_novel, 0, n, 0)); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); best_clear = std::min(best_clear, ms);
        CK(hipEventRecord(e0)); dedup_kernel<<<NBK, 1024>>>(d_recs, d_off, k, d_novel, d_table, d_stats); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); best_dedup = std::min(best_dedup, ms);
        CK(hipMemcpy(st.data(), d_stats, 64, hipMemcpyDeviceToHost));
    }
    u32 mx = 0; for (u64 b = 0; b < NBK; b++) mx = std::max(mx, cnt[b]);
    printf("{\"text_bytes\": %llu, \"k\": %d, \"m\": %d, \"buckets\": %llu, \"records\": %llu, \"kmers_in_records\": %llu, \"kmers_per_record\": %.2f, "
           "\"record_bytes_total\": %llu, \"bytes_per_kmer\": %.2f, \"records_in_fullest_bucket\": %u, \"mean_records_per_bucket\": %.1f, "
           "\"count_ms\": %.3f, \"scatter_ms\": %.3f, \"novel_clear_ms\": %.3f, \"dedup_ms\": %.3f, \"sum_ms\": %.3f, "
           "\"distinct_kmers\": %llu, \"kmers_inserted\": %llu, \"buckets_overflowed\": %llu, \"lds_slots\": %d, \"table_out_bytes\": %llu}\n",
           (unsigned long long)n, k, m, (unsigned long long)NBK, (unsigned long long)n_recs, (unsigned long long)st[1], n_recs ? (double)st[1] / (double)n_recs : 0.0,
           (unsigned long long)(n_recs * 32), st[1] ? (double)(n_recs * 32) / (double)st[1] : 0.0, mx, (double)n_recs / (double)NBK,
           best_count, best_scatter, best_clear, best_dedup, best_count + best_scatter + best_clear + best_dedup,
           (unsigned long long)st[2], (unsigned long long)st[4], (unsigned long long)st[3], LDS_SLOTS, (unsigned long long)(NBK * LDS_SLOTS * 8));
    return 0;
}
