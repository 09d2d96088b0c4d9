// C ABI of libautocycler_hip.so (include/autocycler_hip.h).  No CPU fallback: every build call runs the
// HIP pipeline on a gfx950 device or fails.  (Under -DAC_EMU the very same entry points drive the serial
// emulation; that library is built only by the CPU test-suite and is named libautocycler_emu.so.)
#include <cstring>
#include <memory>
#include <mutex>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <thread>
#include <string>
#include <vector>

#include "../../include/autocycler_hip.h"
#include "graph_build.hpp"
#include "host_tail.hpp"
#include "gfa_writer.hpp"
#include "host_io.hpp"
#include <filesystem>
#include <fstream>
#include <chrono>
#include <fcntl.h>
#include <cerrno>
#include <unistd.h>
#include "device_rt.hpp"
#include "multi_build.hpp"

using namespace ac;

static thread_local std::string g_err;
static std::mutex g_build_mutex;   // one build at a time per process: the device / pinned arenas are shared
static std::atomic<int> g_host_side_device{0};      // ac_set_host_side_device: where ac_seqs_load / ac_seqs_from_raw run the end repair
static int g_live_shards = 0;      // a live sharded build owns the arenas between its phases: no other build may start

struct ac_graph {
    FinalGraph g;
    BuildTimings tm;
    MultiStats multi;          // n_ranks == 0: not built by ac_compress_build_multi
    std::vector<uint16_t> seq_ids;
    std::vector<uint32_t> seq_lens;
    bool positions_built = false;
    bool host_arrays = true;   // false: a rank of a sharded build that did not ask for the unitigs / links
    bool host_paths = true;    // false: ... that did not ask for its paths either
    std::vector<std::string> filenames, headers;   // graphs loaded from a GFA carry them (FN:Z / HD:Z)
};

struct ac_seqs {
    LoadResult lr;
    std::vector<ac_seq_view> views;
    void make_views() {
        views.resize(lr.seqs.size());
        for (size_t i = 0; i < views.size(); i++)
            views[i] = ac_seq_view{(const uint8_t*)lr.seqs[i].forward_seq.data(), lr.seqs[i].length, lr.seqs[i].id};
    }
};

// (a call that failed may have left a scan between its ticket take and its kernel: the calling thread's scan state pool starts over)
template <class F> static int guarded(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { g_err = e.what(); scan_pool().invalidate(); return 1; }
    catch (...) { g_err = "unknown internal error"; scan_pool().invalidate(); return 1; }
}

static void select_device(int device, bool refresh_tuning = true) {
    // (tests / the A/B tool change AC_* variables between builds: AC_TUNING_FOLLOW_ENV; else the knobs were read once.  The rank threads of a
    // multi-device build do not refresh: their entry point did, before it started them)
    if (refresh_tuning) tuning_refresh();
#ifndef AC_EMU
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        throw DeviceError("no HIP device available: the MI355X backend has no CPU fallback");
    if (device < 0 || device >= n) throw DeviceError("invalid HIP device ordinal " + std::to_string(device));
    AC_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    AC_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        throw DeviceError(std::string("device is ") + prop.gcnArchName + "; this library is built for gfx950 only");
#endif
    // The device arena is one process-wide bump allocator: its blocks live on the device they were allocated on.  A call that
    // names another ordinal gives them back first (a build never runs on HBM of another device), together with everything else
    // that is tied to the previous device's memory.  One build at a time per process (g_build_mutex), so nothing is in flight.
    int& arena_device = device_ctx().arena_device;      // (per context: the threads of a multi-device build each have their own)
    if (arena_device != device) {
        if (arena_device >= 0) {
            if (g_live_shards) throw DeviceError("a sharded build is in flight on device " + std::to_string(arena_device) + ": this process cannot use device " + std::to_string(device) + " until it is freed");
            Arena::device().release_all();
#ifndef AC_EMU
            Mailbox::get().release();      // the read-back page is mapped into the previous device's address space
#endif
        }
        arena_device = device;
    }
}

namespace ac { void select_device_checked(int device) { select_device(device, /*refresh_tuning=*/false); } }

static void validate(uint32_t k, const ac_seq_view* seqs, uint32_t n_seqs) {
    if (!seqs || n_seqs == 0) throw DeviceError("no sequences found in input assemblies");
    if (k % 2 == 0) throw DeviceError("--kmer must be odd");
    if (n_seqs > 32767) throw DeviceError("no more than 32767 input sequences are allowed");
    for (uint32_t i = 0; i < n_seqs; i++) {
        if (!seqs[i].fwd) throw DeviceError("null sequence pointer");
        if (seqs[i].length < k) throw DeviceError("sequence shorter than k");
    }
}

// The device entries take the text layout from the caller: check it before any kernel indexes the text with it (the header
// promises errors, not faults).  off[i] = first padded byte of sequence i; every padded sequence is followed by one separator.
static void validate_layout(uint32_t k, uint64_t n_text, const uint64_t* off, const uint32_t* len, const uint16_t* d1, const uint16_t* d2,
                            uint32_t n_seqs) {
    if (k < 1 || k % 2 == 0) throw DeviceError("--kmer must be odd");
    if (!off || !len || !d1 || !d2) throw DeviceError("null sequence table");
    uint64_t prev_end = 0;      // index of the separator before the next sequence
    for (uint32_t i = 0; i < n_seqs; i++) {
        if (len[i] < k) throw DeviceError("sequence " + std::to_string(i + 1) + " is shorter than k");
        const uint64_t plen = (uint64_t)len[i] + k - 1;
        if (off[i] != prev_end + 1) throw DeviceError("sequence table: sequence " + std::to_string(i + 1) + " does not start right behind the separator of the previous one");
        if (off[i] + plen + 1 > n_text) throw DeviceError("sequence table: sequence " + std::to_string(i + 1) + " runs past the end of the text");
        if (d1[i] > k - 1 || d2[i] > k - 1 || (uint32_t)d1[i] + d2[i] > k - 1) throw DeviceError("sequence table: more padding dots than k - 1 on sequence " + std::to_string(i + 1));
        prev_end = off[i] + plen;
    }
    if (n_seqs && prev_end + 1 != n_text) throw DeviceError("sequence table: the text does not end with the separator of the last sequence");
}

// compress.rs:42-44 behind the ABI: one device pipeline from the packed text to the final UnitigGraph.
static void build_graph(GraphBuilder& b, uint32_t assembly_count, ac_graph* h) {
    b.build(assembly_count, &h->g);
    h->tm = b.timings();
}

// Writes the pieces one behind the other into `path`, each at its own offset by its own thread (pwrite): the GFA of config C is
// 95 MB.  With the S and L lines formatted by several threads too (gfa_chunks) the write stage of the whole command went from 48 to
// 26 ms (profiles/r07l_e2e_configC.log).
static void write_pieces(const std::string& path, const std::vector<std::string>& pieces, int threads) {
    // into a temporary file next to the target, renamed over it when every piece is down: a failed write never leaves a truncated
    // input_assemblies.gfa behind
    const std::string tmp = path + ".tmp." + std::to_string((long)::getpid());
    int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw UserError("failed to write " + path);
    std::vector<uint64_t> at(pieces.size() + 1, 0);
    for (size_t i = 0; i < pieces.size(); i++) at[i + 1] = at[i] + pieces[i].size();
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto worker = [&] {
        for (size_t i; (i = next.fetch_add(1)) < pieces.size();) {
            const char* p = pieces[i].data(); uint64_t left = pieces[i].size(), off = at[i];
            while (left) {
                ssize_t w = ::pwrite(fd, p, left, (off_t)off);
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) { bad.store(true); return; }
                p += w; left -= (uint64_t)w; off += (uint64_t)w;
            }
        }
    };
    const int T = std::max(1, std::min<int>({threads, 16, (int)pieces.size()}));
    std::vector<std::thread> pool;
    for (int i = 1; i < T; i++) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    const bool closed = ::close(fd) == 0;
    if (!closed || bad.load() || ::rename(tmp.c_str(), path.c_str()) != 0) { ::unlink(tmp.c_str()); throw UserError("failed to write " + path); }
}

extern "C" {

const char* ac_last_error(void) { return g_err.c_str(); }
const char* ac_version(void) {
#ifdef AC_EMU
    return "autocycler_amd 0.1 (CPU emulation, tests only)";
#else
    return "autocycler_amd 0.1 (gfx950)";
#endif
}
int ac_abi_version(void) { return AC_ABI_VERSION; }
int ac_set_host_side_device(int device) { if (device < 0) { g_err = "invalid HIP device ordinal"; return 1; } g_host_side_device.store(device); return 0; }
void ac_set_stage_timing(int on) { set_stage_timing(on != 0); }
// The device arena and the pool of pinned result blocks stay allocated between builds; this gives them back (e.g. before a
// long-lived host process turns to other work).  Graph handles that are still alive keep their blocks.
int ac_release_memory(void) {
    return guarded([&] {
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        Arena::device().release_all();
        release_multi_contexts();
        PinnedPool::get().trim();
        release_host_stager();
        scan_pool().release();
#ifndef AC_EMU
        Mailbox::get().release();
#endif
    });
}
uint32_t ac_max_kmer(void) { int m = max_supported_k(); return (uint32_t)(m % 2 ? m : m - 1); }
int ac_device_count(void) {
#ifndef AC_EMU
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
#else
    return 0;
#endif
}

uint64_t ac_text_size(uint32_t k, const ac_seq_view* seqs, uint32_t n_seqs) {
    uint64_t n = 1;
    for (uint32_t i = 0; i < n_seqs; i++) n += (uint64_t)seqs[i].length + k - 1 + 1;
    return n;
}
int ac_layout_text(uint32_t k, const ac_seq_view* seqs, uint32_t n_seqs, uint8_t* text, uint64_t* seq_off,
                   uint16_t* seq_d1, uint16_t* seq_d2) {
    return guarded([&] {
        std::vector<SeqView> v(n_seqs);
        for (uint32_t i = 0; i < n_seqs; i++) v[i] = SeqView{seqs[i].fwd, seqs[i].length};
        std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint16_t> d1, d2;
        std::vector<uint8_t> t = layout_text(v, k, &off, &len, &d1, &d2);
        memcpy(text, t.data(), t.size());
        for (uint32_t i = 0; i < n_seqs; i++) { seq_off[i] = off[i]; seq_d1[i] = d1[i]; seq_d2[i] = d2[i]; }
    });
}

int ac_pack_text(const uint8_t* text, uint64_t n_text, uint64_t* bits, uint32_t* mask32, int force_scalar) {
    return guarded([&] {
        if (!text || !bits || !mask32) throw DeviceError("null pointer");
        pack_text_host(text, n_text, bits, mask32, force_scalar != 0);
    });
}

int ac_compress_build(uint32_t k, uint32_t assembly_count, const ac_seq_view* seqs, uint32_t n_seqs, int device,
                      ac_graph** out) {
    return guarded([&] {
        validate(k, seqs, n_seqs);
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        auto h = std::make_unique<ac_graph>();
        std::vector<SeqView> v(n_seqs);
        for (uint32_t i = 0; i < n_seqs; i++) {
            v[i] = SeqView{seqs[i].fwd, seqs[i].length};
            h->seq_ids.push_back(seqs[i].id);
            h->seq_lens.push_back(seqs[i].length);
        }
        GraphBuilder b(k);
        b.set_sequences_host(v);
        build_graph(b, assembly_count, h.get());
        *out = h.release();
    });
}

// compress.rs:42-44 over several devices of one node: one call, one process, the same graph.
int ac_compress_build_multi(uint32_t k, uint32_t assembly_count, const ac_seq_view* seqs, uint32_t n_seqs, const int* devices, int n_devices,
                            ac_graph** out) {
    return guarded([&] {
        validate(k, seqs, n_seqs);
        if (!devices || n_devices < 1) throw DeviceError("ac_compress_build_multi: no devices");
        if (n_devices > 64) throw DeviceError("ac_compress_build_multi: more than 64 devices");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        auto h = std::make_unique<ac_graph>();
        std::vector<SeqView> v(n_seqs);
        for (uint32_t i = 0; i < n_seqs; i++) {
            v[i] = SeqView{seqs[i].fwd, seqs[i].length};
            h->seq_ids.push_back(seqs[i].id);
            h->seq_lens.push_back(seqs[i].length);
        }
        tuning_refresh();
        int transport = MULTI_AUTO;
        if (const int e = tuning_multi_transport()) transport = e == 1 ? MULTI_HOST_STAGED : MULTI_RCCL;
        build_multi(k, assembly_count, v, std::vector<int>(devices, devices + n_devices), transport, &h->g, &h->tm, &h->multi);
        *out = h.release();
    });
}
int ac_multi_info_get(const ac_graph* g, ac_multi_info* o) {
    if (!g || !o) { g_err = "null pointer"; return 1; }
    const MultiStats& m = g->multi;
    memset(o, 0, sizeof *o);
    o->n_ranks = m.n_ranks; o->transport = m.transport;
    o->bytes_fragments = m.bytes_fragments; o->bytes_bitmap = m.bytes_bitmap; o->bytes_degrees = m.bytes_degrees; o->bytes_links = m.bytes_links;
    o->bytes_queries = m.bytes_queries; o->bytes_answers = m.bytes_answers; o->bytes_reduce = m.bytes_reduce;
    o->queries_total = m.queries_total; o->queries_sent_away = m.queries_sent_away;
    o->table_capacity_max = m.table_capacity_max; o->table_capacity_sum = m.table_capacity_sum;
    o->union_text_bytes = m.union_text_bytes; o->fragments = m.fragments; o->distinct = m.distinct;
    o->seconds_total = m.seconds_total; o->seconds_exchange_max = m.seconds_exchange_max;
    o->candidates_total = m.candidates_total; o->candidates_owned_max = m.candidates_owned_max;
    o->bytes_sibling = m.bytes_sibling; o->bytes_tail = m.bytes_tail; o->degrees_open = m.degrees_open; o->bytes_received_max = m.bytes_received_max;
    o->path_runs_copied = m.path_runs_copied;
    return 0;
}
size_t ac_multi_info_get_sized(const ac_graph* g, ac_multi_info* out, size_t out_size) {
    ac_multi_info t;
    if (!g || !out) { g_err = "null pointer"; return sizeof t; }
    ac_multi_info_get(g, &t);
    memcpy(out, &t, out_size < sizeof t ? out_size : sizeof t);
    return sizeof t;
}

int ac_compress_build_device(uint32_t k, uint32_t assembly_count, const void* d_text, uint64_t n_text,
                             const uint64_t* seq_off, const uint32_t* seq_len, const uint16_t* seq_ids,
                             const uint16_t* seq_d1, const uint16_t* seq_d2, uint32_t n_seqs, int device,
                             ac_graph** out) {
    return guarded([&] {
        if (!d_text || n_seqs == 0) throw DeviceError("no sequences found in input assemblies");
        if (n_seqs > 32767) throw DeviceError("no more than 32767 input sequences are allowed");
        validate_layout(k, n_text, seq_off, seq_len, seq_d1, seq_d2, n_seqs);
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        auto h = std::make_unique<ac_graph>();
        std::vector<uint64_t> off(seq_off, seq_off + n_seqs);
        std::vector<uint32_t> len(seq_len, seq_len + n_seqs);
        std::vector<uint16_t> d1(seq_d1, seq_d1 + n_seqs), d2(seq_d2, seq_d2 + n_seqs);
        h->seq_ids.assign(seq_ids, seq_ids + n_seqs);
        h->seq_lens = len;
        GraphBuilder b(k);
        b.set_text_device((const uint8_t*)d_text, n_text, off, len, d1, d2);
        build_graph(b, assembly_count, h.get());
        *out = h.release();
    });
}

// ---- one job sharded by sequence over several devices (the collectives between the phases are the caller's) ------
struct ac_shard {
    std::unique_ptr<GraphBuilder> b;
    std::vector<uint16_t> seq_ids;
    std::vector<uint32_t> seq_lens;
    int device = 0;
    uint32_t n_shards = 1;
    int phase = 0;   // 1 fragments ready, 2 owned k-mers inserted, 3 novel list + degree words, 4 unitigs + link words, 5 links complete + walk
                     // queries ready, 6 walked, 7 reduced quantities imported, 8 finished
};

int ac_shard_begin(uint32_t k, uint32_t local_assembly_count, const void* d_text, uint64_t n_text, const uint64_t* seq_off,
                   const uint32_t* seq_len, const uint16_t* seq_ids, const uint16_t* seq_d1, const uint16_t* seq_d2,
                   uint32_t n_seqs, int device, ac_shard** out) {
    return guarded([&] {
        if (!d_text || n_seqs == 0) throw DeviceError("no sequences found in input assemblies");
        if (n_seqs > 32767) throw DeviceError("no more than 32767 input sequences are allowed");
        validate_layout(k, n_text, seq_off, seq_len, seq_d1, seq_d2, n_seqs);
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("another sharded build is in flight in this process");
        select_device(device);
        auto h = std::make_unique<ac_shard>();
        h->device = device;
        std::vector<uint64_t> off(seq_off, seq_off + n_seqs);
        std::vector<uint32_t> len(seq_len, seq_len + n_seqs);
        std::vector<uint16_t> d1(seq_d1, seq_d1 + n_seqs), d2(seq_d2, seq_d2 + n_seqs);
        h->seq_ids.assign(seq_ids, seq_ids + n_seqs);
        h->seq_lens = len;
        h->b = std::make_unique<GraphBuilder>(k);
        h->b->set_text_device((const uint8_t*)d_text, n_text, off, len, d1, d2);
        h->b->shard_begin(local_assembly_count);
        h->phase = 1;
        g_live_shards++;
        *out = h.release();
    });
}
int ac_shard_fragment_sizes(const ac_shard* s, uint64_t* text_bytes, uint64_t* n_fragments) {
    *text_bytes = s->b->fragment_text_bytes();
    *n_fragments = s->b->fragment_count();
    return 0;
}
uint64_t ac_shard_local_distinct(const ac_shard* s) { return s->b->local_distinct_count(); }
void ac_shard_set_distinct_upper_bound(ac_shard* s, uint64_t n) { s->b->set_distinct_upper_bound(n); }
int ac_shard_fragments_export(ac_shard* s, void* d_text_out, void* d_meta_out) {
    return guarded([&] {
        if (s->phase < 1) throw DeviceError("ac_shard_fragments_export: no fragments yet");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->fragments_export(d_text_out, d_meta_out);
    });
}
// The fragment text as 2-bit codes on the union text's word grid (a quarter of the bytes; nothing to pack on the receiving side).
uint64_t ac_shard_fragment_packed_words(const ac_shard* s, uint64_t union_off) { return s->phase >= 1 ? s->b->fragment_packed_words(union_off) : 0; }
int ac_shard_fragments_export_packed(ac_shard* s, uint64_t union_off, void* d_words_out, void* d_meta_out) {
    return guarded([&] {
        if (s->phase < 1) throw DeviceError("ac_shard_fragments_export_packed: no fragments yet");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->fragments_export_packed(union_off, d_words_out, d_meta_out);
    });
}
int ac_shard_build_union_packed(ac_shard* s, uint32_t rank, uint32_t n_shards, const void* d_staged_words, const uint64_t* first_word,
                                const uint64_t* n_words, uint64_t n_union_text, const void* d_meta, uint64_t n_fragments_total) {
    return guarded([&] {
        if (s->phase != 1) throw DeviceError("ac_shard_build_union_packed: wrong phase");
        if (!first_word || !n_words) throw DeviceError("ac_shard_build_union_packed: no word table");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->shard_build_union_packed(rank, n_shards, d_staged_words, first_word, n_words, n_union_text, d_meta, n_fragments_total);
        s->n_shards = n_shards;
        s->phase = 2;
    });
}
int ac_shard_build_union(ac_shard* s, uint32_t rank, uint32_t n_shards, const void* d_union_text, uint64_t n_union_text,
                         const void* d_meta, uint64_t n_fragments_total) {
    return guarded([&] {
        if (s->phase != 1) throw DeviceError("ac_shard_build_union: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->shard_build_union(rank, n_shards, (const uint8_t*)d_union_text, n_union_text, d_meta, n_fragments_total);
        s->n_shards = n_shards;
        s->phase = 2;
    });
}
uint64_t ac_shard_bitmap_words(const ac_shard* s) { return s->phase >= 2 ? s->b->bitmap_words() : 0; }
int ac_shard_bitmap_export(ac_shard* s, void* d_out_u64) {
    return guarded([&] {
        if (s->phase != 2) throw DeviceError("ac_shard_bitmap_export: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->bitmap_export(d_out_u64);
    });
}
int ac_shard_build_novel(ac_shard* s, const void* d_bitmap_sum_u64) {
    return guarded([&] {
        if (s->phase != 2) throw DeviceError("ac_shard_build_novel: wrong phase");
        if (!d_bitmap_sum_u64 && s->n_shards > 1) throw DeviceError("ac_shard_build_novel: the summed bitmap is required when there are several shards");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->shard_build_novel(d_bitmap_sum_u64);
        s->phase = 3;
    });
}
// round 5: the sibling bits (2 per distinct k-mer, by novel index).  ac_shard_sib_words() > 0 after ac_shard_build_novel: the degree stage
// waits for their sum — ac_shard_sib_export -> all-reduce SUM (uint64) -> ac_shard_degrees; 0: it has run already.
uint64_t ac_shard_sib_words(const ac_shard* s) { return s->phase == 3 ? s->b->sib_words() : 0; }
int ac_shard_sib_export(ac_shard* s, void* d_out_u64) {
    return guarded([&] {
        if (s->phase != 3) throw DeviceError("ac_shard_sib_export: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->sib_export(d_out_u64);
    });
}
int ac_shard_degrees(ac_shard* s, const void* d_sib_sum_u64) {
    return guarded([&] {
        if (s->phase != 3) throw DeviceError("ac_shard_degrees: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->shard_degrees(d_sib_sum_u64);
    });
}
uint64_t ac_shard_degree_bytes(const ac_shard* s) {
    if (s->phase != 3) return 0;
    try { return s->b->degree_bytes(); } catch (const std::exception& e) { g_err = e.what(); return 0; }
}
uint64_t ac_shard_distinct_count(const ac_shard* s) { return s->b->distinct_count(); }
uint64_t ac_shard_table_capacity(const ac_shard* s) { return s->b->timings().table_capacity; }
int ac_shard_degrees_export(ac_shard* s, void* d_out_u32) {
    return guarded([&] {
        if (s->phase != 3) throw DeviceError("ac_shard_degrees_export: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->degrees_export(d_out_u32);
    });
}
int ac_shard_build_graph(ac_shard* s, const void* d_degrees_sum_u32) {
    return guarded([&] {
        if (s->phase != 3) throw DeviceError("ac_shard_build_graph: wrong phase");
        if (!d_degrees_sum_u32 && s->n_shards > 1) throw DeviceError("ac_shard_build_graph: the summed degree bytes are required when there are several shards");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->shard_build_graph(d_degrees_sum_u32);
        s->phase = 4;
    });
}
uint32_t ac_shard_unitig_count(const ac_shard* s) { return s->b->unitig_count(); }
int ac_shard_links_export(ac_shard* s, void* d_links_i32, void* d_wlinks_i64) {
    return guarded([&] {
        if (s->phase != 4) throw DeviceError("ac_shard_links_export: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->links_export(d_links_i32, d_wlinks_i64);
    });
}
int ac_shard_links_import(ac_shard* s, const void* d_links_i32, const void* d_wlinks_i64) {
    return guarded([&] {
        if (s->phase != 4) throw DeviceError("ac_shard_links_import: wrong phase");
        if (!d_links_i32 && s->n_shards > 1) throw DeviceError("ac_shard_links_import: the summed link words are required when there are several shards");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->links_import(d_links_i32, d_wlinks_i64);
        s->phase = 5;
    });
}
uint64_t ac_shard_query_count(const ac_shard* s) { return s->phase >= 5 ? s->b->query_count() : 0; }
uint32_t ac_shard_query_key_words(const ac_shard* s) { return s->b->query_key_words(); }
int ac_shard_queries_export(ac_shard* s, void* d_out_u64) {
    return guarded([&] {
        if (s->phase != 5) throw DeviceError("ac_shard_queries_export: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->queries_export(d_out_u64);
    });
}
int ac_shard_answer(ac_shard* s, const void* d_keys_u64, uint64_t n_queries, void* d_out_u64) {
    return guarded([&] {
        if (s->phase != 5) throw DeviceError("ac_shard_answer: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->answer_queries(d_keys_u64, n_queries, d_out_u64);
    });
}
// The owner-routed form of the walk-start exchange (what ac_compress_build_multi does inside the library, multi_build.cpp): the
// rank's keys ordered by owner, counts[r] of them for rank r — one all-to-all sends each key to the ONE rank whose table can answer it,
// ac_shard_answer looks the received keys up, the reverse all-to-all brings the answers back in the same order.
int ac_shard_queries_route(ac_shard* s, uint32_t n_shards, void* d_routed_keys_u64, uint64_t* counts) {
    return guarded([&] {
        if (s->phase != 5) throw DeviceError("ac_shard_queries_route: wrong phase");
        if (n_shards == 0 || !counts) throw DeviceError("ac_shard_queries_route: no ranks");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->queries_route(n_shards, d_routed_keys_u64, counts);
    });
}
int ac_shard_walk_routed(ac_shard* s, const void* d_routed_answers_u64) {
    return guarded([&] {
        if (s->phase != 5) throw DeviceError("ac_shard_walk_routed: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->shard_walk_routed(d_routed_answers_u64);
        s->phase = 6;
    });
}
int ac_shard_walk(ac_shard* s, const void* d_answers_u64) {
    return guarded([&] {
        if (s->phase != 5) throw DeviceError("ac_shard_walk: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->shard_walk(d_answers_u64);
        s->phase = 6;
    });
}
int ac_shard_reduce_export(ac_shard* s, void* d_sum_i32, void* d_min_i32) {
    return guarded([&] {
        if (s->phase != 6) throw DeviceError("ac_shard_reduce_export: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->reduce_export((int32_t*)d_sum_i32, (int32_t*)d_min_i32);
    });
}
int ac_shard_reduce_import(ac_shard* s, const void* d_sum_i32, const void* d_min_i32) {
    return guarded([&] {
        if (s->phase != 6) throw DeviceError("ac_shard_reduce_import: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->reduce_import((const int32_t*)d_sum_i32, (const int32_t*)d_min_i32);
        s->phase = 7;
    });
}
int ac_shard_set_allreduce(ac_shard* s, ac_allreduce_fn fn, void* user) {
    return guarded([&] {
        if (s->phase > 7) throw DeviceError("ac_shard_set_allreduce: wrong phase");
        if (!fn) { s->b->set_tail_exchange(nullptr); return; }
        s->b->set_tail_exchange([fn, user](void* d_buf, uint64_t count, int dtype, int op) {
            if (fn(user, d_buf, count, dtype, op) != 0) throw DeviceError("the caller's all-reduce failed (ac_shard_set_allreduce)");
        });
    });
}
int ac_device_copy(void* dst, const void* src, uint64_t bytes, int device) {
    return guarded([&] {
        if (!bytes) return;
#ifdef AC_EMU
        (void)device;
        memmove(dst, src, (size_t)bytes);
#else
        AC_HIP_CHECK(hipSetDevice(device));
        AC_HIP_CHECK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDefault));
#endif
    });
}
int ac_shard_finish(ac_shard* s, int want, ac_graph** out) {
    return guarded([&] {
        if (s->phase != 7) throw DeviceError("ac_shard_finish: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        auto h = std::make_unique<ac_graph>();
        h->seq_ids = s->seq_ids;
        h->seq_lens = s->seq_lens;
        s->b->shard_finish(&h->g, (want & 1) != 0, (want & 2) != 0);
        h->tm = s->b->timings();
        h->host_arrays = (want & 1) != 0;
        h->host_paths = (want & 2) != 0;
        s->phase = 8;
        *out = h.release();
    });
}
uint64_t ac_shard_path_entries(const ac_shard* s) { return s->b->path_entry_count(); }
int ac_shard_paths_export(ac_shard* s, void* d_out_i32) {
    return guarded([&] {
        if (s->phase != 8) throw DeviceError("ac_shard_paths_export: wrong phase");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(s->device);
        s->b->paths_export(d_out_i32);
    });
}
void ac_shard_free(ac_shard* s) {
    if (!s) return;
    std::lock_guard<std::mutex> lock(g_build_mutex);
    if (s->phase >= 1) g_live_shards--;
    delete s;
}
// The rank that writes the GFA replaces its own paths by those of ALL sequences of the job (rank order).
int ac_graph_set_paths(ac_graph* g, uint32_t n_seqs_total, const uint16_t* seq_ids, const uint32_t* seq_lens,
                       const uint64_t* path_counts, const void* d_path_i32, int device) {
    return guarded([&] {
        if (!g->host_arrays) throw DeviceError("ac_graph_set_paths: this graph was finished without host arrays");
        if (n_seqs_total == 0 || n_seqs_total > 32767) throw DeviceError("no more than 32767 input sequences are allowed");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        select_device(device);
        std::vector<uint64_t> off((size_t)n_seqs_total + 1, 0);
        for (uint32_t i = 0; i < n_seqs_total; i++) off[i + 1] = off[i] + path_counts[i];
        uint64_t n = off[n_seqs_total];
        HostBlock blk = PinnedPool::get().alloc(n * 4);
        copy_d2h(blk.p, d_path_i32, n * 4);
        const int32_t* p = (const int32_t*)blk.p;
        for (uint32_t s = 0; s < n_seqs_total; s++) {   // every path must spell its sequence (unitig_graph.rs:160-174)
            uint64_t sum = 0;
            for (uint64_t i = off[s]; i < off[s + 1]; i++) {
                uint32_t u = (uint32_t)(p[i] < 0 ? -p[i] : p[i]);
                if (u == 0 || u > g->g.n_unitigs) throw DeviceError("internal error: gathered path names an unknown unitig");
                sum += g->g.seq_len[u - 1];
            }
            if (sum != seq_lens[s]) throw DeviceError("internal error: gathered path length mismatch for sequence " + std::to_string(s + 1));
        }
        g->g.path_block = std::move(blk);
        g->g.path = p;
        g->g.n_path = n;
        g->g.path_off = off;
        g->host_paths = true;
        g->seq_ids.assign(seq_ids, seq_ids + n_seqs_total);
        g->seq_lens.assign(seq_lens, seq_lens + n_seqs_total);
        g->positions_built = false;
    });
}
uint32_t ac_graph_seq_count(const ac_graph* g) { return (uint32_t)g->seq_ids.size(); }
int ac_path_counts(const ac_graph* g, uint64_t* counts) {   // entries per sequence; also valid without host arrays
    for (size_t s = 0; s + 1 < g->g.path_off.size(); s++) counts[s] = g->g.path_off[s + 1] - g->g.path_off[s];
    return 0;
}

// sequence_end_repair (compress.rs:202-270) on a device-resident text of padded, unrepaired sequences.
int ac_end_repair_device(uint32_t k, void* d_text, uint64_t n_text, const uint64_t* seq_off, const uint32_t* seq_len,
                         uint16_t* seq_d1, uint16_t* seq_d2, uint32_t n_seqs, int device, double* seconds, uint64_t* n_matches) {
    return guarded([&] {
        if (!d_text || n_seqs == 0) throw DeviceError("no sequences found in input assemblies");
        if (!seq_d1 || !seq_d2) throw DeviceError("null sequence table");
        {   // seq_d1 / seq_d2 are outputs here (the repair counts the surviving dots itself): only the layout is checked
            std::vector<uint16_t> zero(n_seqs, 0);
            validate_layout(k, n_text, seq_off, seq_len, zero.data(), zero.data(), n_seqs);
        }
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        std::vector<uint64_t> off(seq_off, seq_off + n_seqs);
        std::vector<uint32_t> len(seq_len, seq_len + n_seqs);
        std::vector<uint16_t> d1(n_seqs, 0), d2(n_seqs, 0);
        RepairTimings tm;
        end_repair_device(k, (uint8_t*)d_text, n_text, off, len, &d1, &d2, &tm);
        for (uint32_t i = 0; i < n_seqs; i++) { seq_d1[i] = d1[i]; seq_d2[i] = d2[i]; }
        if (seconds) *seconds = tm.total;
        if (n_matches) *n_matches = tm.matches;
    });
}

// pairwise_contig_distances (cluster.rs:132-157), the first step of `autocycler cluster`, on the graph just built.
int ac_pairwise_distances(const ac_graph* g, int device, double* out) {
    return guarded([&] {
        if (!g->host_arrays || !g->host_paths) throw DeviceError("this rank kept no host arrays (sharded build, not the writing rank)");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        pairwise_distances_device(g->g, (uint32_t)g->seq_ids.size(), out);
    });
}

// UnitigGraph::from_gfa_lines (unitig_graph.rs:55-174) for the GFAs `compress` writes: what `cluster` and `decompress` start from.
int ac_graph_from_gfa(const char* gfa_text, uint64_t len, ac_graph** out) {
    return guarded([&] {
        if (!gfa_text) throw DeviceError("no GFA text");
        auto h = std::make_unique<ac_graph>();
        std::vector<SeqMeta> meta;
        load_gfa(gfa_text, (size_t)len, &h->g, &meta);
        for (auto& m : meta) { h->seq_ids.push_back(m.id); h->seq_lens.push_back(m.length); h->filenames.push_back(m.filename); h->headers.push_back(m.contig_header); }
        *out = h.release();
    });
}
uint32_t ac_graph_kmer_size(const ac_graph* g) { return g->g.k; }
int ac_graph_seq_info(const ac_graph* g, uint32_t i, uint16_t* id, uint32_t* length, const char** filename, const char** header) {
    if (i >= g->seq_ids.size()) { g_err = "sequence index out of range"; return 1; }
    if (id) *id = g->seq_ids[i];
    if (length) *length = g->seq_lens[i];
    if (filename) *filename = i < g->filenames.size() ? g->filenames[i].c_str() : nullptr;
    if (header) *header = i < g->headers.size() ? g->headers[i].c_str() : nullptr;
    return 0;
}
// reconstruct_original_sequences (unitig_graph.rs:362-388) for one sequence; out holds its LN bytes.
int ac_decompress_seq(const ac_graph* g, uint32_t seq_index, uint8_t* out) {
    return guarded([&] {
        if (seq_index >= g->seq_ids.size()) throw DeviceError("sequence index out of range");
        if (!g->host_arrays || !g->host_paths) throw DeviceError("this rank kept no host arrays (sharded build, not the writing rank)");
        decompress_sequence(g->g, seq_index, (char*)out);
    });
}

// ---- the round-trip verifier behind the ABI (kernels_verify.inc) ------------------------------------------------------------------
static void fill_report(const VerifyReport& r, ac_verify_report* o) {
    memset(o, 0, sizeof *o);
    o->failed = r.failed;
    o->first_bad_unitig = r.first_bad_unitig; o->first_bad_link = r.first_bad_link; o->first_bad_path_entry = r.first_bad_path_entry;
    o->first_bad_sequence = r.first_bad_sequence; o->first_bad_base = r.first_bad_base;
    o->unitigs = r.unitigs; o->links = r.links; o->path_entries = r.path_entries; o->bases_checked = r.bases_checked;
    o->self_mirror_links = r.self_mirror_links; o->seconds = r.seconds;
    o->checks = r.checks; o->first_bad_junction = r.first_bad_junction;
}
int ac_verify_graph_device(const ac_graph* g, const void* d_text, uint64_t n_text, const uint64_t* seq_off, const uint32_t* seq_len,
                           uint32_t n_seqs, int device, ac_verify_report* report) {
    return guarded([&] {
        if (!g || !d_text || !seq_off || !seq_len || !report) throw DeviceError("null pointer");
        if (!g->host_arrays || !g->host_paths) throw DeviceError("this rank kept no host arrays (sharded build, not the writing rank)");
        if (n_seqs != g->seq_lens.size()) throw DeviceError("ac_verify_graph: the graph was built from " + std::to_string(g->seq_lens.size()) + " sequences, not " + std::to_string(n_seqs));
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        std::vector<uint64_t> off(seq_off, seq_off + n_seqs);
        std::vector<uint32_t> len(seq_len, seq_len + n_seqs);
        VerifyReport r;
        verify_graph_device(g->g, (const uint8_t*)d_text, n_text, off, len, &r);
        fill_report(r, report);
    });
}
int ac_verify_graph(const ac_graph* g, const ac_seq_view* seqs, uint32_t n_seqs, int device, ac_verify_report* report) {
    return guarded([&] {
        if (!g || !seqs || !report) throw DeviceError("null pointer");
        if (!g->host_arrays || !g->host_paths) throw DeviceError("this rank kept no host arrays (sharded build, not the writing rank)");
        if (n_seqs != g->seq_lens.size()) throw DeviceError("ac_verify_graph: the graph was built from " + std::to_string(g->seq_lens.size()) + " sequences, not " + std::to_string(n_seqs));
        std::vector<SeqView> v(n_seqs);
        for (uint32_t i = 0; i < n_seqs; i++) {
            if (!seqs[i].fwd) throw DeviceError("null sequence");
            v[i] = SeqView{seqs[i].fwd, seqs[i].length};
        }
        std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint16_t> d1, d2;
        std::vector<uint8_t> text = layout_text(v, g->g.k, &off, &len, &d1, &d2);
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        VerifyReport r;
        // (the text goes up through an allocation of its own: the verifier resets the arena for its working set)
#ifdef AC_EMU
        verify_graph_device(g->g, text.data(), text.size(), off, len, &r);
#else
        void* d_text = nullptr;
        AC_HIP_CHECK(hipMalloc(&d_text, text.size() + 64));
        struct Free { void* p; ~Free() { (void)hipFree(p); } } fr{d_text};
        AC_HIP_CHECK(hipMemcpy(d_text, text.data(), text.size(), hipMemcpyHostToDevice));
        verify_graph_device(g->g, (const uint8_t*)d_text, text.size(), off, len, &r);
#endif
        fill_report(r, report);
    });
}

// reconstruct_original_sequences for every sequence of the graph at once, on the device (kernels_verify.inc): out = sum of the sequence
// lengths bytes, sequence i at offset sum(length[0 .. i)).  What ac_decompress_seq does one sequence at a time on the host.
int ac_decompress_device(const ac_graph* g, int device, uint8_t* out, uint64_t out_bytes) {
    return guarded([&] {
        if (!g || !out) throw DeviceError("null pointer");
        if (!g->host_arrays || !g->host_paths) throw DeviceError("this rank kept no host arrays (sharded build, not the writing rank)");
        uint64_t need = 0;
        for (uint32_t l : g->seq_lens) need += l;
        if (out_bytes < need) throw DeviceError("ac_decompress_device: the buffer holds " + std::to_string(out_bytes) + " bytes, the sequences need " + std::to_string(need));
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        decompress_device(g->g, g->seq_lens, out);
    });
}
// The hand-written scan / radix sort / comparator sort (device_prims.hpp) against the host's std:: algorithms — test hook.
int ac_selftest_primitives(int device, uint64_t n, uint64_t seed, int end_bit, int key_kind) {
    return guarded([&] {
        if (end_bit < 1 || end_bit > 64) throw DeviceError("end_bit out of range");
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        primitives_selftest(n, seed, end_bit, key_kind);
    });
}

int ac_random_access_ceilings_at(int device, uint64_t table_slots, double* cas_gops, double* read_gops) {
    return guarded([&] {
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        random_access_ceilings(cas_gops, read_gops, table_slots);
    });
}
int ac_random_access_ceilings(int device, double* cas_gops, double* read_gops) { return ac_random_access_ceilings_at(device, (uint64_t)1 << 24, cas_gops, read_gops); }

// The whole `autocycler decompress` command (decompress.rs:27-39): GFA file -> the assemblies it was built from, one file per
// original filename in out_dir (gzip when the name ends in .gz, decompress.rs:83-105) and / or all contigs in one FASTA file
// (headers ">{filename}__{header}", :117-137).  Either of out_dir / out_file may be NULL, not both.
int ac_decompress(const char* in_gfa, const char* out_dir, const char* out_file, int threads) {
    return guarded([&] {
        namespace fs = std::filesystem;
        if (!in_gfa || !fs::is_regular_file(in_gfa)) throw UserError(std::string("file does not exist: ") + (in_gfa ? in_gfa : ""));
        if (!out_dir && !out_file) throw UserError("either --out_dir or --out_file is required");
        if (out_dir && fs::exists(out_dir) && !fs::is_directory(out_dir)) throw UserError(std::string(out_dir) + " exists but is not a directory");
        std::string text;
        {
            std::ifstream f(in_gfa, std::ios::binary);
            text.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        }
        ac_graph h;
        std::vector<SeqMeta> meta;
        load_gfa(text.data(), text.size(), &h.g, &meta);
        const size_t S = meta.size();
        std::vector<std::string> seqs(S);
        {   // reconstruct_original_sequences (unitig_graph.rs:362-388), one sequence per task
            std::atomic<size_t> next{0};
            auto worker = [&] { for (size_t i; (i = next.fetch_add(1)) < S;) { seqs[i].resize(meta[i].length); decompress_sequence(h.g, i, seqs[i].data()); } };
            int T = std::max(1, std::min<int>(threads, (int)S));
            std::vector<std::thread> pool;
            for (int t = 1; t < T; t++) pool.emplace_back(worker);
            worker();
            for (auto& t : pool) t.join();
        }
        std::map<std::string, std::vector<size_t>> by_file;      // filenames sorted; contigs in GFA order within a file
        for (size_t i = 0; i < S; i++) by_file[meta[i].filename].push_back(i);
        if (out_dir) {
            std::error_code ec;
            fs::create_directories(out_dir, ec);
            if (ec) throw UserError(std::string("failed to create directory ") + out_dir + "\n" + ec.message());
            for (auto& [fname, idx] : by_file) {
                fs::path path = fs::path(out_dir) / fname;
                std::string body;
                for (size_t i : idx) { body += ">" + meta[i].contig_header + "\n"; body += seqs[i]; body += "\n"; }
                if (path.extension() == ".gz") {
                    gzFile gz = gzopen(path.c_str(), "wb");
                    if (!gz) throw UserError("failed to create " + path.string());
                    size_t o = 0;
                    while (o < body.size()) { int n = gzwrite(gz, body.data() + o, (unsigned)std::min<size_t>(body.size() - o, 1u << 30)); if (n <= 0) { gzclose(gz); throw UserError("failed to write " + path.string()); } o += (size_t)n; }
                    gzclose(gz);
                } else {
                    std::ofstream f(path, std::ios::binary);
                    f.write(body.data(), (std::streamsize)body.size());
                    if (!f) throw UserError("failed to write " + path.string());
                }
            }
        }
        if (out_file) {
            std::ofstream f(out_file, std::ios::binary);
            for (auto& [fname, idx] : by_file) {
                std::string clean = fname;
                std::replace(clean.begin(), clean.end(), ' ', '_');
                for (size_t i : idx) { f << ">" << clean << "__" << meta[i].contig_header << "\n"; f.write(seqs[i].data(), (std::streamsize)seqs[i].size()); f << "\n"; }
            }
            if (!f) throw UserError(std::string("failed to write ") + out_file);
        }
    });
}

uint64_t ac_kmer_count(const ac_graph* g) { return g->g.n_kmers; }
ac_stats ac_stats_pre(const ac_graph* g) { return ac_stats{g->g.pre.unitigs, g->g.pre.links_one_way, g->g.pre.total_length}; }
ac_stats ac_stats_post(const ac_graph* g) { return ac_stats{g->g.post.unitigs, g->g.post.links_one_way, g->g.post.total_length}; }
uint32_t ac_unitig_count(const ac_graph* g) { return g->g.n_unitigs; }

int ac_unitig(const ac_graph* g, uint32_t idx, const uint8_t** seq, uint32_t* len, double* depth) {
    if (idx >= g->g.n_unitigs) { g_err = "unitig index out of range"; return 1; }
    if (!g->host_arrays) { g_err = "this rank kept no host arrays (sharded build, not the writing rank)"; return 1; }
    if (seq) *seq = (const uint8_t*)g->g.seq(idx);
    if (len) *len = g->g.seq_len[idx];
    if (depth) *depth = g->g.depth[idx];
    return 0;
}
int ac_unitig_positions(ac_graph* g, uint32_t idx, int forward, const ac_position** positions, uint32_t* n) {
    if (idx >= g->g.n_unitigs) { g_err = "unitig index out of range"; return 1; }
    return guarded([&] {
        if (!g->host_arrays || !g->host_paths) throw DeviceError("this rank kept no host arrays (sharded build, not the writing rank)");
        if (!g->positions_built) { build_positions(&g->g, g->seq_ids, g->seq_lens); g->positions_built = true; }
        auto& v = forward ? g->g.fwd_positions[idx] : g->g.rev_positions[idx];
        static_assert(sizeof(ac_position) == sizeof(Position), "layout");
        *positions = (const ac_position*)v.data();
        *n = (uint32_t)v.size();
    });
}
int ac_links(const ac_graph* g, const ac_link** links, uint64_t* n) {
    static_assert(sizeof(ac_link) == sizeof(Link), "layout");
    if (!g->host_arrays) { g_err = "this rank kept no host arrays (sharded build, not the writing rank)"; return 1; }
    *links = (const ac_link*)g->g.links;
    *n = g->g.n_links;
    return 0;
}
int ac_path(const ac_graph* g, uint32_t seq_index, const int32_t** signed_unitigs, uint32_t* n) {
    if ((size_t)seq_index + 1 >= g->g.path_off.size()) { g_err = "sequence index out of range"; return 1; }
    if (!g->host_paths) { g_err = "this rank kept no paths on the host (sharded build)"; return 1; }
    uint64_t b = g->g.path_off[seq_index], e = g->g.path_off[seq_index + 1];
    *signed_unitigs = g->g.path + b;
    *n = (uint32_t)(e - b);
    return 0;
}
int ac_unitigs_bulk(const ac_graph* g, const uint8_t** seq_bytes, const uint64_t** seq_begin, const uint32_t** seq_len, const double** depth) {
    if (!g->host_arrays) { g_err = "this rank kept no host arrays (sharded build, not the writing rank)"; return 1; }
    if (seq_bytes) *seq_bytes = (const uint8_t*)g->g.seq_block.p;
    if (seq_begin) *seq_begin = g->g.seq_begin;
    if (seq_len) *seq_len = g->g.seq_len;
    if (depth) *depth = g->g.depth;
    return 0;
}
int ac_paths_bulk(const ac_graph* g, const int32_t** path_entries, const uint64_t** path_off, uint64_t* n_entries) {
    if (!g->host_paths) { g_err = "this rank kept no paths on the host (sharded build)"; return 1; }
    if (path_entries) *path_entries = g->g.path;
    if (path_off) *path_off = g->g.path_off.data();
    if (n_entries) *n_entries = g->g.n_path;
    return 0;
}
int ac_timings_get(const ac_graph* g, ac_timings* o) {
    const BuildTimings& t = g->tm;
    o->h2d = t.h2d; o->pack = t.pack; o->insert = t.insert; o->collect_sort = t.collect_sort; o->degree = t.degree;
    o->segment = t.segment; o->minkey = t.minkey; o->rank = t.rank; o->paths = t.paths; o->links = t.links; o->seqs = t.seqs;
    o->d2h = t.d2h; o->total_device = t.total_device; o->expand = t.expand;
    o->insert_kernel_ms = t.insert_kernel_ms; o->insert_positions = t.insert_positions;
    o->table_capacity = t.table_capacity; o->n_distinct = t.n_distinct; o->n_path_entries = t.n_path_entries;
    o->simplify_passes = t.simplify_passes; o->n_candidates = t.n_candidates; o->n_levels = t.n_levels;
    o->insert_launches = t.insert_launches; o->insert_real = t.insert_real;
    o->analysis = t.analysis; o->finalize = t.finalize;
    o->fragments = t.fragments; o->union_pack = t.union_pack; o->union_insert = t.union_insert;
    o->n_local_distinct = t.n_local_distinct; o->n_fragments = t.n_fragments; o->fragment_bytes = t.fragment_bytes;
    o->upload_device_ms = t.upload_device_ms;
    o->path_runs_copied = t.path_runs_copied; o->path_entries_walked = t.path_entries_walked; o->position_retries = t.position_retries;
    o->n_candidates_owned = t.n_candidates_owned;
    o->launches = t.launches; o->readbacks = t.readbacks; o->n_degrees_open = t.n_degrees_open; o->sort_retries = t.sort_retries;
    o->insert_rest_known = t.insert_rest_known; o->insert_rest_sampled = t.insert_rest_sampled; o->path_stretches = t.path_stretches;
    o->expand_sparse_sweeps = t.expand_sparse_sweeps; o->expand_sparse_start = t.expand_sparse_start;
    return 0;
}
// The same for a caller that was compiled against another version of the header: at most out_size bytes are written (the struct only
// ever grows at its end), the library's own size is returned.
size_t ac_timings_get_sized(const ac_graph* g, ac_timings* out, size_t out_size) {
    ac_timings t;
    memset(&t, 0, sizeof t);
    ac_timings_get(g, &t);
    if (out) memcpy(out, &t, std::min(out_size, sizeof t));
    return sizeof t;
}
void ac_free(ac_graph* g) { delete g; }

static int gfa_parts_impl(const ac_graph* g, int parts, const char* const* filenames, const char* const* headers, char** out,
                          uint64_t* out_len) {
    return guarded([&] {
        if ((parts & 1) && !g->host_arrays) throw DeviceError("this rank kept no unitigs / links on the host (sharded build)");
        if ((parts & 2) && !g->host_paths) throw DeviceError("this rank kept no paths on the host (sharded build)");
        std::vector<SeqMeta> meta(g->seq_ids.size());
        for (size_t i = 0; i < meta.size(); i++) meta[i] = SeqMeta{g->seq_ids[i], g->seq_lens[i], filenames[i], headers[i]};
        std::string s = gfa_string(g->g, meta, parts);
        char* p = (char*)malloc(s.size() + 1);
        if (!p) throw DeviceError("out of memory");
        memcpy(p, s.data(), s.size()); p[s.size()] = 0;
        *out = p;
        if (out_len) *out_len = s.size();
    });
}
int ac_gfa_string(const ac_graph* g, const char* const* filenames, const char* const* headers, char** out, uint64_t* out_len) {
    return gfa_parts_impl(g, 3, filenames, headers, out, out_len);
}
int ac_gfa_string_parts(const ac_graph* g, int parts, const char* const* filenames, const char* const* headers, char** out,
                        uint64_t* out_len) {
    return gfa_parts_impl(g, parts, filenames, headers, out, out_len);
}
void ac_string_free(char* p) { free(p); }

// ---- host side: load_sequences / end repair / whole command ---------------------------------------------
// sequence_end_repair (compress.rs:202-270) for sequences that live in host memory (ac_seqs_load, ac_seqs_from_raw, the multi-device
// command): the padded sequences go up as one text, the device kernels repair it in place (neighbours.inc — the only implementation
// the library has), and the k - 1 leading / trailing bytes of every sequence come back.
static void repair_on_device(LoadResult& lr, uint32_t k, int device) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<LoadedSeq>& seqs = lr.seqs;
    const size_t m = (size_t)k - 1;
    if (m == 0 || seqs.empty()) return;
    std::vector<SeqView> v(seqs.size());
    for (size_t i = 0; i < seqs.size(); i++) v[i] = SeqView{(const uint8_t*)seqs[i].forward_seq.data(), seqs[i].length};
    std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint16_t> d1, d2;
    std::vector<uint8_t> text = layout_text(v, k, &off, &len, &d1, &d2);
    {
        std::lock_guard<std::mutex> lock(g_build_mutex);
        if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
        select_device(device);
        RepairTimings rt;
#ifdef AC_EMU
        end_repair_device(k, text.data(), text.size(), off, len, &d1, &d2, &rt);
#else
        void* d_text = nullptr;
        AC_HIP_CHECK(hipMalloc(&d_text, text.size() + 64));
        struct Free { void* p; ~Free() { (void)hipFree(p); } } fr{d_text};
        AC_HIP_CHECK(hipMemcpy(d_text, text.data(), text.size(), hipMemcpyHostToDevice));
        end_repair_device(k, (uint8_t*)d_text, text.size(), off, len, &d1, &d2, &rt);
        // only the two ends of a sequence can have changed: k - 1 bytes from either end of each come back (many short sequences: the whole text)
        if (seqs.size() > 256) AC_HIP_CHECK(hipMemcpy(text.data(), d_text, text.size(), hipMemcpyDeviceToHost));
        else for (size_t i = 0; i < seqs.size(); i++) {
            const size_t plen = seqs[i].forward_seq.size();
            AC_HIP_CHECK(hipMemcpyAsync(&text[off[i]], (const uint8_t*)d_text + off[i], std::min(m, plen), hipMemcpyDeviceToHost, 0));
            if (plen > m) AC_HIP_CHECK(hipMemcpyAsync(&text[off[i] + plen - m], (const uint8_t*)d_text + off[i] + plen - m, m, hipMemcpyDeviceToHost, 0));
        }
        AC_HIP_CHECK(hipStreamSynchronize(0));
#endif
    }
    for (size_t i = 0; i < seqs.size(); i++) {
        std::string& f = seqs[i].forward_seq;
        const size_t plen = f.size(), head = std::min(m, plen);
        memcpy(&f[0], &text[off[i]], head);
        if (plen > m) memcpy(&f[plen - m], &text[off[i] + plen - m], m);
    }
    lr.repair_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int ac_seqs_load(const char* assemblies_dir, uint32_t k, uint32_t max_contigs, int threads, ac_seqs** out) {
    return guarded([&] {
        auto h = std::make_unique<ac_seqs>();
        h->lr = load_sequences(assemblies_dir, k, max_contigs, threads);
        repair_on_device(h->lr, k, g_host_side_device.load());
        h->make_views();
        *out = h.release();
    });
}
int ac_seqs_from_raw(uint32_t k, uint32_t n, const uint8_t* const* seqs, const uint32_t* lens, const char* const* filenames,
                     const char* const* headers, uint32_t assembly_count, int repair, int threads, ac_seqs** out) {
    return guarded([&] {
        auto h = std::make_unique<ac_seqs>();
        if (n > 32767) throw UserError("no more than 32767 input sequences are allowed");
        for (uint32_t i = 0; i < n; i++) {
            LoadedSeq s;
            s.id = (uint16_t)(i + 1);
            s.filename = filenames ? filenames[i] : ("assembly_" + std::to_string(i) + ".fasta");
            s.contig_header = headers ? headers[i] : ("contig_" + std::to_string(i + 1));
            pad_sequence(&s, std::string((const char*)seqs[i], lens[i]), k);
            h->lr.seqs.push_back(std::move(s));
        }
        h->lr.assembly_count = assembly_count;
        h->lr.total_contigs_seen = n;
        (void)threads;      // (the packers of a later build use their own pool; the repair is a device kernel)
        if (repair) repair_on_device(h->lr, k, g_host_side_device.load());
        h->make_views();
        *out = h.release();
    });
}
uint32_t ac_seqs_count(const ac_seqs* s) { return (uint32_t)s->lr.seqs.size(); }
uint32_t ac_seqs_assembly_count(const ac_seqs* s) { return s->lr.assembly_count; }
const ac_seq_view* ac_seqs_views(const ac_seqs* s) { return s->views.data(); }
int ac_seqs_get(const ac_seqs* s, uint32_t i, ac_seq_view* view, const char** filename, const char** header) {
    if (i >= s->lr.seqs.size()) { g_err = "sequence index out of range"; return 1; }
    if (view) *view = s->views[i];
    if (filename) *filename = s->lr.seqs[i].filename.c_str();
    if (header) *header = s->lr.seqs[i].contig_header.c_str();
    return 0;
}
double ac_seqs_repair_seconds(const ac_seqs* s) { return s->lr.repair_seconds; }
int ac_seqs_metrics_yaml(const ac_seqs* s, uint32_t unitig_count, uint64_t unitig_total_length, char** out) {
    return guarded([&] {
        std::string y = metrics_yaml(s->lr, unitig_count, unitig_total_length);
        char* p = (char*)malloc(y.size() + 1);
        if (!p) throw DeviceError("out of memory");
        memcpy(p, y.data(), y.size() + 1);
        *out = p;
    });
}
void ac_seqs_free(ac_seqs* s) { delete s; }

int ac_compress_seqs(uint32_t k, const ac_seqs* s, int device, ac_graph** out) {
    return ac_compress_build(k, s->lr.assembly_count, s->views.data(), (uint32_t)s->views.size(), device, out);
}

// compress.rs:32-50: the whole `autocycler compress` command.  times[4] = load, repair, graph (hot path), write.
int ac_compress_dir(const char* assemblies_dir, const char* autocycler_dir, uint32_t k, uint32_t max_contigs, int threads,
                    int device, ac_graph** graph_out, double* times) {
    return guarded([&] {
        namespace fs = std::filesystem;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        check_compress_settings(assemblies_dir, autocycler_dir, k, threads);
        std::error_code ec;
        fs::create_directories(autocycler_dir, ec);
        if (ec) throw UserError(std::string("failed to create directory ") + autocycler_dir + "\n" + ec.message());
        // load (host) -> text layout -> H2D -> end repair on the device text -> graph build from the same buffer
        // the HIP context and the code objects come up on another thread while the host reads the FASTA files
        // (with the arena and the upload ring it will need, sized from the files' sizes: a .gz holds about four times its size in bases)
        uint64_t est = 0;
        {
            std::error_code ec2;
            for (auto& e : fs::directory_iterator(assemblies_dir, ec2)) {
                if (!e.is_regular_file(ec2)) continue;
                const uint64_t sz = (uint64_t)e.file_size(ec2);
                est += e.path().extension() == ".gz" ? 4 * sz : sz;
            }
        }
        // (the helper thread also notes which device the arena it reserves lives on — select_device — so that the build's own
        // select_device, after the join below, finds it in place; nothing else runs in this process meanwhile: the CLI's only call)
        std::thread warm([device, k, est] {
            try {
                { std::lock_guard<std::mutex> lock(g_build_mutex); if (g_live_shards) return; select_device(device); }
                device_warmup(device, k, est + (est >> 4) + (1u << 20));
            } catch (...) {}
        });
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm};
        ac_seqs s;
        s.lr = load_sequences(assemblies_dir, k, max_contigs, threads);
        s.make_views();
        warm.join();
        double t0 = now();
        ac_graph* g = nullptr;
        {
            const uint32_t n = (uint32_t)s.views.size();
            validate(k, s.views.data(), n);
            std::lock_guard<std::mutex> lock(g_build_mutex);
            if (g_live_shards) throw DeviceError("a sharded build is in flight in this process");
            select_device(device);
            auto h = std::make_unique<ac_graph>();
            std::vector<SeqView> v(n);
            for (uint32_t i = 0; i < n; i++) {
                v[i] = SeqView{s.views[i].fwd, s.views[i].length};
                h->seq_ids.push_back(s.views[i].id);
                h->seq_lens.push_back(s.views[i].length);
            }
            // pinned-ring upload of the padded sequences -> end repair in place on the device text -> pack + build from the same buffer
            GraphBuilder b(k);
            b.set_sequences_host(v, /*pack_now=*/false);
            RepairTimings rt;
            b.repair_ends(&rt);
            s.lr.repair_seconds = now() - t0;        // upload of the text + the repair itself
            t0 = now();
            build_graph(b, s.lr.assembly_count, h.get());
            g = h.release();
        }
        std::unique_ptr<ac_graph> guard(g);
        double t1 = now();
        std::vector<SeqMeta> meta(s.lr.seqs.size());
        for (size_t i = 0; i < meta.size(); i++) meta[i] = SeqMeta{s.lr.seqs[i].id, s.lr.seqs[i].length, s.lr.seqs[i].filename, s.lr.seqs[i].contig_header};
        write_pieces((fs::path(autocycler_dir) / "input_assemblies.gfa").string(), gfa_chunks(g->g, meta, threads), threads);
        {
            std::string y = metrics_yaml(s.lr, g->g.post.unitigs, g->g.post.total_length);
            std::ofstream f(fs::path(autocycler_dir) / "input_assemblies.yaml", std::ios::binary);
            f.write(y.data(), (std::streamsize)y.size());
        }
        double t2 = now();
        if (times) { times[0] = s.lr.load_seconds; times[1] = s.lr.repair_seconds; times[2] = t1 - t0; times[3] = t2 - t1; }
        if (graph_out) *graph_out = guard.release();
    });
}

// The whole command over several devices: load + end repair on the host (the reference's own order, compress.rs:38-41), then ONE
// ac_compress_build_multi call, then the same writers.
int ac_compress_dir_multi(const char* assemblies_dir, const char* autocycler_dir, uint32_t k, uint32_t max_contigs, int threads,
                          const int* devices, int n_devices, ac_graph** graph_out, double* times) {
    return guarded([&] {
        namespace fs = std::filesystem;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        check_compress_settings(assemblies_dir, autocycler_dir, k, threads);
        if (!devices || n_devices < 1) throw DeviceError("ac_compress_dir_multi: no devices");
        std::error_code ec;
        fs::create_directories(autocycler_dir, ec);
        if (ec) throw UserError(std::string("failed to create directory ") + autocycler_dir + "\n" + ec.message());
        ac_seqs s;
        s.lr = load_sequences(assemblies_dir, k, max_contigs, threads);
        repair_on_device(s.lr, k, devices[0]);      // (compress.rs:38-41: load, then repair, then the build — the repair on the first rank's device)
        s.make_views();
        double t0 = now();
        ac_graph* g = nullptr;
        if (ac_compress_build_multi(k, s.lr.assembly_count, s.views.data(), (uint32_t)s.views.size(), devices, n_devices, &g) != 0) throw DeviceError(g_err);
        std::unique_ptr<ac_graph> guard(g);
        double t1 = now();
        std::vector<SeqMeta> meta(s.lr.seqs.size());
        for (size_t i = 0; i < meta.size(); i++) meta[i] = SeqMeta{s.lr.seqs[i].id, s.lr.seqs[i].length, s.lr.seqs[i].filename, s.lr.seqs[i].contig_header};
        write_pieces((fs::path(autocycler_dir) / "input_assemblies.gfa").string(), gfa_chunks(g->g, meta, threads), threads);
        {
            std::string y = metrics_yaml(s.lr, g->g.post.unitigs, g->g.post.total_length);
            std::ofstream f(fs::path(autocycler_dir) / "input_assemblies.yaml", std::ios::binary);
            f.write(y.data(), (std::streamsize)y.size());
        }
        double t2 = now();
        if (times) { times[0] = s.lr.load_seconds; times[1] = s.lr.repair_seconds; times[2] = t1 - t0; times[3] = t2 - t1; }
        if (graph_out) *graph_out = guard.release();
    });
}

}  // extern "C"
