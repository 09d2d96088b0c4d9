// Device graph build for the `autocycler compress` hot path: k-mer set -> unitigs -> links -> paths.
// Replaces KmerGraph::add_sequences (kmer_graph.rs:86-134), UnitigGraph::build_unitigs_from_kmer_graph
// + simplify_seqs + create_links' neighbour discovery + trim_overlaps (unitig_graph.rs:176-293) and the
// per-sequence path walk (unitig_graph.rs:407-465).  Output is the compacted graph in *seed order*
// (reference's initial unitig numbering: rank of each unitig's smallest k-mer, unitig_graph.rs:176-226
// with kmer_graph.rs:168-173); the order-dependent tail (link push order, renumber, expand_repeats) is
// host_tail.cpp.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "graph_types.hpp"

namespace ac {

struct SeqView {          // one padded, end-repaired forward sequence as at compress.rs:41
    const uint8_t* fwd;   // length + k - 1 bytes over ".ACGT"
    uint32_t length;      // unpadded length (number of forward k-mers)
};

struct BuildTimings {                   // seconds; device stages are bracketed by stream syncs
    double h2d = 0, pack = 0, insert = 0, collect_sort = 0, degree = 0, segment = 0, minkey = 0, rank = 0,
           paths = 0, links = 0, seqs = 0, d2h = 0, total_device = 0;
    double expand = 0;                  // expand_repeats passes on the device (level-scheduled)
    double analysis = 0;                // link push order + expand_repeats candidates + first renumber (device)
    double finalize = 0;                // second renumber + final numbering of links and paths (device)
    double insert_kernel_ms = 0;        // event-timed duration of the dominant kernel (k-mer insert)
    uint64_t insert_positions = 0;      // text positions streamed by that launch
    uint64_t table_capacity = 0;
    uint64_t n_distinct = 0;
    uint64_t n_path_entries = 0;
    uint32_t insert_launches = 0;       // phases of the run-following insert (same kernel, launched per phase)
    uint64_t insert_real = 0;           // positions that actually touched the table
    uint32_t simplify_passes = 0, n_candidates = 0, n_levels = 0;   // expand_repeats: passes, candidate junctions, conflict levels
    uint32_t n_candidates_owned = 0;    // ... and the candidate junctions THIS rank ran (a sharded build with a partitioned tail; else all of them)
    // sharded builds (one job over several devices)
    uint32_t local_hint = 0, graph_hint = 0;   // capacity hints of the local / graph table (assembly counts)
    double fragments = 0;               // novel runs of this rank -> fragment text
    double union_pack = 0, union_insert = 0;   // packing / inserting the union of all ranks' fragments
    uint64_t n_local_distinct = 0, n_fragments = 0, fragment_bytes = 0;
    double upload_device_ms = 0;        // host entry: first copy issued -> last chunk landed and packed (HIP events)
    double insert_rest_known = 0;       // share of a sample of the insert's one-launch rest found in the table after the first two stretches (sizes its chunks)
    uint64_t path_stretches = 0;        // the paths crossed to the host as this many stretches of consecutive text-order numbers (0: as entries)
    uint32_t expand_sparse_sweeps = 0;  // passes of expand_repeats that the one-workgroup tail ran from the list of dirty junctions (kernels_tail.inc expand_mopup_kernel)
    uint32_t expand_sparse_start = 0;   // ... and the dirty junctions on that list when it took over
    double insert_rest_sampled = 0;     // ... and the share of that rest the sample could cover (the text that was on the device when it was taken)
    uint32_t sort_retries = 0;          // builds repeated with checked sorts (a deferred "group too large" flag of the seed sort / a renumbering was set)
    uint32_t position_retries = 0;      // builds repeated with exact smallest positions (AC_POS_CAP; kernels_tail.inc exp_avoid_start_of_path)
    uint32_t launches = 0, readbacks = 0;   // kernel / fill launches and host round trips (mailbox, synchronising copies) of this build on its main stream
    uint64_t n_degrees_open = 0;        // sharded builds: k-mers the light degree step left to the probes (the compact degree exchange's size)
    uint64_t path_runs_copied = 0, path_entries_walked = 0;   // K10c: followed runs whose path entries were copied / entries that were really walked (0 / 0: the plain walk)
};

// sequence_end_repair (compress.rs:202-270) on the device.  d_text: the PADDED, unrepaired sequences in the text layout below;
// it is patched in place and d1 / d2 (surviving dots) are updated.
struct RepairTimings { double total = 0, scan_ms = 0; uint64_t hits = 0, matches = 0; uint32_t patterns = 0; };

class GraphBuilder {
  public:
    explicit GraphBuilder(uint32_t k);
    ~GraphBuilder();
    // Host entry: lays the sequences out as one text ('$' separators) chunk by chunk in a pinned staging ring and streams it to
    // the device; pack_now: K1 packs every chunk right behind its copy (the sequences are final, i.e. end-repaired already).
    void set_sequences_host(const std::vector<SeqView>& seqs, bool pack_now = true);
    // sequence_end_repair (compress.rs:202-270) on the text set_sequences_host(seqs, false) uploaded: patches it in place and
    // updates the surviving-dot counts; the build then packs the repaired text.
    void repair_ends(RepairTimings* tm);
    // Device entry: `d_text` is an ASCII text already resident in HBM with the same layout:
    // text[0] = '$', then for each sequence its padded bytes followed by one '$'.
    // off[s] = index of the first padded byte of sequence s.  d1/d2 = leading/trailing dot counts.
    void set_text_device(const uint8_t* d_text, uint64_t n_text, const std::vector<uint64_t>& off,
                         const std::vector<uint32_t>& len, const std::vector<uint16_t>& d1,
                         const std::vector<uint16_t>& d2);
    // The timed region (compress.rs:42-44): packed text -> final UnitigGraph in host memory.  Everything runs on
    // the device, including expand_repeats and both renumberings; the host only receives the results.
    void build(uint32_t assembly_count_hint, FinalGraph* out);

    // One job sharded by sequence over several devices, the k-mer table partitioned by key hash.  This rank's
    // sequences are set with set_text_device; the collectives between the phases are the caller's (torch.distributed / RCCL).
    //   1. shard_begin: pack + insert this rank's sequences into a LOCAL table, cut its novel runs out as "fragments".
    //      -> all-gather the fragment texts and meta records of all ranks (rank order) = the union text, on every rank.
    //   2. shard_build_union: this rank inserts the union-text k-mers it OWNS (owner = home hash mod n_shards) -> its share of
    //      the novel bitmap.  -> all-reduce SUM of the bitmaps (the owners partition the keys: the shares are disjoint).
    //   3. shard_build_novel: novel list (identical everywhere); degrees + first flags of ALL novel k-mers, probing only owned
    //      groups.  -> all-reduce SUM of the degree words.
    //   4. shard_build_graph: unitigs in seed order (identical everywhere); links, probing only owned groups.
    //      -> all-reduce SUM of the link words.
    //   5. links_import: the keys this rank's walkers start from.  -> all-gather of the keys; answer_queries (owned ones) on
    //      the keys of all ranks; all-reduce SUM of the answers.
    //   6. shard_walk: the paths of this rank's sequences.  -> all-reduce (SUM / MIN) the per-unitig buffers of reduce_export.
    //   7. reduce_import + shard_finish: the order-sensitive tail (identical on every rank); paths of this rank's sequences in
    //      final numbers (kept per rank, or gathered with paths_export to the rank that writes the GFA).
    void shard_begin(uint32_t local_assembly_hint);
    uint64_t local_distinct_count() const;                          // distinct canonical k-mers of this rank's slice
    void set_distinct_upper_bound(uint64_t n);                      // optional: sum of all ranks' local counts sizes the owned tables
    uint64_t fragment_text_bytes() const;
    uint64_t fragment_count() const;
    void fragments_export(void* d_text_out, void* d_meta_out);      // device buffers: text bytes, 8 bytes per fragment
    // The same text as 2-bit codes on the union text's word grid — a quarter of the bytes, and nothing to pack on the receiving side.
    // union_off = where this rank's stretch begins in the union text ('$' + the ranks' fragment texts in rank order).
    uint64_t fragment_packed_words(uint64_t union_off) const;
    void fragments_export_packed(uint64_t union_off, void* d_words_out, void* d_meta_out);
    // d_staged_words: the ranks' word stretches one behind the other (n_words[r] words of rank r, which begin at union word first_word[r])
    void shard_build_union_packed(uint32_t rank, uint32_t n_shards, const void* d_staged_words, const uint64_t* first_word, const uint64_t* n_words,
                                  uint64_t n_union_text, const void* d_meta, uint64_t n_frags_total);
    void shard_build_union(uint32_t rank, uint32_t n_shards, const uint8_t* d_union_text, uint64_t n_union_text,
                           const void* d_meta, uint64_t n_frags_total);
    uint64_t bitmap_words() const;                                  // u64 words of the union text's novel bitmap
    void bitmap_export(void* d_out);
    void shard_build_novel(const void* d_bitmap_sum);               // nullptr: single rank
    // round 5: this rank's sibling bits by novel index (2 bits per distinct k-mer); sib_words() == 0: not in use, the degree stage has run
    uint64_t sib_words() const;
    void sib_export(void* d_out);
    void shard_degrees(const void* d_sib_sum);                      // the degree stage, with the ranks' summed sibling bits
    uint64_t degree_bytes() const;                                  // size of the degree exchange (compact: the k-mers the light step left open)
    uint64_t distinct_count() const;                                // N: distinct canonical k-mers of the whole job
    void degrees_export(void* d_out);                               // degree_bytes() bytes: this rank's contributions
    void shard_build_graph(const void* d_kinfo_sum);                // degree_bytes() bytes (nullptr: single rank)
    uint32_t unitig_count() const;
    void links_export(void* d_links_i32, void* d_wlinks_i64);       // 10 U words each: this rank's contributions
    void links_import(const void* d_links_i32, const void* d_wlinks_i64);   // summed (nullptr, nullptr: single rank)
    uint64_t query_count() const;                                   // walk queries of this rank
    uint32_t query_key_words() const;                               // u64 words per query key
    void queries_export(void* d_out);
    void answer_queries(const void* d_keys, uint64_t n, void* d_out);   // n keys of any ranks -> n u64 (0 where this rank does not own the key)
    void shard_walk(const void* d_answers_mine);                    // query_count() u64
    // The same exchange routed by owner (one all-to-all each way instead of all-gather + SUM): the queries in owner order (stable),
    // counts_host[o] = how many of them rank o's table answers; the owners' answers come back in that order.
    void queries_route(uint32_t n_shards, void* d_routed_keys, uint64_t* counts_host);
    void shard_walk_routed(const void* d_routed_answers);
    void reduce_export(int32_t* d_sum, int32_t* d_min);             // 3U and 2U int32
    void reduce_import(const int32_t* d_sum, const int32_t* d_min);
    // Before shard_finish (optional, the same choice on every rank): an in-place all-reduce of a device buffer over the ranks (dtype 0 =
    // uint8, 1 = int32; op 0 = SUM, 1 = MIN).  With it expand_repeats runs on this rank's share of the junctions only — the conflict
    // components it owns — and the ranks' results are merged by two SUM all-reduces (field lengths, then the sequence bytes).
    void set_tail_exchange(std::function<void(void*, uint64_t, int, int)> all_reduce);
    void shard_finish(FinalGraph* out, bool want_graph, bool want_paths);
    uint64_t path_entry_count() const;
    void paths_export(void* d_out);                                 // int32 per entry, final numbers
    const BuildTimings& timings() const { return tm_; }
    void set_sequence_index_base(uint64_t n);   // a rank of a multi-device build: how many sequences of the job precede its slice (error messages)
    static void set_upload_threads_cap(int n);  // for the calling thread's builds: at most n packing threads (the ranks of a multi-device build share the host)
    uint64_t n_text() const;
    uint64_t n_bases() const;   // sum of unpadded lengths

    struct Impl;   // all device state of one build (graph_build.hip)

  private:
    void build_union_impl(uint32_t rank, uint32_t n_shards, const uint8_t* d_union_text, const void* d_staged_words, const uint64_t* first_word,
                          const uint64_t* n_words, uint64_t n_union_text, const void* d_meta, uint64_t n_frags_total);
    void upload_packed(const std::vector<SeqView>& seqs, const std::vector<uint64_t>& off);
    Impl* impl_;
    BuildTimings tm_;
};

// Builds the text layout used by both entries.  Returns the text; fills off/len/d1/d2.
std::vector<uint8_t> layout_text(const std::vector<SeqView>& seqs, uint32_t k, std::vector<uint64_t>* off,
                                 std::vector<uint32_t>* len, std::vector<uint16_t>* d1, std::vector<uint16_t>* d2);

void end_repair_device(uint32_t k, uint8_t* d_text, uint64_t n_text, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len,
                       std::vector<uint16_t>* d1, std::vector<uint16_t>* d2, RepairTimings* tm, bool reset_arena = true);

// pairwise_contig_distances (cluster.rs:132-157) on the final graph: out[a * n_seqs + b], sequences in path order.
void pairwise_distances_device(const FinalGraph& g, uint32_t n_seqs, double* out);

// ac_verify_graph (kernels_verify.inc): the size-independent properties of a finished graph, checked on the device against the job's text.
struct VerifyReport {
    uint32_t failed = 0;                                   // bit mask of the checks that failed (VerifyFlag); 0 = the graph holds
    uint64_t first_bad_unitig = ~0ULL, first_bad_link = ~0ULL, first_bad_path_entry = ~0ULL, first_bad_sequence = ~0ULL, first_bad_base = ~0ULL;
    uint64_t unitigs = 0, links = 0, path_entries = 0, bases_checked = 0, self_mirror_links = 0;
    double seconds = 0;
    uint32_t checks = 0;                                   // VerifyCheck bits: which of the order-sensitive checks ran
    uint64_t first_bad_junction = ~0ULL;                   // 2 * unitig index + side (0 = its inputs, 1 = its outputs) that would still shift
};
void verify_graph_device(const FinalGraph& g, const uint8_t* d_text, uint64_t n_text, const std::vector<uint64_t>& off,
                         const std::vector<uint32_t>& len, VerifyReport* rep);

// reconstruct_original_sequences (unitig_graph.rs:362-400) for ALL sequences on the device: out_host holds sum(seq_len) bytes, sequence s behind s - 1.
void decompress_device(const FinalGraph& g, const std::vector<uint32_t>& seq_len, uint8_t* out_host);
// device_prims.hpp against std:: on n pseudo-random items (tests); throws on a mismatch.
void primitives_selftest(uint64_t n, uint64_t seed, int end_bit, int key_kind);

// Measured ceilings of the device for random atomicCAS / random 8-byte reads on a 134 MB table, in 10^9 operations per second.
void random_access_ceilings(double* cas_gops, double* read_gops, uint64_t table_slots = (uint64_t)1 << 24);

// Brings the HIP context and this library's code objects up on `device` (first use costs ~0.2 s): callable from a helper
// thread while the caller is still busy on the host.
void device_warmup(int device, uint32_t k = 0, uint64_t text_bytes_estimate = 0);

// K1 on the host (what the packed upload of set_sequences_host runs per piece): n_text bytes -> (n_text + 31) / 32 words of 2-bit
// codes (first base most significant) and as many 32-bit mask words (bit i = byte i is not a base).
void pack_text_host(const uint8_t* text, uint64_t n_text, uint64_t* bits, uint32_t* mask32, bool force_scalar);

// Frees the pinned staging ring of the host entry (192 MB; it otherwise stays for the next build of the process).
void release_host_stager();

int max_supported_k();
// The environment knobs are read once per process (graph_impl.hpp: Knobs); these are the ones code outside the graph build asks for.
void tuning_refresh();                       // re-reads them if AC_TUNING_FOLLOW_ENV was set (tests, tools/ab_knobs.py); called under the build lock
int tuning_multi_transport();                // AC_MULTI_TRANSPORT: 0 unset, 1 host, 2 rccl
bool tuning_multi_fragments_as_bytes();      // AC_MULTI_FRAGMENTS=bytes
bool tuning_multi_tail_replicated();         // AC_MULTI_TAIL=replicated
void set_stage_timing(bool on);   // per-stage timers (a stream sync per stage); off by default
bool stage_timing();

}  // namespace ac
