/* autocycler_hip.h — C ABI of libautocycler_hip.so, the MI355X (gfx950) drop-in for the hot path of
 * `autocycler compress` (rrwick/Autocycler v0.7.0).
 *
 * The reference has no FFI or plugin interface; the seam is cut in src/compress.rs:42-44:
 *
 *     let kmer_graph = build_kmer_graph(k_size, assembly_count, &sequences);   // KmerGraph::add_sequences, kmer_graph.rs:86-134
 *     let mut unitig_graph = build_unitig_graph(kmer_graph);                   // UnitigGraph::from_kmer_graph, unitig_graph.rs:36-48
 *     simplify_unitig_graph(&mut unitig_graph, &sequences);                    // simplify_structure, graph_simplification.rs:26-40
 *
 * ac_compress_build() replaces those three calls; the accessors hand back exactly what the consumers
 * (save_gfa unitig_graph.rs:317-331, save_metrics compress.rs:181-189, print_basic_graph_info
 * unitig_graph.rs:509-516) read.  INTEGRATION.md shows the Rust `extern "C"` block and the patch.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returning int returns 0 on
 * success and non-zero on failure; ac_last_error() then holds the text the Rust shim should pass to
 * quit_with_error (misc.rs:131-137).  The library never exits, aborts or unwinds across the ABI.
 * Inputs are only read during the call (the reference's raw pointers into Sequence buffers,
 * kmer_graph.rs:30,115, need not outlive it).  Results are owned by the handle until ac_free().
 * There is no CPU fallback: without a usable gfx950 device every build call fails with an error.
 */
#ifndef AUTOCYCLER_HIP_H
#define AUTOCYCLER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ac_graph ac_graph; /* opaque: the final unitig graph (UnitigGraph after simplify_structure) */

/* One loaded sequence as `load_sequences` returns it (compress.rs:98-133, sequence.rs:19-59). */
typedef struct {
    const uint8_t* fwd; /* Sequence::forward_seq: padded + end-repaired, length + k - 1 bytes over ".ACGT" */
    uint32_t length;    /* Sequence::length (unpadded) */
    uint16_t id;        /* Sequence::id (1-based; gaps allowed, compress.rs:111,120) */
} ac_seq_view;

typedef struct { uint32_t pos; uint16_t seq_id_and_strand; } ac_position; /* position.rs:18-22; strand = bit 15 */
/* A link a -> b (UnitigStrand pairs, unitig_graph.rs:234-287) as two SIGNED unitig numbers: +n = the forward strand of unitig n, -n = its
 * reverse strand — the form of the reference's own path entries (get_unitig_path_for_sequence_i32).  ABI 7: 8 bytes; ABI <= 6 carried
 * { uint32_t a; uint8_t a_fwd; uint32_t b; uint8_t b_fwd; } = 16 bytes with padding (the link array is the largest late result of a
 * mixed-species build and crosses PCIe behind everything else). */
typedef struct { int32_t a; int32_t b; } ac_link;
typedef struct { uint32_t unitigs; uint64_t links_one_way; uint64_t total_length; } ac_stats;

/* Seconds spent in each stage of the last build.  The per-stage fields are only filled while stage timing is on
 * (ac_set_stage_timing(1): one stream synchronisation per stage, ~0.3 ms per build); total_device, insert_kernel_ms and
 * the counts are always filled. */
typedef struct {
    double h2d, pack, insert, collect_sort, degree, segment, minkey, rank, paths, links, seqs, d2h;
    double total_device; /* pack .. d2h: the whole replaced region */
    double expand;       /* expand_repeats passes (device, level-scheduled) */
    double insert_kernel_ms;    /* HIP-event duration of the k-mer insert kernel, summed over its phase launches */
    uint64_t insert_positions;  /* text positions those launches streamed */
    uint64_t table_capacity, n_distinct, n_path_entries;
    uint32_t simplify_passes;
    uint32_t insert_launches;   /* phases of the run-following insert (launches of that kernel per build) */
    uint64_t insert_real;       /* positions that really accessed the k-mer table (the rest were run-followed) */
    double analysis;            /* link push order, expand_repeats candidates, first renumber */
    double finalize;            /* second renumber, final numbering of links and paths */
    uint32_t n_candidates;      /* junctions that pass the static tests of expand_repeats */
    uint32_t n_levels;          /* conflict levels they are scheduled in */
    /* sharded builds only */
    double fragments;           /* cutting this rank's novel runs out of its text */
    double union_pack, union_insert;   /* packing / inserting the union of all ranks' fragments */
    uint64_t n_local_distinct, n_fragments, fragment_bytes;
    double upload_device_ms;    /* ac_compress_build only: the upload from its first byte to its last chunk on the device (host clock when the packers
                                 * store into device memory themselves, HIP events around the copies of the pinned-ring path) */
    uint64_t path_runs_copied, path_entries_walked;   /* the copying path walk: runs whose entries were copied, entries really walked (0, 0: plain walk) */
    uint64_t position_retries;   /* builds repeated with exact smallest positions because expand_repeats met a common sequence longer than the bound kept (AC_POS_CAP) */
    uint32_t n_candidates_owned; /* the candidate junctions THIS rank ran (a job over several devices with a partitioned tail; else = n_candidates) */
    uint32_t launches;           /* kernel launches of this build on its main stream (fused fills count once per batch) */
    uint32_t readbacks;          /* host round trips of this build: small device -> host reads the host waited for */
    uint64_t n_degrees_open;     /* sharded builds: k-mers whose degrees the sibling bits did not settle (= bytes of the compact degree exchange) */
    uint64_t sort_retries;       /* builds repeated because a sort's "group too large" flag, read with the last read-back, was set */
    double insert_rest_known;    /* share of a sample of the insert's one-launch rest that the first two stretches already held (sizes the rest's chunks); 0 = no such rest */
    double insert_rest_sampled;  /* share of that rest the sample could cover: only text that was on the device when it was taken (an upload still in flight: its first chunks) */
    uint64_t path_stretches;     /* > 0: the path entries crossed to the host as this many stretches of consecutive text-order numbers (8 bytes each) and were written out there */
    uint32_t expand_sparse_sweeps; /* passes of expand_repeats run by the one-workgroup tail from the list of dirty junctions (0: level launches to the end) */
    uint32_t expand_sparse_start;  /* ... and the length of that list when it took over */
} ac_timings;

/* Replaces compress.rs:42-44.  k: --kmer (odd).  assembly_count: the reference's capacity hint
 * (kmer_graph.rs:40), used the same way (initial table sizing).  device: HIP device ordinal. */
int ac_compress_build(uint32_t k, uint32_t assembly_count, const ac_seq_view* seqs, uint32_t n_seqs, int device,
                      ac_graph** out);

/* The same over SEVERAL devices of one node, still one call from one process: devices[r] = HIP ordinal of rank r.  The
 * library runs one host thread per device; the sequences are sharded by rank (contiguous slices balanced by bases), the k-mer table is
 * partitioned over the ranks by key hash, and the exchanges between the phases of the build happen inside the library: RCCL over xGMI
 * (ncclAllReduce, and grouped ncclSend / ncclRecv that route every walk-start key to the one rank that owns it), loaded on first use.  A
 * device may be named more than once (those ranks then share it and the exchanges are staged through host memory: tests and dry runs);
 * AC_MULTI_TRANSPORT=host|rccl overrides the choice.  The graph is the one ac_compress_build builds (byte-identical GFA); n_devices = 1
 * is allowed.  Rust: the shim of INTEGRATION.md passes the ordinals it wants instead of one `device`. */
int ac_compress_build_multi(uint32_t k, uint32_t assembly_count, const ac_seq_view* seqs, uint32_t n_seqs, const int* devices,
                            int n_devices, ac_graph** out);
/* What the multi-device build behind a graph moved between its ranks (n_ranks = 0: a single-device build). */
typedef struct {
    uint32_t n_ranks; int transport;   /* transport: 1 = staged through host memory, 2 = RCCL, 3 = one rank: built as a single-device job */
    uint64_t bytes_fragments, bytes_bitmap, bytes_degrees, bytes_links, bytes_queries, bytes_answers, bytes_reduce;   /* received, all ranks */
    uint64_t queries_total, queries_sent_away;        /* walk-start queries of all ranks / those another rank answered */
    uint64_t table_capacity_max, table_capacity_sum;  /* slots of the ranks' shares of the job's k-mer table */
    uint64_t union_text_bytes, fragments, distinct;
    double seconds_total, seconds_exchange_max;
    /* expand_repeats partitioned by conflict component: candidate junctions of the job / the most any one rank ran */
    uint64_t candidates_total, candidates_owned_max;
    /* round 5 */
    uint64_t bytes_sibling;      /* the sibling bits (2 per distinct k-mer), received, all ranks */
    uint64_t bytes_tail;         /* the partitioned tail's merges (field lengths, sequence bytes) */
    uint64_t degrees_open;       /* k-mers the light degree step left to the probes (bytes_degrees is their exchange) */
    uint64_t bytes_received_max; /* the most any ONE rank received from the others over the whole build */
    uint64_t path_runs_copied;   /* pieces of followed runs the ranks' copying walks copied instead of walking (their local inserts note the runs; 0: all text walked) */
} ac_multi_info;
int ac_multi_info_get(const ac_graph*, ac_multi_info* out);
/* The same for a caller compiled against an older header: at most out_size bytes are written (the struct only grows at its end); returns
 * the library's own sizeof(ac_multi_info). */
size_t ac_multi_info_get_sized(const ac_graph*, ac_multi_info* out, size_t out_size);

/* ---- the round-trip verifier: what the reference's own tests hold a compress result to (tests.rs:108-127), as device
 * kernels over the result arrays of `graph` and the job's sequences — for inputs no CPU oracle can hold:
 *   every path spells its input sequence base for base (reconstruct_original_sequences, unitig_graph.rs:362-400; decompress.rs:83-105);
 *   every step of every path is a link, links are unique and come in reverse-complement pairs (check_links, unitig_graph.rs:752-793);
 *   depth == number of path occurrences (unitig.rs:149-156); unitigs are in renumber_unitigs order (unitig_graph.rs:295-315);
 *   the statistics are consistent (total_length, link_count().1, kmers.len() == 2 x pre-simplification length).
 * Returns 0 when the checks RAN (report->failed says what they found: 0 = the graph holds), non-zero on a usage error (ac_last_error).
 * `failed` bits: 1 unitig length / range, 2 renumber order, 4 link endpoint out of range, 8 duplicate link, 16 link without mirror,
 * 32 path entry out of range, 64 path step that is no link, 128 path length != sequence length, 256 a path does not spell its sequence,
 * 512 depth != occurrences, 1024 statistics.  first_bad_*: the smallest offending index of each kind (all ones: none).
 *
 * ABI 6 adds the three ORDER-SENSITIVE guarantees of the reference, so that a graph no CPU oracle can hold is checked as "the reference's
 * graph" and not only as "lossless and consistent" (the struct grew at its end: `checks`, `first_bad_junction`):
 *   2048  L-line order: the links are not in get_links_for_gfa order (unitig_graph.rs:333-350: unitigs ascending, forward_next before
 *         reverse_next, inside a list create_links' push order :248-286).  The order inside a class of one list is by SEED number, which a
 *         built graph carries and a graph reloaded from a GFA does not (`checks` bit 2 says whether that part ran);
 *   4096  maximality: a link a -> b that is a's only successor and b's only predecessor, and none of the walk's break cases
 *         (unitig_graph.rs:192-223: the end of a / start of b on a sequence-strand end, b == -a, b == a): a unitig cut in two
 *         (first_bad_link names it);
 *   8192  expand_repeats has not reached its fixed point (graph_simplification.rs:26-40): junction `first_bad_junction` = 2 x unitig index
 *         + side (0 = its exclusive inputs, 1 = its exclusive outputs) passes the reference's candidate test (:190-280) and its clamps
 *         (:145-181) still leave a shift > 0.
 * `checks`: 1 link order, 2 ... with seed numbers, 4 maximality, 8 fixed point — 4 and 8 need a link set that holds (no 4 / 8 / 16), 8 also
 * paths that add up. */
typedef struct {
    uint32_t failed;
    uint64_t first_bad_unitig, first_bad_link, first_bad_path_entry, first_bad_sequence, first_bad_base;
    uint64_t unitigs, links, path_entries, bases_checked, self_mirror_links;
    double seconds;
    uint32_t checks;
    uint64_t first_bad_junction;
} ac_verify_report;
/* seqs: the sequences the graph was built from, as for ac_compress_build (host memory; they are laid out and uploaded as text) */
int ac_verify_graph(const ac_graph* graph, const ac_seq_view* seqs, uint32_t n_seqs, int device, ac_verify_report* report);
/* the same against a text that is resident on the device (the layout ac_compress_build_device takes) */
int ac_verify_graph_device(const ac_graph* graph, const void* d_text, uint64_t n_text, const uint64_t* seq_off, const uint32_t* seq_len,
                           uint32_t n_seqs, int device, ac_verify_report* report);

/* reconstruct_original_sequences (unitig_graph.rs:362-400; decompress.rs:83-105) for ALL sequences of the graph on the device: sequence i lands
 * at out[sum of the lengths before it ...]; out_bytes >= the sum of the lengths (ac_graph_seq_info).  ac_decompress_seq is the per-sequence
 * host form. */
int ac_decompress_device(const ac_graph*, int device, uint8_t* out, uint64_t out_bytes);

/* Test hook: the library's own scan / radix sort / comparator sort kernels (csrc/device_prims.hpp) against the host's std:: algorithms on
 * n pseudo-random items; key_kind 0 uniform, 1 few distinct values, 2 sorted, 3 reverse sorted, 4 one hot digit.  0 = equal. */
int ac_selftest_primitives(int device, uint64_t n, uint64_t seed, int end_bit, int key_kind);

/* The 2-bit packing the host entry applies before the upload (sequence.rs:39-48 validates the same alphabet): n_text bytes ->
 * (n_text + 31) / 32 words of 2-bit codes (A, C, G, T = 0..3, first base most significant) and as many 32-bit mask words
 * (bit i = byte i is not a base).  force_scalar != 0 selects the portable loop instead of the AVX2 / BMI2 one (both are
 * tested against each other and against the device kernel). */
int ac_pack_text(const uint8_t* text, uint64_t n_text, uint64_t* bits, uint32_t* mask32, int force_scalar);

/* Same, for a text that is already resident in device memory (benchmarks, multi-GPU shards):
 * d_text[0] = '$', then for every sequence its padded bytes followed by one '$'; n_text bytes in total.
 * seq_off[s] = index of the first padded byte of sequence s (host array). */
int ac_compress_build_device(uint32_t k, uint32_t assembly_count, const void* d_text, uint64_t n_text,
                             const uint64_t* seq_off, const uint32_t* seq_len, const uint16_t* seq_ids,
                             const uint16_t* seq_d1, const uint16_t* seq_d2, uint32_t n_seqs, int device,
                             ac_graph** out);

/* ---- one compress job over several devices: sequences sharded by rank, the k-mer table partitioned by key hash --
 * One process per device; every rank holds a slice of the job's sequences (rank order = sequence order) as a device text laid out
 * as above.  The library never communicates: the collectives between the phases belong to the caller (torch.distributed over
 * RCCL in autocycler_amd/sharded.py; anything that moves device buffers works).  Every exported buffer holds this rank's
 * CONTRIBUTIONS — the owners partition the keys, so the ranks' contributions are disjoint and a SUM all-reduce completes them.
 *
 *   ac_shard_begin            pack + insert this rank's sequences into a LOCAL table (KmerGraph::add_sequences on the slice), cut the
 *                             runs of rank-novel positions out as "fragments" (+ the first and last k-mer of every sequence, which
 *                             carry first_position, kmer_graph.rs:57-60)
 *   [all-gather]              fragment texts and 8-byte meta records of all ranks, concatenated in rank order;
 *                             union text = '$' + the concatenated fragment texts (every rank holds it)
 *   ac_shard_build_union      this rank inserts the union-text k-mers it OWNS: owner = hash of the k-mer's canonical middle mod
 *                             n_shards — the four successors of a k-mer share the middle, hence the owner — into a table of about
 *                             1/n_shards of the job's k-mers                          -> ac_shard_bitmap_export (novel positions)
 *   [all-reduce SUM int64]    the novel bitmaps (disjoint bits: the sum is the OR)
 *   ac_shard_build_novel      sorted novel list (identical on every rank).  If ac_shard_sib_words() > 0 afterwards (round 5; k >= 3):
 *   [all-reduce SUM uint64]     ac_shard_sib_export: the sibling bits this rank's insert collected, 2 per distinct k-mer by novel index
 *                               (two k-mers of one canonical middle that share their first or last base: the only way a k-mer's
 *                               text neighbour can have a second successor / predecessor; all k-mers of a middle have one owner)
 *   ac_shard_degrees            next_kmers / prev_kmers counts (kmer_graph.rs:136-166): with the summed sibling bits 97-99 % are settled
 *                               without a table access, the same on every rank; the rest, and the first flags, by probing the groups
 *                               this rank owns                                                       -> ac_shard_degrees_export
 *                             (ac_shard_sib_words() == 0: ac_shard_build_novel has run the degree stage itself, every degree by probing)
 *   [all-reduce SUM uint8]    ac_shard_degree_bytes() bytes: one per k-mer left open + four per flagged fragment end (compact), or
 *                             one per distinct k-mer
 *   ac_shard_build_graph      unitigs in seed order (identical on every rank); links (create_links, unitig_graph.rs:234-287),
 *                             probing only owned groups                                                   -> ac_shard_links_export
 *   [all-reduce SUM int32]    10 U link words, U = ac_shard_unitig_count() (the 10 U walk words are derived from them on import)
 *   ac_shard_links_import     the complete links; the keys this rank's path walkers start from      -> ac_shard_queries_export
 *   [all-gather]              the query keys of all ranks (ac_shard_query_count() x ac_shard_query_key_words() u64 per rank)
 *   ac_shard_answer           looks the owned ones among ALL ranks' keys up in this rank's table
 *   [all-reduce SUM int64]    the answers; every rank keeps the slice that answers its own queries
 *   ac_shard_walk             the paths of this rank's sequences (get_unitig_path_for_sequence, unitig_graph.rs:407-465)
 *   [all-reduce SUM, MIN]     ac_shard_reduce_export -> sum buffer (3U int32: depth, path starts, path ends) and min buffer
 *                             (2U int32: smallest forward / reverse position, biased so signed MIN orders them)
 *   ac_shard_reduce_import    the reduced buffers
 *   ac_shard_finish           link order, renumber, expand_repeats, final numbering (identical on every rank).  want: bit 0 =
 *                             unitigs + links to host memory, bit 1 = this rank's paths to host memory.  Either every rank
 *                             keeps the P lines of its own sequences (ac_gfa_string_parts), or:
 *   [gather]                  ac_shard_paths_export (final numbers) -> the writing rank calls ac_graph_set_paths with the
 *                             paths of all sequences in rank order
 * With n_shards == 1 the import pointers may be NULL (nothing to sum).  All `d_` pointers are device pointers into caller-owned
 * buffers of the stated sizes. */
typedef struct ac_shard ac_shard;
int ac_shard_begin(uint32_t k, uint32_t local_assembly_count, const void* d_text, uint64_t n_text, const uint64_t* seq_off,
                   const uint32_t* seq_len, const uint16_t* seq_ids, const uint16_t* seq_d1, const uint16_t* seq_d2,
                   uint32_t n_seqs, int device, ac_shard** out);
int ac_shard_fragment_sizes(const ac_shard*, uint64_t* text_bytes, uint64_t* n_fragments);
int ac_shard_fragments_export(ac_shard*, void* d_text_out /* text_bytes */, void* d_meta_out /* 8 * n_fragments */);
/* The same fragment text as 2-bit codes laid out on the UNION text's word grid (a quarter of the bytes over the links, and the receivers do
 * not pack again): union_off = where this rank's stretch begins in the union text (1 + the text bytes of the ranks before it).  Every rank's
 * words are all-gathered one behind the other; ac_shard_build_union_packed ORs them into place (rank r's n_words[r] words begin at union
 * word first_word[r] = union_off_r / 32) and derives the mask plane from the fragment records. */
uint64_t ac_shard_fragment_packed_words(const ac_shard*, uint64_t union_off);
int ac_shard_fragments_export_packed(ac_shard*, uint64_t union_off, void* d_words_u64, void* d_meta_out /* 8 * n_fragments */);
int ac_shard_build_union_packed(ac_shard*, uint32_t rank, uint32_t n_shards, const void* d_staged_words_u64, const uint64_t* first_word,
                                const uint64_t* n_words, uint64_t n_union_text, const void* d_meta, uint64_t n_fragments_total);
uint64_t ac_shard_local_distinct(const ac_shard*);                 /* distinct canonical k-mers of this rank's slice */
void ac_shard_set_distinct_upper_bound(ac_shard*, uint64_t n);     /* optional, before ac_shard_build_union: the sum of all ranks'
                                                                      local counts sizes the owned tables without a retry */
int ac_shard_build_union(ac_shard*, uint32_t rank, uint32_t n_shards, const void* d_union_text, uint64_t n_union_text,
                         const void* d_meta, uint64_t n_fragments_total);
uint64_t ac_shard_table_capacity(const ac_shard*);     /* slots of this rank's share of the job's k-mer table */
uint64_t ac_shard_bitmap_words(const ac_shard*);       /* u64 words of the union text's novel bitmap */
int ac_shard_bitmap_export(ac_shard*, void* d_out_u64);
int ac_shard_build_novel(ac_shard*, const void* d_bitmap_sum_u64 /* or NULL */);
uint64_t ac_shard_distinct_count(const ac_shard*);     /* N: distinct canonical k-mers of the whole job */
uint64_t ac_shard_sib_words(const ac_shard*);          /* after ac_shard_build_novel: u64 words of the sibling bits to sum (0: none, the degree stage has run) */
int ac_shard_sib_export(ac_shard*, void* d_out_u64 /* ac_shard_sib_words() */);
int ac_shard_degrees(ac_shard*, const void* d_sib_sum_u64);
uint64_t ac_shard_degree_bytes(const ac_shard*);       /* size of the degree exchange (after the degree stage) */
int ac_shard_degrees_export(ac_shard*, void* d_out_u8 /* ac_shard_degree_bytes() */);
int ac_shard_build_graph(ac_shard*, const void* d_degrees_sum_u8 /* ac_shard_degree_bytes(), or NULL */);
uint32_t ac_shard_unitig_count(const ac_shard*);       /* U: sizes the link and reduce buffers */
int ac_shard_links_export(ac_shard*, void* d_links_i32 /* 10 U */, void* d_wlinks_i64 /* 10 U, or NULL: not wanted */);
int ac_shard_links_import(ac_shard*, const void* d_links_sum_i32 /* or NULL: one rank */, const void* d_wlinks_sum_i64 /* or NULL: derived from the link words */);
uint64_t ac_shard_query_count(const ac_shard*);        /* walk queries of this rank */
uint32_t ac_shard_query_key_words(const ac_shard*);    /* u64 words per query key (depends on k only) */
int ac_shard_queries_export(ac_shard*, void* d_out_u64 /* query_count * query_key_words */);
int ac_shard_answer(ac_shard*, const void* d_keys_u64, uint64_t n_queries, void* d_out_u64 /* n_queries */);
int ac_shard_walk(ac_shard*, const void* d_answers_u64 /* query_count: this rank's slice of the summed answers */);
/* The same exchange routed by owner (north_star's bucket exchange; what ac_compress_build_multi does inside the library): the keys
 * ordered by owner rank, counts[r] of them for rank r -> all-to-all -> ac_shard_answer on what arrived -> reverse all-to-all ->
 * ac_shard_walk_routed with the answers in the order ac_shard_queries_route gave the keys.  A rank receives ~1/n_shards of the keys. */
int ac_shard_queries_route(ac_shard*, uint32_t n_shards, void* d_routed_keys_u64 /* query_count * query_key_words */,
                           uint64_t* counts /* n_shards */);
int ac_shard_walk_routed(ac_shard*, const void* d_routed_answers_u64 /* query_count */);
int ac_shard_reduce_export(ac_shard*, void* d_sum_i32 /* 3U */, void* d_min_i32 /* 2U */);
int ac_shard_reduce_import(ac_shard*, const void* d_sum_i32, const void* d_min_i32);
/* Optional, before ac_shard_finish, the same choice on every rank: an in-place all-reduce of a DEVICE buffer over the ranks (dtype 0 =
 * uint8, 1 = int32; op 0 = SUM, 1 = MIN; returns 0 on success), called from inside ac_shard_finish on the calling thread.  With it the
 * order-sensitive tail is no longer replicated in full: expand_repeats (graph_simplification.rs:43-86) runs on this rank's share of the
 * junctions — the conflict components it owns — and the ranks' results are merged by two SUM all-reduces (field lengths, sequence bytes). */
typedef int (*ac_allreduce_fn)(void* user, void* d_buf, uint64_t count, int dtype, int op);
int ac_shard_set_allreduce(ac_shard*, ac_allreduce_fn fn, void* user);
/* Plain copy between two buffers of `device` (or host memory), finished on return: for callers whose collectives want buffers of their
 * own (autocycler_amd/sharded.py stages ac_allreduce_fn's buffer through a torch tensor with it).  Takes no library lock. */
int ac_device_copy(void* dst, const void* src, uint64_t bytes, int device);
int ac_shard_finish(ac_shard*, int want, ac_graph** out);
uint64_t ac_shard_path_entries(const ac_shard*);
/* Only for ranks that did NOT keep their own paths: after ac_shard_finish(want & 2) the rank's entries got their final numbers in host
 * memory (the handle's paths; the device copy stays in seed numbers) and this call fails with an error (ABI 5). */
int ac_shard_paths_export(ac_shard*, void* d_out_i32 /* ac_shard_path_entries() */);
void ac_shard_free(ac_shard*);
/* path_counts[s] = number of path entries of sequence s; d_path_i32 = all entries, concatenated (device). */
int ac_graph_set_paths(ac_graph*, uint32_t n_seqs_total, const uint16_t* seq_ids, const uint32_t* seq_lens,
                       const uint64_t* path_counts, const void* d_path_i32, int device);
uint32_t ac_graph_seq_count(const ac_graph*);
int ac_path_counts(const ac_graph*, uint64_t* counts /* ac_graph_seq_count() */);   /* path entries per sequence */

/* sequence_end_repair (compress.rs:202-270) on the device, for a text that is already resident there: d_text holds the
 * PADDED, UNREPAIRED sequences (Sequence::new_with_seq, sequence.rs:31-59) in the layout above; the chosen matches are
 * patched into it in place and seq_d1 / seq_d2 (dots surviving at each end) are updated, so the same buffer can go straight
 * into ac_compress_build_device / ac_shard_begin.  One pass over the packed text replaces the reference's 2S regex scans. */
int ac_end_repair_device(uint32_t k, void* d_text, uint64_t n_text, const uint64_t* seq_off, const uint32_t* seq_len,
                         uint16_t* seq_d1, uint16_t* seq_d2, uint32_t n_seqs, int device, double* seconds, uint64_t* n_matches);

/* pairwise_contig_distances (cluster.rs:132-157), the first step of `autocycler cluster`, computed on the device from the
 * graph this library has just built: out[a * S + b] = 1 - len(unitigs shared by the paths of a and b) / len(unitigs of a),
 * S = ac_graph_seq_count(), sequences in input order. */
int ac_pairwise_distances(const ac_graph*, int device, double* out);

/* The loader side of save_gfa for the GFAs `compress` writes (UnitigGraph::from_gfa_lines, unitig_graph.rs:55-174): what
 * `autocycler cluster` (cluster.rs:42-43) and `autocycler decompress` (decompress.rs:27-39) start from.  The handle then serves
 * every accessor above (ac_gfa_string on it reproduces the file: tests.rs:108-112), ac_pairwise_distances and: */
int ac_graph_from_gfa(const char* gfa_text, uint64_t len, ac_graph** out);
uint32_t ac_graph_kmer_size(const ac_graph*);
int ac_graph_seq_info(const ac_graph*, uint32_t seq_index, uint16_t* id, uint32_t* length, const char** filename, const char** header);
/* reconstruct_original_sequences (unitig_graph.rs:362-388) for one sequence: out receives its `length` bytes. */
int ac_decompress_seq(const ac_graph*, uint32_t seq_index, uint8_t* out);

/* The whole `autocycler decompress` command (decompress.rs:27-39): in_gfa -> one FASTA per original file name in out_dir
 * (gzip when the name ends in .gz) and / or every contig in out_file (">{filename}__{header}").  Host code. */
int ac_decompress(const char* in_gfa, const char* out_dir, const char* out_file, int threads);

/* Host helper: lay sequences out as the text described above.  text must hold ac_text_size() bytes. */
uint64_t ac_text_size(uint32_t k, const ac_seq_view* seqs, uint32_t n_seqs);
int ac_layout_text(uint32_t k, const ac_seq_view* seqs, uint32_t n_seqs, uint8_t* text, uint64_t* seq_off,
                   uint16_t* seq_d1, uint16_t* seq_d2);

uint64_t ac_kmer_count(const ac_graph*);                 /* KmerGraph.kmers.len(), both strands (compress.rs:152) */
ac_stats ac_stats_pre(const ac_graph*);                  /* print_basic_graph_info after from_kmer_graph (compress.rs:165) */
ac_stats ac_stats_post(const ac_graph*);                 /* ... and after simplify_structure (compress.rs:177) */
uint32_t ac_unitig_count(const ac_graph*);
/* Unitig idx (0-based, final order; Unitig::number == idx + 1): forward_seq, length, depth. */
int ac_unitig(const ac_graph*, uint32_t idx, const uint8_t** seq, uint32_t* len, double* depth);
/* Unitig::forward_positions / reverse_positions as from_gfa_lines rebuilds them (unitig_graph.rs:151-174). */
int ac_unitig_positions(ac_graph*, uint32_t idx, int forward, const ac_position** positions, uint32_t* n);
int ac_links(const ac_graph*, const ac_link** links, uint64_t* n);   /* get_links_for_gfa order (unitig_graph.rs:333-350) */
/* get_unitig_path_for_sequence_i32 (unitig_graph.rs:467-472) of the seq_index-th input sequence. */
int ac_path(const ac_graph*, uint32_t seq_index, const int32_t** signed_unitigs, uint32_t* n);
/* Bulk views of the finished graph, zero-copy and valid until ac_free(): everything save_gfa (unitig_graph.rs:317-331) reads, in
 * five arrays instead of one call per unitig / sequence — what a caller that rebuilds its own UnitigGraph, or checks a graph of
 * 10^8 unitigs, wants.  seq_bytes + seq_begin[i] .. + seq_len[i] = Unitig::forward_seq of unitig i (number i + 1); depth[i] =
 * Unitig::depth; path_entries[path_off[s] .. path_off[s + 1]) = get_unitig_path_for_sequence_i32 of the s-th input sequence.
 * Any out pointer may be NULL. */
int ac_unitigs_bulk(const ac_graph*, const uint8_t** seq_bytes, const uint64_t** seq_begin, const uint32_t** seq_len,
                    const double** depth);
int ac_paths_bulk(const ac_graph*, const int32_t** path_entries, const uint64_t** path_off /* ac_graph_seq_count() + 1 */,
                  uint64_t* n_entries);
int ac_timings_get(const ac_graph*, ac_timings* out);
/* ac_timings only ever grows at its end.  A client that may meet a newer or older library passes sizeof its own ac_timings: at most
 * that many bytes are written, the library's sizeof(ac_timings) is returned. */
size_t ac_timings_get_sized(const ac_graph*, ac_timings* out, size_t out_size);
void ac_free(ac_graph*);

/* The GFA text save_gfa would write (unitig_graph.rs:317-331): H, S*, L*, P* lines.  filenames/headers:
 * Sequence::filename / contig_header per input sequence (FN:Z / HD:Z tags).  Free with ac_string_free. */
int ac_gfa_string(const ac_graph*, const char* const* filenames, const char* const* headers, char** out,
                  uint64_t* out_len);
/* parts: bit 0 = H, S and L lines, bit 1 = P lines (of the sequences this handle holds paths for). */
int ac_gfa_string_parts(const ac_graph*, int parts, const char* const* filenames, const char* const* headers, char** out,
                        uint64_t* out_len);
void ac_string_free(char*);

/* ---- host side around the hot path ("boundary" and "next" rows of SURVEY.md §8) -------------------------------
 * ac_seqs mirrors what load_sequences returns (compress.rs:98-133): padded, end-repaired Sequences + the
 * per-assembly details for the YAML metrics.  The Rust CLI keeps its own loader; these serve the standalone
 * CLI (autocycler-compress), the tests and the benchmarks. */
typedef struct ac_seqs ac_seqs;
int ac_seqs_load(const char* assemblies_dir, uint32_t k, uint32_t max_contigs, int threads, ac_seqs** out);
/* The device the host-side helpers below run sequence_end_repair on (ac_seqs_load, ac_seqs_from_raw with repair: the padded sequences
 * go up as text, the device kernels of ac_end_repair_device patch it, the sequence ends come back — the library holds no host
 * implementation of the repair).  Default 0; one process per GPU sets its own ordinal. */
int ac_set_host_side_device(int device);
/* Sequence::new_with_seq (sequence.rs:31-59) for ids 1..n + optional sequence_end_repair (compress.rs:202-236, on the device). */
int ac_seqs_from_raw(uint32_t k, uint32_t n, const uint8_t* const* seqs, const uint32_t* lens,
                     const char* const* filenames, const char* const* headers, uint32_t assembly_count, int repair,
                     int threads, ac_seqs** out);
uint32_t ac_seqs_count(const ac_seqs*);
uint32_t ac_seqs_assembly_count(const ac_seqs*);
const ac_seq_view* ac_seqs_views(const ac_seqs*);
int ac_seqs_get(const ac_seqs*, uint32_t i, ac_seq_view* view, const char** filename, const char** header);
double ac_seqs_repair_seconds(const ac_seqs*);
int ac_seqs_metrics_yaml(const ac_seqs*, uint32_t unitig_count, uint64_t unitig_total_length, char** out); /* metrics.rs:65-107 */
void ac_seqs_free(ac_seqs*);
int ac_compress_seqs(uint32_t k, const ac_seqs*, int device, ac_graph** out);
/* The whole command (compress.rs:32-50): writes input_assemblies.gfa / .yaml into autocycler_dir.
 * times[4] = load, end repair, graph build (hot path), write (seconds).  graph_out may be NULL. */
int ac_compress_dir(const char* assemblies_dir, const char* autocycler_dir, uint32_t k, uint32_t max_contigs,
                    int threads, int device, ac_graph** graph_out, double* times);

/* The same over several devices (ac_compress_build_multi behind the host loader; end repair on devices[0]). */
int ac_compress_dir_multi(const char* assemblies_dir, const char* autocycler_dir, uint32_t k, uint32_t max_contigs, int threads,
                          const int* devices, int n_devices, ac_graph** graph_out, double* times);

/* Measured ceilings of the device for the two access patterns the graph build is bound by: random atomicCAS and random 8-byte
 * reads on a 134 MB table, in 10^9 operations per second (a ~20 ms microbenchmark; bench.py prices its kernels against them).
 * _at: on a table of table_slots 8-byte slots (rounded up to a power of two between 2^24 and 2^30) — the size of the k-mer table
 * the workload in question builds (ac_timings.table_capacity): a table beyond the 256 MB Infinity Cache takes fewer claims per second. */
int ac_random_access_ceilings(int device, double* cas_gops, double* read_gops);
int ac_random_access_ceilings_at(int device, uint64_t table_slots, double* cas_gops, double* read_gops);
void ac_set_stage_timing(int on);   /* off by default */
int ac_release_memory(void);        /* frees the device arena and the pinned result pool kept between builds */
const char* ac_last_error(void);
int ac_device_count(void);       /* number of visible HIP devices (0 if none / no driver) */
uint32_t ac_max_kmer(void);      /* largest --kmer this build supports */
const char* ac_version(void);
/* The ABI generation of this header (the round it was last changed incompatibly in); ac_abi_version() is the library's.  A caller
 * compiled against another generation must not pass caller-allocated structs that grew (ac_verify_report) or drive the ac_shard_* phases.
 *   5: ac_shard_*: sib_export / all-reduce / ac_shard_degrees between ac_shard_build_novel and the degree exchange whenever
 *      ac_shard_sib_words() > 0; degree buffers are ac_shard_degree_bytes() bytes; ac_shard_paths_export fails after
 *      ac_shard_finish(want & 2) (the rank's own paths were renumbered on the host: read them from the handle).
 *   6: ac_verify_report grew (checks, first_bad_junction; failed bits 2048 / 4096 / 8192).
 *   7: ac_link is two signed unitig numbers (8 bytes; it was { u32 a; u8 a_fwd; u32 b; u8 b_fwd } = 16). */
#define AC_ABI_VERSION 7
int ac_abi_version(void);
const char* ac_source_hash(void);   /* 16 hex digits: digest of the sources this library was built from (csrc/Makefile; tools/source_hash.py) */

#ifdef __cplusplus
}
#endif
#endif
