// One compress job over several devices of one node, driven from ONE process behind the C ABI (ac_compress_build_multi): a host
// thread per device, each with its own device context (device_rt.hpp: arena, streams, mailbox), the sequences sharded by rank, the
// k-mer table partitioned by key hash, and the exchanges between the phases done inside the library — RCCL over xGMI (ncclAllReduce,
// grouped ncclSend / ncclRecv routed by owner), or staged through host memory where RCCL cannot run (several ranks on one device:
// tests and dry runs; the CPU emulation).  SURVEY.md §8(e); DESIGN.md §7.
#pragma once
#include <cstdint>
#include <vector>

#include "graph_build.hpp"

namespace ac {

enum MultiTransport { MULTI_AUTO = 0, MULTI_HOST_STAGED = 1, MULTI_RCCL = 2, MULTI_DIRECT = 3 };      // DIRECT: one rank, built as a single-device job (no protocol)

struct MultiStats {
    uint32_t n_ranks = 0;
    int transport = 0;                      // MULTI_HOST_STAGED, MULTI_RCCL or MULTI_DIRECT: what ran
    // bytes this build moved between ranks, summed over all ranks (what every rank RECEIVED from others)
    uint64_t bytes_fragments = 0, bytes_bitmap = 0, bytes_degrees = 0, bytes_links = 0, bytes_queries = 0, bytes_answers = 0, bytes_reduce = 0;
    uint64_t queries_total = 0, queries_sent_away = 0;      // walk-start queries of all ranks / those answered by another rank
    uint64_t table_capacity_max = 0, table_capacity_sum = 0;   // the ranks' shares of the job's k-mer table
    uint64_t union_text_bytes = 0, fragments = 0, distinct = 0;
    double seconds_total = 0, seconds_exchange_max = 0;      // wall clock of the call / the slowest rank's time inside exchanges
    uint64_t candidates_total = 0, candidates_owned_max = 0; // expand_repeats: candidate junctions of the job / the most one rank ran (its conflict components)
    uint64_t bytes_tail = 0;                                 // the tail's merge: field lengths + sequence bytes (all-reduces)
    uint64_t bytes_sibling = 0, degrees_open = 0;            // round 5: the sibling bits' exchange; k-mers the light degree step left to the probes
    uint64_t bytes_received_max = 0;                         // the most any one rank received over the whole build
    uint64_t path_runs_copied = 0;                           // pieces of followed runs the ranks' copying walks copied instead of walking (0: every rank walked all of its text)
};

// seqs: all sequences of the job in input order; devices[r] = HIP ordinal of rank r (an ordinal may appear more than once: those ranks
// share the device — host-staged exchanges only).  The graph lands in `out` as from a single-device build.
void build_multi(uint32_t k, uint32_t assembly_count, const std::vector<SeqView>& seqs, const std::vector<int>& devices, int transport,
                 FinalGraph* out, BuildTimings* tm, MultiStats* st);
// Frees the per-rank device contexts (arenas, rings) kept between multi-device builds, and the RCCL communicators.
void release_multi_contexts();
// (capi.cpp) hipSetDevice + gfx950 check + arena / device bookkeeping of the calling thread's context.
void select_device_checked(int device);

}  // namespace ac
