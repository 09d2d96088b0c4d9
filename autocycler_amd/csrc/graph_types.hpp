// Plain host-side data types shared by the device pipeline (graph_build), the sequential host tail and the C ABI.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ac {

// A block of host memory handed to the caller together with its deleter (pinned memory from a recycling pool
// in the HIP build, so device -> host copies into it run at full PCIe rate and cost no allocation in steady state).
struct HostBlock {
    void* p = nullptr;
    size_t bytes = 0;
    void (*release)(void*, size_t) = nullptr;
    HostBlock() {}
    HostBlock(const HostBlock&) = delete;
    HostBlock& operator=(const HostBlock&) = delete;
    HostBlock(HostBlock&& o) noexcept : p(o.p), bytes(o.bytes), release(o.release) { o.p = nullptr; o.release = nullptr; }
    HostBlock& operator=(HostBlock&& o) noexcept {
        if (this != &o) { reset(); p = o.p; bytes = o.bytes; release = o.release; o.p = nullptr; o.release = nullptr; }
        return *this;
    }
    void reset() { if (p && release) release(p, bytes); p = nullptr; release = nullptr; bytes = 0; }
    ~HostBlock() { reset(); }
};

struct GraphStats { uint32_t unitigs = 0; uint64_t links_one_way = 0; uint64_t total_length = 0; };

// A link a -> b as two SIGNED unitig numbers: +n = the forward strand of unitig n, -n = its reverse strand (the form the reference's own
// path entries have, get_unitig_path_for_sequence_i32).  8 bytes (round 6; rounds 1-5 carried {u32, u8, u32, u8} = 16 with padding: the link
// array is the largest late result of a mixed-species build — 307 of 648 MB on mini-E — and crosses PCIe behind everything else).
// 0 is no unitig (a GFA that names one: the verifier's range check).
struct Link {
    int32_t a, b;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t na() const { return a < 0 ? (uint32_t)(-(int64_t)a) : (uint32_t)a; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t nb() const { return b < 0 ? (uint32_t)(-(int64_t)b) : (uint32_t)b; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    bool a_fwd() const { return a > 0; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    bool b_fwd() const { return b > 0; }
};

struct Position { uint32_t pos; uint16_t seq_id_and_strand; };   // position.rs:18-22

// The final UnitigGraph (after simplify_structure), in final order (number = index + 1).  The bulk arrays live
// in host blocks filled straight from the device.
struct FinalGraph {
    uint32_t k = 0;
    uint64_t n_kmers = 0;
    GraphStats pre, post;
    uint32_t n_unitigs = 0;
    HostBlock seq_block;                     // final forward sequences, concatenated (in seed order)
    HostBlock meta_block;                    // seq_begin[U] (u64) | depth[U] (double) | seq_len[U] (u32) [| seed_index[U] (u32): built graphs]
    const uint64_t* seq_begin = nullptr;     // offset of unitig i's sequence in seq_block
    const double* depth = nullptr;
    const uint32_t* seq_len = nullptr;
    // built graphs only (null for a graph loaded from a GFA): the index final unitig i had in SEED order, i.e. its number - 1 when
    // create_links ran (unitig_graph.rs:234-287 pushes links in that order).  Only the verifier reads it (L-line order inside a group).
    const uint32_t* seed_index = nullptr;
    HostBlock links_block;                   // Link[n_links] in get_links_for_gfa order (unitig_graph.rs:333-350)
    const Link* links = nullptr;
    uint64_t n_links = 0;
    std::vector<uint64_t> path_off;          // n_seqs + 1
    HostBlock path_block;                    // signed final numbers
    const int32_t* path = nullptr;
    uint64_t n_path = 0;
    // lazily built by build_positions(): forward/reverse positions per unitig as from_gfa_lines would
    // rebuild them (unitig_graph.rs:151-174)
    std::vector<std::vector<Position>> fwd_positions, rev_positions;
    int simplify_passes = 0;

    const char* seq(uint32_t i) const { return (const char*)seq_block.p + seq_begin[i]; }
};

}  // namespace ac
