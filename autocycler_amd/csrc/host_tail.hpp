// Order-dependent tail of the compress hot path, on the compacted graph (10^4-10^7 unitigs):
// link push order of create_links (unitig_graph.rs:234-287), renumber_unitigs (:295-315),
// simplify_structure / expand_repeats (graph_simplification.rs:26-312), link_count (:478-507).
// These are sequential and order-sensitive in the reference; they run on the host
// over flat arrays (no Rc<RefCell<..>>, no per-position vectors: only the minimum position per strand is
// ever consulted, graph_simplification.rs:164-181).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "graph_build.hpp"

namespace ac {

struct GraphStats { uint32_t unitigs = 0; uint64_t links_one_way = 0; uint64_t total_length = 0; };

struct Link { uint32_t a; uint8_t a_fwd; uint32_t b; uint8_t b_fwd; };

struct Position { uint32_t pos; uint16_t seq_id_and_strand; };   // position.rs:18-22

struct FinalGraph {
    uint32_t k = 0;
    uint64_t n_kmers = 0;
    GraphStats pre, post;
    // final order (number = index + 1)
    std::vector<std::string> seqs;
    std::vector<double> depth;
    std::vector<Link> links;                 // get_links_for_gfa order (unitig_graph.rs:333-350)
    std::vector<uint64_t> path_off;          // n_seqs + 1
    std::vector<int32_t> path;               // signed final numbers
    // lazily built by build_positions(): forward/reverse positions per unitig as from_gfa_lines would
    // rebuild them (unitig_graph.rs:151-174)
    std::vector<std::vector<Position>> fwd_positions, rev_positions;
    double tail_seconds = 0;
    int simplify_passes = 0;
};

// seq_ids / seq_lens: per input sequence (same order as the SeqViews given to the builder).
void run_host_tail(RawGraph& raw, const std::vector<uint16_t>& seq_ids, const std::vector<uint32_t>& seq_lens,
                   FinalGraph* out);

void build_positions(FinalGraph* g, const std::vector<uint16_t>& seq_ids, const std::vector<uint32_t>& seq_lens);

}  // namespace ac
