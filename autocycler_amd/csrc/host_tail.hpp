// The order-dependent, sequential part of the compress hot path on the compacted graph (10^4-10^7 unitigs):
// simplify_structure / expand_repeats (graph_simplification.rs:26-312) with the shift primitives of
// unitig.rs:217-249.  Everything around it that is order-free runs on the device (graph_build.hip): link
// push order of create_links (unitig_graph.rs:234-287), both renumber_unitigs sorts (:295-315), the fixed
// starts/ends and exclusivity tests of graph_simplification.rs:190-280, path renumbering and link_count.
// No Rc<RefCell<..>>, no per-position vectors: only the minimum position per strand is ever consulted
// (graph_simplification.rs:164-181).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "graph_types.hpp"

namespace ac {

void run_expand_repeats(const RawGraph& raw, char* seq_out, TailResult* out);

void build_positions(FinalGraph* g, const std::vector<uint16_t>& seq_ids, const std::vector<uint32_t>& seq_lens);

}  // namespace ac
