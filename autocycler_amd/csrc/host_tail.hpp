// Host-side views of the final graph that are only built on demand: Unitig::forward_positions /
// reverse_positions as UnitigGraph::from_gfa_lines rebuilds them from the paths (unitig_graph.rs:151-174).
// (Everything else of the former host tail — link push order, renumbering, expand_repeats — runs on the device.)
#pragma once
#include <cstdint>
#include <vector>

#include "graph_types.hpp"

namespace ac {

void build_positions(FinalGraph* g, const std::vector<uint16_t>& seq_ids, const std::vector<uint32_t>& seq_lens);

}  // namespace ac
