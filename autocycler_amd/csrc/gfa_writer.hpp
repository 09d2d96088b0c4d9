// GFA text emission of the final graph — save_gfa (unitig_graph.rs:317-360), gfa_segment_line
// (unitig.rs:168-172).  One buffered pass instead of one writeln! syscall per line.
#pragma once
#include <string>
#include <vector>

#include "host_tail.hpp"

namespace ac {
struct SeqMeta { uint16_t id; uint32_t length; std::string filename; std::string contig_header; };
std::string gfa_string(const FinalGraph& g, const std::vector<SeqMeta>& seqs);
}  // namespace ac
