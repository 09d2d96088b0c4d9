// GFA text emission of the final graph — save_gfa (unitig_graph.rs:317-360), gfa_segment_line
// (unitig.rs:168-172).  One buffered pass instead of one writeln! syscall per line.
#pragma once
#include <string>
#include <vector>

#include "host_tail.hpp"

namespace ac {
struct SeqMeta { uint16_t id; uint32_t length; std::string filename; std::string contig_header; };
// parts: bit 0 = H, S and L lines, bit 1 = P lines (a sharded build can keep the P lines of each rank's sequences on that rank)
std::string gfa_string(const FinalGraph& g, const std::vector<SeqMeta>& seqs, int parts = 3);
// gfa_reader.cpp: the loader side (UnitigGraph::from_gfa_lines, unitig_graph.rs:55-174) and decompress's reconstruction
void load_gfa(const char* text, size_t len, FinalGraph* g, std::vector<SeqMeta>* seqs);
void decompress_sequence(const FinalGraph& g, size_t seq_index, char* out);      // out: LN bytes
std::vector<std::string> gfa_chunks(const FinalGraph& g, const std::vector<SeqMeta>& seqs, int threads);
}  // namespace ac
