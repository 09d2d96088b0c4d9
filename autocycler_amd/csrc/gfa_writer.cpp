#include "gfa_writer.hpp"

#include <cstdio>
#include <cstring>

namespace ac {

static inline void put_u64(std::string& s, uint64_t v) {
    char buf[24]; int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(buf[--n]);
}

std::string gfa_string(const FinalGraph& g, const std::vector<SeqMeta>& seqs, int parts) {
    std::string out;
    size_t est = 64;
    if (parts & 1) est += g.post.total_length + (size_t)g.n_unitigs * 32 + g.n_links * 32;
    if (parts & 2) est += g.n_path * 10 + seqs.size() * 256;
    out.reserve(est);
    if (parts & 1) {
    out += "H\tVN:Z:1.0\tKM:i:"; put_u64(out, g.k); out.push_back('\n');
    for (uint32_t i = 0; i < g.n_unitigs; i++) {
        out += "S\t"; put_u64(out, (uint64_t)i + 1); out.push_back('\t'); out.append(g.seq(i), g.seq_len[i]); out += "\tDP:f:";
        char buf[64]; snprintf(buf, sizeof buf, "%.2f", g.depth[i]);   // Rust {:.2}
        out += buf; out.push_back('\n');
    }
    for (uint64_t li = 0; li < g.n_links; li++) {
        const Link& l = g.links[li];
        out += "L\t"; put_u64(out, l.a); out += l.a_fwd ? "\t+\t" : "\t-\t"; put_u64(out, l.b);
        out += l.b_fwd ? "\t+\t0M\n" : "\t-\t0M\n";
    }
    }
    if (parts & 2)
    for (size_t s = 0; s < seqs.size(); s++) {
        out += "P\t"; put_u64(out, seqs[s].id); out.push_back('\t');
        for (uint64_t i = g.path_off[s]; i < g.path_off[s + 1]; i++) {
            if (i != g.path_off[s]) out.push_back(',');
            int32_t v = g.path[i];
            put_u64(out, (uint64_t)(v < 0 ? -v : v)); out.push_back(v < 0 ? '-' : '+');
        }
        out += "\t*\tLN:i:"; put_u64(out, seqs[s].length);
        out += "\tFN:Z:"; out += seqs[s].filename; out += "\tHD:Z:"; out += seqs[s].contig_header;
        out.push_back('\n');   // cluster is 0 in compress output: no CL:i tag (unitig_graph.rs:357)
    }
    return out;
}

}  // namespace ac
