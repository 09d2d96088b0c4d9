#include "gfa_writer.hpp"

#include <cstdio>
#include <cstring>
#include <algorithm>
#include <thread>

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
        out.reserve(out.size() + (size_t)(g.path_off[s + 1] - g.path_off[s]) * 8 + 512);
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

// The same text in pieces built by several threads (H/S/L in one piece, the P lines split by path entries): concatenated in
// order they are exactly gfa_string().  Used by the whole-command driver, which writes the pieces one after the other.
std::vector<std::string> gfa_chunks(const FinalGraph& g, const std::vector<SeqMeta>& seqs, int threads) {
    size_t S = seqs.size();
    int T = std::max(1, std::min<int>(threads, 64));
    std::vector<size_t> cut{0};          // sequence ranges with about equal numbers of path entries
    uint64_t total = g.path_off.empty() ? 0 : g.path_off[S], per = total / (uint64_t)T + 1;
    for (size_t s = 0; s < S; s++)
        if (g.path_off[s + 1] >= per * cut.size() && s + 1 < S) cut.push_back(s + 1);
    cut.push_back(S);
    std::vector<std::string> out(cut.size());          // out[0] = H, S, L; out[i] = P lines of sequences [cut[i-1], cut[i])
    auto p_lines = [&](size_t a, size_t b, std::string* dst) {
        FinalGraph view;       // a shallow view restricted to [a, b): same arrays, shifted offsets
        std::vector<SeqMeta> sub(seqs.begin() + (long)a, seqs.begin() + (long)b);
        view.k = g.k; view.path = g.path; view.path_off.assign(g.path_off.begin() + (long)a, g.path_off.begin() + (long)b + 1);
        view.n_path = view.path_off.back() - view.path_off.front();
        *dst = gfa_string(view, sub, 2);
    };
    std::vector<std::thread> pool;
    for (size_t i = 1; i < cut.size(); i++) pool.emplace_back(p_lines, cut[i - 1], cut[i], &out[i]);
    out[0] = gfa_string(g, seqs, 1);
    for (auto& t : pool) t.join();
    return out;
}

}  // namespace ac
