// See host_tail.hpp.
#include "host_tail.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ac {

namespace {
inline uint32_t idx_of(int32_t signed_number) { return (uint32_t)(signed_number < 0 ? -signed_number : signed_number) - 1; }
}  // namespace

void build_positions(FinalGraph* g, const std::vector<uint16_t>& seq_ids, const std::vector<uint32_t>& seq_lens) {
    size_t U = g->n_unitigs;
    g->fwd_positions.assign(U, {});
    g->rev_positions.assign(U, {});
    for (size_t s = 0; s < seq_ids.size(); s++) {
        uint64_t b = g->path_off[s], e = g->path_off[s + 1];
        uint32_t pos = 0;
        for (uint64_t i = b; i < e; i++) {          // forward path, strand bit set (position.rs:24-33)
            int32_t v = g->path[i];
            uint32_t u = idx_of(v);
            (v > 0 ? g->fwd_positions : g->rev_positions)[u].push_back(Position{pos, (uint16_t)(seq_ids[s] | 0x8000)});
            pos += g->seq_len[u];
        }
        pos = 0;
        for (uint64_t i = e; i-- > b;) {            // reverse path (unitig_graph.rs:982-984)
            int32_t v = -g->path[i];
            uint32_t u = idx_of(v);
            (v > 0 ? g->fwd_positions : g->rev_positions)[u].push_back(Position{pos, seq_ids[s]});
            pos += g->seq_len[u];
        }
        (void)seq_lens;
    }
}

}  // namespace ac
