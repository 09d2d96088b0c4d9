// See host_tail.hpp.  Every step names the reference code whose observable order it reproduces.
#include "host_tail.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace ac {

namespace {

inline char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}

// A unitig's forward sequence: a view into RawGraph::seqs until something is appended to it.
struct USeq {
    const char* p = nullptr;
    uint32_t len = 0;
    std::string owned;
    void remove_start(uint32_t n) { p += n; len -= n; }
    void remove_end(uint32_t n) { len -= n; }
    void add_start(const std::string& s) {
        std::string t;
        t.reserve(s.size() + len);
        t.append(s); t.append(p, len);
        owned.swap(t); p = owned.data(); len = (uint32_t)owned.size();
    }
    void add_end(const std::string& s) {
        std::string t;
        t.reserve(s.size() + len);
        t.append(p, len); t.append(s);
        owned.swap(t); p = owned.data(); len = (uint32_t)owned.size();
    }
    // i-th character of the given strand's sequence counted from its start / from its end
    char from_start(bool strand, uint32_t i) const { return strand ? p[i] : comp(p[len - 1 - i]); }
    char from_end(bool strand, uint32_t i) const { return strand ? p[len - 1 - i] : comp(p[i]); }
};

struct Csr {
    std::vector<uint64_t> off;
    std::vector<int32_t> v;
    const int32_t* begin(uint32_t u) const { return v.data() + off[u]; }
    uint32_t count(uint32_t u) const { return (uint32_t)(off[u + 1] - off[u]); }
};

inline uint32_t idx_of(int32_t signed_number) { return (uint32_t)(signed_number < 0 ? -signed_number : signed_number) - 1; }

}  // namespace

void run_host_tail(RawGraph& raw, const std::vector<uint16_t>& seq_ids, const std::vector<uint32_t>& seq_lens,
                   FinalGraph* out) {
    auto t_begin = std::chrono::steady_clock::now();
    const uint32_t U = raw.n_unitigs;
    const size_t S = seq_ids.size();
    out->k = raw.k;
    out->n_kmers = raw.n_kmers;

    // ---- 1. link vectors in create_links' push order (unitig_graph.rs:248-286) ----------
    // forward_next(a): all b+ (seed order asc) then all b- (asc).
    // reverse_next(a): a'- for a' <= a (asc; a' == a is the a+ -> a+ self loop pushed in case 1 of iteration a),
    //                  then b+ (case 3 of iteration a, asc), then a'- for a' > a (asc).
    Csr fnext, rnext;
    fnext.off.assign(U + 1, 0); rnext.off.assign(U + 1, 0);
    auto gather = [&](uint32_t a, int side, int32_t* tmp) {   // successors by symbol -> compact list
        int n = 0;
        const int32_t* src = raw.links.data() + ((size_t)2 * a + side) * 5;
        for (int c = 0; c < 5; c++) if (src[c] != 0) tmp[n++] = src[c];
        return n;
    };
    {
        int32_t tmp[5];
        for (uint32_t a = 0; a < U; a++) {
            fnext.off[a + 1] = fnext.off[a] + gather(a, 0, tmp);
            rnext.off[a + 1] = rnext.off[a] + gather(a, 1, tmp);
        }
    }
    fnext.v.resize(fnext.off[U]); rnext.v.resize(rnext.off[U]);
    uint64_t n_self_rc = 0;
    for (uint32_t a = 0; a < U; a++) {
        int32_t num = (int32_t)a + 1;
        {
            int32_t tmp[5]; int n = gather(a, 0, tmp);
            std::sort(tmp, tmp + n, [](int32_t x, int32_t y) {
                bool xp = x > 0, yp = y > 0;
                if (xp != yp) return xp;                 // positives first
                return std::abs(x) < std::abs(y);
            });
            for (int i = 0; i < n; i++) { fnext.v[fnext.off[a] + i] = tmp[i]; if (tmp[i] == -num) n_self_rc++; }
        }
        {
            int32_t tmp[5]; int n = gather(a, 1, tmp);
            auto cls = [num](int32_t x) { if (x > 0) return 1; return (-x <= num) ? 0 : 2; };
            std::sort(tmp, tmp + n, [&](int32_t x, int32_t y) {
                int cx = cls(x), cy = cls(y);
                if (cx != cy) return cx < cy;
                return std::abs(x) < std::abs(y);
            });
            for (int i = 0; i < n; i++) { rnext.v[rnext.off[a] + i] = tmp[i]; if (tmp[i] == num) n_self_rc++; }
        }
    }
    uint64_t total_links = fnext.v.size() + rnext.v.size();
    // link_count().1 (unitig_graph.rs:478-507): a link and its mirror count once; a link that is its own
    // mirror (a+ -> a-, a- -> a+) counts once.
    uint64_t links_one_way = (total_links + n_self_rc) / 2;

    // ---- 2. sequences, first renumber_unitigs (unitig_graph.rs:295-315): STABLE sort on seed order ---------
    std::vector<USeq> us(U);
    for (uint32_t r = 0; r < U; r++) { us[r].p = raw.seqs.data() + raw.seq_off[r]; us[r].len = raw.len[r]; }
    std::vector<uint32_t> minf(raw.minpos_fwd.begin(), raw.minpos_fwd.end()), minr(raw.minpos_rev.begin(), raw.minpos_rev.end());
    auto unitig_less = [&](uint32_t a, uint32_t b) {
        if (us[a].len != us[b].len) return us[a].len > us[b].len;
        int c = memcmp(us[a].p, us[b].p, us[a].len);
        if (c != 0) return c < 0;
        return raw.depth[a] > raw.depth[b];
    };
    std::vector<uint32_t> order(U);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), unitig_less);

    uint64_t total_len = 0;
    for (uint32_t r = 0; r < U; r++) total_len += us[r].len;
    out->pre = GraphStats{U, links_one_way, total_len};

    // ---- 3. fixed starts/ends (graph_simplification.rs:190-230); invariant across passes ---------------------
    std::vector<uint8_t> fixed_start(U, 0), fixed_end(U, 0);
    {
        std::vector<uint8_t> fs0(U, 0), fe0(U, 0);
        for (size_t s = 0; s < S; s++) {
            uint64_t b = raw.path_off[s], e = raw.path_off[s + 1];
            if (b == e) continue;
            int32_t first = raw.path[b], last = raw.path[e - 1];
            if (first > 0) fs0[idx_of(first)] = 1; else fe0[idx_of(first)] = 1;
            if (last > 0) fe0[idx_of(last)] = 1; else fs0[idx_of(last)] = 1;
        }
        fixed_start = fs0; fixed_end = fe0;
        for (uint32_t u = 0; u < U; u++) {
            if (fs0[u]) {   // upstream of a fixed start: forward_prev(u) = { -e : e in reverse_next(u) }
                const int32_t* p = rnext.begin(u);
                for (uint32_t i = 0; i < rnext.count(u); i++) {
                    int32_t up = -p[i];
                    if (up > 0) fixed_end[idx_of(up)] = 1; else fixed_start[idx_of(up)] = 1;
                }
            }
            if (fe0[u]) {   // downstream of a fixed end
                const int32_t* p = fnext.begin(u);
                for (uint32_t i = 0; i < fnext.count(u); i++) {
                    int32_t down = p[i];
                    if (down > 0) fixed_start[idx_of(down)] = 1; else fixed_end[idx_of(down)] = 1;
                }
            }
        }
    }

    // ---- 4. expand_repeats until nothing moves (graph_simplification.rs:26-142) --------------------------------
    auto next_of = [&](int32_t x, const int32_t** p, uint32_t* n) {   // next links of a unitig strand
        uint32_t u = idx_of(x);
        if (x > 0) { *p = fnext.begin(u); *n = fnext.count(u); } else { *p = rnext.begin(u); *n = rnext.count(u); }
    };
    std::vector<int32_t> srcs;
    auto exclusive = [&](uint32_t x, bool inputs) -> bool {
        // inputs:  forward_prev(x) = { -e : e in reverse_next(x) }, each must lead only to x+      (:233-255)
        // outputs: forward_next(x), each must be reached only from x+                               (:258-280)
        srcs.clear();
        int32_t xnum = (int32_t)x + 1;
        const Csr& lst = inputs ? rnext : fnext;
        const int32_t* p = lst.begin(x);
        uint32_t n = lst.count(x);
        for (uint32_t i = 0; i < n; i++) {
            int32_t other = inputs ? -p[i] : p[i];
            // inputs: other's next list must be exactly [x+].  outputs: other's prev list must be exactly [x+],
            // i.e. the next list of other's opposite strand must be exactly [x-].
            const int32_t* q; uint32_t m;
            next_of(inputs ? other : -other, &q, &m);
            if (!(m == 1 && q[0] == (inputs ? xnum : -xnum))) return false;
            srcs.push_back(other);
        }
        for (int32_t s : srcs) if (idx_of(s) == x) return false;
        return true;
    };
    auto has_dup = [&](const std::vector<int32_t>& v) {
        for (size_t i = 0; i < v.size(); i++)
            for (size_t j = i + 1; j < v.size(); j++)
                if (idx_of(v[i]) == idx_of(v[j])) return true;
        return false;
    };
    int passes = 0;
    for (;;) {
        uint64_t shifted_total = 0;
        for (uint32_t oi = 0; oi < U; oi++) {
            uint32_t x = order[oi];
            // --- inputs side: shift_sequence_1 (:89-119)
            if (rnext.count(x) >= 2 && !fixed_start[x] && exclusive(x, true) && srcs.size() >= 2) {
                bool can = true;
                for (int32_t s : srcs) {
                    uint32_t u = idx_of(s);
                    if ((s > 0 && fixed_end[u]) || (s < 0 && fixed_start[u])) { can = false; break; }
                }
                if (can) {
                    // get_common_end_seq (:298-312): longest common suffix of the source strand sequences
                    uint32_t min_len = UINT32_MAX;
                    for (int32_t s : srcs) min_len = std::min(min_len, us[idx_of(s)].len);
                    uint32_t amount = 0;
                    const USeq& s0 = us[idx_of(srcs[0])];
                    while (amount < min_len) {
                        char c = s0.from_end(srcs[0] > 0, amount);
                        bool same = true;
                        for (size_t i = 1; i < srcs.size(); i++)
                            if (us[idx_of(srcs[i])].from_end(srcs[i] > 0, amount) != c) { same = false; break; }
                        if (!same) break;
                        amount++;
                    }
                    if (amount > 0) {
                        // avoid_zero_len_unitigs (:145-161): trim while min_source_len <= len * dup
                        uint32_t dup = has_dup(srcs) ? 2 : 1;
                        amount = std::min(amount, (min_len - 1) / dup);
                    }
                    if (amount > 0) {
                        // avoid_start_of_path (:164-181): trim while any forward position <= len
                        uint32_t m = minf[x];
                        if (m == 0) throw std::logic_error("avoid_start_of_path on a path start");
                        amount = std::min(amount, m - 1);
                    }
                    if (amount > 0) {
                        std::string common(amount, 'N');   // the LAST `amount` characters of the common suffix
                        for (uint32_t i = 0; i < amount; i++) common[amount - 1 - i] = s0.from_end(srcs[0] > 0, i);
                        for (int32_t s : srcs) {
                            uint32_t u = idx_of(s);
                            if (s > 0) { us[u].remove_end(amount); minr[u] += amount; }     // unitig.rs:226-233
                            else { us[u].remove_start(amount); minf[u] += amount; }          // unitig.rs:217-224
                        }
                        us[x].add_start(common); minf[x] -= amount;                          // unitig.rs:235-241
                        shifted_total += amount;
                    }
                }
            }
            // --- outputs side: shift_sequence_2 (:122-142)
            if (fnext.count(x) >= 2 && !fixed_end[x] && exclusive(x, false) && srcs.size() >= 2) {
                bool can = true;
                for (int32_t s : srcs) {
                    uint32_t u = idx_of(s);
                    if ((s > 0 && fixed_start[u]) || (s < 0 && fixed_end[u])) { can = false; break; }
                }
                if (can) {
                    uint32_t min_len = UINT32_MAX;
                    for (int32_t s : srcs) min_len = std::min(min_len, us[idx_of(s)].len);
                    uint32_t amount = 0;
                    const USeq& s0 = us[idx_of(srcs[0])];
                    while (amount < min_len) {   // get_common_start_seq (:283-295)
                        char c = s0.from_start(srcs[0] > 0, amount);
                        bool same = true;
                        for (size_t i = 1; i < srcs.size(); i++)
                            if (us[idx_of(srcs[i])].from_start(srcs[i] > 0, amount) != c) { same = false; break; }
                        if (!same) break;
                        amount++;
                    }
                    if (amount > 0) {
                        uint32_t dup = has_dup(srcs) ? 2 : 1;
                        amount = std::min(amount, (min_len - 1) / dup);
                    }
                    if (amount > 0) {
                        uint32_t m = minr[x];
                        if (m == 0) throw std::logic_error("avoid_start_of_path on a path start");
                        amount = std::min(amount, m - 1);
                    }
                    if (amount > 0) {
                        std::string common(amount, 'N');   // the FIRST `amount` characters of the common prefix
                        for (uint32_t i = 0; i < amount; i++) common[i] = s0.from_start(srcs[0] > 0, i);
                        for (int32_t s : srcs) {
                            uint32_t u = idx_of(s);
                            if (s > 0) { us[u].remove_start(amount); minf[u] += amount; }
                            else { us[u].remove_end(amount); minr[u] += amount; }
                        }
                        us[x].add_end(common); minr[x] -= amount;                            // unitig.rs:243-249
                        shifted_total += amount;
                    }
                }
            }
        }
        passes++;
        if (shifted_total == 0) break;
    }
    out->simplify_passes = passes;

    // ---- 5. second renumber_unitigs (graph_simplification.rs:39): stable sort of the CURRENT order -----------
    std::stable_sort(order.begin(), order.end(), unitig_less);
    std::vector<uint32_t> final_number(U);
    for (uint32_t i = 0; i < U; i++) final_number[order[i]] = i + 1;

    total_len = 0;
    for (uint32_t r = 0; r < U; r++) total_len += us[r].len;
    out->post = GraphStats{U, links_one_way, total_len};

    // ---- 6. outputs in final order ---------------------------------------------------------------------------
    out->seqs.resize(U); out->depth.resize(U);
    for (uint32_t i = 0; i < U; i++) {
        uint32_t r = order[i];
        out->seqs[i].assign(us[r].p, us[r].len);
        out->depth[i] = (double)raw.depth[r];
    }
    out->links.clear();
    out->links.reserve(total_links);
    for (uint32_t i = 0; i < U; i++) {   // get_links_for_gfa (unitig_graph.rs:333-350)
        uint32_t r = order[i];
        const int32_t* p = fnext.begin(r);
        for (uint32_t j = 0; j < fnext.count(r); j++)
            out->links.push_back(Link{i + 1, 1, final_number[idx_of(p[j])], (uint8_t)(p[j] > 0)});
        p = rnext.begin(r);
        for (uint32_t j = 0; j < rnext.count(r); j++)
            out->links.push_back(Link{i + 1, 0, final_number[idx_of(p[j])], (uint8_t)(p[j] > 0)});
    }
    out->path_off.assign(raw.path_off.begin(), raw.path_off.end());
    out->path.resize(raw.path.size());
    for (size_t i = 0; i < raw.path.size(); i++) {
        int32_t v = raw.path[i];
        int32_t f = (int32_t)final_number[idx_of(v)];
        out->path[i] = v > 0 ? f : -f;
    }
    // The path of every sequence must spell its full length (unitig_graph.rs:160-174, decompress.rs).
    for (size_t s = 0; s < S; s++) {
        uint64_t sum = 0;
        for (uint64_t i = out->path_off[s]; i < out->path_off[s + 1]; i++) sum += out->seqs[idx_of(out->path[i])].size();
        if (sum != seq_lens[s]) throw std::logic_error("internal error: path length mismatch for sequence " + std::to_string(seq_ids[s]));
    }
    out->tail_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
}

void build_positions(FinalGraph* g, const std::vector<uint16_t>& seq_ids, const std::vector<uint32_t>& seq_lens) {
    size_t U = g->seqs.size();
    g->fwd_positions.assign(U, {});
    g->rev_positions.assign(U, {});
    for (size_t s = 0; s < seq_ids.size(); s++) {
        uint64_t b = g->path_off[s], e = g->path_off[s + 1];
        uint32_t pos = 0;
        for (uint64_t i = b; i < e; i++) {          // forward path, strand bit set (position.rs:24-33)
            int32_t v = g->path[i];
            uint32_t u = idx_of(v);
            (v > 0 ? g->fwd_positions : g->rev_positions)[u].push_back(Position{pos, (uint16_t)(seq_ids[s] | 0x8000)});
            pos += (uint32_t)g->seqs[u].size();
        }
        pos = 0;
        for (uint64_t i = e; i-- > b;) {            // reverse path (unitig_graph.rs:982-984)
            int32_t v = -g->path[i];
            uint32_t u = idx_of(v);
            (v > 0 ? g->fwd_positions : g->rev_positions)[u].push_back(Position{pos, seq_ids[s]});
            pos += (uint32_t)g->seqs[u].size();
        }
        (void)seq_lens;
    }
}

}  // namespace ac
