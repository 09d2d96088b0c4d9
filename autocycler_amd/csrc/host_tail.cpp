// See host_tail.hpp.  Every step names the reference code whose observable behaviour it reproduces.
#include "host_tail.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ac {

namespace {

inline char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}

inline uint32_t idx_of(int32_t signed_number) { return (uint32_t)(signed_number < 0 ? -signed_number : signed_number) - 1; }

// A unitig's forward sequence as [bytes shifted onto its start][what is left of the device-built core][bytes
// shifted onto its end].  The core is a view into RawGraph::seqs and is never copied while shifting; the
// shifted-on bytes live in one shared pool (offsets, so the pool may grow).
struct USeq {
    const char* core = nullptr;
    uint32_t core_len = 0;
    uint32_t pre_off = 0, pre_len = 0, post_off = 0, post_len = 0;
    uint32_t len() const { return pre_len + core_len + post_len; }
};

struct Seqs {
    std::vector<USeq> us;
    std::vector<char> pool;
    char at(const USeq& u, uint32_t i) const {
        if (i < u.pre_len) return pool[u.pre_off + i];
        i -= u.pre_len;
        if (i < u.core_len) return u.core[i];
        return pool[u.post_off + (i - u.core_len)];
    }
    // i-th character of the given strand's sequence counted from its start / from its end
    char from_start(const USeq& u, bool strand, uint32_t i) const { return strand ? at(u, i) : comp(at(u, u.len() - 1 - i)); }
    char from_end(const USeq& u, bool strand, uint32_t i) const { return strand ? at(u, u.len() - 1 - i) : comp(at(u, i)); }
    void remove_start(USeq& u, uint32_t n) {
        uint32_t d = std::min(n, u.pre_len); u.pre_off += d; u.pre_len -= d; n -= d;
        d = std::min(n, u.core_len); u.core += d; u.core_len -= d; n -= d;
        u.post_off += n; u.post_len -= n;
    }
    void remove_end(USeq& u, uint32_t n) {
        uint32_t d = std::min(n, u.post_len); u.post_len -= d; n -= d;
        d = std::min(n, u.core_len); u.core_len -= d; n -= d;
        u.pre_len -= n;
    }
    void add_start(USeq& u, const char* s, uint32_t n) {   // new prefix = s + old prefix
        uint32_t off = (uint32_t)pool.size();
        pool.insert(pool.end(), s, s + n);
        for (uint32_t i = 0; i < u.pre_len; i++) pool.push_back(pool[u.pre_off + i]);
        u.pre_off = off; u.pre_len += n;
    }
    void add_end(USeq& u, const char* s, uint32_t n) {     // new suffix = old suffix + s
        uint32_t off = (uint32_t)pool.size();
        for (uint32_t i = 0; i < u.post_len; i++) pool.push_back(pool[u.post_off + i]);
        pool.insert(pool.end(), s, s + n);
        u.post_off = off; u.post_len += n;
    }
};

}  // namespace

void run_expand_repeats(const RawGraph& raw, char* seq_out, TailResult* out) {
    auto t_begin = std::chrono::steady_clock::now();
    auto t_last = t_begin;
    const bool prof = getenv("AC_TAIL_PROFILE") != nullptr;
    auto lap = [&](const char* what, uint64_t n) {
        if (!prof) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[tail] %-24s %8.3f ms  (%llu)\n", what, std::chrono::duration<double>(t - t_last).count() * 1e3, (unsigned long long)n);
        t_last = t;
    };
    const uint32_t U = raw.n_unitigs;
    Seqs S;
    S.us.resize(U);
    for (uint32_t r = 0; r < U; r++) { S.us[r].core = raw.seqs.data() + raw.seq_off[r]; S.us[r].core_len = raw.len[r]; }
    std::vector<uint32_t> minf(raw.minpos_fwd.begin(), raw.minpos_fwd.end()), minr(raw.minpos_rev.begin(), raw.minpos_rev.end());

    auto fnext = [&](uint32_t u, uint32_t* n) { *n = raw.link_cnt[2 * (size_t)u]; return raw.links.data() + (2 * (size_t)u) * 5; };
    auto rnext = [&](uint32_t u, uint32_t* n) { *n = raw.link_cnt[2 * (size_t)u + 1]; return raw.links.data() + (2 * (size_t)u + 1) * 5; };

    // Candidate junctions in the reference's visiting order (graph_simplification.rs:57-84: unitigs in the order
    // of the first renumbering, inputs side then outputs side).  raw.cand already holds every test that does not
    // depend on sequence content; what is left per visit is the amount to shift.
    std::vector<uint32_t> clist;
    for (uint32_t oi = 0; oi < U; oi++) {
        uint32_t x = raw.order1[oi];
        if (raw.cand[2 * (size_t)x]) clist.push_back(2 * x);
        if (raw.cand[2 * (size_t)x + 1]) clist.push_back(2 * x + 1);
    }
    // A junction whose unitigs have not changed since it was last found to shift nothing shifts nothing again, so
    // only junctions touching a changed unitig are revisited (same visits with a non-zero result, same order,
    // same number of passes as the reference, which re-tests every junction in every pass).
    std::vector<uint8_t> dirty(raw.cand.begin(), raw.cand.end());
    auto mark_dirty = [&](uint32_t u) {
        dirty[2 * (size_t)u] = raw.cand[2 * (size_t)u];
        dirty[2 * (size_t)u + 1] = raw.cand[2 * (size_t)u + 1];
        for (int side = 0; side < 2; side++) {   // u's strand is a source of the junction its only link leads to
            uint32_t n; const int32_t* p = side == 0 ? fnext(u, &n) : rnext(u, &n);
            if (n != 1) continue;
            size_t c = 2 * (size_t)idx_of(p[0]) + (p[0] > 0 ? 0 : 1);
            dirty[c] = raw.cand[c];
        }
    };

    lap("setup + candidate list", clist.size());
    int32_t srcs[5];
    std::string common;
    int passes = 0;
    for (;;) {
        uint64_t shifted_total = 0;
        for (uint32_t c : clist) {
            if (!dirty[c]) continue;
            dirty[c] = 0;
            const uint32_t x = c >> 1;
            const bool inputs = (c & 1) == 0;
            uint32_t n;
            const int32_t* p = inputs ? rnext(x, &n) : fnext(x, &n);
            // inputs:  forward_prev(x) = { -e : e in reverse_next(x) }   (graph_simplification.rs:233-255)
            // outputs: forward_next(x)                                    (:258-280)
            for (uint32_t i = 0; i < n; i++) srcs[i] = inputs ? -p[i] : p[i];
            uint32_t min_len = UINT32_MAX;
            for (uint32_t i = 0; i < n; i++) min_len = std::min(min_len, S.us[idx_of(srcs[i])].len());
            // get_common_end_seq (:298-312) / get_common_start_seq (:283-295) of the source strand sequences
            uint32_t amount = 0;
            const USeq& s0 = S.us[idx_of(srcs[0])];
            while (amount < min_len) {
                char ch = inputs ? S.from_end(s0, srcs[0] > 0, amount) : S.from_start(s0, srcs[0] > 0, amount);
                bool same = true;
                for (uint32_t i = 1; i < n; i++) {
                    const USeq& si = S.us[idx_of(srcs[i])];
                    char ci = inputs ? S.from_end(si, srcs[i] > 0, amount) : S.from_start(si, srcs[i] > 0, amount);
                    if (ci != ch) { same = false; break; }
                }
                if (!same) break;
                amount++;
            }
            if (amount > 0) {
                // avoid_zero_len_unitigs (:145-161): trim while min_source_len <= len * dup
                bool dup = false;
                for (uint32_t i = 0; i < n; i++)
                    for (uint32_t j = i + 1; j < n; j++)
                        if (idx_of(srcs[i]) == idx_of(srcs[j])) dup = true;
                amount = std::min(amount, (min_len - 1) / (dup ? 2u : 1u));
            }
            if (amount > 0) {
                // avoid_start_of_path (:164-181): trim while any forward (inputs) / reverse (outputs) position <= len
                uint32_t m = inputs ? minf[x] : minr[x];
                if (m == 0) throw std::logic_error("avoid_start_of_path on a path start");
                amount = std::min(amount, m - 1);
            }
            if (amount == 0) continue;
            common.assign(amount, 'N');
            if (inputs) {   // shift_sequence_1 (:89-119): the LAST `amount` characters of the common suffix move onto x's start
                for (uint32_t i = 0; i < amount; i++) common[amount - 1 - i] = S.from_end(s0, srcs[0] > 0, i);
                for (uint32_t i = 0; i < n; i++) {
                    uint32_t u = idx_of(srcs[i]);
                    if (srcs[i] > 0) { S.remove_end(S.us[u], amount); minr[u] += amount; }      // unitig.rs:226-233
                    else { S.remove_start(S.us[u], amount); minf[u] += amount; }                 // unitig.rs:217-224
                }
                S.add_start(S.us[x], common.data(), amount); minf[x] -= amount;                  // unitig.rs:235-241
            } else {        // shift_sequence_2 (:122-142): the FIRST `amount` characters of the common prefix move onto x's end
                for (uint32_t i = 0; i < amount; i++) common[i] = S.from_start(s0, srcs[0] > 0, i);
                for (uint32_t i = 0; i < n; i++) {
                    uint32_t u = idx_of(srcs[i]);
                    if (srcs[i] > 0) { S.remove_start(S.us[u], amount); minf[u] += amount; }
                    else { S.remove_end(S.us[u], amount); minr[u] += amount; }
                }
                S.add_end(S.us[x], common.data(), amount); minr[x] -= amount;                    // unitig.rs:243-249
            }
            shifted_total += amount;
            mark_dirty(x);
            for (uint32_t i = 0; i < n; i++) mark_dirty(idx_of(srcs[i]));
        }
        passes++;
        lap("pass", shifted_total);
        if (shifted_total == 0) break;
    }

    // final forward sequences, in seed order
    out->final_off.resize(U); out->final_len.resize(U);
    uint64_t w = 0;
    for (uint32_t r = 0; r < U; r++) {
        const USeq& u = S.us[r];
        out->final_off[r] = w;
        out->final_len[r] = u.len();
        if (u.pre_len) { memcpy(seq_out + w, S.pool.data() + u.pre_off, u.pre_len); w += u.pre_len; }
        if (u.core_len) { memcpy(seq_out + w, u.core, u.core_len); w += u.core_len; }
        if (u.post_len) { memcpy(seq_out + w, S.pool.data() + u.post_off, u.post_len); w += u.post_len; }
    }
    lap("assemble sequences", w);
    out->total_len = w;
    out->passes = passes;
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
}

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
