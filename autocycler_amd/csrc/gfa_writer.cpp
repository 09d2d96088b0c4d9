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

// The lines are put together in a 64 KB block on the stack with plain pointer writes and appended to the string a block at a time: one
// character at a time through std::string::push_back the 91 MB of config C took 0.26 s on one core, this way 0.07 s (tools/microbench/
// gfa_write_bench.cpp) — 10.6 M path entries are most of that text.
namespace {
struct Out {
    std::string& s; char buf[1 << 16]; size_t n = 0;
    explicit Out(std::string& dst) : s(dst) {}
    ~Out() { flush(); }
    void flush() { s.append(buf, n); n = 0; }
    void room(size_t need) { if (n + need > sizeof buf) flush(); }
    void ch(char c) { room(1); buf[n++] = c; }
    void lit(const char* p, size_t len) { if (len > sizeof buf / 2) { flush(); s.append(p, len); return; } room(len); memcpy(buf + n, p, len); n += len; }
    void str(const std::string& x) { lit(x.data(), x.size()); }
    void u64(uint64_t v) {      // decimal, two digits per step
        static const char pairs[] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        room(24);
        char tmp[24]; int k = 24;
        while (v >= 100) { const unsigned r = (unsigned)(v % 100); v /= 100; tmp[--k] = pairs[2 * r + 1]; tmp[--k] = pairs[2 * r]; }
        if (v >= 10) { tmp[--k] = pairs[2 * v + 1]; tmp[--k] = pairs[2 * v]; } else tmp[--k] = (char)('0' + v);
        memcpy(buf + n, tmp + k, (size_t)(24 - k)); n += (size_t)(24 - k);
    }
};
}  // namespace
#define AC_LIT(o, text) (o).lit(text, sizeof(text) - 1)

// Rust's {:.2} of a depth.  Depths of a compress graph are whole numbers (occurrence counts): those skip snprintf.
static inline void put_depth(Out& o, double d) {
    if (d >= 0 && d < 9e15 && d == (double)(uint64_t)d) { o.u64((uint64_t)d); AC_LIT(o, ".00"); return; }
    char buf[64]; const int len = snprintf(buf, sizeof buf, "%.2f", d);
    o.lit(buf, (size_t)len);
}
static void s_lines(const FinalGraph& g, uint32_t a, uint32_t b, std::string& out) {      // S lines of unitigs [a, b)
    Out o(out);
    for (uint32_t i = a; i < b; i++) {
        AC_LIT(o, "S\t"); o.u64((uint64_t)i + 1); o.ch('\t'); o.lit(g.seq(i), g.seq_len[i]); AC_LIT(o, "\tDP:f:");
        put_depth(o, g.depth[i]); o.ch('\n');
    }
}
static void l_lines(const FinalGraph& g, uint64_t a, uint64_t b, std::string& out) {      // L lines [a, b)
    Out o(out);
    for (uint64_t li = a; li < b; li++) {
        const Link& l = g.links[li];
        AC_LIT(o, "L\t"); o.u64(l.na()); if (l.a_fwd()) AC_LIT(o, "\t+\t"); else AC_LIT(o, "\t-\t"); o.u64(l.nb());
        if (l.b_fwd()) AC_LIT(o, "\t+\t0M\n"); else AC_LIT(o, "\t-\t0M\n");
    }
}

std::string gfa_string(const FinalGraph& g, const std::vector<SeqMeta>& seqs, int parts) {
    std::string out;
    size_t est = 64;
    if (parts & 1) est += g.post.total_length + (size_t)g.n_unitigs * 32 + g.n_links * 32;
    if (parts & 2) est += g.n_path * 10 + seqs.size() * 256;
    out.reserve(est);
    if (parts & 1) {
        out += "H\tVN:Z:1.0\tKM:i:"; put_u64(out, g.k); out.push_back('\n');
        s_lines(g, 0, g.n_unitigs, out);
        l_lines(g, 0, g.n_links, out);
    }
    if (parts & 2) {
        Out o(out);
        for (size_t s = 0; s < seqs.size(); s++) {
            AC_LIT(o, "P\t"); o.u64(seqs[s].id); o.ch('\t');
            for (uint64_t i = g.path_off[s]; i < g.path_off[s + 1]; i++) {
                if (i != g.path_off[s]) o.ch(',');
                const int32_t v = g.path[i];
                o.u64((uint64_t)(v < 0 ? -(int64_t)v : (int64_t)v)); o.ch(v < 0 ? '-' : '+');
            }
            AC_LIT(o, "\t*\tLN:i:"); o.u64(seqs[s].length);
            AC_LIT(o, "\tFN:Z:"); o.str(seqs[s].filename); AC_LIT(o, "\tHD:Z:"); o.str(seqs[s].contig_header);
            o.ch('\n');   // cluster is 0 in compress output: no CL:i tag (unitig_graph.rs:357)
        }
    }
    return out;
}

// The same text in pieces built by several threads — the header, the S lines split by sequence bytes, the L lines, the P lines
// split by path entries: concatenated in order they are exactly gfa_string().  Used by the whole-command driver.
std::vector<std::string> gfa_chunks(const FinalGraph& g, const std::vector<SeqMeta>& seqs, int threads) {
    size_t S = seqs.size();
    int T = std::max(1, std::min<int>(threads, 64));
    std::vector<size_t> cut{0};          // sequence ranges with about equal numbers of path entries
    uint64_t total = g.path_off.empty() ? 0 : g.path_off[S], per = total / (uint64_t)T + 1;
    for (size_t s = 0; s < S; s++)
        if (g.path_off[s + 1] >= per * cut.size() && s + 1 < S) cut.push_back(s + 1);
    cut.push_back(S);
    std::vector<uint32_t> ucut{0};       // unitig ranges with about equal S-line bytes (the unitigs come longest first)
    {
        uint64_t bytes = 0, all = (uint64_t)g.post.total_length + (uint64_t)g.n_unitigs * 24, per_s = all / (uint64_t)T + 1;
        for (uint32_t i = 0; i < g.n_unitigs; i++) {
            bytes += (uint64_t)g.seq_len[i] + 24;
            if (bytes >= per_s * ucut.size() && i + 1 < g.n_unitigs) ucut.push_back(i + 1);
        }
        ucut.push_back(g.n_unitigs);
    }
    const size_t n_s = ucut.size() - 1, n_l = g.n_links ? (size_t)std::min<uint64_t>((uint64_t)T, g.n_links / 65536 + 1) : 0, n_p = cut.size() - 1;
    std::vector<std::string> out(1 + n_s + n_l + n_p);      // header | S pieces | L pieces | P pieces
    auto p_lines = [&](size_t a, size_t b, std::string* dst) {
        FinalGraph view;       // a shallow view restricted to [a, b): same arrays, shifted offsets
        std::vector<SeqMeta> sub(seqs.begin() + (long)a, seqs.begin() + (long)b);
        view.k = g.k; view.path = g.path; view.path_off.assign(g.path_off.begin() + (long)a, g.path_off.begin() + (long)b + 1);
        view.n_path = view.path_off.back() - view.path_off.front();
        *dst = gfa_string(view, sub, 2);
    };
    std::vector<std::thread> pool;
    for (size_t i = 0; i < n_p; i++) pool.emplace_back(p_lines, cut[i], cut[i + 1], &out[1 + n_s + n_l + i]);
    for (size_t i = 0; i < n_s; i++)
        pool.emplace_back([&g, &ucut, &out, i] {
            std::string& d = out[1 + i];
            uint64_t est = 0;
            for (uint32_t u = ucut[i]; u < ucut[i + 1]; u++) est += (uint64_t)g.seq_len[u] + 32;
            d.reserve(est);
            s_lines(g, ucut[i], ucut[i + 1], d);
        });
    for (size_t i = 0; i < n_l; i++)
        pool.emplace_back([&g, &out, n_s, n_l, i] {
            const uint64_t a = g.n_links * i / n_l, b = g.n_links * (i + 1) / n_l;
            std::string& d = out[1 + n_s + i];
            d.reserve((b - a) * 28);
            l_lines(g, a, b, d);
        });
    out[0] = "H\tVN:Z:1.0\tKM:i:"; put_u64(out[0], g.k); out[0].push_back('\n');
    for (auto& t : pool) t.join();
    return out;
}

}  // namespace ac
