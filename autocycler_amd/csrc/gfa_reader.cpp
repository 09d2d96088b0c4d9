// GFA text -> FinalGraph: the loader side of save_gfa, for the GFAs `compress` writes (UnitigGraph::from_gfa_lines,
// unitig_graph.rs:55-174: H line with KM:i, S lines with DP:f, L lines with 0M overlaps, P lines with LN:i / FN:Z / HD:Z).
// It is what `autocycler cluster` starts from (cluster.rs:42-43) and what `decompress` reads (decompress.rs:27-39).
#include "gfa_writer.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace ac {

namespace {
struct GfaError : std::runtime_error { using std::runtime_error::runtime_error; };
HostBlock heap_block(size_t bytes) {
    HostBlock b;
    b.p = malloc(bytes ? bytes : 1);
    if (!b.p) throw GfaError("out of memory");
    b.bytes = bytes;
    b.release = [](void* p, size_t) { free(p); };
    return b;
}
std::vector<std::pair<const char*, size_t>> split(const char* s, size_t n, char sep) {
    std::vector<std::pair<const char*, size_t>> out;
    size_t a = 0;
    for (size_t i = 0; i <= n; i++)
        if (i == n || s[i] == sep) { out.push_back({s + a, i - a}); a = i + 1; }
    return out;
}
uint64_t to_u64(const char* s, size_t n, const char* what) {
    if (n == 0) throw GfaError(std::string("Error parsing ") + what);
    uint64_t v = 0;
    for (size_t i = 0; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') throw GfaError(std::string("Error parsing ") + what);
        v = v * 10 + (uint64_t)(s[i] - '0');
    }
    return v;
}
}  // namespace

void load_gfa(const char* text, size_t len, FinalGraph* g, std::vector<SeqMeta>* seqs) {
    struct Seg { const char* seq; size_t n; double depth; };
    std::vector<Seg> segs;
    std::vector<Link> links;
    struct PathLine { uint64_t id; const char* p; size_t n; uint64_t ln; std::string fn, hd; };
    std::vector<PathLine> paths;
    uint32_t k = 0;
    size_t a = 0;
    while (a < len) {
        size_t e = a;
        while (e < len && text[e] != '\n') e++;
        size_t ll = e - a;
        if (ll && text[a + ll - 1] == '\r') ll--;
        if (ll >= 2 && text[a + 1] == '\t') {
            auto f = split(text + a, ll, '\t');
            switch (text[a]) {
                case 'H':
                    for (auto& t : f) if (t.second > 5 && !memcmp(t.first, "KM:i:", 5)) k = (uint32_t)to_u64(t.first + 5, t.second - 5, "KM:i");
                    break;
                case 'S': {   // Unitig::from_segment_line (unitig.rs:63-92): number, sequence, DP:f tag required
                    if (f.size() < 3) throw GfaError("segment line does not have enough parts");
                    uint64_t number = to_u64(f[1].first, f[1].second, "unitig number");
                    if (number != segs.size() + 1)
                        throw GfaError("this loader reads GFAs as `autocycler compress` writes them: segments numbered 1, 2, 3, ... in order");
                    double depth = -1;
                    for (size_t i = 3; i < f.size(); i++)
                        if (f[i].second > 5 && !memcmp(f[i].first, "DP:f:", 5)) depth = strtod(std::string(f[i].first + 5, f[i].second - 5).c_str(), nullptr);
                    if (depth < 0) throw GfaError("could not find a depth tag (e.g. DP:f:10.00) in the GFA segment line");
                    segs.push_back(Seg{f[2].first, f[2].second, depth});
                    break;
                }
                case 'L': {   // build_links_from_gfa (unitig_graph.rs:91-115)
                    if (f.size() < 6 || f[5].second != 2 || memcmp(f[5].first, "0M", 2))
                        throw GfaError("non-zero overlap found on the GFA link line.\nAre you sure this is an Autocycler-generated GFA file?");
                    // (numbers beyond what a signed entry holds name no unitig of a graph this library can hold: kept as 0x7FFFFFFF, refused below)
                    const uint64_t na = std::min<uint64_t>(to_u64(f[1].first, f[1].second, "segment 1 as integer"), 0x7FFFFFFFu);
                    const uint64_t nb = std::min<uint64_t>(to_u64(f[3].first, f[3].second, "segment 2 as integer"), 0x7FFFFFFFu);
                    if (na == 0) throw GfaError("link refers to nonexistent unitig: 0");
                    if (nb == 0) throw GfaError("link refers to nonexistent unitig: 0");
                    Link l;
                    l.a = (f[2].second == 1 && f[2].first[0] == '+') ? (int32_t)na : -(int32_t)na;
                    l.b = (f[4].second == 1 && f[4].first[0] == '+') ? (int32_t)nb : -(int32_t)nb;
                    links.push_back(l);
                    break;
                }
                case 'P': {   // build_paths_from_gfa (unitig_graph.rs:117-149)
                    if (f.size() < 3) throw GfaError("path line does not have enough parts");
                    PathLine pl;
                    pl.id = to_u64(f[1].first, f[1].second, "sequence id");
                    pl.p = f[2].first; pl.n = f[2].second; pl.ln = ~0ULL;
                    bool has_fn = false, has_hd = false;
                    for (size_t i = 3; i < f.size(); i++) {
                        if (f[i].second > 5 && !memcmp(f[i].first, "LN:i:", 5)) pl.ln = to_u64(f[i].first + 5, f[i].second - 5, "LN:i");
                        else if (f[i].second >= 5 && !memcmp(f[i].first, "FN:Z:", 5)) { pl.fn.assign(f[i].first + 5, f[i].second - 5); has_fn = true; }
                        else if (f[i].second >= 5 && !memcmp(f[i].first, "HD:Z:", 5)) { pl.hd.assign(f[i].first + 5, f[i].second - 5); has_hd = true; }
                    }
                    if (pl.ln == ~0ULL || !has_fn || !has_hd) throw GfaError("missing required tag in GFA path line.");
                    if (pl.id == 0 || pl.id > 32767) throw GfaError("sequence id out of range in GFA path line");
                    paths.push_back(std::move(pl));
                    break;
                }
                default: break;
            }
        }
        a = e + 1;
    }
    const size_t U = segs.size();
    g->k = k;
    g->n_kmers = 0;      // not recorded in a GFA
    g->n_unitigs = (uint32_t)U;
    uint64_t total = 0;
    for (auto& s : segs) total += s.n;
    g->seq_block = heap_block(total);
    g->meta_block = heap_block(U * 20);
    uint64_t* seq_begin = (uint64_t*)g->meta_block.p;
    double* depth = (double*)((char*)g->meta_block.p + U * 8);
    uint32_t* seq_len = (uint32_t*)((char*)g->meta_block.p + U * 16);
    uint64_t off = 0;
    for (size_t i = 0; i < U; i++) {
        memcpy((char*)g->seq_block.p + off, segs[i].seq, segs[i].n);
        seq_begin[i] = off; depth[i] = segs[i].depth; seq_len[i] = (uint32_t)segs[i].n;
        off += segs[i].n;
    }
    g->seq_begin = seq_begin; g->depth = depth; g->seq_len = seq_len;
    // link vectors are per unitig strand in file order; save_gfa walks the unitigs: forward_next, then reverse_next
    for (auto& l : links) {
        if (l.na() == 0 || l.na() > U) throw GfaError("link refers to nonexistent unitig: " + std::to_string(l.na()));
        if (l.nb() == 0 || l.nb() > U) throw GfaError("link refers to nonexistent unitig: " + std::to_string(l.nb()));
    }
    std::stable_sort(links.begin(), links.end(), [](const Link& x, const Link& y) {
        if (x.na() != y.na()) return x.na() < y.na();
        return x.a_fwd() > y.a_fwd();
    });
    g->links_block = heap_block(links.size() * sizeof(Link));
    if (!links.empty()) memcpy(g->links_block.p, links.data(), links.size() * sizeof(Link));
    g->links = (const Link*)g->links_block.p;
    g->n_links = links.size();
    // paths
    std::vector<int32_t> ent;
    g->path_off.assign(1, 0);
    seqs->clear();
    for (auto& pl : paths) {
        uint64_t sum = 0;
        for (auto& t : split(pl.p, pl.n, ',')) {
            if (t.second < 2) throw GfaError("invalid path strand");
            char st = t.first[t.second - 1];
            if (st != '+' && st != '-') throw GfaError("invalid path strand");
            uint64_t u = to_u64(t.first, t.second - 1, "unitig number in path");
            if (u == 0 || u > U) throw GfaError("unitig " + std::to_string(u) + " not found in unitig index");
            ent.push_back(st == '+' ? (int32_t)u : -(int32_t)u);
            sum += seq_len[u - 1];
        }
        if (sum != pl.ln) throw GfaError("Position calculation mismatch");      // unitig_graph.rs:173
        g->path_off.push_back(ent.size());
        seqs->push_back(SeqMeta{(uint16_t)pl.id, (uint32_t)pl.ln, pl.fn, pl.hd});
    }
    g->path_block = heap_block(ent.size() * 4);
    if (!ent.empty()) memcpy(g->path_block.p, ent.data(), ent.size() * 4);
    g->path = (const int32_t*)g->path_block.p;
    g->n_path = ent.size();
    uint64_t self_mirror = 0;
    for (auto& l : links) if (l.a == -l.b) self_mirror++;
    GraphStats st{(uint32_t)U, (links.size() + self_mirror) / 2, total};
    g->pre = st; g->post = st;
}

// reconstruct_original_sequences (unitig_graph.rs:362-388) for one sequence: the unitig strand sequences along its path.
void decompress_sequence(const FinalGraph& g, size_t seq_index, char* out) {
    for (uint64_t i = g.path_off[seq_index]; i < g.path_off[seq_index + 1]; i++) {
        int32_t v = g.path[i];
        uint32_t u = (uint32_t)(v < 0 ? -v : v) - 1;
        const char* s = g.seq(u);
        uint32_t n = g.seq_len[u];
        if (v > 0) memcpy(out, s, n);
        else for (uint32_t j = 0; j < n; j++) { char c = s[n - 1 - j]; out[j] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == '.' ? '.' : 'N'; }
        out += n;
    }
}

}  // namespace ac
