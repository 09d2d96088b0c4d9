// ORACLE — TEST INFRASTRUCTURE ONLY (see ac_oracle.hpp for the rules and the parity statement).
//
// Literal, sequential CPU restatement of the `autocycler compress` path.  Every function names the
// reference file:line it follows.  It deliberately keeps the reference's data-structure shape
// (hash map of k-mer -> occurrence list, seed-ordered walk, seen set, per-unitig link vectors built
// in the reference's push order) so that its output is the ground truth the order-free GPU
// formulation is compared with.
#include "ac_oracle.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <set>
#include <sstream>
#include <thread>
#include <unordered_set>

namespace fs = std::filesystem;

namespace oracle {

static const bool FORWARD = true, REVERSE = false;  // misc.rs `strand` consts

void quit_with_error(const std::string& text) { throw QuitError(text); }

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// misc.rs:346-376
static char complement_base(char b) {
    switch (b) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'G': return 'C';
        case 'C': return 'G';
        case '.': return '.';
        default: return 'N';
    }
}

std::string reverse_complement(std::string_view seq) {
    std::string out;
    out.reserve(seq.size());
    for (size_t i = seq.size(); i-- > 0;) out.push_back(complement_base(seq[i]));
    return out;
}

// ------------------------------------------------------------------------------------------------
// position.rs:24-52
Position Position::make(uint16_t seq_id, bool strand, size_t pos) {
    uint16_t v = seq_id;
    if (strand) v |= 0x8000;
    return Position{(uint32_t)pos, v};
}
std::string Position::to_string() const {
    return std::to_string(seq_id()) + (strand() ? "+" : "-") + std::to_string(pos);
}

// ------------------------------------------------------------------------------------------------
static std::vector<std::string> split_whitespace(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}
static std::string to_lower(std::string s) {
    for (auto& c : s) c = (char)tolower((unsigned char)c);
    return s;
}

// sequence.rs:31-59
Sequence Sequence::new_with_seq(size_t id, std::string seq, std::string filename, std::string contig_header,
                                size_t length, uint32_t half_k) {
    for (char c : seq)
        if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T'))
            quit_with_error(filename + " contains non-ACGT characters");
    std::string padding(half_k, '.');
    Sequence s;
    s.id = (uint16_t)id;
    s.forward_seq = padding + seq + padding;
    s.reverse_seq = reverse_complement(s.forward_seq);
    s.filename = std::move(filename);
    s.contig_header = std::move(contig_header);
    s.length = length;
    s.cluster = 0;
    return s;
}
// sequence.rs:61-77
Sequence Sequence::new_without_seq(uint16_t id, std::string filename, std::string contig_header, size_t length,
                                   uint16_t cluster) {
    Sequence s;
    s.id = id; s.filename = std::move(filename); s.contig_header = std::move(contig_header);
    s.length = length; s.cluster = cluster;
    return s;
}
// misc.rs:496-505
std::string Sequence::contig_name() const {
    auto parts = split_whitespace(contig_header);
    return parts.empty() ? "" : parts[0];
}
std::string Sequence::contig_description() const {
    for (size_t i = 0; i < contig_header.size(); i++)
        if (isspace((unsigned char)contig_header[i])) return contig_header.substr(i + 1);
    return "";
}
// sequence.rs:95-97
bool Sequence::is_ignored() const { return to_lower(contig_header).find("autocycler_ignore") != std::string::npos; }

// ------------------------------------------------------------------------------------------------
// kmer_graph.rs:57-69
bool Kmer::first_position() const {
    for (auto& p : positions) if (p.pos == 0) return true;
    return false;
}
std::string Kmer::to_string() const {
    std::string s(seq());
    s += ":";
    for (size_t i = 0; i < positions.size(); i++) { if (i) s += ","; s += positions[i].to_string(); }
    return s;
}

// fxhash 0.2.1 style word-at-a-time multiply-rotate hash (kmer_graph.rs:14,75).  Hash-map iteration
// order never reaches the output (kmer_graph.rs:170-171 sorts), so this has no parity obligation.
size_t FxLikeHash::operator()(std::string_view s) const noexcept {
    const uint64_t SEED = 0x517cc1b727220a95ULL;
    uint64_t h = 0;
    auto add = [&](uint64_t w) { h = ((h << 5) | (h >> 59)) ^ w; h *= SEED; };
    add(s.size());
    const char* p = s.data();
    size_t n = s.size();
    while (n >= 8) { uint64_t w; memcpy(&w, p, 8); add(w); p += 8; n -= 8; }
    if (n >= 4) { uint32_t w; memcpy(&w, p, 4); add(w); p += 4; n -= 4; }
    if (n >= 2) { uint16_t w; memcpy(&w, p, 2); add(w); p += 2; n -= 2; }
    if (n >= 1) add((uint8_t)*p);
    return (size_t)h;
}

// kmer_graph.rs:86-134
void KmerGraph::add_sequences(const std::vector<Sequence>& seqs, size_t assembly_count) {
    for (auto& s : seqs) add_sequence(s, assembly_count);
}
void KmerGraph::add_sequence(const Sequence& seq, size_t assembly_count) {
    // The reference pre-sizes every k-mer's occurrence vector (Vec::with_capacity(assembly_count), kmer_graph.rs:40): a pure
    // capacity hint, 8 x assembly_count bytes per distinct k-mer.  On diverse inputs (config E': nearly every k-mer distinct,
    // 100 assemblies) that alone is ~70 GB, so ORACLE_NO_POSITION_RESERVE=1 lets the vectors grow on demand instead — same
    // contents, same output (tests/test_oracle_kats.py::test_position_reserve_is_only_a_hint compares both).
    static const bool no_reserve = getenv("ORACLE_NO_POSITION_RESERVE") != nullptr;
    if (no_reserve) assembly_count = 0;
    size_t k = k_size, half_k = k_size / 2, two_half_k = half_k + half_k;
    const char* fraw = seq.forward_seq.data();
    const char* rraw = seq.reverse_seq.data();
    for (size_t forward_start = 0; forward_start < seq.length; forward_start++) {
        size_t forward_end = forward_start + k;
        size_t reverse_start = seq.length + two_half_k - forward_end;
        std::string_view forward_k(fraw + forward_start, k);
        std::string_view reverse_k(rraw + reverse_start, k);
        {
            auto it = kmers.find(forward_k);
            if (it == kmers.end()) {
                Kmer km{fraw + forward_start, k, {}};
                km.positions.reserve(assembly_count);  // kmer_graph.rs:40
                it = kmers.emplace(forward_k, std::move(km)).first;
            }
            it->second.positions.push_back(Position::make(seq.id, FORWARD, forward_start));
        }
        {
            auto it = kmers.find(reverse_k);
            if (it == kmers.end()) {
                Kmer km{rraw + reverse_start, k, {}};
                km.positions.reserve(assembly_count);
                it = kmers.emplace(reverse_k, std::move(km)).first;
            }
            it->second.positions.push_back(Position::make(seq.id, REVERSE, reverse_start));
        }
    }
}

static const char ALPHABET[5] = {'.', 'A', 'C', 'G', 'T'};  // kmer_graph.rs:23

// kmer_graph.rs:136-150
std::vector<const Kmer*> KmerGraph::next_kmers(std::string_view kmer) const {
    std::vector<const Kmer*> out;
    std::string next(kmer.substr(1));
    next.push_back('N');
    for (char b : ALPHABET) {
        next.back() = b;
        auto it = kmers.find(std::string_view(next));
        if (it != kmers.end()) out.push_back(&it->second);
    }
    return out;
}
// kmer_graph.rs:152-166
std::vector<const Kmer*> KmerGraph::prev_kmers(std::string_view kmer) const {
    std::vector<const Kmer*> out;
    std::string prev = "N";
    prev.append(kmer.substr(0, kmer.size() - 1));
    for (char b : ALPHABET) {
        prev[0] = b;
        auto it = kmers.find(std::string_view(prev));
        if (it != kmers.end()) out.push_back(&it->second);
    }
    return out;
}
// kmer_graph.rs:168-173 — byte-lexicographic order ('.' < 'A' < 'C' < 'G' < 'T' in ASCII).
std::vector<const Kmer*> KmerGraph::iterate_kmers() const {
    std::vector<const Kmer*> v;
    v.reserve(kmers.size());
    for (auto& kv : kmers) v.push_back(&kv.second);
    size_t k = k_size;
    std::sort(v.begin(), v.end(), [k](const Kmer* a, const Kmer* b) { return memcmp(a->pointer, b->pointer, k) < 0; });
    return v;
}
// kmer_graph.rs:175-181
const Kmer* KmerGraph::reverse(const Kmer* kmer) const {
    std::string rc = reverse_complement(kmer->seq());
    auto it = kmers.find(std::string_view(rc));
    if (it == kmers.end()) throw std::logic_error("reverse k-mer missing");
    return &it->second;
}

// ------------------------------------------------------------------------------------------------
// unitig.rs:343-365
uint32_t UnitigStrand::number() const { return unitig->number; }
int32_t UnitigStrand::signed_number() const { return strand ? (int32_t)unitig->number : -(int32_t)unitig->number; }
uint32_t UnitigStrand::length() const { return unitig->length(); }
std::string UnitigStrand::get_seq() const { return unitig->get_seq(strand); }

// unitig.rs:49-61
Unitig Unitig::from_kmers(uint32_t number, const Kmer* f, const Kmer* r) {
    Unitig u;
    u.number = number;
    u.forward_kmers.push_back(f);
    u.reverse_kmers.push_back(r);
    return u;
}
static std::vector<std::string> split_tab(const std::string& line) {
    std::vector<std::string> parts;
    size_t i = 0;
    while (true) {
        size_t j = line.find('\t', i);
        if (j == std::string::npos) { parts.push_back(line.substr(i)); break; }
        parts.push_back(line.substr(i, j - i));
        i = j + 1;
    }
    return parts;
}
// unitig.rs:63-92 (colour tags are irrelevant to compress output: Other + use_other_colour=false)
Unitig Unitig::from_segment_line(const std::string& line) {
    auto parts = split_tab(line);
    if (parts.size() < 3) quit_with_error("Segment line does not have enough parts.");
    Unitig u;
    try { size_t used; unsigned long n = std::stoul(parts[1], &used); if (used != parts[1].size()) throw 1; u.number = (uint32_t)n; }
    catch (...) { quit_with_error("Unable to parse unitig number."); }
    u.forward_seq = parts[2];
    u.reverse_seq = reverse_complement(u.forward_seq);
    bool found = false;
    for (auto& p : parts) {
        if (p.rfind("DP:f:", 0) == 0) {
            try { size_t used; double d = std::stod(p.substr(5), &used); if (used == p.size() - 5) { u.depth = d; found = true; } }
            catch (...) {}
            break;  // `.find(...)` takes the first DP tag; `.and_then(parse.ok())`
        }
    }
    if (!found)
        quit_with_error("Could not find a depth tag (e.g. DP:f:10.00) in the GFA segment line.\n"
                        "Are you sure this is an Autocycler-generated GFA file?");
    return u;
}
// unitig.rs:101-111
void Unitig::add_kmer_to_end(const Kmer* f, const Kmer* r) { forward_kmers.push_back(f); reverse_kmers.push_front(r); }
void Unitig::add_kmer_to_start(const Kmer* f, const Kmer* r) { forward_kmers.push_front(f); reverse_kmers.push_back(r); }

// unitig.rs:113-156
void Unitig::simplify_seqs() {
    if (!forward_kmers.empty()) {
        forward_seq.assign(forward_kmers.front()->seq());
        for (size_t i = 1; i < forward_kmers.size(); i++) forward_seq.push_back(forward_kmers[i]->seq().back());
    }
    if (!reverse_kmers.empty()) {
        reverse_seq.assign(reverse_kmers.front()->seq());
        for (size_t i = 1; i < reverse_kmers.size(); i++) reverse_seq.push_back(reverse_kmers[i]->seq().back());
    }
    if (!forward_kmers.empty())
        forward_positions.insert(forward_positions.end(), forward_kmers.front()->positions.begin(),
                                 forward_kmers.front()->positions.end());
    if (!reverse_kmers.empty())
        reverse_positions.insert(reverse_positions.end(), reverse_kmers.front()->positions.begin(),
                                 reverse_kmers.front()->positions.end());
    double fsum = 0, rsum = 0;
    for (auto* k : forward_kmers) fsum += (double)k->depth();
    for (auto* k : reverse_kmers) rsum += (double)k->depth();
    double favg = fsum / (double)forward_kmers.size(), ravg = rsum / (double)reverse_kmers.size();
    if (favg != ravg) throw std::logic_error("assert_eq!(forward_avg, reverse_avg) failed (unitig.rs:154)");
    depth = favg;
    forward_kmers.clear();
    reverse_kmers.clear();
}
// unitig.rs:158-166
void Unitig::trim_overlaps(size_t k_size) {
    size_t overlap = k_size / 2;
    if (forward_seq.size() < k_size) throw std::logic_error("assert forward_seq.len() >= k_size (unitig.rs:160)");
    forward_seq = forward_seq.substr(overlap);
    reverse_seq = reverse_seq.substr(0, reverse_seq.size() - overlap);
    forward_seq = forward_seq.substr(0, forward_seq.size() - overlap);
    reverse_seq = reverse_seq.substr(overlap);
    if (forward_seq.empty()) throw std::logic_error("assert !forward_seq.is_empty() (unitig.rs:165)");
}
// unitig.rs:168-172 — Rust `{:.2}`; printf("%.2f") agrees on every finite double.
std::string Unitig::gfa_segment_line() const {
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", depth);
    return "S\t" + std::to_string(number) + "\t" + forward_seq + "\tDP:f:" + buf;
}
// unitig.rs:217-249
void Unitig::remove_seq_from_start(size_t amount) {
    for (auto& p : forward_positions) p.pos += (uint32_t)amount;
    if (amount > forward_seq.size()) throw std::logic_error("assert amount <= len (unitig.rs:221)");
    forward_seq.erase(0, amount);
    reverse_seq.resize(reverse_seq.size() - amount);
}
void Unitig::remove_seq_from_end(size_t amount) {
    for (auto& p : reverse_positions) p.pos += (uint32_t)amount;
    if (amount > forward_seq.size()) throw std::logic_error("assert amount <= len (unitig.rs:230)");
    forward_seq.resize(reverse_seq.size() - amount);
    reverse_seq.erase(0, amount);
}
void Unitig::add_seq_to_start(const std::string& seq) {
    for (auto& p : forward_positions) p.pos -= (uint32_t)seq.size();
    forward_seq.insert(0, seq);
    reverse_seq = reverse_complement(forward_seq);
}
void Unitig::add_seq_to_end(const std::string& seq) {
    for (auto& p : reverse_positions) p.pos -= (uint32_t)seq.size();
    forward_seq.append(seq);
    reverse_seq = reverse_complement(forward_seq);
}

// ------------------------------------------------------------------------------------------------
// unitig_graph.rs:36-48
UnitigGraph UnitigGraph::from_kmer_graph(const KmerGraph& kg) {
    UnitigGraph g;
    g.k_size = kg.k_size;
    g.build_unitigs_from_kmer_graph(kg);
    for (auto& u : g.unitigs) u->simplify_seqs();            // :228-232
    g.create_links();
    for (auto& u : g.unitigs) u->trim_overlaps(g.k_size);    // :289-293
    g.renumber_unitigs();
    g.check_links();
    return g;
}

// unitig_graph.rs:176-226 — the sequential, seed-ordered walk with the `seen` set.
void UnitigGraph::build_unitigs_from_kmer_graph(const KmerGraph& kg) {
    std::unordered_set<std::string_view, FxLikeHash> seen;
    seen.reserve(kg.kmers.size());
    uint32_t unitig_number = 0;
    for (const Kmer* forward_kmer : kg.iterate_kmers()) {
        if (seen.count(forward_kmer->seq())) continue;
        const Kmer* reverse_kmer = kg.reverse(forward_kmer);
        unitig_number++;
        auto unitig = std::make_unique<Unitig>(Unitig::from_kmers(unitig_number, forward_kmer, reverse_kmer));
        seen.insert(forward_kmer->seq());
        seen.insert(reverse_kmer->seq());

        // Extend unitig forward
        const Kmer* for_k = forward_kmer;
        const Kmer* rev_k = reverse_kmer;
        while (true) {
            if (rev_k->first_position()) break;
            auto next = kg.next_kmers(for_k->seq());
            if (next.size() != 1) break;
            for_k = next[0];
            if (seen.count(for_k->seq())) break;
            auto prev = kg.prev_kmers(for_k->seq());
            if (prev.size() != 1) break;
            rev_k = kg.reverse(for_k);
            if (for_k->first_position()) break;
            unitig->add_kmer_to_end(for_k, rev_k);
            seen.insert(for_k->seq());
            seen.insert(rev_k->seq());
        }

        // Extend unitig backward
        for_k = forward_kmer;
        while (true) {
            if (for_k->first_position()) break;
            auto prev = kg.prev_kmers(for_k->seq());
            if (prev.size() != 1) break;
            for_k = prev[0];
            if (seen.count(for_k->seq())) break;
            auto next = kg.next_kmers(for_k->seq());
            if (next.size() != 1) break;
            rev_k = kg.reverse(for_k);
            if (rev_k->first_position()) break;
            unitig->add_kmer_to_start(for_k, rev_k);
            seen.insert(for_k->seq());
            seen.insert(rev_k->seq());
        }
        unitigs.push_back(std::move(unitig));
    }
}

// unitig_graph.rs:234-287 — push order is part of the output contract (L-line order).
void UnitigGraph::create_links() {
    size_t piece_len = k_size - 1;
    std::unordered_map<std::string, std::vector<size_t>> forward_starts, reverse_starts;
    for (size_t i = 0; i < unitigs.size(); i++) {
        forward_starts[unitigs[i]->forward_seq.substr(0, piece_len)].push_back(i);
        reverse_starts[unitigs[i]->reverse_seq.substr(0, piece_len)].push_back(i);
    }
    for (size_t i = 0; i < unitigs.size(); i++) {
        Unitig* a = unitigs[i].get();
        std::string ending_forward_seq = a->forward_seq.substr(a->forward_seq.size() - piece_len);
        std::string ending_reverse_seq = a->reverse_seq.substr(a->reverse_seq.size() - piece_len);
        auto it = forward_starts.find(ending_forward_seq);
        if (it != forward_starts.end())
            for (size_t j : it->second) {
                Unitig* b = unitigs[j].get();
                a->forward_next.push_back({b, FORWARD});   // a+ -> b+
                b->forward_prev.push_back({a, FORWARD});
                b->reverse_next.push_back({a, REVERSE});   // b- -> a-
                a->reverse_prev.push_back({b, REVERSE});
            }
        it = reverse_starts.find(ending_forward_seq);
        if (it != reverse_starts.end())
            for (size_t j : it->second) {
                Unitig* b = unitigs[j].get();
                a->forward_next.push_back({b, REVERSE});   // a+ -> b-
                b->reverse_prev.push_back({a, FORWARD});
            }
        it = forward_starts.find(ending_reverse_seq);
        if (it != forward_starts.end())
            for (size_t j : it->second) {
                Unitig* b = unitigs[j].get();
                a->reverse_next.push_back({b, FORWARD});   // a- -> b+
                b->forward_prev.push_back({a, REVERSE});
            }
    }
}

void UnitigGraph::build_unitig_index() {
    unitig_index.clear();
    for (auto& u : unitigs) unitig_index[u->number] = u.get();
}

// unitig_graph.rs:295-315 — Rust's sort_by is stable.
void UnitigGraph::renumber_unitigs() {
    std::stable_sort(unitigs.begin(), unitigs.end(),
                     [](const std::unique_ptr<Unitig>& a, const std::unique_ptr<Unitig>& b) {
                         if (a->length() != b->length()) return a->length() > b->length();
                         int c = a->forward_seq.compare(b->forward_seq);
                         if (c != 0) return c < 0;
                         return a->depth > b->depth;
                     });
    for (size_t i = 0; i < unitigs.size(); i++) unitigs[i]->number = (uint32_t)(i + 1);
    build_unitig_index();
}

// unitig_graph.rs:723-750
bool UnitigGraph::link_exists(uint32_t a, bool as, uint32_t b, bool bs) const {
    auto it = unitig_index.find(a);
    if (it == unitig_index.end()) return false;
    auto& next = as ? it->second->forward_next : it->second->reverse_next;
    for (auto& n : next) if (n.number() == b && n.strand == bs) return true;
    return false;
}
bool UnitigGraph::link_exists_prev(uint32_t a, bool as, uint32_t b, bool bs) const {
    auto it = unitig_index.find(b);
    if (it == unitig_index.end()) return false;
    auto& prev = bs ? it->second->forward_prev : it->second->reverse_prev;
    for (auto& p : prev) if (p.number() == a && p.strand == as) return true;
    return false;
}
// unitig_graph.rs:752-793
void UnitigGraph::check_links() const {
    auto fail = [](const char* m) { throw std::logic_error(std::string("check_links: ") + m); };
    for (auto& up : unitigs) {
        const Unitig& a = *up;
        auto check_next = [&](const UnitigStrand& b, bool a_strand) {
            if (!link_exists(a.number, a_strand, b.number(), b.strand)) fail("missing next link");
            if (!link_exists_prev(a.number, a_strand, b.number(), b.strand)) fail("missing prev link");
            if (!link_exists(b.number(), !b.strand, a.number, !a_strand)) fail("missing next link");
            if (!link_exists_prev(b.number(), !b.strand, a.number, !a_strand)) fail("missing prev link");
            if (!unitig_index.count(b.number())) fail("unitig missing from index");
        };
        auto check_prev = [&](const UnitigStrand& b, bool a_strand) {
            if (!link_exists(b.number(), b.strand, a.number, a_strand)) fail("missing next link");
            if (!link_exists_prev(b.number(), b.strand, a.number, a_strand)) fail("missing prev link");
            if (!link_exists(a.number, !a_strand, b.number(), !b.strand)) fail("missing next link");
            if (!link_exists_prev(a.number, !a_strand, b.number(), !b.strand)) fail("missing prev link");
            if (!unitig_index.count(b.number())) fail("unitig missing from index");
        };
        for (auto& b : a.forward_next) check_next(b, FORWARD);
        for (auto& b : a.reverse_next) check_next(b, REVERSE);
        for (auto& b : a.forward_prev) check_prev(b, FORWARD);
        for (auto& b : a.reverse_prev) check_prev(b, REVERSE);
    }
}

// unitig_graph.rs:55-74
std::pair<UnitigGraph, std::vector<Sequence>> UnitigGraph::from_gfa_lines(const std::vector<std::string>& lines) {
    UnitigGraph g;
    std::vector<std::string> link_lines, path_lines;
    for (auto& raw : lines) {
        std::string line = raw;
        while (!line.empty() && line.back() == '\n') line.pop_back();
        auto parts = split_tab(line);
        if (parts.empty()) continue;
        if (parts[0] == "H") {                       // :80-89
            for (auto& p : parts)
                if (p.rfind("KM:i:", 0) == 0) {
                    try { size_t used; unsigned long k = std::stoul(p.substr(5), &used);
                          if (used == p.size() - 5) { g.k_size = (uint32_t)k; break; } } catch (...) {}
                }
        } else if (parts[0] == "S") {
            g.unitigs.push_back(std::make_unique<Unitig>(Unitig::from_segment_line(line)));
        } else if (parts[0] == "L") {
            link_lines.push_back(line);
        } else if (parts[0] == "P") {
            path_lines.push_back(line);
        }
    }
    g.build_unitig_index();
    g.build_links_from_gfa(link_lines);
    auto seqs = g.build_paths_from_gfa(path_lines);
    g.check_links();
    return {std::move(g), std::move(seqs)};
}

// unitig_graph.rs:91-115
void UnitigGraph::build_links_from_gfa(const std::vector<std::string>& link_lines) {
    for (auto& line : link_lines) {
        auto parts = split_tab(line);
        if (parts.size() < 6 || parts[5] != "0M")
            quit_with_error("non-zero overlap found on the GFA link line.\n"
                            "Are you sure this is an Autocycler-generated GFA file?");
        uint32_t seg_1 = (uint32_t)std::stoul(parts[1]), seg_2 = (uint32_t)std::stoul(parts[3]);
        bool strand_1 = parts[2] == "+", strand_2 = parts[4] == "+";
        auto i1 = unitig_index.find(seg_1);
        if (i1 == unitig_index.end()) quit_with_error("link refers to nonexistent unitig: " + std::to_string(seg_1));
        auto i2 = unitig_index.find(seg_2);
        if (i2 == unitig_index.end()) quit_with_error("link refers to nonexistent unitig: " + std::to_string(seg_2));
        Unitig* u1 = i1->second; Unitig* u2 = i2->second;
        if (strand_1) u1->forward_next.push_back({u2, strand_2}); else u1->reverse_next.push_back({u2, strand_2});
        if (strand_2) u2->forward_prev.push_back({u1, strand_1}); else u2->reverse_prev.push_back({u1, strand_1});
    }
}

// unitig_graph.rs:971-984
static std::vector<std::pair<uint32_t, bool>> parse_unitig_path(const std::string& s) {
    std::vector<std::pair<uint32_t, bool>> out;
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        std::string u = s.substr(i, j - i);
        if (u.empty()) throw std::logic_error("Invalid path strand");
        bool strand;
        if (u.back() == '+') strand = FORWARD; else if (u.back() == '-') strand = REVERSE;
        else throw std::logic_error("Invalid path strand");
        out.push_back({(uint32_t)std::stoul(u.substr(0, u.size() - 1)), strand});
        i = j + 1;
    }
    return out;
}
static std::vector<std::pair<uint32_t, bool>> reverse_path(const std::vector<std::pair<uint32_t, bool>>& p) {
    std::vector<std::pair<uint32_t, bool>> out;
    for (size_t i = p.size(); i-- > 0;) out.push_back({p[i].first, !p[i].second});
    return out;
}

// unitig_graph.rs:117-174
std::vector<Sequence> UnitigGraph::build_paths_from_gfa(const std::vector<std::string>& path_lines) {
    std::vector<Sequence> sequences;
    for (auto& line : path_lines) {
        auto parts = split_tab(line);
        uint16_t seq_id = (uint16_t)std::stoul(parts.at(1));
        bool has_len = false, has_fn = false, has_hd = false;
        uint32_t length = 0; std::string filename, header; uint16_t cluster = 0;
        for (size_t i = 2; i < parts.size(); i++) {
            auto& p = parts[i];
            if (p.rfind("LN:i:", 0) == 0) { length = (uint32_t)std::stoul(p.substr(5)); has_len = true; }
            else if (p.rfind("FN:Z:", 0) == 0) { filename = p.substr(5); has_fn = true; }
            else if (p.rfind("HD:Z:", 0) == 0) { header = p.substr(5); has_hd = true; }
            else if (p.rfind("CL:i:", 0) == 0) { cluster = (uint16_t)std::stoul(p.substr(5)); }
        }
        if (!has_len || !has_fn || !has_hd) quit_with_error("missing required tag in GFA path line.");
        auto forward_path = parse_unitig_path(parts.at(2));
        auto rpath = reverse_path(forward_path);
        add_positions_from_path(forward_path, FORWARD, seq_id, length);
        add_positions_from_path(rpath, REVERSE, seq_id, length);
        sequences.push_back(Sequence::new_without_seq(seq_id, filename, header, length, cluster));
    }
    return sequences;
}
void UnitigGraph::add_positions_from_path(const std::vector<std::pair<uint32_t, bool>>& path, bool path_strand,
                                          uint16_t seq_id, uint32_t length) {
    uint32_t pos = 0;
    for (auto& [num, ustrand] : path) {
        auto it = unitig_index.find(num);
        if (it == unitig_index.end()) quit_with_error("unitig " + std::to_string(num) + " not found in unitig index");
        Unitig* u = it->second;
        (ustrand ? u->forward_positions : u->reverse_positions).push_back(Position::make(seq_id, path_strand, pos));
        pos += u->length();
    }
    if (pos != length) throw std::logic_error("Position calculation mismatch (unitig_graph.rs:173)");
}

// unitig_graph.rs:407-425 — full scan of all positions of all unitigs.
UnitigStrand UnitigGraph::find_starting_unitig(uint16_t seq_id) const {
    std::vector<UnitigStrand> starting;
    for (auto& u : unitigs) {
        for (auto& p : u->forward_positions)
            if (p.seq_id() == seq_id && p.strand() && p.pos == 0) starting.push_back({u.get(), FORWARD});
        for (auto& p : u->reverse_positions)
            if (p.seq_id() == seq_id && p.strand() && p.pos == 0) starting.push_back({u.get(), REVERSE});
    }
    if (starting.size() != 1) throw std::logic_error("assert_eq!(starting_unitigs.len(), 1) (unitig_graph.rs:423)");
    return starting[0];
}
// unitig_graph.rs:427-445
bool UnitigGraph::get_next_unitig(uint16_t seq_id, bool seq_strand, const Unitig* u, bool strand, uint32_t pos,
                                  UnitigStrand* next_out, uint32_t* next_pos_out) const {
    uint32_t next_pos = pos + u->length();
    auto& next_edges = strand ? u->forward_next : u->reverse_next;
    for (auto& next : next_edges) {
        auto& positions = next.strand ? next.unitig->forward_positions : next.unitig->reverse_positions;
        for (auto& p : positions)
            if (p.seq_id() == seq_id && p.strand() == seq_strand && p.pos == next_pos) {
                *next_out = next; *next_pos_out = next_pos;
                return true;
            }
    }
    return false;
}
// unitig_graph.rs:447-465
std::vector<std::pair<uint32_t, bool>> UnitigGraph::get_unitig_path_for_sequence(const Sequence& seq) const {
    std::vector<std::pair<uint32_t, bool>> path;
    UnitigStrand u = find_starting_unitig(seq.id);
    uint32_t pos = 0;
    while (true) {
        path.push_back({u.number(), u.strand});
        UnitigStrand next; uint32_t next_pos;
        if (!get_next_unitig(seq.id, FORWARD, u.unitig, u.strand, pos, &next, &next_pos)) break;
        u = next; pos = next_pos;
    }
    return path;
}
// unitig_graph.rs:390-400
std::string UnitigGraph::get_sequence_from_path(const std::vector<std::pair<uint32_t, bool>>& path) const {
    std::string s;
    for (auto& [num, strand] : path) s += unitig_index.at(num)->get_seq(strand);
    return s;
}
// unitig_graph.rs:362-388
std::vector<std::tuple<std::string, std::string, std::string>>
UnitigGraph::reconstruct_original_sequences(const std::vector<Sequence>& seqs) const {
    std::vector<std::tuple<std::string, std::string, std::string>> out;
    for (auto& seq : seqs) {
        auto path = get_unitig_path_for_sequence(seq);
        std::string s = get_sequence_from_path(path);
        if (s.size() != seq.length) throw std::logic_error("reconstructed sequence does not have expected length");
        out.emplace_back(seq.filename, seq.contig_header, std::move(s));
    }
    return out;
}
// unitig_graph.rs:474-476
uint64_t UnitigGraph::total_length() const {
    uint64_t t = 0;
    for (auto& u : unitigs) t += u->length();
    return t;
}
// unitig_graph.rs:478-507
std::pair<size_t, size_t> UnitigGraph::link_count() const {
    std::set<std::pair<int32_t, int32_t>> all_links, one_way;
    for (auto& up : unitigs) {
        int32_t a_num = (int32_t)up->number;
        for (auto& b : up->forward_next) {
            int32_t b_num = b.signed_number();
            std::pair<int32_t, int32_t> link{a_num, b_num}, rev{-b_num, -a_num};
            all_links.insert(link); all_links.insert(rev);
            one_way.insert(link > rev ? link : rev);
        }
        for (auto& b : up->reverse_next) {
            int32_t b_num = b.signed_number();
            std::pair<int32_t, int32_t> link{-a_num, b_num}, rev{-b_num, a_num};
            all_links.insert(link); all_links.insert(rev);
            one_way.insert(link > rev ? link : rev);
        }
    }
    return {all_links.size(), one_way.size()};
}
// unitig_graph.rs:317-360
std::string UnitigGraph::save_gfa_string(const std::vector<Sequence>& seqs) const {
    std::string out;
    out += "H\tVN:Z:1.0\tKM:i:" + std::to_string(k_size) + "\n";
    for (auto& u : unitigs) { out += u->gfa_segment_line(); out += "\n"; }
    for (auto& up : unitigs) {   // get_links_for_gfa(0)
        const Unitig& a = *up;
        for (auto& b : a.forward_next)
            out += "L\t" + std::to_string(a.number) + "\t+\t" + std::to_string(b.number()) + "\t" + (b.strand ? "+" : "-") + "\t0M\n";
        for (auto& b : a.reverse_next)
            out += "L\t" + std::to_string(a.number) + "\t-\t" + std::to_string(b.number()) + "\t" + (b.strand ? "+" : "-") + "\t0M\n";
    }
    for (auto& s : seqs) {       // get_gfa_path_line
        auto path = get_unitig_path_for_sequence(s);
        std::string path_str;
        for (size_t i = 0; i < path.size(); i++) {
            if (i) path_str += ",";
            path_str += std::to_string(path[i].first) + (path[i].second ? "+" : "-");
        }
        std::string cluster_tag = s.cluster > 0 ? "\tCL:i:" + std::to_string(s.cluster) : "";
        out += "P\t" + std::to_string(s.id) + "\t" + path_str + "\t*\tLN:i:" + std::to_string(s.length) +
               "\tFN:Z:" + s.filename + "\tHD:Z:" + s.contig_header + cluster_tag + "\n";
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// graph_simplification.rs:184-186
bool check_for_duplicates(const std::vector<UnitigStrand>& unitigs) {
    std::set<uint32_t> s;
    for (auto& u : unitigs) s.insert(u.number());
    return s.size() != unitigs.size();
}
// graph_simplification.rs:190-230
static void get_fixed_unitig_starts_and_ends(const UnitigGraph& graph, const std::vector<Sequence>& sequences,
                                             std::set<uint32_t>& fixed_starts, std::set<uint32_t>& fixed_ends) {
    for (auto& seq : sequences) {
        auto path = graph.get_unitig_path_for_sequence(seq);
        if (path.empty()) continue;
        if (path.front().second) fixed_starts.insert(path.front().first); else fixed_ends.insert(path.front().first);
        if (path.back().second) fixed_ends.insert(path.back().first); else fixed_starts.insert(path.back().first);
    }
    auto starts_copy = fixed_starts, ends_copy = fixed_ends;
    for (uint32_t u : starts_copy)
        for (auto& up : graph.unitig_index.at(u)->forward_prev) {
            if (up.strand) fixed_ends.insert(up.number()); else fixed_starts.insert(up.number());
        }
    for (uint32_t u : ends_copy)
        for (auto& down : graph.unitig_index.at(u)->forward_next) {
            if (down.strand) fixed_starts.insert(down.number()); else fixed_ends.insert(down.number());
        }
}
// graph_simplification.rs:233-255
std::vector<UnitigStrand> get_exclusive_inputs(const Unitig* unitig) {
    std::vector<UnitigStrand> inputs;
    for (auto& prev : unitig->forward_prev) {
        auto& next = prev.strand ? prev.unitig->forward_next : prev.unitig->reverse_next;
        bool exclusive = next.size() == 1 && next[0].strand && next[0].number() == unitig->number;
        if (!exclusive) return {};
        inputs.push_back(prev);
    }
    for (auto& inp : inputs) if (inp.number() == unitig->number) return {};
    return inputs;
}
// graph_simplification.rs:258-280
std::vector<UnitigStrand> get_exclusive_outputs(const Unitig* unitig) {
    std::vector<UnitigStrand> outputs;
    for (auto& next : unitig->forward_next) {
        auto& prevs = next.strand ? next.unitig->forward_prev : next.unitig->reverse_prev;
        bool exclusive = prevs.size() == 1 && prevs[0].strand && prevs[0].number() == unitig->number;
        if (!exclusive) return {};
        outputs.push_back(next);
    }
    for (auto& o : outputs) if (o.number() == unitig->number) return {};
    return outputs;
}
static bool starts_with(const std::string& s, const std::string& p) {
    return s.size() >= p.size() && memcmp(s.data(), p.data(), p.size()) == 0;
}
// graph_simplification.rs:283-295
std::string get_common_start_seq(const std::vector<UnitigStrand>& unitigs) {
    std::vector<std::string> seqs;
    for (auto& u : unitigs) seqs.push_back(u.get_seq());
    if (seqs.empty()) return "";
    std::string prefix = seqs[0];
    for (auto& seq : seqs)
        while (!starts_with(seq, prefix)) {
            prefix.pop_back();
            if (prefix.empty()) return "";
        }
    return prefix;
}
// graph_simplification.rs:298-312
std::string get_common_end_seq(const std::vector<UnitigStrand>& unitigs) {
    std::vector<std::string> seqs;
    for (auto& u : unitigs) { std::string s = u.get_seq(); std::reverse(s.begin(), s.end()); seqs.push_back(s); }
    if (seqs.empty()) return "";
    std::string suffix = seqs[0];
    for (auto& seq : seqs)
        while (!starts_with(seq, suffix)) {
            suffix.pop_back();
            if (suffix.empty()) return "";
        }
    std::reverse(suffix.begin(), suffix.end());
    return suffix;
}
// graph_simplification.rs:145-161
static void avoid_zero_len_unitigs(std::string& common_seq, const std::vector<UnitigStrand>& sources, bool trim_from_start) {
    if (common_seq.empty()) return;
    uint32_t dup = check_for_duplicates(sources) ? 2 : 1;
    uint32_t min_source_len = UINT32_MAX;
    for (auto& s : sources) min_source_len = std::min(min_source_len, s.length());
    while (min_source_len <= (uint32_t)common_seq.size() * dup) {
        if (trim_from_start) common_seq.erase(0, 1); else common_seq.pop_back();
    }
}
// graph_simplification.rs:164-181
static void avoid_start_of_path(std::string& common_seq, const Unitig* dest, bool trim_from_start) {
    if (common_seq.empty()) return;
    auto any_le = [&](const std::vector<Position>& ps) {
        for (auto& p : ps) if (p.pos <= (uint32_t)common_seq.size()) return true;
        return false;
    };
    if (trim_from_start) { while (any_le(dest->forward_positions)) common_seq.erase(0, 1); }
    else { while (any_le(dest->reverse_positions)) common_seq.pop_back(); }
}
// graph_simplification.rs:89-119
static size_t shift_sequence_1(const std::vector<UnitigStrand>& sources, Unitig* dest) {
    std::string common_seq = get_common_end_seq(sources);
    avoid_zero_len_unitigs(common_seq, sources, true);
    avoid_start_of_path(common_seq, dest, true);
    size_t amount = common_seq.size();
    if (amount == 0) return 0;
    for (auto& s : sources) {
        if (s.strand) s.unitig->remove_seq_from_end(amount); else s.unitig->remove_seq_from_start(amount);
    }
    dest->add_seq_to_start(common_seq);
    return amount;
}
// graph_simplification.rs:122-142
static size_t shift_sequence_2(Unitig* dest, const std::vector<UnitigStrand>& sources) {
    std::string common_seq = get_common_start_seq(sources);
    avoid_zero_len_unitigs(common_seq, sources, false);
    avoid_start_of_path(common_seq, dest, false);
    size_t amount = common_seq.size();
    if (amount == 0) return 0;
    for (auto& s : sources) {
        if (s.strand) s.unitig->remove_seq_from_start(amount); else s.unitig->remove_seq_from_end(amount);
    }
    dest->add_seq_to_end(common_seq);
    return amount;
}
// graph_simplification.rs:43-86
size_t expand_repeats(UnitigGraph& graph, const std::vector<Sequence>& seqs) {
    std::set<uint32_t> fixed_starts, fixed_ends;
    get_fixed_unitig_starts_and_ends(graph, seqs, fixed_starts, fixed_ends);
    size_t total = 0;
    for (auto& up : graph.unitigs) {
        Unitig* unitig = up.get();
        uint32_t number = unitig->number;
        auto inputs = get_exclusive_inputs(unitig);
        if (inputs.size() >= 2 && !fixed_starts.count(number)) {
            bool can_shift = true;
            for (auto& in : inputs)
                if ((in.strand && fixed_ends.count(in.number())) || (!in.strand && fixed_starts.count(in.number()))) can_shift = false;
            if (can_shift) total += shift_sequence_1(inputs, unitig);
        }
        auto outputs = get_exclusive_outputs(unitig);
        if (outputs.size() >= 2 && !fixed_ends.count(number)) {
            bool can_shift = true;
            for (auto& o : outputs)
                if ((o.strand && fixed_starts.count(o.number())) || (!o.strand && fixed_ends.count(o.number()))) can_shift = false;
            if (can_shift) total += shift_sequence_2(unitig, outputs);
        }
    }
    return total;
}
// graph_simplification.rs:26-40
void simplify_structure(UnitigGraph& graph, const std::vector<Sequence>& seqs) {
    while (expand_repeats(graph, seqs) > 0) {}
    graph.renumber_unitigs();
}

// ------------------------------------------------------------------------------------------------
// compress.rs:239-270 — min_by returns the FIRST minimum; the comparator is a total order on
// (dots asc, freq desc, bytes asc) so the first minimum is the unique best value anyway.
std::string find_best_match(const std::vector<std::string>& matches) {
    if (matches.empty()) throw std::logic_error("There should be at least one match");
    std::unordered_map<std::string, std::pair<size_t, size_t>> counts;
    for (auto& m : matches) {
        auto& e = counts[m];
        e.first += 1;
        e.second = (size_t)std::count(m.begin(), m.end(), '.');
    }
    const std::string* best = &matches[0];
    for (size_t i = 1; i < matches.size(); i++) {
        const std::string& a = matches[i];
        auto& ca = counts[a]; auto& cb = counts[*best];
        bool less;
        if (ca.second != cb.second) less = ca.second < cb.second;
        else if (ca.first != cb.first) less = ca.first > cb.first;
        else less = a < *best;
        if (less) best = &a;
    }
    return *best;
}

// regex::bytes find_iter semantics for a pattern made only of literals and '.': leftmost,
// non-overlapping; '.' matches any one byte (the haystack never contains '\n').
static void find_iter_all(const std::string& pat, const std::string& hay, std::vector<std::string>& out) {
    size_t m = pat.size();
    if (m == 0 || hay.size() < m) return;
    size_t i = 0;
    while (i + m <= hay.size()) {
        bool ok = true;
        for (size_t j = 0; j < m; j++)
            if (pat[j] != '.' && pat[j] != hay[i + j]) { ok = false; break; }
        if (ok) { out.push_back(hay.substr(i, m)); i += m; } else i++;
    }
}

// compress.rs:202-236
void sequence_end_repair(std::vector<Sequence>& sequences, uint32_t k_size, int threads) {
    size_t overlap_size = k_size - 1;
    if (overlap_size == 0) return;  // k=1: empty patterns, empty splices
    std::vector<std::string> all_seqs;
    for (auto& s : sequences) { all_seqs.push_back(s.forward_seq); all_seqs.push_back(s.reverse_seq); }
    auto work = [&](size_t idx) {
        Sequence& seq = sequences[idx];
        std::string start = seq.forward_seq.substr(0, overlap_size);
        std::string end = seq.forward_seq.substr(seq.forward_seq.size() - overlap_size);
        std::vector<std::string> all_matches;
        for (auto& s : all_seqs) find_iter_all(start, s, all_matches);
        std::string best = find_best_match(all_matches);
        seq.forward_seq.replace(0, overlap_size, best);
        all_matches.clear();
        for (auto& s : all_seqs) find_iter_all(end, s, all_matches);
        best = find_best_match(all_matches);
        seq.forward_seq.replace(seq.forward_seq.size() - overlap_size, overlap_size, best);
        seq.reverse_seq = reverse_complement(seq.forward_seq);
    };
    if (threads <= 1 || sequences.size() <= 1) {
        for (size_t i = 0; i < sequences.size(); i++) work(i);
    } else {  // rayon par_iter_mut: each sequence is independent (matches come from the snapshot)
        std::vector<std::thread> pool;
        std::atomic<size_t> next{0};
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < sequences.size();) work(i); });
        for (auto& t : pool) t.join();
    }
}

// ------------------------------------------------------------------------------------------------
// misc.rs:87-96 — note the reference's operator precedence: `gz && stem.ends_with(".fasta") ||
// stem.ends_with(".fna") || stem.ends_with(".fa")`.
static bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
static bool is_assembly_file(const fs::path& p) {
    if (!fs::is_regular_file(p)) return false;
    std::string ext = p.extension().string();
    if (!ext.empty()) ext = ext.substr(1);
    std::string stem = p.stem().string();
    return ext == "fasta" || ext == "fna" || ext == "fa" ||
           ((ext == "gz" && ends_with(stem, ".fasta")) || ends_with(stem, ".fna") || ends_with(stem, ".fa"));
}
// misc.rs:65-84
std::vector<std::string> find_all_assemblies(const std::string& dir) {
    std::vector<std::string> all;
    std::error_code ec;
    fs::directory_iterator it(dir, ec);
    if (ec) quit_with_error("unable to read directory " + dir + "\n" + ec.message());
    for (auto& e : it) if (is_assembly_file(e.path())) all.push_back(e.path().string());
    std::sort(all.begin(), all.end());
    if (all.empty()) quit_with_error("no assemblies found in " + dir);
    return all;
}

static std::string read_file_maybe_gz(const std::string& filename) {
    // misc.rs:268-280 is_file_gzipped; gzread handles multi-member gzip like MultiGzDecoder (:323).
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) quit_with_error("unable to open " + filename);
    unsigned char magic[2] = {0, 0};
    size_t n = fread(magic, 1, 2, f);
    fclose(f);
    std::string data;
    if (n == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        gzFile g = gzopen(filename.c_str(), "rb");
        if (!g) quit_with_error("unable to load " + filename);
        char buf[1 << 16];
        int r;
        while ((r = gzread(g, buf, sizeof buf)) > 0) data.append(buf, (size_t)r);
        if (r < 0) { gzclose(g); quit_with_error("unable to load " + filename); }
        gzclose(g);
    } else {
        std::ifstream in(filename, std::ios::binary);
        std::stringstream ss; ss << in.rdbuf();
        data = ss.str();
    }
    return data;
}

// misc.rs:145-195, 282-355
std::vector<std::tuple<std::string, std::string, std::string>> load_fasta(const std::string& filename) {
    std::error_code ec;
    if (fs::file_size(filename, ec) == 0 && !ec) quit_with_error(filename + " is an empty file");
    std::string data = read_file_maybe_gz(filename);
    std::vector<std::tuple<std::string, std::string, std::string>> seqs;
    std::string name, header, sequence;
    size_t i = 0;
    auto flush = [&] {
        for (auto& c : sequence) c = (char)toupper((unsigned char)c);  // make_ascii_uppercase
        seqs.emplace_back(name, header, sequence);
        sequence.clear();
    };
    while (i < data.size()) {   // BufRead::lines: split on '\n', strip one trailing '\r'
        size_t j = data.find('\n', i);
        if (j == std::string::npos) j = data.size();
        std::string text = data.substr(i, j - i);
        if (!text.empty() && text.back() == '\r') text.pop_back();
        i = j + 1;
        if (text.empty()) continue;
        if (text[0] == '>') {
            if (!name.empty()) flush();
            header = text.substr(1);
            auto pieces = split_whitespace(header);
            if (pieces.empty()) quit_with_error(filename + " is not correctly formatted");
            name = pieces[0];
        } else {
            if (name.empty()) quit_with_error(filename + " is not correctly formatted");
            sequence += text;
        }
    }
    if (!name.empty()) flush();
    // check_load_fasta, misc.rs:173-193
    if (seqs.empty()) quit_with_error(filename + " contains no sequences");
    for (auto& [n, h, s] : seqs) {
        if (n.empty()) quit_with_error(filename + " has an unnamed sequence");
        if (s.empty()) quit_with_error(filename + " has an empty sequence");
    }
    std::set<std::string> names;
    for (auto& [n, h, s] : seqs)
        if (!names.insert(n).second) quit_with_error(filename + " has a duplicate name: " + n);
    return seqs;
}

// compress.rs:84-95
static void check_sequence_count(const std::vector<Sequence>& sequences, size_t assembly_count, uint32_t max_contigs) {
    double n = (double)sequences.size();
    if (n == 0.0) quit_with_error("no sequences found in input assemblies");
    double mean = n / (double)assembly_count;
    if (mean > (double)max_contigs) {
        char buf[64]; snprintf(buf, sizeof buf, "%.1f", mean);
        quit_with_error(std::string("the mean number of contigs per input assembly (") + buf +
                        ") exceeds the allowed threshold (" + std::to_string(max_contigs) +
                        "). Are your input assemblies fragmented or contaminated?");
    }
}

// compress.rs:98-133
std::pair<std::vector<Sequence>, size_t> load_sequences(const std::string& assemblies_dir, uint32_t k_size,
                                                        InputAssemblyMetrics& metrics, uint32_t max_contigs, int threads) {
    auto assemblies = find_all_assemblies(assemblies_dir);
    uint32_t half_k = k_size / 2;
    size_t seq_id = 0;
    std::vector<Sequence> sequences;
    for (auto& assembly : assemblies) {
        AssemblyDetails details;
        details.filename = assembly;   // metrics.rs:85 — full path string
        for (auto& [name, header, seq] : load_fasta(assembly)) {
            size_t seq_len = seq.size();
            if (seq_len < k_size) continue;
            seq_id++;
            if (seq_id > 32767) quit_with_error("no more than 32767 input sequences are allowed");
            auto toks = split_whitespace(header);
            std::string contig_header;
            for (size_t i = 0; i < toks.size(); i++) { if (i) contig_header += " "; contig_header += toks[i]; }
            std::string filename = fs::path(assembly).filename().string();
            Sequence s = Sequence::new_with_seq(seq_id, seq, filename, contig_header, seq_len, half_k);
            details.contigs.push_back({s.contig_name(), s.contig_description(), (uint64_t)s.length});
            if (!s.is_ignored()) sequences.push_back(std::move(s));
        }
        metrics.details.push_back(std::move(details));
    }
    check_sequence_count(sequences, assemblies.size(), max_contigs);
    sequence_end_repair(sequences, k_size, threads);
    return {std::move(sequences), assemblies.size()};
}

// metrics.rs:65-107,256-260 — serde_yaml 0.9.34 block style.  No reference test pins these bytes
// ("parity unpinned" for the YAML side output); strings are quoted with the common serde_yaml rules.
static std::string yaml_scalar(const std::string& s) {
    auto plain_ok = [&] {
        if (s.empty()) return false;
        static const char* specials[] = {"true", "false", "null", "~", "yes", "no", "on", "off", "y", "n",
                                         "True", "False", "Null", "NULL", "TRUE", "FALSE", ".nan", ".inf", "-.inf"};
        for (auto sp : specials) if (s == sp) return false;
        if (isspace((unsigned char)s.front()) || isspace((unsigned char)s.back())) return false;
        if (strchr("-?:,[]{}#&*!|>'\"%@`", s.front())) {
            if (!((s.front() == '-' || s.front() == '?' || s.front() == ':') && s.size() > 1 && !isspace((unsigned char)s[1]))) return false;
        }
        for (size_t i = 0; i < s.size(); i++) {
            unsigned char c = (unsigned char)s[i];
            if (c < 0x20 || c == 0x7f) return false;
            if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) return false;
            if (c == '#' && i > 0 && s[i - 1] == ' ') return false;
        }
        // things that parse as numbers must be quoted
        char* end = nullptr;
        strtod(s.c_str(), &end);
        if (end && *end == '\0') return false;
        return true;
    };
    if (plain_ok()) return s;
    std::string out = "'";
    for (char c : s) { if (c == '\'') out += "''"; else out.push_back(c); }
    out += "'";
    return out;
}
std::string InputAssemblyMetrics::to_yaml() const {
    std::string y;
    y += "input_assemblies_count: " + std::to_string(input_assemblies_count) + "\n";
    y += "input_assemblies_total_contigs: " + std::to_string(input_assemblies_total_contigs) + "\n";
    y += "input_assemblies_total_length: " + std::to_string(input_assemblies_total_length) + "\n";
    y += "compressed_unitig_count: " + std::to_string(compressed_unitig_count) + "\n";
    y += "compressed_unitig_total_length: " + std::to_string(compressed_unitig_total_length) + "\n";
    if (details.empty()) { y += "input_assembly_details: []\n"; return y; }
    y += "input_assembly_details:\n";
    for (auto& a : details) {
        y += "- filename: " + yaml_scalar(a.filename) + "\n";
        if (a.contigs.empty()) { y += "  contigs: []\n"; continue; }
        y += "  contigs:\n";
        for (auto& c : a.contigs) {
            y += "  - name: " + yaml_scalar(c.name) + "\n";
            y += "    description: " + yaml_scalar(c.description) + "\n";
            y += "    length: " + std::to_string(c.length) + "\n";
        }
    }
    return y;
}

// compress.rs:42-47 (build_kmer_graph, build_unitig_graph, simplify_unitig_graph, save_gfa)
std::string compress_sequences(const std::vector<Sequence>& seqs, size_t assembly_count, uint32_t k_size,
                               GraphStats* stats, StageTimes* times) {
    double t0 = now_s();
    KmerGraph kg(k_size);
    kg.add_sequences(seqs, assembly_count);
    double t1 = now_s();
    UnitigGraph ug = UnitigGraph::from_kmer_graph(kg);
    double t2 = now_s();
    if (stats) {
        stats->kmers = kg.kmers.size();
        stats->unitigs_pre = ug.unitigs.size(); stats->links_pre = ug.link_count().second; stats->length_pre = ug.total_length();
    }
    double t2b = now_s();
    simplify_structure(ug, seqs);
    double t3 = now_s();
    if (stats) {
        stats->unitigs_post = ug.unitigs.size(); stats->links_post = ug.link_count().second; stats->length_post = ug.total_length();
    }
    double t3b = now_s();
    std::string gfa = ug.save_gfa_string(seqs);
    double t4 = now_s();
    if (times) { times->kmer_graph = t1 - t0; times->unitig_graph = t2 - t1; times->simplify = t3 - t2b; times->save = t4 - t3b; }
    return gfa;
}

// compress.rs:32-50
void compress_dir(const std::string& assemblies_dir, const std::string& autocycler_dir, uint32_t k_size,
                  uint32_t max_contigs, int threads, GraphStats* stats, StageTimes* times) {
    // check_settings, compress.rs:53-62 (the CLI range checks; tests may call load_sequences with other k)
    if (!fs::exists(assemblies_dir)) quit_with_error("directory does not exist: " + assemblies_dir);
    if (!fs::is_directory(assemblies_dir)) quit_with_error(assemblies_dir + " is not a directory");
    if (fs::exists(autocycler_dir) && !fs::is_directory(autocycler_dir)) quit_with_error(autocycler_dir + " exists but is not a directory");
    if (k_size < 11) quit_with_error("--kmer cannot be less than 11");
    if (k_size > 501) quit_with_error("--kmer cannot be greater than 501");
    if (k_size % 2 == 0) quit_with_error("--kmer must be odd");
    if (threads < 1) quit_with_error("--threads cannot be less than 1");
    if (threads > 100) quit_with_error("--threads cannot be greater than 100");
    std::error_code ec;
    fs::create_directories(autocycler_dir, ec);
    if (ec) quit_with_error("failed to create directory " + autocycler_dir + "\n" + ec.message());
    InputAssemblyMetrics metrics;
    double t0 = now_s();
    auto [sequences, assembly_count] = load_sequences(assemblies_dir, k_size, metrics, max_contigs, threads);
    double t1 = now_s();
    GraphStats st;
    std::string gfa = compress_sequences(sequences, assembly_count, k_size, &st, times);
    if (times) times->load = t1 - t0;
    if (stats) *stats = st;
    { std::ofstream out(fs::path(autocycler_dir) / "input_assemblies.gfa", std::ios::binary); out << gfa; }
    metrics.input_assemblies_count = (uint32_t)assembly_count;
    metrics.input_assemblies_total_contigs = (uint32_t)sequences.size();
    uint64_t total = 0; for (auto& s : sequences) total += s.length;
    metrics.input_assemblies_total_length = total;
    metrics.compressed_unitig_count = (uint32_t)st.unitigs_post;
    metrics.compressed_unitig_total_length = st.length_post;
    { std::ofstream out(fs::path(autocycler_dir) / "input_assemblies.yaml", std::ios::binary); out << metrics.to_yaml(); }
}

}  // namespace oracle
