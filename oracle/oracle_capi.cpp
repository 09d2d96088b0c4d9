// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over ac_oracle.cpp for the pytest harness
// (ctypes), smoke() and bench.py's cpu_baseline leg.  Returned strings are malloc'd; free with
// orc_free().  Every call returns 0 on success; on failure orc_last_error() holds the message
// (the reference would have printed "Error: ..." and exited, or panicked: misc.rs:131-142).
#include <map>
#include <set>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "ac_oracle.hpp"

using namespace oracle;

static thread_local std::string g_err;

static char* dup_str(const std::string& s) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}
static std::vector<std::string> split_lines(const char* text) {
    std::vector<std::string> lines;
    std::string s(text);
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        lines.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    return lines;
}
template <class F>
static int guarded(F&& f) {
    try { f(); return 0; }
    catch (const QuitError& e) { g_err = std::string("Error: ") + e.what(); return 1; }
    catch (const std::exception& e) { g_err = std::string("panic: ") + e.what(); return 2; }
    catch (...) { g_err = "panic: unknown"; return 2; }
}

struct orc_seqs { std::vector<Sequence> seqs; size_t assembly_count = 0; InputAssemblyMetrics metrics; };

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
void orc_free(void* p) { free(p); }

// compress.rs:98-133
int orc_load_sequences(const char* dir, uint32_t k, uint32_t max_contigs, int threads, orc_seqs** out) {
    return guarded([&] {
        auto h = std::make_unique<orc_seqs>();
        auto r = load_sequences(dir, k, h->metrics, max_contigs, threads);
        h->seqs = std::move(r.first);
        h->assembly_count = r.second;
        *out = h.release();
    });
}
// Sequence::new_with_seq (sequence.rs:31-59) for ids 1..n, optionally followed by
// sequence_end_repair (compress.rs:202-236).
int orc_seqs_from_raw(uint32_t k, uint32_t n, const char** seqs, const char** filenames, const char** headers,
                      int repair, int threads, uint32_t assembly_count, orc_seqs** out) {
    return guarded([&] {
        auto h = std::make_unique<orc_seqs>();
        for (uint32_t i = 0; i < n; i++) {
            std::string s(seqs[i]);
            size_t len = s.size();
            h->seqs.push_back(Sequence::new_with_seq(i + 1, s, filenames[i], headers[i], len, k / 2));
        }
        if (repair) sequence_end_repair(h->seqs, k, threads);
        h->assembly_count = assembly_count;
        *out = h.release();
    });
}
uint32_t orc_seqs_count(const orc_seqs* h) { return (uint32_t)h->seqs.size(); }
uint32_t orc_seqs_assembly_count(const orc_seqs* h) { return (uint32_t)h->assembly_count; }
int orc_seq_get(const orc_seqs* h, uint32_t i, const char** fwd, const char** rev, uint32_t* length, uint16_t* id,
                const char** filename, const char** header) {
    if (i >= h->seqs.size()) { g_err = "index out of range"; return 1; }
    const Sequence& s = h->seqs[i];
    if (fwd) *fwd = s.forward_seq.c_str();
    if (rev) *rev = s.reverse_seq.c_str();
    if (length) *length = (uint32_t)s.length;
    if (id) *id = s.id;
    if (filename) *filename = s.filename.c_str();
    if (header) *header = s.contig_header.c_str();
    return 0;
}
void orc_seqs_free(orc_seqs* h) { delete h; }

// compress.rs:42-47 on loaded sequences.  stats[7] = kmers, (unitigs, links, length) pre, post.
// times[6] = load, repair, kmer_graph, unitig_graph, simplify, save (seconds).
int orc_compress(const orc_seqs* h, uint32_t k, char** gfa, uint64_t* stats, double* times) {
    return guarded([&] {
        GraphStats st; StageTimes tm;
        std::string g = compress_sequences(h->seqs, h->assembly_count, k, &st, &tm);
        if (gfa) *gfa = dup_str(g);
        if (stats) { stats[0] = st.kmers; stats[1] = st.unitigs_pre; stats[2] = st.links_pre; stats[3] = st.length_pre;
                     stats[4] = st.unitigs_post; stats[5] = st.links_post; stats[6] = st.length_post; }
        if (times) { times[0] = tm.load; times[1] = tm.repair; times[2] = tm.kmer_graph; times[3] = tm.unitig_graph;
                     times[4] = tm.simplify; times[5] = tm.save; }
    });
}
// compress.rs:32-50
int orc_compress_dir(const char* in_dir, const char* out_dir, uint32_t k, uint32_t max_contigs, int threads,
                     uint64_t* stats, double* times) {
    return guarded([&] {
        GraphStats st; StageTimes tm;
        compress_dir(in_dir, out_dir, k, max_contigs, threads, &st, &tm);
        if (stats) { stats[0] = st.kmers; stats[1] = st.unitigs_pre; stats[2] = st.links_pre; stats[3] = st.length_pre;
                     stats[4] = st.unitigs_post; stats[5] = st.links_post; stats[6] = st.length_post; }
        if (times) { times[0] = tm.load; times[1] = tm.repair; times[2] = tm.kmer_graph; times[3] = tm.unitig_graph;
                     times[4] = tm.simplify; times[5] = tm.save; }
    });
}
int orc_metrics_yaml(const orc_seqs* h, uint32_t unitig_count, uint64_t unitig_total_length, char** yaml) {
    return guarded([&] {
        InputAssemblyMetrics m = h->metrics;
        m.input_assemblies_count = (uint32_t)h->assembly_count;
        m.input_assemblies_total_contigs = (uint32_t)h->seqs.size();
        uint64_t t = 0; for (auto& s : h->seqs) t += s.length;
        m.input_assemblies_total_length = t;
        m.compressed_unitig_count = unitig_count;
        m.compressed_unitig_total_length = unitig_total_length;
        *yaml = dup_str(m.to_yaml());
    });
}

// ---- known-answer-test hooks (each mirrors one reference unit test's set-up) --------------------

// kmer_graph.rs:189-212,266-282: sorted Kmer displays, one per line, for one sequence id=1.
int orc_kat_kmers(const char* seq, uint32_t k, char** out) {
    return guarded([&] {
        std::string s(seq);
        Sequence sq = Sequence::new_with_seq(1, s, "assembly.fasta", "contig_1", s.size(), k / 2);
        KmerGraph kg(k);
        kg.add_sequence(sq, 1);
        std::string o;
        for (auto* km : kg.iterate_kmers()) { o += km->to_string(); o += "\n"; }
        *out = dup_str(o);
    });
}
// kmer_graph.rs:214-263
int orc_kat_neighbours(const char* seq, uint32_t k, const char* kmer, int next, char** out) {
    return guarded([&] {
        std::string s(seq);
        Sequence sq = Sequence::new_with_seq(1, s, "assembly.fasta", "contig_1", s.size(), k / 2);
        KmerGraph kg(k);
        kg.add_sequence(sq, 1);
        auto v = next ? kg.next_kmers(kmer) : kg.prev_kmers(kmer);
        std::string o;
        for (size_t i = 0; i < v.size(); i++) { if (i) o += ","; o += std::string(v[i]->seq()); }
        *out = dup_str(o);
    });
}
// position.rs:64-72
int orc_kat_position(uint16_t id, int strand, uint64_t pos, char** out) {
    return guarded([&] { *out = dup_str(Position::make(id, strand != 0, pos).to_string()); });
}
// kmer_graph.rs:189-197
int orc_kat_kmer_display(char** out) {
    return guarded([&] {
        std::string seq = "ACGACTGACATCAGCACTGA";
        Kmer k{seq.data(), 4, {}};
        k.positions.push_back(Position::make(1, true, 123));
        k.positions.push_back(Position::make(2, false, 456));
        *out = dup_str(k.to_string());
    });
}
// unitig.rs:411-441
int orc_kat_unitig_from_kmers(char** out) {
    return guarded([&] {
        uint32_t k = 5;
        Sequence seq = Sequence::new_with_seq(1, "ACGCATAGCACTAGCTACGA", "assembly.fasta", "contig_1", 20, k / 2);
        const char* f = seq.forward_seq.data(); const char* r = seq.reverse_seq.data();
        Kmer fk1{f + 4, 5, {}}, rk1{r + 15, 5, {}}, fk2{f + 5, 5, {}}, rk2{r + 14, 5, {}}, fk3{f + 6, 5, {}}, rk3{r + 13, 5, {}};
        for (Kmer* km : {&fk1, &rk1, &fk2, &rk2, &fk3, &rk3}) km->positions.push_back(Position::make(1, true, 1));
        Unitig u = Unitig::from_kmers(123, &fk2, &rk2);
        u.add_kmer_to_start(&fk1, &rk1);
        u.add_kmer_to_end(&fk3, &rk3);
        u.simplify_seqs();
        std::string o = std::to_string(u.length()) + "," + u.forward_seq + "," + u.reverse_seq;
        u.trim_overlaps(k);
        o += "," + std::to_string(u.length()) + "," + u.forward_seq + "," + u.reverse_seq;
        *out = dup_str(o);
    });
}
// unitig.rs:458-556: op in {remove_start, remove_end, add_start, add_end}
int orc_kat_shift(const char* op, const char* arg, char** out) {
    return guarded([&] {
        Unitig u = Unitig::from_segment_line("S\t1\tGCTGAAGGGC\tDP:f:1");
        u.forward_positions.push_back(Position::make(1, true, 100));
        u.reverse_positions.push_back(Position::make(2, false, 890));
        u.forward_positions.push_back(Position::make(2, false, 200));
        u.reverse_positions.push_back(Position::make(2, true, 790));
        std::string o(op);
        if (o == "remove_start") u.remove_seq_from_start((size_t)atoi(arg));
        else if (o == "remove_end") u.remove_seq_from_end((size_t)atoi(arg));
        else if (o == "add_start") u.add_seq_to_start(arg);
        else if (o == "add_end") u.add_seq_to_end(arg);
        else throw std::logic_error("bad op");
        std::ostringstream ss;
        ss << u.forward_seq << "," << u.reverse_seq << "," << u.forward_positions[0].pos << "," << u.reverse_positions[0].pos
           << "," << u.forward_positions[1].pos << "," << u.reverse_positions[1].pos;
        *out = dup_str(ss.str());
    });
}
// compress.rs:281-344: newline-separated candidate matches.
int orc_find_best_match(const char* matches, char** out) {
    return guarded([&] { *out = dup_str(find_best_match(split_lines(matches))); });
}
// misc.rs:589-593
int orc_reverse_complement(const char* seq, char** out) {
    return guarded([&] { *out = dup_str(reverse_complement(seq)); });
}
// misc.rs:780-826: "name\theader\tseq\n" per record
int orc_load_fasta(const char* path, char** out) {
    return guarded([&] {
        std::string o;
        for (auto& [n, h, s] : load_fasta(path)) o += n + "\t" + h + "\t" + s + "\n";
        *out = dup_str(o);
    });
}
int orc_find_all_assemblies(const char* dir, char** out) {
    return guarded([&] {
        std::string o;
        for (auto& p : find_all_assemblies(dir)) o += p + "\n";
        *out = dup_str(o);
    });
}
// unitig_graph.rs:993-1043: "k unitigs total_length all_links one_way_links"
int orc_gfa_stats(const char* gfa, char** out) {
    return guarded([&] {
        auto [g, seqs] = UnitigGraph::from_gfa_lines(split_lines(gfa));
        g.check_links();
        auto lc = g.link_count();
        std::ostringstream ss;
        ss << g.k_size << " " << g.unitigs.size() << " " << g.total_length() << " " << lc.first << " " << lc.second;
        *out = dup_str(ss.str());
    });
}
// tests.rs:108-112: load -> save
int orc_gfa_resave(const char* gfa, char** out) {
    return guarded([&] {
        auto [g, seqs] = UnitigGraph::from_gfa_lines(split_lines(gfa));
        *out = dup_str(g.save_gfa_string(seqs));
    });
}
// graph_simplification.rs:627-671: load -> simplify_structure -> forward seqs in order, one per line
int orc_gfa_simplify(const char* gfa, char** out) {
    return guarded([&] {
        auto [g, seqs] = UnitigGraph::from_gfa_lines(split_lines(gfa));
        simplify_structure(g, seqs);
        std::string o;
        for (auto& u : g.unitigs) o += u->forward_seq + "\n";
        *out = dup_str(o);
    });
}
// graph_simplification.rs:582-625: per unitig (file order) "inputs|outputs", each sorted like the test helper
int orc_gfa_exclusive(const char* gfa, char** out) {
    return guarded([&] {
        auto [g, seqs] = UnitigGraph::from_gfa_lines(split_lines(gfa));
        auto fmt = [](std::vector<UnitigStrand> v) {
            std::sort(v.begin(), v.end(), [](const UnitigStrand& a, const UnitigStrand& b) {
                if (a.number() != b.number()) return a.number() < b.number();
                return a.strand < b.strand;
            });
            std::string s;
            for (size_t i = 0; i < v.size(); i++) { if (i) s += ","; s += std::to_string(v[i].number()) + (v[i].strand ? "+" : "-"); }
            return s;
        };
        std::string o;
        for (auto& u : g.unitigs) o += fmt(get_exclusive_inputs(u.get())) + "|" + fmt(get_exclusive_outputs(u.get())) + "\n";
        *out = dup_str(o);
    });
}
// graph_simplification.rs:540-580: segs = "SEQ+\nSEQ-\n..." (unitig sequence and strand)
int orc_common_seq(const char* segs, int start, char** out) {
    return guarded([&] {
        auto lines = split_lines(segs);
        std::vector<std::unique_ptr<Unitig>> us;
        std::vector<UnitigStrand> v;
        uint32_t n = 0;
        for (auto& l : lines) {
            if (l.empty()) continue;
            bool strand = l.back() == '+';
            us.push_back(std::make_unique<Unitig>(Unitig::from_segment_line("S\t" + std::to_string(++n) + "\t" + l.substr(0, l.size() - 1) + "\tDP:f:1")));
            v.push_back({us.back().get(), strand});
        }
        *out = dup_str(start ? get_common_start_seq(v) : get_common_end_seq(v));
    });
}
// graph_simplification.rs:673-684: numbers "1,2,1"
int orc_check_duplicates(const char* numbers, int* result) {
    return guarded([&] {
        std::vector<std::unique_ptr<Unitig>> us;
        std::vector<UnitigStrand> v;
        std::stringstream ss(numbers);
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            auto u = std::make_unique<Unitig>();
            u->number = (uint32_t)atoi(tok.c_str());
            v.push_back({u.get(), true});
            us.push_back(std::move(u));
        }
        *result = check_for_duplicates(v) ? 1 : 0;
    });
}
// decompress.rs:83-105 + unitig_graph.rs:362-388: "filename\theader\tsequence\n" per path
int orc_decompress(const char* gfa, char** out) {
    return guarded([&] {
        auto [g, seqs] = UnitigGraph::from_gfa_lines(split_lines(gfa));
        std::string o;
        for (auto& [f, h, s] : g.reconstruct_original_sequences(seqs)) o += f + "\t" + h + "\t" + s + "\n";
        *out = dup_str(o);
    });
}

// cluster.rs:132-157 pairwise_contig_distances: distance(a, b) = 1 - (summed length of the unitigs shared by the paths of a
// and b) / (summed length of the unitigs of a's path); unitig sets ignore strand and multiplicity.  out: S*S doubles,
// out[a*S + b], sequences in GFA P-line order.
int orc_pairwise_distances(const char* gfa, double* out, uint32_t* n_seqs) {
    return guarded([&] {
        auto [g, seqs] = UnitigGraph::from_gfa_lines(split_lines(gfa));
        std::map<uint32_t, uint32_t> unitig_lengths;
        for (auto& u : g.unitigs) unitig_lengths[u->number] = u->length();
        std::vector<std::set<uint32_t>> sets;
        for (auto& s : seqs) {
            std::set<uint32_t> st;
            for (auto& [number, strand] : g.get_unitig_path_for_sequence(s)) { (void)strand; st.insert(number); }
            sets.push_back(std::move(st));
        }
        size_t S = seqs.size();
        if (n_seqs) *n_seqs = (uint32_t)S;
        if (!out) return;
        for (size_t a = 0; a < S; a++) {
            uint32_t a_sum = 0;
            for (uint32_t u : sets[a]) a_sum += unitig_lengths[u];
            double a_len = (double)a_sum;
            for (size_t b = 0; b < S; b++) {
                double ab_len = 0;
                for (uint32_t u : sets[a]) if (sets[b].count(u)) ab_len += (double)unitig_lengths[u];
                out[a * S + b] = 1.0 - (ab_len / a_len);
            }
        }
    });
}

// position.rs:18-46 + unitig_graph.rs:151-174: the forward / reverse positions of every unitig as from_gfa_lines rebuilds them
// from the P lines.  One line per unitig: "number\tF id,strand,pos;...\tR id,strand,pos;..." in vector order.
int orc_gfa_positions(const char* gfa, char** out) {
    return guarded([&] {
        auto [g, seqs] = UnitigGraph::from_gfa_lines(split_lines(gfa));
        (void)seqs;
        std::ostringstream os;
        for (auto& u : g.unitigs) {
            os << u->number;
            for (int fwd = 1; fwd >= 0; fwd--) {
                os << (fwd ? "\tF" : "\tR");
                const auto& v = fwd ? u->forward_positions : u->reverse_positions;
                for (size_t i = 0; i < v.size(); i++) os << (i ? ";" : " ") << v[i].seq_id() << "," << (v[i].strand() ? 1 : 0) << "," << v[i].pos;
            }
            os << "\n";
        }
        *out = dup_str(os.str());
    });
}

}  // extern "C"
