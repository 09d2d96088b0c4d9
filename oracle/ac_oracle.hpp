// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++17) of the `autocycler compress` hot path of rrwick/Autocycler v0.7.0,
// written from the reference's behaviour (file:line citations are relative to
// /root/reference/src).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load or run anything under oracle/.  The product path (autocycler_amd/) never links or calls it.
//
// Parity pinning: the reference is Rust and there is no Rust toolchain in the build image, so the
// reference itself cannot be compiled or run here (oracle/_ref is not buildable).  The oracle is
// pinned against every known-answer test the reference's own test-suite holds for this path
// (tests/test_oracle_kats.py).  Exact `input_assemblies.gfa` bytes for
// whole inputs are pinned by NO reference test — for those, parity is "restatement
// + round-trip properties", and is stated as such in DESIGN.md.
#pragma once
#include <cstdint>
#include <deque>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <utility>
#include <vector>

namespace oracle {

// misc.rs:131-142 — user-facing errors.  The reference exits (or panics under cfg(test)); the
// oracle throws and the C boundary turns it into a status + message.
struct QuitError : std::runtime_error { using std::runtime_error::runtime_error; };
[[noreturn]] void quit_with_error(const std::string& text);

// misc.rs:358-376
std::string reverse_complement(std::string_view seq);

// position.rs:18-52
struct Position {
    uint32_t pos;
    uint16_t seq_id_and_strand;
    static Position make(uint16_t seq_id, bool strand, size_t pos);
    uint16_t seq_id() const { return seq_id_and_strand & 0x7FFF; }
    bool strand() const { return (seq_id_and_strand & 0x8000) != 0; }
    std::string to_string() const;
};

// sequence.rs:19-59
struct Sequence {
    uint16_t id = 0;
    std::string forward_seq, reverse_seq;
    std::string filename, contig_header;
    size_t length = 0;
    uint16_t cluster = 0;
    static Sequence new_with_seq(size_t id, std::string seq, std::string filename,
                                 std::string contig_header, size_t length, uint32_t half_k);
    static Sequence new_without_seq(uint16_t id, std::string filename, std::string contig_header,
                                    size_t length, uint16_t cluster);
    std::string contig_name() const;
    std::string contig_description() const;
    bool is_ignored() const;
};

// kmer_graph.rs:26-61
struct Kmer {
    const char* pointer;
    size_t length;
    std::vector<Position> positions;
    std::string_view seq() const { return std::string_view(pointer, length); }
    size_t depth() const { return positions.size(); }
    bool first_position() const;
    std::string to_string() const;
};

struct FxLikeHash { size_t operator()(std::string_view s) const noexcept; };

// kmer_graph.rs:73-181
struct KmerGraph {
    uint32_t k_size;
    std::unordered_map<std::string_view, Kmer, FxLikeHash> kmers;
    explicit KmerGraph(uint32_t k) : k_size(k) {}
    void add_sequences(const std::vector<Sequence>& seqs, size_t assembly_count);
    void add_sequence(const Sequence& seq, size_t assembly_count);
    std::vector<const Kmer*> next_kmers(std::string_view kmer) const;
    std::vector<const Kmer*> prev_kmers(std::string_view kmer) const;
    std::vector<const Kmer*> iterate_kmers() const;
    const Kmer* reverse(const Kmer* kmer) const;
};

struct Unitig;
// unitig.rs:323-373
struct UnitigStrand {
    Unitig* unitig;
    bool strand;
    uint32_t number() const;
    int32_t signed_number() const;
    uint32_t length() const;
    std::string get_seq() const;
};

// unitig.rs:30-249
struct Unitig {
    uint32_t number = 0;
    std::deque<const Kmer*> forward_kmers, reverse_kmers;
    std::string forward_seq, reverse_seq;
    double depth = 0.0;
    std::vector<Position> forward_positions, reverse_positions;
    std::vector<UnitigStrand> forward_next, forward_prev, reverse_next, reverse_prev;

    static Unitig from_kmers(uint32_t number, const Kmer* f, const Kmer* r);
    static Unitig from_segment_line(const std::string& line);
    void add_kmer_to_end(const Kmer* f, const Kmer* r);
    void add_kmer_to_start(const Kmer* f, const Kmer* r);
    void simplify_seqs();
    void trim_overlaps(size_t k_size);
    std::string gfa_segment_line() const;
    uint32_t length() const { return (uint32_t)forward_seq.size(); }
    std::string get_seq(bool strand) const { return strand ? forward_seq : reverse_seq; }
    void remove_seq_from_start(size_t amount);
    void remove_seq_from_end(size_t amount);
    void add_seq_to_start(const std::string& seq);
    void add_seq_to_end(const std::string& seq);
};

// unitig_graph.rs:28-516, 723-793
struct UnitigGraph {
    std::vector<std::unique_ptr<Unitig>> unitigs;
    uint32_t k_size = 0;
    std::unordered_map<uint32_t, Unitig*> unitig_index;

    static UnitigGraph from_kmer_graph(const KmerGraph& kg);
    static std::pair<UnitigGraph, std::vector<Sequence>> from_gfa_lines(const std::vector<std::string>& lines);
    void build_unitig_index();
    void renumber_unitigs();
    void check_links() const;
    bool link_exists(uint32_t a, bool as, uint32_t b, bool bs) const;
    bool link_exists_prev(uint32_t a, bool as, uint32_t b, bool bs) const;
    std::string save_gfa_string(const std::vector<Sequence>& seqs) const;
    std::vector<std::pair<uint32_t, bool>> get_unitig_path_for_sequence(const Sequence& seq) const;
    std::string get_sequence_from_path(const std::vector<std::pair<uint32_t, bool>>& path) const;
    std::vector<std::tuple<std::string, std::string, std::string>>
        reconstruct_original_sequences(const std::vector<Sequence>& seqs) const;
    uint64_t total_length() const;
    std::pair<size_t, size_t> link_count() const;

  private:
    void build_unitigs_from_kmer_graph(const KmerGraph& kg);
    void create_links();
    void build_links_from_gfa(const std::vector<std::string>& link_lines);
    std::vector<Sequence> build_paths_from_gfa(const std::vector<std::string>& path_lines);
    void add_positions_from_path(const std::vector<std::pair<uint32_t, bool>>& path, bool path_strand,
                                 uint16_t seq_id, uint32_t length);
    UnitigStrand find_starting_unitig(uint16_t seq_id) const;
    bool get_next_unitig(uint16_t seq_id, bool seq_strand, const Unitig* u, bool strand, uint32_t pos,
                         UnitigStrand* next, uint32_t* next_pos) const;
};

// graph_simplification.rs:26-312
void simplify_structure(UnitigGraph& graph, const std::vector<Sequence>& seqs);
size_t expand_repeats(UnitigGraph& graph, const std::vector<Sequence>& seqs);
std::vector<UnitigStrand> get_exclusive_inputs(const Unitig* u);
std::vector<UnitigStrand> get_exclusive_outputs(const Unitig* u);
std::string get_common_start_seq(const std::vector<UnitigStrand>& unitigs);
std::string get_common_end_seq(const std::vector<UnitigStrand>& unitigs);
bool check_for_duplicates(const std::vector<UnitigStrand>& unitigs);

// compress.rs:202-270
void sequence_end_repair(std::vector<Sequence>& sequences, uint32_t k_size, int threads);
std::string find_best_match(const std::vector<std::string>& matches);

// misc.rs:65-96, 145-195, 282-355
std::vector<std::string> find_all_assemblies(const std::string& dir);
std::vector<std::tuple<std::string, std::string, std::string>> load_fasta(const std::string& filename);

struct ContigDetails { std::string name, description; uint64_t length; };
struct AssemblyDetails { std::string filename; std::vector<ContigDetails> contigs; };
// metrics.rs:65-107
struct InputAssemblyMetrics {
    uint32_t input_assemblies_count = 0, input_assemblies_total_contigs = 0;
    uint64_t input_assemblies_total_length = 0;
    uint32_t compressed_unitig_count = 0;
    uint64_t compressed_unitig_total_length = 0;
    std::vector<AssemblyDetails> details;
    std::string to_yaml() const;
};

// compress.rs:98-133
std::pair<std::vector<Sequence>, size_t> load_sequences(const std::string& assemblies_dir, uint32_t k_size,
                                                        InputAssemblyMetrics& metrics, uint32_t max_contigs,
                                                        int threads);

struct StageTimes { double load = 0, repair = 0, kmer_graph = 0, unitig_graph = 0, simplify = 0, save = 0; };
struct GraphStats { uint64_t kmers = 0; uint64_t unitigs_pre = 0, links_pre = 0, length_pre = 0;
                    uint64_t unitigs_post = 0, links_post = 0, length_post = 0; };

// compress.rs:42-47 on in-memory sequences (the replaced region): returns the GFA text.
std::string compress_sequences(const std::vector<Sequence>& seqs, size_t assembly_count, uint32_t k_size,
                               GraphStats* stats, StageTimes* times);
// compress.rs:32-50 end to end on a directory.
void compress_dir(const std::string& assemblies_dir, const std::string& autocycler_dir, uint32_t k_size,
                  uint32_t max_contigs, int threads, GraphStats* stats, StageTimes* times);

}  // namespace oracle
