// Host side of `autocycler compress` around the device graph build: the rows SURVEY.md §8 marks "boundary"
// and "next" — assembly discovery + FASTA load (misc.rs:65-96,145-195,282-355), load_sequences
// (compress.rs:98-133) up to the padded sequences, the YAML metrics (metrics.rs:65-107,256-260) and the compress driver
// (compress.rs:32-50).  sequence_end_repair (compress.rs:202-270) is NOT here: it is a device kernel (neighbours.inc) and nothing else.
#pragma once
#include <cstdint>
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

namespace ac {

struct UserError : std::runtime_error { using std::runtime_error::runtime_error; };   // -> quit_with_error

struct LoadedSeq {
    uint16_t id = 0;
    std::string forward_seq;      // padded with k/2 dots each side, then end-repaired
    std::string filename;
    std::string contig_header;
    uint32_t length = 0;
};
struct ContigDetails { std::string name, description; uint64_t length; };
struct AssemblyDetails { std::string filename; std::vector<ContigDetails> contigs; };
struct LoadResult {
    std::vector<LoadedSeq> seqs;
    uint32_t assembly_count = 0;
    uint32_t total_contigs_seen = 0;      // ids handed out (ignored contigs consume one, compress.rs:111,120)
    std::vector<AssemblyDetails> details;
    double load_seconds = 0, repair_seconds = 0;
};

std::vector<std::string> find_all_assemblies(const std::string& dir);
std::vector<std::array<std::string, 3>> load_fasta(const std::string& filename);   // (name, header, sequence)
void pad_sequence(LoadedSeq* s, const std::string& seq, uint32_t k);               // sequence.rs:31-59
// the padded sequences as Sequence::new_with_seq leaves them (the device end repair then works on the text)
LoadResult load_sequences(const std::string& assemblies_dir, uint32_t k, uint32_t max_contigs, int threads);
std::string metrics_yaml(const LoadResult& lr, uint32_t unitig_count, uint64_t unitig_total_length);
void check_compress_settings(const std::string& assemblies_dir, const std::string& autocycler_dir, uint32_t k, int threads);
std::string format_duration(double seconds);   // misc.rs:379-385

}  // namespace ac
