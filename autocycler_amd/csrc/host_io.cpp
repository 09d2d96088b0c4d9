// See host_io.hpp.
#include "host_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <exception>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <set>
#include <thread>
#include <unordered_map>

namespace fs = std::filesystem;

namespace ac {

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool has_suffix(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && memcmp(s.data() + s.size() - n, suf, n) == 0;
}

// misc.rs:65-96.  `a && b || c || d` in the reference makes any file whose stem ends in .fna/.fa qualify.
std::vector<std::string> find_all_assemblies(const std::string& dir) {
    std::error_code ec;
    fs::directory_iterator it(dir, ec);
    if (ec) throw UserError("unable to read directory " + dir + "\n" + ec.message());
    std::vector<std::string> found;
    for (const auto& entry : it) {
        const fs::path& p = entry.path();
        if (!fs::is_regular_file(p)) continue;
        std::string ext = p.extension().string(), stem = p.stem().string();
        bool ok = ext == ".fasta" || ext == ".fna" || ext == ".fa" || (ext == ".gz" && has_suffix(stem, ".fasta")) ||
                  has_suffix(stem, ".fna") || has_suffix(stem, ".fa");
        if (ok) found.push_back(p.string());
    }
    std::sort(found.begin(), found.end());
    if (found.empty()) throw UserError("no assemblies found in " + dir);
    return found;
}

static std::string slurp(const std::string& filename) {
    unsigned char magic[2] = {0, 0};
    {
        FILE* f = fopen(filename.c_str(), "rb");
        if (!f) throw UserError("unable to open " + filename);
        size_t n = fread(magic, 1, 2, f);
        fclose(f);
        if (n < 2) magic[0] = 0;
    }
    std::string data;
    if (magic[0] == 0x1f && magic[1] == 0x8b) {   // gzip, multi-member like flate2's MultiGzDecoder (misc.rs:323)
        gzFile g = gzopen(filename.c_str(), "rb");
        if (!g) throw UserError("unable to load " + filename);
        gzbuffer(g, 1 << 20);
        std::vector<char> buf(1 << 20);
        int r;
        while ((r = gzread(g, buf.data(), (unsigned)buf.size())) > 0) data.append(buf.data(), (size_t)r);
        gzclose(g);
        if (r < 0) throw UserError("unable to load " + filename);
    } else {
        std::ifstream in(filename, std::ios::binary | std::ios::ate);
        if (!in) throw UserError("unable to load " + filename);
        std::streamsize sz = in.tellg();
        in.seekg(0);
        data.resize((size_t)sz);
        in.read(&data[0], sz);
    }
    return data;
}

// misc.rs:145-195,282-355
std::vector<std::array<std::string, 3>> load_fasta(const std::string& filename) {
    std::error_code ec;
    auto sz = fs::file_size(filename, ec);
    if (!ec && sz == 0) throw UserError(filename + " is an empty file");
    std::string data = slurp(filename);
    std::vector<std::array<std::string, 3>> recs;
    bool have = false;
    std::string name, header, seq;
    auto finish = [&] {
        for (char& c : seq) if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        recs.push_back({name, header, std::move(seq)});
        seq.clear();
    };
    size_t pos = 0, n = data.size();
    while (pos < n) {
        const char* nl = (const char*)memchr(data.data() + pos, '\n', n - pos);
        size_t eol = nl ? (size_t)(nl - data.data()) : n;
        size_t end = eol;
        if (end > pos && data[end - 1] == '\r') end--;
        if (end > pos) {
            if (data[pos] == '>') {
                if (have) finish();
                header.assign(data, pos + 1, end - pos - 1);
                size_t a = 0;
                while (a < header.size() && isspace((unsigned char)header[a])) a++;
                size_t b = a;
                while (b < header.size() && !isspace((unsigned char)header[b])) b++;
                if (b == a) throw UserError(filename + " is not correctly formatted");
                name.assign(header, a, b - a);
                have = true;
            } else {
                if (!have) throw UserError(filename + " is not correctly formatted");
                seq.append(data, pos, end - pos);
            }
        }
        pos = eol + 1;
    }
    if (have) finish();
    if (recs.empty()) throw UserError(filename + " contains no sequences");
    std::set<std::string> names;
    for (auto& r : recs) {
        if (r[0].empty()) throw UserError(filename + " has an unnamed sequence");
        if (r[2].empty()) throw UserError(filename + " has an empty sequence");
    }
    for (auto& r : recs)
        if (!names.insert(r[0]).second) throw UserError(filename + " has a duplicate name: " + r[0]);
    return recs;
}

// sequence.rs:31-59
void pad_sequence(LoadedSeq* s, const std::string& seq, uint32_t k) {
    for (char c : seq)
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') throw UserError(s->filename + " contains non-ACGT characters");
    uint32_t h = k / 2;
    s->forward_seq.clear();
    s->forward_seq.reserve(seq.size() + 2 * h);
    s->forward_seq.append(h, '.');
    s->forward_seq.append(seq);
    s->forward_seq.append(h, '.');
    s->length = (uint32_t)seq.size();
}

// compress.rs:84-133
LoadResult load_sequences(const std::string& assemblies_dir, uint32_t k, uint32_t max_contigs, int threads) {
    LoadResult lr;
    double t0 = now_s();
    std::vector<std::string> assemblies = find_all_assemblies(assemblies_dir);
    // Files are read, checked and padded in parallel; ids are handed out afterwards in file-then-record order, and the first
    // failing file (in that order) is the one reported, exactly as a sequential reader would.
    struct PerFile { AssemblyDetails det; std::vector<LoadedSeq> seqs; std::vector<char> ignored; std::exception_ptr err; };
    std::vector<PerFile> files(assemblies.size());
    auto load_one = [&](size_t fi) {
        PerFile& pf = files[fi];
        try {
            const std::string& assembly = assemblies[fi];
            pf.det.filename = assembly;                              // full path (metrics.rs:85)
            std::string base = fs::path(assembly).filename().string();
            for (auto& rec : load_fasta(assembly)) {
                if (rec[2].size() < k) continue;                    // skipped silently, consumes no id
                LoadedSeq s;
                s.filename = base;
                {   // header whitespace runs -> single spaces (compress.rs:115)
                    const std::string& hd = rec[1];
                    size_t i = 0;
                    while (i < hd.size()) {
                        while (i < hd.size() && isspace((unsigned char)hd[i])) i++;
                        size_t j = i;
                        while (j < hd.size() && !isspace((unsigned char)hd[j])) j++;
                        if (j > i) { if (!s.contig_header.empty()) s.contig_header.push_back(' '); s.contig_header.append(hd, i, j - i); }
                        i = j;
                    }
                }
                pad_sequence(&s, rec[2], k);
                size_t sp = s.contig_header.find(' ');
                pf.det.contigs.push_back({s.contig_header.substr(0, sp), sp == std::string::npos ? "" : s.contig_header.substr(sp + 1), s.length});
                std::string lower = s.contig_header;
                for (char& c : lower) c = (char)tolower((unsigned char)c);
                pf.ignored.push_back(lower.find("autocycler_ignore") != std::string::npos);
                pf.seqs.push_back(std::move(s));
            }
        } catch (...) { pf.err = std::current_exception(); }
    };
    {
        int T = std::max(1, std::min<int>(threads, (int)assemblies.size()));
        std::atomic<size_t> next{0};
        auto worker = [&] { for (size_t fi; (fi = next.fetch_add(1)) < assemblies.size();) load_one(fi); };
        std::vector<std::thread> pool;
        for (int t = 1; t < T; t++) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
    }
    size_t seq_id = 0;
    for (PerFile& pf : files) {
        if (pf.err) std::rethrow_exception(pf.err);
        for (size_t i = 0; i < pf.seqs.size(); i++) {
            if (++seq_id > 32767) throw UserError("no more than 32767 input sequences are allowed");
            pf.seqs[i].id = (uint16_t)seq_id;
            if (!pf.ignored[i]) lr.seqs.push_back(std::move(pf.seqs[i]));
        }
        lr.details.push_back(std::move(pf.det));
    }
    lr.assembly_count = (uint32_t)assemblies.size();
    lr.total_contigs_seen = (uint32_t)seq_id;
    if (lr.seqs.empty()) throw UserError("no sequences found in input assemblies");
    double mean = (double)lr.seqs.size() / (double)assemblies.size();
    if (mean > (double)max_contigs) {
        char buf[32]; snprintf(buf, sizeof buf, "%.1f", mean);
        throw UserError(std::string("the mean number of contigs per input assembly (") + buf + ") exceeds the allowed threshold (" +
                        std::to_string(max_contigs) + "). Are your input assemblies fragmented or contaminated?");
    }
    lr.load_seconds = now_s() - t0;      // (sequence_end_repair, compress.rs:202-270, is a device kernel: neighbours.inc — the callers in capi.cpp run it)
    lr.repair_seconds = 0;
    return lr;
}

// serde_yaml 0.9 block style of InputAssemblyMetrics (metrics.rs:65-107).  No reference test pins these bytes.
static std::string yaml_str(const std::string& s) {
    bool plain = !s.empty() && !isspace((unsigned char)s.front()) && !isspace((unsigned char)s.back());
    static const char* reserved[] = {"true", "false", "null", "~", "True", "False", "Null", "TRUE", "FALSE", "NULL", "y", "n", "yes", "no", "on", "off", ".nan", ".inf", "-.inf"};
    for (auto r : reserved) if (s == r) plain = false;
    if (plain && strchr("-?:,[]{}#&*!|>'\"%@`", s.front()) && !((s.front() == '-' || s.front() == '?' || s.front() == ':') && s.size() > 1 && !isspace((unsigned char)s[1]))) plain = false;
    for (size_t i = 0; plain && i < s.size(); i++) {
        unsigned char c = (unsigned char)s[i];
        if (c < 0x20 || c == 0x7f) plain = false;
        if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) plain = false;
        if (c == '#' && i > 0 && s[i - 1] == ' ') plain = false;
    }
    if (plain) { char* e = nullptr; strtod(s.c_str(), &e); if (e && *e == 0) plain = false; }
    if (plain) return s;
    std::string q = "'";
    for (char c : s) { if (c == '\'') q += "''"; else q.push_back(c); }
    return q + "'";
}
std::string metrics_yaml(const LoadResult& lr, uint32_t unitig_count, uint64_t unitig_total_length) {
    uint64_t total = 0;
    for (auto& s : lr.seqs) total += s.length;
    std::string y;
    y += "input_assemblies_count: " + std::to_string(lr.assembly_count) + "\n";
    y += "input_assemblies_total_contigs: " + std::to_string(lr.seqs.size()) + "\n";
    y += "input_assemblies_total_length: " + std::to_string(total) + "\n";
    y += "compressed_unitig_count: " + std::to_string(unitig_count) + "\n";
    y += "compressed_unitig_total_length: " + std::to_string(unitig_total_length) + "\n";
    if (lr.details.empty()) return y + "input_assembly_details: []\n";
    y += "input_assembly_details:\n";
    for (auto& a : lr.details) {
        y += "- filename: " + yaml_str(a.filename) + "\n";
        if (a.contigs.empty()) { y += "  contigs: []\n"; continue; }
        y += "  contigs:\n";
        for (auto& c : a.contigs) {
            y += "  - name: " + yaml_str(c.name) + "\n";
            y += "    description: " + yaml_str(c.description) + "\n";
            y += "    length: " + std::to_string(c.length) + "\n";
        }
    }
    return y;
}

// compress.rs:53-62
void check_compress_settings(const std::string& assemblies_dir, const std::string& autocycler_dir, uint32_t k, int threads) {
    if (!fs::exists(assemblies_dir)) throw UserError("directory does not exist: " + assemblies_dir);
    if (!fs::is_directory(assemblies_dir)) throw UserError(assemblies_dir + " is not a directory");
    if (fs::exists(autocycler_dir) && !fs::is_directory(autocycler_dir)) throw UserError(autocycler_dir + " exists but is not a directory");
    if (k < 11) throw UserError("--kmer cannot be less than 11");
    if (k > 501) throw UserError("--kmer cannot be greater than 501");
    if (k % 2 == 0) throw UserError("--kmer must be odd");
    if (threads < 1) throw UserError("--threads cannot be less than 1");
    if (threads > 100) throw UserError("--threads cannot be greater than 100");
}

std::string format_duration(double seconds) {
    uint64_t us = (uint64_t)(seconds * 1e6);
    char buf[64];
    snprintf(buf, sizeof buf, "%llu:%02llu:%02llu.%06llu", (unsigned long long)(us / 1000000 / 3600),
             (unsigned long long)(us / 1000000 / 60 % 60), (unsigned long long)(us / 1000000 % 60), (unsigned long long)(us % 1000000));
    return buf;
}

}  // namespace ac
