// Host-only microbenchmark of the GFA emission of the whole command (gfa_chunks + the parallel pwrite of capi.cpp's write_pieces) on a
// synthetic graph of config C's shape: 158 639 unitigs (5.7 M bases), 212 353 links, 171 paths with 10.6 M entries -> a 95 MB file.
//   g++ -O2 -std=c++17 -pthread -I autocycler_amd/csrc tools/microbench/gfa_write_bench.cpp autocycler_amd/csrc/gfa_writer.cpp -o /tmp/gfa_write_bench
//   /tmp/gfa_write_bench [threads] [dir]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <random>
#include <thread>
#include <unistd.h>

#include "gfa_writer.hpp"
using namespace ac;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
    const uint32_t U = 158639; const uint64_t NL = 212353, NP = 10630307; const size_t S = 171;
    std::mt19937_64 rng(7);
    std::vector<uint64_t> seq_begin(U); std::vector<double> depth(U); std::vector<uint32_t> seq_len(U);
    std::string bases; uint64_t off = 0;
    for (uint32_t i = 0; i < U; i++) { seq_begin[i] = off; seq_len[i] = 1 + (uint32_t)(rng() % 71); depth[i] = 1 + (double)(rng() % 96); off += seq_len[i]; }
    bases.resize(off); for (auto& c : bases) c = "ACGT"[rng() & 3];
    std::vector<Link> links(NL);
    for (auto& l : links) { l.a = 1 + (int32_t)(rng() % U); l.b = 1 + (int32_t)(rng() % U); if (rng() & 1) l.a = -l.a; if (rng() & 1) l.b = -l.b; }
    std::vector<int32_t> path(NP);
    for (auto& v : path) { int32_t u = 1 + (int32_t)(rng() % U); v = (rng() & 1) ? u : -u; }
    FinalGraph g; g.k = 51; g.n_unitigs = U; g.seq_block.p = (void*)bases.data(); g.seq_begin = seq_begin.data(); g.depth = depth.data(); g.seq_len = seq_len.data();
    g.links = links.data(); g.n_links = NL; g.path = path.data(); g.n_path = NP; g.post.total_length = off;
    std::vector<SeqMeta> seqs(S);
    g.path_off.assign(S + 1, 0);
    for (size_t s = 0; s < S; s++) { seqs[s] = SeqMeta{(uint16_t)(s + 1), 5000000u, "assembly_0001.fasta", "contig_1 length=5000000"}; g.path_off[s + 1] = NP * (s + 1) / S; }
    for (int rep = 0; rep < 5; rep++) {
        double t0 = now();
        std::vector<std::string> pieces = gfa_chunks(g, seqs, T);
        double t1 = now();
        const std::string tmp = dir + "/gfa_write_bench.tmp";
        int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        std::vector<uint64_t> at(pieces.size() + 1, 0);
        for (size_t i = 0; i < pieces.size(); i++) at[i + 1] = at[i] + pieces[i].size();
        std::atomic<size_t> next{0};
        auto worker = [&] { for (size_t i; (i = next.fetch_add(1)) < pieces.size();) { const char* p = pieces[i].data(); uint64_t left = pieces[i].size(), o = at[i];
            while (left) { ssize_t w = ::pwrite(fd, p, left, (off_t)o); if (w <= 0) return; p += w; left -= (uint64_t)w; o += (uint64_t)w; } } };
        std::vector<std::thread> pool; for (int i = 1; i < std::min(T, 16); i++) pool.emplace_back(worker); worker(); for (auto& t : pool) t.join();
        ::close(fd);
        double t2 = now();
        pieces.clear(); pieces.shrink_to_fit();
        double t3 = now();
        printf("threads %d: format %.1f ms, write %.1f ms, free %.1f ms, file %.1f MB\n", T, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, at.back() / 1e6);
        ::unlink(tmp.c_str());
    }
    return 0;
}
