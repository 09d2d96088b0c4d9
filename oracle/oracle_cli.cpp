// ORACLE — TEST INFRASTRUCTURE ONLY.  `autocycler_oracle compress -i DIR -a DIR [--kmer K]
// [--max_contigs N] [-t T]` — the reference's compress flag surface (main.rs:140-160) over the CPU
// restatement; prints the stage times the reference's sections correspond to.
#include <cstdio>
#include <cstring>
#include <string>
#include "ac_oracle.hpp"
int main(int argc, char** argv) {
    std::string in, out; unsigned k = 51, max_contigs = 25; int threads = 8;
    int i = 1;
    if (argc > 1 && strcmp(argv[1], "compress") == 0) i = 2;
    for (; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "-i" || a == "--assemblies_dir") in = val();
        else if (a == "-a" || a == "--autocycler_dir") out = val();
        else if (a == "--kmer") k = (unsigned)atoi(val());
        else if (a == "--max_contigs") max_contigs = (unsigned)atoi(val());
        else if (a == "-t" || a == "--threads") threads = atoi(val());
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (in.empty() || out.empty()) { fprintf(stderr, "usage: autocycler_oracle compress -i DIR -a DIR [--kmer K] [--max_contigs N] [-t T]\n"); return 2; }
    try {
        oracle::GraphStats st; oracle::StageTimes tm;
        oracle::compress_dir(in, out, k, max_contigs, threads, &st, &tm);
        fprintf(stderr, "Graph contains %llu k-mers\n", (unsigned long long)st.kmers);
        fprintf(stderr, "%llu unitigs, %llu links\ntotal length: %llu bp\n", (unsigned long long)st.unitigs_pre, (unsigned long long)st.links_pre, (unsigned long long)st.length_pre);
        fprintf(stderr, "%llu unitigs, %llu links\ntotal length: %llu bp\n", (unsigned long long)st.unitigs_post, (unsigned long long)st.links_post, (unsigned long long)st.length_post);
        fprintf(stderr, "times: load+repair %.3f kmer_graph %.3f unitig_graph %.3f simplify %.3f save %.3f\n", tm.load, tm.kmer_graph, tm.unitig_graph, tm.simplify, tm.save);
    } catch (const oracle::QuitError& e) { fprintf(stderr, "\nError: %s\n", e.what()); return 1; }
    return 0;
}
