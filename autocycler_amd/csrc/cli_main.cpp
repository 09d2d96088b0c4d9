// autocycler-compress — standalone driver with the flag surface of `autocycler compress` (main.rs:140-160):
//   -i/--assemblies_dir DIR  -a/--autocycler_dir DIR  [--kmer 51] [--max_contigs 25] [-t/--threads 8] [--device 0 | --devices 0,1,..]
// (--devices: one job over several GPUs of the node through ac_compress_build_multi; not a flag of the reference)
// Writes DIR/input_assemblies.gfa and DIR/input_assemblies.yaml (compress.rs:45-46) through the C ABI
// (ac_compress_dir: C++ loader + end repair, HIP graph build, host tail, buffered GFA writer).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

#include "../../include/autocycler_hip.h"

static void usage() {
    fprintf(stderr, "Usage: autocycler-compress --assemblies_dir <DIR> --autocycler_dir <DIR> [--kmer 51] [--max_contigs 25] [--threads 8] [--device 0 | --devices 0,1,..]\n");
}

// `autocycler decompress` (main.rs:163-175): -i/--in_gfa FILE  [-o/--out_dir DIR]  [-f/--out_file FASTA]
static int decompress_main(int argc, char** argv) {
    std::string in, dir, file;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "error: a value is required for '%s'\n", a.c_str()); exit(2); }
            return argv[++i];
        };
        if (a == "-i" || a == "--in_gfa") in = val();
        else if (a == "-o" || a == "--out_dir") dir = val();
        else if (a == "-f" || a == "--out_file") file = val();
        else { fprintf(stderr, "error: unexpected argument '%s'\nUsage: autocycler-compress decompress --in_gfa <GFA> [--out_dir <DIR>] [--out_file <FASTA>]\n", a.c_str()); return 2; }
    }
    fprintf(stderr, "\nStarting autocycler decompress\nSettings:\n  --in_gfa %s\n", in.c_str());
    if (!dir.empty()) fprintf(stderr, "  --out_dir %s\n", dir.c_str());
    if (!file.empty()) fprintf(stderr, "  --out_file %s\n", file.c_str());
    if (ac_decompress(in.c_str(), dir.empty() ? nullptr : dir.c_str(), file.empty() ? nullptr : file.c_str(), 8) != 0) {
        fprintf(stderr, "\nError: %s\n", ac_last_error());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "decompress") == 0) return decompress_main(argc, argv);
    std::string in, out;
    unsigned k = 51, max_contigs = 25;
    int threads = 8, device = 0;
    std::vector<int> devices;
    int i = 1;
    if (argc > 1 && strcmp(argv[1], "compress") == 0) i = 2;
    for (; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "error: a value is required for '%s'\n", a.c_str()); usage(); exit(2); }
            return argv[++i];
        };
        if (a == "-i" || a == "--assemblies_dir") in = val();
        else if (a == "-a" || a == "--autocycler_dir") out = val();
        else if (a == "--kmer") k = (unsigned)strtoul(val(), nullptr, 10);
        else if (a == "--max_contigs") max_contigs = (unsigned)strtoul(val(), nullptr, 10);
        else if (a == "-t" || a == "--threads") threads = atoi(val());
        else if (a == "--device") device = atoi(val());
        else if (a == "--devices") { const char* v = val(); for (const char* q = v; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; } }
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(); return 2; }
    }
    if (in.empty() || out.empty()) { usage(); return 2; }
    auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "\nStarting autocycler compress (%s)\n", ac_version());
    fprintf(stderr, "Settings:\n  --assemblies_dir %s\n  --autocycler_dir %s\n  --kmer %u\n  --threads %d\n\n", in.c_str(), out.c_str(), k, threads);
    ac_graph* g = nullptr;
    double times[4] = {0, 0, 0, 0};
    const int rc = devices.size() > 1 ? ac_compress_dir_multi(in.c_str(), out.c_str(), k, max_contigs, threads, devices.data(), (int)devices.size(), &g, times)
                                      : ac_compress_dir(in.c_str(), out.c_str(), k, max_contigs, threads, devices.empty() ? device : devices[0], &g, times);
    if (rc != 0) {
        fprintf(stderr, "\nError: %s\n", ac_last_error());    // quit_with_error, misc.rs:131-137
        return 1;
    }
    ac_stats pre = ac_stats_pre(g), post = ac_stats_post(g);
    fprintf(stderr, "Graph contains %llu k-mers\n\n", (unsigned long long)ac_kmer_count(g));
    fprintf(stderr, "%u unitig%s, %llu link%s\ntotal length: %llu bp\n\n", pre.unitigs, pre.unitigs == 1 ? "" : "s",
            (unsigned long long)pre.links_one_way, pre.links_one_way == 1 ? "" : "s", (unsigned long long)pre.total_length);
    fprintf(stderr, "%u unitig%s, %llu link%s\ntotal length: %llu bp\n\n", post.unitigs, post.unitigs == 1 ? "" : "s",
            (unsigned long long)post.links_one_way, post.links_one_way == 1 ? "" : "s", (unsigned long long)post.total_length);
    ac_free(g);
    double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "Compressed unitig graph: %s/input_assemblies.gfa\nInput assembly stats:    %s/input_assemblies.yaml\n", out.c_str(), out.c_str());
    fprintf(stderr, "Stage times: load %.3fs, end repair %.3fs, graph build (GPU hot path) %.3fs, write %.3fs\n", times[0], times[1], times[2], times[3]);
    unsigned long long us = (unsigned long long)(total * 1e6);
    fprintf(stderr, "Time to run: %llu:%02llu:%02llu.%06llu\n\n", us / 1000000 / 3600, us / 1000000 / 60 % 60, us / 1000000 % 60, us % 1000000);
    // the output files are written and closed: leave without tearing down the HIP runtime, the device arena and the pinned pools
    // (tens of milliseconds of a sub-second command)
    fflush(stdout); fflush(stderr);
    _exit(0);
}
