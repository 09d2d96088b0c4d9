/* A plain C99 client of include/autocycler_hip.h — what a foreign-function binding (the Rust shim of INTEGRATION.md, cgo, JNI ...)
 * sees: no C++ types, no torch.  Reads padded sequences from stdin (one per line: "<id> <unpadded length> <padded forward bytes>"),
 * calls ac_compress_build, prints the statistics compress.rs:152,165,177 prints and the GFA text (file names / headers = f<i> / h<i>).
 *     client <k> <assembly_count> <device> < seqs.txt > out.gfa            (tests/test_host_side.py, tests/test_gpu_boundary.py) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "autocycler_hip.h"

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: client k assembly_count device\n"); return 2; }
    uint32_t k = (uint32_t)atoi(argv[1]), assemblies = (uint32_t)atoi(argv[2]);
    int device = atoi(argv[3]);
    size_t cap = 16, n = 0;
    ac_seq_view* views = (ac_seq_view*)malloc(cap * sizeof *views);
    char** names = (char**)malloc(cap * sizeof *names);
    char** heads = (char**)malloc(cap * sizeof *heads);
    char* line = NULL; size_t lcap = 0; ssize_t got;
    while ((got = getline(&line, &lcap, stdin)) > 0) {
        unsigned id, len; int off = 0;
        if (sscanf(line, "%u %u %n", &id, &len, &off) < 2) continue;
        size_t plen = (size_t)len + k - 1;
        if (n == cap) { cap *= 2; views = (ac_seq_view*)realloc(views, cap * sizeof *views); names = (char**)realloc(names, cap * sizeof *names); heads = (char**)realloc(heads, cap * sizeof *heads); }
        uint8_t* buf = (uint8_t*)malloc(plen);
        memcpy(buf, line + off, plen);
        views[n].fwd = buf; views[n].length = len; views[n].id = (uint16_t)id;
        names[n] = (char*)malloc(32); heads[n] = (char*)malloc(32);
        snprintf(names[n], 32, "f%zu", n); snprintf(heads[n], 32, "h%zu", n);
        n++;
    }
    ac_graph* g = NULL;
    if (ac_compress_build(k, assemblies, views, (uint32_t)n, device, &g) != 0) {
        fprintf(stderr, "\nError: %s\n", ac_last_error());      /* quit_with_error, misc.rs:131-137 */
        return 1;
    }
    ac_stats pre = ac_stats_pre(g), post = ac_stats_post(g);
    fprintf(stderr, "kmers %llu pre %u %llu %llu post %u %llu %llu\n", (unsigned long long)ac_kmer_count(g), pre.unitigs,
            (unsigned long long)pre.links_one_way, (unsigned long long)pre.total_length, post.unitigs,
            (unsigned long long)post.links_one_way, (unsigned long long)post.total_length);
    /* the accessors a shim would walk: every unitig, every link, every path */
    uint64_t total = 0;
    for (uint32_t i = 0; i < ac_unitig_count(g); i++) { const uint8_t* s; uint32_t len; double d; if (ac_unitig(g, i, &s, &len, &d)) return 1; total += len; }
    if (total != post.total_length) { fprintf(stderr, "total length mismatch\n"); return 1; }
    const ac_link* links; uint64_t n_links;
    if (ac_links(g, &links, &n_links)) return 1;
    for (uint32_t s = 0; s < n; s++) { const int32_t* p; uint32_t np; if (ac_path(g, s, &p, &np) || np == 0) return 1; }
    char* gfa; uint64_t gfa_len;
    if (ac_gfa_string(g, (const char* const*)names, (const char* const*)heads, &gfa, &gfa_len) != 0) { fprintf(stderr, "\nError: %s\n", ac_last_error()); return 1; }
    fwrite(gfa, 1, gfa_len, stdout);
    ac_string_free(gfa);
    ac_free(g);
    return 0;
}
