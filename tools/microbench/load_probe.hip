// How long does the runtime take to load a code object (hipModuleLoadData), and what of it is the file's size (symbol and string
// tables a device never sees) as opposed to its kernels?  Round 4: a fresh `autocycler-compress` process waits 65-105 ms for the main
// code object.   hipcc -O2 --offload-arch=gfx950 tools/microbench/load_probe.hip -o /tmp/load_probe && /tmp/load_probe a.co b.co ...
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (hipSetDevice(0) != hipSuccess) return 1;
    void* p; if (hipMalloc(&p, 4096) != hipSuccess) return 1;
    for (int rep = 0; rep < 3; rep++)
        for (int i = 1; i < argc; i++) {
            std::ifstream f(argv[i], std::ios::binary);
            std::vector<char> img((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            const double t0 = now();
            hipModule_t m;
            hipError_t e = hipModuleLoadData(&m, img.data());
            const double t1 = now();
            if (e != hipSuccess) { printf("{\"file\": \"%s\", \"error\": \"%s\"}\n", argv[i], hipGetErrorString(e)); continue; }
            (void)hipModuleUnload(m);
            printf("{\"file\": \"%s\", \"bytes\": %zu, \"rep\": %d, \"load_ms\": %.2f, \"unload_ms\": %.2f}\n", argv[i], img.size(), rep, (t1 - t0) * 1e3, (now() - t1) * 1e3);
        }
    return 0;
}
