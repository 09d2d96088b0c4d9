// The lockstep emulation of the wavefront primitives (autocycler_amd/csrc/wave_rt.hpp) checked on its own: ballots, shuffles, a barrier
// over LDS, lane groups that diverge from one another, lanes that return early, and the error it must report instead of guessing.
#define AC_EMU
#define AC_EMU_DEFINE_CTX_SWITCH
#include "wave_rt.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace ac;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); fails++; } } while (0)
struct Out { unsigned long long bal, first; int sh, sx, su, sd, nb, gs; unsigned long long gb; bool all_even; };
static void kern(Out* out, int n_live) {
    const unsigned t = wv::tid(), lane = (unsigned)wv::lane();
    if ((int)t >= n_live) return;                       // the tail of the last wavefront returns: it takes no part in what follows
    Out o{};
    const int v = (int)t * 3 + 1;
    o.bal = wv::ballot((t % 3) == 0);
    o.first = (unsigned long long)wv::uniform((int)t + 7);
    o.sh = wv::shfl(v, 5);
    o.sx = wv::shfl_xor(v, 1);
    o.su = wv::shfl_up(v, 2);
    o.sd = wv::shfl_down(v, 3);
    o.all_even = wv::all((v & 1) == ((int)t * 3 + 1) % 2);
    AC_SHARED int sh[256];
    sh[t] = v;
    wv::block_sync();
    o.nb = sh[(t + 1) % (unsigned)n_live];
    const int g = (int)(lane & ~15u);
    if ((t / 16) % 2 == 0) {                            // every other 16-lane group goes through two more operations, the others do not
        o.gb = wv::grp_ballot<16>((t & 1) != 0, g);
        o.gs = wv::grp_shfl<16>(v, 3, g);
    }
    out[wv::bid() * 256 + t] = o;
}
int main() {
    const int n_live = 200;                             // 3 full wavefronts + 8 lanes of a fourth
    std::vector<Out> out(512);
    wv::launch_kernel(kern, 2, 256, out.data(), n_live);
    for (int b = 0; b < 2; b++)
        for (int t = 0; t < n_live; t++) {
            const Out& o = out[b * 256 + t];
            const int w0 = t & ~63, lane = t & 63, live = (w0 + 64 <= n_live) ? 64 : n_live - w0;
            unsigned long long bal = 0;
            for (int l = 0; l < live; l++) if (((w0 + l) % 3) == 0) bal |= 1ULL << l;
            CHECK(o.bal == bal);
            CHECK(o.first == (unsigned long long)(w0 + 7));
            // a source lane that has RETURNED reads as 0 (EXEC-disabled on the device: ds_bpermute gives 0), one beyond the wavefront as the own value
            CHECK(o.sh == (5 < live ? (w0 + 5) * 3 + 1 : 0));
            CHECK(o.sx == ((lane ^ 1) < live ? (w0 + (lane ^ 1)) * 3 + 1 : 0));
            CHECK(o.su == (lane >= 2 ? (t - 2) * 3 + 1 : t * 3 + 1));
            CHECK(o.sd == (lane + 3 < live ? (t + 3) * 3 + 1 : (lane + 3 < 64 ? 0 : t * 3 + 1)));
            CHECK(o.all_even);
            CHECK(o.nb == ((t + 1) % n_live) * 3 + 1);
            if ((t / 16) % 2 == 0) {
                const int g0 = t & ~15, glive = (g0 + 16 <= n_live) ? 16 : n_live - g0;
                unsigned long long gb = 0;
                for (int l = 0; l < glive; l++) if ((g0 + l) & 1) gb |= 1ULL << l;
                CHECK(o.gb == gb);
                CHECK(o.gs == (3 < glive ? (g0 + 3) * 3 + 1 : 0));
            }
        }
    bool caught = false;
    try { wv::launch_kernel(+[](int) { if (wv::tid() < 32) wv::ballot(true); else wv::block_sync(); }, 1, 64, 0); } catch (const std::exception& e) { caught = true; printf("reported: %s\n", e.what()); }
    CHECK(caught);
    // a lane group that does a group operation first rejoins its wavefront at the next wavefront-wide one
    std::vector<unsigned long long> joined(64);
    wv::launch_kernel(+[](unsigned long long* o) { unsigned long long g = 0; if (wv::tid() < 16) g = wv::grp_ballot<16>((wv::tid() & 3) == 0, 0); o[wv::tid()] = wv::ballot(wv::tid() >= 60) ^ g; }, 1, 64, joined.data());
    CHECK(joined[0] == (0xF000000000000000ULL ^ 0x1111ULL) && joined[40] == 0xF000000000000000ULL);
    // a kernel whose locals outgrow the fiber's stack is reported (a canary below every stack), not left to corrupt the heap
    caught = false;
    try {
        wv::launch_kernel(+[](int* sink) { volatile char big[(128 << 10) - 4096]; for (size_t i = 0; i < 512; i++) big[i] = (char)(i + 1);      /* exactly the usable stack: the frames above it push its low end into the guard */ sink[0] = big[1024] + (int)wv::ballot(true); }, 1, 64, (int*)joined.data());
    } catch (const std::exception& e) { caught = true; printf("reported: %s\n", e.what()); }
    CHECK(caught);
    // the workgroup order knob of the launcher (AC_EMU_ORDER): every workgroup still runs exactly once
    for (const char* ord : {"1", "2", "0"}) {
        setenv("AC_EMU_ORDER", ord, 1);
        std::vector<unsigned long long> seen(37);
        wv::launch_kernel(+[](unsigned long long* o) { if (wv::tid() == 0) o[wv::bid()] += 1 + wv::bid(); }, 37, 64, seen.data());
        for (unsigned b = 0; b < 37; b++) CHECK(seen[b] == 1 + b);
    }
    // the runtime is usable after an error
    wv::launch_kernel(kern, 1, 256, out.data(), 256);
    CHECK(out[255].sh == 5 * 3 + 1 + 192 * 3);
    printf(fails ? "wave_rt_check: %d FAILED\n" : "wave_rt_check: OK\n", fails);
    return fails ? 1 : 0;
}
