// Wavefront / workgroup primitives behind one set of names, so that a kernel that uses ballots, shuffles, LDS and barriers has ONE
// source: on the device the names are the gfx950 intrinsics (forced inline: the same code as writing them out); under AC_EMU — the CPU
// build the test-suite loads — they are a LOCKSTEP EMULATION: every thread of a workgroup is a fiber (its own stack; a context switch of a dozen instructions), a fiber runs until it
// reaches a cross-lane operation, and the operation completes when all the live lanes of its wavefront (or of its lane group, or of its
// workgroup, for a barrier) have arrived.  Round 4: before this the CPU suite ran hand-written serial twins of the wave kernels
// (InsertWaveEmuFunctor, ExpandFunctor, ...), i.e. not the code that ships.
//
// What the emulation asks of a kernel (and checks): a wavefront-wide operation is reached by all live lanes of the wavefront (lanes that
// have returned no longer take part, as on the device); kernels whose lane GROUPS diverge from one another use the grp_* forms, which
// only involve the G lanes of the caller's group.  A lane that waits at one kind of operation while a lane of the same wavefront waits
// at another is reported as an error, not guessed at.
#pragma once
#include <cstdint>

#ifndef AC_EMU
#include <hip/hip_runtime.h>
#define AC_KERNEL __global__
#define AC_SHARED __shared__
#define AC_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
namespace ac { namespace wv {
__device__ __forceinline__ unsigned tid() { return threadIdx.x; }
__device__ __forceinline__ unsigned bid() { return blockIdx.x; }
__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
// the value of the wavefront's first live lane, in every lane (v_readfirstlane: makes a polled flag wave-uniform by construction)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, src); }
__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int shfl_xor(int v, int m) { return __shfl_xor(v, m); }
__device__ __forceinline__ unsigned long long shfl_xor64(unsigned long long v, int m) { return __shfl_xor(v, m); }
__device__ __forceinline__ bool all(bool p) { return __all(p) != 0; }
__device__ __forceinline__ int shfl_up(int v, int d) { return __shfl_up(v, d); }
__device__ __forceinline__ int shfl_down(int v, int d) { return __shfl_down(v, d); }
// the G lanes of the caller's group only (gshift = the group's first lane within the wavefront)
template <int G> __device__ __forceinline__ unsigned long long grp_ballot(bool p, int gshift) {
    return (__ballot(p) >> gshift) & (G == 64 ? ~0ULL : ((1ULL << G) - 1));
}
template <int G> __device__ __forceinline__ int grp_shfl(int v, int src_in_group, int gshift) { return __shfl(v, gshift + src_in_group); }
__device__ __forceinline__ void block_sync() { __syncthreads(); }
} }
#else
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#define AC_KERNEL
#define AC_SHARED static thread_local
#define AC_WAVES_PER_EU(lo, hi)
#define __launch_bounds__(n)
namespace ac { namespace wv {
enum { OP_NONE = 0, OP_BALLOT, OP_SHFL, OP_SHFL_XOR, OP_SHFL_UP, OP_SHFL_DOWN, OP_SYNC, OP_FIRST };
// A fiber switch without system calls (glibc's swapcontext saves the signal mask: a syscall per switch, and a kernel makes several
// switches per text position): the callee-saved registers and the stack pointer, x86-64 System V.  Other hosts (or -DAC_EMU_UCONTEXT, which
// the suite uses to test this path on x86-64 too) fall back to <ucontext.h>: slower, portable.  Test infrastructure only.
#if defined(__x86_64__) && !defined(AC_EMU_UCONTEXT)
extern "C" void ac_emu_ctx_switch(void** save_sp, void* load_sp);
#ifdef AC_EMU_DEFINE_CTX_SWITCH
asm(R"(
.text
.globl ac_emu_ctx_switch
.type ac_emu_ctx_switch,@function
ac_emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size ac_emu_ctx_switch,.-ac_emu_ctx_switch
)");
#endif
typedef void* ctx_t;
inline void ctx_switch(ctx_t* save, ctx_t* load) { ac_emu_ctx_switch(save, *load); }
// a fresh stack whose first switch-in "returns" into `entry`: six zero registers below the entry address, which sits at a 16-byte
// boundary (so that `entry` starts with the stack alignment a call would have left)
inline void ctx_make(ctx_t* c, char* lo, size_t bytes, void (*entry)()) {
    char* top = (char*)((uintptr_t)(lo + bytes) & ~(uintptr_t)15);
    void** sp = (void**)top;
    *--sp = nullptr;                        // (where a return address of entry's caller would be: never used)
    *--sp = (void*)entry;
    for (int r = 0; r < 6; r++) *--sp = nullptr;
    *c = sp;
}
#else
} }
#include <ucontext.h>
namespace ac { namespace wv {
typedef ucontext_t ctx_t;
inline void ctx_switch(ctx_t* save, ctx_t* load) { swapcontext(save, load); }
inline void ctx_make(ctx_t* c, char* lo, size_t bytes, void (*entry)()) {
    getcontext(c);
    c->uc_stack.ss_sp = lo; c->uc_stack.ss_size = bytes; c->uc_link = nullptr;
    makecontext(c, entry, 0);
}
#endif
struct Lane {
    ctx_t ctx{};
    bool done = true;
    int op = OP_NONE, scope = 0;      // scope: 64 = wavefront, G < 64 = lane group, 0 = workgroup (barrier)
    unsigned long long arg = 0, res = 0;
    int src = 0;
};
struct Block {
    std::vector<Lane> lanes;
    char* stacks = nullptr; size_t stacks_bytes = 0;      // (malloc: never zeroed — 32 MB per emulating thread)
    ~Block() { free(stacks); }
    ctx_t sched{};
    unsigned n = 0, cur = 0, block_idx = 0;
    const std::function<void()>* body = nullptr;
    std::string error;
};
inline Block*& tl_block() { static thread_local Block* b = nullptr; return b; }
inline Block& blk() { return *tl_block(); }
inline unsigned tid() { return blk().cur; }
inline unsigned bid() { return blk().block_idx; }
inline int lane() { return (int)(blk().cur & 63); }
static const size_t STACK_BYTES = 128 << 10;
static const size_t STACK_GUARD = 4096;
static const unsigned long long STACK_CANARY = 0xAC57AC4B0F1BE255ULL;
inline void lane_entry() {
    Block& b = blk();
    try { (*b.body)(); } catch (const std::exception& e) { if (b.error.empty()) b.error = e.what(); } catch (...) { if (b.error.empty()) b.error = "unknown exception in a kernel"; }
    b.lanes[b.cur].done = true;
    ctx_switch(&b.lanes[b.cur].ctx, &b.sched);      // never comes back
    abort();
}
inline unsigned long long wait_op(int op, int scope, unsigned long long arg, int src) {
    Block& b = blk();
    Lane& l = b.lanes[b.cur];
    l.op = op; l.scope = scope; l.arg = arg; l.src = src;
    ctx_switch(&l.ctx, &b.sched);
    return l.res;
}
// Completes every operation whose participants have all arrived.  Returns whether anything was released.
inline bool resolve(Block& b) {
    bool any = false;
    // barriers: all live lanes of the workgroup
    {
        bool all = true, some = false;
        for (unsigned i = 0; i < b.n; i++) if (!b.lanes[i].done) { some = true; if (b.lanes[i].op != OP_SYNC) all = false; }
        if (some && all) { for (unsigned i = 0; i < b.n; i++) if (!b.lanes[i].done) b.lanes[i].op = OP_NONE; return true; }
    }
    for (unsigned w0 = 0; w0 < b.n; w0 += 64) {
        const unsigned w1 = w0 + 64 < b.n ? w0 + 64 : b.n;
        // the scope the waiting lanes of this wavefront ask for (they must agree lane group by lane group)
        for (unsigned g0 = w0; g0 < w1;) {
            unsigned first = g0;
            while (first < w1 && (b.lanes[first].done || b.lanes[first].op == OP_NONE || b.lanes[first].op == OP_SYNC)) first++;
            if (first >= w1) break;
            const int scope = b.lanes[first].scope, op = b.lanes[first].op;
            const unsigned s0 = scope == 64 ? w0 : w0 + ((first - w0) / (unsigned)scope) * (unsigned)scope;
            const unsigned s1 = scope == 64 ? w1 : (s0 + (unsigned)scope < w1 ? s0 + (unsigned)scope : w1);
            bool ready = true;
            for (unsigned i = s0; i < s1; i++) {
                const Lane& l = b.lanes[i];
                if (l.done) continue;
                if (l.op == OP_NONE) { ready = false; continue; }      // still running towards it (cannot happen: resolve runs when all are blocked)
                if (l.op == OP_SYNC) { ready = false; continue; }      // at a barrier while a neighbour is at a wave operation: wait for the barrier's turn
                if (l.op != op || l.scope != scope) {
                    if (b.error.empty()) b.error = "lockstep emulation: lanes of one wavefront wait at different cross-lane operations (a kernel whose lane groups diverge must use the grp_* forms)";
                    ready = false;
                }
            }
            if (ready) {
                unsigned long long bal = 0;
                if (op == OP_BALLOT) for (unsigned i = s0; i < s1; i++) if (!b.lanes[i].done && b.lanes[i].arg) bal |= 1ULL << (i - s0);
                for (unsigned i = s0; i < s1; i++) {
                    Lane& l = b.lanes[i];
                    if (l.done) continue;
                    if (op == OP_BALLOT) l.res = bal;
                    else if (op == OP_FIRST) { for (unsigned q = s0; q < s1; q++) if (!b.lanes[q].done) { l.res = b.lanes[q].arg; break; } }
                    else {
                        int src = op == OP_SHFL ? l.src : (op == OP_SHFL_XOR ? (int)((i - s0) ^ (unsigned)l.src) : (op == OP_SHFL_UP ? (int)(i - s0) - l.src : (int)(i - s0) + l.src));
                        const int width = (int)(s1 - s0);
                        if (op == OP_SHFL) src = ((src % width) + width) % width;
                        // out of range: the own value, as the device's shuffles do; a lane that has RETURNED is EXEC-disabled on the device and
                        // ds_bpermute reads 0 from it — the emulation answers 0 too, so that a kernel relying on such a read fails here as there
                        const bool in_range = src >= 0 && src < width;
                        l.res = in_range ? (b.lanes[s0 + (unsigned)src].done ? 0ULL : b.lanes[s0 + (unsigned)src].arg) : l.arg;
                    }
                }
                for (unsigned i = s0; i < s1; i++) if (!b.lanes[i].done) b.lanes[i].op = OP_NONE;
                any = true;
            }
            g0 = s1;
        }
    }
    return any;
}
// One workgroup of `threads` lanes running `body` (which reads tid() / bid()).
inline void run_block(unsigned block_idx, unsigned threads, const std::function<void()>& body) {
    static thread_local Block b;
    Block* prev = tl_block();
    tl_block() = &b;
    b.n = threads; b.block_idx = block_idx; b.body = &body; b.error.clear();
    if (b.lanes.size() < threads) b.lanes.resize(threads);
    if (b.stacks_bytes < (size_t)threads * STACK_BYTES) { free(b.stacks); b.stacks_bytes = (size_t)threads * STACK_BYTES; b.stacks = (char*)malloc(b.stacks_bytes); if (!b.stacks) throw std::runtime_error("out of memory for the emulation's fiber stacks"); }
    for (unsigned i = 0; i < threads; i++) {
        Lane& l = b.lanes[i];
        l.done = false; l.op = OP_NONE;
        char* lo = b.stacks + (size_t)i * STACK_BYTES;
        // the lowest STACK_GUARD bytes are not handed out: eight canary words right below the usable part (checked when the workgroup is
        // done: a kernel whose locals outgrow the fiber's stack), the rest slack that keeps a small overrun off the neighbour's stack
        for (int q = 1; q <= 8; q++) ((unsigned long long*)(lo + STACK_GUARD))[-q] = STACK_CANARY;
        ctx_make(&l.ctx, lo + STACK_GUARD, STACK_BYTES - STACK_GUARD, &lane_entry);
    }
    for (;;) {
        bool alive = false, ran = false;
        for (unsigned i = 0; i < threads; i++) {
            Lane& l = b.lanes[i];
            if (l.done) continue;
            alive = true;
            if (l.op != OP_NONE) continue;
            b.cur = i;
            ctx_switch(&b.sched, &l.ctx);
            ran = true;
        }
        if (!alive) break;
        const bool released = resolve(b);
        if (!b.error.empty()) break;
        if (!ran && !released) { b.error = "lockstep emulation: deadlock (a cross-lane operation that not all of its lanes reach)"; break; }
    }
    tl_block() = prev;
    if (b.error.empty())
        for (unsigned i = 0; i < threads && b.error.empty(); i++)
            for (int q = 1; q <= 8; q++)
                if (((const unsigned long long*)(b.stacks + (size_t)i * STACK_BYTES + STACK_GUARD))[-q] != STACK_CANARY) { b.error = "lockstep emulation: a fiber overran its stack (kernel locals beyond " + std::to_string((STACK_BYTES - STACK_GUARD) >> 10) + " KB)"; break; }
    if (!b.error.empty()) {
        for (unsigned i = 0; i < threads; i++) b.lanes[i].done = true;
        throw std::runtime_error(b.error);
    }
}
inline unsigned long long ballot(bool p) { return wait_op(OP_BALLOT, 64, p ? 1 : 0, 0); }
inline int uniform(int v) { return (int)(unsigned)wait_op(OP_FIRST, 64, (unsigned)v, 0); }
inline int shfl(int v, int src) { return (int)(unsigned)wait_op(OP_SHFL, 64, (unsigned)v, src); }
inline unsigned long long shfl64(unsigned long long v, int src) { return wait_op(OP_SHFL, 64, v, src); }
inline int shfl_xor(int v, int m) { return (int)(unsigned)wait_op(OP_SHFL_XOR, 64, (unsigned)v, m); }
inline unsigned long long shfl_xor64(unsigned long long v, int m) { return wait_op(OP_SHFL_XOR, 64, v, m); }
inline bool all(bool p) { return wait_op(OP_BALLOT, 64, p ? 0 : 1, 0) == 0; }
inline int shfl_up(int v, int d) { return (int)(unsigned)wait_op(OP_SHFL_UP, 64, (unsigned)v, d); }
inline int shfl_down(int v, int d) { return (int)(unsigned)wait_op(OP_SHFL_DOWN, 64, (unsigned)v, d); }
template <int G> inline unsigned long long grp_ballot(bool p, int) { return wait_op(OP_BALLOT, G, p ? 1 : 0, 0); }
template <int G> inline int grp_shfl(int v, int src_in_group, int) { return (int)(unsigned)wait_op(OP_SHFL, G, (unsigned)v, src_in_group); }
inline void block_sync() { wait_op(OP_SYNC, 0, 0, 0); }
} }
inline long long clock64() { return 0; }      // (the profiling variants of a kernel are never launched by the emulation)
namespace ac { namespace wv {
// hipLaunchKernelGGL for the emulation: the workgroups one after the other
// AC_EMU_ORDER=1 / 2 (the order knob of the functor launcher, device_rt.hpp): the workgroups in descending / pseudo-random order, so that
// the tests can check that no result of a wave kernel depends on which workgroup runs first
template <class K, class... A> void launch_kernel(K kernel, unsigned blocks, unsigned threads, A... args) {
    const std::function<void()> body = [&] { kernel(args...); };
    const char* now = getenv("AC_EMU_ORDER");      // (read per launch: the tests switch it between builds of one process)
    const int ord = now ? atoi(now) : 0;
    if (ord == 1) { for (unsigned bi = blocks; bi-- > 0;) run_block(bi, threads, body); }
    else if (ord == 2) {
        std::vector<unsigned> perm(blocks);
        for (unsigned i = 0; i < blocks; i++) perm[i] = i;
        unsigned long long st = 0x9E3779B97F4A7C15ULL ^ blocks;
        for (unsigned i = blocks; i > 1; i--) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; std::swap(perm[i - 1], perm[(st >> 33) % i]); }
        for (unsigned i = 0; i < blocks; i++) run_block(perm[i], threads, body);
    } else { for (unsigned bi = 0; bi < blocks; bi++) run_block(bi, threads, body); }
}
} }
#endif
