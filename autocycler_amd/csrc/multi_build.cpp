// ac_compress_build_multi: one compress job over several devices of one node from ONE process (see multi_build.hpp).
//
// Rank r = one host thread + one device context + one GraphBuilder.  The phases are those of the sharded build (graph_build.hpp:
// shard_begin .. shard_finish); what used to be the caller's job — the collectives between them — happens here:
//
//   fragments + records     all-gather (variable sizes), straight into their places in the union text / record table
//   novel bitmaps           all-reduce SUM (uint64; the owners partition the keys: disjoint bits)
//   sibling bits            all-reduce SUM (uint64; 2 bits per distinct k-mer, by novel index)
//   degree bytes            all-reduce SUM (uint8), compact: what the owners' probes found for the 1-3 % of the k-mers the sibling bits do not settle
//   link words              all-reduce SUM (int32; the walk words are derived from them on arrival)
//   walk-start keys         ROUTED BY OWNER: every rank sorts its queries by owner = home hash mod N (queries_route) and one all-to-all
//   + their answers         sends each key to the one rank whose table can answer it; the answers come back by the reverse all-to-all —
//                           a rank receives ~1/N of the keys instead of all of them (north_star's bucket exchange)
//   per-unitig sums / mins  all-reduce SUM, MIN (int32)
//   paths                   none: every rank's copy of its own paths lands in host memory of this one process
//
// Transports: RCCL (dlopen'd librccl: ncclCommInitAll over the device list, ncclAllReduce, grouped ncclSend / ncclRecv on each rank's
// stream) when every rank has a device of its own; host-staged (device -> shared host buffer -> device, thread barriers) when ranks share a
// device or under the CPU emulation.  Small host values (sizes, counts) always go through shared memory: the ranks are threads.
#include "multi_build.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "device_rt.hpp"

#ifndef AC_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

namespace ac {
namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// A barrier for the rank threads that gives up when a rank has failed (nobody may wait for a thread that has thrown).
class RankBarrier {
  public:
    explicit RankBarrier(int n) : n_(n) {}
    void fail(const std::string& why) {
        std::lock_guard<std::mutex> lock(mu_);
        if (!failed_) { failed_ = true; why_ = why; }
        cv_.notify_all();
    }
    bool failed() { std::lock_guard<std::mutex> lock(mu_); return failed_; }
    std::string why() { std::lock_guard<std::mutex> lock(mu_); return why_; }
    void wait() {
        std::unique_lock<std::mutex> lock(mu_);
        if (failed_) throw DeviceError("another rank of the multi-device build failed: " + why_);
        const uint64_t gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); return; }
        cv_.wait(lock, [&] { return gen_ != gen || failed_; });
        if (gen_ == gen) throw DeviceError("another rank of the multi-device build failed: " + why_);
    }
  private:
    std::mutex mu_; std::condition_variable cv_;
    int n_, count_ = 0; uint64_t gen_ = 0; bool failed_ = false; std::string why_;
};

enum XType { X_U8, X_I32, X_I64, X_U64 };
enum XOp { X_SUM, X_MIN };
size_t xsize(XType t) { return t == X_U8 ? 1 : (t == X_I32 ? 4 : 8); }

// The device-buffer collectives of one multi-device build.  Every method is called by all ranks, each from its own thread.
class Exchange {
  public:
    Exchange(int n, RankBarrier* bar) : R(n), bar_(bar), values_(n) {}
    virtual ~Exchange() {}
    const int R;
    // small host values: all[r * n + i] = rank r's mine[i]
    void gather_values(int rank, const uint64_t* mine, int n, uint64_t* all) {
        values_[rank].assign(mine, mine + n);
        bar_->wait();
        for (int r = 0; r < R; r++) memcpy(all + (size_t)r * n, values_[r].data(), (size_t)n * 8);
        bar_->wait();
    }
    // d_recv + displ[r] receives rank r's bytes[r] bytes (every rank's send is its own slice: send_bytes == bytes[rank])
    virtual void all_gather_v(int rank, const void* d_send, void* d_recv, const size_t* bytes, const size_t* displ) = 0;
    virtual void all_reduce(int rank, void* d_buf, size_t count, XType t, XOp op) = 0;      // in place
    // send_displ / send_bytes: what goes to each rank out of d_send; recv_displ / recv_bytes: where each rank's part lands in d_recv
    virtual void all_to_all_v(int rank, const void* d_send, const size_t* send_bytes, const size_t* send_displ, void* d_recv,
                              const size_t* recv_bytes, const size_t* recv_displ) = 0;
    virtual const char* name() const = 0;
  protected:
    RankBarrier* bar_;
    std::vector<std::vector<uint64_t>> values_;
};

// Through host memory: rank r's buffer -> stage[r] (device -> host), a barrier, every rank picks up what is meant for it.
class HostStagedExchange : public Exchange {
  public:
    HostStagedExchange(int n, RankBarrier* bar) : Exchange(n, bar), stage_(n), sb_(n), sd_(n) {}
    const char* name() const override { return "host-staged"; }
    void all_gather_v(int rank, const void* d_send, void* d_recv, const size_t* bytes, const size_t* displ) override {
        stage_[rank].resize(bytes[rank]);
        copy_d2h(stage_[rank].data(), d_send, bytes[rank]);
        bar_->wait();
        for (int r = 0; r < R; r++) copy_h2d((uint8_t*)d_recv + displ[r], stage_[r].data(), bytes[r]);
        stream_sync();
        bar_->wait();
    }
    void all_reduce(int rank, void* d_buf, size_t count, XType t, XOp op) override {
        const size_t es = xsize(t);
        stage_[rank].resize(count * es);
        copy_d2h(stage_[rank].data(), d_buf, count * es);
        if (rank == 0) result_.resize(count * es);
        bar_->wait();
        const size_t a = count * (size_t)rank / (size_t)R, b = count * (size_t)(rank + 1) / (size_t)R;      // this rank reduces its slice
        auto run = [&](auto zero) {
            typedef decltype(zero) T;
            T* out = (T*)result_.data();
            for (size_t i = a; i < b; i++) {
                T acc = ((const T*)stage_[0].data())[i];
                for (int r = 1; r < R; r++) { const T v = ((const T*)stage_[r].data())[i]; acc = op == X_SUM ? (T)(acc + v) : (v < acc ? v : acc); }
                out[i] = acc;
            }
        };
        if (t == X_U8) run((uint8_t)0); else if (t == X_I32) run((int32_t)0); else if (t == X_I64) run((int64_t)0); else run((uint64_t)0);
        bar_->wait();
        copy_h2d(d_buf, result_.data(), count * es);
        stream_sync();
        bar_->wait();
    }
    void all_to_all_v(int rank, const void* d_send, const size_t* send_bytes, const size_t* send_displ, void* d_recv, const size_t* recv_bytes,
                      const size_t* recv_displ) override {
        size_t total = 0;
        for (int r = 0; r < R; r++) total = std::max(total, send_displ[r] + send_bytes[r]);
        stage_[rank].resize(total);
        copy_d2h(stage_[rank].data(), d_send, total);
        sb_[rank].assign(send_bytes, send_bytes + R); sd_[rank].assign(send_displ, send_displ + R);
        bar_->wait();
        for (int src = 0; src < R; src++) {
            if (sb_[src][rank] != recv_bytes[src]) throw DeviceError("internal error: all-to-all sizes disagree between ranks");
            copy_h2d((uint8_t*)d_recv + recv_displ[src], stage_[src].data() + sd_[src][rank], recv_bytes[src]);
        }
        stream_sync();
        bar_->wait();
    }
  private:
    std::vector<std::vector<uint8_t>> stage_;
    std::vector<std::vector<size_t>> sb_, sd_;
    std::vector<uint8_t> result_;
};

#ifndef AC_EMU
// RCCL, loaded on first use (a single-device user never loads it).  One communicator per rank, created together (ncclCommInitAll) and
// kept for the next build over the same device list.
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::vector<ncclComm_t> comms; std::vector<int> devices;
    static RcclApi& get() { static RcclApi* a = new RcclApi; return *a; }
    void load() {
        if (lib) return;
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) throw DeviceError(std::string("RCCL is not available (") + dlerror() + "): a multi-device build needs librccl, or the host-staged transport");
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) throw DeviceError(std::string("librccl lacks ") + n); return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString"); AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    }
    void check(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw DeviceError(std::string("RCCL error in ") + what + ": " + GetErrorString(r)); }
    void ensure_comms(const std::vector<int>& devs) {
        load();
        if (devs == devices && !comms.empty()) return;
        destroy_comms();
        comms.assign(devs.size(), nullptr);
        check(CommInitAll(comms.data(), (int)devs.size(), devs.data()), "ncclCommInitAll");
        devices = devs;
    }
    void destroy_comms() { for (auto c : comms) if (c) (void)CommDestroy(c); comms.clear(); devices.clear(); }
};
class RcclExchange : public Exchange {
  public:
    RcclExchange(int n, RankBarrier* bar) : Exchange(n, bar), api_(RcclApi::get()) {}
    const char* name() const override { return "rccl"; }
    // (a host barrier in front of every collective: a rank that has failed never leaves the others inside one)
    void all_gather_v(int rank, const void* d_send, void* d_recv, const size_t* bytes, const size_t* displ) override {
        bar_->wait();
        flush_fills();
        api_.check(api_.GroupStart(), "ncclGroupStart");
        for (int r = 0; r < R; r++) {
            if (bytes[rank]) api_.check(api_.Send(d_send, bytes[rank], ncclUint8, r, api_.comms[rank], 0), "ncclSend");
            if (bytes[r]) api_.check(api_.Recv((uint8_t*)d_recv + displ[r], bytes[r], ncclUint8, r, api_.comms[rank], 0), "ncclRecv");
        }
        api_.check(api_.GroupEnd(), "ncclGroupEnd");
    }
    void all_reduce(int rank, void* d_buf, size_t count, XType t, XOp op) override {
        bar_->wait();
        flush_fills();
        const ncclDataType_t dt = t == X_U8 ? ncclUint8 : (t == X_I32 ? ncclInt32 : (t == X_I64 ? ncclInt64 : ncclUint64));
        api_.check(api_.AllReduce(d_buf, d_buf, count, dt, op == X_SUM ? ncclSum : ncclMin, api_.comms[rank], 0), "ncclAllReduce");
    }
    void all_to_all_v(int rank, const void* d_send, const size_t* send_bytes, const size_t* send_displ, void* d_recv, const size_t* recv_bytes,
                      const size_t* recv_displ) override {
        bar_->wait();
        flush_fills();
        api_.check(api_.GroupStart(), "ncclGroupStart");
        for (int r = 0; r < R; r++) {
            if (send_bytes[r]) api_.check(api_.Send((const uint8_t*)d_send + send_displ[r], send_bytes[r], ncclUint8, r, api_.comms[rank], 0), "ncclSend");
            if (recv_bytes[r]) api_.check(api_.Recv((uint8_t*)d_recv + recv_displ[r], recv_bytes[r], ncclUint8, r, api_.comms[rank], 0), "ncclRecv");
        }
        api_.check(api_.GroupEnd(), "ncclGroupEnd");
    }
  private:
    RcclApi& api_;
};
#endif

// What a rank keeps between multi-device builds: its device context and the arena its exchange buffers come from.
struct RankCtx {
    DeviceCtx ctx;
    Arena xarena;      // exchange buffers (the builder owns the context's arena for the build)
};
std::vector<std::unique_ptr<RankCtx>>& rank_ctxs() { static auto* v = new std::vector<std::unique_ptr<RankCtx>>; return *v; }

struct Shared {
    uint32_t k; uint32_t assembly_count; int R;
    const std::vector<SeqView>* seqs;
    std::vector<size_t> first;               // rank r holds sequences [first[r], first[r + 1])
    std::vector<int> devices;
    Exchange* x; RankBarrier* bar;
    std::vector<FinalGraph> graphs;          // per rank: root's holds unitigs + links, every rank's its own paths
    std::vector<BuildTimings> tms;
    std::vector<double> xsec;
    MultiStats st; std::mutex st_mu;
};

void rank_main(Shared& S, int rank) {
    RankCtx& rc = *rank_ctxs()[rank];
    set_device_ctx(&rc.ctx);
    struct Unset { ~Unset() { set_device_ctx(nullptr); } } unset;
    select_device_checked(S.devices[rank]);
    rc.xarena.reset();
    const int R = S.R;
    Exchange& X = *S.x;
    double xs = 0;
    auto timed = [&](auto&& f) { const double t0 = now_s(); f(); xs += now_s() - t0; };
    auto xalloc = [&](size_t bytes) { return rc.xarena.alloc(bytes ? bytes : 1); };
    static const bool dbg = getenv("AC_MULTI_DEBUG") != nullptr;
    double t_lap = now_s();
    auto lapmsg = [&](const char* what) { if (!dbg) return; stream_sync(); const double t = now_s(); fprintf(stderr, "[multi rank %d] %-22s %8.3f ms\n", rank, what, (t - t_lap) * 1e3); t_lap = t; };

    // ---- this rank's slice: host-side pack + upload, local insert, fragments
    const std::vector<SeqView> mine(S.seqs->begin() + (long)S.first[rank], S.seqs->begin() + (long)S.first[rank + 1]);
    uint64_t my_bases = 0, all_bases = 0;
    for (auto& s : mine) my_bases += s.length;
    for (auto& s : *S.seqs) all_bases += s.length;
    GraphBuilder::set_upload_threads_cap((int)std::max(2u, std::thread::hardware_concurrency() / (unsigned)std::max(S.R, 1)));
    GraphBuilder b(S.k);
    b.set_sequence_index_base(S.first[rank]);
    b.set_sequences_host(mine);
    const uint32_t local_hint = (uint32_t)std::max<uint64_t>(1, (uint64_t)((double)S.assembly_count * (double)my_bases / (double)std::max<uint64_t>(all_bases, 1) + 0.5));
    b.shard_begin(local_hint);
    lapmsg("shard_begin (upload, local insert, fragments)");

    // ---- fragments of all ranks -> the union text ('$' + the ranks' fragment texts in rank order) and the record table, on every rank
    uint64_t v3[3] = {b.fragment_count(), b.fragment_text_bytes(), b.local_distinct_count()};
    std::vector<uint64_t> all3((size_t)R * 3);
    X.gather_values(rank, v3, 3, all3.data());
    uint64_t nf_total = 0, nb_total = 1, distinct_upper = 0;
    std::vector<size_t> tb(R), td(R), mb(R), md(R);
    for (int r = 0; r < R; r++) {
        md[r] = nf_total * 8; mb[r] = all3[3 * r] * 8; td[r] = nb_total; tb[r] = all3[3 * r + 1];
        nf_total += all3[3 * r]; nb_total += all3[3 * r + 1]; distinct_upper += all3[3 * r + 2];
    }
    b.set_distinct_upper_bound(distinct_upper);
    uint64_t frag_text_received = 0;      // bytes of other ranks' fragment text this rank received
    uint8_t* d_meta = (uint8_t*)xalloc(nf_total * 8);
    {
        // the fragment texts travel as 2-bit codes on the union text's word grid (a quarter of the bytes; AC_MULTI_FRAGMENTS=bytes: as text)
        const bool as_bytes = tuning_multi_fragments_as_bytes();
        const Arena::Mark mk = rc.xarena.mark();
        uint8_t* my_meta = (uint8_t*)xalloc(mb[rank]);
        if (as_bytes) {
            uint8_t* d_union = (uint8_t*)xalloc(nb_total + 64);
            uint8_t* my_text = (uint8_t*)xalloc(tb[rank]);
            b.fragments_export(my_text, my_meta);
            const uint8_t dollar = '$';
            copy_h2d(d_union, &dollar, 1);
            lapmsg("fragments export");
            timed([&] { X.all_gather_v(rank, my_text, d_union, tb.data(), td.data()); X.all_gather_v(rank, my_meta, d_meta, mb.data(), md.data()); });
            lapmsg("fragments all-gather");
            stream_sync();
            b.shard_build_union((uint32_t)rank, (uint32_t)R, d_union, nb_total, d_meta, nf_total);
            frag_text_received = (nb_total - 1 - tb[rank]);
        } else {
            std::vector<uint64_t> first_word(R), n_words(R);
            std::vector<size_t> wb(R), wd(R);
            size_t staged = 0;
            for (int r = 0; r < R; r++) {
                first_word[r] = (uint64_t)td[r] >> 5;
                n_words[r] = tb[r] ? (((uint64_t)td[r] + tb[r] - 1) >> 5) - first_word[r] + 1 : 0;
                wb[r] = (size_t)n_words[r] * 8; wd[r] = staged; staged += wb[r];
            }
            uint8_t* d_staged = (uint8_t*)xalloc(staged + 8);
            uint8_t* my_words = (uint8_t*)xalloc(wb[rank] + 8);
            b.fragments_export_packed(td[rank], my_words, my_meta);
            lapmsg("fragments export");
            timed([&] { X.all_gather_v(rank, my_words, d_staged, wb.data(), wd.data()); X.all_gather_v(rank, my_meta, d_meta, mb.data(), md.data()); });
            lapmsg("fragments all-gather");
            stream_sync();
            b.shard_build_union_packed((uint32_t)rank, (uint32_t)R, d_staged, first_word.data(), n_words.data(), nb_total, d_meta, nf_total);
            frag_text_received = staged - wb[rank];
        }
        stream_sync();
        rc.xarena.rewind(mk);
    }
    lapmsg("build_union");

    // ---- novel bitmap, degree bytes, link words: the owners' contributions add up
    {
        const Arena::Mark mk = rc.xarena.mark();
        const uint64_t words = b.bitmap_words();
        void* bm = xalloc(words * 8);
        b.bitmap_export(bm);
        timed([&] { X.all_reduce(rank, bm, words, X_U64, X_SUM); });
        b.shard_build_novel(bm);
        rc.xarena.rewind(mk);
    }
    lapmsg("bitmap + build_novel");
    const uint64_t N = b.distinct_count();
    const uint64_t sib_words = b.sib_words();
    if (sib_words) {      // the sibling bits, 2 per distinct k-mer by novel index: their sum lets every rank settle 97-99 % of the degrees without a probe
        const Arena::Mark mk = rc.xarena.mark();
        void* sb = xalloc(sib_words * 8);
        b.sib_export(sb);
        timed([&] { X.all_reduce(rank, sb, sib_words, X_U64, X_SUM); });
        b.shard_degrees(sb);
        rc.xarena.rewind(mk);
    }
    lapmsg("sibling bits + degrees");
    const uint64_t deg_bytes = b.degree_bytes();
    {
        const Arena::Mark mk = rc.xarena.mark();
        void* deg = xalloc(deg_bytes + 8);
        b.degrees_export(deg);
        timed([&] { X.all_reduce(rank, deg, deg_bytes, X_U8, X_SUM); });
        b.shard_build_graph(deg);
        rc.xarena.rewind(mk);
    }
    lapmsg("degrees + build_graph");
    const uint64_t U = b.unitig_count();
    {
        const Arena::Mark mk = rc.xarena.mark();
        void* lk = xalloc(U * 10 * 4);      // (the walk words follow from these: links_import derives them — 40 instead of 120 bytes per unitig cross)
        b.links_export(lk, nullptr);
        timed([&] { X.all_reduce(rank, lk, U * 10, X_I32, X_SUM); });
        b.links_import(lk, nullptr);
        rc.xarena.rewind(mk);
    }

    lapmsg("links");
    // ---- walk-start keys to their owners, answers back (one all-to-all each way)
    uint64_t nq = b.query_count();
    const uint32_t kw = b.query_key_words();
    uint64_t sent_away = 0, q_recv_total = 0;
    {
        const Arena::Mark mk = rc.xarena.mark();
        void* routed = xalloc(nq * kw * 8);
        std::vector<uint64_t> to(R), from_all((size_t)R * R);
        b.queries_route((uint32_t)R, routed, to.data());
        X.gather_values(rank, to.data(), R, from_all.data());      // from_all[src * R + dst] = queries src sends to dst
        std::vector<size_t> sb(R), sd(R), rb(R), rd(R);
        size_t so = 0, ro = 0;
        for (int r = 0; r < R; r++) {
            sb[r] = (size_t)to[r] * kw * 8; sd[r] = so; so += sb[r];
            rb[r] = (size_t)from_all[(size_t)r * R + rank] * kw * 8; rd[r] = ro; ro += rb[r];
            if (r != rank) { sent_away += to[r]; q_recv_total += from_all[(size_t)r * R + rank]; }
        }
        const uint64_t n_in = ro / ((size_t)kw * 8);
        void* in_keys = xalloc(ro); void* in_ans = xalloc(n_in * 8); void* my_ans = xalloc(nq * 8);
        timed([&] { X.all_to_all_v(rank, routed, sb.data(), sd.data(), in_keys, rb.data(), rd.data()); });
        b.answer_queries(in_keys, n_in, in_ans);
        for (int r = 0; r < R; r++) { sb[r] /= kw; sd[r] /= kw; rb[r] /= kw; rd[r] /= kw; }      // answers: one word per key, the same routes backwards
        timed([&] { X.all_to_all_v(rank, in_ans, rb.data(), rd.data(), my_ans, sb.data(), sd.data()); });
        b.shard_walk_routed(my_ans);
        rc.xarena.rewind(mk);
    }

    lapmsg("queries + walk");
    // ---- per-unitig quantities over all sequences
    {
        const Arena::Mark mk = rc.xarena.mark();
        int32_t* red = (int32_t*)xalloc(U * 5 * 4);
        b.reduce_export(red, red + 3 * U);
        timed([&] { X.all_reduce(rank, red, 3 * U, X_I32, X_SUM); X.all_reduce(rank, red + 3 * U, 2 * U, X_I32, X_MIN); });
        b.reduce_import(red, red + 3 * U);
        rc.xarena.rewind(mk);
    }
    lapmsg("reduce");
    // ---- the order-sensitive tail; expand_repeats on this rank's share of the junctions (conflict components), the results merged by two
    // SUM all-reduces inside shard_finish (kernels_tail.inc; AC_MULTI_TAIL=replicated: every rank runs every junction, as before round 4);
    // rank 0 keeps unitigs + links, every rank the paths of its own sequences
    uint64_t tail_bytes = 0;
    const bool tail_replicated = tuning_multi_tail_replicated();
    if (!tail_replicated)
        b.set_tail_exchange([&](void* d_buf, uint64_t count, int dtype, int op) {
            tail_bytes += count * (dtype == 0 ? 1 : 4);
            timed([&] { X.all_reduce(rank, d_buf, count, dtype == 0 ? X_U8 : X_I32, op == 0 ? X_SUM : X_MIN); });
        });
    b.shard_finish(&S.graphs[rank], rank == 0, true);
    S.tms[rank] = b.timings();
    lapmsg("finish");
    S.xsec[rank] = xs;
    {
        std::lock_guard<std::mutex> lock(S.st_mu);
        MultiStats& st = S.st;
        const uint64_t others_frag = frag_text_received + (nf_total * 8 - mb[rank]);
        st.bytes_fragments += others_frag;
        st.bytes_bitmap += b.bitmap_words() * 8 * (uint64_t)(R - 1) / (uint64_t)R * 2;      // reduce-scatter + all-gather of an all-reduce: 2 (R - 1) / R of the buffer per rank
        st.bytes_degrees += deg_bytes * (uint64_t)(R - 1) / (uint64_t)R * 2;
        st.bytes_sibling += sib_words * 8 * (uint64_t)(R - 1) / (uint64_t)R * 2;
        st.degrees_open = b.timings().n_degrees_open;
        st.bytes_links += U * 40 * (uint64_t)(R - 1) / (uint64_t)R * 2;
        st.bytes_reduce += U * 20 * (uint64_t)(R - 1) / (uint64_t)R * 2;
        st.bytes_queries += q_recv_total * kw * 8; st.bytes_answers += sent_away * 8;
        st.queries_total += nq; st.queries_sent_away += sent_away;
        st.table_capacity_max = std::max<uint64_t>(st.table_capacity_max, b.timings().table_capacity);
        st.table_capacity_sum += b.timings().table_capacity;
        st.union_text_bytes = nb_total; st.fragments = nf_total; st.distinct = N;
        st.candidates_total = b.timings().n_candidates;
        st.candidates_owned_max = std::max<uint64_t>(st.candidates_owned_max, b.timings().n_candidates_owned);
        st.bytes_tail += tail_bytes * (uint64_t)(R - 1) / (uint64_t)R * 2;
        // what THIS rank received over the whole build (an all-reduce of S bytes: 2 (R - 1) / R x S on a ring)
        const uint64_t ar = (b.bitmap_words() * 8 + sib_words * 8 + deg_bytes + U * 40 + U * 20 + tail_bytes) * (uint64_t)(R - 1) / (uint64_t)R * 2;
        const uint64_t mine = others_frag + ar + q_recv_total * kw * 8 + sent_away * 8;
        st.bytes_received_max = std::max(st.bytes_received_max, mine);
        st.path_runs_copied += b.timings().path_runs_copied;
    }
}

}  // namespace

void release_multi_contexts() {
    for (auto& rc : rank_ctxs()) {
        if (!rc) continue;
        // the objects of a context free device / pinned memory: with the device they live on current
#ifndef AC_EMU
        if (rc->ctx.arena_device >= 0) (void)hipSetDevice(rc->ctx.arena_device);
#endif
        rc.reset();
    }
    rank_ctxs().clear();
#ifndef AC_EMU
    RcclApi::get().destroy_comms();
#endif
}

void build_multi(uint32_t k, uint32_t assembly_count, const std::vector<SeqView>& seqs, const std::vector<int>& devices_in, int transport,
                 FinalGraph* out, BuildTimings* tm, MultiStats* st_out) {
    const double t_begin = now_s();
    if (devices_in.empty()) throw DeviceError("ac_compress_build_multi: no devices");
    if (seqs.empty()) throw DeviceError("no sequences found in input assemblies");
    // fewer ranks than devices when there are fewer sequences; contiguous slices balanced by bases (rank order = sequence order)
    const int R = (int)std::min<size_t>(devices_in.size(), seqs.size());
    std::vector<int> devices(devices_in.begin(), devices_in.begin() + R);
    uint64_t total = 0;
    for (auto& s : seqs) total += s.length;
    Shared S;
    S.k = k; S.assembly_count = assembly_count; S.R = R; S.seqs = &seqs; S.devices = devices;
    S.first.assign((size_t)R + 1, seqs.size());
    S.first[0] = 0;
    {
        uint64_t acc = 0; int r = 1;      // r = next rank to start
        for (size_t i = 0; i < seqs.size() && r < R; i++) {
            acc += seqs[i].length;
            const size_t rem = seqs.size() - (i + 1), need = (size_t)(R - r);      // sequences left / ranks that still have to get one
            // rank r starts behind sequence i when the ranks before it hold their share of the bases — and at the latest when the
            // sequences that are left are just enough for the ranks that are left
            if (rem >= need && (acc * (uint64_t)R >= total * (uint64_t)r || rem == need)) S.first[r++] = i + 1;
        }
    }
    for (int r = 0; r < R; r++) if (S.first[r + 1] <= S.first[r]) throw DeviceError("internal error: a rank of the multi-device build got no sequence");
    // One device (or one sequence): there is nothing to exchange and nobody to dedup against — the job IS a single-device build, and it
    // takes that path (round 5: the protocol at world size 1 cost 1.7x the build it stands for).  A transport named explicitly
    // (AC_MULTI_TRANSPORT: the device suite's RCCL-in-a-world-of-one tests) still runs every phase and every collective.
    if (R == 1 && transport == MULTI_AUTO) {
        select_device_checked(devices[0]);
        GraphBuilder b(k);
        b.set_sequences_host(seqs);
        b.build(assembly_count, out);
        *tm = b.timings();
        MultiStats st;
        st.n_ranks = 1; st.transport = MULTI_DIRECT;
        st.table_capacity_max = st.table_capacity_sum = tm->table_capacity;
        st.distinct = tm->n_distinct;
        st.candidates_total = st.candidates_owned_max = tm->n_candidates;
        st.seconds_total = now_s() - t_begin;
        if (st_out) *st_out = st;
        return;
    }
    [[maybe_unused]] bool distinct = true;
    for (int a = 0; a < R; a++) for (int c = a + 1; c < R; c++) if (devices[a] == devices[c]) distinct = false;
#ifdef AC_EMU
    transport = MULTI_HOST_STAGED;
#else
    if (transport == MULTI_AUTO) transport = distinct ? MULTI_RCCL : MULTI_HOST_STAGED;
    if (transport == MULTI_RCCL && !distinct) throw DeviceError("the RCCL transport needs a device of its own for every rank");
#endif
    RankBarrier bar(R);
    std::unique_ptr<Exchange> X;
#ifndef AC_EMU
    if (transport == MULTI_RCCL) { RcclApi::get().ensure_comms(devices); X.reset(new RcclExchange(R, &bar)); }
#endif
    if (!X) X.reset(new HostStagedExchange(R, &bar));
    S.x = X.get(); S.bar = &bar;
    S.graphs.resize(R); S.tms.resize(R); S.xsec.assign(R, 0.0);
    while ((int)rank_ctxs().size() < R) rank_ctxs().emplace_back(new RankCtx);
    std::vector<std::thread> threads;
    for (int r = 0; r < R; r++)
        threads.emplace_back([&S, r, &bar] {
            try { rank_main(S, r); }
            catch (const std::exception& e) { bar.fail(std::string("rank ") + std::to_string(r) + ": " + e.what()); }
            catch (...) { bar.fail("rank " + std::to_string(r) + ": unknown error"); }
        });
    for (auto& t : threads) t.join();
    if (bar.failed()) throw DeviceError(bar.why());

    // ---- one graph: rank 0's unitigs and links, the ranks' paths one behind the other (rank order = sequence order)
    *out = std::move(S.graphs[0]);
    uint64_t n_ent = 0;
    std::vector<uint64_t> base(R);
    for (int r = 0; r < R; r++) { base[r] = n_ent; n_ent += (r == 0 ? out->n_path : S.graphs[r].n_path); }
    if (R > 1) {
        HostBlock all = PinnedPool::get().alloc(n_ent * 4);
        std::vector<uint64_t> off; off.reserve(seqs.size() + 1);
        std::vector<std::thread> copiers;
        for (int r = 0; r < R; r++) {
            const FinalGraph& g = r == 0 ? *out : S.graphs[r];
            for (size_t s = 0; s + 1 < g.path_off.size(); s++) off.push_back(base[r] + g.path_off[s]);
            copiers.emplace_back([&, r] { const FinalGraph& gg = r == 0 ? *out : S.graphs[r]; memcpy((int32_t*)all.p + base[r], gg.path, gg.n_path * 4); });
        }
        off.push_back(n_ent);
        for (auto& t : copiers) t.join();
        out->path_block = std::move(all);
        out->path = (const int32_t*)out->path_block.p;
        out->n_path = n_ent;
        out->path_off = off;
    }
    *tm = S.tms[0];
    tm->n_path_entries = n_ent;
    S.st.n_ranks = (uint32_t)R; S.st.transport = transport;
    S.st.seconds_total = now_s() - t_begin;
    for (double x : S.xsec) S.st.seconds_exchange_max = std::max(S.st.seconds_exchange_max, x);
    if (st_out) *st_out = S.st;
}

}  // namespace ac
