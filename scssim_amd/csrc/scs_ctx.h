// scs_ctx.h -- private to the library's host side: the context behind the C ABI's opaque scs_ctx, its device buffers, and the
// functions the host files share (scs_pipeline.cpp: mailbox, C ABI; scs_stage.cpp: profile, genome, fragments; scs_amplify.cpp:
// Malbac::amplify; scs_reads.cpp: read allocation, yieldReads, the sink).
#pragma once
#include "../../include/scssim_hip.h"
#include <sched.h>
#include <pthread.h>
#include "scs_device.h"
#include "scs_seams.h"
#include "scs_tables.h"
#include "scs_comm.h"
#include "scs_bgzf.h"

#include <atomic>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>


namespace scs {

struct ScsError : std::runtime_error { int code; ScsError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    throw ScsError(SCS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// growable device buffer.  Small buffers are plain hipMalloc blocks.  A buffer that grows past 64 MB moves (once) into
// a reserved virtual address range and from then on grows IN PLACE by mapping more physical memory behind it
// (hipMemAddressReserve / hipMemCreate / hipMemMap): no reallocate-copy-free cycles while the amplicon arrays of a
// whole-genome job grow cycle by cycle, no transient 2.5x footprint -- and fresh hipMalloc memory costs about 20 ms per
// GB on its first touch on this platform (measured), mapped chunks do not.
struct DevBuf {
    void* p = nullptr; size_t cap = 0;       // cap: usable (mapped) bytes
    size_t va = 0;                           // reserved address range in bytes (0: plain hipMalloc block)
    // equal-sized chunks: on ROCm 7.2 hipMemSetAccess rejects a chunk mapped right behind one of a different size (probed)
    static constexpr size_t kRange = 384ull << 30, kGran = 128ull << 20;
    static size_t virtual_from() {           // SCS_VMM_FROM_MB: tests lower it so that small jobs run on mapped buffers too
        static const size_t v = seam_env("SCS_VMM_FROM_MB") ? (size_t)atol(seam_env("SCS_VMM_FROM_MB")) << 20 : 64ull << 20;
        return v;
    }
    static bool& virtual_ok() { static bool ok = seam_env("SCS_NO_VMM") == nullptr; return ok; }
    void map_more(size_t ncap) {             // map [cap, ncap) of the reserved range, kGran at a time
        int dev = 0; HIP_OK(hipGetDevice(&dev));
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
        while (cap < ncap) {
            hipMemGenericAllocationHandle_t h;
            HIP_OK(hipMemCreate(&h, kGran, &prop, 0));
            hipError_t e = hipMemMap((char*)p + cap, kGran, 0, h, 0);
            if (e == hipSuccess) { e = hipMemSetAccess((char*)p + cap, kGran, &acc, 1); if (e != hipSuccess) (void)hipMemUnmap((char*)p + cap, kGran); }
            (void)hipMemRelease(h);          // the mapping keeps the memory alive
            if (e != hipSuccess) throw ScsError(SCS_EDEVICE, std::string("device memory map: ") + hipGetErrorString(e));
            cap += kGran;
        }
    }
    void reserve(size_t bytes, hipStream_t s, size_t keep_bytes = 0) {
        if (bytes <= cap) return;
        // a mapped buffer grows in place, chunk by chunk: it takes what is asked for plus 3 % (growing it by half, as a block that
        // must be copied is, mapped tens of GB in the middle of a job whose amplicon count came out 0.01 % above the last one's)
        size_t ncap = va ? bytes + bytes / 32 : std::max(bytes, cap + cap / 2);
        if (va) { map_more(std::min((ncap + kGran - 1) / kGran * kGran, va)); if (bytes > cap) throw ScsError(SCS_EOVERFLOW, "device buffer larger than its address range"); return; }
        if (ncap > virtual_from() && virtual_ok()) {
            void* base = nullptr;
            if (hipMemAddressReserve(&base, kRange, kGran, nullptr, 0) == hipSuccess) {
                void* old = p; const size_t old_cap = cap;
                p = base; cap = 0; va = kRange;
                try { map_more((ncap + kGran - 1) / kGran * kGran); }
                catch (...) { (void)hipMemAddressFree(base, kRange); p = old; cap = old_cap; va = 0; throw; }
                if (old && keep_bytes) { HIP_OK(hipMemcpyAsync(p, old, keep_bytes, hipMemcpyDeviceToDevice, s)); HIP_OK(hipStreamSynchronize(s)); }
                if (old) HIP_OK(hipFree(old));
                return;
            }
            (void)hipGetLastError(); virtual_ok() = false;                        // no virtual memory management here: classic path from now on
        }
        void* np = nullptr;
        HIP_OK(hipMalloc(&np, ncap));
        if (p && keep_bytes) { HIP_OK(hipMemcpyAsync(np, p, keep_bytes, hipMemcpyDeviceToDevice, s)); HIP_OK(hipStreamSynchronize(s)); }
        if (p) HIP_OK(hipFree(p));
        p = np; cap = ncap;
    }
    void release() {
        if (p && va) { for (size_t o = 0; o < cap; o += kGran) (void)hipMemUnmap((char*)p + o, kGran); (void)hipMemAddressFree(p, va); }
        else if (p) (void)hipFree(p);
        p = nullptr; cap = 0; va = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

struct AmpStore {            // SoA amplicon arrays (DevAmps) with capacity management
    DevBuf parent, sl, gc, primers, uid, errs; uint32_t n = 0, cap = 0;
    DevBuf pool, pool_head; uint32_t pool_cap = 0;
    void reserve(uint64_t want, hipStream_t s) {
        if (want > 0xFFFFFFF0ull) throw ScsError(SCS_EOVERFLOW, "amplicon count exceeds 2^32 (reference limit: Malbac.cpp:376,386)");
        if (want <= cap) return;
        uint64_t ncap = uid.va ? want + want / 32 : std::max<uint64_t>(want, (uint64_t)cap + cap / 2);   // (mapped arrays grow in place: see DevBuf::reserve)
        ncap = std::min<uint64_t>(std::max<uint64_t>(ncap, 1u << 16), 0xFFFFFFF0ull);
        parent.reserve(ncap * 4, s, (size_t)n * 4); sl.reserve(ncap * 4, s, (size_t)n * 4);
        gc.reserve(ncap * 2, s, (size_t)n * 2); primers.reserve(ncap * 2, s, (size_t)n * 2);
        uid.reserve(ncap * 8, s, (size_t)n * 8); errs.reserve(ncap * 8, s, (size_t)n * 8);
        cap = (uint32_t)ncap;
    }
    void reserve_pool(uint32_t entries, hipStream_t s) {
        if (!pool_head.p) { pool_head.reserve(256, s); HIP_OK(hipMemsetAsync(pool_head.p, 0, 4, s)); }
        if (entries > pool_cap) {
            uint32_t used = 0;
            if (pool_cap) { HIP_OK(hipMemcpyAsync(&used, pool_head.p, 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s)); used = std::min(used, pool_cap); }
            pool.reserve((size_t)entries * 4, s, (size_t)used * 4); pool_cap = entries;
        }
    }
    DevAmps view() const { return DevAmps{parent.as<uint32_t>(), sl.as<uint32_t>(), gc.as<uint16_t>(), primers.as<uint16_t>(), uid.as<uint64_t>(), errs.as<uint64_t>()}; }
    DevErrPool pool_view() const { return DevErrPool{pool.as<uint32_t>(), pool_head.as<uint32_t>(), pool_cap}; }
    void reset(hipStream_t s) { n = 0; if (pool_head.p) HIP_OK(hipMemsetAsync(pool_head.p, 0, 4, s)); }
    void reset_counts() { n = 0; }                                                // the pool head is zeroed by k_amplify_init
    void release() { parent.release(); sl.release(); gc.release(); primers.release(); uid.release(); errs.release(); pool.release(); pool_head.release(); n = cap = pool_cap = 0; }
};

static const bool kAlwaysTimed = true;
struct KernelTimer {         // HIP events on the ctx stream around the launches of one kernel (scs_set_kernel_timing turns one off)
    const char* name; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; double ms = 0; uint64_t launches = 0; uint64_t units = 0; bool on = true; const bool* gate = &kAlwaysTimed;
    void add_units(uint64_t n) { if (on && *gate) units += n; }
    void begin(hipStream_t s) {
        if (!on || !*gate) return;
        if (used == ev.size()) { hipEvent_t a, b; HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); ev.push_back({a, b}); }
        HIP_OK(hipEventRecord(ev[used].first, s));
    }
    void end(hipStream_t s) { if (!on || !*gate) return; HIP_OK(hipEventRecord(ev[used].second, s)); ++used; }
    void collect() {         // call after a stream sync
        for (size_t i = 0; i < used; ++i) { float t = 0; HIP_OK(hipEventElapsedTime(&t, ev[i].first, ev[i].second)); ms += t; ++launches; }
        used = 0;
    }
    void reset() { ms = 0; launches = 0; units = 0; used = 0; }
    void release() { for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } ev.clear(); }
};


struct SinkPipe;
}  // namespace scs
using namespace scs;

struct Mail {                // a batch of device scalars for one k_mail post (at most 16)
    const void* src[16]; int wd[16]; int dst[16]; int n = 0; unsigned clear = 0;
    void add(const void* p, int width, int slot, bool clear_after = false) { src[n] = p; wd[n] = width; dst[n] = slot; if (clear_after) clear |= 1u << n; ++n; }
};

struct scs_ctx {
    scs_config cfg; std::string err;
    hipStream_t stream = nullptr; bool own_stream = false;
    RngKey key{0, 0};
    // model
    ProfileTables prof; bool have_profile = false; DevTables dtb{};
    DevBuf d_tables, t_gap, t_qcompact, t_guide, t_ring1, t_ring2, t_ring1u, t_ring2u, t_subs1, t_subs2, t_qual, t_ins, t_del, t_isize, d_subs1, d_subs2, d_qual, d_ins, d_del, d_isize, d_gcmeans;
    // genome + fragments
    DevBuf gx_gc_bits, gx_n_bits, gx_gc_cnt, gx_n_cnt, gx_gc_pref, gx_n_pref, gx_gc_pair, d_binom;
    std::vector<FastaRecord> recs; bool have_genome = false; DevBuf genome, genome2; std::vector<uint64_t> rec_off, rec_len; uint64_t genome_bases = 0;   // recs: names only once staged
    std::vector<uint64_t> f_goff; std::vector<uint32_t> f_len; std::vector<int8_t> f_strand; std::vector<uint32_t> f_primers;
    uint64_t f_gidx_base = 0; bool have_frags = false;
    uint64_t slice_base = 0, slice_len = 0; bool sliced = false;   // sharded job staged from a FASTA: only the bases of this shard's fragments are resident (genome coordinate slice_base ..)
    DevBuf df_blob, df_primers, df_hasn; size_t df_len_off = 0, df_strand_off = 0;           // fragments: offsets | lengths | strands in one block
    uint8_t* h_frag = nullptr; size_t h_frag_cap = 0; bool frag_copy_pending = false;   // its pinned staging copy
    // amplicons
    AmpStore semis, fulls;
    DevBuf budget_f, budget_s, slot_off_f, slot_off_s, dsums, poisson_part; uint64_t* h_rb = nullptr;   // dsums: device scalars; h_rb: pinned, device-mapped mailbox (32 words)
    unsigned long long* d_rb = nullptr; uint64_t mail_seq = 0;                      // device address of h_rb; sequence of the last post
    Mail pend;                                                                     // counts of the passes launched since the last collect
    bool timing_gate = true; uint32_t timing_every = 1; uint64_t amplify_calls = 0, yield_calls = 0;   // scs_set_kernel_timing: events on every n-th call
    uint64_t frag_total_len = 0, semi_total_len = 0; uint32_t slots_f = 0, slots_s = 0, budget_ns = 0;
    uint64_t nf_all = 0, frag_len_all = 0; bool budgets_pending = false;            // sharded job: fragments of ALL shards; budgets not yet exchanged
    DevBuf primer_cnt, primer_delta, primer_cut, primer_gdelta; uint64_t total_primers = 0; bool amplified = false;   // stock, what the running pass took (this shard / all shards), the cuts k_attach reads
    DevBuf st_eidx, st_etype, st_estart, st_info, st_list, st_sorted, st_tmp, att_wave_first; uint64_t min_stock_lb = 0;   // exact_stock's work arrays; lower bound of every primer stock in use
    DevBuf slots, slot_tmpl, valid, valid_off, valid_f, valid_off_f, scan_tmp, flags;
    // allocation + reads
    DevBuf weights, read_numbers, pair_off, pairs, odd_before, a_part, a_tp, a_probs, a_quota, a_poff, a_plan, a_crn, a_scratch, a_brow, a_bmap, a_send, a_gath, a_odd; SegMap gmap{}; std::vector<uint32_t> h_read_numbers; uint64_t reads_requested = 0, n_pairs_planned = 0; bool allocated = false;
    DevBuf slot_b, slot_q, lens, ev_hdr, ev_dat, sizes1, sizes2, off1, off2, out1, out2, out1b, out2b, rl_cls, rl_pos, rl_lists, d_bounds; SinkPipe* pipe = nullptr;
    hipStream_t errs_stream = nullptr; hipEvent_t ev_att = nullptr, ev_errs = nullptr; bool errs_pending = false;   // k_errs<semi->full> of a cycle runs beside the fragment pass that follows it
    DevBuf slots_fr, slot_tmpl_fr;                        // the fragment passes' own slot arrays (the semi pass's are still being read then)
    // BGZF made on the device (scs_bgzf.hip): per mate the blocks' plans / sizes / offsets, two sets of output buffers, the CRC tables; the
    // blocks' total per batch reaches the host through a small pinned array (h_z) behind an event, one batch late (see do_yield)
    DevBuf z_plan[2], z_sizes[2], z_offs[2], z_out[2][2], z_crc; uint32_t* h_z = nullptr; hipEvent_t ev_z[2] = {nullptr, nullptr};
    bool want_cks = false; DevBuf d_cks; std::vector<uint64_t> cks;   // scs_set_batch_checksums: per batch and mate, computed where the text lies in HBM
    ReadsSide reads_side;                                 // k_reads' two small class kernels run beside the big one on these (per ctx: two contexts on one device do not share events)
    hipStream_t pre_stream = nullptr; hipEvent_t ev_pre[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_plan = nullptr;   // the reads stage's pre-pass on its own stream, beside the previous batch's base pass
    hipStream_t mail_stream = nullptr;                                             // the stream of the last post (mail_wait watches it)
    hipStream_t copy_stream = nullptr; hipEvent_t ev_made[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};   // sink mode: D2H on its own stream, behind the batch's k_reads
    // sharded single job: collectives supplied by the caller + segment bookkeeping of the local amplicon lists
    scs_allreduce_fn allreduce = nullptr; scs_allgatherv_fn allgatherv = nullptr; void* coll_user = nullptr;
    scs_allreduce_dev_fn allreduce_dev = nullptr; scs_allgather_dev_fn allgather_dev = nullptr; void* coll_dev_user = nullptr;
    DevBuf d_tot, d_stage, d_mail;
    std::vector<uint32_t> semi_block_end;                  // local semi count after each fragment pass
    struct Seg { int c, p; uint32_t count; }; std::vector<Seg> full_segs;   // local fulls list = these, in order
    DevBuf d_hostred;
    RcclComm* rccl = nullptr;                              // scs_comm_init: RCCL communicator of this shard (the device collectives then run on it)
    uint32_t seg_lo[ALLOC_SLOTS + 1] = {0};                // first local amplicon of each list segment slot (do_allocate); [ALLOC_SLOTS] = amplicon count
    int pending_seg_cycle = -1;
    // collectives run when the job is sharded -- or whenever hooks are installed (1-shard jobs then exercise them too)
    bool sharded() const { return cfg.shard_count > 1 || allreduce || allreduce_dev; }
    void reduce(uint64_t* v, uint64_t n) {
        if (!sharded()) return;
        if (allreduce_dev) {                                                       // device hook installed: two small copies beat the host hook's round trip
            d_hostred.reserve(n * 8, stream);
            HIP_OK(hipMemcpyAsync(d_hostred.p, v, n * 8, hipMemcpyHostToDevice, stream));
            if (allreduce_dev(coll_dev_user, d_hostred.p, n, 8)) throw ScsError(SCS_EINVAL, "sharded job: device all-reduce hook failed");
            HIP_OK(hipMemcpyAsync(v, d_hostred.p, n * 8, hipMemcpyDeviceToHost, stream)); HIP_OK(hipStreamSynchronize(stream));
            return;
        }
        if (!allreduce || allreduce(coll_user, v, n)) throw ScsError(SCS_EINVAL, "sharded job: all-reduce hook missing or failed (scs_set_collectives)");
    }
    // sum a device array over all shards, in place, ordered on the ctx stream when the device hook is set
    void reduce_dev(void* d, uint64_t n, int elem_bytes) {
        if (!sharded()) return;
        if (allreduce_dev) { if (allreduce_dev(coll_dev_user, d, n, elem_bytes)) throw ScsError(SCS_EINVAL, "sharded job: device all-reduce hook failed"); return; }
        std::vector<uint64_t> v(n);                                                // fallback: stage through the host hook
        if (elem_bytes == 8) { HIP_OK(hipMemcpyAsync(v.data(), d, n * 8, hipMemcpyDeviceToHost, stream)); HIP_OK(hipStreamSynchronize(stream)); }
        else { std::vector<uint32_t> w(n); HIP_OK(hipMemcpyAsync(w.data(), d, n * 4, hipMemcpyDeviceToHost, stream)); HIP_OK(hipStreamSynchronize(stream)); for (uint64_t i = 0; i < n; ++i) v[i] = w[i]; }
        reduce(v.data(), n);
        if (elem_bytes == 8) { HIP_OK(hipMemcpyAsync(d, v.data(), n * 8, hipMemcpyHostToDevice, stream)); HIP_OK(hipStreamSynchronize(stream)); }
        else { std::vector<uint32_t> w(n); for (uint64_t i = 0; i < n; ++i) w[i] = (uint32_t)std::min<uint64_t>(v[i], 0xFFFFFFFFull);
               HIP_OK(hipMemcpyAsync(d, w.data(), n * 4, hipMemcpyHostToDevice, stream)); HIP_OK(hipStreamSynchronize(stream)); }
    }
    // every shard's `bytes` at d_send -> d_recv[r * bytes ..], ordered on the ctx stream when the device hook is set
    void gather_dev(const void* d_send, void* d_recv, uint64_t bytes) {
        if (allgather_dev) { if (allgather_dev(coll_dev_user, d_send, d_recv, bytes)) throw ScsError(SCS_EINVAL, "sharded job: device all-gather hook failed"); return; }
        std::vector<uint8_t> h(bytes), all((size_t)bytes * cfg.shard_count); std::vector<uint64_t> sizes(cfg.shard_count, 0);
        HIP_OK(hipMemcpyAsync(h.data(), d_send, bytes, hipMemcpyDeviceToHost, stream)); HIP_OK(hipStreamSynchronize(stream));
        if (!allgatherv || allgatherv(coll_user, h.data(), bytes, all.data(), bytes, sizes.data())) throw ScsError(SCS_EINVAL, "sharded job: all-gather hook missing or failed (scs_set_collectives)");
        HIP_OK(hipMemcpyAsync(d_recv, all.data(), all.size(), hipMemcpyHostToDevice, stream)); HIP_OK(hipStreamSynchronize(stream));
    }
    scs_stats st{};
    KernelTimer tm_errscan{"k_errs<semi->full>"}, tm_errscan_f{"k_errs<frag->semi>"}, tm_reads{"k_reads"}, tm_attach{"k_attach<semi>"}, tm_indels{"k_indels"}, tm_attach_f{"k_attach<frag>"};

    DevFrags frags_view() const {
        uint8_t* b = df_blob.as<uint8_t>();
        return DevFrags{(uint64_t*)b, (uint32_t*)(b + df_len_off), (int8_t*)(b + df_strand_off), df_primers.as<uint32_t>(), (uint32_t)f_len.size(), f_gidx_base, df_hasn.as<uint8_t>()};
    }
};


namespace scs {
// ---- mailbox: device scalars -> pinned host words (scs_pipeline.cpp)
void mail_post(scs_ctx* c, const Mail& m, bool last, hipStream_t st = nullptr);   // last: the post the host will wait for; st: the ctx stream unless given
void mail_wait(scs_ctx* c);                                                      // everything posted so far has landed in h_rb
void flags_eval(scs_ctx* c);
void check_flags(scs_ctx* c);
template <class T>
inline void upload(DevBuf& b, const std::vector<T>& v, hipStream_t s, size_t extra = 0) {
    b.reserve(std::max<size_t>((v.size() + extra) * sizeof(T), 16), s);
    if (!v.empty()) HIP_OK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
}


// ---- scs_stage.cpp
void do_load_profile(scs_ctx* c, const char* path);
void index_genome(scs_ctx* c, uint64_t tot);
void stage_genome(scs_ctx* c, const void* d_ascii = nullptr, const uint64_t* d_lens = nullptr);
void stage_fasta_on_device(scs_ctx* c, const std::string& path_in);
bool stage_fasta_slice(scs_ctx* c, const std::string& path);
void do_create_frags(scs_ctx* c);
// ---- scs_amplify.cpp
void do_amplify(scs_ctx* c);
// ---- scs_reads.cpp
void do_allocate(scs_ctx* c, uint64_t reads);
struct CallbackSink : BatchSink {           // a caller's scs_sink_fn as a BatchSink: one region, the batches in record order
    scs_sink_fn fn; void* user;
    CallbackSink(scs_sink_fn f, void* u) : fn(f), user(u) {}
    int put(int, const char* a, size_t na, const char* b, size_t nb) override { return fn(user, a, na, b, nb); }
};
struct OutTarget { bool device; char* d1; char* d2; size_t cap1, cap2; BatchSink* sink;
                   std::vector<uint64_t>* seg_off1 = nullptr; std::vector<uint64_t>* seg_off2 = nullptr;
                   bool bgzf = false; };                                          // bgzf: the sink gets BGZF blocks made on the device instead of the text   // seg_off: byte offset of each list segment's first record (shard index)

void do_yield(scs_ctx* c, const OutTarget& tg, uint64_t* n1_out, uint64_t* n2_out, uint64_t* pairs_out);
void sink_pipe_free(scs_ctx* c);            // releases and deletes c->pipe
std::vector<int> gpu_local_cpus(int device);

template <class F>
int guarded(scs_ctx* c, F f) {
    if (!c) return SCS_EINVAL;
    try { if (c->cfg.device >= 0) (void)hipSetDevice(c->cfg.device); f(); return SCS_OK; }
    catch (const ScsError& e) { c->err = e.what(); return e.code; }
    catch (const std::exception& e) { c->err = e.what(); return SCS_EIO; }
}

}  // namespace scs
