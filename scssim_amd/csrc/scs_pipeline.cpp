// scs_pipeline.cpp -- host orchestration behind the C ABI (include/scssim_hip.h).
// One scs_ctx = one HIP device + one stream; all amplicon state lives in HBM as flat SoA arrays.
// Reference call sequence reproduced: src/scssim.cpp:46-67 (genreads branch of main()).
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

using namespace scs;

namespace {

thread_local std::string g_create_error;

struct ScsError : std::runtime_error { int code; ScsError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    throw ScsError(SCS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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

}  // namespace

namespace { struct SinkPipe; }
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
    DevBuf gx_gc_bits, gx_n_bits, gx_gc_cnt, gx_n_cnt, gx_gc_pref, gx_n_pref, d_binom;
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

namespace {

// ---- mailbox: device scalars -> pinned host words, no copy and no stream sync (k_mail)
void mail_post(scs_ctx* c, const Mail& m, bool last, hipStream_t st = nullptr) {   // last: the post the host will wait for; st: the ctx stream unless given
    c->mail_stream = st ? st : c->stream;
    launch_mail(c->mail_stream, m.src, m.wd, m.dst, m.n, m.clear, c->d_rb, last ? ++c->mail_seq : 0ull);
}
void mail_wait(scs_ctx* c) {                                                      // everything posted so far has landed in h_rb
    volatile uint64_t* flag = c->h_rb + MAIL_SEQ_SLOT;
    // a post usually lands within tens of microseconds: spin (with the CPU's pause hint) for about that long, then back
    // off -- yield, then short sleeps -- so that a long device phase (a whole-genome pass, a collective waiting for another
    // rank) does not burn a host core; a failed or drained stream must not leave the host waiting either
    for (uint64_t spin = 1;; ++spin) {
        if (*flag == c->mail_seq) break;
        if (spin < 20000) { __builtin_ia32_pause(); continue; }
        if ((spin & 0x3F) == 0) {
            const hipError_t q = hipStreamQuery(c->mail_stream ? c->mail_stream : c->stream);
            if (q == hipSuccess) { if (*flag == c->mail_seq) break; throw ScsError(SCS_EDEVICE, "mailbox: stream drained without the expected post"); }
            if (q != hipErrorNotReady) throw ScsError(SCS_EDEVICE, std::string("mailbox: ") + hipGetErrorString(q));
        }
        if (spin < 20200) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(spin < 21000 ? 20 : 200));
    }
    std::atomic_thread_fence(std::memory_order_acquire);
}

// device-side overflow flags: slot 30 of the mailbox.  flags_eval reads what a post already brought (the caller has
// waited for that post, or synchronised the stream after it); check_flags posts and waits itself.
void flags_eval(scs_ctx* c) {
    { const hipError_t le = take_launch_error(); if (le != hipSuccess) throw ScsError(SCS_EDEVICE, std::string("kernel launch failed: ") + hipGetErrorString(le)); }
    const uint32_t f = (uint32_t)c->h_rb[30];
    if (f) {
        HIP_OK(hipMemsetAsync(c->flags.p, 0, 4, c->stream));
        std::string m = "device work buffer overflow:";
        if (f & FLAG_ERRCAP) m += " per-amplicon error list";
        if (f & FLAG_ERRPOOL) m += " error overflow pool";
        if (f & FLAG_READSLOT) m += " read slot (indel-extended read longer than the slot)";
        if (f & FLAG_INTERNAL) m += " internal";
        throw ScsError(SCS_EOVERFLOW, m);
    }
}
void check_flags(scs_ctx* c) {
    Mail m; m.add(c->flags.p, 4, 30); mail_post(c, m, true); mail_wait(c);
    flags_eval(c);
}

template <class T>
void upload(DevBuf& b, const std::vector<T>& v, hipStream_t s, size_t extra = 0) {
    b.reserve(std::max<size_t>((v.size() + extra) * sizeof(T), 16), s);
    if (!v.empty()) HIP_OK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

// ---------------------------------------------------------------- profile
void do_load_profile(scs_ctx* c, const char* path) {
    load_profile(path, c->cfg.paired != 0, c->cfg.isize, c->prof);
    ProfileTables& P = c->prof; hipStream_t s = c->stream;
    upload(c->t_subs1, P.subs1_t, s); upload(c->t_subs2, P.subs2_t, s); upload(c->t_qual, P.qual_t, s);
    upload(c->t_qcompact, P.qual_alias, s); upload(c->t_ins, P.ins_t, s); upload(c->t_del, P.del_t, s); upload(c->t_isize, P.isize_t, s); upload(c->t_gap, P.gap_t, s);
    upload(c->d_subs1, P.subs1, s); upload(c->d_subs2, P.subs2, s); upload(c->d_qual, P.qual, s);
    upload(c->d_ins, P.ins_cdf, s); upload(c->d_del, P.del_cdf, s); upload(c->d_isize, P.isize_cdf, s);
    std::vector<double> gm(P.gc_means, P.gc_means + 101); upload(c->d_gcmeans, gm, s);
    // the bins as k_reads keeps them in its LDS ring (scs_kernels.hip RingBin): the four diagonal alias rows (c, c), then the
    // threshold triples of the 64 clean 3-mers; a workgroup refills its ring with straight 16-byte copies of this image
    auto ring_image = [&](const std::vector<uint32_t>& subs_t) {
        const size_t B = (size_t)P.bins, qw = (size_t)P.qual_k + (size_t)P.qual_k / 4, bw = 4 * qw + 192, Bpad = (B + 7) & ~(size_t)7;
        std::vector<uint32_t> img(Bpad * bw + 64, 0u);                              // + the head: threshold triples of the 1-mers at bin 0 and the 2-mers at bin 1
        for (size_t ki = 0; ki < 20 && B >= 2; ++ki) memcpy(img.data() + Bpad * bw + ki * 3, subs_t.data() + (ki * B + (ki < 4 ? 0 : 1)) * 4, 12);
        for (size_t b = 0; b < B; ++b) {
            uint32_t* d = img.data() + b * bw;
            for (size_t cc = 0; cc < 4; ++cc) memcpy(d + cc * qw, P.qual_alias.data() + ((cc * 5) * B + b) * qw, qw * 4);
            for (size_t kk = 0; kk < 64; ++kk) memcpy(d + 4 * qw + kk * 3, subs_t.data() + ((20 + kk) * B + b) * 4, 12);
        }
        return img;
    };
    // the uniform walk's image (RingBinU): instead of the three thresholds, the interval of draws that KEEP the window's base c
    // of the 3-mer -- k = (x >= T0) + (x >= T1) + (x >= T2) equals c  <=>  lo <= x < hi with lo = T[c-1] (0 for c = 0), hi = T[c]
    // (2^32 for c = 3) -- as (lo, width): kept <=> x - lo < width, one subtraction and one compare.  hi is capped at 2^32 - 1, so the
    // draw 0xFFFFFFFF (whose base call needs the double tables) is never "kept" and takes the walk's rare path like a substitution.
    auto keep_pair = [&](const uint32_t* T, uint32_t cbase, uint32_t* out) {
        const uint32_t lo = cbase ? T[cbase - 1] : 0u, hi = cbase < 3 ? T[cbase] : 0xFFFFFFFFu;
        out[0] = lo; out[1] = hi > lo ? hi - lo : 0u;
    };
    auto ring_image_u = [&](const std::vector<uint32_t>& subs_t) {
        const size_t B = (size_t)P.bins, qw = (size_t)P.qual_k + (size_t)P.qual_k / 4, bw = 4 * qw + 128, Bpad = (B + 7) & ~(size_t)7;
        std::vector<uint32_t> img(Bpad * bw + 64, 0u);                              // + the head: the 1-mers at bin 0 and the 2-mers at bin 1
        for (size_t ki = 0; ki < 20 && B >= 2; ++ki) keep_pair(subs_t.data() + (ki * B + (ki < 4 ? 0 : 1)) * 4, (uint32_t)(ki & 3), img.data() + Bpad * bw + ki * 2);
        for (size_t b = 0; b < B; ++b) {
            uint32_t* d = img.data() + b * bw;
            for (size_t cc = 0; cc < 4; ++cc) memcpy(d + cc * qw, P.qual_alias.data() + ((cc * 5) * B + b) * qw, qw * 4);
            for (size_t kk = 0; kk < 64; ++kk)                                       // 3-mer (c0, c1, c2) = table row 20 + 16 c0 + 4 c1 + c2, kept at c0 | c1 << 2 | c2 << 4: the window's own bit order
                keep_pair(subs_t.data() + ((20 + kk) * B + b) * 4, (uint32_t)(kk & 3), d + 4 * qw + (((kk >> 4) & 3) | (kk & 12) | ((kk & 3) << 4)) * 2);
        }
        return img;
    };
    upload(c->t_ring1, ring_image(P.subs1_t), s);
    if (P.have_cdf2) upload(c->t_ring2, ring_image(P.subs2_t), s);
    upload(c->t_ring1u, ring_image_u(P.subs1_t), s);
    if (P.have_cdf2) upload(c->t_ring2u, ring_image_u(P.subs2_t), s);
    HIP_OK(hipStreamSynchronize(s));
    DevTables& t = c->dtb;
    t.L = P.read_length; t.bins = P.bins; t.t_insert = P.t_insert; t.t_delete = P.t_delete; t.t_indel = P.t_indel; t.t_ber = threshold_lt(c->cfg.ber); t.gap_t = c->t_gap.as<uint32_t>(); t.t_kind = P.t_kind;
    t.subs1 = c->t_subs1.as<uint32_t>(); t.subs2 = P.have_cdf2 ? c->t_subs2.as<uint32_t>() : nullptr; t.qual = c->t_qual.as<uint32_t>(); t.qual_alias = c->t_qcompact.as<uint32_t>(); t.qual_k = P.qual_k;
    t.ring1 = c->t_ring1.as<uint4>(); t.ring2 = P.have_cdf2 ? c->t_ring2.as<uint4>() : nullptr;
    t.ring1u = c->t_ring1u.as<uint4>(); t.ring2u = P.have_cdf2 ? c->t_ring2u.as<uint4>() : nullptr;
    t.ins_t = c->t_ins.as<uint32_t>(); t.n_ins = (int)P.ins_t.size(); t.del_t = c->t_del.as<uint32_t>(); t.n_del = (int)P.del_t.size();
    t.isize_t = c->t_isize.as<uint32_t>(); t.n_isize = (int)P.isize_t.size(); t.isize_min = P.isize_min;
    t.subs1_d = c->d_subs1.as<double>(); t.subs2_d = P.have_cdf2 ? c->d_subs2.as<double>() : nullptr; t.qual_d = c->d_qual.as<double>();
    t.ins_d = c->d_ins.as<double>(); t.del_d = c->d_del.as<double>(); t.isize_d = c->d_isize.as<double>();
    t.gc_means = c->d_gcmeans.as<double>(); t.gc_std = P.gc_std;
    // inject_errors keeps 256 read windows + indel events + a 16 KB table ring in one workgroup's LDS, and its bin index
    // is a 32-bit multiply-high (exact while position * bins * length < 2^32)
    if (P.read_length < 4) throw ScsError(SCS_EINVAL, "read length < 4 not supported by the inject_errors kernel");
    if (reads_lds_bytes(t) > 160u * 1024u - 64u) throw ScsError(SCS_EINVAL, "read length too large for the inject_errors kernel (LDS tile)");
    if ((uint64_t)(P.read_length + 128) * (uint64_t)(P.read_length + 128) * (uint64_t)P.bins >= (1ull << 32))
        throw ScsError(SCS_EINVAL, "read length x bin count too large for the inject_errors kernel");
    c->d_tables.reserve(sizeof(DevTables), s);                                    // the table descriptor itself also lives in HBM (kernels fetch fields on use)
    HIP_OK(hipMemcpyAsync(c->d_tables.p, &c->dtb, sizeof(DevTables), hipMemcpyHostToDevice, s)); HIP_OK(hipStreamSynchronize(s));
    c->have_profile = true;
    if (c->cfg.verbose) fprintf(stderr, "profile was loaded from file %s\n", path);
}

// ---------------------------------------------------------------- genome
// d_ascii: the records' ASCII bases already concatenated in device memory (scs_upload_genome_device), or null: host
// records in c->recs[i].code.  The host copies are dropped once the genome is resident (6 GB at whole-genome size).
void index_genome(scs_ctx* c, uint64_t tot);
void stage_genome(scs_ctx* c, const void* d_ascii = nullptr, const uint64_t* d_lens = nullptr) {
    c->rec_off.clear(); c->rec_len.clear(); uint64_t tot = 0;
    for (size_t i = 0; i < c->recs.size(); ++i) { const uint64_t l = d_ascii ? d_lens[i] : c->recs[i].code.size(); c->rec_off.push_back(tot); c->rec_len.push_back(l); tot += l; }
    c->genome_bases = tot;
    c->genome.reserve(std::max<uint64_t>(tot, 16), c->stream);
    if (d_ascii) { if (tot && d_ascii != c->genome.p) HIP_OK(hipMemcpyAsync(c->genome.p, d_ascii, tot, hipMemcpyDeviceToDevice, c->stream)); }   // simuvars builds in place
    else for (size_t i = 0; i < c->recs.size(); ++i)
        if (!c->recs[i].code.empty())
            HIP_OK(hipMemcpyAsync((uint8_t*)c->genome.p + c->rec_off[i], c->recs[i].code.data(), c->recs[i].code.size(), hipMemcpyHostToDevice, c->stream));
    c->sliced = false; c->slice_base = 0; c->slice_len = tot;
    index_genome(c, tot);
    HIP_OK(hipStreamSynchronize(c->stream));
    for (auto& r : c->recs) std::vector<uint8_t>().swap(r.code);
    c->have_genome = true; c->have_frags = false; c->amplified = false; c->allocated = false;
    c->st.records = c->recs.size(); c->st.genome_bases = tot; c->st.staged_bases = tot;
}
// the resident bases (tot of them, raw ASCII in c->genome) -> base codes, bit index, two-bit copy
void index_genome(scs_ctx* c, uint64_t tot) {
    launch_encode_bases(c->stream, c->genome.as<uint8_t>(), tot);                 // raw ASCII -> base codes on the device
    {   // bit index: GC count / any-N of any window in O(1)
        hipStream_t s = c->stream; const uint64_t nw = (tot + 63) / 64;
        c->gx_gc_bits.reserve((nw + 1) * 8, s); c->gx_n_bits.reserve((nw + 1) * 8, s); c->gx_gc_cnt.reserve((nw + 2) * 4, s); c->gx_n_cnt.reserve((nw + 2) * 4, s);
        c->gx_gc_pref.reserve((nw + 2) * 8, s); c->gx_n_pref.reserve((nw + 2) * 8, s); c->scan_tmp.reserve(scan_temp_bytes(nw + 1), s);
        c->genome2.reserve((nw + 1) * 16 + 256, s);                               // two bits per base, 64 bytes of slack in front and 192 behind (the window gather over-reads by up to a dozen words)
        launch_genome_bits(s, c->genome.as<uint8_t>(), tot, nw, c->gx_gc_bits.as<unsigned long long>(), c->gx_n_bits.as<unsigned long long>(), c->gx_gc_cnt.as<uint32_t>(),
                           c->gx_n_cnt.as<uint32_t>(), c->gx_gc_pref.as<uint64_t>(), c->gx_n_pref.as<uint64_t>(), c->scan_tmp.p, c->scan_tmp.cap, c->genome2.as<uint32_t>() + 16);
    }
}

// Genome::loadRefSeq for whole-genome inputs: the FASTA is mmap'ed and its RAW bytes go to the device in
// 64 MB chunks through two pinned buffers (four host threads copy a chunk out of the page cache while the GPU works on the
// one before); the device separates bases from line ends, headers and comments (k_fa_*), compacts them into the genome
// buffer and lists the headers; the host only reads the header lines.  Then encode + bit index as for every genome.
void stage_fasta_on_device(scs_ctx* c, const std::string& path_in) {
    const std::string path = fasta_plain_path(path_in);
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw ScsError(SCS_EIO, "could not open " + path);
    struct stat st_;
    if (fstat(fd, &st_) != 0) { close(fd); throw ScsError(SCS_EIO, "could not stat " + path); }
    const size_t size = (size_t)st_.st_size;
    if (size == 0) { close(fd); throw ScsError(SCS_EIO, "ERROR: reference sequence cannot be empty!"); }
    const char* base = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) { close(fd); throw ScsError(SCS_EIO, "could not map " + path); }
    (void)madvise((void*)base, size, MADV_SEQUENTIAL);
    struct Unmap { const char* b; size_t n; int fd; ~Unmap() { munmap((void*)b, n); close(fd); } } unmap{base, size, fd};
    hipStream_t s = c->stream;
    const size_t CH = 64u << 20; const uint32_t hdr_cap = 1u << 20;
    DevBuf d_raw[2], d_kind, d_keep, d_pos, d_st, d_hdr, d_tmp; char* h_raw[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool ev_used[2] = {false, false};
    struct Rel { DevBuf* b[8]; char** h; hipEvent_t* e; ~Rel() { for (DevBuf* x : b) x->release(); for (int k = 0; k < 2; ++k) { if (h[k]) (void)hipHostFree(h[k]); if (e[k]) (void)hipEventDestroy(e[k]); } } }
        rel{{&d_raw[0], &d_raw[1], &d_kind, &d_keep, &d_pos, &d_st, &d_hdr, &d_tmp}, h_raw, ev};
    const size_t ch = std::min(CH, size);
    for (int k = 0; k < 2; ++k) { d_raw[k].reserve(ch + 16, s); HIP_OK(hipHostMalloc((void**)&h_raw[k], ch, hipHostMallocDefault)); HIP_OK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
    d_kind.reserve(ch + 16, s); d_keep.reserve((ch + 2) * 4, s); d_pos.reserve((ch + 2) * 4, s); d_st.reserve(64, s); d_hdr.reserve((size_t)hdr_cap * 16, s);
    d_tmp.reserve(fasta_chunk_temp_bytes((uint32_t)ch), s);
    HIP_OK(hipMemsetAsync(d_st.p, 0, 64, s));
    c->genome.reserve(size + 16, s);                                              // the bases are fewer than the file's bytes
    for (size_t off = 0, k = 0; off < size; off += ch, ++k) {
        const int b = (int)(k & 1); const size_t n = std::min(ch, size - off);
        if (ev_used[b]) HIP_OK(hipEventSynchronize(ev[b]));                       // the pinned buffer's last upload is done
        {   // page cache -> pinned, four slices in parallel
            std::vector<std::thread> th; const size_t parts = n >= (8u << 20) ? 4 : 1, per = (n + parts - 1) / parts;
            for (size_t q = 1; q < parts; ++q) th.emplace_back([&, q] { const size_t o = q * per; if (o < n) memcpy(h_raw[b] + o, base + off + o, std::min(per, n - o)); });
            memcpy(h_raw[b], base + off, std::min(per, n));
            for (auto& t : th) t.join();
        }
        HIP_OK(hipMemcpyAsync(d_raw[b].p, h_raw[b], n, hipMemcpyHostToDevice, s));
        HIP_OK(hipEventRecord(ev[b], s)); ev_used[b] = true;
        launch_fasta_chunk(s, d_raw[b].as<uint8_t>(), (uint32_t)n, (unsigned long long)off, d_st.as<unsigned long long>(), d_kind.as<uint8_t>(), d_keep.as<uint32_t>(), d_pos.as<uint32_t>(),
                           c->genome.as<uint8_t>(), d_hdr.as<unsigned long long>(), hdr_cap, d_tmp.p, d_tmp.cap);
    }
    unsigned long long stv[3] = {0, 0, 0};
    HIP_OK(hipMemcpyAsync(stv, d_st.p, 24, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
    { const hipError_t le = take_launch_error(); if (le != hipSuccess) throw ScsError(SCS_EDEVICE, std::string("FASTA staging kernels: ") + hipGetErrorString(le)); }
    const uint64_t total = stv[0], nh = stv[1];
    if (nh > hdr_cap) throw ScsError(SCS_EOVERFLOW, "more than 2^20 FASTA records");
    if (nh == 0) throw ScsError(SCS_EIO, total ? "malformed FASTA (sequence before header): " + path : std::string("ERROR: reference sequence cannot be empty!"));
    std::vector<unsigned long long> hp(2 * nh);
    HIP_OK(hipMemcpyAsync(hp.data(), d_hdr.p, hp.size() * 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
    std::vector<std::pair<uint64_t, uint64_t>> hs(nh);
    for (uint64_t k = 0; k < nh; ++k) hs[k] = {hp[2 * k], hp[2 * k + 1]};
    std::sort(hs.begin(), hs.end());                                               // by file offset (the list is filled by atomics)
    if (hs[0].second != 0) throw ScsError(SCS_EIO, "malformed FASTA (sequence before header): " + path);
    std::vector<uint64_t> hoff(nh), lens(nh);
    c->recs.assign(nh, FastaRecord());
    for (uint64_t k = 0; k < nh; ++k) {
        hoff[k] = hs[k].first; lens[k] = (k + 1 < nh ? hs[k + 1].second : total) - hs[k].second;
        const char* nl = (const char*)memchr(base + hoff[k], '\n', size - hoff[k]);
        size_t hend = nl ? (size_t)(nl - base) : size; if (hend > hoff[k] && base[hend - 1] == '\r') --hend;
        c->recs[k].name = fasta_index_name(std::string(base + hoff[k] + 1, base + hend));
    }
    fasta_write_fai(path, base, size, hoff, lens);                                 // fastahack leaves <file>.fai beside its input (Fasta.cpp:241-249)
    stage_genome(c, c->genome.p, lens.data());
}

// ---------------------------------------------------------------- a1: Genome::splitToFrags (Genome.cpp:753-782)
// the whole job's fragment list (genome coordinates) and this shard's contiguous range [lo, hi) of it, balanced by bases
void split_frags(scs_ctx* c, std::vector<uint64_t>& goff, std::vector<uint32_t>& len, std::vector<int8_t>& strand, size_t& lo, size_t& hi) {
    const scs_config& cf = c->cfg;
    goff.clear(); len.clear(); strand.clear();
    for (size_t r = 0; r < c->recs.size(); ++r) {
        const int64_t chr_len = (int64_t)c->rec_len[r]; int64_t pos = 1; uint32_t k = 0;
        while (pos <= chr_len) {
            const U4 d = draw4(c->key, ST_FRAGSPLIT, 0, r, k++);
            const int64_t fl = scale_draw(d.w[0], (uint32_t)cf.frag_min, (uint32_t)(cf.frag_max + 1 - cf.frag_min));   // randomInteger(minSize, maxSize+1)
            if (pos + fl - 1 > chr_len) break;
            for (int sgn : {1, -1}) { goff.push_back(c->rec_off[r] + (uint64_t)(pos - 1)); len.push_back((uint32_t)fl); strand.push_back((int8_t)sgn); }
            pos += fl;
        }
        if (pos <= chr_len)                                                     // tail: emitted twice, both strand +1 (Genome.cpp:772-777)
            for (int rep = 0; rep < 2; ++rep) { goff.push_back(c->rec_off[r] + (uint64_t)(pos - 1)); len.push_back((uint32_t)(chr_len - pos + 1)); strand.push_back(1); }
    }
    // fragment-lineage sharding: contiguous fragment ranges balanced by bases
    lo = 0; hi = len.size();
    if (cf.shard_count > 1) {
        uint64_t tot = 0; for (auto l : len) tot += l;
        std::vector<size_t> cut(cf.shard_count + 1, len.size()); cut[0] = 0;
        uint64_t acc = 0; int sh = 1;
        for (size_t i = 0; i < len.size() && sh < cf.shard_count; ++i) { acc += len[i]; while (sh < cf.shard_count && acc * cf.shard_count >= tot * (uint64_t)sh) cut[sh++] = i + 1; }
        lo = cut[cf.shard_rank]; hi = cut[cf.shard_rank + 1];
    }
}

// Sharded job, regular FASTA with an index beside it ("genome slices needed per GPU = its own fragments only";
// lib/genome/Genome.cpp:753-782 splits by record length alone): the record lengths come from the .fai, the fragment split from
// them, and only the byte ranges of THIS shard's fragments are read, uploaded, stripped of their line ends (the .fai's line
// geometry), encoded and indexed.  Returns false when the file has no usable index (absent, older than the file, or lines
// that are not what it states: a ragged file) -- the caller then stages the whole file, which also writes the index.
bool stage_fasta_slice(scs_ctx* c, const std::string& path) {
    struct stat sf, si;
    const std::string fai = path + ".fai";
    if (stat(path.c_str(), &sf) != 0 || stat(fai.c_str(), &si) != 0 || si.st_mtime < sf.st_mtime) return false;
    struct Ent { std::string name; uint64_t len, off; uint32_t lb, lw; };
    std::vector<Ent> ents;
    {   FILE* f = fopen(fai.c_str(), "r"); if (!f) return false;
        char line[4096];
        while (fgets(line, sizeof line, f)) {
            char nm[2048]; unsigned long long l, o; unsigned lb, lw;
            if (sscanf(line, "%2047s %llu %llu %u %u", nm, &l, &o, &lb, &lw) != 5) { fclose(f); return false; }
            ents.push_back(Ent{fasta_index_name(nm), l, o, lb, lw});
        }
        fclose(f); }
    if (ents.empty()) return false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct Close { int fd; ~Close() { close(fd); } } closer{fd};
    const uint64_t size = (uint64_t)sf.st_size;
    auto byte_of = [](const Ent& e, uint64_t b) { return e.off + (e.lb ? b / e.lb * e.lw + b % e.lb : 0); };   // file offset of base b of the record
    // the index must describe THIS file: every record's header and last line end where the geometry puts them
    for (size_t r = 0; r < ents.size(); ++r) {
        const Ent& e = ents[r];
        if (e.len && (e.lb == 0 || e.lw <= e.lb || e.lw - e.lb > 2)) return false;
        const uint64_t end = e.len ? byte_of(e, e.len - 1) + 1 : e.off;            // one past the record's last base
        char b[4] = {0, 0, 0, 0};
        if (e.off == 0 || e.off > size || end > size) return false;
        if (pread(fd, b, 1, (off_t)(e.off - 1)) != 1 || b[0] != '\n') return false;   // the header line ends right before the first base
        if (end < size) {                                                           // then a line end, then the next header or the end of the file
            const ssize_t got = pread(fd, b, 3, (off_t)end);
            int k = 0; if (got > k && b[k] == '\r') ++k; if (!(got > k && b[k] == '\n')) return false; ++k;
            const uint64_t next = end + (uint64_t)k;
            if (r + 1 < ents.size()) { if (next >= size || (got > k ? b[k] : 0) != '>') return false; }
            else if (next != size) return false;
        } else if (r + 1 < ents.size()) return false;
    }
    // (the fragment split reads the records from the ctx: what was there comes back if this staging gives up below)
    struct Keep { scs_ctx* c; std::vector<FastaRecord> recs; std::vector<uint64_t> off, len; uint64_t bases; bool done = false;
                  ~Keep() { if (!done) { c->recs.swap(recs); c->rec_off.swap(off); c->rec_len.swap(len); c->genome_bases = bases; } } } keep{c, c->recs, c->rec_off, c->rec_len, c->genome_bases};
    c->recs.assign(ents.size(), FastaRecord()); c->rec_off.clear(); c->rec_len.clear(); uint64_t tot = 0;
    for (size_t r = 0; r < ents.size(); ++r) { c->recs[r].name = ents[r].name; c->rec_off.push_back(tot); c->rec_len.push_back(ents[r].len); tot += ents[r].len; }
    c->genome_bases = tot;
    std::vector<uint64_t> goff; std::vector<uint32_t> len; std::vector<int8_t> strand; size_t lo, hi;
    split_frags(c, goff, len, strand, lo, hi);
    uint64_t g_lo = 0, g_hi = 0;
    if (hi > lo) { g_lo = goff[lo]; for (size_t i = lo; i < hi; ++i) g_hi = std::max(g_hi, goff[i] + len[i]); }
    hipStream_t s = c->stream;
    const uint64_t n_slice = g_hi - g_lo;
    if (n_slice == 0) return false;                                                 // (more shards than fragments: nothing of its own to stage)
    c->genome.reserve(std::max<uint64_t>(n_slice, 16), s);
    DevBuf d_ragged; struct RelR { DevBuf* b; ~RelR() { b->release(); } } relr{&d_ragged};
    d_ragged.reserve(16, s); HIP_OK(hipMemsetAsync(d_ragged.p, 0, 4, s));
    // record by record: the bytes of [a, b) -> pinned -> device, line ends dropped by the gather
    const size_t CH = 64u << 20; DevBuf d_raw; char* h_raw = nullptr;
    struct Rel { DevBuf* b; char** h; ~Rel() { b->release(); if (*h) (void)hipHostFree(*h); } } rel{&d_raw, &h_raw};
    HIP_OK(hipHostMalloc((void**)&h_raw, CH, hipHostMallocDefault)); d_raw.reserve(CH + 16, s);
    for (size_t r = 0; r < ents.size() && n_slice; ++r) {
        if (ents[r].len == 0) continue;                                             // an empty record (index line "name 0 off 0 0"): nothing to read, no line geometry
        const uint64_t r0 = c->rec_off[r], r1 = r0 + ents[r].len;
        uint64_t a = std::max(g_lo, r0), b = std::min(g_hi, r1);
        const uint64_t per = (uint64_t)(CH / ents[r].lw) * ents[r].lb;              // bases whose lines fit the buffer (two lines of slack: a piece starts and ends inside a line)
        while (a < b) {
            const uint64_t take = std::min<uint64_t>(b - a, per > 2ull * ents[r].lb ? per - 2ull * ents[r].lb : 1), ba = a - r0;
            const uint64_t f0 = byte_of(ents[r], ba), f1 = byte_of(ents[r], ba + take - 1) + 1;
            HIP_OK(hipStreamSynchronize(s));                                         // the pinned buffer's last upload is done
            if (pread(fd, h_raw, (size_t)(f1 - f0), (off_t)f0) != (ssize_t)(f1 - f0)) throw ScsError(SCS_EIO, "could not read " + path);
            HIP_OK(hipMemcpyAsync(d_raw.p, h_raw, (size_t)(f1 - f0), hipMemcpyHostToDevice, s));
            launch_fa_gather_regular(s, d_raw.as<uint8_t>(), c->genome.as<uint8_t>() + (a - g_lo), take, (uint32_t)(ba % ents[r].lb), ents[r].lb, ents[r].lw, d_ragged.as<uint32_t>());
            a += take;
        }
    }
    {   // a line end or a '>' among the bases: the lines are not what the index says (ragged lines that cancel out, a blank line, a
        // file rewritten within the index's second) -- not this file's index: the whole file is staged by the parser instead
        uint32_t ragged = 0; HIP_OK(hipMemcpyAsync(&ragged, d_ragged.p, 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
        if (ragged) return false; }
    keep.done = true;
    index_genome(c, n_slice);
    HIP_OK(hipStreamSynchronize(s));
    { const hipError_t le = take_launch_error(); if (le != hipSuccess) throw ScsError(SCS_EDEVICE, std::string("FASTA slice staging: ") + hipGetErrorString(le)); }
    c->sliced = true; c->slice_base = g_lo; c->slice_len = n_slice;
    c->have_genome = true; c->have_frags = false; c->amplified = false; c->allocated = false;
    c->st.records = c->recs.size(); c->st.genome_bases = tot; c->st.staged_bases = n_slice;
    return true;
}

void do_create_frags(scs_ctx* c) {
    if (!c->have_genome) throw ScsError(SCS_EINVAL, "scs_create_frags: no genome loaded");
    std::vector<uint64_t> goff; std::vector<uint32_t> len; std::vector<int8_t> strand; size_t lo, hi;
    split_frags(c, goff, len, strand, lo, hi);
    if (c->sliced) {   // only this shard's bases are resident: the split (a function of the seed) must still ask for them
        for (size_t i = lo; i < hi; ++i)
            if (goff[i] < c->slice_base || goff[i] + len[i] > c->slice_base + c->slice_len)
                throw ScsError(SCS_EINVAL, "the genome was staged for another seed's fragment split (sharded staging): load it again after scs_set_seed");
        for (size_t i = lo; i < hi; ++i) goff[i] -= c->slice_base;
    }
    c->nf_all = len.size(); c->frag_len_all = 0; for (auto l : len) c->frag_len_all += l;
    c->f_goff.assign(goff.begin() + lo, goff.begin() + hi); c->f_len.assign(len.begin() + lo, len.begin() + hi);
    c->f_strand.assign(strand.begin() + lo, strand.begin() + hi); c->f_primers.assign(hi - lo, 0); c->f_gidx_base = lo;
    // one asynchronous copy from a pinned staging block (offsets | lengths | strands); the stream orders it before the kernels
    // that read it, and the block is not rewritten before that copy is done (frag_copy_pending, cleared by the next host wait)
    {
        const size_t nfr = c->f_len.size(), o_len = nfr * 8, o_str = o_len + nfr * 4, bytes = std::max<size_t>(o_str + nfr, 16);
        if (c->frag_copy_pending) { HIP_OK(hipStreamSynchronize(c->stream)); c->frag_copy_pending = false; }
        if (bytes > c->h_frag_cap) {
            if (c->h_frag) HIP_OK(hipHostFree(c->h_frag));
            c->h_frag_cap = bytes + bytes / 2; HIP_OK(hipHostMalloc((void**)&c->h_frag, c->h_frag_cap, hipHostMallocDefault));
        }
        if (nfr) { memcpy(c->h_frag, c->f_goff.data(), nfr * 8); memcpy(c->h_frag + o_len, c->f_len.data(), nfr * 4); memcpy(c->h_frag + o_str, c->f_strand.data(), nfr); }
        c->df_blob.reserve(bytes, c->stream);
        if (nfr) { HIP_OK(hipMemcpyAsync(c->df_blob.p, c->h_frag, o_str + nfr, hipMemcpyHostToDevice, c->stream)); c->frag_copy_pending = true; }
        c->df_len_off = o_len; c->df_strand_off = o_str;
    }
    c->df_primers.reserve(std::max<size_t>(c->f_len.size() * 4, 16), c->stream);
    c->df_hasn.reserve(std::max<size_t>(c->f_len.size(), 16), c->stream);
    {   const DevGenomeIdx gx{c->gx_gc_bits.as<unsigned long long>(), c->gx_n_bits.as<unsigned long long>(), c->gx_gc_pref.as<uint64_t>(), c->gx_n_pref.as<uint64_t>()};
        const DevFrags fv = c->frags_view();
        launch_frag_has_n(c->stream, fv.goff, fv.len, fv.n, gx, c->df_hasn.as<uint8_t>()); }
    c->have_frags = true; c->amplified = false; c->allocated = false;
    c->st.fragments = c->f_len.size();
}

// ---------------------------------------------------------------- a3: Malbac::setPrimers (Malbac.cpp:236-283) on the device
// One launch gives every template (fragments, then all semis so far) its Poisson budget; the scans
// turn budgets into slot offsets.  One host sync: the sums feed totalPrimers and the buffer sizes.
// ns_cap: upper bound of the semi amplicon count (the count itself is on the device: the passes that made the newest
// semis have not been read back yet -- their counts arrive with this call's mail, ONE wait per cycle).
void set_primers_launch(scs_ctx* c, bool only_frags, uint32_t call, uint32_t ns_cap) {
    hipStream_t s = c->stream;
    const uint32_t nf = (uint32_t)c->f_len.size(), ns = only_frags ? 0u : ns_cap;
    PoissonParams p; p.key = c->key; p.call = call; p.gamma = c->cfg.gamma; p.total_primers = c->total_primers;
    p.nf = nf; p.frag_len = c->frag_total_len; p.dev = c->dsums.as<unsigned long long>(); p.totals = nullptr; p.total_primers_dev = nullptr;
    if (c->sharded()) {
        // whole-job {templateNum, totalLen} and the pool size are device scalars, kept current by the tail of the per-pass
        // primer all-reduce (launch_pass): no collective of its own here
        p.totals = c->dsums.as<uint64_t>() + DS_G_TOTALS; p.total_primers_dev = c->dsums.as<unsigned long long>() + DS_G_PRIMERS;
    }
    c->budget_f.reserve(((size_t)nf + 1) * 4, s); c->budget_s.reserve(((size_t)ns + 2) * 4, s);
    c->slot_off_f.reserve(((size_t)nf + 1) * 4, s); c->slot_off_s.reserve(((size_t)ns + 2) * 4, s);
    c->scan_tmp.reserve(scan_temp_bytes(std::max(nf, ns)), s);
    // sums[0..1] are zero here: the previous call's mail cleared them after reading (k_amplify_init zeroes them first)
    c->poisson_part.reserve(((size_t)nf + (size_t)ns / 256 + 4) * 8, s);
    launch_poisson(s, c->frags_view(), c->semis.view(), ns, p, c->budget_f.as<uint32_t>(), c->budget_s.as<uint32_t>(), c->dsums.as<unsigned long long>(), c->poisson_part.as<unsigned long long>());
    exclusive_scan_u32_pair(s, c->budget_f.as<uint32_t>(), c->slot_off_f.as<uint32_t>(), nf, ns ? c->budget_s.as<uint32_t>() : nullptr, c->slot_off_s.as<uint32_t>(), ns, c->scan_tmp.p, c->scan_tmp.cap);
    const bool sh = c->sharded();                                                  // sharded: the budget sums ride on the next pass's all-reduce (and are cleared there)
    c->budgets_pending = sh;
    Mail& m = c->pend;                                                             // together with the counts of the passes before (collect_post)
    m.add(c->dsums.p, 8, 0, !sh); m.add(c->dsums.as<unsigned long long>() + 1, 8, 1, !sh); m.add(c->slot_off_f.as<uint32_t>() + nf, 4, 2);
    m.add(ns ? (const void*)(c->slot_off_s.as<uint32_t>() + ns) : nullptr, 4, 3);  // budgets beyond the real count are 0: the total sits at [ns_cap] too
    mail_post(c, m, true); c->pend = Mail();
}
void set_primers_finish(scs_ctx* c) {                                              // after mail_wait (and collect_read: semis.n is current)
    const uint64_t* rb = c->h_rb;
    if (!c->sharded()) c->total_primers -= rb[0] + rb[1];                          // sharded: the whole-job pool size comes back with collect_read
    c->slots_f = (uint32_t)rb[2]; c->slots_s = (uint32_t)rb[3]; c->budget_ns = c->semis.n;
}

// sharded job, end of a pass: what the shards owe each other besides the primer stock (new semi amplicons of a fragment pass,
// the budgets of the last setPrimers) is summed by a small all-reduce behind the stock counters; the update takes what the
// pass took from the stock (summed over the shards by attach_pass) and folds the rest into the device scalars the next
// setPrimers reads.
void shard_close(scs_ctx* c, const uint32_t* new_semis) {
    hipStream_t s = c->stream;
    const int wb = c->budgets_pending ? 1 : 0;
    launch_shard_tail(s, c->primer_gdelta.as<uint32_t>(), c->dsums.as<unsigned long long>(), new_semis, wb);
    c->budgets_pending = false;
    c->reduce_dev(c->primer_gdelta.as<uint32_t>() + 65536, SHARD_TAIL_WORDS, 4);
    launch_primer_update_sharded(s, c->primer_cnt.as<int64_t>(), c->primer_gdelta.as<uint32_t>(), c->primer_delta.as<uint32_t>(), c->primer_cut.as<unsigned long long>(),
                                 c->dsums.as<unsigned long long>(), c->flags.as<uint32_t>(), wb);
}

// ---------------------------------------------------------------- a2: the primer stock, exactly (Malbac::updatePrimerCount, Malbac.cpp:91-103)
// The kernels and the argument are in scs_kernels.hip ("the primer stock, exactly").  Here: the loop.
static void attach_range(scs_ctx* c, bool from_frag, const AmplifyParams& p, uint32_t lo, uint32_t hi, int undo, const unsigned long long* t_from) {
    hipStream_t s = c->stream;
    const uint32_t* slot_off = (from_frag ? c->slot_off_f : c->slot_off_s).as<uint32_t>();
    DevBuf& valid = from_frag ? c->valid_f : c->valid; DevBuf& slots = from_frag ? c->slots_fr : c->slots; DevBuf& slot_tmpl = from_frag ? c->slot_tmpl_fr : c->slot_tmpl;
    DevFrags fr = c->frags_view(); fr.primers = c->budget_f.as<uint32_t>();
    if (from_frag) launch_attach_frags(s, c->genome.as<uint8_t>(), fr, slot_off, slots.as<uint32_t>(), slot_tmpl.as<uint32_t>(), valid.as<uint32_t>(),
                                       c->primer_cut.as<unsigned long long>(), c->primer_delta.as<uint32_t>(), c->poisson_part.as<unsigned long long>(), p, lo, hi, undo, t_from);
    else if (lo == 0 && hi == c->budget_ns && c->slots_s && !seam_env("SCS_ATTACH_GROUPS")) {
        // the whole pass: the dense form (one lane = one primer, scs_k_amplify.hip); its plan is made with the pass's first run
        if (!undo) {
            c->att_wave_first.reserve(((size_t)attach_dense_waves(c->slots_s) + 2) * 4, s);
            launch_attach_plan(s, slot_off, hi, c->slots_s, slot_tmpl.as<uint32_t>(), c->att_wave_first.as<uint32_t>(), valid.as<uint32_t>());
        }
        launch_attach_dense(s, c->genome.as<uint8_t>(), fr, c->semis.view(), c->semis.pool_view(), slot_off, slots.as<uint32_t>(), slot_tmpl.as<uint32_t>(), c->att_wave_first.as<uint32_t>(),
                            c->slots_s, valid.as<uint32_t>(), c->primer_cut.as<unsigned long long>(), c->primer_delta.as<uint32_t>(), p, undo, t_from);
    }
    else launch_attach_semis(s, c->genome.as<uint8_t>(), fr, c->semis.view(), c->budget_ns, c->semis.pool_view(), slot_off, slots.as<uint32_t>(), slot_tmpl.as<uint32_t>(),
                             valid.as<uint32_t>(), c->primer_cut.as<unsigned long long>(), c->primer_delta.as<uint32_t>(), p, lo, hi, undo, t_from);   // a range of the list (a sharded pass run again segment by segment), or SCS_ATTACH_GROUPS: a lane group per template
}
// The templates [lo, hi) of a pass have been run against the cuts as they stand, primer_delta = what they took, primer_cnt = the
// stock they started from.  Until no type is over its stock (and no cut type under it): cut the over-demanded types at their
// stock-th attachment in list order, run the templates behind the earliest new cut again.  One host wait per round.
static void exact_stock(scs_ctx* c, bool from_frag, const AmplifyParams& p, uint32_t lo, uint32_t hi, const uint32_t* taken) {
    hipStream_t s = c->stream;
    c->st_eidx.reserve(65536 * 4, s); c->st_etype.reserve(65536 * 4, s); c->st_estart.reserve(65536 * 4, s); c->st_info.reserve(64, s);
    unsigned long long* info = c->st_info.as<unsigned long long>();
    const uint32_t* slot_off = (from_frag ? c->slot_off_f : c->slot_off_s).as<uint32_t>();
    DevBuf& valid = from_frag ? c->valid_f : c->valid; DevBuf& slots = from_frag ? c->slots_fr : c->slots;
    for (int round = 0;; ++round) {
        launch_stock_check(s, c->primer_cnt.as<int64_t>(), taken, c->primer_cut.as<unsigned long long>(), from_frag, info);
        Mail m; m.add(info, 8, 24); m.add(info + 1, 8, 25); m.add(info + 2, 8, 26); mail_post(c, m, true); mail_wait(c);
        const uint64_t n_over = c->h_rb[24], n_att = c->h_rb[25], n_under = c->h_rb[26];
        if (round == 0) { c->st.stock_checks++; if (n_over) c->st.stock_exhausted_passes++; }
        if (!n_over && !n_under) return;
        if (taken != c->primer_delta.as<uint32_t>()) return;                       // a sharded job's first look at the pass (all shards' demand): attach_pass takes over
        if (round >= 500) throw ScsError(SCS_EOVERFLOW, "internal: the primer stock of a pass did not settle");
        launch_stock_list(s, c->primer_cnt.as<int64_t>(), taken, c->st_eidx.as<uint32_t>(), c->st_etype.as<uint32_t>(), c->st_estart.as<uint32_t>(), info);
        if (n_over) {
            c->st_list.reserve(n_att * 8 + 64, s); c->st_sorted.reserve(n_att * 8 + 64, s); c->st_tmp.reserve(stock_sort_temp_bytes(n_att), s);
            launch_stock_collect(s, c->genome.as<uint8_t>(), c->frags_view(), c->semis.view(), c->semis.pool_view(), from_frag, slot_off, slots.as<uint32_t>(), valid.as<uint32_t>(),
                                 c->st_eidx.as<uint32_t>(), c->st_list.as<unsigned long long>(), info, lo, hi);
            launch_stock_sort(s, c->st_list.as<unsigned long long>(), c->st_sorted.as<unsigned long long>(), n_att, c->st_tmp.p, c->st_tmp.cap);
            launch_stock_pick(s, c->primer_cnt.as<int64_t>(), c->st_etype.as<uint32_t>(), c->st_estart.as<uint32_t>(), (uint32_t)n_over, c->st_sorted.as<unsigned long long>(),
                              c->primer_cut.as<unsigned long long>(), from_frag, info);
        }
        attach_range(c, from_frag, p, lo, hi, 1, info + 6);                        // info[6]: the first template behind a moved cut (the kernel skips the others)
        c->st.stock_rounds++;
    }
}
// One pass's attachments over this shard's templates [0, nt), exact.  Unsharded: one run; the over-demand check (a host wait)
// only when the pass has more primers to place than the smallest stock in use.  Sharded: one run, the shards' demand summed; if
// a type is over its stock the pass is run again segment by segment in the whole job's list order, every segment by its owner
// against the stock the segments before it left (handed on by an all-reduce to which only the owner contributes).
static void attach_pass(scs_ctx* c, bool from_frag, const AmplifyParams& p, uint32_t nt, uint32_t n_slots) {
    hipStream_t s = c->stream;
    const bool some = nt != 0 && n_slots != 0;
    if (some) attach_range(c, from_frag, p, 0, nt, 0, nullptr);
    if (!c->sharded()) {
        if (some && (uint64_t)n_slots > c->min_stock_lb) exact_stock(c, from_frag, p, 0, nt, c->primer_delta.as<uint32_t>());
        c->min_stock_lb = c->min_stock_lb > n_slots ? c->min_stock_lb - n_slots : 0;
        return;
    }
    uint32_t* delta = c->primer_delta.as<uint32_t>(); uint32_t* gdelta = c->primer_gdelta.as<uint32_t>();
    HIP_OK(hipMemcpyAsync(gdelta, delta, 65536 * 4, hipMemcpyDeviceToDevice, s));
    c->reduce_dev(gdelta, 65536, 4);
    const uint64_t before = c->st.stock_exhausted_passes;
    exact_stock(c, from_frag, p, 0, nt, gdelta);                                   // the check alone: taken != primer_delta
    if (c->st.stock_exhausted_passes == before) return;                            // gdelta = what the pass took, all shards: applied by shard_close
    HIP_OK(hipMemsetAsync(delta, 0, 65536 * 4, s)); HIP_OK(hipMemsetAsync(gdelta, 0, 65536 * 4, s));
    std::vector<std::pair<uint32_t, uint32_t>> segs;                               // local template ranges, in list order
    if (from_frag) segs.push_back({0u, nt});
    else for (size_t b = 0; b < c->semi_block_end.size(); ++b) segs.push_back({b ? c->semi_block_end[b - 1] : 0u, std::min(c->semi_block_end[b], nt)});
    const int R = c->cfg.shard_count;
    for (auto& sg : segs) for (int k = 0; k < R; ++k) {
        // fragments ascend with the shard; the semis of a fragment pass lie in the list with their fragments DEscending
        const int owner = from_frag ? k : R - 1 - k;
        if (owner == c->cfg.shard_rank && sg.second > sg.first && n_slots) {
            attach_range(c, from_frag, p, sg.first, sg.second, 0, nullptr);
            exact_stock(c, from_frag, p, sg.first, sg.second, delta);
            HIP_OK(hipMemcpyAsync(gdelta, delta, 65536 * 4, hipMemcpyDeviceToDevice, s));
        }
        c->reduce_dev(gdelta, 65536, 4);
        launch_stock_apply(s, c->primer_cnt.as<int64_t>(), gdelta, delta, c->primer_cut.as<unsigned long long>(), c->flags.as<uint32_t>());
    }
}

// ---------------------------------------------------------------- one amplification pass (a4 / a5)
// rb_slot: where the number of amplicons created is read back to (pinned host memory, stream-ordered).
static void join_errs(scs_ctx* c) { if (c->errs_pending) { HIP_OK(hipStreamWaitEvent(c->stream, c->ev_errs, 0)); c->errs_pending = false; } }
void launch_pass(scs_ctx* c, bool from_frag, uint32_t pass, int rb_slot) {
    hipStream_t s = c->stream;
    const uint32_t nt = from_frag ? (uint32_t)c->f_len.size() : c->budget_ns;
    const uint32_t n_slots = from_frag ? c->slots_f : c->slots_s;
    const bool some = nt != 0 && n_slots != 0;                                     // a shard with nothing local still joins the pass's collectives
    const uint32_t* slot_off = (from_frag ? c->slot_off_f : c->slot_off_s).as<uint32_t>();
    // the two passes of a group keep their own count arrays: their totals are mailed together at the group's collect
    DevBuf& valid = from_frag ? c->valid_f : c->valid; DevBuf& valid_off = from_frag ? c->valid_off_f : c->valid_off;
    DevBuf& slots = from_frag ? c->slots_fr : c->slots; DevBuf& slot_tmpl = from_frag ? c->slot_tmpl_fr : c->slot_tmpl;
    AmpStore& out = from_frag ? c->semis : c->fulls;
    AmplifyParams p; p.key = c->key; p.pass = pass; p.amp_min = (uint32_t)c->cfg.amplicon_min_len; p.amp_max = (uint32_t)c->cfg.amplicon_max_len; p.t_ber = c->dtb.t_ber;
    if (some) {
        valid.reserve(((size_t)nt + 1) * 4, s); valid_off.reserve(((size_t)nt + 1) * 4, s);
        slots.reserve((size_t)n_slots * 4, s); slot_tmpl.reserve((size_t)n_slots * 4, s);   // k_attach marks its own slots unused first
        c->scan_tmp.reserve(scan_temp_bytes(nt), s);
        out.reserve((uint64_t)out.n + n_slots, s);
        out.reserve_pool(std::max<uint32_t>(1u << 16, (uint32_t)std::min<uint64_t>(((uint64_t)out.n + n_slots) / 256 + 4096, 0xFFFFFFF0ull)), s);
    }
    KernelTimer& tma = from_frag ? c->tm_attach_f : c->tm_attach;
    if (some) tma.begin(s);
    attach_pass(c, from_frag, p, nt, n_slots);
    if (some) { tma.end(s); tma.add_units(nt); }
    if (!some) {
        if (c->sharded()) shard_close(c, nullptr);
        c->pend.add(nullptr, 8, rb_slot);
        if (!from_frag) { for (int b = 0; b < 8; ++b) c->pend.add(nullptr, 8, 16 + b); c->pending_seg_cycle = (int)pass; }
        return;
    }
    DevFrags fr = c->frags_view(); fr.primers = c->budget_f.as<uint32_t>();
    const uint8_t* g = c->genome.as<uint8_t>();
    if (from_frag) launch_frag_len_sum(s, c->poisson_part.as<unsigned long long>(), nt, c->dsums.as<unsigned long long>() + DS_SEMI_LEN);
    exclusive_scan_u32(s, valid.as<uint32_t>(), valid_off.as<uint32_t>(), nt, c->scan_tmp.p, c->scan_tmp.cap);
    KernelTimer& tm = from_frag ? c->tm_errscan_f : c->tm_errscan;
    // the stock update rides on k_errs (launched with at least 256 workgroups: one primer type per thread); a sharded job
    // closes the pass with shard_close
    const bool ride = !c->sharded();
    // k_errs<semi->full> writes only the new full amplicons, which nothing reads before the allocation: it runs on its own
    // stream beside the fragment pass that follows (its chain of dependent gathers beside the attach kernel's ALU work); the
    // stock update it used to carry runs on the main stream.  Joined before the next setPrimers rewrites the slot offsets.
    hipStream_t es = s;
    if (!from_frag && !seam_env("SCS_ERRS_INLINE")) {
        if (!c->errs_stream) {
            HIP_OK(hipStreamCreateWithFlags(&c->errs_stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&c->ev_att, hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&c->ev_errs, hipEventDisableTiming));
        }
        HIP_OK(hipEventRecord(c->ev_att, s)); HIP_OK(hipStreamWaitEvent(c->errs_stream, c->ev_att, 0));
        es = c->errs_stream;
    }
    tm.begin(es);
    const DevGenomeIdx gx{c->gx_gc_bits.as<unsigned long long>(), c->gx_n_bits.as<unsigned long long>(), c->gx_gc_pref.as<uint64_t>(), c->gx_n_pref.as<uint64_t>()};
    if (from_frag) launch_errs_frags(s, g, gx, fr, n_slots, slot_off, slots.as<uint32_t>(), slot_tmpl.as<uint32_t>(), valid_off.as<uint32_t>(),
                                     out.view(), out.n, out.pool_view(), c->flags.as<uint32_t>(), c->d_binom.as<unsigned long long>(), p,
                                     ride ? c->primer_cnt.as<int64_t>() : nullptr, c->primer_delta.as<uint32_t>(), c->primer_cut.as<unsigned long long>(), c->dsums.as<unsigned long long>(),
                                     c->dsums.as<unsigned long long>() + DS_SEMIS_N);
    else launch_errs_semis(es, g, gx, fr, c->semis.view(), nt, c->semis.pool_view(), n_slots, slot_off, slots.as<uint32_t>(), slot_tmpl.as<uint32_t>(),
                           valid_off.as<uint32_t>(), out.view(), out.n, out.pool_view(), c->flags.as<uint32_t>(), c->d_binom.as<unsigned long long>(), p,
                           ride && es == s ? c->primer_cnt.as<int64_t>() : nullptr, c->primer_delta.as<uint32_t>(), c->primer_cut.as<unsigned long long>(), c->dsums.as<unsigned long long>());
    tm.end(es);
    if (es != s) {
        HIP_OK(hipEventRecord(c->ev_errs, es)); c->errs_pending = true;
        if (ride) launch_primer_update(s, c->primer_cnt.as<int64_t>(), c->primer_delta.as<uint32_t>(), c->primer_cut.as<unsigned long long>(), c->dsums.as<unsigned long long>(), c->flags.as<uint32_t>());
    }
    if (c->sharded()) shard_close(c, from_frag ? valid_off.as<uint32_t>() + nt : nullptr);
    {   // counts of this pass -> mailbox (read by the host at the group's sync): new amplicons, and for a semi pass the
        // fulls made from the semis of each fragment pass (segments)
        c->pend.add(valid_off.as<uint32_t>() + nt, 4, rb_slot);
        if (!from_frag) {
            for (size_t b = 0; b < c->semi_block_end.size() && b < 8; ++b) c->pend.add(valid_off.as<uint32_t>() + std::min(c->semi_block_end[b], nt), 4, 16 + (int)b);
            c->pending_seg_cycle = (int)pass;
        }
    }
}
// closing a group of passes: their counts go to the mailbox (and the new semi count into the device scalars) ...
void collect_post(scs_ctx* c, bool post_now) {
    c->pend.add(c->dsums.as<unsigned long long>() + DS_SEMI_LEN, 8, 8);
    if (c->sharded()) c->pend.add(c->dsums.as<unsigned long long>() + DS_G_PRIMERS, 8, 9);
    c->pend.add(c->dsums.as<unsigned long long>() + DS_MIN_STOCK, 8, 10);
    if (post_now) { mail_post(c, c->pend, true); c->pend = Mail(); }               // else: rides on the next setPrimers mail
}
// ... and are taken over by the host after the next mail_wait: counts of new amplicons, total length of the semis
void collect_read(scs_ctx* c, int rb_fulls, int rb_semis) {
    if (rb_fulls >= 0) {
        c->fulls.n += (uint32_t)c->h_rb[rb_fulls]; c->tm_errscan.add_units(c->h_rb[rb_fulls]);
        if (c->pending_seg_cycle >= 0) {                                          // stored order within a cycle: fragment pass p descending
            const size_t nb = std::min<size_t>(c->semi_block_end.size(), 8);
            for (int b = (int)nb - 1; b >= 0; --b) {
                const uint32_t hi = (uint32_t)c->h_rb[16 + b], lo = b ? (uint32_t)c->h_rb[16 + b - 1] : 0u;
                c->full_segs.push_back(scs_ctx::Seg{c->pending_seg_cycle, b, hi - lo});
            }
            c->pending_seg_cycle = -1;
        }
    }
    if (rb_semis >= 0) { c->semis.n += (uint32_t)c->h_rb[rb_semis]; c->tm_errscan_f.add_units(c->h_rb[rb_semis]); c->semi_block_end.push_back(c->semis.n); }
    c->semi_total_len = c->h_rb[8];
    if (c->sharded()) c->total_primers = c->h_rb[9];                               // whole-job pool size after the budgets exchanged so far
    c->min_stock_lb = c->h_rb[10];                                                 // the smallest primer stock in use after the passes mailed so far
}

// ---------------------------------------------------------------- Malbac::amplify (Malbac.cpp:173-201)
void do_amplify(scs_ctx* c) {
    if (!c->have_frags) throw ScsError(SCS_EINVAL, "scs_amplify: call scs_create_frags first");
    if (!c->have_profile) throw ScsError(SCS_EINVAL, "scs_amplify: load a profile first");
    hipStream_t s = c->stream;
    if (c->cfg.verbose) fprintf(stderr, "\nMALBAC amplification...\n");
    c->semis.reset_counts(); c->fulls.reset_counts(); c->semi_block_end.clear(); c->full_segs.clear(); c->pending_seg_cycle = -1; c->pend = Mail();
    c->timing_gate = (c->amplify_calls++ % c->timing_every) == 0;
    c->tm_errscan.reset(); c->tm_errscan_f.reset(); c->tm_attach.reset(); c->tm_attach_f.reset();
    c->primer_cnt.reserve(65536 * 8, s); c->primer_cut.reserve(65536 * 8, s); c->primer_delta.reserve(65536 * 4, s);   // createPrimers: 4^8 types x `primers` copies
    if (c->sharded()) c->primer_gdelta.reserve((65536 + SHARD_TAIL_WORDS) * 4, s);
    c->min_stock_lb = c->cfg.primers > 0 ? (uint64_t)c->cfg.primers : 0; c->st.stock_checks = c->st.stock_exhausted_passes = c->st.stock_rounds = 0;
    launch_amplify_init(s, c->primer_cnt.as<int64_t>(), c->primer_cut.as<unsigned long long>(), (int64_t)c->cfg.primers, c->primer_delta.as<uint32_t>(),
                        c->sharded() ? c->primer_gdelta.as<uint32_t>() : nullptr, c->flags.as<uint32_t>(), c->dsums.as<unsigned long long>(),
                        c->nf_all, c->frag_len_all, 65536ull * (uint64_t)c->cfg.primers, c->semis.pool_head.as<uint32_t>(), c->fulls.pool_head.as<uint32_t>());
    if (!c->d_binom.p) {   // [REMAP] error-count thresholds for every window length (cfg is fixed for the ctx lifetime)
        std::vector<uint64_t> bt = binom_table(c->cfg.ber, c->cfg.amplicon_min_len - 8, c->cfg.amplicon_max_len - 8);
        upload(c->d_binom, bt, s); HIP_OK(hipStreamSynchronize(s));
    }
    c->total_primers = 65536ull * (uint64_t)c->cfg.primers;
    c->frag_total_len = 0; for (uint32_t l : c->f_len) c->frag_total_len += l;
    c->semi_total_len = 0;
    set_primers_launch(c, true, 0, 0); mail_wait(c); c->frag_copy_pending = false; set_primers_finish(c);
    launch_pass(c, true, 0, 5);
    int open_fulls = -1, open_semis = 5; uint32_t semis_in_flight = c->slots_f;     // the group of passes not read back yet
    for (uint32_t i = 0; i < 5; ++i) {
        if (c->total_primers == 0) break;
        if (c->cfg.verbose) fprintf(stderr, "cycle number: %u\n", i + 1);
        // ONE wait per cycle: the counts of the previous group and this cycle's budgets come back together.  setPrimers runs
        // on the device's own semi count; the host only bounds it (count so far + slots of the fragment pass in flight).
        collect_post(c, false);
        join_errs(c);
        set_primers_launch(c, false, i + 1, c->semis.n + semis_in_flight);
        mail_wait(c);
        collect_read(c, open_fulls, open_semis);
        set_primers_finish(c);
        launch_pass(c, false, i, 4);
        if (i < 4) launch_pass(c, true, i + 1, 5);
        open_fulls = 4; open_semis = i < 4 ? 5 : -1; semis_in_flight = i < 4 ? c->slots_f : 0;
        if (c->cfg.verbose) { fprintf(stderr, "semi amplicon amplification done!\n"); if (i < 4) fprintf(stderr, "fragment amplification done!\n"); }
    }
    join_errs(c);
    c->pend.add(c->flags.p, 4, 30);                                              // the overflow flags ride on the last collect: one wait, not two
    collect_post(c, true); mail_wait(c); collect_read(c, open_fulls, open_semis);
    flags_eval(c);
    c->tm_errscan.collect(); c->tm_errscan_f.collect(); c->tm_attach.collect(); c->tm_attach_f.collect();
    c->amplified = true; c->allocated = false;
    c->st.semi_amplicons = c->semis.n; c->st.full_amplicons = c->fulls.n; c->st.primers_left = c->total_primers;
}

// ---------------------------------------------------------------- a8 + a9: Malbac::setReadCounts (Malbac.cpp:370-408) on the device
void do_allocate(scs_ctx* c, uint64_t reads) {
    if (!c->amplified) throw ScsError(SCS_EINVAL, "scs_allocate_reads: call scs_amplify first");
    hipStream_t s = c->stream;
    if (reads == 0) {                                                             // Malbac::yieldReads, Malbac.cpp:413-420
        uint64_t ref_len = 0;
        for (auto& r : c->recs) { size_t p = r.name.rfind('_'); ref_len += (uint64_t)atoi(r.name.c_str() + (p == std::string::npos ? 0 : p + 1)); }
        ref_len /= 2;
        reads = (uint64_t)(ref_len * c->cfg.coverage / (long)c->prof.read_length);
    }
    if (c->cfg.verbose) fprintf(stderr, "\nNumber of reads to generate: %llu\n", (unsigned long long)reads);
    c->reads_requested = reads; c->st.reads_requested = reads;
    const uint32_t ac = c->fulls.n;
    double t0 = now_s();
    c->weights.reserve(std::max<size_t>((size_t)ac * 8, 16), s);
    c->read_numbers.reserve(((size_t)ac + 1) * 4, s); c->pair_off.reserve(((size_t)ac + 1) * 4, s);
    launch_weights(s, c->fulls.view(), ac, c->dtb, c->key, (uint32_t)c->cfg.frag_size, c->weights.as<double>());
    double* d_w = c->weights.as<double>(); uint32_t* d_rn = c->read_numbers.as<uint32_t>();

    // ---- the plan: where the chunks of the whole job's list lie relative to this shard's list.
    // slot = cycle * 8 + (7 - fragment pass): this shard's segments in local order; the whole job's list takes the
    // shards' segments slot by slot, shard by shard
    const int R = c->cfg.shard_count, me = c->cfg.shard_rank; const bool multi = c->sharded();
    std::vector<uint64_t> segc((size_t)R * ALLOC_SLOTS, 0);
    for (auto& sg : c->full_segs) { if (sg.c < 0 || sg.c >= 5 || sg.p < 0 || sg.p >= 8) throw ScsError(SCS_EINVAL, "allocation: segment out of range"); segc[(size_t)me * ALLOC_SLOTS + sg.c * 8 + (7 - sg.p)] += sg.count; }
    if (multi) c->reduce(segc.data(), segc.size());
    std::vector<AllocGSeg> gseg; std::vector<uint32_t> loff(R, 0); uint64_t total = 0;
    AllocPlan pl{}; pl.rank = (uint32_t)me;
    for (int sl = 0; sl < ALLOC_SLOTS; ++sl) for (int r = 0; r < R; ++r) {
        const uint64_t n = segc[(size_t)r * ALLOC_SLOTS + sl];
        if (r == me) pl.my_seg[sl] = AllocMySeg{total, loff[r], (uint32_t)n, (uint32_t)(sl * R + r), 0};
        if (!n) continue;
        gseg.push_back(AllocGSeg{total, loff[r], (uint32_t)n, (uint32_t)r, (uint32_t)sl});
        loff[r] += (uint32_t)n; total += n;
    }
    if (loff[me] != ac) throw ScsError(SCS_EINVAL, "sharded allocation: segment bookkeeping mismatch");
    if (total > 0xFFFFFFF0ull) throw ScsError(SCS_EOVERFLOW, "more than 2^32 amplicons in the whole job");
    const uint32_t nch = (uint32_t)((total + ALLOC_CHUNK - 1) / ALLOC_CHUNK);
    std::vector<AllocRange> rng; std::vector<AllocBChunk> bch; uint32_t nq = 0;
    {
        auto owner_of = [&](uint64_t gi) { size_t lo = 0, hi = gseg.size(); while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (gseg[mid].go <= gi) lo = mid; else hi = mid; } return gseg[lo].owner; };
        auto add_boundary = [&](uint32_t ch) { for (auto& b : bch) if (b.c == ch) return; bch.push_back(AllocBChunk{ch, (uint32_t)std::min<uint64_t>(ALLOC_CHUNK, total - (uint64_t)ch * ALLOC_CHUNK), owner_of((uint64_t)ch * ALLOC_CHUNK) == (uint32_t)me ? 1u : 0u}); };
        for (size_t k = 0; k < gseg.size();) {                                    // my segments, merged while they are contiguous in the whole list
            if (gseg[k].owner != (uint32_t)me) { ++k; continue; }
            uint64_t go = gseg[k].go, n = gseg[k].n; const uint32_t lo = gseg[k].lo; size_t j = k + 1;
            while (j < gseg.size() && gseg[j].owner == (uint32_t)me && gseg[j].go == go + n) { n += gseg[j].n; ++j; }
            k = j;
            const uint64_t cA = (go + ALLOC_CHUNK - 1) / ALLOC_CHUNK, cB = go + n == total ? nch : (go + n) / ALLOC_CHUNK;   // whole chunks inside [go, go+n)
            if (cA < cB) { rng.push_back(AllocRange{nq, (uint32_t)cA, (uint32_t)(lo + (cA * ALLOC_CHUNK - go))}); nq += (uint32_t)(cB - cA); }
            if (cA >= cB) { for (uint64_t ch = go / ALLOC_CHUNK; ch <= (go + n - 1) / ALLOC_CHUNK; ++ch) add_boundary((uint32_t)ch); }   // shorter than a chunk (or two partial ones)
            else {
                if (go % ALLOC_CHUNK) add_boundary((uint32_t)(go / ALLOC_CHUNK));
                if (cB * ALLOC_CHUNK < go + n) add_boundary((uint32_t)cB);
            }
        }
    }
    pl.total = total; pl.n_interior = nq; pl.n_boundary = (uint32_t)bch.size(); pl.n_ranges = (uint32_t)rng.size(); pl.n_gseg = (uint32_t)gseg.size();
    const uint32_t nwork = pl.n_interior + pl.n_boundary;
    {   // the plan's arrays: one small upload
        const size_t o_b = rng.size() * sizeof(AllocRange), o_g = o_b + bch.size() * sizeof(AllocBChunk), bytes = o_g + gseg.size() * sizeof(AllocGSeg);
        std::vector<uint8_t> blob(std::max<size_t>(bytes, 16));
        if (!rng.empty()) memcpy(blob.data(), rng.data(), o_b);
        if (!bch.empty()) memcpy(blob.data() + o_b, bch.data(), o_g - o_b);
        if (!gseg.empty()) memcpy(blob.data() + o_g, gseg.data(), bytes - o_g);
        c->a_plan.reserve(blob.size(), s);
        HIP_OK(hipMemcpyAsync(c->a_plan.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s)); HIP_OK(hipStreamSynchronize(s));
        pl.rng = (const AllocRange*)c->a_plan.p; pl.bchunk = (const AllocBChunk*)((char*)c->a_plan.p + o_b); pl.gseg = (const AllocGSeg*)((char*)c->a_plan.p + o_g);
    }
    for (int sl = 0; sl < ALLOC_SLOTS; ++sl) c->seg_lo[sl] = pl.my_seg[sl].lo;
    c->seg_lo[ALLOC_SLOTS] = ac;
    c->gmap = SegMap{};
    if (multi) { uint32_t k = 0; for (int sl = 0; sl < ALLOC_SLOTS; ++sl) if (pl.my_seg[sl].n) { c->gmap.lo[k] = pl.my_seg[sl].lo; c->gmap.cnt[k] = pl.my_seg[sl].n; c->gmap.go[k] = pl.my_seg[sl].go; ++k; } c->gmap.n = k; }

    // ---- buffers: per-chunk partials of the WHOLE job (8 B per 1000 amplicons), per-work-chunk partials of this shard
    const size_t tree_scratch = (size_t)nch / ALLOC_CHUNK * 3 + 4096;
    c->a_part.reserve(((size_t)nch + 2) * 8, s); c->a_tp.reserve(((size_t)nch + 2) * 8, s); c->a_probs.reserve(((size_t)nch + 2) * 8, s);
    c->a_quota.reserve(((size_t)nch + 2) * 4, s); c->a_crn.reserve(((size_t)nwork + 2) * 4, s); c->a_scratch.reserve(tree_scratch * 8, s);
    c->a_brow.reserve(std::max<size_t>((size_t)pl.n_boundary * ALLOC_CHUNK * 8, 16), s); c->a_bmap.reserve(std::max<size_t>((size_t)pl.n_boundary * ALLOC_CHUNK * 4, 16), s);
    c->odd_before.reserve(((size_t)ac + 1) * 4, s); c->scan_tmp.reserve(scan_temp_bytes(ac), s);
    AllocState* st = (AllocState*)((char*)c->dsums.p + 128);
    double* d_part = c->a_part.as<double>(); double* d_tp = c->a_tp.as<double>();
    unsigned long long* d_sum_rn = (unsigned long long*)(d_tp + nch);               // rides behind tp[] on the same all-reduce
    if (R > 1) {   // first / last 1000 weights of every segment of every shard: what the boundary rows of the others need
        const size_t per = (size_t)ALLOC_SLOTS * 2 * ALLOC_CHUNK * 8;
        c->a_send.reserve(per, s); c->a_gath.reserve(per * R, s);
        launch_alloc_bpack(s, d_w, pl, c->a_send.as<double>());
        c->gather_dev(c->a_send.p, c->a_gath.p, per);
    }
    launch_alloc_bgather(s, d_w, pl, c->a_gath.as<double>(), c->a_brow.as<double>(), c->a_bmap.as<int>());
    if (multi) HIP_OK(hipMemsetAsync(d_part, 0, (size_t)nch * 8, s));               // owners fill their chunks; the all-reduce sums disjoint entries (x + 0 = x)
    launch_alloc_chunk_sum(s, d_w, c->a_brow.as<double>(), pl, d_part);
    if (multi) c->reduce_dev(d_part, nch, 8);
    launch_tree_sum(s, d_part, nch, c->a_scratch.as<double>(), &st->total);
    if (multi) HIP_OK(hipMemsetAsync(d_tp, 0, ((size_t)nch + 1) * 8, s));
    launch_alloc_norm(s, d_w, c->a_brow.as<double>(), c->a_bmap.as<int>(), pl, &st->total, reads, d_rn, d_tp, c->a_crn.as<uint32_t>(), d_sum_rn);
    if (multi) c->reduce_dev(d_tp, (uint64_t)nch + 1, 8);
    launch_alloc_quota(s, d_tp, nch, reads, d_sum_rn, &st->sum_quota, c->a_quota.as<uint32_t>(), c->a_probs.as<double>(), c->a_scratch.as<double>(), c->key);
    launch_alloc_sample(s, d_w, c->a_brow.as<double>(), c->a_bmap.as<int>(), pl, d_tp, c->a_quota.as<uint32_t>(), c->key, d_rn);
    if (c->cfg.paired && !multi) launch_parity_pair_offsets(s, d_rn, ac, c->pair_off.as<uint32_t>(), c->scan_tmp.p, c->scan_tmp.cap);
    else if (c->cfg.paired) {
        launch_alloc_odd_scan(s, d_rn, ac, c->odd_before.as<uint32_t>(), c->scan_tmp.p, c->scan_tmp.cap);
        unsigned long long* table = nullptr;
        if (multi) {   // odd entries of every segment of every shard, in list order
            c->a_odd.reserve((size_t)R * ALLOC_SLOTS * 8, s); table = c->a_odd.as<unsigned long long>();
            HIP_OK(hipMemsetAsync(table, 0, (size_t)R * ALLOC_SLOTS * 8, s));
            launch_alloc_odd_counts(s, c->odd_before.as<uint32_t>(), pl, table);
            c->reduce_dev(table, (uint64_t)R * ALLOC_SLOTS, 8);
        }
        launch_alloc_parity(s, d_rn, c->odd_before.as<uint32_t>(), ac, pl, table);
    }
    if (!c->cfg.paired || multi) launch_pair_offsets(s, d_rn, ac, c->cfg.paired != 0, c->pair_off.as<uint32_t>(), c->scan_tmp.p, c->scan_tmp.cap);
    { Mail m; m.add(ac ? (const void*)(c->pair_off.as<uint32_t>() + ac) : nullptr, 4, 0); mail_post(c, m, true); }
    mail_wait(c);
    c->n_pairs_planned = (uint32_t)c->h_rb[0];
    c->st.t_stage[3] = 0; c->st.t_stage[4] = now_s() - t0;
    c->allocated = true;
}

// ---------------------------------------------------------------- a10/a11/a13/a16: yieldReads
// FASTQ sink pipeline (replaces the mutexed ofstream of lib/seqwriter/SeqWriter.cpp:41-54).  A batch's text is
// copied D2H on the copy stream into a free pinned slot and handed to the writer thread of its REGION (BatchSink: the job's
// records are cut into `regions` contiguous ranges, visited round-robin, one writer thread and one pair of files each), which
// waits for the copy's event, writes, and frees the slot -- while the GPU already produces the next batches.  writers + 2
// slots: every writer can hold one while one is being filled and one crosses PCIe.  (regions = writers x generations: writer w
// serves the regions r = w mod writers, one after the other.)
// ---- where the sink's host work runs.  A GPU hangs on one NUMA node of the host; a copy into pinned memory of the OTHER node runs at
// half the rate (profiles/r03_numa_probe.log: 29 against 57 GB/s), and on a node with several GPUs every rank's writers should stay
// on their own GPU's node.  gpu_local_cpus: the CPUs of the ctx device's node that this process may run on (empty: unknown, or no
// choice to make); NumaScope binds the calling thread to them for its lifetime (pinned allocations: first touch).
static std::vector<int> gpu_local_cpus(int device) {
    std::vector<int> out; char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return out;
    for (char* q = bdf; *q; ++q) *q = (char)tolower(*q);
    int node = -1;
    { FILE* f = fopen((std::string("/sys/bus/pci/devices/") + bdf + "/numa_node").c_str(), "r"); if (!f) return out; if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return out;
    char list[4096] = {0};
    { FILE* f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r"); if (!f) return out; if (!fgets(list, sizeof list, f)) list[0] = 0; fclose(f); }
    cpu_set_t allowed; CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
    for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0; const int k = sscanf(tok, "%d-%d", &a, &b); if (k < 1) continue; if (k == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) out.push_back(c);
    }
    if ((int)out.size() == CPU_COUNT(&allowed)) out.clear();                       // the whole mask is local already
    return out;
}
struct NumaScope {
    cpu_set_t old; bool on = false;
    explicit NumaScope(const std::vector<int>& cpus) {
        if (cpus.empty() || pthread_getaffinity_np(pthread_self(), sizeof old, &old) != 0) return;
        cpu_set_t s; CPU_ZERO(&s); for (int c : cpus) CPU_SET(c, &s);
        on = pthread_setaffinity_np(pthread_self(), sizeof s, &s) == 0;
    }
    ~NumaScope() { if (on) (void)pthread_setaffinity_np(pthread_self(), sizeof old, &old); }
};

struct SinkPipe {
    std::vector<int> local_cpus;                                                   // of the device's NUMA node (gpu_local_cpus)
    struct Slot { char* h[2] = {nullptr, nullptr}; size_t cap[2] = {0, 0}; hipEvent_t ev = nullptr; bool busy = false; };
    struct Job { int slot, region; size_t n1, n2; };
    struct Writer { std::thread th; std::vector<Job> q; };
    std::vector<Slot> slots; std::vector<Writer> writers;
    std::mutex mu; std::condition_variable cv; bool done = false, failed = false;
    BatchSink* sink = nullptr; bool paired = true; int device = 0;
    void start(BatchSink* f, bool pe, int dev) {
        sink = f; paired = pe; device = dev; done = failed = false;
        local_cpus = gpu_local_cpus(dev);
        const size_t nw = (size_t)std::max(1, f->writers), want = nw + 2;
        // (blocking events: a writer that waits for its batch's copy sleeps instead of spinning -- the host's cores are the sink's bottleneck)
        while (slots.size() < want) { Slot sl; HIP_OK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming | hipEventBlockingSync)); slots.push_back(sl); }
        for (auto& sl : slots) sl.busy = false;
        writers = std::vector<Writer>(nw);
        for (size_t w = 0; w < writers.size(); ++w) writers[w].th = std::thread([this, w] {
            (void)hipSetDevice(device);
            if (!local_cpus.empty()) { cpu_set_t cs; CPU_ZERO(&cs); for (int c : local_cpus) CPU_SET(c, &cs); (void)pthread_setaffinity_np(pthread_self(), sizeof cs, &cs); }   // a writer stays on its GPU's node
            Writer& W = writers[w];
            for (;;) {
                Job j;
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !W.q.empty() || done; }); if (W.q.empty()) return; j = W.q.front(); W.q.erase(W.q.begin()); }
                Slot& sl = slots[(size_t)j.slot];
                bool bad = hipEventSynchronize(sl.ev) != hipSuccess;
                if (!bad && !failed) bad = sink->put(j.region, sl.h[0], j.n1, paired ? sl.h[1] : nullptr, j.n2) != 0;
                { std::lock_guard<std::mutex> lk(mu); sl.busy = false; if (bad) failed = true; }
                cv.notify_all();
            }
        });
    }
    // a free pinned slot with room for the batch (blocks while every slot is with a writer); -1: the sink failed
    int acquire(size_t need1, size_t need2) {
        int k = -1;
        { std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { if (failed) return true; for (size_t i = 0; i < slots.size(); ++i) if (!slots[i].busy) { k = (int)i; return true; } return false; });
          if (failed) return -1;
          slots[(size_t)k].busy = true; }
        Slot& sl = slots[(size_t)k];
        for (int f = 0; f < 2; ++f) {
            const size_t need = f == 0 ? need1 : need2;
            if (need > sl.cap[f]) {
                if (sl.h[f]) HIP_OK(hipHostFree(sl.h[f]));
                sl.h[f] = nullptr; sl.cap[f] = 0;
                const size_t nc = std::max<size_t>(need + need / 8, 1 << 20);
                NumaScope here(local_cpus);                                        // the slot's pages on the GPU's node
                HIP_OK(hipHostMalloc((void**)&sl.h[f], nc, hipHostMallocDefault)); sl.cap[f] = nc;
            }
        }
        return k;
    }
    void submit(int region, int slot, size_t n1, size_t n2) { { std::lock_guard<std::mutex> lk(mu); writers[(size_t)region % writers.size()].q.push_back(Job{slot, region, n1, n2}); } cv.notify_all(); }
    bool finish() { { std::lock_guard<std::mutex> lk(mu); done = true; } cv.notify_all(); for (auto& W : writers) if (W.th.joinable()) W.th.join(); writers.clear(); return !failed; }
    void release() { for (auto& sl : slots) { for (int f = 0; f < 2; ++f) if (sl.h[f]) (void)hipHostFree(sl.h[f]); if (sl.ev) (void)hipEventDestroy(sl.ev); } slots.clear(); }
};
// a caller's scs_sink_fn as a BatchSink: one region, the batches in record order
struct CallbackSink : BatchSink {
    scs_sink_fn fn; void* user;
    CallbackSink(scs_sink_fn f, void* u) : fn(f), user(u) {}
    int put(int, const char* a, size_t na, const char* b, size_t nb) override { return fn(user, a, na, b, nb); }
};

struct OutTarget { bool device; char* d1; char* d2; size_t cap1, cap2; BatchSink* sink;
                   std::vector<uint64_t>* seg_off1 = nullptr; std::vector<uint64_t>* seg_off2 = nullptr;
                   bool bgzf = false; };                                          // bgzf: the sink gets BGZF blocks made on the device instead of the text   // seg_off: byte offset of each list segment's first record (shard index)

void do_yield(scs_ctx* c, const OutTarget& tg, uint64_t* n1_out, uint64_t* n2_out, uint64_t* pairs_out) {
    if (!c->allocated) throw ScsError(SCS_EINVAL, "scs_yield_reads: call scs_allocate_reads first");
    hipStream_t s = c->stream; const int paired = c->cfg.paired != 0;
    if (c->cfg.verbose) fprintf(stderr, "\n*****Producing reads*****\n");
    c->timing_gate = (c->yield_calls++ % c->timing_every) == 0;
    c->tm_reads.reset(); c->tm_indels.reset();
    const uint64_t P = c->n_pairs_planned;
    const uint32_t L = (uint32_t)c->prof.read_length, slot = ((L + 64 + 63) / 64) * 64;
    c->pairs.reserve(std::max<size_t>(P * sizeof(PairRec), 16), s);
    HIP_OK(hipMemsetAsync(c->dsums.as<unsigned long long>() + DS_HOLES, 0, 8, s));
    const bool to_sink = !tg.device && tg.sink;
    const int regions = to_sink ? std::max(1, tg.sink->regions) : 1;
    // pairs per batch: 8 M with the text staying in HBM (5 GB of text per batch: the base pass' grids are long enough for their tails and
    // the per-batch pre-pass not to matter: 2 M -> 8 M gave -11 % on the stage).  Towards a sink a batch fills a pinned slot and every
    // writer holds one: as large as leaves each part file of each generation a couple of batches -- 2 M pairs (1.3 GB of text) on a
    // whole-genome job, where the base pass then runs at the rate it has in HBM (256 k-pair launches ran at 0.09 of the HBM roofline
    // with the chip half empty through their tails, 2 M-pair ones at 0.15: profiles/r04_sink_batch_sizes.log; the job, bound by the
    // host's copies, is the same to within its run-to-run spread) --, never fewer than 256 k (512 k with few writers)
    static const int batch_shift = seam_env("SCS_TEST_BATCH_SHIFT") ? atoi(seam_env("SCS_TEST_BATCH_SHIFT")) : 0;   // tests: many small batches
    uint64_t sink_batch = 1ull << 19;
    if (to_sink && tg.sink->writers > 4) {
        const uint64_t per_part = P / (2ull * (uint64_t)std::max(1, regions));     // two batches per part file
        sink_batch = 1ull << 18; while (sink_batch < (1ull << 21) && sink_batch * 2 <= per_part) sink_batch <<= 1;
    }
    const uint64_t batch = std::min<uint64_t>(std::max<uint64_t>(P, 1), batch_shift ? (1ull << batch_shift) : to_sink ? sink_batch : (1ull << 23));
    // The pairs are planned (k_plan_pairs: insert sizes, positions, the amplicon resolved to an index map) batch by batch, at the
    // head of each batch's pre-pass: bounds[b] = the amplicon that holds the batch's first pair.
    const uint32_t nbatch = (uint32_t)((P + batch - 1) / batch);
    std::vector<uint32_t> bounds(nbatch + 1, 0);
    if (P) {
        c->d_bounds.reserve(((size_t)nbatch + 1) * 4, s);
        launch_batch_bounds(s, c->pair_off.as<uint32_t>(), c->fulls.n, batch, nbatch, c->d_bounds.as<uint32_t>());
        HIP_OK(hipMemcpyAsync(bounds.data(), c->d_bounds.p, ((size_t)nbatch + 1) * 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
    }
    // The order the batches are made in.  One region: record order.  Several (a sink with `writers` threads and regions = writers x
    // generations): region r owns the contiguous batches [r nbatch / regions, (r + 1) nbatch / regions); generation after generation,
    // the `writers` regions of a generation are visited round-robin, so every writer always has a batch of its own range on the way
    // while each range still arrives in record order -- and a generation's parts are complete when the next one starts.
    const int n_writers = to_sink ? std::max(1, std::min(tg.sink->writers, regions)) : 1;
    std::vector<uint32_t> order, region_of; order.reserve(nbatch); region_of.reserve(nbatch);
    for (int g0 = 0; g0 < regions; g0 += n_writers) {
        const int g1 = std::min(regions, g0 + n_writers);
        std::vector<uint32_t> next((size_t)(g1 - g0)), end((size_t)(g1 - g0)); size_t left = 0;
        for (int r = g0; r < g1; ++r) { next[(size_t)(r - g0)] = (uint32_t)((uint64_t)nbatch * r / regions); end[(size_t)(r - g0)] = (uint32_t)((uint64_t)nbatch * (r + 1) / regions); left += end[(size_t)(r - g0)] - next[(size_t)(r - g0)]; }
        while (left) for (int r = g0; r < g1; ++r) if (next[(size_t)(r - g0)] < end[(size_t)(r - g0)]) { order.push_back(next[(size_t)(r - g0)]++); region_of.push_back((uint32_t)r); --left; }
    }
    struct PipeGuard { SinkPipe* p; ~PipeGuard() { if (p) (void)p->finish(); } } guard{nullptr};
    if (to_sink) {
        if (!c->pipe) c->pipe = new SinkPipe;
        if (!c->copy_stream) { HIP_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)); for (int k = 0; k < 2; ++k) { HIP_OK(hipEventCreateWithFlags(&c->ev_made[k], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&c->ev_d2h[k], hipEventDisableTiming)); } }
        c->pipe->start(tg.sink, paired != 0, c->cfg.device); guard.p = c->pipe;
    }
    const bool bgzf = to_sink && tg.bgzf;
    if (bgzf && !c->h_z) {
        HIP_OK(hipHostMalloc((void**)&c->h_z, 64, hipHostMallocDefault)); memset(c->h_z, 0, 64);
        for (int k = 0; k < 2; ++k) HIP_OK(hipEventCreateWithFlags(&c->ev_z[k], hipEventDisableTiming | hipEventBlockingSync));
        std::vector<uint32_t> tabs(512); bgzf_host_tables(tabs.data(), tabs.data() + 256);
        upload(c->z_crc, tabs, s); HIP_OK(hipStreamSynchronize(s));
    }
    uint64_t bi = 0;                                                               // batches handed to the sink so far
    // Per batch a PRE-PASS (indel events -> record sizes -> offsets, class lists; k_indels + scans) must finish before the host
    // can launch the base pass (it needs the batch's byte counts and class counts).  The pre-pass of batch i+1 is therefore
    // queued BEFORE the base pass of batch i, into a second set of buffers: while the host waits for its mail the GPU
    // still has a base pass to run.
    const uint64_t nreads_b = paired ? 2 * batch : batch;
    c->ev_hdr.reserve(2 * nreads_b * 4, s); c->ev_dat.reserve(2 * nreads_b * 16, s);
    c->sizes1.reserve(2 * (batch + 1) * 4, s); c->sizes2.reserve(2 * (batch + 1) * 4, s); c->off1.reserve(2 * (batch + 1) * 8, s); c->off2.reserve(2 * (batch + 1) * 8, s);
    c->scan_tmp.reserve(scan_temp_bytes(batch), s);
    // the reads of a batch split by class (with / without indel events): flags, their scans, four lists of pair indices
    c->rl_cls.reserve(2 * (batch + 1) * 2 * 4, s); c->rl_pos.reserve(2 * (batch + 1) * 2 * 4, s); c->rl_lists.reserve(2 * batch * 6 * 4, s);
    struct BatchSet { uint32_t* ev_hdr; uint4* ev_dat; uint32_t *sizes1, *sizes2; uint64_t *off1, *off2; uint32_t *d1f1, *d1f2, *d1p1, *d1p2, *slist1, *slist2, *clist1, *clist2, *dlist1, *dlist2; } bs[2];   // d1f / d1p: the one-deletion class' flags and their scan
    for (int k = 0; k < 2; ++k) {
        bs[k].ev_hdr = c->ev_hdr.as<uint32_t>() + k * nreads_b; bs[k].ev_dat = c->ev_dat.as<uint4>() + k * nreads_b;
        bs[k].sizes1 = c->sizes1.as<uint32_t>() + k * (batch + 1); bs[k].sizes2 = c->sizes2.as<uint32_t>() + k * (batch + 1);
        bs[k].off1 = c->off1.as<uint64_t>() + k * (batch + 1); bs[k].off2 = c->off2.as<uint64_t>() + k * (batch + 1);
        bs[k].d1f1 = c->rl_cls.as<uint32_t>() + k * 2 * (batch + 1); bs[k].d1f2 = bs[k].d1f1 + batch + 1;
        bs[k].d1p1 = c->rl_pos.as<uint32_t>() + k * 2 * (batch + 1); bs[k].d1p2 = bs[k].d1p1 + batch + 1;
        bs[k].slist1 = c->rl_lists.as<uint32_t>() + k * 6 * batch; bs[k].slist2 = bs[k].slist1 + batch; bs[k].clist1 = bs[k].slist2 + batch; bs[k].clist2 = bs[k].clist1 + batch;
        bs[k].dlist1 = bs[k].clist2 + batch; bs[k].dlist2 = bs[k].dlist1 + batch;
    }
    // The pre-pass runs on a stream of its own, BESIDE the previous batch's base pass (it is memory-bound and short, the base pass
    // compute-bound).  Its buffer set must be free (the base pass two batches back, which read it, is over: ev_free) and the
    // base pass of its batch starts when the host has seen its mail.  SCS_READS_SERIAL=1: everything on the ctx stream.
    static const bool serial_pre = seam_env("SCS_READS_SERIAL") != nullptr;
    hipStream_t ps = s; bool free_rec[2] = {false, false};
    if (!serial_pre) {
        if (!c->pre_stream) {
            HIP_OK(hipStreamCreateWithFlags(&c->pre_stream, hipStreamNonBlocking)); HIP_OK(hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming));
            for (int k = 0; k < 2; ++k) { HIP_OK(hipEventCreateWithFlags(&c->ev_pre[k], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&c->ev_free[k], hipEventDisableTiming)); }
        }
        ps = c->pre_stream;
        HIP_OK(hipEventRecord(c->ev_plan, s)); HIP_OK(hipStreamWaitEvent(ps, c->ev_plan, 0));   // the pair records (and everything before) are made
    }
    auto prepass = [&](uint64_t p0, const BatchSet& B, int k) {
        hipStream_t s = ps;                                                        // (shadows the ctx stream inside the pre-pass)
        if (ps != c->stream && free_rec[k]) HIP_OK(hipStreamWaitEvent(ps, c->ev_free[k], 0));
        const uint32_t np = (uint32_t)std::min<uint64_t>(batch, P - p0);
        const PairRec* pr = c->pairs.as<PairRec>() + p0;
        {   // this batch's pair records: its amplicons, the one that straddles the next batch's start included
            const uint32_t b = (uint32_t)(p0 / batch), a_lo = bounds[b], a_hi = std::min<uint32_t>(c->fulls.n, bounds[b + 1] + 1u);
            launch_plan_pairs(s, c->frags_view(), c->semis.view(), c->fulls.view(), a_lo, a_hi - a_lo, (uint32_t)p0, (uint32_t)(p0 + np), c->read_numbers.as<uint32_t>(), c->pair_off.as<uint32_t>(),
                              c->gmap, c->dtb, c->key, paired, c->pairs.as<PairRec>(), c->dsums.as<unsigned long long>() + DS_HOLES);
        }
        // the indel pass fixes every read's length, hence the record sizes and (prefix sums) the record offsets
        c->tm_indels.begin(s);
        launch_indels(s, pr, np, paired, c->dtb, c->key, slot, B.ev_hdr, B.ev_dat, B.sizes1, B.sizes2, B.d1f1, B.d1f2, c->flags.as<uint32_t>());
        c->tm_indels.end(s);
        c->tm_indels.add_units(np);
        exclusive_scan_sizes(s, B.sizes1, B.off1, np, c->scan_tmp.p, c->scan_tmp.cap);   // byte offsets + positions in the class lists: one scan per mate
        if (paired) exclusive_scan_sizes(s, B.sizes2, B.off2, np, c->scan_tmp.p, c->scan_tmp.cap);
        launch_read_lists(s, np, paired, B.sizes1, B.off1, B.d1f1, B.d1p1, B.sizes2, B.off2, B.d1f2, B.d1p2, B.slist1, B.slist2, B.clist1, B.clist2, B.dlist1, B.dlist2,
                          c->scan_tmp.p, c->scan_tmp.cap);
        Mail m; m.add(B.off1 + np, 8, 0); m.add(paired ? (const void*)(B.off2 + np) : nullptr, 8, 1);
        m.add(B.d1p1 + np, 4, 2); m.add(paired ? (const void*)(B.d1p2 + np) : nullptr, 4, 3); mail_post(c, m, true, s);
        if (ps != c->stream) HIP_OK(hipEventRecord(c->ev_pre[k], ps));
    };
    uint64_t tot1 = 0, tot2 = 0, pairs_written = 0;
    // shard index: the pair index at which each list segment starts (pair_off at the segment's first amplicon); the byte offset of
    // that record = the bytes of the batches before its batch (known once every batch is made) + its offset inside the batch
    std::vector<uint64_t> bpair; std::vector<uint64_t> bb1(nbatch, 0), bb2(nbatch, 0);
    struct SegAt { size_t seg; uint32_t b; uint64_t o1, o2; }; std::vector<SegAt> seg_at;
    if (tg.seg_off1) {
        std::vector<uint32_t> v(ALLOC_SLOTS + 1, 0);
        for (int k = 0; k <= ALLOC_SLOTS; ++k) HIP_OK(hipMemcpyAsync(&v[k], c->pair_off.as<uint32_t>() + c->seg_lo[k], 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        bpair.assign(v.begin(), v.end()); tg.seg_off1->assign(ALLOC_SLOTS + 1, 0); if (tg.seg_off2) tg.seg_off2->assign(ALLOC_SLOTS + 1, 0);
    }
    bool d2h_rec[2] = {false, false};
    uint64_t sunk1 = 0, sunk2 = 0;                                                  // bytes handed to the sink (= the text's, or its BGZF blocks')
    struct Ship { char* p1; char* p2; uint64_t n1, n2; int dsl; uint32_t region; };
    Ship pending{}; bool have_pending = false;
    auto ship = [&](Ship sh) {                                                      // D2H on the copy stream into a free pinned slot, then to the region's writer
        SinkPipe* pp = c->pipe;
        if (bgzf) { HIP_OK(hipEventSynchronize(c->ev_z[sh.dsl])); sh.n1 = c->h_z[sh.dsl * 2]; sh.n2 = c->h_z[sh.dsl * 2 + 1]; }   // the blocks' totals have arrived
        const int hs = pp->acquire(sh.n1, sh.n2);                                   // (a pinned slot no writer holds: the host waits here when the sink is the slower side)
        if (hs < 0) throw ScsError(SCS_EIO, "sink aborted");
        SinkPipe::Slot& H = pp->slots[(size_t)hs];
        HIP_OK(hipStreamWaitEvent(c->copy_stream, c->ev_made[sh.dsl], 0));          // ... and crosses PCIe on the copy stream, beside the next batch's kernels
        if (sh.n1) HIP_OK(hipMemcpyAsync(H.h[0], sh.p1, sh.n1, hipMemcpyDeviceToHost, c->copy_stream));
        if (sh.n2) HIP_OK(hipMemcpyAsync(H.h[1], sh.p2, sh.n2, hipMemcpyDeviceToHost, c->copy_stream));
        HIP_OK(hipEventRecord(H.ev, c->copy_stream));
        HIP_OK(hipEventRecord(c->ev_d2h[sh.dsl], c->copy_stream)); d2h_rec[sh.dsl] = true;
        pp->submit((int)sh.region, hs, sh.n1, sh.n2);
        sunk1 += sh.n1; sunk2 += sh.n2;
    };
    c->cks.clear();
    if (c->want_cks && !tg.device) c->d_cks.reserve(std::max<size_t>((size_t)nbatch * 16, 16), s);
    if (P) prepass((uint64_t)order[0] * batch, bs[0], 0);
    for (uint64_t it = 0; it < nbatch; ++it) {
        const uint32_t bidx = order[it]; const uint64_t p0 = (uint64_t)bidx * batch;
        const uint32_t np = (uint32_t)std::min<uint64_t>(batch, P - p0);
        const PairRec* pr = c->pairs.as<PairRec>() + p0;
        const BatchSet& B = bs[it & 1];
        mail_wait(c);                                                              // this batch's byte and class counts
        const uint64_t b1 = c->h_rb[0] & OFF_MASK, b2 = c->h_rb[1] & OFF_MASK; const uint32_t nc1 = (uint32_t)(c->h_rb[0] >> OFF_BITS), nc2 = (uint32_t)(c->h_rb[1] >> OFF_BITS), nd1 = (uint32_t)c->h_rb[2], nd2 = (uint32_t)c->h_rb[3];
        if (ps != s) HIP_OK(hipStreamWaitEvent(s, c->ev_pre[it & 1], 0));          // (the host has seen the pre-pass' mail already: ordering for the device's sake)
        if (it + 1 < nbatch) prepass((uint64_t)order[it + 1] * batch, bs[(it + 1) & 1], (int)((it + 1) & 1));   // the next batch's pre-pass starts now, beside this batch's base pass
        bb1[bidx] = b1; bb2[bidx] = b2;
        for (size_t j = (size_t)(std::lower_bound(bpair.begin(), bpair.end(), p0) - bpair.begin()); j < bpair.size() && bpair[j] < p0 + np; ++j) {   // segments that start inside this batch
            uint64_t o1v = 0, o2v = 0; const uint64_t idx = bpair[j] - p0;
            HIP_OK(hipMemcpyAsync(&o1v, B.off1 + idx, 8, hipMemcpyDeviceToHost, s));
            if (paired) HIP_OK(hipMemcpyAsync(&o2v, B.off2 + idx, 8, hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            seg_at.push_back(SegAt{j, bidx, o1v & OFF_MASK, o2v & OFF_MASK});
        }
        char *o1, *o2;
        SinkPipe* pp = to_sink ? c->pipe : nullptr; const int dsl = (int)(bi & 1);
        if (tg.device) {
            if (tot1 + b1 > tg.cap1 || tot2 + b2 > tg.cap2) throw ScsError(SCS_EOVERFLOW, "scs_yield_reads_device: output buffer too small");
            o1 = tg.d1 + tot1; o2 = tg.d2 ? tg.d2 + tot2 : nullptr;
        } else {
            // sink mode: two device buffers.  One is free for this batch's k_reads once the D2H of the batch two back has left it
            // (ev_d2h: the stream waits, not the host), so the text of a batch crosses PCIe beside the next batch's kernels.
            DevBuf& d1 = (pp && dsl) ? c->out1b : c->out1; DevBuf& d2 = (pp && dsl) ? c->out2b : c->out2;
            const uint64_t want1 = std::max<uint64_t>(b1 + b1 / 16, 16), want2 = std::max<uint64_t>(b2 + b2 / 16, 16);
            if (pp && d2h_rec[dsl]) {
                if (want1 > d1.cap || want2 > d2.cap) HIP_OK(hipEventSynchronize(c->ev_d2h[dsl]));   // the buffer is about to move: its last copy must be out
                else HIP_OK(hipStreamWaitEvent(s, c->ev_d2h[dsl], 0));
            }
            d1.reserve(want1, s); d2.reserve(want2, s);
            o1 = d1.as<char>(); o2 = d2.as<char>();
        }
        c->tm_reads.begin(s);                                                      // the base pass writes the FASTQ text at the record offsets
        launch_reads(s, c->genome.as<uint8_t>(), c->genome2.as<uint32_t>() + 16, c->semis.pool_view(), c->fulls.pool_view(), pr, np, 0,
                     c->dtb, c->d_tables.as<DevTables>(), c->key, paired, slot, B.ev_hdr, B.ev_dat,
                     B.off1, B.off2, o1, o2, c->flags.as<uint32_t>(), b1, b2, B.slist1, B.slist2, B.clist1, B.clist2, nc1, nc2, B.dlist1, B.dlist2, nd1, nd2, &c->reads_side);
        c->tm_reads.end(s);
        c->tm_reads.add_units(np);
        if (c->want_cks && !tg.device) {
            launch_text_checksum(s, o1, b1, c->d_cks.as<unsigned long long>() + 2 * (size_t)bidx);
            launch_text_checksum(s, o2, paired ? b2 : 0, c->d_cks.as<unsigned long long>() + 2 * (size_t)bidx + 1);
        }
        if (ps != s) { HIP_OK(hipEventRecord(c->ev_free[it & 1], s)); free_rec[it & 1] = true; }   // this batch's buffer set is free for the pre-pass after next
        { const hipError_t le = take_launch_error(); if (le != hipSuccess) throw ScsError(SCS_EDEVICE, std::string("k_reads launch failed: ") + hipGetErrorString(le)); }
        if (pp) {
            Ship sh{o1, o2, b1, b2, dsl, region_of[it]};
            if (bgzf) {
                // the text becomes BGZF blocks where it lies: plan (code lengths, exact block sizes), prefix sum, emit at the final offsets.
                // The blocks' total is only known on the device: it travels to a pinned word behind ev_z, and the batch is shipped ONE
                // ITERATION LATER, when the host reads it without waiting while the GPU works on the next batch.
                for (int m = 0; m < (paired ? 2 : 1); ++m) {
                    const uint64_t nb = m ? b2 : b1; const uint32_t nblk = bgzf_blocks(nb);
                    c->z_plan[m].reserve(std::max<size_t>((size_t)nblk * BGZF_PLAN_BYTES, 16), s); c->z_sizes[m].reserve(((size_t)nblk + 2) * 4, s); c->z_offs[m].reserve(((size_t)nblk + 2) * 4, s);
                    DevBuf& zo = c->z_out[dsl][m];
                    if (bgzf_bound(nb) > zo.cap && d2h_rec[dsl]) HIP_OK(hipEventSynchronize(c->ev_d2h[dsl]));
                    zo.reserve(bgzf_bound(nb), s);
                    if (bgzf_bound(nb) > 0xFFFFFFF0ull) throw ScsError(SCS_EOVERFLOW, "BGZF: a batch's text exceeds 4 GB");
                    launch_bgzf_plan(s, m ? o2 : o1, nb, c->z_plan[m].as<uint8_t>(), c->z_sizes[m].as<uint32_t>());
                    exclusive_scan_u32(s, c->z_sizes[m].as<uint32_t>(), c->z_offs[m].as<uint32_t>(), nblk, nullptr, 0);   // (n <= 256 k: the one-workgroup scan, no scratch)
                    launch_bgzf_emit(s, m ? o2 : o1, nb, c->z_plan[m].as<uint8_t>(), c->z_sizes[m].as<uint32_t>(), c->z_offs[m].as<uint32_t>(),
                                     c->z_crc.as<uint32_t>(), c->z_crc.as<uint32_t>() + 256, zo.as<char>(), 0);
                    HIP_OK(hipMemcpyAsync(c->h_z + (dsl * 2 + m), c->z_offs[m].as<uint32_t>() + nblk, 4, hipMemcpyDeviceToHost, s));
                }
                if (!paired) c->h_z[dsl * 2 + 1] = 0;
                HIP_OK(hipEventRecord(c->ev_z[dsl], s));
                sh.p1 = c->z_out[dsl][0].as<char>(); sh.p2 = paired ? c->z_out[dsl][1].as<char>() : nullptr;
            }
            HIP_OK(hipEventRecord(c->ev_made[dsl], s));                             // the batch's text (its blocks) is complete ...
            if (bgzf) { if (have_pending) ship(pending); pending = sh; have_pending = true; }
            else ship(sh);
            ++bi;
        }
        tot1 += b1; tot2 += b2;
    }
    if (have_pending) ship(pending);
    if (tg.seg_off1) {                                                               // record order = batch order: the bytes before each batch
        std::vector<uint64_t> pre1(nbatch + 1, 0), pre2(nbatch + 1, 0);
        for (uint32_t b = 0; b < nbatch; ++b) { pre1[b + 1] = pre1[b] + bb1[b]; pre2[b + 1] = pre2[b] + bb2[b]; }
        for (size_t j = 0; j < bpair.size(); ++j) { (*tg.seg_off1)[j] = tot1; if (tg.seg_off2) (*tg.seg_off2)[j] = tot2; }   // segments that start behind the last pair
        for (const SegAt& a : seg_at) { (*tg.seg_off1)[a.seg] = pre1[a.b] + a.o1; if (tg.seg_off2) (*tg.seg_off2)[a.seg] = pre2[a.b] + a.o2; }
    }
    // pairs produced = planned - holes; a hole arises only when > 1000 insert sizes in a row miss [readLength, ampliconLen]
    // (Amplicon.cpp:484-489): k_plan_pairs counted them on the device
    { Mail m; m.add(c->flags.p, 4, 30); m.add(c->dsums.as<unsigned long long>() + DS_HOLES, 8, 2); mail_post(c, m, true); }   // flags + hole count land before the final synchronize: no second round trip
    HIP_OK(hipStreamSynchronize(s));
    if (to_sink) { HIP_OK(hipStreamSynchronize(c->copy_stream)); guard.p = nullptr; if (!c->pipe->finish()) throw ScsError(SCS_EIO, "sink aborted"); }
    mail_wait(c); flags_eval(c);
    if (c->want_cks && !tg.device && nbatch) { c->cks.assign((size_t)nbatch * 2, 0); HIP_OK(hipMemcpyAsync(c->cks.data(), c->d_cks.p, (size_t)nbatch * 16, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s)); }
    pairs_written = P - c->h_rb[2];
    c->tm_reads.collect(); c->tm_indels.collect();
    c->st.pairs_written = pairs_written; c->st.reads_written = paired ? 2 * pairs_written : pairs_written;
    c->st.fastq_bytes[0] = tot1; c->st.fastq_bytes[1] = tot2;
    c->st.sink_bytes[0] = to_sink ? sunk1 : 0; c->st.sink_bytes[1] = to_sink ? sunk2 : 0;
    // 1526 B per created amplicon + per pair (insert size + FASTQ bytes of both records)
    const uint64_t per_pair_tmpl = paired ? (uint64_t)(c->cfg.isize + 1) : (uint64_t)L;
    c->st.algorithmic_bytes = 1526ull * (c->st.semi_amplicons + c->st.full_amplicons) + pairs_written * per_pair_tmpl + tot1 + tot2;
    if (n1_out) *n1_out = tot1; if (n2_out) *n2_out = tot2; if (pairs_out) *pairs_out = pairs_written;
    if (seam_env("SCS_PHASE_CLOCK")) phase_clock_report();                         // (prints only in a -DSCS_PHASE_CLOCK build)
    if (c->cfg.verbose) fprintf(stderr, "\nReads generation done!\n");
}

template <class F>
int guarded(scs_ctx* c, F f) {
    if (!c) return SCS_EINVAL;
    try { if (c->cfg.device >= 0) (void)hipSetDevice(c->cfg.device); f(); return SCS_OK; }
    catch (const ScsError& e) { c->err = e.what(); return e.code; }
    catch (const std::exception& e) { c->err = e.what(); return SCS_EIO; }
}

}  // namespace

// =================================================================== C ABI
extern "C" {

void scs_default_config(scs_config* cfg) {
    memset(cfg, 0, sizeof *cfg);
    cfg->device = 0; cfg->stream = nullptr; cfg->seed = 1;
    cfg->primers = 100000; cfg->gamma = 1e-9; cfg->coverage = 5; cfg->isize = 260; cfg->paired = 1;
    cfg->ber = 3.4e-4; cfg->amplicon_min_len = 1000; cfg->amplicon_max_len = 2000; cfg->frag_size = 1000;
    cfg->frag_min = 10000; cfg->frag_max = 100000; cfg->shard_rank = 0; cfg->shard_count = 1; cfg->verbose = 0;
}

int scs_create(const scs_config* cfg, scs_ctx** out) {
    if (!cfg || !out) { g_create_error = "scs_create: null argument"; return SCS_EINVAL; }
    *out = nullptr;
    if (cfg->primers < 1000) { g_create_error = "Error: the value of parameter \"primers\" should be at least 1000!"; return SCS_EINVAL; }
    if (cfg->gamma <= 0 || cfg->gamma > 1e-8) { g_create_error = "Error: the value of parameter \"gamma\" should be in 0~1e-8!"; return SCS_EINVAL; }
    if (cfg->coverage <= 0) { g_create_error = "Error: sequencing coverage not properly specified!"; return SCS_EINVAL; }
    if (cfg->shard_count < 1 || cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_count) { g_create_error = "scs_create: bad shard rank/count"; return SCS_EINVAL; }
    if (cfg->amplicon_max_len > 2047 || cfg->frag_max > 131071 || cfg->amplicon_min_len < 64) { g_create_error = "scs_create: amplicon/fragment size outside the packed-record limits"; return SCS_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device) {
        g_create_error = "scs_create: no HIP device " + std::to_string(cfg->device) + " (this library has no CPU fallback)"; return SCS_EDEVICE;
    }
    scs_ctx* c = new scs_ctx; c->cfg = *cfg;
    try {
        HIP_OK(hipSetDevice(cfg->device));
        if (cfg->stream) c->stream = (hipStream_t)cfg->stream; else { HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
        c->key = RngKey{(uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32)};
        for (KernelTimer* t : {&c->tm_errscan, &c->tm_errscan_f, &c->tm_reads, &c->tm_attach, &c->tm_indels, &c->tm_attach_f}) t->gate = &c->timing_gate;
        c->flags.reserve(256, c->stream); HIP_OK(hipMemsetAsync(c->flags.p, 0, 256, c->stream));
        c->dsums.reserve(256, c->stream); HIP_OK(hipMemsetAsync(c->dsums.p, 0, 256, c->stream));
        c->d_tot.reserve(256, c->stream);
        HIP_OK(hipHostMalloc((void**)&c->h_rb, 256, hipHostMallocMapped | hipHostMallocCoherent)); memset(c->h_rb, 0, 256);
        HIP_OK(hipHostGetDevicePointer((void**)&c->d_rb, c->h_rb, 0));
        HIP_OK(hipStreamSynchronize(c->stream));
    } catch (const std::exception& e) { g_create_error = e.what(); delete c; return SCS_EDEVICE; }
    *out = c; return SCS_OK;
}

void scs_destroy(scs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->d_tables, &c->t_gap, &c->t_qcompact, &c->t_guide, &c->t_ring1, &c->t_ring2, &c->t_ring1u, &c->t_ring2u, &c->t_subs1, &c->t_subs2, &c->t_qual, &c->t_ins, &c->t_del, &c->t_isize, &c->d_subs1, &c->d_subs2, &c->d_qual, &c->d_ins, &c->d_del,
                      &c->d_isize, &c->d_gcmeans, &c->genome, &c->genome2, &c->gx_gc_bits, &c->gx_n_bits, &c->gx_gc_cnt, &c->gx_n_cnt, &c->gx_gc_pref, &c->gx_n_pref, &c->d_binom, &c->df_blob, &c->df_primers, &c->df_hasn, &c->primer_cnt, &c->primer_delta, &c->primer_cut, &c->primer_gdelta, &c->st_eidx, &c->st_etype, &c->st_estart, &c->st_info, &c->st_list, &c->st_sorted, &c->st_tmp, &c->att_wave_first,
                      &c->slots, &c->slot_tmpl, &c->slots_fr, &c->slot_tmpl_fr, &c->valid, &c->valid_off, &c->valid_f, &c->valid_off_f, &c->scan_tmp, &c->flags, &c->weights, &c->read_numbers,
                      &c->pair_off, &c->pairs, &c->odd_before, &c->a_part, &c->a_tp, &c->a_probs, &c->a_quota, &c->a_poff, &c->a_plan, &c->a_crn, &c->a_scratch, &c->a_brow, &c->a_bmap, &c->a_send, &c->a_gath, &c->a_odd, &c->d_hostred, &c->d_tot, &c->d_stage, &c->d_mail, &c->budget_f, &c->budget_s, &c->poisson_part, &c->slot_off_f,
                      &c->slot_off_s, &c->dsums, &c->slot_b, &c->slot_q, &c->lens, &c->ev_hdr, &c->ev_dat, &c->sizes1, &c->sizes2, &c->off1, &c->off2, &c->out1, &c->out2, &c->out1b, &c->out2b, &c->rl_cls, &c->rl_pos, &c->rl_lists, &c->d_bounds, &c->d_cks}) b->release();
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->errs_stream) { (void)hipStreamDestroy(c->errs_stream); (void)hipEventDestroy(c->ev_att); (void)hipEventDestroy(c->ev_errs); }
    if (c->pre_stream) { (void)hipStreamDestroy(c->pre_stream); (void)hipEventDestroy(c->ev_plan); for (int k = 0; k < 2; ++k) { (void)hipEventDestroy(c->ev_pre[k]); (void)hipEventDestroy(c->ev_free[k]); } }
    for (int k = 0; k < 2; ++k) { if (c->ev_made[k]) (void)hipEventDestroy(c->ev_made[k]); if (c->ev_d2h[k]) (void)hipEventDestroy(c->ev_d2h[k]); }
    c->reads_side.release();
    for (int k = 0; k < 2; ++k) { c->z_plan[k].release(); c->z_sizes[k].release(); c->z_offs[k].release(); c->z_out[k][0].release(); c->z_out[k][1].release(); if (c->ev_z[k]) (void)hipEventDestroy(c->ev_z[k]); }
    c->z_crc.release(); if (c->h_z) (void)hipHostFree(c->h_z);
    c->semis.release(); c->fulls.release();
    for (KernelTimer* t : {&c->tm_errscan, &c->tm_errscan_f, &c->tm_reads, &c->tm_attach, &c->tm_indels, &c->tm_attach_f}) t->release();
    if (c->h_rb) (void)hipHostFree(c->h_rb);
    if (c->h_frag) (void)hipHostFree(c->h_frag);
    if (c->pipe) { c->pipe->release(); delete c->pipe; }
    if (c->rccl) rccl_destroy(c->rccl);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* scs_last_error(const scs_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int scs_set_seed(scs_ctx* c, uint64_t seed) {
    if (!c) return SCS_EINVAL;
    c->cfg.seed = seed; c->key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)}; return SCS_OK;
}

int scs_load_profile(scs_ctx* c, const char* path) { return guarded(c, [&] { if (!path) throw ScsError(SCS_EINVAL, "null path"); do_load_profile(c, path); }); }
int scs_read_length(const scs_ctx* c) { return c && c->have_profile ? c->prof.read_length : -1; }

int scs_load_genome_fasta(scs_ctx* c, const char* path) {
    return guarded(c, [&] {
        if (!path) throw ScsError(SCS_EINVAL, "null path");
        if (seam_env("SCS_HOST_FASTA")) { load_fasta(path, c->recs, true); stage_genome(c); }   // the host parser (what scs_fasta_probe checks); debugging aid
        else if (c->cfg.shard_count > 1 && !seam_env("SCS_STAGE_WHOLE") && stage_fasta_slice(c, fasta_plain_path(path))) {
            if (c->cfg.verbose) fprintf(stderr, "(shard %d of %d: %llu of %llu bases staged)\n", c->cfg.shard_rank, c->cfg.shard_count, (unsigned long long)c->slice_len, (unsigned long long)c->genome_bases);
        }
        else stage_fasta_on_device(c, path);
        if (c->cfg.verbose) fprintf(stderr, "\nReference sequence was loaded from file %s\n", path);
    });
}
int scs_upload_genome(scs_ctx* c, int n, const char* const* names, const char* const* seqs, const uint64_t* lens) {
    return guarded(c, [&] {
        if (n <= 0 || !names || !seqs || !lens) throw ScsError(SCS_EINVAL, "scs_upload_genome: bad arguments");
        c->recs.resize(n);
        for (int i = 0; i < n; ++i) encode_record(names[i], seqs[i], lens[i], c->recs[i]);
        stage_genome(c);
    });
}
int scs_upload_genome_device(scs_ctx* c, int n, const char* const* names, const uint64_t* lens, const void* d_bases) {
    return guarded(c, [&] {
        if (n <= 0 || !names || !lens || !d_bases) throw ScsError(SCS_EINVAL, "scs_upload_genome_device: bad arguments");
        c->recs.resize(n);
        for (int i = 0; i < n; ++i) encode_record(names[i], nullptr, 0, c->recs[i]);
        stage_genome(c, d_bases, lens);
    });
}
// `scssim simuvars` (src/scssim.cpp:33-38: Genome::loadData + Genome::saveSequence) on the data plane
int scs_simuvars(scs_ctx* c, const char* ref_fasta, const char* snp_file, const char* var_file, const char* out_fasta) {
    return guarded(c, [&] {
        if (!ref_fasta) throw ScsError(SCS_EINVAL, "reference sequence file not specified!");
        hipStream_t s = c->stream;
        std::vector<FastaRecord> ref; load_fasta(ref_fasta, ref, true);                // Genome::loadRefSeq (Genome.cpp:176-195): index names, .fai beside the file
        if (c->cfg.verbose) fprintf(stderr, "\nReference sequence was loaded from file %s\n", ref_fasta);
        std::vector<SvChrom> chroms; uint64_t rtot = 0;
        for (auto& r : ref) { chroms.push_back(SvChrom{r.name, rtot, (uint64_t)r.code.size()}); rtot += r.code.size(); }
        DevBuf d_ref, d_lit, d_pc, d_sb;
        struct Rel { DevBuf* b[4]; ~Rel() { for (DevBuf* x : b) x->release(); } } rel{{&d_ref, &d_lit, &d_pc, &d_sb}};
        d_ref.reserve(std::max<uint64_t>(rtot, 16), s);
        for (size_t i = 0; i < ref.size(); ++i) if (!ref[i].code.empty()) HIP_OK(hipMemcpyAsync((uint8_t*)d_ref.p + chroms[i].off, ref[i].code.data(), ref[i].code.size(), hipMemcpyHostToDevice, s));
        SvPlan P;
        try { simuvars_plan(chroms, snp_file ? snp_file : "", var_file ? var_file : "", c->cfg.verbose != 0, P); }
        catch (const std::exception& e) { HIP_OK(hipStreamSynchronize(s)); throw ScsError(SCS_EIO, e.what()); }
        if (P.pieces.size() > 0xFFFFFFF0ull || P.substs.size() > 0xFFFFFFF0ull) throw ScsError(SCS_EOVERFLOW, "simuvars: too many edits");
        upload(d_pc, P.pieces, s); upload(d_sb, P.substs, s);
        d_lit.reserve(std::max<size_t>(P.literals.size(), 16), s);
        if (!P.literals.empty()) HIP_OK(hipMemcpyAsync(d_lit.p, P.literals.data(), P.literals.size(), hipMemcpyHostToDevice, s));
        c->genome.reserve(std::max<uint64_t>(P.total, 16), s);
        launch_sv_build(s, d_ref.as<uint8_t>(), d_lit.as<uint8_t>(), d_pc.as<SvPiece>(), (uint32_t)P.pieces.size(), d_sb.as<SvSubst>(), (uint32_t)P.substs.size(), c->genome.as<uint8_t>(), P.total);
        HIP_OK(hipStreamSynchronize(s));                                                 // the host vectors behind the uploads may go
        { const hipError_t le = take_launch_error(); if (le != hipSuccess) throw ScsError(SCS_EDEVICE, std::string("simuvars kernel launch failed: ") + hipGetErrorString(le)); }
        if (out_fasta && *out_fasta) {                                                   // Genome::saveSequence's file: >chr_hap_len, 100 columns (Genome.cpp:365-381)
            FILE* o = fopen(out_fasta, "w");
            if (!o) throw ScsError(SCS_EIO, std::string("can not open file ") + out_fasta);
            const size_t chunk = 64u << 20; char* hb = nullptr; HIP_OK(hipHostMalloc((void**)&hb, chunk, hipHostMallocDefault));
            std::vector<char> line_buf; uint64_t off = 0; bool okw = true;
            for (size_t r = 0; r < P.rec_names.size() && okw; ++r) {
                okw = fprintf(o, ">%s\n", P.rec_names[r].c_str()) > 0;
                uint64_t col = 0;
                for (uint64_t done = 0; done < P.rec_lens[r] && okw; done += chunk) {
                    const size_t n = (size_t)std::min<uint64_t>(chunk, P.rec_lens[r] - done);
                    HIP_OK(hipMemcpyAsync(hb, (uint8_t*)c->genome.p + off + done, n, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
                    line_buf.clear(); line_buf.reserve(n + n / 100 + 2);
                    for (size_t i = 0; i < n;) { const size_t take = (size_t)std::min<uint64_t>(100 - col, n - i); line_buf.insert(line_buf.end(), hb + i, hb + i + take); i += take; col += take; if (col == 100) { line_buf.push_back('\n'); col = 0; } }
                    okw = fwrite(line_buf.data(), 1, line_buf.size(), o) == line_buf.size();
                }
                if (col && okw) okw = fputc('\n', o) != EOF;
                off += P.rec_lens[r];
            }
            (void)hipHostFree(hb);
            if (fclose(o) != 0 || !okw) throw ScsError(SCS_EIO, std::string("writing ") + out_fasta + " failed");
        }
        c->recs.resize(P.rec_names.size());
        for (size_t i = 0; i < P.rec_names.size(); ++i) encode_record(P.rec_names[i].c_str(), nullptr, 0, c->recs[i]);
        stage_genome(c, c->genome.p, P.rec_lens.data());                                 // encode + bit index in place: ready for scs_create_frags
    });
}
int scs_create_frags(scs_ctx* c) { return guarded(c, [&] { double t = now_s(); do_create_frags(c); c->st.t_stage[1] = now_s() - t; }); }
int scs_amplify(scs_ctx* c) { return guarded(c, [&] { double t = now_s(); do_amplify(c); c->st.t_stage[2] = now_s() - t; }); }
int scs_allocate_reads(scs_ctx* c, uint64_t reads) { return guarded(c, [&] { do_allocate(c, reads); }); }
int scs_yield_reads(scs_ctx* c, scs_sink_fn sink, void* user) {
    return guarded(c, [&] {
        double t = now_s(); CallbackSink cb(sink, user);
        OutTarget tg{false, nullptr, nullptr, 0, 0, sink ? &cb : nullptr}; do_yield(c, tg, nullptr, nullptr, nullptr); c->st.t_stage[5] = now_s() - t;
    });
}
int scs_yield_reads_device(scs_ctx* c, void* d1, size_t cap1, void* d2, size_t cap2, uint64_t* n1, uint64_t* n2, uint64_t* pairs) {
    return guarded(c, [&] {
        if (!d1 || (c->cfg.paired && !d2)) throw ScsError(SCS_EINVAL, "scs_yield_reads_device: null output buffer");
        double t = now_s(); OutTarget tg{true, (char*)d1, (char*)d2, cap1, cap2, nullptr}; do_yield(c, tg, n1, n2, pairs); c->st.t_stage[5] = now_s() - t;
    });
}
// SeqWriter (lib/seqwriter/SeqWriter.cpp:12-64).  writers <= 1: the reference's files <prefix>_1.fq / _2.fq (.fq) -- a shard of a
// sharded job: <prefix>.r<rank>_1.fq ... + <prefix>.r<rank>.idx.  writers = K > 1: K part files per mate, each a contiguous
// range of the job's (shard's) records written by its own thread, + <base>.parts (scs_comm.h: FastqParts).
int scs_yield_reads_files_ex(scs_ctx* c, const char* prefix, int writers, int generations, int bgzf) {
    return guarded(c, [&] {
        if (!prefix || !*prefix) throw ScsError(SCS_EINVAL, "scs_yield_reads_files: no output prefix");
        if (writers > 64 || generations > 64 || (int64_t)std::max(1, writers) * std::max(1, generations) > 99) throw ScsError(SCS_EINVAL, "scs_yield_reads_files: at most 64 writers and 99 parts");
        const bool pe = c->cfg.paired != 0, shard = c->cfg.shard_count > 1; const std::string pre = prefix;
        const std::string base = shard ? shard_base(pre, c->cfg.shard_rank) : pre;
        FastqParts files; std::string err;
        if (!files.open(base, pe, std::max(1, writers), std::max(1, generations), bgzf ? ".fq.gz" : ".fq", bgzf != 0, err)) throw ScsError(SCS_EIO, err);
        std::vector<uint64_t> so1, so2;
        double t = now_s(); OutTarget tg{false, nullptr, nullptr, 0, 0, &files}; tg.bgzf = bgzf != 0;
        if (shard && !bgzf) { tg.seg_off1 = &so1; tg.seg_off2 = &so2; }              // (byte ranges of compressed shards cannot be spliced: BGZF shards stay shards)
        do_yield(c, tg, nullptr, nullptr, nullptr);
        if (!files.close(err)) throw ScsError(SCS_EIO, err);
        if (shard && !bgzf && !write_shard_index(shard_index_path(pre, c->cfg.shard_rank), so1, so2, err)) throw ScsError(SCS_EIO, err);
        c->st.t_stage[5] = now_s() - t;
    });
}
int scs_yield_reads_files(scs_ctx* c, const char* prefix, int writers) { return scs_yield_reads_files_ex(c, prefix, writers, 1, 0); }
int scs_merge_fastq_parts(const char* prefix, int paired, int keep_parts, char* errbuf, size_t errlen) {
    if (!prefix) return SCS_EINVAL;
    std::string err;
    if (merge_parts(prefix, paired != 0, ".fq", keep_parts != 0, err)) return SCS_OK;
    if (errbuf && errlen) { strncpy(errbuf, err.c_str(), errlen - 1); errbuf[errlen - 1] = 0; }
    return SCS_EIO;
}
int scs_merge_fastq_shards(const char* prefix, int nranks, int paired, int keep_shards, char* errbuf, size_t errlen) {
    if (!prefix) return SCS_EINVAL;
    std::string err;
    if (merge_shards(prefix, nranks, paired != 0, keep_shards != 0, err)) return SCS_OK;
    if (errbuf && errlen) { strncpy(errbuf, err.c_str(), errlen - 1); errbuf[errlen - 1] = 0; }
    return SCS_EIO;
}
int scs_comm_unique_id(void* id_out) {
    std::string err;
    if (!id_out) return SCS_EINVAL;
    if (rccl_unique_id(id_out, err)) { g_create_error = err; return SCS_EDEVICE; }
    return SCS_OK;
}
static int rccl_allreduce_hook(void* user, void* d_vals, uint64_t n, int elem_bytes) {
    scs_ctx* c = (scs_ctx*)user; std::string err;
    if (rccl_allreduce_sum(c->rccl, d_vals, n, elem_bytes, c->stream, err)) { c->err = err; return 1; }
    return 0;
}
static int rccl_allgather_hook(void* user, const void* d_send, void* d_recv, uint64_t bytes) {
    scs_ctx* c = (scs_ctx*)user; std::string err;
    if (rccl_allgather(c->rccl, d_send, d_recv, bytes, c->stream, err)) { c->err = err; return 1; }
    return 0;
}
int scs_comm_init(scs_ctx* c, const void* id, int rank, int nranks) {
    return guarded(c, [&] {
        if (!id || nranks < 1 || rank < 0 || rank >= nranks) throw ScsError(SCS_EINVAL, "scs_comm_init: bad arguments");
        if (rank != c->cfg.shard_rank || nranks != c->cfg.shard_count) throw ScsError(SCS_EINVAL, "scs_comm_init: rank / size differ from the ctx's shard_rank / shard_count");
        if (c->rccl) { rccl_destroy(c->rccl); c->rccl = nullptr; }
        std::string err;
        c->rccl = rccl_init(id, rank, nranks, err);
        if (!c->rccl) throw ScsError(SCS_EDEVICE, err);
        c->allreduce_dev = rccl_allreduce_hook; c->allgather_dev = rccl_allgather_hook; c->coll_dev_user = c;
    });
}
int scs_comm_count(const scs_ctx* c) { return c && c->rccl ? rccl_count(c->rccl) : 0; }
int scs_comm_abort(scs_ctx* c) { if (!c) return SCS_EINVAL; if (c->rccl) rccl_abort(c->rccl); return SCS_OK; }
int scs_run_genreads(scs_ctx* c, scs_sink_fn sink, void* user) {
    int rc; double t = now_s();
    if ((rc = scs_create_frags(c))) return rc;
    if ((rc = scs_amplify(c))) return rc;
    if ((rc = scs_allocate_reads(c, 0))) return rc;
    if ((rc = scs_yield_reads(c, sink, user))) return rc;
    c->st.t_stage[7] = now_s() - t; return SCS_OK;
}
int scs_set_collectives(scs_ctx* c, scs_allreduce_fn ar, scs_allgatherv_fn ag, void* user) {
    if (!c) return SCS_EINVAL;
    c->allreduce = ar; c->allgatherv = ag; c->coll_user = user; return SCS_OK;
}
int scs_set_collectives_device(scs_ctx* c, scs_allreduce_dev_fn ar, scs_allgather_dev_fn ag, void* user) {
    if (!c) return SCS_EINVAL;
    c->allreduce_dev = ar; c->allgather_dev = ag; c->coll_dev_user = user; return SCS_OK;
}
int scs_set_batch_checksums(scs_ctx* c, int on) { if (!c) return SCS_EINVAL; c->want_cks = on != 0; return SCS_OK; }
int scs_batch_checksums(const scs_ctx* c, uint64_t* out, size_t cap, size_t* n_batches) {
    if (!c || !n_batches) return SCS_EINVAL;
    *n_batches = c->cks.size() / 2;
    if (out) memcpy(out, c->cks.data(), std::min(cap, c->cks.size()) * 8);
    return SCS_OK;
}
int scs_get_stats(const scs_ctx* c, scs_stats* out) { if (!c || !out) return SCS_EINVAL; *out = c->st; return SCS_OK; }

int scs_kernel_time(const scs_ctx* c, int which, const char** name, uint64_t* launches, double* ms, uint64_t* units) {
    if (!c) return SCS_EINVAL;
    const KernelTimer* t[] = {&c->tm_errscan, &c->tm_errscan_f, &c->tm_reads, &c->tm_attach, &c->tm_indels, &c->tm_attach_f};
    if (which < 0 || which >= 6) return SCS_EINVAL;
    if (name) *name = t[which]->name; if (launches) *launches = t[which]->launches; if (ms) *ms = t[which]->ms; if (units) *units = t[which]->units;
    return SCS_OK;
}

int scs_set_kernel_timing(scs_ctx* c, unsigned mask, unsigned every) {
    if (!c || every == 0) return SCS_EINVAL;
    KernelTimer* t[] = {&c->tm_errscan, &c->tm_errscan_f, &c->tm_reads, &c->tm_attach, &c->tm_indels, &c->tm_attach_f};
    for (int i = 0; i < 6; ++i) t[i]->on = (mask >> i) & 1u;
    c->timing_every = every; c->amplify_calls = 0; c->yield_calls = 0;
    return SCS_OK;
}

int scs_predict_batch(scs_ctx* c, const uint8_t* windows, size_t n_reads, const uint64_t* uids, const uint32_t* attempts, const uint8_t* is_read1,
                      char* out_bases, char* out_quals, int32_t* out_len, int out_stride) {
    return guarded(c, [&] {
        if (!c->have_profile) throw ScsError(SCS_EINVAL, "scs_predict_batch: load a profile first");
        if (n_reads > 0x7FFFFFFFull) throw ScsError(SCS_EINVAL, "too many reads");
        hipStream_t s = c->stream; const uint32_t L = (uint32_t)c->prof.read_length, slot = ((L + 64 + 63) / 64) * 64, n = (uint32_t)n_reads;
        if (out_stride < (int)slot) throw ScsError(SCS_EINVAL, "out_stride must be >= " + std::to_string(slot));
        DevBuf dw, du, da, dr; dw.reserve(std::max<size_t>((size_t)n * L, 16), s); du.reserve(std::max<size_t>((size_t)n * 8, 16), s);
        da.reserve(std::max<size_t>((size_t)n * 4, 16), s); dr.reserve(std::max<size_t>(n, 16), s);
        HIP_OK(hipMemcpyAsync(dw.p, windows, (size_t)n * L, hipMemcpyHostToDevice, s)); HIP_OK(hipMemcpyAsync(du.p, uids, (size_t)n * 8, hipMemcpyHostToDevice, s));
        HIP_OK(hipMemcpyAsync(da.p, attempts, (size_t)n * 4, hipMemcpyHostToDevice, s)); HIP_OK(hipMemcpyAsync(dr.p, is_read1, n, hipMemcpyHostToDevice, s));
        c->slot_b.reserve((size_t)n * slot, s); c->slot_q.reserve((size_t)n * slot, s); c->lens.reserve(std::max<size_t>((size_t)n * 4, 16), s);
        launch_predict_windows(s, dw.as<uint8_t>(), n, du.as<uint64_t>(), da.as<uint32_t>(), dr.as<uint8_t>(), c->dtb, c->d_tables.as<DevTables>(), c->key, slot, c->slot_b.as<char>(),
                               c->slot_q.as<char>(), c->lens.as<uint32_t>(), c->flags.as<uint32_t>());
        std::vector<char> hb((size_t)n * slot), hq((size_t)n * slot); std::vector<uint32_t> hl(n);
        HIP_OK(hipMemcpyAsync(hb.data(), c->slot_b.p, hb.size(), hipMemcpyDeviceToHost, s)); HIP_OK(hipMemcpyAsync(hq.data(), c->slot_q.p, hq.size(), hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(hl.data(), c->lens.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        dw.release(); du.release(); da.release(); dr.release();
        check_flags(c);
        for (uint32_t i = 0; i < n; ++i) { out_len[i] = (int32_t)hl[i]; memcpy(out_bases + (size_t)i * out_stride, hb.data() + (size_t)i * slot, hl[i]); memcpy(out_quals + (size_t)i * out_stride, hq.data() + (size_t)i * slot, hl[i]); }
    });
}

int scs_philox_batch(scs_ctx* c, const uint32_t* ctr, size_t n, const uint32_t* key, uint32_t* out) {
    return guarded(c, [&] {
        hipStream_t s = c->stream; DevBuf a, b; a.reserve(std::max<size_t>(n * 16, 16), s); b.reserve(std::max<size_t>(n * 16, 16), s);
        HIP_OK(hipMemcpyAsync(a.p, ctr, n * 16, hipMemcpyHostToDevice, s));
        launch_philox(s, a.as<uint32_t>(), (uint32_t)n, RngKey{key[0], key[1]}, b.as<uint32_t>());
        HIP_OK(hipMemcpyAsync(out, b.p, n * 16, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s)); a.release(); b.release();
    });
}
int scs_detlog_batch(scs_ctx* c, const double* x, size_t n, double* out) {
    return guarded(c, [&] {
        hipStream_t s = c->stream; DevBuf a, b; a.reserve(std::max<size_t>(n * 8, 16), s); b.reserve(std::max<size_t>(n * 8, 16), s);
        HIP_OK(hipMemcpyAsync(a.p, x, n * 8, hipMemcpyHostToDevice, s));
        launch_detlog(s, a.as<double>(), (uint32_t)n, b.as<double>());
        HIP_OK(hipMemcpyAsync(out, b.p, n * 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s)); a.release(); b.release();
    });
}

int scs_download_amplicons(scs_ctx* c, int kind, uint32_t* parent, uint32_t* spos, uint32_t* len, uint32_t* gc, uint32_t* primers, uint64_t* uid,
                           uint32_t* errs, uint32_t* nerr) {
    return guarded(c, [&] {
        if (!c->amplified) throw ScsError(SCS_EINVAL, "scs_download_amplicons: call scs_amplify first");
        AmpStore& A = kind == 0 ? c->semis : c->fulls; const uint32_t n = A.n; hipStream_t s = c->stream; DevAmps v = A.view();
        std::vector<uint32_t> hsl(n), hp(n); std::vector<uint16_t> hgc(n), hpr(n); std::vector<uint64_t> hu(n), he(n);
        if (n) {
            HIP_OK(hipMemcpyAsync(hp.data(), v.parent, (size_t)n * 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipMemcpyAsync(hsl.data(), v.sl, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            HIP_OK(hipMemcpyAsync(hgc.data(), v.gc, (size_t)n * 2, hipMemcpyDeviceToHost, s)); HIP_OK(hipMemcpyAsync(hpr.data(), v.primers, (size_t)n * 2, hipMemcpyDeviceToHost, s));
            HIP_OK(hipMemcpyAsync(hu.data(), v.uid, (size_t)n * 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipMemcpyAsync(he.data(), v.errs, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        }
        uint32_t used = 0; HIP_OK(hipMemcpyAsync(&used, A.pool_head.p, 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
        std::vector<uint32_t> pool(std::min(used, A.pool_cap));
        if (!pool.empty()) { HIP_OK(hipMemcpyAsync(pool.data(), A.pool.p, pool.size() * 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s)); }
        for (uint32_t i = 0; i < n; ++i) {
            if (parent) parent[i] = hp[i]; if (spos) spos[i] = sl_spos(hsl[i]); if (len) len[i] = sl_len(hsl[i]); if (gc) gc[i] = hgc[i];
            if (primers) primers[i] = hpr[i]; if (uid) uid[i] = hu[i];
            uint32_t cnt = 0; uint32_t e4[4] = {0, 0, 0, 0};
            if (he[i] & ERR_OVERFLOW_BIT) { const uint32_t off = (uint32_t)he[i]; cnt = (uint32_t)(he[i] >> 32) & 0xFFFF; for (uint32_t k = 0; k < std::min(cnt, 4u); ++k) { uint32_t e = pool[off + k]; e4[k] = (err_pos(e) << 3) | err_alt(e); } }
            else for (int k = 0; k < 4; ++k) { uint32_t e = (uint32_t)(he[i] >> (16 * k)) & 0xFFFF; if (e) e4[cnt++] = (err_pos(e) << 3) | err_alt(e); }
            if (errs) memcpy(errs + 4 * (size_t)i, e4, 16); if (nerr) nerr[i] = cnt;
        }
    });
}
int scs_gpu_local_cpus(int device, int* cpus, int cap) {
    try { const std::vector<int> v = gpu_local_cpus(device); for (int i = 0; i < (int)v.size() && i < cap && cpus; ++i) cpus[i] = v[(size_t)i]; return (int)v.size(); } catch (...) { return 0; }
}
const char* scs_test_seam(const char* name) { return name ? seam_env(name) : nullptr; }
int scs_download_primer_stock(scs_ctx* c, int64_t* stock) {
    return guarded(c, [&] {
        if (!stock) throw ScsError(SCS_EINVAL, "null pointer");
        if (!c->amplified) throw ScsError(SCS_EINVAL, "scs_download_primer_stock: call scs_amplify first");
        HIP_OK(hipMemcpyAsync(stock, c->primer_cnt.p, 65536 * 8, hipMemcpyDeviceToHost, c->stream)); HIP_OK(hipStreamSynchronize(c->stream));
    });
}
int scs_download_read_numbers(scs_ctx* c, uint32_t* rn) {
    return guarded(c, [&] {
        if (!c->allocated) throw ScsError(SCS_EINVAL, "call scs_allocate_reads first");
        if (c->fulls.n) HIP_OK(hipMemcpyAsync(rn, c->read_numbers.p, (size_t)c->fulls.n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    });
}

int scs_fasta_probe(const char* path, int* n_records, uint64_t* total_bases, uint64_t* checksum, char* names_buf, size_t names_len, char* errbuf, size_t errlen) {
    if (!path) return SCS_EINVAL;
    std::vector<FastaRecord> recs;
    try { load_fasta(path, recs); }
    catch (const std::exception& e) { if (errbuf && errlen) { strncpy(errbuf, e.what(), errlen - 1); errbuf[errlen - 1] = 0; } return SCS_EIO; }
    uint64_t tot = 0, h = 1469598103934665603ull; std::string names;
    for (auto& r : recs) {
        tot += r.code.size(); names += r.name; names += '\n';
        for (uint8_t b : r.code) { h ^= (uint64_t)(b >= 'a' && b <= 'z' ? b - 32 : b); h *= 1099511628211ull; }
    }
    if (n_records) *n_records = (int)recs.size(); if (total_bases) *total_bases = tot; if (checksum) *checksum = h;
    if (names_buf && names_len) { strncpy(names_buf, names.c_str(), names_len - 1); names_buf[names_len - 1] = 0; }
    return SCS_OK;
}
int scs_devbuf_probe(int device, uint64_t first_bytes, uint64_t second_bytes, uint64_t* caps, int* in_place) {
    if (!caps) return SCS_EINVAL;
    DevBuf b;
    try {
        HIP_OK(hipSetDevice(device));
        b.reserve((size_t)first_bytes, nullptr); caps[0] = b.cap;
        const void* at = b.p;
        b.reserve((size_t)second_bytes, nullptr); caps[1] = b.cap;
        if (in_place) *in_place = b.p == at ? 1 : 0;
        b.release();
        return SCS_OK;
    } catch (const std::exception& e) { b.release(); g_create_error = e.what(); return SCS_EDEVICE; }
}
// host-only: the simuvars plan applied to the host copy of the reference, folded into a checksum of the FASTA text that
// scs_simuvars would write (test seam for the planner; the product builds the sequences on the device)
int scs_simuvars_probe(const char* ref_fasta, const char* snp_file, const char* var_file, int* n_records, uint64_t* total_bases, uint64_t* checksum, char* errbuf, size_t errlen) {
    if (!ref_fasta) return SCS_EINVAL;
    try {
        std::vector<FastaRecord> ref; load_fasta(ref_fasta, ref);
        std::vector<SvChrom> chroms; std::vector<uint8_t> flat;
        for (auto& r : ref) { chroms.push_back(SvChrom{r.name, (uint64_t)flat.size(), (uint64_t)r.code.size()}); flat.insert(flat.end(), r.code.begin(), r.code.end()); }
        SvPlan P; simuvars_plan(chroms, snp_file ? snp_file : "", var_file ? var_file : "", false, P);
        std::vector<uint8_t> out(P.total);
        for (const SvPiece& pc : P.pieces) for (uint32_t i = 0; i < pc.len; ++i) { uint8_t ch = pc.lit ? (uint8_t)P.literals[pc.src + i] : flat[pc.src + i]; out[pc.dst + i] = (uint8_t)(ch >= 'a' && ch <= 'z' ? ch - 32 : ch); }
        for (const SvSubst& sb : P.substs) out[sb.dst] = (uint8_t)sb.ch;
        uint64_t h = 1469598103934665603ull, off = 0;
        auto eat = [&](const char* p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)p[i]; h *= 1099511628211ull; } };
        for (size_t r = 0; r < P.rec_names.size(); ++r) {
            const std::string hd = ">" + P.rec_names[r] + "\n"; eat(hd.data(), hd.size());
            for (uint64_t x = 0; x < P.rec_lens[r]; x += 100) { eat((const char*)out.data() + off + x, (size_t)std::min<uint64_t>(100, P.rec_lens[r] - x)); eat("\n", 1); }
            off += P.rec_lens[r];
        }
        if (n_records) *n_records = (int)P.rec_names.size(); if (total_bases) *total_bases = P.total; if (checksum) *checksum = h;
        return SCS_OK;
    } catch (const std::exception& e) { if (errbuf && errlen) { strncpy(errbuf, e.what(), errlen - 1); errbuf[errlen - 1] = 0; } return SCS_EIO; }
}
// host-only test seam: the BGZF kernels' arithmetic run on the CPU ("thread" by "thread" over the same functions: scs_bgzf.hip)
int scs_bgzf_probe(const void* text, uint64_t nbytes, uint32_t lds_out_cap, void* out, uint64_t cap, uint64_t* n_out) {
    if ((!text && nbytes) || !n_out) return SCS_EINVAL;
    std::vector<uint8_t> z; bgzf_compress_host((const uint8_t*)text, nbytes, lds_out_cap ? lds_out_cap : BGZF_LDS_OUT, z);
    *n_out = z.size();
    if (out) { if (z.size() > cap) return SCS_EOVERFLOW; memcpy(out, z.data(), z.size()); }
    return SCS_OK;
}
int scs_fasta_write_index(const char* path, char* errbuf, size_t errlen) {
    if (!path) return SCS_EINVAL;
    std::vector<FastaRecord> recs;
    try { load_fasta(path, recs, true); }
    catch (const std::exception& e) { if (errbuf && errlen) { strncpy(errbuf, e.what(), errlen - 1); errbuf[errlen - 1] = 0; } return SCS_EIO; }
    return SCS_OK;
}
int scs_profile_open(const char* path, int paired, int isize, void** handle, char* errbuf, size_t errlen) {
    if (!path || !handle) return SCS_EINVAL;
    ProfileTables* T = new ProfileTables;
    try { load_profile(path, paired != 0, isize, *T); }
    catch (const std::exception& e) { if (errbuf && errlen) { strncpy(errbuf, e.what(), errlen - 1); errbuf[errlen - 1] = 0; } delete T; *handle = nullptr; return SCS_EIO; }
    *handle = T; return SCS_OK;
}
int scs_profile_table(void* handle, int which, const uint32_t** thr, const double** cdf, size_t* n) {
    if (!handle) return SCS_EINVAL;
    ProfileTables* T = (ProfileTables*)handle;
    const std::vector<uint32_t>* t; const std::vector<double>* d;
    switch (which) {
        case 0: t = &T->subs1_t; d = &T->subs1; break; case 1: t = &T->subs2_t; d = &T->subs2; break; case 2: t = &T->qual_t; d = &T->qual; break;
        case 3: t = &T->ins_t; d = &T->ins_cdf; break; case 4: t = &T->del_t; d = &T->del_cdf; break; case 5: t = &T->isize_t; d = &T->isize_cdf; break;
        case 6: if (thr) *thr = T->qual_alias.data(); if (cdf) *cdf = nullptr; if (n) *n = T->qual_alias.size(); return SCS_OK;   // alias quality rows
        default: return SCS_EINVAL;
    }
    if (thr) *thr = t->data(); if (cdf) *cdf = d->data(); if (n) *n = t->size();
    return SCS_OK;
}
int scs_profile_scalars(void* handle, double* out) {
    if (!handle || !out) return SCS_EINVAL;
    ProfileTables* T = (ProfileTables*)handle;
    out[0] = T->read_length; out[1] = T->bins; out[2] = T->t_insert; out[3] = T->t_delete; out[4] = T->isize_min; out[5] = T->have_cdf2; out[6] = T->insert_rate; out[7] = T->del_rate;
    out[8] = T->t_indel; out[9] = T->qual_k;
    return SCS_OK;
}
void scs_profile_close(void* handle) { delete (ProfileTables*)handle; }

}  // extern "C"
