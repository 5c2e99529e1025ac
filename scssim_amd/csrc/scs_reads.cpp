// scs_reads.cpp -- Malbac::setReadCounts and Malbac::yieldReads on the device, and the FASTQ sink (SeqWriter's replacement)
#include "scs_ctx.h"

namespace scs {
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
std::vector<int> gpu_local_cpus(int device) {
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
    char* save = nullptr;                                                          // (strtok_r: two ctxs on two host threads come through here at once)
    for (char* tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
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
void do_yield(scs_ctx* c, const OutTarget& tg, uint64_t* n1_out, uint64_t* n2_out, uint64_t* pairs_out) {
    if (!c->allocated) throw ScsError(SCS_EINVAL, "scs_yield_reads: call scs_allocate_reads first");
    hipStream_t s = c->stream; const int paired = c->cfg.paired != 0;
    if (c->cfg.verbose) fprintf(stderr, "\n*****Producing reads*****\n");
    c->timing_gate = (c->yield_calls++ % c->timing_every) == 0;
    c->tm_reads.reset(); c->tm_indels.reset();
    const uint64_t P = c->n_pairs_planned;
    // A paired-end job on a model whose [Insert Size Standard Deviation] is 0 has no insert-size alphabet (Profile.cpp:908: built only when
    // stdISize > 0); the reference's first yieldInsertSize then asks its Config for a parameter that does not exist and exit(1)s
    // (Profile.cpp:1482-1485 -> Config.cpp:85-93) -- after the amplification, with the output files opened and empty.  Same here, as an error code.
    if (paired && P > 0 && c->prof.isize_t.empty()) throw ScsError(SCS_EIO, "Error: unrecognized parameter name \"insertSize\"");
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
        // writers + 2 pinned slots of two mates each stay allocated until the ctx goes: at most 24 GB of them per ctx (12 writers x 2 M pairs of
        // PE150 = 19.5 GB; 64 writers would pin 92 GB per rank)
        const uint64_t per_pair = 4ull * L + 64ull;
        while (sink_batch > (1ull << 18) && ((uint64_t)tg.sink->writers + 2ull) * sink_batch * per_pair > (24ull << 30)) sink_batch >>= 1;
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
        if (seam_env("SCS_DEBUG_CLASSES")) fprintf(stderr, "[classes] batch of %u pairs: general %u / %u, one-event %u / %u\n", np, nc1, nc2, nd1, nd2);
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

void sink_pipe_free(scs_ctx* c) { if (c->pipe) { c->pipe->release(); delete c->pipe; c->pipe = nullptr; } }
}  // namespace scs
