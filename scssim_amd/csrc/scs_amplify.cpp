// scs_amplify.cpp -- Malbac::amplify on the device: setPrimers, the passes of primer attachment with the exact primer stock, the new amplicons
#include "scs_ctx.h"

namespace scs {
namespace {
// ---------------------------------------------------------------- a3: Malbac::setPrimers (Malbac.cpp:236-283) on the device
// One launch gives every template (fragments, then all semis so far) its Poisson budget; the scans
// turn budgets into slot offsets.  One host sync: the sums feed totalPrimers and the buffer sizes.
// ns_cap: upper bound of the semi amplicon count (the count itself is on the device: the passes that made the newest
// semis have not been read back yet -- their counts arrive with this call's mail, ONE wait per cycle).
void set_primers_launch(scs_ctx* c, bool only_frags, uint32_t call, uint32_t ns_cap) {
    hipStream_t s = c->stream;
    const uint32_t nf = (uint32_t)c->f_len.size(), ns = only_frags ? 0u : ns_cap;
    if (nf >= (1u << 26)) throw ScsError(SCS_EOVERFLOW, "more than 2^26 fragments on one GPU: shard the job (the stock keys hold 26 bits of fragment index)");
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
    launch_poisson(s, c->frags_view(), c->semis.view(), ns, p, c->budget_f.as<uint32_t>(), c->budget_s.as<uint32_t>(), c->dsums.as<unsigned long long>(), c->poisson_part.as<unsigned long long>(), c->flags.as<uint32_t>());
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
// The kernels and the argument are in scs_k_amplify.hip ("the primer stock, exactly").  Here: the loop.
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
    const DevGenomeIdx gx{c->gx_gc_bits.as<unsigned long long>(), c->gx_n_bits.as<unsigned long long>(), c->gx_gc_pref.as<uint64_t>(), c->gx_n_pref.as<uint64_t>(), c->gx_gc_pair.as<ulonglong2>()};
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

}  // namespace

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

}  // namespace scs
