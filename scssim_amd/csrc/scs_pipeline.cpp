// scs_pipeline.cpp -- the C ABI (include/scssim_hip.h) over the host files of the library, and what they share: the mailbox.
// One scs_ctx = one HIP device + one stream; all amplicon state lives in HBM as flat SoA arrays (scs_ctx.h).
// Reference call sequence reproduced: src/scssim.cpp:46-67 (genreads branch of main()).
#include "scs_ctx.h"

namespace { thread_local std::string g_create_error; }

namespace scs {

// ---- mailbox: device scalars -> pinned host words, no copy and no stream sync (k_mail)
void mail_post(scs_ctx* c, const Mail& m, bool last, hipStream_t st) {   // last: the post the host will wait for; st: the ctx stream unless given
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
        if (f & FLAG_KEYSPACE) m += " primer budget of a fragment beyond 2^20 (-p / -r far outside the reference's ranges)";
        throw ScsError(SCS_EOVERFLOW, m);
    }
}
void check_flags(scs_ctx* c) {
    Mail m; m.add(c->flags.p, 4, 30); mail_post(c, m, true); mail_wait(c);
    flags_eval(c);
}

}  // namespace scs

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
                      &c->d_isize, &c->d_gcmeans, &c->genome, &c->genome2, &c->gx_gc_bits, &c->gx_n_bits, &c->gx_gc_cnt, &c->gx_n_cnt, &c->gx_gc_pref, &c->gx_n_pref, &c->gx_gc_pair, &c->d_binom, &c->df_blob, &c->df_primers, &c->df_hasn, &c->primer_cnt, &c->primer_delta, &c->primer_cut, &c->primer_gdelta, &c->st_eidx, &c->st_etype, &c->st_estart, &c->st_info, &c->st_list, &c->st_sorted, &c->st_tmp, &c->att_wave_first,
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
    sink_pipe_free(c);
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
int scs_yield_reads_files_ex(scs_ctx* c, const char* prefix, int writers, int generations, int flags) {
    const int bgzf = flags & SCS_SINK_BGZF; const bool in_place = (flags & SCS_SINK_IN_PLACE) != 0;
    return guarded(c, [&] {
        if (flags & ~(SCS_SINK_BGZF | SCS_SINK_IN_PLACE)) throw ScsError(SCS_EINVAL, "scs_yield_reads_files: unknown sink flag");
        if (!prefix || !*prefix) throw ScsError(SCS_EINVAL, "scs_yield_reads_files: no output prefix");
        if (writers > 64 || generations > 64 || (int64_t)std::max(1, writers) * std::max(1, generations) > 99) throw ScsError(SCS_EINVAL, "scs_yield_reads_files: at most 64 writers and 99 parts");
        const bool pe = c->cfg.paired != 0, shard = c->cfg.shard_count > 1; const std::string pre = prefix;
        const std::string base = shard ? shard_base(pre, c->cfg.shard_rank) : pre;
        FastqParts files; std::string err;
        if (!files.open(base, pe, std::max(1, writers), std::max(1, generations), bgzf ? ".fq.gz" : ".fq", bgzf != 0, err, in_place)) throw ScsError(SCS_EIO, err);
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

