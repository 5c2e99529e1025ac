// scs_stage.cpp -- the model and the input side: Profile::train(file), the genome (FASTA parsed on the device, per-shard slices), Genome::splitToFrags
#include "scs_ctx.h"

namespace scs {
// ---------------------------------------------------------------- profile
void do_load_profile(scs_ctx* c, const char* path) {
    load_profile(path, c->cfg.paired != 0, c->cfg.isize, c->prof);
    ProfileTables& P = c->prof; hipStream_t s = c->stream;
    upload(c->t_subs1, P.subs1_t, s); upload(c->t_subs2, P.subs2_t, s); upload(c->t_qual, P.qual_t, s);
    upload(c->t_qcompact, P.qual_alias, s); upload(c->t_ins, P.ins_t, s); upload(c->t_del, P.del_t, s); upload(c->t_isize, P.isize_t, s); upload(c->t_gap, P.gap_t, s);
    upload(c->d_subs1, P.subs1, s); upload(c->d_subs2, P.subs2, s); upload(c->d_qual, P.qual, s);
    upload(c->d_ins, P.ins_cdf, s); upload(c->d_del, P.del_cdf, s); upload(c->d_isize, P.isize_cdf, s);
    std::vector<double> gm(P.gc_means, P.gc_means + 101); upload(c->d_gcmeans, gm, s);
    // the bins as k_reads keeps them in its LDS ring (scs_k_reads.hip RingBin): the four diagonal alias rows (c, c), then the
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
        std::vector<uint32_t> img(Bpad * bw + 128, 0u);                             // + the head: the 1-mers at bin 0 and the 2-mers at bin 1 (rows 0..19), the 2-mers at bin 0 (rows 20..35: a read with one inserted base has its second position there)
        for (size_t ki = 0; ki < 20 && B >= 2; ++ki) keep_pair(subs_t.data() + (ki * B + (ki < 4 ? 0 : 1)) * 4, (uint32_t)(ki & 3), img.data() + Bpad * bw + ki * 2);
        for (size_t ki = 4; ki < 20 && B >= 2; ++ki) keep_pair(subs_t.data() + (ki * B + 0) * 4, (uint32_t)(ki & 3), img.data() + Bpad * bw + (16 + ki) * 2);
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
void stage_genome(scs_ctx* c, const void* d_ascii, const uint64_t* d_lens) {
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
        c->gx_gc_pref.reserve((nw + 2) * 8, s); c->gx_n_pref.reserve((nw + 2) * 8, s); c->gx_gc_pair.reserve((nw + 2) * 16, s); c->scan_tmp.reserve(scan_temp_bytes(nw + 1), s);
        c->genome2.reserve((nw + 1) * 16 + 256, s);                               // two bits per base, 64 bytes of slack in front and 192 behind (the window gather over-reads by up to a dozen words)
        launch_genome_bits(s, c->genome.as<uint8_t>(), tot, nw, c->gx_gc_bits.as<unsigned long long>(), c->gx_n_bits.as<unsigned long long>(), c->gx_gc_cnt.as<uint32_t>(),
                           c->gx_n_cnt.as<uint32_t>(), c->gx_gc_pref.as<uint64_t>(), c->gx_n_pref.as<uint64_t>(), c->scan_tmp.p, c->scan_tmp.cap, c->genome2.as<uint32_t>() + 16, c->gx_gc_pair.as<ulonglong2>());
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
    {   const DevGenomeIdx gx{c->gx_gc_bits.as<unsigned long long>(), c->gx_n_bits.as<unsigned long long>(), c->gx_gc_pref.as<uint64_t>(), c->gx_n_pref.as<uint64_t>(), c->gx_gc_pair.as<ulonglong2>()};
        const DevFrags fv = c->frags_view();
        launch_frag_has_n(c->stream, fv.goff, fv.len, fv.n, gx, c->df_hasn.as<uint8_t>()); }
    c->have_frags = true; c->amplified = false; c->allocated = false;
    c->st.fragments = c->f_len.size();
}

}  // namespace scs
