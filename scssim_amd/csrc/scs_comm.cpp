// scs_comm.cpp -- RCCL binding, FASTQ file sink, shard merge (host code; see scs_comm.h).
#include "scs_comm.h"

#include <dlfcn.h>
#include <link.h>
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include <rccl/rccl.h>

namespace scs {

// ------------------------------------------------------------------------------------------------ RCCL
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
RcclApi& api() {
    static RcclApi a = [] {
        RcclApi r;
        // a process that already carries RCCL (torch ships its own librccl.so, without a soname) shares that copy ...
        std::string loaded;
        dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* out) -> int {
            const char* n = info->dlpi_name;
            if (n && *n) { const char* base = strrchr(n, '/'); base = base ? base + 1 : n; if (strncmp(base, "librccl.so", 10) == 0) { *(std::string*)out = n; return 1; } }
            return 0;
        }, &loaded);
        if (!loaded.empty()) r.h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
        // ... otherwise the system's
        if (!r.h) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (!r.h) { r.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return r; }
        auto sym = [&](const char* n) { void* p = dlsym(r.h, n); if (!p && r.why.empty()) r.why = std::string("RCCL symbol missing: ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId"); r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy"); r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort"); r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather"); r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        return r;
    }();
    return a;
}
bool ok(ncclResult_t r, const char* what, std::string& err) {
    if (r == ncclSuccess) return true;
    err = std::string(what) + ": " + (api().GetErrorString ? api().GetErrorString(r) : "RCCL error"); return false;
}
}  // namespace

struct RcclComm { ncclComm_t comm = nullptr; int rank = 0, nranks = 1; std::atomic<bool> aborted{false}; };

int rccl_unique_id(void* id128, std::string& err) {
    RcclApi& a = api();
    if (!a.why.empty()) { err = a.why; return 1; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (!ok(a.GetUniqueId(&id), "ncclGetUniqueId", err)) return 1;
    memcpy(id128, &id, sizeof id);
    return 0;
}
RcclComm* rccl_init(const void* id128, int rank, int nranks, std::string& err) {
    RcclApi& a = api();
    if (!a.why.empty()) { err = a.why; return nullptr; }
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    RcclComm* c = new RcclComm; c->rank = rank; c->nranks = nranks;
    if (!ok(a.CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank", err)) { delete c; return nullptr; }
    return c;
}
int rccl_allreduce_sum(RcclComm* c, void* d, uint64_t n, int elem_bytes, hipStream_t s, std::string& err) {
    const ncclDataType_t t = elem_bytes == 8 ? ncclUint64 : ncclUint32;
    return ok(api().AllReduce(d, d, (size_t)n, t, ncclSum, c->comm, s), "ncclAllReduce", err) ? 0 : 1;
}
int rccl_allgather(RcclComm* c, const void* d_send, void* d_recv, uint64_t bytes, hipStream_t s, std::string& err) {
    return ok(api().AllGather(d_send, d_recv, (size_t)bytes, ncclUint8, c->comm, s), "ncclAllGather", err) ? 0 : 1;
}
void rccl_destroy(RcclComm* c) { if (!c) return; if (c->comm && !c->aborted.load() && api().CommDestroy) (void)api().CommDestroy(c->comm); delete c; }
int rccl_count(RcclComm* c) { int n = 0; if (!c || !c->comm || !api().CommCount || api().CommCount(c->comm, &n) != ncclSuccess) return 0; return n; }
void rccl_abort(RcclComm* c) { if (c && c->comm && api().CommAbort && !c->aborted.exchange(true)) (void)api().CommAbort(c->comm); }

// ------------------------------------------------------------------------------------------------ files
static bool pwrite_all(int fd, const char* p, size_t n, uint64_t off) {
    while (n) { const ssize_t w = ::pwrite(fd, p, n, (off_t)off); if (w < 0) { if (errno == EINTR) continue; return false; } p += w; n -= (size_t)w; off += (uint64_t)w; }
    return true;
}
static const unsigned char kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

std::string part_path(const std::string& base, int part, int parts, int mate, bool paired, const std::string& suffix) {
    std::string p = base;
    if (parts > 1) { char t[16]; snprintf(t, sizeof t, ".p%02d", part); p += t; }
    if (paired) p += mate == 0 ? "_1" : "_2";
    return p + suffix;
}
std::string parts_index_path(const std::string& base) { return base + ".parts"; }

FastqParts::~FastqParts() { std::string e; (void)close(e); }
// (close() opens every part, so no set-aside file of an in-place job outlives it; a part that failed to open leaves its ".prev" behind, unlinked here)
void FastqParts::drop_set_aside() {
    if (!in_place_) return;
    for (int k = writers; k < regions; ++k) for (int m = 0; m < (paired_ ? 2 : 1); ++m) (void)::unlink((part_path(base_, k, regions, m, paired_, suffix_) + ".prev").c_str());
}
bool FastqParts::open_part(int k, std::string& err) {
    Part& P = part_[(size_t)k];
    if (P.opened) return true;
    for (int m = 0; m < (paired_ ? 2 : 1); ++m) {
        const std::string p = part_path(base_, k, regions, m, paired_, suffix_);
        // in place: a file that is there already keeps its pages and is overwritten where it lies; its length is set when the part is
        // finished.  (Freeing a file system's pages and taking them again costs more than the copy into them: tools/probes/overwrite_probe.py)
        if (in_place_ && k >= writers) (void)::rename((p + ".prev").c_str(), p.c_str());   // a later generation's old file, set aside by open(): back under its name now
        P.fd[m] = ::open(p.c_str(), O_WRONLY | O_CREAT | (in_place_ ? 0 : O_TRUNC), 0644);
        if (P.fd[m] < 0) { err = "Error: can not open fastq file to save results:\n" + p; P.failed = true; return false; }
    }
    P.opened = true;
    return true;
}
void FastqParts::finish_part(int k) {                                                // the part is complete: BGZF end-of-file block, close
    Part& P = part_[(size_t)k];
    if (!P.opened || P.done) return;
    for (int m = 0; m < 2; ++m) if (P.fd[m] >= 0) {
        if (eof_ && !P.failed) { if (pwrite_all(P.fd[m], (const char*)kBgzfEof, sizeof kBgzfEof, P.pos[m])) P.pos[m] += sizeof kBgzfEof; else P.failed = true; }
        if (in_place_ && ftruncate(P.fd[m], (off_t)P.pos[m]) != 0) P.failed = true;
        if (::close(P.fd[m]) != 0) P.failed = true;
        P.fd[m] = -1;
    }
    P.done = true;
}
bool FastqParts::open(const std::string& base, bool paired, int nwriters, int generations, const std::string& suffix, bool bgzf_eof, std::string& err, bool in_place) {
    base_ = base; paired_ = paired; eof_ = bgzf_eof; suffix_ = suffix; in_place_ = in_place;
    writers = std::max(1, nwriters); regions = writers * std::max(1, generations);
    part_.assign((size_t)regions, Part()); cur_.assign((size_t)writers, -1);
    first_ = part_path(base, 0, regions, 0, paired, suffix);
    (void)::unlink(parts_index_path(base).c_str());                                  // a stale index must not describe the new files
    // the first generation's files are opened here, so that an output that cannot be written is reported before the job runs;
    // the later ones when their first batch arrives (a part's existence tells a consumer that the part `writers` before it is final)
    // In place, the files of an earlier job are still there under the later generations' names, and a consumer would take them for the
    // signal that the generation before is final: they are set aside (renamed, pages kept) until their part's first batch arrives.
    if (in_place_) for (int k = writers; k < regions; ++k) for (int m = 0; m < (paired_ ? 2 : 1); ++m) {
        const std::string p = part_path(base_, k, regions, m, paired_, suffix_);
        if (::rename(p.c_str(), (p + ".prev").c_str()) != 0 && errno != ENOENT) { err = "Error: can not open fastq file to save results:\n" + p; return false; }
    }
    for (int k = 0; k < writers; ++k) if (!open_part(k, err)) return false;
    return true;
}
// One writer thread per region r % writers (SinkPipe); a region's two files are written one after the other -- except for the
// single region of the reference's two-file layout, whose second file gets a thread of its own for the batch.
int FastqParts::put(int region, const char* a, size_t na, const char* b, size_t nb) {
    if (region < 0 || region >= (int)part_.size()) return 1;
    int& cur = cur_[(size_t)(region % writers)];
    if (cur != region) {                                                            // this writer moves on to its next generation's part
        if (cur >= 0) finish_part(cur);
        for (int k = cur < 0 ? region % writers : cur + writers; k < region; k += writers) { std::string e; if (open_part(k, e)) finish_part(k); }   // (ranges without a batch: empty parts)
        cur = region;
    }
    Part& P = part_[(size_t)region];
    std::string err;
    if (P.failed || P.done || !open_part(region, err)) return 1;
    bool ok2 = true; std::thread t2;
    const bool par = regions == 1 && nb && na && P.fd[1] >= 0;
    if (par) t2 = std::thread([&] { ok2 = pwrite_all(P.fd[1], b, nb, P.pos[1]); });
    bool ok1 = !na || pwrite_all(P.fd[0], a, na, P.pos[0]);
    if (par) t2.join(); else if (nb && P.fd[1] >= 0) ok2 = pwrite_all(P.fd[1], b, nb, P.pos[1]);
    P.pos[0] += na; if (P.fd[1] >= 0) P.pos[1] += nb;
    if (!ok1 || !ok2) { P.failed = true; return 1; }
    return 0;
}
bool FastqParts::close(std::string& err) {
    bool good = true; const bool any = !part_.empty();
    for (size_t k = 0; k < part_.size(); ++k) {
        std::string e;
        if (!part_[k].opened && !part_[k].failed) (void)open_part((int)k, e);      // a range that got no batch: an empty part, so that the set is whole
        finish_part((int)k);
        if (part_[k].failed) { good = false; if (err.empty()) err = e.empty() ? "writing " + part_path(base_, (int)k, regions, 0, paired_, suffix_) + " failed" : e; }
    }
    if (any) drop_set_aside();
    if (!good) return false;
    if (any && regions > 1) {
        FILE* f = fopen(parts_index_path(base_).c_str(), "w");
        if (!f) { err = "can not write " + parts_index_path(base_); return false; }
        fprintf(f, "# scssim FASTQ parts: part, bytes in file 1, bytes in file 2 (the files' concatenation in this order is the whole file)\n");
        for (size_t k = 0; k < part_.size(); ++k) fprintf(f, "%zu\t%llu\t%llu\n", k, (unsigned long long)part_[k].pos[0], (unsigned long long)part_[k].pos[1]);
        if (fclose(f) != 0) { err = "can not write " + parts_index_path(base_); return false; }
    }
    part_.clear();
    return true;
}

bool LogicalFile::open(const std::string& base, int mate, bool paired, const std::string& suffix, std::string& err) {
    close();
    std::vector<uint64_t> sizes; int parts = 1;
    if (FILE* f = fopen(parts_index_path(base).c_str(), "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) { if (line[0] == '#') continue; unsigned long long k, a, b; if (sscanf(line, "%llu %llu %llu", &k, &a, &b) == 3) sizes.push_back(mate == 0 ? a : b); }
        fclose(f);
        if (sizes.empty()) { err = "malformed parts index " + parts_index_path(base); return false; }
        parts = (int)sizes.size();
    }
    for (int k = 0; k < parts; ++k) {
        const std::string p = part_path(base, k, parts, mate, paired, suffix);
        const int d = ::open(p.c_str(), O_RDONLY);
        if (d < 0) { err = "can not open " + p; close(); return false; }
        struct stat st_; if (fstat(d, &st_) != 0) { ::close(d); err = "can not stat " + p; close(); return false; }
        if (!sizes.empty() && (uint64_t)st_.st_size != sizes[(size_t)k]) { ::close(d); err = p + " does not have the size its index states"; close(); return false; }
        fd.push_back(d); start.push_back(size); path.push_back(p); size += (uint64_t)st_.st_size;
    }
    return true;
}
void LogicalFile::close() { for (int d : fd) if (d >= 0) ::close(d); fd.clear(); start.clear(); path.clear(); size = 0; }

std::string shard_base(const std::string& prefix, int rank) { return prefix + ".r" + std::to_string(rank); }
std::string shard_index_path(const std::string& prefix, int rank) { return prefix + ".r" + std::to_string(rank) + ".idx"; }

bool write_shard_index(const std::string& path, const std::vector<uint64_t>& off1, const std::vector<uint64_t>& off2, std::string& err) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) { err = "can not write shard index " + path; return false; }
    fprintf(f, "# scssim FASTQ shard index: list segment slot, byte offset in file 1, byte offset in file 2 (last line: totals)\n");
    for (size_t i = 0; i < off1.size(); ++i) fprintf(f, "%zu\t%llu\t%llu\n", i, (unsigned long long)off1[i], (unsigned long long)(i < off2.size() ? off2[i] : 0ull));
    return fclose(f) == 0;
}
static bool read_shard_index(const std::string& path, std::vector<uint64_t>& off1, std::vector<uint64_t>& off2, std::string& err) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) { err = "can not read shard index " + path; return false; }
    char line[256]; off1.clear(); off2.clear();
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '#') continue;
        unsigned long long i, a, b;
        if (sscanf(line, "%llu %llu %llu", &i, &a, &b) == 3) { off1.push_back(a); off2.push_back(b); }
    }
    fclose(f);
    if (off1.size() < 2) { err = "malformed shard index " + path; return false; }
    return true;
}
static bool copy_range(int in, int out, uint64_t off_in, uint64_t off_out, uint64_t n) {
    std::vector<char> buf;
    while (n) {
        off64_t oi = (off64_t)off_in, oo = (off64_t)off_out;
        ssize_t r = copy_file_range(in, &oi, out, &oo, (size_t)std::min<uint64_t>(n, 1ull << 30), 0);
        if (r < 0 && (errno == EXDEV || errno == EINVAL || errno == ENOSYS || errno == EOPNOTSUPP)) {   // not supported here: through a buffer
            if (buf.empty()) buf.resize(8u << 20);
            r = ::pread(in, buf.data(), (size_t)std::min<uint64_t>(n, buf.size()), (off_t)off_in);
            if (r > 0 && !pwrite_all(out, buf.data(), (size_t)r, off_out)) return false;
        }
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) return false;                                                  // shard shorter than its index says
        off_in += (uint64_t)r; off_out += (uint64_t)r; n -= (uint64_t)r;
    }
    return true;
}
// bytes [off_in, off_in + n) of a logical file (its parts in order) -> out at off_out
static bool copy_logical(const LogicalFile& in, int out, uint64_t off_in, uint64_t off_out, uint64_t n) {
    if (off_in + n > in.size) return false;
    size_t k = (size_t)(std::upper_bound(in.start.begin(), in.start.end(), off_in) - in.start.begin()) - 1;
    while (n) {
        const uint64_t end = k + 1 < in.start.size() ? in.start[k + 1] : in.size, take = std::min<uint64_t>(n, end - off_in);
        if (take && !copy_range(in.fd[k], out, off_in - in.start[k], off_out, take)) return false;
        off_in += take; off_out += take; n -= take; ++k;
    }
    return true;
}
struct CopyJob { const LogicalFile* in; int out; uint64_t off_in, off_out, n; };
static bool run_copies(const std::vector<CopyJob>& jobs) {
    const unsigned nt = std::max(1u, std::min<unsigned>(8u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th; std::vector<char> okv(nt, 1);
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
        for (size_t j = t; j < jobs.size(); j += nt) if (!copy_logical(*jobs[j].in, jobs[j].out, jobs[j].off_in, jobs[j].off_out, jobs[j].n)) { okv[t] = 0; return; }
    });
    for (auto& t : th) t.join();
    for (char v : okv) if (!v) return false;
    return true;
}
static void unlink_logical(const std::string& base, const std::vector<LogicalFile>& lf) {
    for (auto& l : lf) for (auto& p : l.path) (void)::unlink(p.c_str());
    (void)::unlink(parts_index_path(base).c_str());
}
bool merge_parts(const std::string& base, bool paired, const std::string& suffix, bool keep, std::string& err) {
    const int nm = paired ? 2 : 1;
    std::vector<LogicalFile> in((size_t)nm); int ofd[2] = {-1, -1}; bool good = true;
    for (int m = 0; m < nm && good; ++m) good = in[(size_t)m].open(base, m, paired, suffix, err);
    if (good && in[0].fd.size() == 1 && in[0].path[0] == part_path(base, 0, 1, 0, paired, suffix)) { for (auto& l : in) l.close(); return true; }   // already the single files
    std::vector<CopyJob> jobs;
    for (int m = 0; m < nm && good; ++m) {
        const std::string out = part_path(base, 0, 1, m, paired, suffix);
        ofd[m] = ::open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (ofd[m] < 0) { err = "Error: can not open fastq file to save results:\n" + out; good = false; break; }
        for (size_t k = 0; k < in[(size_t)m].fd.size(); ++k) {
            const uint64_t st = in[(size_t)m].start[k], en = k + 1 < in[(size_t)m].start.size() ? in[(size_t)m].start[k + 1] : in[(size_t)m].size;
            if (en > st) jobs.push_back(CopyJob{&in[(size_t)m], ofd[m], st, st, en - st});
        }
    }
    if (good && !run_copies(jobs)) { good = false; err = "copying a part failed"; }
    for (int m = 0; m < 2; ++m) if (ofd[m] >= 0 && ::close(ofd[m]) != 0) { good = false; err = "closing the merged file failed"; }
    if (good && !keep) unlink_logical(base, in);
    for (auto& l : in) l.close();
    return good;
}
bool merge_shards(const std::string& prefix, int nranks, bool paired, bool keep_shards, std::string& err, const std::string& suffix) {
    if (nranks < 1) { err = "merge_shards: bad rank count"; return false; }
    std::vector<std::vector<uint64_t>> o1(nranks), o2(nranks);
    for (int r = 0; r < nranks; ++r) if (!read_shard_index(shard_index_path(prefix, r), o1[r], o2[r], err)) return false;
    const size_t nslot = o1[0].size() - 1;
    for (int r = 0; r < nranks; ++r) if (o1[r].size() != nslot + 1) { err = "shard indexes disagree on the number of segments"; return false; }
    const int nm = paired ? 2 : 1;
    std::vector<LogicalFile> in((size_t)nranks * 2); bool good = true;
    for (int r = 0; r < nranks && good; ++r) for (int k = 0; k < nm && good; ++k) good = in[(size_t)r * 2 + k].open(shard_base(prefix, r), k, paired, suffix, err);
    int ofd[2] = {-1, -1};
    const std::string out_name[2] = {part_path(prefix, 0, 1, 0, paired, suffix), part_path(prefix, 0, 1, 1, paired, suffix)};
    for (int k = 0; k < nm && good; ++k) { ofd[k] = ::open(out_name[k].c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (ofd[k] < 0) { err = "Error: can not open fastq file to save results:\n" + out_name[k]; good = false; } }
    std::vector<CopyJob> jobs; uint64_t out[2] = {0, 0};
    if (good) for (size_t s = 0; s < nslot; ++s) for (int r = 0; r < nranks; ++r) for (int k = 0; k < nm; ++k) {
        const std::vector<uint64_t>& o = k ? o2[r] : o1[r];
        const uint64_t n = o[s + 1] - o[s];
        if (n) jobs.push_back(CopyJob{&in[(size_t)r * 2 + k], ofd[k], o[s], out[k], n});
        out[k] += n;
    }
    if (good) {
        for (int k = 0; k < nm; ++k) if (ftruncate(ofd[k], (off_t)out[k]) != 0) { /* sparse pre-size is an optimisation only */ }
        if (!run_copies(jobs)) { good = false; err = "copying a shard range failed"; }
    }
    for (int k = 0; k < 2; ++k) if (ofd[k] >= 0 && ::close(ofd[k]) != 0) { good = false; err = "closing " + out_name[k] + " failed"; }
    if (good && !keep_shards)
        for (int r = 0; r < nranks; ++r) {
            std::vector<LogicalFile> mine; for (int k = 0; k < nm; ++k) mine.push_back(in[(size_t)r * 2 + k]);
            unlink_logical(shard_base(prefix, r), mine); (void)::unlink(shard_index_path(prefix, r).c_str());
        }
    for (auto& l : in) l.close();
    return good;
}

}  // namespace scs
