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

struct RcclComm { ncclComm_t comm = nullptr; int rank = 0, nranks = 1; };

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
void rccl_destroy(RcclComm* c) { if (!c) return; if (c->comm && api().CommDestroy) (void)api().CommDestroy(c->comm); delete c; }

// ------------------------------------------------------------------------------------------------ files
FastqFiles::~FastqFiles() { (void)close(); }
bool FastqFiles::open(const std::string& p1, const std::string& p2, int threads, std::string& err) {
    threads_ = std::max(1, threads);
    fd_[0] = ::open(p1.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd_[0] < 0) { err = "Error: can not open fastq file to save results:\n" + p1; return false; }
    if (!p2.empty()) { fd_[1] = ::open(p2.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (fd_[1] < 0) { err = "Error: can not open fastq file to save results:\n" + p2; return false; } }
    pos_[0] = pos_[1] = 0; total_[0] = total_[1] = 0; failed_ = false;
    return true;
}
static bool pwrite_all(int fd, const char* p, size_t n, uint64_t off) {
    while (n) { const ssize_t w = ::pwrite(fd, p, n, (off_t)off); if (w < 0) { if (errno == EINTR) continue; return false; } p += w; n -= (size_t)w; off += (uint64_t)w; }
    return true;
}
// A batch = one pwrite() per file, the two files in parallel.  More writers per file do not help: writes into ONE file
// serialise on its inode lock (measured on the GPU box's tmpfs: 5.7 GB/s per file with 1, 4, 8 or 16 threads slicing it,
// against 39 GB/s for 8 threads on 8 files), and filling a shared mapping of the file from several threads is slower still
// (3.8 GB/s: the page faults contend).  `threads_` > 1 slices a file anyway for file systems where it pays.
bool FastqFiles::write(const char* a, size_t na, const char* b, size_t nb) {
    if (failed_) return false;
    struct Slice { int fd; const char* p; size_t n; uint64_t off; };
    std::vector<Slice> sl;
    const size_t grain = 8u << 20;
    auto cut = [&](int k, const char* p, size_t n) {
        if (!n || fd_[k] < 0) return;
        if (recycle_ && pos_[k] > recycle_) { if (ftruncate(fd_[k], 0) != 0) failed_ = true; pos_[k] = 0; }
        const size_t parts = std::max<size_t>(1, std::min<size_t>((size_t)threads_, n / grain)), per = (n + parts - 1) / parts;
        for (size_t o = 0; o < n; o += per) sl.push_back(Slice{fd_[k], p + o, std::min(per, n - o), pos_[k] + o});
        pos_[k] += n; total_[k] += n;
    };
    cut(0, a, na); cut(1, b, nb);
    if (sl.empty()) return !failed_;
    std::vector<std::thread> th; std::vector<char> okv(sl.size(), 1);
    for (size_t i = 1; i < sl.size(); ++i) th.emplace_back([&, i] { okv[i] = pwrite_all(sl[i].fd, sl[i].p, sl[i].n, sl[i].off); });
    okv[0] = pwrite_all(sl[0].fd, sl[0].p, sl[0].n, sl[0].off);
    for (auto& t : th) t.join();
    for (char v : okv) if (!v) failed_ = true;
    return !failed_;
}
bool FastqFiles::close() {
    bool good = !failed_;
    for (int k = 0; k < 2; ++k) if (fd_[k] >= 0) {
        if (::close(fd_[k]) != 0) good = false;
        fd_[k] = -1;
    }
    return good;
}

std::string shard_path(const std::string& prefix, int rank, int file, bool paired) {
    return prefix + ".r" + std::to_string(rank) + (paired ? (file == 0 ? "_1.fq" : "_2.fq") : ".fq");
}
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
bool merge_shards(const std::string& prefix, int nranks, bool paired, bool keep_shards, std::string& err) {
    if (nranks < 1) { err = "merge_shards: bad rank count"; return false; }
    std::vector<std::vector<uint64_t>> o1(nranks), o2(nranks);
    for (int r = 0; r < nranks; ++r) if (!read_shard_index(shard_index_path(prefix, r), o1[r], o2[r], err)) return false;
    const size_t nslot = o1[0].size() - 1;
    for (int r = 0; r < nranks; ++r) if (o1[r].size() != nslot + 1) { err = "shard indexes disagree on the number of segments"; return false; }
    struct Job { int file, rank; uint64_t in, out, n; };
    std::vector<Job> jobs; uint64_t out[2] = {0, 0};
    for (size_t s = 0; s < nslot; ++s) for (int r = 0; r < nranks; ++r) for (int k = 0; k < (paired ? 2 : 1); ++k) {
        const std::vector<uint64_t>& o = k ? o2[r] : o1[r];
        const uint64_t n = o[s + 1] - o[s];
        if (n) jobs.push_back(Job{k, r, o[s], out[k], n});
        out[k] += n;
    }
    int ofd[2] = {-1, -1}; std::vector<int> ifd((size_t)nranks * 2, -1); bool good = true;
    const std::string out_name[2] = {prefix + (paired ? "_1.fq" : ".fq"), prefix + "_2.fq"};
    for (int k = 0; k < (paired ? 2 : 1) && good; ++k) { ofd[k] = ::open(out_name[k].c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (ofd[k] < 0) { err = "Error: can not open fastq file to save results:\n" + out_name[k]; good = false; } }
    for (int r = 0; r < nranks && good; ++r) for (int k = 0; k < (paired ? 2 : 1); ++k) {
        ifd[(size_t)r * 2 + k] = ::open(shard_path(prefix, r, k, paired).c_str(), O_RDONLY);
        if (ifd[(size_t)r * 2 + k] < 0) { err = "can not open shard " + shard_path(prefix, r, k, paired); good = false; break; }
    }
    if (good) {
        for (int k = 0; k < (paired ? 2 : 1); ++k) if (ftruncate(ofd[k], (off_t)out[k]) != 0) { /* sparse pre-size is an optimisation only */ }
        const unsigned nt = std::max(1u, std::min<unsigned>(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th; std::vector<char> okv(nt, 1);
        for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] {
            for (size_t j = t; j < jobs.size(); j += nt) if (!copy_range(ifd[(size_t)jobs[j].rank * 2 + jobs[j].file], ofd[jobs[j].file], jobs[j].in, jobs[j].out, jobs[j].n)) { okv[t] = 0; return; }
        });
        for (auto& t : th) t.join();
        for (char v : okv) if (!v) { good = false; err = "copying a shard range failed"; }
    }
    for (int fd : ifd) if (fd >= 0) ::close(fd);
    for (int k = 0; k < 2; ++k) if (ofd[k] >= 0 && ::close(ofd[k]) != 0) { good = false; err = "closing " + out_name[k] + " failed"; }
    if (good && !keep_shards)
        for (int r = 0; r < nranks; ++r) { for (int k = 0; k < (paired ? 2 : 1); ++k) (void)::unlink(shard_path(prefix, r, k, paired).c_str()); (void)::unlink(shard_index_path(prefix, r).c_str()); }
    return good;
}

}  // namespace scs
