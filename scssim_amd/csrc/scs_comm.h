// scs_comm.h -- RCCL inside the library (one process per GPU) and the FASTQ file sink / shard merge.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>

namespace scs {

// ---- RCCL, bound at run time (dlopen of librccl.so.1: a process that already carries RCCL -- torch -- shares that copy,
// and the library still loads where RCCL is absent).  Collectives run on the stream given, no host sync.
struct RcclComm;
int  rccl_unique_id(void* id128, std::string& err);                                   // ncclGetUniqueId
RcclComm* rccl_init(const void* id128, int rank, int nranks, std::string& err);       // ncclCommInitRank on the current device
int  rccl_allreduce_sum(RcclComm* c, void* d_vals, uint64_t n, int elem_bytes, hipStream_t s, std::string& err);   // in place; 4 = uint32, 8 = uint64
int  rccl_allgather(RcclComm* c, const void* d_send, void* d_recv, uint64_t bytes_per_rank, hipStream_t s, std::string& err);
void rccl_destroy(RcclComm* c);

// ---- FASTQ files: SeqWriter (lib/seqwriter/SeqWriter.cpp:12-64) without its mutex: a batch is cut into slices that a few
// one pwrite() per file and batch, the two files in parallel (page-cache copies are the cost of a tmpfs / buffered write).
class FastqFiles {
public:
    FastqFiles() = default;
    ~FastqFiles();
    bool open(const std::string& p1, const std::string& p2, int threads, std::string& err);   // p2 empty: single end
    bool write(const char* a, size_t na, const char* b, size_t nb);                             // appends to both files
    bool close();
    uint64_t bytes(int k) const { return total_[k]; }
    // measurement only (SCS_SINK_RECYCLE_MB): rewind a file once it holds this many bytes, so that a whole-genome job's
    // ~190 GB of FASTQ can be timed through D2H + write() into tmpfs without keeping them in the page cache
    void set_recycle(uint64_t bytes) { recycle_ = bytes; }
private:
    int fd_[2] = {-1, -1}; uint64_t pos_[2] = {0, 0}, total_[2] = {0, 0}, recycle_ = 0; int threads_ = 1; bool failed_ = false;
};

// Shard index: for every list segment slot (ALLOC_SLOTS of them) the byte offset in each of the shard's two files where
// the slot's records start, plus the totals: 41 entries per file.
bool write_shard_index(const std::string& path, const std::vector<uint64_t>& off1, const std::vector<uint64_t>& off2, std::string& err);
// Concatenates the byte ranges of the shards in whole-job list order (slot by slot, shard by shard) into <prefix>_1.fq /
// _2.fq (or <prefix>.fq): in-kernel copies (copy_file_range) by a few threads at known output offsets -- no record is parsed.
bool merge_shards(const std::string& prefix, int nranks, bool paired, bool keep_shards, std::string& err);
std::string shard_path(const std::string& prefix, int rank, int file, bool paired);    // <prefix>.r<k>_1.fq / _2.fq / .fq
std::string shard_index_path(const std::string& prefix, int rank);                     // <prefix>.r<k>.idx

}  // namespace scs
