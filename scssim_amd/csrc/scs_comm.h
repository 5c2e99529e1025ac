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
int  rccl_count(RcclComm* c);                                                          // ncclCommCount (0 on error)
void rccl_abort(RcclComm* c);                                                          // ncclCommAbort: from another thread than the one inside a collective

// ---- FASTQ files: SeqWriter (lib/seqwriter/SeqWriter.cpp:12-64) without its mutex and its single stream.
// A job's FASTQ text reaches the sink batch by batch.  The reads stage cuts the job's records into `regions` contiguous
// ranges and visits them round-robin, so that `regions` writer threads are busy at the same time, each appending the batches
// of ITS range, in order, to ITS OWN pair of files: buffered writes into one file serialise on its inode lock (measured on
// the GPU box's tmpfs: 5.7 GB/s per file from 1, 4, 8 or 16 threads; 39 GB/s for 8 threads on 8 files).
struct BatchSink {
    int regions = 1;                                       // how many contiguous ranges the job's records are cut into (one part file per mate each)
    int writers = 1;                                       // threads: region r is written by thread r % writers; regions = writers x generations
    // called by the region's writer thread, the batches of a region in record order; a / b: the two mates' text (b null: single end)
    virtual int put(int region, const char* a, size_t na, const char* b, size_t nb) = 0;   // 0 = ok
    virtual ~BatchSink() {}
};
// parts == 1: the reference's files <base>_1.fq / <base>_2.fq (or <base>.fq), the two written in parallel.
// parts  > 1: part files <base>.p<kk>_1.fq / _2.fq (.p<kk>.fq), kk = 00 .. parts-1 -- their concatenation in that order
// IS the single file (`cat <base>.p*_1.fq`) -- and an index <base>.parts with their sizes.  parts = writers x generations:
// the records are made generation by generation (the first parts / writers parts, then the next ...), so the early parts are
// complete -- closed, final -- long before the job ends: part p is complete as soon as part p + writers exists.  A consumer
// can stream them, and the page cache holds a couple of generations instead of the whole job's text.
// suffix: ".fq", or ".fq.gz" for BGZF text (close() / the part's completion then ends it with the BGZF end-of-file block).
class FastqParts : public BatchSink {
public:
    ~FastqParts() override;
    // in_place: files that exist are not truncated when they are opened but overwritten where they lie (their pages stay) and cut
    // to their new length when the part is finished -- the same files in the end.  With generations the "part p + writers exists => part p is
    // final" protocol still holds: the earlier job's files of the later generations are renamed to <name>.prev by open() and come back under
    // their names (pages kept) when their first batch arrives
    bool open(const std::string& base, bool paired, int writers, int generations, const std::string& suffix, bool bgzf_eof, std::string& err, bool in_place = false);
    int put(int region, const char* a, size_t na, const char* b, size_t nb) override;
    bool close(std::string& err);                          // sizes final, index written (parts > 1)
    uint64_t bytes(int mate) const { uint64_t t = 0; for (auto& p : part_) t += p.pos[mate]; return t; }
    const std::string& first_path() const { return first_; }
private:
    struct Part { int fd[2] = {-1, -1}; uint64_t pos[2] = {0, 0}; bool failed = false, opened = false, done = false; };
    bool open_part(int k, std::string& err); void finish_part(int k); void drop_set_aside();
    std::vector<Part> part_; std::vector<int> cur_; std::string base_, first_, suffix_; bool paired_ = true, eof_ = false, in_place_ = false;
};
std::string part_path(const std::string& base, int part, int parts, int mate, bool paired, const std::string& suffix);
std::string parts_index_path(const std::string& base);                                 // <base>.parts
// The parts of <base> (or the single file when there is no index) as ONE logical file per mate: open read-only.
struct LogicalFile { std::vector<int> fd; std::vector<uint64_t> start; uint64_t size = 0; std::vector<std::string> path;
                     bool open(const std::string& base, int mate, bool paired, const std::string& suffix, std::string& err); void close(); };
// <base>.p*_1.fq ... -> <base>_1.fq ... by byte-range copies (the parts are removed unless keep)
bool merge_parts(const std::string& base, bool paired, const std::string& suffix, bool keep, std::string& err);

// Shard index: for every list segment slot (ALLOC_SLOTS of them) the byte offset in each of the shard's two files where
// the slot's records start, plus the totals: 41 entries per file.
bool write_shard_index(const std::string& path, const std::vector<uint64_t>& off1, const std::vector<uint64_t>& off2, std::string& err);
// Concatenates the byte ranges of the shards in whole-job list order (slot by slot, shard by shard) into <prefix>_1.fq /
// _2.fq (or <prefix>.fq): in-kernel copies (copy_file_range) by a few threads at known output offsets -- no record is parsed.
bool merge_shards(const std::string& prefix, int nranks, bool paired, bool keep_shards, std::string& err, const std::string& suffix = ".fq");
std::string shard_base(const std::string& prefix, int rank);                           // <prefix>.r<k>: a shard's files are FastqParts on this base
std::string shard_index_path(const std::string& prefix, int rank);                     // <prefix>.r<k>.idx

}  // namespace scs
