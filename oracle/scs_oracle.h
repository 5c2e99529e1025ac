// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the SCSsim `genreads` hot path (reference: qasimyu/scssim,
// cited as <file>:<line> relative to the reference root).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
// run this; the product (scssim_amd/, include/) never does.
//
// Parity status: PINNED.  In `--rng ref` mode this program consumes the same
// random streams as the reference (2 x std::mt19937 per worker, glibc rand(),
// libstdc++ normal_distribution over minstd_rand0) in the same order, and its
// FASTQ output is byte-identical to oracle/_ref/scssim_ref run under
// oracle/seedshim.cpp at -t 1 (tests/test_oracle_golden.py, tests/golden/).
// In `--rng counter` mode the same algorithm draws from Philox4x32-10 keyed by
// logical ids (DESIGN.md "RNG remapping"); that mode is what the HIP path is
// compared against bit-for-bit.
#pragma once
#include <cstdint>
#include <cstddef>

extern "C" {

// Philox4x32-10 block (Salmon et al. 2011); ctr[4], key[2] -> out[4].
void scso_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out);

// Deterministic natural log (software, +,-,*,/ only; see scs_oracle.cpp).
double scso_det_log(double x);
double scso_det_exp(double x);   /* -256 <= x <= 0 */

// Run the whole genreads pipeline.  Returns 0 on success; on failure a message
// is left in scso_last_error().  `rng_mode`: 0 = reference streams, 1 = counter.
//   dump_prefix (nullable): writes <prefix>.frags.tsv / .semis.tsv / .fulls.tsv /
//   .reads.tsv intermediate tables for stage-level parity tests.
// Collectives for the sharded mode (fragment-lineage sharding of ONE job; world_size > 1 tests run them over
// torch.distributed/gloo).  allreduce: element-wise sum of n uint64 in place across shards.  allgatherv: every shard
// contributes send_bytes; recv holds shard_count slots of stride_bytes; sizes[r] = bytes received from shard r.
typedef int (*scso_allreduce_fn)(void* user, uint64_t* vals, uint64_t n);
typedef int (*scso_allgatherv_fn)(void* user, const void* send, uint64_t send_bytes, void* recv, uint64_t stride_bytes, uint64_t* sizes);

struct scso_params {
    const char* input_fasta;
    const char* profile;
    const char* output_prefix;
    const char* dump_prefix;     // may be NULL
    long   primers;              // -p  (default 100000)
    double gamma;                // -r  (default 1e-9)
    double coverage;             // -c  (default 5)
    int    isize;                // -s  (default 260)
    int    paired;               // -l  PE=1 / SE=0
    int    threads;              // -t  (counter mode only; ref mode is -t 1)
    int    rng_mode;             // 0 ref, 1 counter
    uint64_t seed;               // counter-mode seed
    long long fixed_time;        // ref-mode pinned clock (seconds), = SCS_FIXED_TIME
    int    verbose;
    int    shard_rank, shard_count;              // counter mode only; 0/1 = whole job
    scso_allreduce_fn  allreduce;                // required when shard_count > 1
    scso_allgatherv_fn allgatherv;
    void*  coll_user;
    const char* checksum_file;   // may be NULL.  Not NULL: no FASTQ files; per batch of batch_pairs planned pairs (0: 2^23) and mate the
    uint64_t batch_pairs;        //   checksum the library computes for its batches (scs_set_batch_checksums), one line per batch
    const char* checksum_batches;  // may be NULL: every batch.  "0,1,-1": those batches alone (negative = counted from the end)
};
void scso_default_params(scso_params* p);
int  scso_genreads(const scso_params* p);
const char* scso_last_error(void);

// Stage timings of the last scso_genreads call, seconds:
// [0] load+tables [1] fragments [2] amplify [3] allocate [4] readgen [5] pairs written
void scso_last_timings(double out[6]);

// ---- table-level access for unit tests ----------------------------------
// Loads a profile and builds the CDF tables exactly as Profile::load/normParas/
// initCDFs do.  Returns an opaque handle (NULL on error).
void* scso_profile_load(const char* path, int paired, int isize);
void  scso_profile_free(void* h);
int   scso_profile_read_length(void* h);
int   scso_profile_kmer_count(void* h);              // 84
// which: 0 subsCdf1 [84][bins][4], 1 subsCdf2, 2 qualCdf [16][bins][94],
//        3 insCdf, 4 delCdf, 5 isizeCdf, 6 gcMeans[101]
size_t scso_profile_table(void* h, int which, const double** data);
void  scso_profile_scalars(void* h, double out[8]);  // insertRate, delRate, stdISize, gcStd, isize_min, isize_count, haveCdf2, bins

// predict() on one window in counter mode (kernel-level parity).
//   window: n bases as codes 0..3 (ACGT), 4 = N.   uid/attempt/read select the
//   Philox substream exactly as the pipeline does.  out_bases/out_quals must
//   hold 2*n+64 bytes; returns the produced length n'.
int scso_predict_counter(void* h, const uint8_t* window, int n, int is_read1,
                         uint64_t seed, uint64_t uid, uint32_t attempt,
                         char* out_bases, char* out_quals);
// `count` windows in one call, outputs at stride 2 n + 64, lens[i] = n' : from the reference's streams (running on from read
// to read) / in counter mode (uid = first_uid + i).  For the statistical comparison of the two (tests/test_oracle_stats.py).
int scso_predict_ref_batch(void* h, const uint8_t* windows, int n, int count, const uint8_t* is_read1, unsigned seed,
                           char* out_bases, char* out_quals, int* lens);
int scso_predict_counter_batch(void* h, const uint8_t* windows, int n, int count, const uint8_t* is_read1, uint64_t seed, uint64_t first_uid,
                               char* out_bases, char* out_quals, int* lens);
// two more remapped units, the reference's draws (counter = 0) against counter mode (counter = 1): the attach tries of one primer on an
// empty template (tries used, 51 = gave up; position and length of the try that fit) and the GC factor of one GC percentage
int scso_attach_tries_batch(unsigned length, int amin, int amax, int count, int counter, uint64_t seed, uint32_t* tries, uint32_t* spos, uint32_t* alen);
int scso_gc_factor_batch(void* h, int gc, int count, int counter, uint64_t seed, double* out);
}
