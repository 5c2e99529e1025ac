/* scssim_hip.h -- C ABI of the MI355X-native `genreads` hot path.
 *
 * Drop-in boundary for qasimyu/scssim (reference paths below are relative to
 * the reference root).  The reference has no plugin/FFI layer: its genreads
 * driver (src/scssim.cpp:46-67) calls five methods on global objects.  Each
 * entry point here replaces one of those calls (or the pool job behind it), so
 * a maintainer swaps the bodies of those five calls for the functions below
 * (INTEGRATION.md shows the patch).  Plain C types only; every function
 * returns 0 on success or an SCS_E* code and never throws or exits.
 * One scs_ctx per host thread; a ctx owns one HIP device and one stream.
 */
#ifndef SCSSIM_HIP_H
#define SCSSIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCS_OK          0
#define SCS_EINVAL      1   /* bad argument / call order                         */
#define SCS_EIO         2   /* file could not be opened / malformed (the reference exit(1)/exit(-1)s) */
#define SCS_EDEVICE     3   /* HIP error, no device, out of memory               */
#define SCS_EOVERFLOW   4   /* a fixed-size device work buffer overflowed (never silent) */

typedef struct scs_ctx scs_ctx;

/* Mirrors the reference's Config defaults (lib/config/Config.cpp:13-49) and the
 * genreads options (src/scssim.cpp:285-404). */
typedef struct scs_config {
    int      device;            /* HIP device ordinal                                  */
    void*    stream;            /* hipStream_t to run on, or NULL = ctx-owned stream   */
    uint64_t seed;              /* counter-RNG seed (the reference seeds from time(): scssim.cpp:47) */
    long     primers;           /* -p  [100000]                                        */
    double   gamma;             /* -r  [1e-9]                                          */
    double   coverage;          /* -c  [5]                                             */
    int      isize;             /* -s  [260]                                           */
    int      paired;            /* -l  PE=1 / SE=0 [1]                                 */
    double   ber;               /* Config "ber" 3.4e-4                                 */
    int      amplicon_min_len;  /* 1000                                                */
    int      amplicon_max_len;  /* 2000                                                */
    int      frag_size;         /* Config "fragSize" 1000 (weight denominator)         */
    int      frag_min;          /* Fragment::minSize 10000 (lib/fragment/Fragment.cpp:15-16) */
    int      frag_max;          /* Fragment::maxSize 100000                            */
    int      shard_rank;        /* this process' shard (fragment-lineage sharding)     */
    int      shard_count;       /* number of shards; 1 = whole job                     */
    int      verbose;           /* progress lines on stderr as the reference prints    */
} scs_config;

typedef struct scs_stats {
    uint64_t records, genome_bases, fragments, semi_amplicons, full_amplicons;
    uint64_t primers_left;       /* Malbac::totalPrimers after amplify                 */
    uint64_t reads_requested, pairs_written, reads_written;
    uint64_t fastq_bytes[2];
    uint64_t algorithmic_bytes;  /* 1526 B per created amplicon + per pair (isize + FASTQ bytes) */
    double   t_stage[8];         /* seconds: load, frags, amplify, weights, allocate, yield, -, total */
    uint64_t sink_bytes[2];      /* bytes handed to the sink per mate: fastq_bytes, or their BGZF blocks' (scs_yield_reads_files_ex) */
    uint64_t staged_bases;       /* bases resident on this GPU: genome_bases, or -- a shard of a sharded job loaded from an indexed FASTA --
                                    only the stretch its own fragments cover (scs_load_genome_fasta) */
    /* the primer stock of scs_amplify (Malbac::updatePrimerCount, lib/malbac/Malbac.cpp:91-103): passes whose demand was compared
     * with the stock (a pass that cannot reach the smallest stock is not), passes in which a primer type ran dry, and the
     * rounds it took to give such a type to exactly its first `stock` attachments in list order */
    uint64_t stock_checks, stock_exhausted_passes, stock_rounds;
} scs_stats;

void        scs_default_config(scs_config* cfg);
int         scs_create(const scs_config* cfg, scs_ctx** out);
void        scs_destroy(scs_ctx* ctx);
/* message of the last failure on ctx (or of the last failed scs_create when ctx == NULL) */
const char* scs_last_error(const scs_ctx* ctx);
int         scs_set_seed(scs_ctx* ctx, uint64_t seed);

/* Profile::train(file) = load + normParas(true) + initCDFs  (lib/profile/Profile.cpp:1432-1436).
 * Parses the .profile text, builds the CDF tables in double exactly as the reference, converts
 * every CDF entry to the exact uint32 draw threshold, uploads them.  Sets the read length. */
int         scs_load_profile(scs_ctx* ctx, const char* profile_path);
int         scs_read_length(const scs_ctx* ctx);

/* Genome::loadData for genreads = Genome::loadRefSeq (lib/genome/Genome.cpp:18-25,176-195):
 * simuvars-style FASTA (records <chr>_<hap>_<reflen>); ".gz" is inflated with `gzip -cd` as there. */
int         scs_load_genome_fasta(scs_ctx* ctx, const char* fasta_path);
/* A shard of a sharded job (shard_count > 1) stages only ITS OWN stretch of the genome when the file has a fastahack index beside
 * it (<fasta>.fai, not older than the file) that describes it (regular lines): the fragment split needs the record lengths alone
 * (lib/genome/Genome.cpp:753-782), so the shard reads, uploads, encodes and indexes just the byte ranges its fragments cover
 * (scs_stats.staged_bases).  The split depends on the seed: after scs_set_seed load the genome again (scs_create_frags says so).
 * Without a usable index the whole file is staged, which also writes the index. */
/* Same, from memory: names[i] as they appear after '>' ; seqs[i] = lens[i] ASCII bases. */
int         scs_upload_genome(scs_ctx* ctx, int n_records, const char* const* names,
                              const char* const* seqs, const uint64_t* lens);

/* Same, with the bases already in device memory (HBM-resident producers: a genome edited on the GPU, a synthetic
 * benchmark genome): d_bases = the records' ASCII bases concatenated without separators, sum(lens) bytes, readable
 * on the ctx device; copied, the caller keeps ownership. */
int         scs_upload_genome_device(scs_ctx* ctx, int n_records, const char* const* names, const uint64_t* lens,
                                     const void* d_bases);

/* `scssim simuvars` (src/scssim.cpp:33-38, 108-170) on the data plane: Genome::loadData (loadAbers lib/genome/Genome.cpp:35-165,
 * loadSNPs -> lib/snp/snp.cpp:147-203, loadRefSeq 176-195) + Genome::saveSequence (329-384) / generateSegment (386-691).
 * ref_fasta: plain reference (records chr<k>); snp_file / var_file: the reference's formats, either may be NULL.
 * The host only plans (segments, copies, substitutions, insertions, deletions as a list of pieces; the reference's rand()
 * draws reproduced); the two haplotypes of every chromosome are built in HBM from the resident reference and stay resident
 * exactly as if scs_load_genome_fasta had read the simuvars output -- genreads can follow with no intermediate FASTA.
 * out_fasta != NULL additionally writes that file, byte-identical to the reference's (records <chr>_<hap>_<reflen>, 100 columns). */
int         scs_simuvars(scs_ctx* ctx, const char* ref_fasta, const char* snp_file, const char* var_file, const char* out_fasta);

/* Malbac::createFrags -> Genome::splitToFrags + Fragment::createSequence
 * (lib/malbac/Malbac.cpp:143-145, lib/genome/Genome.cpp:753-782, lib/fragment/Fragment.cpp:40-50) */
int         scs_create_frags(scs_ctx* ctx);

/* Malbac::amplify (lib/malbac/Malbac.cpp:173-201): createPrimers, setPrimers, and the 1+5 cycles of
 * Fragment::batchAmplify / Amplicon::batchAmplify pool jobs (Fragment.cpp:52-152, Amplicon.cpp:156-253).
 * Amplicons stay resident in HBM. */
int         scs_amplify(scs_ctx* ctx);

/* Malbac::setReadCounts (Malbac.cpp:370-408) incl. Amplicon::getWeightedLength (Amplicon.cpp:396-400),
 * Profile::getGCFactor (Profile.cpp:1503-1513) and randIndx_hp/batchSampling (MyDefine.cpp:191-272).
 * reads == 0: derive it from the record names and coverage as Malbac::yieldReads does (Malbac.cpp:413-420). */
int         scs_allocate_reads(scs_ctx* ctx, uint64_t reads);

/* Sink = SeqWriter::write(char*) / write(char*,char*) (lib/seqwriter/SeqWriter.cpp:41-54).
 * Called in output order with host buffers valid only during the call; fq2/n2 are NULL/0 for SE.
 * Return non-zero to abort. */
typedef int (*scs_sink_fn)(void* user, const char* fq1, size_t n1, const char* fq2, size_t n2);

/* Malbac::yieldReads fan-out + Amplicon::yieldReads jobs (Malbac.cpp:436-457, Amplicon.cpp:402-565)
 * with Profile::predict (Profile.cpp:1582-1697) per read.  FASTQ text is produced on the device
 * and handed to `sink` batch by batch (sink may be NULL: generate and count only). */
int         scs_yield_reads(scs_ctx* ctx, scs_sink_fn sink, void* user);

/* Same, but the FASTQ pool stays in HBM in caller-owned device buffers (for the RCCL gather of the
 * read pool).  Fails with SCS_EOVERFLOW if a capacity is too small; n1 / n2 receive the byte counts. */
int         scs_yield_reads_device(scs_ctx* ctx, void* d_fq1, size_t cap1, void* d_fq2, size_t cap2,
                                   uint64_t* n1, uint64_t* n2, uint64_t* pairs);

/* createFrags + amplify + allocate + yield in one call (the body of main()'s genreads branch,
 * src/scssim.cpp:59-65). */
int         scs_run_genreads(scs_ctx* ctx, scs_sink_fn sink, void* user);

int         scs_get_stats(const scs_ctx* ctx, scs_stats* out);

/* Integrity of text that never leaves the GPU (a NULL sink) or crosses PCIe: with on != 0 every batch of the next scs_yield_reads /
 * scs_yield_reads_files gets a 64-bit checksum per mate, computed by a kernel where the text lies in HBM -- the text as
 * little-endian 64-bit words w_i (the last zero-padded): sum_i fmix64(w_i + (i + 1) * 0x9E3779B97F4A7C15) mod 2^64, fmix64 = the
 * MurmurHash3 finaliser.  scs_batch_checksums: out[2 b], out[2 b + 1] = batch b's two mates, in record order (cap: entries of
 * out); *n_batches = batches of the last call.  Not computed for scs_yield_reads_device. */
int         scs_set_batch_checksums(scs_ctx* ctx, int on);
int         scs_batch_checksums(const scs_ctx* ctx, uint64_t* out, size_t cap, size_t* n_batches);

/* ---- one job over several GPUs (scs_config.shard_rank / shard_count: fragment-lineage sharding) -------------
 * The reference is single-process; these are the exchange steps its globals imply once fragments are split over
 * ranks: Malbac::setPrimers totals (Malbac.cpp:242-262,282), the primer stock (Malbac.cpp:91-103) and the weight
 * normalisation / chunked sampling of Malbac::setReadCounts (Malbac.cpp:370-408).  The caller supplies them
 * (torch.distributed over RCCL or gloo: scssim_amd/dist.py); buffers are host memory.
 *   allreduce : element-wise sum of n uint64 values in place over all shards
 *   allgatherv: every shard sends send_bytes; recv has shard_count slots of stride_bytes; sizes[r] = bytes of shard r
 * With the hooks set, a sharded job writes record names / read counts identical to the unsharded job; each shard's
 * FASTQ pool is sorted by the amplicon index in the record name, so the writer k-way merges the pools. */
typedef int (*scs_allreduce_fn)(void* user, uint64_t* vals, uint64_t n);
typedef int (*scs_allgatherv_fn)(void* user, const void* send, uint64_t send_bytes, void* recv, uint64_t stride_bytes, uint64_t* sizes);
int         scs_set_collectives(scs_ctx* ctx, scs_allreduce_fn allreduce, scs_allgatherv_fn allgatherv, void* user);
/* Device-memory variants, ordered on the ctx stream (no host sync): used for the per-cycle scalars, the per-pass
 * primer-stock decrements and the weight gather when set; the host hooks above remain the fallback.
 *   allreduce_dev: sum n elements of elem_bytes (4 = uint32, 8 = uint64) in place
 *   allgather_dev: d_recv[r * bytes_per_rank ..] = shard r's d_send[0 .. bytes_per_rank) */
typedef int (*scs_allreduce_dev_fn)(void* user, void* d_vals, uint64_t n, int elem_bytes);
typedef int (*scs_allgather_dev_fn)(void* user, const void* d_send, void* d_recv, uint64_t bytes_per_rank);
int         scs_set_collectives_device(scs_ctx* ctx, scs_allreduce_dev_fn allreduce_dev, scs_allgather_dev_fn allgather_dev, void* user);

/* RCCL inside the library (one process per GPU; no caller-side hooks needed).  Rank 0 obtains an id (ncclGetUniqueId) and
 * hands its SCS_COMM_ID_BYTES to the other ranks by any means (a pipe, a file, MPI, torch.distributed); every rank then
 * calls scs_comm_init on its ctx (ncclCommInitRank on the ctx device).  From then on the exchanges above run as RCCL
 * all-reduce / all-gather on the ctx stream.  RCCL is bound at run time (dlopen), so the library loads without it. */
#define SCS_COMM_ID_BYTES 128
int         scs_comm_unique_id(void* id_out);
int         scs_comm_init(scs_ctx* ctx, const void* id, int rank, int nranks);
/* ranks of the ctx's communicator as RCCL reports them (ncclCommCount); 0 without a communicator */
int         scs_comm_count(const scs_ctx* ctx);
/* ncclCommAbort on the ctx's communicator, callable from ANOTHER thread than the one blocked in a collective: how a driver
 * that has seen a rank die releases its own rank before it leaves (the ctx is only good for scs_destroy afterwards) */
int         scs_comm_abort(scs_ctx* ctx);

/* ---- FASTQ straight to files (SeqWriter, lib/seqwriter/SeqWriter.cpp:12-64; opened by Malbac::yieldReads, Malbac.cpp:426-435)
 * writers <= 1: the reference's files.  Whole job (shard_count == 1): <prefix>_1.fq / <prefix>_2.fq, or <prefix>.fq for SE.
 * Sharded job: this shard's records go to <prefix>.r<rank>_1.fq / _2.fq (.fq) and <prefix>.r<rank>.idx lists the byte offset
 * at which each of the shard's list segments starts.  The whole job's file is the shards' segments interleaved in list
 * order, so scs_merge_fastq_shards rebuilds it by copying byte ranges (copy_file_range, a few threads) -- it parses no
 * record -- and the result equals the unsharded job's files byte for byte.
 * writers = K > 1 (at most 64): buffered writes into ONE file serialise on its inode lock (5.7 GB/s per file on the GPU
 * box's tmpfs whatever the thread count), so the job's (shard's) records are cut into K contiguous ranges, made round-robin,
 * and written by K threads into K PART files per mate: <base>.p00_1.fq ... <base>.p<K-1>_1.fq (+ _2.fq; .p<kk>.fq for SE),
 * base = <prefix> or <prefix>.r<rank>.  Their concatenation in that order IS the single file (`cat <base>.p*_1.fq`);
 * <base>.parts lists their sizes; scs_merge_fastq_parts / scs_merge_fastq_shards read parts and single files alike. */
int         scs_yield_reads_files(scs_ctx* ctx, const char* prefix, int writers);
/* The same with two more choices.
 * generations = G > 1: writers x G parts per mate, made generation by generation (the first `writers` parts, then the next ...):
 *   part p is complete -- closed, final -- as soon as part p + writers exists, so a consumer can stream the early parts while the
 *   job runs and the page cache holds a couple of generations instead of the whole job's text (at most 99 parts).
 * bgzf != 0: the files are <...>.fq.gz in BGZF (blocked gzip: what `bgzip` writes; gzip / zcat, htslib, bwa, samtools read it).
 *   The blocks are made ON THE GPU from the text where it lies in HBM (scs_bgzf.hip: one dynamic-Huffman deflate block of
 *   literals per 63 KB of text, CRC-32 included), so 3-4x fewer bytes cross PCIe and reach the file system -- the two walls of a
 *   job.  `zcat` of a part is the text of that part; every part ends with the BGZF end-of-file block.  An extension: the
 *   reference writes plain text only.  The shards of a sharded job stay shards (compressed byte ranges cannot be spliced).
 * flags: SCS_SINK_BGZF (1; `bgzf` was this argument's name when it was the only choice) | SCS_SINK_IN_PLACE (2): output files that
 *   exist already are not truncated when they are opened (what `ofstream` does, SeqWriter.cpp:17-30) but overwritten where they lie
 *   and cut to their new length when they are finished -- the same files in the end, and a job that replaces the files of an earlier
 *   one does not pay for giving their pages back and taking them again.  Until a file is finished its tail is the old file's.
 *   With generations > 1 the "part p + writers exists => part p is final" signal is kept: the call first renames the earlier job's
 *   files of the LATER generations to <name>.prev, and each comes back under its name (its pages kept) when its part's first batch
 *   arrives; no .prev file is left when the call returns. */
#define SCS_SINK_BGZF     1
#define SCS_SINK_IN_PLACE 2
int         scs_yield_reads_files_ex(scs_ctx* ctx, const char* prefix, int writers, int generations, int flags);
int         scs_merge_fastq_shards(const char* prefix, int nranks, int paired, int keep_shards, char* errbuf, size_t errlen);
/* host only: <prefix>.p*_1.fq ... -> <prefix>_1.fq ... (byte-range copies; parts removed unless keep_parts) */
int         scs_merge_fastq_parts(const char* prefix, int paired, int keep_parts, char* errbuf, size_t errlen);

/* ---- kernel-level entry points (unit parity tests; same kernels as the pipeline) ------------ */

/* char* Profile::predict(char* refSeq, int isRead1)  (lib/profile/Profile.cpp:1582-1697) for a batch:
 * windows = n_reads x L base codes (0..3 = ACGT, 4 = N) on the HOST; per read the lineage uid,
 * the attempt number and the read-1 flag select the counter-RNG substream.  out_bases/out_quals:
 * n_reads x out_stride chars; out_len[i] = produced length n'. */
int         scs_predict_batch(scs_ctx* ctx, const uint8_t* windows, size_t n_reads,
                              const uint64_t* uids, const uint32_t* attempts, const uint8_t* is_read1,
                              char* out_bases, char* out_quals, int32_t* out_len, int out_stride);

/* Philox4x32-10 on the device for n counters (ctr: n x 4, key: 2, out: n x 4; host pointers). */
int         scs_philox_batch(scs_ctx* ctx, const uint32_t* ctr, size_t n, const uint32_t* key, uint32_t* out);
/* det_log on the device (host pointers); an argument <= 0 is evaluated as det_exp instead (the product-form Poisson's
 * exp(-lambda), -256 <= x <= 0), so one entry point serves both deterministic functions. */
int         scs_detlog_batch(scs_ctx* ctx, const double* x, size_t n, double* out);

/* Download the amplicon tables (kind 0 = semi, 1 = full) for stage-level parity tests.  Any pointer
 * may be NULL.  Arrays must hold scs_stats.{semi,full}_amplicons entries; errs: up to 4 packed
 * (pos<<3|alt) entries per amplicon in errs[4*i..], count in nerr[i] (>4 = overflow list truncated). */
int         scs_download_amplicons(scs_ctx* ctx, int kind, uint32_t* parent, uint32_t* spos, uint32_t* len,
                                   uint32_t* gc, uint32_t* primers, uint64_t* uid, uint32_t* errs, uint32_t* nerr);
int         scs_download_read_numbers(scs_ctx* ctx, uint32_t* read_numbers);
/* The primer pool after scs_amplify: stock[65536], copies left of every primer type (PrimerIndex.count, lib/malbac/Malbac.h:18-24;
 * index = the 8-mer at two bits per base, first base in the top bits). */
int         scs_download_primer_stock(scs_ctx* ctx, int64_t* stock);
/* The CPUs of the NUMA node `device` hangs on that this process may run on (cpus[0..cap), returns their number; 0: unknown, or the
 * whole affinity mask is local).  The library's sink threads and pinned buffers bind themselves to them; a caller with host threads of
 * its own around the sink (bench.py's cleaners) can do the same. */
int         scs_gpu_local_cpus(int device, int* cpus, int cap);
/* Test seams (csrc/scs_seams.h: small batches, forced kernel variants, injected failures) exist only in libscssim_hip_seams.so, the
 * build the tests load; there this returns the seam's value.  In libscssim_hip.so it returns NULL for every name: the product reads
 * no such knob. */
const char* scs_test_seam(const char* name);

/* ---- host-only table access (no GPU needed): the thresholds scs_load_profile uploads -------------
 * which: 0 subs read1 [84][bins][4], 1 subs read2, 2 quality [16][bins][94], 3 insert length,
 *        4 deletion length, 5 insert size, 6 the alias rows of the quality tables that the inject_errors kernel draws from
 *        (uint32 words, K + K/4 per row, K = 16 / 64 / 128 columns: scssim_amd/csrc/scs_tables.h; no cdf).
 *        thr/cdf point into memory owned by the handle. */
int         scs_profile_open(const char* profile_path, int paired, int isize, void** handle, char* errbuf, size_t errlen);
int         scs_profile_table(void* handle, int which, const uint32_t** thr, const double** cdf, size_t* n);
/* out[0..9] = read length, bins, t_insert, t_delete, isize_min, have_cdf2, insert_rate, del_rate, t_indel (one-draw
 * insertion/deletion test), columns per alias quality row (16, 64 or 128) */
int         scs_profile_scalars(void* handle, double* out);
void        scs_profile_close(void* handle);

/* Host-only: parse a FASTA exactly as scs_load_genome_fasta stages it (no GPU).  names_buf receives the index
 * names joined by '\n'; checksum = FNV-1a over the upper-cased sequence bytes of all records in order. */
int         scs_fasta_probe(const char* fasta_path, int* n_records, uint64_t* total_bases, uint64_t* checksum,
                            char* names_buf, size_t names_len, char* errbuf, size_t errlen);
/* Host-only: plan scs_simuvars for these inputs and return the record count, the total haplotype bases and the FNV-1a
 * checksum of the FASTA text scs_simuvars would write (no GPU; the test seam of the planner). */
int         scs_simuvars_probe(const char* ref_fasta, const char* snp_file, const char* var_file, int* n_records, uint64_t* total_bases,
                               uint64_t* checksum, char* errbuf, size_t errlen);
/* Test seam of the device-buffer policy (needs a GPU, touches no ctx): a library buffer is reserved with first_bytes, then
 * with second_bytes; caps[0..1] receive its usable capacity after each step and *in_place whether the second step kept its
 * address.  Buffers above 64 MB (SCS_VMM_FROM_MB) live in a reserved address range and grow in place, by the request + 3 %. */
int         scs_devbuf_probe(int device, uint64_t first_bytes, uint64_t second_bytes, uint64_t* caps, int* in_place);
/* Host-only test seam of the BGZF kernels (scs_yield_reads_files_ex, bgzf): their arithmetic -- Huffman lengths, header, chunked
 * bit packing, chunked CRC-32 combined by carry-less multiplication, the stored fallback when a block's deflate data exceeds
 * lds_out_cap (0 = the kernels' limit) -- run on the CPU over the same functions; out receives the BGZF blocks of the text
 * (no end-of-file block), *n_out their size (out may be NULL to ask for it).  The checker is zlib. */
int         scs_bgzf_probe(const void* text, uint64_t nbytes, uint32_t lds_out_cap, void* out, uint64_t cap, uint64_t* n_out);
/* Host-only: leave <fasta_path>.fai beside the file if there is none, exactly as scs_load_genome_fasta does (the
 * reference indexes its input through fastahack, lib/fastahack/Fasta.cpp:241-249: name, length, offset, bases per
 * line, bytes per line). */
int         scs_fasta_write_index(const char* fasta_path, char* errbuf, size_t errlen);

/* Per-kernel timing (HIP events recorded on the ctx stream around every launch, accumulated over the
 * last scs_amplify / scs_yield_reads call): name, launches, total milliseconds, and the units the
 * launches processed (amplicons created for the errscan kernels, read pairs for k_reads/k_indels,
 * templates for the two k_attach instances).  which = 0..5: k_errs<semi->full>, k_errs<frag->semi>, k_reads,
 * k_attach<semi>, k_indels, k_attach<frag>. */
int         scs_kernel_time(const scs_ctx* ctx, int which, const char** name, uint64_t* launches, double* ms, uint64_t* units);
/* Which of the six kernels get their HIP event pairs: bit `which` of mask (default: all), and on which calls: every
 * `every`-th scs_amplify / scs_yield_reads call counted from this call (default 1 = all).  Every event record is a
 * packet on the stream (about 6 us each on the latency-bound 1 Mb configuration), so a measurement run times only the
 * kernel of interest, on a sample of the steps.  scs_kernel_time reports an untimed call as 0 launches / 0 units. */
int         scs_set_kernel_timing(scs_ctx* ctx, unsigned mask, unsigned every);

#ifdef __cplusplus
}
#endif
#endif /* SCSSIM_HIP_H */
