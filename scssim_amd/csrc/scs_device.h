// scs_device.h -- device-side data layout (SoA in HBM) and kernel launch wrappers.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "scs_common.h"
#include "scs_simuvars.h"

namespace scs {

// ---- genome + fragments (reference: Fragment, lib/fragment/Fragment.h:20-31) ---------------------
// The genome stays resident once, 1 byte per base (codes 0..3, 4 = N).  A fragment is an index map
// into it, never a copy: template strand T_f[i] = strand>0 ? comp(G[goff+len-1-i]) : G[goff+i]
// (Fragment::createSequence + the complement taken in Fragment::amplify, Fragment.cpp:40-50,65-68).
struct DevFrags {
    const uint64_t* goff;      // offset of the slice start in the genome array
    const uint32_t* len;
    const int8_t*   strand;    // +1 / -1
    const uint32_t* primers;   // budget of the current pass (Fragment::primerNum)
    uint32_t n;
    uint64_t gidx_base;        // global index of local fragment 0 (sharding)
    const uint8_t* has_n;      // 1: the fragment contains a non-ACGT base (k_frag_has_n); windows of the others skip the N count
};

// ---- amplicons (reference: Amplicon + AmpliconNode list, lib/amplicon/Amplicon.h:47-76) -----------
// Flat SoA, index = position in the reference's -t 1 list order.
struct DevAmps {
    uint32_t* parent;          // semis: local fragment index; fulls: semi index
    uint32_t* sl;              // pack_sl(spos, len)
    uint16_t* gc;
    uint16_t* primers;         // semis: 12-bit budget of the current cycle (Amplicon.cpp:76-79)
    uint64_t* uid;             // lineage uid (counter-RNG key)
    uint64_t* errs;            // 4 inline u16 entries or overflow reference
};

// genome bit index (k_genome_bits): per 64-base word, G/C and N masks + counts before the word
struct DevGenomeIdx { const unsigned long long* gc_bits; const unsigned long long* n_bits; const uint64_t* gc_pref; const uint64_t* n_pref;
                      const ulonglong2* gc_pair; };   // gc_pair[w] = {gc_bits[w], gc_pref[w]}: ONE 16-byte load per end of a window (k_errs reads two random places of the index per amplicon: two lines instead of four)

struct DevErrPool { uint32_t* data; uint32_t* head; uint32_t cap; };

// view of a template strand as an index map into the genome: T[i] = maybe_comp(G[base + dir*i])
struct View { int64_t base; int32_t dir; uint32_t comp; };

struct DevTables {
    int L, bins;
    uint32_t t_insert, t_delete, t_indel, t_ber;       // t_indel: one-draw indel test (scs_tables.h)
    const uint32_t* gap_t; uint32_t t_kind;            // [REMAP] geometric gap between indel events (scs_tables.h): gap >= g <=> x < gap_t[g]
    const uint32_t* subs1; const uint32_t* subs2;   // [84][bins][4] thresholds (uint4 rows)
    const uint32_t* qual;                            // [16][bins][94]
    const uint4* ring1; const uint4* ring2;          // per mate, per bin (padded to a multiple of 8 bins): the bin's image in k_reads' LDS ring (RingBin), ready to copy
    const uint4* ring1u; const uint4* ring2u;        // the same for the uniform walk (RingBinU): alias rows + the 64 3-mers' KEEP intervals (lo, width) instead of threshold triples
    const uint32_t* qual_alias; int qual_k;          // [16*bins] alias rows of qual_k (16 / 64 / 128) columns: qual_k words + qual_k symbol bytes (scs_tables.h)
    const uint32_t* ins_t; int n_ins;
    const uint32_t* del_t; int n_del;
    const uint32_t* isize_t; int n_isize; int isize_min;
    const double* subs1_d; const double* subs2_d; const double* qual_d;   // slow path (x == 0xFFFFFFFF)
    const double* ins_d; const double* del_d; const double* isize_d;
    const double* gc_means; double gc_std;
};

struct AmplifyParams {
    RngKey key;
    uint32_t pass;             // fragment pass 0..4 / semi cycle 0..4
    uint32_t amp_min, amp_max; // 1000, 2000
    uint32_t t_ber;
};

struct PoissonParams {
    RngKey key; uint32_t call; double gamma;
    uint64_t total_primers;                            // Malbac::setPrimers inputs (Malbac.cpp:236-262)
    const uint64_t* totals;                            // sharded job: device {template_num, total_len} over ALL shards; else null
    // unsharded job: templateNum = nf + dev[DS_SEMIS_N], totalLen = frag_len + dev[DS_SEMI_LEN].  The semi amplicon
    // count and length live on the device: setPrimers of a cycle is launched before the host has read them back.
    uint64_t nf, frag_len; const unsigned long long* dev;
    const unsigned long long* total_primers_dev;       // sharded job: the pool size lives on the device too (else null)
};
// device scalars of an amplification run (scs_ctx::dsums): [0],[1] budget sums of the current setPrimers call,
// [4] total length of this shard's semi amplicons, [5] their number.  Sharded job, whole-job values kept current by the
// tail of the per-pass primer all-reduce (k_shard_tail / k_primer_update_sharded): [6] semi length already reported,
// [8] semis, [9] their length, [10] primers left in the pool, [11],[12] {templateNum, totalLen} for setPrimers,
// [13],[14] fragments and their length over all shards (constants)
enum { DS_HOLES = 20 };                                // pairs planned but not produced (k_plan_pairs; zeroed by scs_yield_reads)
enum { DS_SEMI_LEN = 4, DS_SEMIS_N = 5, DS_REPORTED_LEN = 6, DS_G_SEMIS_N = 8, DS_G_SEMI_LEN = 9, DS_G_PRIMERS = 10, DS_G_TOTALS = 11, DS_G_NF = 13, DS_G_FRAG_LEN = 14,
       DS_MIN_STOCK = 15 };                            // the smallest primer stock > 0 seen at the end of any pass so far (a lower bound of every stock in use)
enum { SHARD_TAIL_WORDS = 16 };                       // u32 words behind the 65536 primer decrements that ride on the same all-reduce
struct AllocState { double total; unsigned long long sum_rn, sum_quota; };

// ---- read allocation over the whole job's amplicon list, computed by the shards (scs_k_allocate.hip, K3) ---------------
// The list is cut into chunks of 1000 (randIndx_hp's chunks, MyDefine.cpp:203-253).  A shard's local list is the
// concatenation of its segments (cycle ascending, fragment pass descending: slot = cycle * 8 + (7 - pass)); the whole
// job's list interleaves the shards' segments slot by slot (order = slot * shards + rank).
#define ALLOC_CHUNK 1000u
enum { ALLOC_SLOTS = 40 };
struct AllocRange { uint32_t q0, c0, local0; };            // work chunks q0.. (up to the next range) = whole-job chunks c0.., in place at local0 + 1000 (q - q0)
struct AllocBChunk { uint32_t c, n, owner; };              // a chunk that straddles a segment boundary: materialised row; owner: this shard holds its first amplicon
struct AllocGSeg { unsigned long long go; uint32_t lo, n, owner, slot; };   // every non-empty segment of every shard, in list order
struct AllocMySeg { unsigned long long go; uint32_t lo, n, order, pad; };  // this shard's segment of slot s
struct AllocPlan {
    unsigned long long total;                              // amplicons of the whole job
    uint32_t rank, n_interior, n_boundary, n_ranges, n_gseg;
    const AllocRange* rng; const AllocBChunk* bchunk; const AllocGSeg* gseg;    // device arrays
    AllocMySeg my_seg[ALLOC_SLOTS];
};
// local amplicon index -> index in the whole job's list (the number printed in the record name); n == 0: identity
struct SegMap { uint32_t n; uint32_t lo[ALLOC_SLOTS], cnt[ALLOC_SLOTS]; unsigned long long go[ALLOC_SLOTS]; };

// error flags raised by kernels (never silent): bit 0 error-list cap, 1 error pool, 2 read slot, 3 other
enum DevFlag : uint32_t { FLAG_ERRCAP = 1, FLAG_ERRPOOL = 2, FLAG_READSLOT = 4, FLAG_INTERNAL = 8, FLAG_KEYSPACE = 16 };

// one planned read pair (or SE read) with its amplicon already resolved to an index map into the genome:
// U[t] = maybe_comp(G[base + dir*t]) patched by the semi's errors (at t = k1 - pos(e), value comp(alt))
// and then the full amplicon's own (at t = pos(e), value alt).  56 bytes per pair.
struct PairRec {
    uint32_t amp, att, pos, isz;      // isz == 0: hole
    int64_t  base; uint32_t flags;    // flags: bit0 complement, bit1 direction is -1, bit2 the fragment holds a non-ACGT base
    int32_t  k1;                      // l_semi - 1 - spos_full
    uint64_t e1, e2, uid;             // error words of the semi / the full amplicon; lineage uid
};

// ---- launch wrappers (scs_k_*.hip) --------------------------------------------------------------
// one pass of primer attachment over the templates [t_first, t_end) of the pass's list; primer_cut: the stock as k_attach sees it
// (scs_k_amplify.hip, "the primer stock, exactly"); undo: the templates were run before in this pass -- what they took then is taken back first
void launch_attach_frags(hipStream_t s, const uint8_t* g, DevFrags fr, const uint32_t* slot_off, uint32_t* slots,
                         uint32_t* slot_tmpl, uint32_t* valid, const unsigned long long* primer_cut, uint32_t* primer_delta,
                         unsigned long long* len_part, AmplifyParams p, uint32_t t_first, uint32_t t_end, int undo, const unsigned long long* t_from);   // len_part: one slot per fragment; t_from (device, or null): skip the templates before it
void launch_frag_len_sum(hipStream_t s, const unsigned long long* len_part, uint32_t nf, unsigned long long* len_sum);   // *len_sum += the pass's amplicon lengths
void launch_poisson(hipStream_t s, DevFrags fr, DevAmps semis, uint32_t n_semis, PoissonParams p, uint32_t* budget_f, uint32_t* budget_s,
                    unsigned long long* sums, unsigned long long* part, uint32_t* flags);   // part: one slot per workgroup (fragments + ceil((n_semis + 1) / 256))
// read allocation, stage by stage (the pipeline puts the shards' exchanges between them)
void launch_alloc_bpack(hipStream_t s, const double* w, const AllocPlan& pl, double* send);
void launch_alloc_bgather(hipStream_t s, const double* w, const AllocPlan& pl, const double* gathered, double* brow, int* bmap);
void launch_alloc_chunk_sum(hipStream_t s, const double* w, const double* brow, const AllocPlan& pl, double* part);
void launch_tree_sum(hipStream_t s, const double* part, uint32_t nch, double* scratch, double* total);     // scratch: nch/1000 + 1024 doubles
void launch_alloc_norm(hipStream_t s, double* w, double* brow, const int* bmap, const AllocPlan& pl, const double* total, unsigned long long reads,
                       uint32_t* rn, double* tp, uint32_t* crn, unsigned long long* sum_rn);
void launch_alloc_quota(hipStream_t s, const double* tp, uint32_t nch, unsigned long long reads, const unsigned long long* sum_rn, unsigned long long* sum_quota,
                        uint32_t* quota, double* probs, double* scratch, RngKey key);   // scratch: 3 * (nch / 1000) + 4096 doubles
void launch_alloc_sample(hipStream_t s, const double* w, const double* brow, const int* bmap, const AllocPlan& pl, const double* tp, const uint32_t* quota, RngKey key, uint32_t* rn);
void launch_alloc_odd_scan(hipStream_t s, const uint32_t* rn, uint32_t ac, uint32_t* odd_before, void* temp, size_t temp_bytes);
void launch_alloc_odd_counts(hipStream_t s, const uint32_t* odd_before, const AllocPlan& pl, unsigned long long* table);
void launch_alloc_parity(hipStream_t s, uint32_t* rn, const uint32_t* odd_before, uint32_t ac, const AllocPlan& pl, const unsigned long long* table);
void launch_attach_semis(hipStream_t s, const uint8_t* g, DevFrags fr, DevAmps semis, uint32_t n_semis, DevErrPool spool,
                         const uint32_t* slot_off, uint32_t* slots, uint32_t* slot_tmpl, uint32_t* valid,
                         const unsigned long long* primer_cut, uint32_t* primer_delta, AmplifyParams p, uint32_t t_first, uint32_t t_end, int undo, const unsigned long long* t_from);
// the dense form of the semi pass (one lane = one primer): a plan per pass, then the pass over every wave's templates (scs_k_amplify.hip)
uint32_t attach_dense_waves(uint32_t n_slots);
void launch_attach_plan(hipStream_t s, const uint32_t* slot_off, uint32_t nt, uint32_t n_slots, uint32_t* item_tmpl, uint32_t* wave_first, uint32_t* valid);
void launch_attach_dense(hipStream_t s, const uint8_t* g, DevFrags fr, DevAmps semis, DevErrPool spool, const uint32_t* slot_off, uint32_t* slots, const uint32_t* item_tmpl,
                         const uint32_t* wave_first, uint32_t n_slots, uint32_t* valid, const unsigned long long* primer_cut, uint32_t* primer_delta, AmplifyParams p, int undo,
                         const unsigned long long* t_from);
// exact primer stock: over-/under-demand of the pass so far (info: 8 words), the attachments of the over-demanded types, their sort, the new cuts
void launch_stock_check(hipStream_t s, const int64_t* cnt, const uint32_t* taken, unsigned long long* cut, bool from_frag, unsigned long long* info);
void launch_stock_list(hipStream_t s, const int64_t* cnt, const uint32_t* taken, uint32_t* eidx, uint32_t* etype, uint32_t* estart, unsigned long long* info);   // only when the check found something
void launch_stock_collect(hipStream_t s, const uint8_t* g, DevFrags fr, DevAmps semis, DevErrPool spool, bool from_frag, const uint32_t* slot_off, const uint32_t* slots,
                          const uint32_t* valid, const uint32_t* eidx, unsigned long long* list, unsigned long long* info, uint32_t t_first, uint32_t t_end);
size_t stock_sort_temp_bytes(size_t n);
void launch_stock_sort(hipStream_t s, const unsigned long long* in, unsigned long long* out, size_t n, void* temp, size_t temp_bytes);
void launch_stock_pick(hipStream_t s, const int64_t* cnt, const uint32_t* etype, const uint32_t* estart, uint32_t ne, const unsigned long long* sorted, unsigned long long* cut,
                       bool from_frag, unsigned long long* info);
void launch_stock_apply(hipStream_t s, int64_t* cnt, uint32_t* gdelta, uint32_t* delta, unsigned long long* cut, uint32_t* flags);
void launch_errs_frags(hipStream_t s, const uint8_t* g, DevGenomeIdx gx, DevFrags fr, uint32_t n_slots, const uint32_t* slot_off, const uint32_t* slots,
                       const uint32_t* slot_tmpl, const uint32_t* valid_off, DevAmps out, uint32_t out_base, DevErrPool pool, uint32_t* flags,
                       const unsigned long long* binom, AmplifyParams p, int64_t* primer_cnt, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* sums,
                       unsigned long long* semis_n);
void launch_errs_semis(hipStream_t s, const uint8_t* g, DevGenomeIdx gx, DevFrags fr, DevAmps semis, uint32_t n_semis, DevErrPool spool, uint32_t n_slots,
                       const uint32_t* slot_off, const uint32_t* slots, const uint32_t* slot_tmpl, const uint32_t* valid_off,
                       DevAmps out, uint32_t out_base, DevErrPool pool, uint32_t* flags, const unsigned long long* binom, AmplifyParams p,
                       int64_t* primer_cnt, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* sums);   // primer_cnt non-null: the pass's stock update rides along (unsharded job)
void launch_encode_bases(hipStream_t s, uint8_t* g, uint64_t n);
void launch_fa_gather_regular(hipStream_t s, const uint8_t* raw, uint8_t* dst, uint64_t n, uint32_t col0, uint32_t lb, uint32_t lw, uint32_t* ragged);   // a piece of a regular FASTA record without its line ends
void launch_frag_has_n(hipStream_t s, const uint64_t* goff, const uint32_t* len, uint32_t nf, DevGenomeIdx gx, uint8_t* has_n);
// one chunk of a FASTA file parsed on the device (k_fa_*): st = {bases so far, headers so far, kind of the open line}; kind n bytes,
// keep / pos n + 1 words; hdr = pairs {file offset of a header, bases before it}, at most hdr_cap of them
size_t fasta_chunk_temp_bytes(uint32_t n);
void launch_fasta_chunk(hipStream_t s, const uint8_t* raw, uint32_t n, unsigned long long chunk_off, unsigned long long* st, uint8_t* kind, uint32_t* keep, uint32_t* pos,
                        uint8_t* out, unsigned long long* hdr, uint32_t hdr_cap, void* temp, size_t temp_bytes);
// simuvars: out[piece.dst ..] = upper(ref | literal pool), then the SNP / SNV alleles
void launch_sv_build(hipStream_t s, const uint8_t* ref, const uint8_t* lit, const SvPiece* pieces, uint32_t np, const SvSubst* subs, uint32_t nsub, uint8_t* out, uint64_t total);
void launch_genome_bits(hipStream_t s, const uint8_t* g, uint64_t n, uint64_t nwords, unsigned long long* gc_bits, unsigned long long* n_bits,
                        uint32_t* gc_cnt, uint32_t* n_cnt, uint64_t* gc_pref, uint64_t* n_pref, void* temp, size_t temp_bytes, uint32_t* g2, ulonglong2* gc_pair);   // g2: the genome at two bits per base ((nwords + 1) * 4 words)
void launch_amplify_init(hipStream_t s, int64_t* primer_cnt, unsigned long long* primer_cut, int64_t copies, uint32_t* primer_delta, uint32_t* primer_gdelta, uint32_t* flags, unsigned long long* sums,
                         unsigned long long nf_all, unsigned long long frag_len_all, unsigned long long total_primers, uint32_t* pool_head_a, uint32_t* pool_head_b);   // primer_gdelta: a sharded job's exchange buffer (else null)
void launch_primer_update(hipStream_t s, int64_t* primer_cnt, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* dsums, uint32_t* flags);
void launch_weights(hipStream_t s, DevAmps fulls, uint32_t n, DevTables tb, RngKey key, uint32_t frag_size, double* w);
void launch_plan_pairs(hipStream_t s, DevFrags fr, DevAmps semis, DevAmps fulls, uint32_t first, uint32_t n_fulls, uint32_t pair_lo, uint32_t pair_hi, const uint32_t* read_numbers,
                       const uint32_t* pair_off, SegMap gmap, DevTables tb, RngKey key, int paired, PairRec* pairs, unsigned long long* holes);   // amplicons [first, first + n_fulls); writes (and counts the holes of) the pairs [pair_lo, pair_hi) only
void launch_batch_bounds(hipStream_t s, const uint32_t* pair_off, uint32_t ac, unsigned long long batch, uint32_t nb, uint32_t* bounds);
void launch_parity_pair_offsets(hipStream_t s, uint32_t* rn, uint32_t ac, uint32_t* pair_cnt_off, void* temp, size_t temp_bytes);   // PE, one shard: parity fix + pair offsets in one scan
void launch_pair_offsets(hipStream_t s, const uint32_t* rn, uint32_t ac, int paired, uint32_t* pair_cnt_off, void* temp, size_t temp_bytes);
// reads of pairs [p0, p0+np): slot layout [2*np][slot] bases / quals (SE: [np][slot])
// the two side streams the small class kernels of a batch run on beside the big one, with their fork / join events: owned by
// the ctx whose batches they order (created by the first launch_reads that uses them, on the current device)
struct ReadsSide { hipStream_t st[2] = {nullptr, nullptr}; hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr}; void release(); };
size_t reads_lds_bytes(const DevTables& tb, bool uni = false);   // uni: the uniform-walk variant (event-free ACGT-only reads)
//          // dynamic LDS of one inject_errors workgroup for this profile
void launch_reads(hipStream_t s, const uint8_t* g, const uint32_t* g2, DevErrPool spool, DevErrPool fpool,
                  const PairRec* pairs, uint32_t np, uint32_t amp_index_base, DevTables tb, const DevTables* d_tb, RngKey key, int paired, uint32_t slot,
                  const uint32_t* ev_hdr, const uint4* ev_dat, const uint64_t* off1, const uint64_t* off2, char* out1, char* out2, uint32_t* flags,
                  uint64_t cap1, uint64_t cap2,   // writes FASTQ text; cap: bytes of the batch's text in each file (records are checked against it)
                  const uint32_t* slist1, const uint32_t* slist2, const uint32_t* clist1, const uint32_t* clist2, uint32_t nc1, uint32_t nc2,
                  const uint32_t* dlist1, const uint32_t* dlist2, uint32_t nd1, uint32_t nd2, struct ReadsSide* side);   // launch_read_lists' lists; nc: reads with indel events per mate; side: below (null: one stream)
// splits the batch's reads into those without indel events and the rest (cls from launch_indels): ascending pair-index lists per mate
void launch_read_lists(hipStream_t s, uint32_t np, int paired, const uint32_t* sizes1, const uint64_t* off1, const uint32_t* d1f1, uint32_t* d1p1,
                       const uint32_t* sizes2, const uint64_t* off2, const uint32_t* d1f2, uint32_t* d1p2,
                       uint32_t* slist1, uint32_t* slist2, uint32_t* clist1, uint32_t* clist2, uint32_t* dlist1, uint32_t* dlist2, void* temp, size_t temp_bytes);
// the indel pass of a batch (n' and events per read, FASTQ record sizes per pair and mate), ahead of launch_reads
void phase_clock_report();   // -DSCS_PHASE_CLOCK builds: prints and zeroes the uniform walk's phase times (no-op otherwise)
void launch_indels(hipStream_t s, const PairRec* pairs, uint32_t np, int paired, DevTables tb, RngKey key, uint32_t slot, uint32_t* ev_hdr, uint4* ev_dat,
                   uint32_t* sizes1, uint32_t* sizes2, uint32_t* d1f1, uint32_t* d1f2, uint32_t* flags);
void launch_predict_windows(hipStream_t s, const uint8_t* windows, uint32_t n_reads, const uint64_t* uids, const uint32_t* atts,
                            const uint8_t* is_read1, DevTables tb, const DevTables* d_tb, RngKey key, uint32_t slot, char* slot_b, char* slot_q,
                            uint32_t* lens, uint32_t* flags);
void launch_text_checksum(hipStream_t s, const char* text, uint64_t nbytes, unsigned long long* out);   // *out = position-mixed 64-bit sum of the text (16-byte aligned) in HBM
void launch_philox(hipStream_t s, const uint32_t* ctr, uint32_t n, RngKey key, uint32_t* out);
void launch_detlog(hipStream_t s, const double* x, uint32_t n, double* out);

// gathers up to 12 device scalars (4 or 8 bytes wide) into mail[dsts[i]] (u64 each)
#define MAIL_SEQ_SLOT 31                                   // mailbox word the sequence number of a post lands in
void launch_mail(hipStream_t s, const void* const* srcs, const int* widths, const int* dsts, int n, unsigned clear, unsigned long long* mail, unsigned long long seq);   // n <= 16
void launch_shard_tail(hipStream_t s, uint32_t* primer_gdelta, const unsigned long long* dsums, const uint32_t* new_semis, int with_budgets);
void launch_primer_update_sharded(hipStream_t s, int64_t* primer_cnt, uint32_t* primer_gdelta, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* dsums, uint32_t* flags, int with_budgets);

// first error of any kernel launch / attribute call since the last call (hipSuccess if none)
hipError_t take_launch_error();

// device-wide exclusive scans (n inputs -> n+1 outputs, last = total)
size_t scan_temp_bytes(size_t n);
size_t parity_scan_temp_bytes(size_t n);
void exclusive_scan_u32(hipStream_t s, const uint32_t* in, uint32_t* out, size_t n, void* temp, size_t temp_bytes);
void exclusive_scan_u32_pair(hipStream_t s, const uint32_t* in0, uint32_t* out0, size_t n0, const uint32_t* in1, uint32_t* out1, size_t n1, void* temp, size_t temp_bytes);
// record sizes (class flag in bit 31) -> byte offsets in the low OFF_BITS bits, count of flagged reads above (one scan for both)
#define OFF_BITS 40
#define OFF_MASK ((1ull << OFF_BITS) - 1ull)
void exclusive_scan_sizes(hipStream_t s, const uint32_t* in, uint64_t* out, size_t n, void* temp, size_t temp_bytes);
void exclusive_scan_u32_to_u64(hipStream_t s, const uint32_t* in, uint64_t* out, size_t n, void* temp, size_t temp_bytes);

}  // namespace scs
