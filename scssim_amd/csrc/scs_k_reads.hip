// scs_k_reads.hip -- gfx950 (CDNA4, wave64) kernels of Malbac::yieldReads: pair planning, the indel pass, the base pass that writes the
// FASTQ text (k_reads: Profile::predict + record formatting), the batch checksum.  Integer / byte work bounded by HBM and VALU issue.
#include <utility>
#include <type_traits>
#include "scs_device.h"
#include "scs_seams.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "scs_kernels_common.h"

namespace scs {
#ifdef SCS_PHASE_CLOCK
__device__ unsigned long long g_phase[16];
#endif
// ------------------------------------------------------------------------------------------------
// K4a  plan pairs: one thread per full amplicon runs the attempt loop of Amplicon::yieldReads
//      (Amplicon.cpp:448-491): insert size, rejection, position.
// ------------------------------------------------------------------------------------------------
// (amplicons [first, first + n_fulls): the reads stage plans a batch's pairs right before the batch's pre-pass.  An amplicon
// whose pairs straddle two batches runs its attempt loop in both, but every batch WRITES only the pair records of its own
// range [pair_lo, pair_hi) and counts only the holes among them: each record is written once -- the base pass of the batch
// before may still be reading its part of the amplicon's records on another stream -- and each hole is counted once)
__global__ void k_plan_pairs(DevFrags fr, DevAmps semis, DevAmps fulls, uint32_t first, uint32_t n_fulls, uint32_t pair_lo, uint32_t pair_hi, const uint32_t* __restrict__ read_numbers,
                             const uint32_t* __restrict__ pair_off, const SegMap gmap, DevTables tb, RngKey key, int paired,
                             PairRec* __restrict__ pairs, unsigned long long* __restrict__ holes) {
    // the insert-size thresholds (a few hundred) go to LDS: the lookup is a nine-step bisection per attempt, and from global
    // memory those dependent loads are what the kernel waits for
    __shared__ uint32_t s_isz[1024];
    __shared__ uint32_t s_act[256], s_wcnt[4];
    const uint32_t n_isz = (uint32_t)tb.n_isize;
    const bool isz_lds = n_isz <= 1024u;
    if (isz_lds) for (uint32_t k = threadIdx.x; k < n_isz; k += blockDim.x) s_isz[k] = tb.isize_t[k];
    // Three amplicons in four get no read at 30x: the block's amplicons WITH pairs in this batch are handed to its first threads,
    // densely (a thread per amplicon left 14 of 64 lanes working through the attempt loop's Philox blocks: profiles/r05_bench_sq.csv)
    {
        const uint32_t j0 = blockIdx.x * blockDim.x + threadIdx.x;
        bool act = false;
        if (j0 < n_fulls && read_numbers[first + j0] != 0u) {
            const uint32_t po0 = pair_off[first + j0], want0 = pair_off[first + j0 + 1] - po0;
            act = !(po0 >= pair_hi || po0 + want0 <= pair_lo);                       // some of its pairs lie in this batch
        }
        const unsigned long long m = __ballot(act);
        const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
        if (lane == 0) s_wcnt[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t k = 0; k < w; ++k) base += s_wcnt[k];
        if (act) s_act[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j0;
        __syncthreads();
    }
    const uint32_t n_act = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (threadIdx.x >= n_act) return;
    const uint32_t j = s_act[threadIdx.x];
    const uint32_t i = first + j;
    int n = (int)read_numbers[i];
    const uint32_t po = pair_off[i], want = pair_off[i + 1] - po;
    PairRec* dst = pairs + po;
    const uint32_t q_lo = pair_lo > po ? pair_lo - po : 0u, q_hi = pair_hi - po < want ? pair_hi - po : want;   // its pairs [q_lo, q_hi) are this batch's
    const uint32_t fsl = fulls.sl[i], amp_len = sl_len(fsl), s2 = sl_spos(fsl);
    const uint32_t L = (uint32_t)tb.L;
    // resolve U = full amplicon sequence to an index map once (Amplicon::getSequence, Amplicon.cpp:266-340, without the copies)
    const uint32_t sm = fulls.parent[i], ssl = semis.sl[sm], l1 = sl_len(ssl), f = semis.parent[sm];
    const View uv = shift_view(semi_tmpl_view(frag_view(fr.goff[f], fr.len[f], fr.strand[f]), sl_spos(ssl), l1), s2);
    PairRec r; r.amp = i;                                                          // index in the whole job's list (record names)
    for (uint32_t k = 0; k < gmap.n; ++k) if (i - gmap.lo[k] < gmap.cnt[k]) { r.amp = (uint32_t)(gmap.go[k] + (i - gmap.lo[k])); break; }
    r.base = uv.base; r.k1 = (int32_t)(l1 - 1 - s2);
    r.flags = (uv.comp & 1u) | (uv.dir < 0 ? 2u : 0u) | (fr.has_n[f] ? 4u : 0u);   // bit 2: the fragment holds a non-ACGT base
    r.e1 = semis.errs[sm]; r.e2 = fulls.errs[i]; r.uid = fulls.uid[i];
    uint32_t made = 0;
    if (amp_len >= L) {
        uint32_t att = 0, fails = 0;
        while (n > 0 && made < q_hi) {                                               // (what lies beyond q_hi is the next batch's)
            const U4 d = draw4(key, ST_PAIR, 0, r.uid, att);
            if (!paired) {
                r.att = att; r.pos = scale_draw(d.w[1], 0, amp_len - L + 1); r.isz = L;
                if (made >= q_lo && made < q_hi) dst[made] = r;
                ++made; ++att; --n; continue;
            }
            const uint32_t isz = (uint32_t)tb.isize_min + rand_indx_thr(isz_lds ? (const uint32_t*)s_isz : tb.isize_t, tb.isize_d, n_isz, d.w[0]);
            if (isz < L || isz > amp_len) { ++att; if (++fails > 1000) break; continue; }
            r.att = att; r.pos = scale_draw(d.w[1], 0, amp_len - isz + 1); r.isz = isz;
            if (made >= q_lo && made < q_hi) dst[made] = r;
            ++made; ++att; n -= 2;
        }
    }
    r.att = 0; r.pos = 0; r.isz = 0;
    const uint32_t h_lo = made > q_lo ? made : q_lo;
    for (uint32_t q = h_lo; q < q_hi; ++q) dst[q] = r;                             // holes
    if (h_lo < q_hi) atomicAdd(holes, (unsigned long long)(q_hi - h_lo));            // rare: the host reports pairs produced = planned - holes
}

// ------------------------------------------------------------------------------------------------
// K5  inject_errors = Profile::predict (lib/profile/Profile.cpp:1582-1697) + window extraction and record formatting
//     (Amplicon::yieldReads, Amplicon.cpp:459-541).  One THREAD per read, one workgroup per 256 reads of the same mate:
//       * staging: the read windows are gathered through the pair records' index maps, a dword (4 bases) per lane,
//         one load instruction per read, into an LDS tile with two bases per byte;
//       * phase 1, the indel tests of every input base (event list, n'), ran in k_indels over the whole batch first
//         (pair mode; explicit-window mode does it here): n' fixes the FASTQ record sizes, hence the record offsets;
//       * phase 2, the base pass, is workgroup-synchronous over the TABLE BINS: position j of a read uses bin
//         j*bins/n', so all 256 reads look up the same bin at the same time and a small ring of bins in LDS
//         (64 k-mer substitution rows + the 4 diagonal quality rows per bin), refilled a group of bins ahead
//         through registers, serves every lookup.  A wave whose reads all sit on clean k-mers takes a branch-free
//         fast step; first-two-bases / N k-mers take the general step;
//       * output (pair mode): the FASTQ text itself, realigned in registers to the record's byte offset and stored as
//         whole 32-byte aligned sectors (BlockOut); the workgroup's reads are handed to its lanes ordered by the sector
//         phase of their records, so that the lanes of a wave cross sector boundaries together.  Explicit-window mode
//         writes sequence/quality slots.
//     Three instantiations per batch in pair mode (CLS; lists from k_indels + k_read_lists), the workgroups of ONE launch
//     (k_reads_all):
//       1  reads without indel events in fragments without a non-ACGT base: the UNIFORM WALK -- position t at bin t, one
//          step of stream B per position, windows from the two-bit genome, straight-line code unrolled by 16 positions
//          with a one-position software pipeline; the base call is ONE compare of the draw against the interval that keeps
//          the window's base (RingBinU), and a position whose draw does not keep it is set aside in LDS and resolved after
//          the pass, base and quality patched into the text (redo_read when a read runs out of room);
//       3  the same walk for reads whose only event is the deletion of one base (n' = L - 1: bins j L / (L - 1) = j) or the insertion
//          of one (n' = L + 1: bins j L / (L + 1) = 0, 0, 1, 2, ...: position 0 is made ahead of the walk, step t makes position t + 1);
//       2  everything else: the general loop described above.  (0: explicit-window mode, the general loop.)
//     The prologue is a chain of dependent loads (list entry -> pair record + offset -> window gather) at four workgroups
//     per CU: the ring's first groups and the event words are requested early, the records are parked in LDS for the
//     hand-out by sector phase, the gather issues all its loads before it uses one, and every barrier orders LDS only
//     (lds_barrier).  -DSCS_PHASE_CLOCK times the phases of a workgroup's life.
//     LDS per workgroup at L = 150: 14 KB ring + 19 KB rows (uniform walks) / 16 KB ring + 4 KB events + 19 KB windows
//     (general) -> 4 workgroups per CU.
// ------------------------------------------------------------------------------------------------
#ifndef SCS_RB
#define SCS_RB 256
#endif
#define RB SCS_RB
#define EV_MAX 8
// ring geometry: two groups of bins (one being served, one being filled).  A bin image = the 4 diagonal quality rows as
// alias rows (QK columns: QK words + QK symbol bytes each, scs_tables.h) + the 64 k-mer substitution rows (3 thresholds; the
// uniform walk's image, RingBinU: 2 words -- the interval of draws that keep the base -- and 256 bytes less per bin).
//   QK = 16  (binned-quality models, e.g. HiSeq X):   80 B rows, 1088 B bins, groups of 8
//   QK = 64  (8-bit-quality models):                 320 B rows, 2048 B bins, groups of 4
//   QK = 128 (a row with more than 64 symbols):      640 B rows, 3328 B bins, groups of 2
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>), in order: an unrolled loop whose index is a compile-time constant
template <class F, int... I>
__device__ __forceinline__ void unroll_steps(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int QK> struct RingGeo;
template <> struct RingGeo<16>  { enum { SLOTS = 16, GROUP = 8, QROW = 5,  ABITS = 4 }; };     // QROW: uint4 per quality row
template <> struct RingGeo<64>  { enum { SLOTS = 8,  GROUP = 4, QROW = 20, ABITS = 6 }; };
template <> struct RingGeo<128> { enum { SLOTS = 4,  GROUP = 2, QROW = 40, ABITS = 7 }; };
template <int QK> struct RingBin { uint4 qd[4][RingGeo<QK>::QROW]; uint32_t subs[64][3]; };
// the uniform walk's bin: the 3-mers' KEEP intervals (lo, width) instead of their threshold triples (scs_stage.cpp ring_image_u)
template <int QK> struct RingBinU { uint4 qd[4][RingGeo<QK>::QROW]; uint32_t keep[64][2]; };
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// LDS-qualified pointer types: keep the compiler from merging LDS and global accesses into FLAT ones
typedef __attribute__((address_space(3))) u32x4_t LdsU4;
typedef __attribute__((address_space(3))) u32x2_t LdsU2;
typedef __attribute__((address_space(3))) uint8_t LdsU8;
typedef __attribute__((address_space(3))) uint16_t LdsU16;
typedef __attribute__((address_space(3))) uint32_t LdsU32;

// window row stride in bytes (two bases per byte, one spare byte, an odd number of dwords: conflict-free columns)
__host__ __device__ static inline uint32_t win_stride(uint32_t n) {
    uint32_t ws = ((n >> 1) + 1u + 3u) & ~3u;
    if (((ws >> 2) & 1u) == 0) ws += 4;
    return ws;
}
// indel events, 16 bits: pos:10 | del:1 | len:5.  A read with an event that does not fit (position >= 1024, length
// >= 32, more than EV_MAX events) is "replayed": phase 2 re-draws its indel tests from stream A as it goes.
__device__ __forceinline__ uint32_t ev_pack(uint32_t pos, uint32_t del, uint32_t len) { return pos | (del << 10) | (len << 11); }
__device__ __forceinline__ uint32_t ev_pos(uint32_t v) { return v & 1023u; }
__device__ __forceinline__ uint32_t ev_del(uint32_t v) { return (v >> 10) & 1u; }
__device__ __forceinline__ uint32_t ev_len(uint32_t v) { return v >> 11; }

__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u :
           v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ void win_put2(LdsU8* w, int i, uint32_t v) {           // two bits per base
    LdsU32* d = (LdsU32*)w + (i >> 4); const uint32_t sh = 2u * (uint32_t)(i & 15);
    *d = (*d & ~(3u << sh)) | (v << sh);
}
// row of the uniform walk: 36 bytes of pending-position slots (three entries of three words) + the window at two bits per base, an
// odd number of dwords in all
__host__ __device__ static inline uint32_t uni_row_bytes(uint32_t n) {
    uint32_t d = 9u + ((n + 15u) >> 4);
    if ((d & 1u) == 0) ++d;
    return 4u * d;
}
__device__ __forceinline__ uint32_t win_get(const LdsU8* w, int i) { return ((uint32_t)w[i >> 1] >> ((i & 1) * 4)) & 15u; }
__device__ __forceinline__ void win_put(LdsU8* w, int i, uint32_t v) {
    const uint32_t sh = (uint32_t)(i & 1) * 4u, old = w[i >> 1];
    w[i >> 1] = (uint8_t)((old & ~(15u << sh)) | (v << sh));
}

// A workgroup barrier that orders LDS accesses ONLY (the HIP __syncthreads() also waits for every outstanding global load and store
// of the wave -- s_waitcnt vmcnt(0) -- before it lets the wave arrive).  k_reads synchronises nothing but LDS between its waves, and
// keeps global loads (the next phase's inputs, the ring's prefetch) and the text's stores in flight across its barriers.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// [REMAP] quality symbol by the alias method (scs_tables.h): column = the draw's top bits; its low bits against the column's
// threshold pick the column's own symbol or its alias; every symbol is hit by exactly as many of the 2^32 draws as in
// the reference's CDF comparison.  One 4-byte and one 1-byte read, wherever the row lives.
template <int QK, class W, class S>
__device__ __forceinline__ uint32_t alias_pick(const W* __restrict__ row, const S* __restrict__ syms, uint32_t x) {
    constexpr uint32_t AB = RingGeo<QK>::ABITS;
    const uint32_t col = x >> (32u - AB), e = row[col];
    // e = t << AB | alias: (x's low bits) < t  <=>  (low bits << AB | QK-1) < e  (both sides compared with their low AB bits in place)
    const uint32_t pick = ((x << AB) | (uint32_t)(QK - 1)) < e ? col : (e & (uint32_t)(QK - 1));
    return syms[pick];
}
// base call + quality of one position entirely from the global tables (k-mer rows outside the LDS ring, substituted bases,
// the xs == 0xFFFFFFFF draw).  ki < 0: the base is not re-drawn, k comes in.
template <int QK>
__device__ __forceinline__ uint32_t call_global_body(const uint32_t* __restrict__ subs, const double* __restrict__ subs_d, const uint32_t* __restrict__ qalias,
                                                     uint32_t B, int ki, uint32_t k, uint32_t c2, uint32_t bin, uint32_t xs, uint32_t xq) {
    if (ki >= 0) {
        const uint32_t row = ((uint32_t)ki * B + bin) * 4u;
        if (xs == 0xFFFFFFFFu) k = rand_indx_slow(subs_d + row, 4, xs);
        else { const uint4 T = *reinterpret_cast<const uint4*>(subs + row); k = (xs >= T.x) + (xs >= T.y) + (xs >= T.z); }
    }
    const uint32_t* __restrict__ qrow = qalias + (size_t)((c2 * 4u + k) * B + bin) * (QK + QK / 4);
    const uint32_t qv = alias_pick<QK>(qrow, reinterpret_cast<const uint8_t*>(qrow + QK), xq);
    return k | (qv << 8);
}
template <int QK>
__device__ __noinline__ uint32_t call_global(const uint32_t* __restrict__ subs, const double* __restrict__ subs_d, const uint32_t* __restrict__ qalias,
                                             uint32_t B, int ki, uint32_t k, uint32_t c2, uint32_t bin, uint32_t xs, uint32_t xq) {
    return call_global_body<QK>(subs, subs_d, qalias, B, ki, k, c2, bin, xs, xq);
}
// The uniform walk of the event-free class (k_reads, CLS 1) sets a substituted base's quality aside in LDS; a read that runs
// out of room for that (or draws x == 0xFFFFFFFF, whose base call needs the double tables) is made AGAIN here, one lane at
// a time, from the genome and the global tables, and its bases and qualities are stored over what the walk wrote.  Rare
// (three substitutions within the first 24 bases, ...): correctness path, no care for speed.  gb / gf: the window's first
// base and its flags (bit0 complement, bit1 backwards) as in the staging; e1 / e2: the error words of the pair record.
template <int QK>
__device__ __noinline__ void redo_read(const uint8_t* __restrict__ g, int64_t gb, uint32_t gf, const uint32_t* __restrict__ spool, const uint32_t* __restrict__ fpool,
                                       uint64_t e1, uint64_t e2, int k1, uint32_t pos, uint32_t isz, uint32_t rd, int n, uint32_t B,
                                       const uint32_t* __restrict__ subs, const double* __restrict__ subs_d, const uint32_t* __restrict__ qalias,
                                       U4 seed, char* __restrict__ out_b, char* __restrict__ out_q, uint32_t del_pos, uint32_t ins_pos) {
    // (del_pos: the one deleted base of a read of the one-event class, 0xFFFF for none: n - 1 positions, bins j * n / (n - 1) = j;
    //  ins_pos: the base behind which its one inserted base follows: n + 1 positions, bins j * n / (n + 1) = 0, 0, 1, 2, ...)
    Xoshiro xb; xb.seed(seed);
    uint32_t c0 = 5u, c1 = 5u;
    const bool ins = ins_pos != 0xFFFFu;
    const int np = ins ? n + 1 : del_pos == 0xFFFFu ? n : n - 1;
    for (int tp = 0; tp < np; ++tp) {
        const int t = ins ? tp - ((uint32_t)tp > ins_pos ? 1 : 0) : tp + ((uint32_t)tp >= del_pos ? 1 : 0);   // window base of output position tp
        uint32_t c2;
        if (ins && (uint32_t)tp == ins_pos + 1u) { uint32_t xi, xu; xb.next2(xi, xu); c2 = scale_draw(xi, 0, 3); }   // the inserted base: a step of its own
        else {
            c2 = g[(gf & 2u) ? gb - t : gb + t];
            if ((gf & 1u) && c2 < 4u) c2 = 3u - c2;
            for_each_err(e1, spool, [&](uint32_t e) {
                const int tt = k1 - (int)err_pos(e); const int k = rd ? (int)(pos + isz - 1) - tt : tt - (int)pos;
                if (k == t) c2 = rd ? err_alt(e) : 3u - err_alt(e);
            });
            for_each_err(e2, fpool, [&](uint32_t e) {
                const int tt = (int)err_pos(e); const int k = rd ? (int)(pos + isz - 1) - tt : tt - (int)pos;
                if (k == t) c2 = rd ? 3u - err_alt(e) : err_alt(e);
            });
        }
        const int ki = kmer_index(c0, c1, c2);
        uint32_t xs, xq; xb.next2(xs, xq);
        uint32_t bc, qc;
        const uint32_t bin = ins ? (tp ? (uint32_t)tp - 1u : 0u) : (uint32_t)tp;
        if (ki < 0 && c2 > 3u) { bc = 'N'; qc = 33u + scale_draw(xq, 0, 20); }
        else { const uint32_t kq = call_global_body<QK>(subs, subs_d, qalias, B, ki, c2, c2, bin, xs, xq); bc = (0x54474341u >> (8u * (kq & 255u))) & 255u; qc = 33u + (kq >> 8); }
        out_b[tp] = (char)bc; out_q[tp] = (char)qc;
        c0 = c1; c1 = c2;
    }
}

// [REMAP] number of event-free bases before the next indel event among the `rem` bases left: the per-base tests of
// getIndelSeq (Profile.cpp:1552-1570) are i.i.d. with probability p = t_indel / 2^32, so the gap is geometric and ONE draw
// x gives it: gap >= g <=> x < T[g], T[g] = floor((1-p)^g 2^32) (non-increasing, host-built: scs_tables.h).  Returns rem when
// no event falls among the bases left (86 % of 150-base reads with the shipped models: one compare).
__device__ __forceinline__ uint32_t indel_gap(const uint32_t* __restrict__ T, uint32_t x, uint32_t rem) {
    if (x < T[rem]) return rem;
    uint32_t lo = 1, hi = rem;                                                     // first g in [1, rem] with x >= T[g] (g = rem qualifies)
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (x >= T[mid]) hi = mid; else lo = mid + 1; }
    return lo - 1;
}
// phase 1 of Profile::predict: the indel events of a read (getIndelSeq, Profile.cpp:1552-1570 / loop 1606-1630): stream A
// gives, event by event, the gap to the next event and its kind; the length is a keyed Philox draw.
// put(i, v) stores event i (16 bits).  Returns n' (0 = the read does not fit its slot), the event count and the replay flag.
struct IndelPass { int n_out; int nev; bool replay; };
template <class Put>
__device__ __forceinline__ IndelPass indel_pass(const DevTables& tb, RngKey key, uint32_t aux, uint64_t uid, uint32_t force_replay, uint32_t slot,
                                                uint32_t* __restrict__ flags, Put put) {
    const int n = tb.L; const uint32_t t_kind = tb.t_kind;
    int nev = 0, delta = 0; bool replay = false;
    Xoshiro xa; xa.seed(draw4(key, ST_READ, aux, uid, 0));                         // stream A: gap, kind, gap, kind, ...
    if (tb.t_indel) for (int ji = 0; ji < n;) {
        ji += (int)indel_gap(tb.gap_t, xa.next(), (uint32_t)(n - ji));
        if (ji >= n) break;
        const uint32_t y = xa.next();                                              // an event at base ji: insertion | deletion in the ratio of their rates
        const uint32_t x = draw4(key, ST_INDEL_LEN, aux, uid, (uint32_t)ji).w[0];
        if (y < t_kind) {
            const uint32_t k = rand_indx_thr(tb.ins_t, tb.ins_d, (uint32_t)tb.n_ins, x);
            if (k > 0) {
                if (nev < EV_MAX && ji < 1024 && k < 32u) put(nev, ev_pack((uint32_t)ji, 0u, k)); else replay = true;
                ++nev; delta += (int)k;
            }
            ++ji;
        } else {
            const uint32_t k = rand_indx_thr(tb.del_t, tb.del_d, (uint32_t)tb.n_del, x);
            if (k > 0) {
                const int kk = (int)k < n - ji ? (int)k : n - ji;
                if (nev < EV_MAX && ji < 1024 && kk < 32) put(nev, ev_pack((uint32_t)ji, 1u, (uint32_t)kk)); else replay = true;
                ++nev; delta -= kk; ji += kk;
            }
            else ++ji;
        }
    }
    if ((force_replay & 1u) && nev > 0) replay = true;
    if (n + delta < 50) { nev = 0; delta = 0; replay = false; }                    // Profile.cpp:1623-1630: drop all indels
    int n_out = n + delta;
    if (n_out > (int)slot) { atomicOr(flags, (uint32_t)FLAG_READSLOT); n_out = 0; nev = 0; replay = false; }
    if (replay) nev = 0;                                                           // phase 2 draws the tests again
    return IndelPass{n_out, nev, replay};
}

// K5a  the indel pass of every read of a batch, ahead of the base pass: n' fixes the size of the FASTQ record, so the
//      record offsets (prefix sums) are known before k_reads runs.  Thread per read; events (8 x 16 bits) and
//      header {n' | events << 16 | replay << 24 | live << 25} go to global memory.
__global__ void __launch_bounds__(256) k_indels(const PairRec* __restrict__ pairs, uint32_t np, int paired, const DevTables tb, RngKey key, uint32_t slot,
                                                uint32_t force_replay, uint32_t* __restrict__ ev_hdr, uint4* __restrict__ ev_dat,
                                                uint32_t* __restrict__ sizes1, uint32_t* __restrict__ sizes2, uint32_t* __restrict__ d1f1, uint32_t* __restrict__ d1f2,
                                                uint32_t* __restrict__ flags) {
    // 86 % of the reads have no indel event and are done after ONE draw of stream A; the others walk an event loop with a Philox block per
    // event.  A thread per read left 22 of 64 lanes working there (profiles/r05_bench_sq.csv): every thread seeds stream A and looks at its
    // read's first gap; the reads WITH an event are then handed to the block's first threads, densely, and run the whole pass there.
    __shared__ uint32_t s_act[256], s_wcnt[4];
    const uint32_t r0 = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nreads = paired ? 2 * np : np;
    // what a read leaves behind: header, event words, record size (+ class bit), one-event flag
    auto emit = [&](uint32_t r, const IndelPass& ip, unsigned long long e_lo, unsigned long long e_hi, uint32_t att, uint32_t amp, uint32_t has_n) {
        const uint32_t pi = paired ? r >> 1 : r, rd = paired ? (r & 1u) : 0u;
        uint32_t* sz = rd ? sizes2 : sizes1;
        uint32_t* d1f = rd ? d1f2 : d1f1;                                          // 1: the read's only event is the deletion or the insertion of one base (k_reads' one-event walk)
        ev_hdr[r] = (uint32_t)ip.n_out | ((uint32_t)ip.nev << 16) | (ip.replay ? 1u << 24 : 0u) | (1u << 25);
        if (ip.nev > 0) ev_dat[r] = make_uint4((uint32_t)e_lo, (uint32_t)(e_lo >> 32), (uint32_t)e_hi, (uint32_t)(e_hi >> 32));   // (86 % of the reads have no event: nothing reads their slots)
        const uint32_t e0 = (uint32_t)e_lo & 0xFFFFu;
        // the one-event class of k_reads: the deletion of one base (n' = L - 1) -- or the insertion of one (n' = L + 1; not when L - 1 is a
        // multiple of 16: reads_body, NP)
        const bool one = ip.nev == 1 && !ip.replay && !has_n && !(force_replay & 12u) && ev_len(e0) == 1u && tb.bins == tb.L;
        const bool d1 = one && (ev_del(e0) ? ip.n_out == tb.L - 1 : (ip.n_out == tb.L + 1 && !(force_replay & 16u) && ((tb.L - 1) & 15) != 0));
        d1f[pi] = d1 ? 1u : 0u;
        const uint32_t cls = ((ip.nev > 0 || ip.replay || has_n || (force_replay & 4u)) && !d1) ? 1u : 0u;   // the uniform walk takes ACGT-only windows without events
        // "@<ampIdx>#<fragCount>[/1|/2]\n" + seq + "\n+\n" + qual + "\n"   (Amplicon.cpp:459-466,497-504)
        sz[pi] = (ip.n_out == 0 ? 0u : 1u + dec_digits(amp) + 1u + dec_digits(att + 1) + (paired ? 2u : 0u) + 1u + 2u * (uint32_t)ip.n_out + 4u) | (cls << 31);   // bit 31: the class rides along into the offsets' scan
    };
    bool with_event = false;
    if (r0 < nreads) {
        const uint32_t pi = paired ? r0 >> 1 : r0, rd = paired ? (r0 & 1u) : 0u;
        const uint64_t uid = pairs[pi].uid; const uint32_t att = pairs[pi].att, isz = pairs[pi].isz, amp = pairs[pi].amp, has_n = pairs[pi].flags & 4u;
        if (isz == 0) { ev_hdr[r0] = 0; (rd ? sizes2 : sizes1)[pi] = 0; (rd ? d1f2 : d1f1)[pi] = 0; }   // hole: the insert-size loop gave up (Amplicon.cpp:484-489)
        else {
            if (tb.t_indel) { Xoshiro xa; xa.seed(draw4(key, ST_READ, rd | (att << 1), uid, 0)); with_event = xa.next() >= tb.gap_t[tb.L]; }   // indel_pass' first gap: an event among the L bases?
            if (!with_event) {
                IndelPass ip{tb.L, 0, false};
                if (ip.n_out > (int)slot) { atomicOr(flags, (uint32_t)FLAG_READSLOT); ip.n_out = 0; }
                emit(r0, ip, 0ull, 0ull, att, amp, has_n);
            }
        }
    }
    {
        const unsigned long long m = __ballot(with_event);
        const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
        if (lane == 0) s_wcnt[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t k = 0; k < w; ++k) base += s_wcnt[k];
        if (with_event) s_act[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r0;
        __syncthreads();
    }
    if (threadIdx.x >= s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]) return;
    const uint32_t r = s_act[threadIdx.x];
    const uint32_t pi = paired ? r >> 1 : r, rd = paired ? (r & 1u) : 0u;
    const uint64_t uid = pairs[pi].uid; const uint32_t att = pairs[pi].att, amp = pairs[pi].amp, has_n = pairs[pi].flags & 4u;
    unsigned long long e_lo = 0, e_hi = 0;
    const IndelPass ip = indel_pass(tb, key, rd | (att << 1), uid, force_replay, slot, flags, [&](int i, uint32_t v) {
        if (i < 4) e_lo |= (unsigned long long)v << (16 * i); else e_hi |= (unsigned long long)v << (16 * (i - 4));
    });
    emit(r, ip, e_lo, e_hi, att, amp, has_n);
}

// ---- FASTQ text straight from the base pass (pair mode).  A record is two byte streams per read: the name line + bases +
// "\n+\n", and the qualities + "\n".  A lane produces its characters four to a register word; a stream starts at an
// arbitrary byte address T, so words are realigned in registers (v_alignbyte against the previous word) into the ALIGNED
// dwords from Ta = T & ~3.  The bytes of a stream's first aligned dword that lie before T are the end of what precedes
// it in the record, and are known: the tail of the name line for the bases, the tail of "\n+\n" for the qualities -- so
// every dword except the record's very last one is written whole, exactly once.
// A lane collects 16 characters (four raw words) of each stream, all lanes at the same steps.  At the block's end the four
// words are byte-aligned (above) and then DWORD-aligned to the sector grid by a two-stage funnel over the previous and the
// new aligned words (dq = dword of Ta inside its 16 bytes): that gives one 16-byte aligned half sector.  A lower half
// waits in registers for its upper half, and the two leave as ONE WHOLE 32-BYTE ALIGNED SECTOR (two dwordx4 stores back
// to back): a lane's partial lines do not survive in L2 until its next store 16 positions later (the open lines of all
// lanes exceed the L2), so anything smaller than a sector is written to memory as a masked sector each time.  Only a
// stream's first sector (masked dwords) and its end (single dwords, once per wave after the pass) are not whole.
struct BlockOut {
    uint32_t R[4];                                                                 // raw words of the block being filled (characters 16m .. 16m+15)
    uint32_t P[3];                                                                 // byte-aligned dwords 1..3 of the previous block: its last dq are not placed yet
    uint32_t H[4];                                                                 // a finished lower half sector waiting for its upper half
    uint32_t carry;                                                                // the raw word before R[0]
    static __device__ __forceinline__ uint32_t al(uint32_t x, uint32_t prev, uint32_t s) {   // stream bytes 4k-s .. 4k-s+3
        return s ? __builtin_amdgcn_alignbyte(x, prev, 4u - s) : x;
    }
    // a = T & 31 of the stream (s = a & 3, dq = (a >> 2) & 3, first half = (a >> 4) & 1); sec0 = offset of its first sector
    __device__ __forceinline__ void block(char* __restrict__ base, uint32_t sec0, uint32_t a, uint32_t m) {   // block m is complete
        asm volatile("" : "+v"(a));                                                // the lane masks derived from a are made here, per block: kept in SGPRs through the whole pass they spill
        const uint32_t s = a & 3u, dq = (a >> 2) & 3u, slot = ((a >> 4) & 1u) + m;  // slot: 16-byte slots from sec0
        const uint32_t w0 = al(R[0], carry, s), w1 = al(R[1], R[0], s), w2 = al(R[2], R[1], s), w3 = al(R[3], R[2], s);
        carry = R[3];
        const bool t2 = dq & 2u, t1 = dq & 1u;                                     // A[j] = C[4 + j - dq], C = P[0..2] (1..3), w0..w3 (4..7)
        const uint32_t e3 = t2 ? P[0] : P[2], e4 = t2 ? P[1] : w0, e5 = t2 ? P[2] : w1, e6 = t2 ? w0 : w2, e7 = t2 ? w1 : w3;
        const uint32_t a0 = t1 ? e3 : e4, a1 = t1 ? e4 : e5, a2 = t1 ? e5 : e6, a3 = t1 ? e6 : e7;
        P[0] = w1; P[1] = w2; P[2] = w3;
        if (!(slot & 1u)) { H[0] = a0; H[1] = a1; H[2] = a2; H[3] = a3; }
        else {
            char* __restrict__ d = base + (sec0 + 16u * (slot - 1u));
            uint32_t lo = a >> 2;                                                  // the stream's first dword inside its first sector
            if (slot == 1u && lo) {                                                // first sector: only the dwords from Ta on are mine
                uint32_t* q = reinterpret_cast<uint32_t*>(d);
                asm volatile("" : "+v"(lo));                                       // compared here, once: not as six lane masks kept in SGPRs through the whole pass
                // dwords lo .. 7 (0..3 = H, 4..7 = a) in at most three stores per lane -- each store instruction of a wave touches 64
                // different lines, and these partial ones were a tenth of the kernel's time as seven single-dword stores:
                // 16 bytes (dwords 4..7) when lo <= 4; 8 bytes (2, 3 or 6, 7) when lo = 1, 2, 5, 6; 4 bytes at lo when lo is odd
                if (lo <= 4u) *reinterpret_cast<uint4*>(q + 4) = make_uint4(a0, a1, a2, a3);
                const bool low = lo < 4u;
                if ((lo & 3u) == 1u || (lo & 3u) == 2u) *reinterpret_cast<uint2*>(q + (low ? 2 : 6)) = make_uint2(low ? H[2] : a2, low ? H[3] : a3);
                if (lo & 1u) q[lo] = low ? (lo == 1u ? H[1] : H[3]) : (lo == 5u ? a1 : a3);
            } else {
                reinterpret_cast<uint4*>(d)[0] = make_uint4(H[0], H[1], H[2], H[3]);
                reinterpret_cast<uint4*>(d)[1] = make_uint4(a0, a1, a2, a3);
            }
        }
    }
    // block m >= 2 (never the stream's first sector) as STRAIGHT-LINE code: every call issues the sector's two stores, from a lane
    // whose half sector only waits (or that has no read: `on` false) to the 32 spare bytes at `spare`.  No divergent region
    // between two positions of the uniform walk: the scheduler keeps overlapping the positions around the call (measured:
    // -7 % on the kernel against the branching form, although half of these stores go nowhere).
    __device__ __forceinline__ void block_flat(char* __restrict__ base, uint32_t sec0, uint32_t a, uint32_t m, char* __restrict__ spare, bool on) {
        asm volatile("" : "+v"(a));
        const uint32_t s = a & 3u, dq = (a >> 2) & 3u, slot = ((a >> 4) & 1u) + m;
        const uint32_t w0 = al(R[0], carry, s), w1 = al(R[1], R[0], s), w2 = al(R[2], R[1], s), w3 = al(R[3], R[2], s);
        carry = R[3];
        const bool t2 = dq & 2u, t1 = dq & 1u;
        const uint32_t e3 = t2 ? P[0] : P[2], e4 = t2 ? P[1] : w0, e5 = t2 ? P[2] : w1, e6 = t2 ? w0 : w2, e7 = t2 ? w1 : w3;
        const uint32_t a0 = t1 ? e3 : e4, a1 = t1 ? e4 : e5, a2 = t1 ? e5 : e6, a3 = t1 ? e6 : e7;
        P[0] = w1; P[1] = w2; P[2] = w3;
        char* __restrict__ d = ((slot & 1u) && on) ? base + (sec0 + 16u * (slot - 1u)) : spare;
        reinterpret_cast<uint4*>(d)[0] = make_uint4(H[0], H[1], H[2], H[3]);
        reinterpret_cast<uint4*>(d)[1] = make_uint4(a0, a1, a2, a3);
        H[0] = a0; H[1] = a1; H[2] = a2; H[3] = a3;                                // (dead after a store, the waiting lower half otherwise)
    }
    // the stream's end, after the pass: block m holds nw (1..4) raw words, the last with nv (1..4) characters; then `sep`.
    // The stream has `nd` whole aligned dwords and `rem` (< 4) bytes after them (only a record's very end has rem != 0).
    __device__ __forceinline__ void tail(char* __restrict__ base, uint32_t sec0, uint32_t a, uint32_t m, uint32_t nw, uint32_t nv, uint32_t sep,
                                         uint32_t nd, uint32_t rem) {
        const uint32_t s = a & 3u, dq = (a >> 2) & 3u, slot = ((a >> 4) & 1u) + m;
        const int k0 = (int)(4u * m) - (int)dq;                                    // aligned dword of the stream that sits in slot `slot`, dword 0
        if ((slot & 1u) && m) {                                                    // the lower half still waiting: dword j is aligned dword k0 - 4 + j
            uint32_t* q = reinterpret_cast<uint32_t*>(base + (sec0 + 16u * (slot - 1u)));
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k0 - 4 < 0 && k0 - 4 + j >= 0) q[j] = H[j];
            if (k0 - 4 >= 0) *reinterpret_cast<uint4*>(q) = make_uint4(H[0], H[1], H[2], H[3]);   // (all four are the stream's: one store)
        }
        const unsigned long long sv = (unsigned long long)sep << (8u * (nv & 3u));
        const uint32_t last = nv < 4u ? (uint32_t)sv : 0u, after = nv < 4u ? (uint32_t)(sv >> 32) : sep;
        // (named scalars, not arrays: a select between two array elements is turned into an indexed load and the array into scratch)
        // the block's raw words with the separator behind them ...
        const uint32_t e0 = 0u + 1u < nw ? R[0] : 0u + 1u == nw ? (R[0] | last) : 0u == nw ? after : 0u;
        const uint32_t e1 = 1u + 1u < nw ? R[1] : 1u + 1u == nw ? (R[1] | last) : 1u == nw ? after : 0u;
        const uint32_t e2 = 2u + 1u < nw ? R[2] : 2u + 1u == nw ? (R[2] | last) : 2u == nw ? after : 0u;
        const uint32_t e3 = 3u + 1u < nw ? R[3] : 3u + 1u == nw ? (R[3] | last) : 3u == nw ? after : 0u;
        const uint32_t e4 = 4u + 1u < nw ? 0u : 4u + 1u == nw ? (0u | last) : 4u == nw ? after : 0u;
        const uint32_t e5 = 5u + 1u < nw ? 0u : 5u + 1u == nw ? (0u | last) : 5u == nw ? after : 0u;
        // ... byte-aligned: c1..c3 = P, c4.. = the tail's aligned dwords; dword t of the slot = c[4 + t - dq] = aligned dword k0 + t
        const uint32_t c1 = P[0], c2 = P[1], c3 = P[2], c4 = al(e0, carry, s), c5 = al(e1, e0, s), c6 = al(e2, e1, s), c7 = al(e3, e2, s),
                       c8 = al(e4, e3, s), c9 = al(e5, e4, s), c10 = 0u, c11 = 0u, c12 = 0u;
        const bool t2 = dq & 2u, t1 = dq & 1u;
        const uint32_t f3 = t2 ? c1 : c3, f4 = t2 ? c2 : c4, f5 = t2 ? c3 : c5, f6 = t2 ? c4 : c6, f7 = t2 ? c5 : c7, f8 = t2 ? c6 : c8, f9 = t2 ? c7 : c9, f10 = t2 ? c8 : c10, f11 = t2 ? c9 : c11, f12 = t2 ? c10 : c12;
        const uint32_t v0 = t1 ? f3 : f4, v1 = t1 ? f4 : f5, v2 = t1 ? f5 : f6, v3 = t1 ? f6 : f7, v4 = t1 ? f7 : f8, v5 = t1 ? f8 : f9, v6 = t1 ? f9 : f10, v7 = t1 ? f10 : f11, v8 = t1 ? f11 : f12;
        uint32_t* q = reinterpret_cast<uint32_t*>(base + (sec0 + 16u * slot));
        auto put = [&](int t, uint32_t v) {
            const int k = k0 + t;
            if (k >= 0 && k < (int)nd) q[t] = v;
            else if (k == (int)nd) for (uint32_t b = 0; b < rem; ++b) reinterpret_cast<char*>(q + t)[b] = (char)(v >> (8u * b));
        };
        put(0, v0); put(1, v1); put(2, v2); put(3, v3); put(4, v4); put(5, v5); put(6, v6); put(7, v7); put(8, v8);
    }
};

// CLS: 0 = the workgroup takes 256 consecutive pairs; 1 / 2 = it takes 256 consecutive entries of a LIST of pair indices:
// the reads without any indel event (CLS 1: 86 % of 150-base reads with the shipped models) and the rest (CLS 2), split
// by k_read_lists from k_indels' result.  The event-free reads need none of the event handling -- no lookahead for the
// next event, exactly one output position per table bin -- and their waves run with every lane busy at every bin.
template <bool FROM_PAIRS, int QK, int CLS>
__device__ __forceinline__ void reads_body(const uint32_t bid, const uint8_t* __restrict__ g, DevErrPool spool, DevErrPool fpool, const PairRec* __restrict__ pairs,
                                              uint32_t np, int paired, const uint8_t* __restrict__ windows, const uint64_t* __restrict__ uids,
                                              const uint32_t* __restrict__ atts, const uint8_t* __restrict__ is_read1, uint32_t n_explicit,
                                              const DevTables tb, RngKey key, uint32_t slot, uint32_t n_slots_cap, uint32_t force_replay,
                                              const uint32_t* __restrict__ ev_hdr, const uint4* __restrict__ ev_dat,
                                              const uint64_t* __restrict__ off1, const uint64_t* __restrict__ off2, char* __restrict__ out1, char* __restrict__ out2,
                                              uint32_t amp_index_base, char* __restrict__ slot_b, char* __restrict__ slot_q, uint32_t* __restrict__ lens,
                                              uint32_t* __restrict__ flags, uint64_t cap1, uint64_t cap2,
                                              const uint32_t* __restrict__ list1, const uint32_t* __restrict__ list2, uint32_t nlist1, uint32_t nlist2) {
    constexpr bool SIMPLE = CLS == 1 || CLS == 3;                                 // 3: the uniform walk for reads with ONE deletion of ONE base (below)
    constexpr bool D1 = CLS == 3;
    typedef RingGeo<QK> Geo;
    constexpr bool UNI = SIMPLE && FROM_PAIRS;
    typedef typename std::conditional<UNI, RingBinU<QK>, RingBin<QK>>::type Bin;
    constexpr int SLOTS = Geo::SLOTS, GROUP = Geo::GROUP, QROW = Geo::QROW;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    // the table descriptor is a by-value kernel argument: pointers loaded from the kernarg segment are known to be
    // global (a descriptor fetched through a pointer makes every table access a FLAT load)
    const int n = tb.L, B = tb.bins;
    const uint32_t WS = win_stride((uint32_t)n);
    Bin* s_ring = reinterpret_cast<Bin*>(s_dyn);                           // [SLOTS]
    int64_t* s_gbase = reinterpret_cast<int64_t*>(s_dyn);                  // [RB]  staging only: aliases the ring, which is filled later
    uint32_t* s_gflag = reinterpret_cast<uint32_t*>(s_gbase + RB);         // [RB]  bit0 complement, bit1 direction -1, bit2 valid
    uint16_t* s_ev = reinterpret_cast<uint16_t*>(s_dyn + SLOTS * sizeof(Bin));   // [RB][EV_MAX]; a replayed read keeps its stream-A state here
    // rows: [RB][WS] windows behind the event slots; the uniform walk (UNI) has no events and keeps 36 bytes per lane
    // IN FRONT of its window instead -- three set-aside entries (the name line is composed there first), later entries overlay
    // the consumed start of the window -- and the window itself with TWO bits per base (its reads see no N): uni_row_bytes
    constexpr uint32_t WOFF = UNI ? 36u : 0u;
    const uint32_t ROW = UNI ? uni_row_bytes((uint32_t)n) : WS;
    uint8_t* s_win = UNI ? reinterpret_cast<uint8_t*>(s_ev) : reinterpret_cast<uint8_t*>(s_ev + RB * EV_MAX);
    uint32_t* s_head = reinterpret_cast<uint32_t*>(s_dyn + SLOTS * sizeof(Bin) + (size_t)RB * ROW);   // [128] UNI: keep intervals of the 1- and 2-mers (head rows: scs_stage.cpp ring_image_u)
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
#ifdef SCS_PHASE_CLOCK
    unsigned long long ph_t_ = 0;
    const unsigned long long ph_c0_ = __builtin_amdgcn_s_memtime(), ph_r0_ = wall_clock64();   // shader clock beside the 100 MHz one: the clock the chip holds
#endif
    SCS_PHASE(-1);

    const bool second_file = FROM_PAIRS && paired && (bid & 1u);
    const uint64_t* __restrict__ offs = second_file ? off2 : off1;
    char* __restrict__ outp = second_file ? out2 : out1;
    const uint32_t* __restrict__ wlist = second_file ? list2 : list1;              // CLS != 0: this mate's list
    const uint32_t nwork = CLS == 0 ? np : (second_file ? nlist2 : nlist1), wq = (paired ? bid >> 1 : bid) * RB;
    if (FROM_PAIRS && wq >= nwork) return;                                          // the grid covers the longer of the two mates' lists
    // ---- the table ring's first two groups of bins and the head rows: their loads leave FIRST and ride through the whole prologue
    // in registers (nothing they need is computed here; the LDS they go to is used for staging until the windows are in place).
    // Ring maintenance: the bins of group gq = t/GROUP live in half (gq & 1) of the ring.  Group 0 is loaded up front; the
    // bins of the next group are prefetched into registers one group ahead and written to LDS at the group boundary,
    // into the half that group gq-2 used -- every wave left that group before the previous boundary's barrier, so one
    // barrier per group is enough.
    // the bins' images come ready-made from global memory (DevTables::ring1/2, ring1u/2u): a group of GROUP bins is one contiguous
    // run of GROUP * EPB 16-byte entries there and in the ring
    constexpr int GE = GROUP * (int)(sizeof(Bin) / 16), NPRE = (GE + RB - 1) / RB;
    const bool second_wg = FROM_PAIRS && paired && (bid & 1u) && tb.subs2 != nullptr;      // (slot mode: the ring holds read 1's rows)
    const uint4* __restrict__ ring_img = UNI ? ((second_wg && tb.ring2u) ? tb.ring2u : tb.ring1u) : ((second_wg && tb.ring2) ? tb.ring2 : tb.ring1);
    u32x4_t* ring16 = reinterpret_cast<u32x4_t*>(s_dyn);
    u32x4_t pre[NPRE], ring0[NPRE];
    auto prefetch = [&](int first) __attribute__((always_inline)) {               // bins [first, first+GROUP) -> registers
        // (unconditional loads: past the table's end the last group is fetched again, and an entry index past the group's is clamped)
        const u32x4_t* __restrict__ src = reinterpret_cast<const u32x4_t*>(ring_img) + (size_t)min(first, ((B + 7) & ~7) - GROUP) * (sizeof(Bin) / 16);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) pre[u] = src[min(tid + u * RB, GE - 1)];
        __builtin_amdgcn_sched_barrier(0);                                         // the loads leave HERE, a group ahead of their use (the scheduler would sink them to the commit and wait there)
    };
    auto commit = [&](int first) __attribute__((always_inline)) {                 // registers -> LDS slots of bins [first, first+GROUP)
        u32x4_t* dst = ring16 + (first & (SLOTS - 1)) * (int)(sizeof(Bin) / 16);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) { const int idx = tid + u * RB; if (idx < GE) dst[idx] = pre[u]; }
    };
#pragma unroll
    for (int u = 0; u < NPRE; ++u) ring0[u] = reinterpret_cast<const u32x4_t*>(ring_img)[min(tid + u * RB, GE - 1)];
    const uint32_t head_w = UNI ? reinterpret_cast<const uint32_t*>(ring_img + (size_t)((B + 7) & ~7) * (sizeof(Bin) / 16))[tid & 127] : 0u;
    prefetch(GROUP);

    // ---- which read is mine
    uint32_t r, pi = 0, rd; bool valid; PairRec pr{}; uint64_t uid = 0; uint32_t att = 0;
    uint32_t rec_rel = 0, rec_h = 0;                                               // pair mode: my record's offset from wg_out, length of its name line
    uint64_t off0 = 0, my_off = 0;                                                 // pair mode: the byte offsets of the chunk's first record and of mine in the batch's text
    uint32_t ev_h = 0; uint4 ev_e = make_uint4(0, 0, 0, 0);                        // pair mode: my read's event words (k_indels), loaded as soon as the read is known
    if (FROM_PAIRS) {
        const uint32_t q = paired ? bid >> 1 : bid;
        rd = paired ? (bid & 1u) : 0u;
        // The workgroup's 256 reads are handed to its lanes ORDERED BY THE SECTOR PHASE of their bases (byte address & 31):
        // a lane stores a sector whenever its stream crosses a 32-byte boundary, and lanes of one wave that do so at the
        // same positions share the store instructions.  (Which lane makes which read does not show in the output.)
        // Every thread fetches ONE record (list entry -> pair record + text offset: two dependent rounds of loads), parks it in LDS
        // and picks up the record its sorted place gives it from there: the second trip to global memory this used to be is gone.
        constexpr uint32_t SR = 19;                                                // parked record: 14 words PairRec, offset (2), pair index, name-line length; odd stride
        uint32_t* s_park = reinterpret_cast<uint32_t*>(s_dyn);                     // [RB][SR]  (everything in LDS is free until the windows are staged)
        uint32_t* s_cnt = s_park + RB * SR; uint32_t* s_perm = s_cnt + 64;
        const uint32_t out_lo = (uint32_t)reinterpret_cast<uintptr_t>(outp);
        uint32_t keyp = 32u;
        auto pair_of = [&](uint32_t e) -> uint32_t { return e < nwork ? (CLS == 0 ? e : wlist[e]) : 0xFFFFFFFFu; };
        {
            const uint32_t p = pair_of(q * RB + tid);
            uint32_t* st = s_park + (uint32_t)tid * SR;
            st[16] = p;
            if (p != 0xFFFFFFFFu) {
                const PairRec o = pairs[p]; const uint64_t of = offs[p] & OFF_MASK;
                const uint32_t amp = amp_index_base + o.amp, cnt = o.att + 1u;
                const uint32_t h = 1u + dec_digits(amp) + 1u + dec_digits(cnt) + (paired ? 2u : 0u) + 1u;   // "@<amp>#<cnt>[/1|/2]\n"
                keyp = (out_lo + (uint32_t)of + h) & 31u;                           // sector phase of the record's first base
                uint32_t w[14]; __builtin_memcpy(w, &o, 56);
#pragma unroll
                for (int i = 0; i < 14; ++i) st[i] = w[i];
                st[14] = (uint32_t)of; st[15] = (uint32_t)(of >> 32); st[17] = h;
            }
        }
        if (tid < 64) s_cnt[tid] = 0;
        lds_barrier();
        const uint32_t rank = atomicAdd(&s_cnt[keyp], 1u);
        lds_barrier();
        if (tid < 64) {                                                            // exclusive prefix of the 33 bucket counts
            const uint32_t c = s_cnt[tid]; uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (tid >= d) incl += o; }
            s_cnt[tid] = incl - c;
        }
        lds_barrier();
        s_perm[s_cnt[keyp] + rank] = (uint32_t)tid;
        lds_barrier();
        const uint32_t* sm = s_park + s_perm[tid] * SR;
        pi = sm[16]; valid = pi != 0xFFFFFFFFu;
        r = paired ? 2 * pi + rd : pi;
        off0 = ((uint64_t)s_park[15] << 32) | s_park[14];                          // lists ascend: the chunk's first record (always there) is its lowest
        if (valid) {
            ev_h = ev_hdr[r]; ev_e = ev_dat[r];
            uint32_t w[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) w[i] = sm[i];
            __builtin_memcpy(&pr, w, 56);
            my_off = ((uint64_t)sm[15] << 32) | sm[14]; rec_h = sm[17];
            uid = pr.uid; att = pr.att;
        }
        lds_barrier();                                                           // the parked records are read before the staging overwrites them
        SCS_PHASE(0);
    } else {
        r = bid * RB + tid; valid = r < n_explicit; rd = 0;
        if (valid) { uid = uids[r]; att = atts[r]; rd = is_read1[r] ? 0u : 1u; }
    }
    // the workgroup's records are contiguous: stores address them as a uniform base (aligned down to a sector) + a 32-bit offset
    const uint32_t adj = (uint32_t)(reinterpret_cast<uintptr_t>(outp) + off0) & 31u;
    char* __restrict__ wg_out = outp + off0 - adj;
    if (FROM_PAIRS && valid) rec_rel = (uint32_t)(my_off - off0) + adj;
    if (!FROM_PAIRS && valid && r >= n_slots_cap) { atomicOr(flags, (uint32_t)FLAG_INTERNAL); valid = false; }   // never write outside the slot buffers
    bool live = valid && (!FROM_PAIRS || pr.isz != 0);

    // ---- stage the windows (coalesced), then patch the amplification errors
    LdsU8* my_win = (LdsU8*)(s_win + (size_t)tid * ROW + WOFF);
    if (FROM_PAIRS) {
        int64_t gb = 0; uint32_t gf = 0;
        if (live) {
            const int64_t dir = (pr.flags & 2u) ? -1 : 1; const uint32_t comp = pr.flags & 1u;
            if (rd == 0) { gb = pr.base + dir * (int64_t)pr.pos; gf = comp | ((dir < 0) ? 2u : 0u) | 4u; }
            else { gb = pr.base + dir * (int64_t)(pr.pos + pr.isz - 1); gf = (comp ^ 1u) | ((dir < 0) ? 0u : 2u) | 4u; }   // read 2 = revcomp of the far end
        }
        s_gbase[tid] = gb; s_gflag[tid] = gf;
        lds_barrier();
        // a lane takes 4 consecutive window bases = one dword of the genome (byte-reversed when the view runs backwards),
        // complements them in place and packs them into two LDS bytes: one load instruction covers 256 bases of a read
        if constexpr (UNI) {
            // windows from the two-bit genome (its reads see no N): W2 dwords of 16 bases per read.  A read takes W2 + 1 lanes of an
            // instruction -- lane j loads word T + j (T - j when the view runs backwards; T = the word of the window's first
            // base) and borrows lane j + 1's word for the funnel shift to the window's bit offset -- so one load instruction
            // serves 64 / (W2 + 1) reads (5 at L = 150).  Backwards: the 16 bases are reversed in the dword; complement: ~.
            const uint32_t* __restrict__ g2 = reinterpret_cast<const uint32_t*>(windows);   // pair mode: the `windows` argument carries the two-bit genome
            const int W2 = (n + 15) >> 4, LPR = W2 + 1, RPI = LPR <= WAVE ? WAVE / LPR : 1;   // (L <= 1008: the host sends longer reads to the general variant)
            const int r5 = lane / LPR, j = lane - r5 * LPR;
            // (all the loads of a batch leave before the first is used: the gather is latency bound -- 13 rounds one after the other
            // were 15 of a workgroup's 80 microseconds)
            constexpr int NB = 16;
            for (int ib = 0; ib < WAVE; ib += NB * RPI) {
                uint32_t own[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int rq = ib + u * RPI + r5; const bool on = r5 < RPI && rq < WAVE;
                    const int rr = wib * WAVE + (on ? rq : 0);
                    const bool bwd = (s_gflag[rr] & 2u) != 0; const int64_t T = s_gbase[rr] >> 4;
                    own[u] = g2[bwd ? T - j : T + j];                          // (always inside the padded array)
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int rq = ib + u * RPI + r5; const bool on = r5 < RPI && rq < WAVE;
                    const int rr = wib * WAVE + (on ? rq : 0);
                    const uint32_t f = s_gflag[rr]; const uint32_t q = (uint32_t)s_gbase[rr] & 15u;
                    const bool bwd = (f & 2u) != 0;
                    const uint32_t nbr = (uint32_t)__shfl_down((int)own[u], 1);
                    uint32_t v = bwd ? __builtin_amdgcn_alignbit(own[u], nbr, 2u * (q + 1u)) : __builtin_amdgcn_alignbit(nbr, own[u], 2u * q);
                    if (bwd) {
                        if (q == 15u) v = own[u];
                        v = __builtin_bitreverse32(v);                          // reverses the bases AND the two bits of each: swap those back
                        v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
                    }
                    if (f & 1u) v = ~v;                                           // complement: 3 - c
                    if (on && (f & 4u) && j < W2) reinterpret_cast<uint32_t*>(s_win + (size_t)rr * ROW + WOFF)[j] = v;
                }
            }
        } else
        for (int kb = 0; kb < n; kb += 4 * WAVE) {                                  // one round for L <= 256
            const int k4 = kb + 4 * lane;
            constexpr int FLY = 32;                                                 // reads whose loads are in flight per lane: the gather is HBM-latency bound
            for (int q0 = 0; q0 < 64; q0 += FLY) {
                uint32_t cv[FLY];
#pragma unroll
                for (int u = 0; u < FLY; ++u) {
                    const int rr = wib * 64 + q0 + u; const uint32_t f = s_gflag[rr];
                    const int64_t b0 = s_gbase[rr];
                    // branch-free: one unconditional dword load per lane and read, so that all FLY loads are in flight together
                    // (a load under a divergent branch is waited for at the join).  The window's last, partial group loads
                    // the last whole dword of the window and shifts; lanes past the window load it too and drop it.
                    const int kk = k4 < n - 4 ? k4 : n - 4, drop = k4 - kk;            // n >= 4
                    uint32_t v;
                    __builtin_memcpy(&v, g + ((f & 2u) ? b0 - kk - 3 : b0 + kk), 4);
                    if (f & 2u) v = __builtin_bswap32(v);
                    v = drop < 4 ? v >> (8 * drop) : 0u;
                    cv[u] = v;
                }
#pragma unroll
                for (int u = 0; u < FLY; ++u) {
                    const int rr = wib * 64 + q0 + u; const uint32_t f = s_gflag[rr];
                    uint32_t v = cv[u];
                    if (f & 1u) v ^= 0x03030303u & ~(((v >> 2) & 0x01010101u) * 3u);   // complement: 3 - c for ACGT codes, N (4) stays
                    const uint32_t pk = (v & 0xFu) | ((v >> 4) & 0xF0u) | ((v >> 8) & 0xF00u) | ((v >> 12) & 0xF000u);
                    if ((f & 4u) && k4 < n) *reinterpret_cast<uint16_t*>(s_win + (size_t)rr * ROW + WOFF + (k4 >> 1)) = (uint16_t)pk;
                }
            }
        }
        lds_barrier();
        if (live) {
            // U[t] patched at t = k1 - pos(e) with comp(alt) (semi) and at t = pos(e) with alt (full); window index of t:
            // read 1: t - pos ; read 2: pos + isz - 1 - t, complemented
            for_each_err(pr.e1, spool.data, [&](uint32_t e) {
                const int t = pr.k1 - (int)err_pos(e); const int k = rd ? (int)(pr.pos + pr.isz - 1) - t : t - (int)pr.pos;
                if (k >= 0 && k < n) { if (UNI) win_put2(my_win, k, rd ? err_alt(e) : 3u - err_alt(e)); else win_put(my_win, k, rd ? err_alt(e) : 3u - err_alt(e)); }
            });
            for_each_err(pr.e2, fpool.data, [&](uint32_t e) {
                const int t = (int)err_pos(e); const int k = rd ? (int)(pr.pos + pr.isz - 1) - t : t - (int)pr.pos;
                if (k >= 0 && k < n) { if (UNI) win_put2(my_win, k, rd ? 3u - err_alt(e) : err_alt(e)); else win_put(my_win, k, rd ? 3u - err_alt(e) : err_alt(e)); }
            });
        }
    } else {
        const size_t base_off = (size_t)bid * RB * (size_t)n;
        const uint32_t nblk = min((uint32_t)RB, n_explicit - bid * RB);
        const uint32_t hb = ((uint32_t)n + 1u) >> 1;
        for (uint32_t idx = tid; idx < nblk * hb; idx += RB) {
            const uint32_t row = idx / hb, b = idx % hb;
            const uint32_t lo = windows[base_off + (size_t)row * n + 2 * b] & 15u;
            const uint32_t hi = 2 * b + 1 < (uint32_t)n ? windows[base_off + (size_t)row * n + 2 * b + 1] & 15u : 0u;
            s_win[(size_t)row * ROW + WOFF + b] = (uint8_t)(lo | (hi << 4));
        }
    }

    SCS_PHASE(1);
    // ---- phase 1: the indel events of my read: computed by k_indels ahead of this launch (pair mode), or here
    const uint32_t aux = rd | (att << 1);
    LdsU16* my_ev = (LdsU16*)(s_ev + tid * EV_MAX);
    LdsU32* my_xa = (LdsU32*)(s_ev + tid * EV_MAX);                                // the same 16 bytes, as a stream-A state (replayed reads)
    int nev = 0, n_out = 0; bool replay = false; uint32_t replay_first = 0, del_pos = 0xFFFFu; bool is_i1 = false;   // D1 class: its one event -- the deleted base, or (is_i1) the base behind which one base is inserted
    if (live) {
        if (FROM_PAIRS) {
            const uint32_t h = ev_h; const uint4 e = ev_e;
            n_out = (int)(h & 0xFFFFu); nev = SIMPLE ? 0 : (int)((h >> 16) & 0xFFu); replay = SIMPLE ? false : (h >> 24) & 1u;
            if (D1) { del_pos = ev_pos(e.x & 0xFFFFu); is_i1 = !ev_del(e.x & 0xFFFFu); }
            if (!UNI) { my_xa[0] = e.x; my_xa[1] = e.y; my_xa[2] = e.z; my_xa[3] = e.w; }   // 8 x 16-bit events
        } else {
            const IndelPass ip = indel_pass(tb, key, aux, uid, force_replay, slot, flags, [&](int i, uint32_t v) { my_ev[i] = (uint16_t)v; });
            n_out = ip.n_out; nev = ip.nev; replay = ip.replay;
        }
        if (replay) {                                                              // phase 2 draws the events again: stream A from its start
            Xoshiro xa; xa.seed(draw4(key, ST_READ, aux, uid, 0));
            replay_first = indel_gap(tb.gap_t, xa.next(), (uint32_t)n);             // where the first event sits
            my_xa[0] = xa.s0; my_xa[1] = xa.s1; my_xa[2] = xa.s2; my_xa[3] = xa.s3;
        }
    }
    lds_barrier();                                                               // also: everyone is done with s_gbase/s_gflag (ring alias)
    SCS_PHASE(2);

    // ---- phase 2: the base pass (Profile.cpp:1632-1694), workgroup-synchronous over the TABLE BINS.  Output position j of a
    // read uses the rows of bin j*binCount/n'; the workgroup walks the bins together and every read emits the positions
    // that fall into the current bin: one each for a read of unchanged length (binCount == L), none or two at the places
    // where indels changed n'.  So every lookup of every read hits the one bin the ring is serving.
    const bool second = rd != 0 && tb.subs2 != nullptr;
    const uint32_t* __restrict__ subs = second ? tb.subs2 : tb.subs1;
    const double* __restrict__ subs_d = second ? tb.subs2_d : tb.subs1_d;
    // the ring holds the substitution rows of the workgroup's mate (explicit-window mode: of read 1)
    const bool ring_subs_ok = FROM_PAIRS ? true : !second;
    int ji = 0, jo = 0, ins_left = 0, evi = 0;
    uint32_t next_ev = replay ? replay_first : nev > 0 ? ev_pos(my_ev[0]) : 0xFFFFFFFFu;   // input position of the next indel event
    // binIndx = j*binCount/n' (Profile.cpp:1668) as a multiply-high: exact while j*binCount*n' < 2^32 (checked on the host)
    const uint32_t mdiv = n_out > 0 ? 0xFFFFFFFFu / (uint32_t)n_out + 1u : 0u;     // ceil(2^32 / n')
    uint32_t nb = 0;                                                               // bin of my position jo
    uint32_t c0 = 5u, c1 = 5u;
    uint32_t cur_b = 0, cur_q = 0, ob0 = 0, ob1 = 0, ob2 = 0, ob3 = 0, oq0 = 0, oq1 = 0, oq2 = 0, oq3 = 0;   // ob/oq: slot mode's 16-byte blocks
    Xoshiro xb; xb.seed(draw4(key, ST_READ, aux, uid, 1));                         // stream B: substitution / quality draws, in output order
    char* my_b = FROM_PAIRS ? nullptr : slot_b + (size_t)r * slot; char* my_q = FROM_PAIRS ? nullptr : slot_q + (size_t)r * slot;
    // pair mode: the two byte streams of my FASTQ record (see BlockOut above)
    BlockOut bo_b, bo_q; uint32_t a1 = 0, a2 = 0, sec1 = 0, sec2 = 0;             // a = T & 31 of each stream, sec = offset of its first sector from wg_out
    bo_b.carry = 0; bo_q.carry = 0x0A2B0A00u;                                      // "\n+\n" rides ahead of the qualities
#pragma unroll
    for (int i = 0; i < 4; ++i) { bo_b.R[i] = bo_q.R[i] = 0; bo_b.H[i] = bo_q.H[i] = 0; if (i < 3) bo_b.P[i] = bo_q.P[i] = 0; }
    // the record must lie inside the batch's text (its offset and size come from k_indels' n'; the walk below emits exactly
    // n' characters per stream): a disagreement would be an internal error, reported, never a store outside the buffer
    if (FROM_PAIRS && live && n_out > 0 && my_off + rec_h + 2ull * (uint32_t)n_out + 4ull > (second_file ? cap2 : cap1)) { atomicOr(flags, (uint32_t)FLAG_INTERNAL); live = false; n_out = 0; }
    LdsU32* my_pend_lds = (LdsU32*)(s_win + (size_t)tid * ROW);                    // rows are dword aligned (win_stride)
    // One-event class, a read with ONE INSERTED base (is_i1): n' = L + 1 positions in the bins j L / (L + 1) = 0, 0, 1, 2, ... -- position 0
    // is made HERE, ahead of the walk (bin 0: the ring's first group and the head rows go to LDS first), and rides in front of the two
    // streams: its base as the last character of the name line, its quality behind "\n+\n".  The walk's step t then makes position
    // t + 1 at bin t, as the other walks' step t makes position t.
    uint32_t b0ch = 0, q0ch = 0, c1_pre = 0;
    const uint32_t sh1 = (D1 && is_i1 && live && n_out > 0) ? 1u : 0u;
    if constexpr (D1) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) { const int idx = tid + u * RB; if (idx < GE) ring16[idx] = ring0[u]; }
        if (tid < 128) s_head[tid] = head_w;
        lds_barrier();
        if (__any(sh1 != 0u)) {
            if (sh1) {
                const uint32_t cb = ((const LdsU32*)(s_win + (size_t)tid * ROW + WOFF))[0] & 3u;   // window base 0
                uint32_t x1, x2; xb.next2(x1, x2);
                const u32x2_t kp = *(const LdsU2*)((const LdsU8*)s_head + cb * 8u);  // the 1-mer's keep interval at bin 0
                uint32_t k = cb, qv;
                if (x1 - kp.x < kp.y) {
                    const LdsU32* qrow = (const LdsU32*)((const LdsU8*)s_dyn + cb * (uint32_t)(QROW * 16));   // bin 0 (ring slot 0): the diagonal row (cb, cb)
                    qv = alias_pick<QK>(qrow, (const LdsU8*)(qrow + QK), x2);
                } else { const uint32_t kq = call_global<QK>(subs, subs_d, tb.qual_alias, (uint32_t)B, (int)cb, cb, cb, 0u, x1, x2); k = kq & 255u; qv = kq >> 8; }
                b0ch = (0x54474341u >> (8u * k)) & 255u; q0ch = 33u + qv; c1_pre = cb;
            }
        }
    }
    if (UNI && live && n_out > 0) {
        // The name line "@<amp>#<cnt>[/1|/2]\n" is WRITTEN INTO LDS FIRST, character by character at its place from the end -- into the
        // 28 bytes in front of my window that the walk's pending entries use later, the line's last character in byte 27 (h <= 25) --
        // and read back as seven words: no six-way choice per character, the digit loops are the only data-dependent part.  The words are
        // shifted to the record's alignment: the last s1 characters ride in the first dword of the bases, the rest ends on the aligned
        // address ta1 and goes out as whole dwords, then the <= 3 leading bytes.
        // (sh1: the line carries the base of position 0 behind its "\n"; the streams start one character later)
        const uint32_t amp = amp_index_base + pr.amp, cnt = pr.att + 1u, d2 = dec_digits(cnt), h = rec_h + sh1, dbase = paired ? 3u : 1u, da = rec_h - 2u - dbase - d2;
        const uint32_t o1 = rec_rel + h, o2 = o1 + (uint32_t)n_out + 3u;          // where the bases / the qualities start
        a1 = o1 & 31u; sec1 = o1 - a1; a2 = o2 & 31u; sec2 = o2 - a2;
        const uint32_t s1 = a1 & 3u; char* ta1 = wg_out + (o1 - s1);
        LdsU8* nb = (LdsU8*)my_pend_lds;
        if (sh1) { nb[27] = (uint8_t)b0ch; bo_q.carry = (q0ch << 24) | 0x000A2B0Au; }
        nb[27u - sh1] = (uint8_t)'\n';
        if (paired) { nb[26u - sh1] = (uint8_t)(rd ? '2' : '1'); nb[25u - sh1] = (uint8_t)'/'; }
        uint32_t at = 27u - sh1 - dbase, v = cnt;                                  // byte of the next character to the left
#pragma unroll
        for (uint32_t j = 0; j < 10; ++j) if (j < d2) { const uint32_t qv = v / 10u; nb[at - j] = (uint8_t)('0' + (v - qv * 10u)); v = qv; }
        at -= d2; nb[at] = (uint8_t)'#'; at -= 1u; v = amp;
#pragma unroll
        for (uint32_t j = 0; j < 10; ++j) if (j < da) { const uint32_t qv = v / 10u; nb[at - j] = (uint8_t)('0' + (v - qv * 10u)); v = qv; }
        nb[at - da] = (uint8_t)'@';
        uint32_t nm[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) nm[k] = ((const LdsU32*)nb)[k];
        bo_b.carry = s1 ? nm[6] & (0xFFFFFFFFu << (8u * (4u - s1))) : 0u;
        uint32_t w[7];                                                             // w[k]: the aligned dword that ends 4 (6 - k) bytes before ta1
#pragma unroll
        for (int k = 6; k >= 0; --k) w[k] = s1 ? __builtin_amdgcn_alignbyte(nm[k], k ? nm[k - 1] : 0u, 4u - s1) : nm[k];
        const uint32_t nd = (h - s1) >> 2, nl = (h - s1) & 3u;                     // whole dwords, leading bytes (h >= 5 > s1)
        uint32_t lead = 0;
#pragma unroll
        for (uint32_t m = 0; m < 7; ++m) {
            if (m < nd) *reinterpret_cast<uint32_t*>(ta1 - 4u * (m + 1u)) = w[6 - m];
            lead = m == nd ? w[6 - m] : lead;
        }
        char* lp = ta1 - 4u * nd - nl;
        for (uint32_t i = 0; i < nl; ++i) lp[i] = (char)(lead >> (8u * (4u - nl + i)));
    }
    if (!UNI && FROM_PAIRS && live && n_out > 0) {
        const uint32_t amp = amp_index_base + pr.amp, cnt = pr.att + 1u, d2 = dec_digits(cnt), h = rec_h;
        const uint32_t o1 = rec_rel + h, o2 = o1 + (uint32_t)n_out + 3u;          // where the bases / the qualities start
        a1 = o1 & 31u; sec1 = o1 - a1; a2 = o2 & 31u; sec2 = o2 - a2;
        const uint32_t s1 = a1 & 3u; char* ta1 = wg_out + (o1 - s1);
        // the name line is produced backwards from its end: its last s1 characters ride in the first dword of the bases,
        // the rest ends on the aligned address ta1 and goes out as whole dwords, then the <= 3 leading bytes
        uint32_t q = 0, vc = cnt, va = amp; const uint32_t dbase = paired ? 3u : 1u;
        auto next_char = [&]() -> uint32_t {
            uint32_t ch;
            if (q == 0) ch = '\n';
            else if (q < dbase) ch = q == 1 ? (rd ? '2' : '1') : '/';
            else if (q < dbase + d2) { ch = '0' + vc % 10u; vc /= 10u; }
            else if (q == dbase + d2) ch = '#';
            else if (q < h - 1u) { ch = '0' + va % 10u; va /= 10u; }
            else ch = '@';
            ++q; return ch;
        };
        for (uint32_t i = 0; i < 3; ++i) if (i < s1) bo_b.carry |= next_char() << (8u * (3u - i));
        char* wp = ta1;
        for (uint32_t m = 0; m < 7; ++m) {
            uint32_t w = 0, nb4 = 0;
            for (uint32_t b = 0; b < 4; ++b) if (q < h) { w = (w << 8) | next_char(); ++nb4; }
            if (nb4 == 4u) { wp -= 4; *reinterpret_cast<uint32_t*>(wp) = w; }
            else if (nb4) { wp -= nb4; for (uint32_t i = 0; i < nb4; ++i) wp[i] = (char)(w >> (8u * i)); }
        }
    }
    // the ring's first group and the head rows leave their registers (loaded at the kernel's start) for the LDS the staging has freed
    if constexpr (!D1) {                                                           // (the one-event class: done above, ahead of its first position)
#pragma unroll
        for (int u = 0; u < NPRE; ++u) { const int idx = tid + u * RB; if (idx < GE) ring16[idx] = ring0[u]; }
        if (UNI && tid < 128) s_head[tid] = head_w;
        lds_barrier();
    }
    SCS_PHASE(3);

    // a substituted base (k != c2) needs an off-diagonal quality row, which only global memory holds.  Its quality does
    // not feed back into the walk, so the lookup is deferred: (position, k, c2, bin, draw) is set aside and resolved after
    // the loop, off the workgroup-synchronous path (resolved in place, one lane of the wave fetches from global memory while
    // the other 63 wait: with a substitution every few hundred bases that was 15 % of this kernel).  Pair mode keeps the
    // entries in the START OF THE READ'S OWN WINDOW ROW in LDS -- entry e over the bases 16e .. 16e+15, dead once the walk
    // has passed them -- and patches the quality character into the FASTQ text after the record is written; slot mode in
    // the free tail of the read's quality slot.  No room (an early position, a fifth substitution) -> resolved in place.
    constexpr uint32_t PEND_MAX = 4, PENDU_MAX = 6;                                // (PENDU_MAX: the uniform walk's entries, three words each)
    const bool can_defer = !FROM_PAIRS && n_out + 15 + (int)(8 * PEND_MAX) <= (int)slot;
    uint32_t npend = 0;
    uint2* my_pend = FROM_PAIRS ? nullptr : reinterpret_cast<uint2*>(my_q + slot - 8 * PEND_MAX);

    bool redo = false;                                                             // UNI: the read is made again after the pass (redo_read)
    if constexpr (UNI) {
        // The event-free, ACGT-only class: every read emits exactly position t at bin t and every position takes exactly two
        // draws, so the whole walk is WAVE-UNIFORM -- window base t (a dword of 16 bases is fetched every 16th step), output
        // word t >> 2, a 16-character block at t & 15 == 15 -- and, between two blocks, STRAIGHT-LINE code but for one wave-uniform
        // branch per position: ring slot, word index and table addresses are compile-time or lane arithmetic.  The base call
        // k = (x1 >= T0) + (x1 >= T1) + (x1 >= T2) keeps the window's base c2 in all but a few draws per thousand, and k = c2 is ONE
        // compare against the 3-mer's keep interval (lo, width) from the ring (RingBinU): the walk writes c2 and the diagonal quality
        // row's symbol.  A position whose draw does not keep the base (or draws 0xFFFFFFFF, whose call needs the double tables) is
        // set aside -- (position | table row, x1, x2), three words in the lane's row -- inside a block that only a wave with such a
        // lane enters, and resolved after the pass from the global tables (finish_b and the loop behind the walk).  No room, or the
        // draw 0xFFFFFFFF: the read is flagged and made again by redo_read.
        // Lanes without a read run along on an all-'A' window and store nothing.
        const bool mine = live && n_out > 0;
        const LdsU8* ring8 = (const LdsU8*)s_dyn;
        const LdsU32* win32 = (const LdsU32*)my_win;
        const LdsU8* head8 = (const LdsU8*)s_head;
        const bool force_redo = (force_replay & 2u) != 0;
        char* __restrict__ spare = reinterpret_cast<char*>(flags) + 128;               // 32 bytes nobody reads (the flags buffer is 256 bytes)
        uint32_t wreg = 0, wnext = 0, sel = 0, qacc = 0, nbad = 0;
        c0 = 0; c1 = c1_pre;
        // D1 (CLS 3): the reads with exactly ONE indel event, the deletion of ONE base, run the same walk.  Their n' = L - 1 positions
        // fall into bins j * L / (L - 1) = j, one step of stream B each; only the source base differs: position j reads window
        // base j before the deleted base and j + 1 from it on (a shift by one two-bit field, the next dword kept beside the
        // current one).  NP: the positions of a read of this class.
        // ... and the reads whose one event is the INSERTION of one base (is_i1; ahead of the walk: position 0, above): step t makes position
        // t + 1 at bin t from window base t + 1 before the event's base e, the inserted base (a draw of its own) at t = e, window base t behind
        // it; B steps.  A one-deletion read in the same wave idles through the last step (its draw kept, its character cleared).  When B - 1 is
        // a multiple of 16 that step would open a block of its own: k_indels sends no insertion reads here then and the walk has B - 1 steps.
        const bool i1_ok = D1 && ((B - 1) & 15) != 0;
        const int NP = (D1 && !i1_ok) ? B - 1 : B;
        // SOFTWARE PIPELINE, one position deep.  The compiler's scheduler waits for an LDS read right where it issues it; here
        // every read gets a stage's worth of independent work before its use.  Step u runs, in this order,
        //   finish_a(u-1): thresholds and alias entry of the previous position are back -> its base k, its alias column -> issue the symbol read
        //   start(u):      this position's window base, stream-B step, table addresses -> issue the threshold and alias-entry reads
        //   finish_b(u-1): the symbol is back -> pending-quality bookkeeping, the output words
        // with the ring's refill (commit, barrier, prefetch) between finish_a and start: every ring read of a group is issued
        // before its wave arrives at the next group's barrier, as the refill's slot reuse assumes.
        uint32_t aLo = 0, aWid = 0, aE = 0, aX1 = 0, aX2 = 0, aC2 = 0, aRow = 0; const LdsU8* aQ = ring8; bool aIdle = false;   // start -> finish_a
        uint32_t bC2 = 0, bX1 = 0, bX2 = 0, bSym = 0, bRow = 0; bool bKept = true;                          // finish_a -> finish_b
        // (FIRST: t0 == 0, as a compile-time constant -- a run-time test would put branches between the positions, and a branch around
        // the stores makes the compiler's s_waitcnt before the next ring commit cover them on every path: vmcnt(1) instead of vmcnt(5))
        auto start = [&](auto U, auto FIRST, int t0) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            constexpr bool first = decltype(FIRST)::value;
            uint32_t c2;
            if constexpr (D1) {
                if (u == 0) { wreg = first ? win32[0] : wnext; wnext = win32[(t0 >> 4) + 1]; if (!mine) { wreg = 0; wnext = 0; } }   // 16 bases + the 16 behind them
                const bool after = ((uint32_t)(t0 + u) >= del_pos) != is_i1;         // from the deleted base on -- or up to the base the insertion follows: the next window base
                if (u < 15) c2 = __builtin_amdgcn_ubfe(wreg, 2u * u + (after ? 2u : 0u), 2u);
                else c2 = after ? (wnext & 3u) : (wreg >> 30);
                if (__ballot(is_i1 && (uint32_t)(t0 + u) == del_pos)) {                // the inserted base: randomInteger(0, 3), a step of stream B of its own (Profile.cpp:1640-1646)
                    if (is_i1 && (uint32_t)(t0 + u) == del_pos) { uint32_t xi, xu; xb.next2(xi, xu); c2 = scale_draw(xi, 0, 3); }
                }
            } else {
                if (u == 0) { wnext = wreg; wreg = win32[t0 >> 4]; if (!mine) wreg = 0; }   // the block's 16 bases (wnext: the block before)
                c2 = __builtin_amdgcn_ubfe(wreg, 2u * u, 2u);
            }
            uint32_t x1, x2; xb.next2(x1, x2);                                       // one step of stream B per position
            const LdsU8* bin8 = ring8 + (u & (SLOTS - 1)) * sizeof(Bin);
            // the clean 3-mer, OLDEST base in the low bits (the ring's keep table is laid out that way): six bits of the window as they lie
            uint32_t rowi;
            if constexpr (D1) rowi = c0 | (c1 << 2) | (c2 << 4);
            else if constexpr (u >= 2) rowi = __builtin_amdgcn_ubfe(wreg, 2u * (u - 2), 6u);
            else rowi = __builtin_amdgcn_alignbit(wreg, wnext, u == 0 ? 28u : 30u) & 63u;
            const LdsU8* kp8 = bin8 + 4 * QROW * 16 + rowi * 8u;
            if constexpr (u < 2 && first) {                                          // the read's first two bases: 1-mer / 2-mer rows (table rows 0..19)
                const uint32_t row3 = rowi;
                rowi = u == 0 ? c2 : 4u + c1 * 4u + c2; kp8 = head8 + rowi * 8u;
                if constexpr (D1) if (is_i1) {                                       // these steps make positions 1 and 2: the 2-mer at bin 0 (head rows 20..35), the 3-mer at bin 1
                    if (u == 0) { rowi = 4u + c1 * 4u + c2; kp8 = head8 + (16u + rowi) * 8u; }
                    else { rowi = 20u + (((row3 & 3u) << 4) | (row3 & 12u) | (row3 >> 4)); kp8 = bin8 + 4 * QROW * 16 + row3 * 8u; }   // (rowi: the TABLE row, as finish_b's HEAD form takes it)
                }
            }
            const LdsU32* qrow = (const LdsU32*)(bin8 + c2 * (uint32_t)(QROW * 16));   // the diagonal row (c2, c2) as an alias row
            const u32x2_t kp = *(const LdsU2*)kp8;                                    // the draws that keep the base: lo <= x1 < lo + width
            aLo = kp.x; aWid = kp.y; aE = qrow[x2 >> (32u - Geo::ABITS)];
            aQ = (const LdsU8*)(qrow + QK); aX1 = x1; aX2 = x2; aC2 = c2; aRow = rowi;
            if constexpr (D1) aIdle = i1_ok && !is_i1 && t0 + u == B - 1;
            c0 = c1; c1 = c2;
            __builtin_amdgcn_sched_barrier(0);
        };
        auto finish_a = [&]() __attribute__((always_inline)) {
            bKept = aX1 - aLo < aWid;                                                 // k == c2 (never for the draw 0xFFFFFFFF)
            if constexpr (D1) if (aIdle) bKept = true;                                // (a one-deletion read's idle last step)
            const uint32_t col = aX2 >> (32u - Geo::ABITS);
            const uint32_t pick = ((aX2 << Geo::ABITS) | (uint32_t)(QK - 1)) < aE ? col : (aE & (uint32_t)(QK - 1));   // alias_pick
            bSym = aQ[pick];
            bC2 = aC2; bX1 = aX1; bX2 = aX2; bRow = aRow;
            __builtin_amdgcn_sched_barrier(0);
        };
        auto finish_b = [&](auto U, auto HEAD, int t, bool last) __attribute__((always_inline)) {   // t: the position being finished, u = t & 15
            constexpr int u = decltype(U)::value;
            // The walk writes the WINDOW's base and the diagonal row's quality; a position whose draw does not keep the base (a
            // substitution: a few per thousand; or the draw 0xFFFFFFFF) is set aside -- (position | table row, x1, x2), three words --
            // and resolved after the pass from the global tables, base and quality patched into the text.  Only a wave in which SOME
            // lane has one enters the block (a wave-uniform branch: one position in four or five); lanes without a read count along,
            // ignored later.  Entry e lies in the row's dwords 3e .. 3e + 2: the first three in front of the window, entry e >= 3 over the
            // window's dwords 3e - 9 .. 3e - 7, which the walk has read once t >= 16 (3e - 7); at most PENDU_MAX entries.  (With two
            // entries in front, one read in a thousand ran out of room and was made again by redo_read: a workgroup in four had one,
            // 3.5 of its 77 microseconds on average.)
            if (__ballot(!bKept)) {
                const bool bad = !bKept, ugly = bX1 == 0xFFFFFFFFu;
                const bool wr = bad & !ugly & (npend <= min((((uint32_t)t >> 4) + 7u) / 3u, PENDU_MAX - 1u)) & !force_redo;
                nbad += bad ? 1u : 0u;                                                // nbad != npend after the pass: the read is made again
                const uint32_t ki = decltype(HEAD)::value ? bRow : 20u + (((bRow & 3u) << 4) | (bRow & 12u) | (bRow >> 4));   // table row: newest base in the low bits
                if (wr) { LdsU32* e = my_pend_lds + 3u * npend; e[0] = (uint32_t)t | (ki << 10); e[1] = bX1; e[2] = bX2; }
                npend += wr ? 1u : 0u;
            }
            sel |= bC2 << (8 * (u & 3)); qacc |= bSym << (8 * (u & 3));               // base selectors and raw qualities, four to a word
            if ((u & 3) == 3 || last) {
                uint32_t wb = __builtin_amdgcn_perm(0x4Eu, 0x54474341u, sel), wq = qacc + 0x21212121u;   // selector 0..3 -> "ACGT"; + 33
                if constexpr ((u & 3) != 3) { const uint32_t m = (1u << (8 * ((u & 3) + 1))) - 1u; wb &= m; wq &= m; }   // the read's last, partial word
                if constexpr (D1) if (last && i1_ok && !is_i1) { const uint32_t m = (1u << (8 * (u & 3))) - 1u; wb &= m; wq &= m; }   // (one deletion: the idle step's character is not the read's)
                bo_b.R[u >> 2] = wb; bo_q.R[u >> 2] = wq; sel = 0; qacc = 0;
            }
        };
        // MODE 2: all 16 positions exist, t0 >= 48: the previous block (m >= 2) leaves as straight-line stores; 1: all 16 exist, t0 < 48
        // (blocks 0 and 1, stored at t0 = 16 and 32, may hold a stream's first, partial sector: branching stores); 0: the read's last block
        auto steps = [&](auto MODE, auto FIRST, int t0) __attribute__((always_inline)) {
            unroll_steps([&](auto U) __attribute__((always_inline)) {
                constexpr int u = decltype(U)::value;
                constexpr int mode = decltype(MODE)::value;
                constexpr bool first = decltype(FIRST)::value;                       // t0 == 0
                const int t = t0 + u;
                if (mode == 0 && t >= NP) return;
                if constexpr (u > 0 || !first) finish_a();
                if constexpr ((u & (GROUP - 1)) == 0 && (u > 0 || !first)) { commit(t); lds_barrier(); prefetch(t + GROUP); }
                start(U, FIRST, t0);
                if constexpr (u > 0 || !first) finish_b(std::integral_constant<int, (u + 15) & 15>{}, std::integral_constant<bool, first && (u == 1 || u == 2)>{}, t - 1, false);
                // The previous 16 characters leave HERE, right behind the ring's loads.  On this hardware loads and stores complete
                // out of order with each other, so a wait for a load (the next commit) is a wait for EVERY outstanding store too
                // (s_waitcnt vmcnt(0)); placed here that wait comes a whole group of positions after the stores, when their round
                // trip to L2 is over.  (The raw words R[] of the stored block are not overwritten before u = 4.)
                if constexpr (u == 0 && !first) {
                    if constexpr (mode == 2) {
                        bo_b.block_flat(wg_out, sec1, a1, (uint32_t)(t0 >> 4) - 1u, spare, mine);
                        bo_q.block_flat(wg_out, sec2, a2, (uint32_t)(t0 >> 4) - 1u, spare, mine);
                    } else if (mine) {
                        bo_b.block(wg_out, sec1, a1, (uint32_t)(t0 >> 4) - 1u);
                        bo_q.block(wg_out, sec2, a2, (uint32_t)(t0 >> 4) - 1u);
                    }
                }
            }, std::make_integer_sequence<int, 16>{});
        };
        int t0 = 0;
        if (16 < NP) { steps(std::integral_constant<int, 1>{}, std::true_type{}, 0); t0 = 16; }
        for (; t0 < 48 && t0 + 16 < NP; t0 += 16) steps(std::integral_constant<int, 1>{}, std::false_type{}, t0);
        for (; t0 + 16 < NP; t0 += 16) steps(std::integral_constant<int, 2>{}, std::false_type{}, t0);
        if (t0 == 0) steps(std::integral_constant<int, 0>{}, std::true_type{}, 0); else steps(std::integral_constant<int, 0>{}, std::false_type{}, t0);
        finish_a();                                                                 // drain: the read's last position
        unroll_steps([&](auto U) __attribute__((always_inline)) { if (decltype(U)::value == ((NP - 1) & 15)) finish_b(U, std::false_type{}, NP - 1, true); }, std::make_integer_sequence<int, 16>{});
        redo = mine && nbad != npend;
        if (!mine) npend = 0;
        SCS_PHASE(4);
    } else
    for (int t = 0; t < B; ++t) {
        if ((t & (GROUP - 1)) == 0 && t > 0) {
            commit(t);
            lds_barrier();
            prefetch(t + GROUP);
        }
        const Bin* rb = &s_ring[t & (SLOTS - 1)];
        // base call + quality from the ring (clean k-mer kk): 0 = done, 1 = base substituted (its quality row is not in the
        // ring), 2 = needs the global tables altogether
        auto call_lds = [&](uint32_t kk, uint32_t c2, uint32_t xs, uint32_t xq, uint32_t& k, uint32_t& qv) -> uint32_t {
            const LdsU32* st = (const LdsU32*)rb->subs[kk];
            k = (xs >= st[0]) + (xs >= st[1]) + (xs >= st[2]);
            const LdsU32* qrow = (const LdsU32*)rb->qd[c2 & 3u];                     // the diagonal row (c2, c2) as an alias row
            qv = alias_pick<QK>(qrow, (const LdsU8*)(qrow + QK), xq);
            return xs == 0xFFFFFFFFu ? 2u : (k != c2 ? 1u : 0u);
        };
        auto defer = [&](uint32_t k, uint32_t c2, uint32_t xq) -> bool {
            const uint32_t w0 = (uint32_t)jo | (k << 12) | (c2 << 14) | ((uint32_t)t << 16);
            if (FROM_PAIRS) {
                if (npend >= PEND_MAX || (uint32_t)ji < 16u * (npend + 1u)) return false;   // ji: the next base the walk reads
                my_pend_lds[2u * npend] = w0; my_pend_lds[2u * npend + 1u] = xq; ++npend;
            } else {
                if (!can_defer || npend >= PEND_MAX) return false;
                my_pend[npend++] = make_uint2(w0, xq);
            }
            return true;
        };
        for (;;) {
            // my position jo falls into bin t (an event-free read has n' = binCount: position t, once)
            const bool mine = SIMPLE ? jo < n_out : (jo < n_out && nb == (uint32_t)t);
            if (!__any(mine)) break;
            // ---- (A) the source base of this output position (Profile.cpp:1632-1654, walked lazily)
            uint32_t c2 = win_get(my_win, ji);                                     // the common case: the next window base
            if (!SIMPLE && mine && (ins_left > 0 || (uint32_t)ji == next_ev)) {    // rare lanes: inside an insertion / at an indel event
                if (ins_left > 0) { uint32_t xi, xu; xb.next2(xi, xu); c2 = scale_draw(xi, 0, 3); --ins_left; }   // inserted base (a step of its own): randomInteger(0, N-1) -> never 'T'
                else {
                    if (replay) {                                                  // the events of phase 1, drawn again (same stream, same order)
                        Xoshiro xa; xa.s0 = my_xa[0]; xa.s1 = my_xa[1]; xa.s2 = my_xa[2]; xa.s3 = my_xa[3];
                        for (;;) {                                                 // an event at base ji
                            const uint32_t y = xa.next();
                            const uint32_t x = draw4(key, ST_INDEL_LEN, aux, uid, (uint32_t)ji).w[0];
                            if (y < tb.t_kind) ins_left = (int)rand_indx_thr(tb.ins_t, tb.ins_d, (uint32_t)tb.n_ins, x);
                            else {
                                const uint32_t k = rand_indx_thr(tb.del_t, tb.del_d, (uint32_t)tb.n_del, x);
                                if (k > 0) {                                       // the walk resumes behind the deleted bases, which may start with an event again
                                    ji += (int)k < n - ji ? (int)k : n - ji;
                                    const uint32_t gq = ji < n ? indel_gap(tb.gap_t, xa.next(), (uint32_t)(n - ji)) : 1u;
                                    if (gq == 0) continue;
                                    c2 = win_get(my_win, ji); next_ev = (uint32_t)ji + gq; ++ji;
                                    break;
                                }
                            }
                            c2 = win_get(my_win, ji); ++ji;                         // base ji is kept (an insertion follows it, or nothing happened)
                            next_ev = ji < n ? (uint32_t)ji + indel_gap(tb.gap_t, xa.next(), (uint32_t)(n - ji)) : 0xFFFFFFFFu;
                            break;
                        }
                        my_xa[0] = xa.s0; my_xa[1] = xa.s1; my_xa[2] = xa.s2; my_xa[3] = xa.s3;
                    } else {
                        while (evi < nev) {                                        // deletions starting here
                            const uint32_t ev = my_ev[evi];
                            if (ev_pos(ev) != (uint32_t)ji || !ev_del(ev)) break;
                            ji += (int)ev_len(ev); ++evi;
                        }
                        c2 = win_get(my_win, ji);
                        if (evi < nev) { const uint32_t ev = my_ev[evi]; if (ev_pos(ev) == (uint32_t)ji) { ins_left = (int)ev_len(ev); ++evi; } }
                        ++ji;
                        next_ev = evi < nev ? ev_pos(my_ev[evi]) : 0xFFFFFFFFu;
                    }
                }
            } else if (mine) ++ji;
            // ---- (B) base call + quality (Profile.cpp:1666-1694)
            uint32_t bc = 0, qc = 0;
            if (__any(mine && ((c0 | c1 | c2) > 3u || !ring_subs_ok))) {           // some read of the wave: first two bases, an N in the k-mer
                if (mine) {
                    const int ki = kmer_index(c0, c1, c2);
                    uint32_t xs, xq; xb.next2(xs, xq);                             // (xs unused when the k-mer has no row)
                    if (ki < 0 && c2 > 3u) { bc = 'N'; qc = 33 + scale_draw(xq, 0, 20); }   // getRandBaseQuality
                    else {
                        uint32_t k = c2, qv = 0, odd = 2u;
                        if (ki >= 20 && ring_subs_ok) odd = call_lds((uint32_t)ki - 20u, c2, xs, xq, k, qv);
                        if (odd == 1u && defer(k, c2, xq)) { qv = 0; odd = 0u; }
                        if (odd) {
                            const uint32_t kq = call_global<QK>(subs, subs_d, tb.qual_alias, (uint32_t)B, ki, c2, c2, (uint32_t)t, xs, xq);
                            k = kq & 255u; qv = kq >> 8;
                        }
                        bc = (0x54474341u >> (8u * k)) & 255u; qc = 33 + qv;       // "ACGT"[k]
                    }
                }
            } else if (mine) {                                                     // the whole wave on clean k-mers
                uint32_t xs, xq; xb.next2(xs, xq);
                const uint32_t kk = (c0 << 4) | (c1 << 2) | c2;
                uint32_t k, qv;
                uint32_t odd = call_lds(kk, c2, xs, xq, k, qv);
                if (odd == 1u && defer(k, c2, xq)) { qv = 0; odd = 0u; }
                if (odd) {
                    const uint32_t kq = call_global<QK>(subs, subs_d, tb.qual_alias, (uint32_t)B, (int)kk + 20, c2, c2, (uint32_t)t, xs, xq);
                    k = kq & 255u; qv = kq >> 8;
                }
                bc = (0x54474341u >> (8u * k)) & 255u; qc = 33 + qv;
            }
            // ---- output: 4 characters per word, 16 per store
            if (mine) {
                const uint32_t sh = 8u * ((uint32_t)jo & 3u);
                cur_b |= bc << sh; cur_q |= qc << sh; c0 = c1; c1 = c2;
                const bool lastp = jo == n_out - 1;
                if (((uint32_t)jo & 3u) == 3u || lastp) {
                    if (!FROM_PAIRS) {                                             // slot mode (explicit windows): 16-byte blocks
                        const uint32_t w = ((uint32_t)jo >> 2) & 3u;
                        ob0 = w == 0u ? cur_b : ob0; ob1 = w == 1u ? cur_b : ob1; ob2 = w == 2u ? cur_b : ob2; ob3 = w == 3u ? cur_b : ob3;
                        oq0 = w == 0u ? cur_q : oq0; oq1 = w == 1u ? cur_q : oq1; oq2 = w == 2u ? cur_q : oq2; oq3 = w == 3u ? cur_q : oq3;
                        if (((uint32_t)jo & 15u) == 15u || lastp) {
                            const int o = jo & ~15;
                            *reinterpret_cast<uint4*>(my_b + o) = make_uint4(ob0, ob1, ob2, ob3);
                            *reinterpret_cast<uint4*>(my_q + o) = make_uint4(oq0, oq1, oq2, oq3);
                            ob0 = ob1 = ob2 = ob3 = 0; oq0 = oq1 = oq2 = oq3 = 0;
                        }
                    } else {                                                       // FASTQ text: 16-character blocks of each stream -> whole sectors
                        const uint32_t w = ((uint32_t)jo >> 2) & 3u;
                        bo_b.R[0] = w == 0u ? cur_b : bo_b.R[0]; bo_b.R[1] = w == 1u ? cur_b : bo_b.R[1]; bo_b.R[2] = w == 2u ? cur_b : bo_b.R[2]; bo_b.R[3] = w == 3u ? cur_b : bo_b.R[3];
                        bo_q.R[0] = w == 0u ? cur_q : bo_q.R[0]; bo_q.R[1] = w == 1u ? cur_q : bo_q.R[1]; bo_q.R[2] = w == 2u ? cur_q : bo_q.R[2]; bo_q.R[3] = w == 3u ? cur_q : bo_q.R[3];
                        if (((uint32_t)jo & 15u) == 15u && !lastp) {
                            bo_b.block(wg_out, sec1, a1, (uint32_t)jo >> 4);
                            bo_q.block(wg_out, sec2, a2, (uint32_t)jo >> 4);
                        }
                    }
                    cur_b = 0; cur_q = 0;
                }
                ++jo; if (!SIMPLE) nb = __umulhi(__umul24((uint32_t)jo, (uint32_t)B), mdiv);
            }
            if (SIMPLE) break;
        }
    }
    auto tails = [&]() __attribute__((always_inline)) {
        if (FROM_PAIRS && live && n_out > 0) {
            // the streams' ends, once per wave after the pass (reads of different lengths end at different steps): the last
            // characters + "\n+\n" up to the qualities' first dword / + "\n" to the record's end
            // (a read with one inserted base: the streams hold its positions 1 .. n' - 1, and position 0's quality rides behind "\n+\n")
            const uint32_t ns = (uint32_t)n_out - sh1;
            const uint32_t lastj = ns - 1u, m = lastj >> 4, nw = ((lastj & 15u) >> 2) + 1u, nv = (lastj & 3u) + 1u;
            const uint32_t s1 = a1 & 3u, s2 = a2 & 3u, nq = s2 + ns + 1u;
            bo_b.tail(wg_out, sec1, a1, m, nw, nv, 0x0A2B0Au | (sh1 ? q0ch << 24 : 0u), (s1 + ns + 3u + sh1 - s2) >> 2, 0u);   // the bases' dwords end where the qualities' first one starts
            bo_q.tail(wg_out, sec2, a2, m, nw, nv, 0x0Au, nq >> 2, nq & 3u);
        }
    };
    if constexpr (!UNI) tails();                                                   // (the uniform walk: below, behind the deferred positions' loads)
    SCS_PHASE(5);
    if constexpr (UNI) {
        // the positions the uniform walk set aside: base call and quality from the global tables (bin = position), patched into the
        // text behind the record's own stores (same lane: program order).  Three dependent loads per entry -- threshold row, alias
        // entry, symbol -- and a lane has up to six entries: the loads of ALL its entries leave together, level by level (three round
        // trips to memory for the wave instead of three per entry).
        uint32_t nmax = 0;
#pragma unroll
        for (uint32_t e = 0; e < PENDU_MAX; ++e) nmax += __any(npend > e) ? 1u : 0u;   // the wave's largest count (uniform)
        uint32_t pw0[PENDU_MAX], px2[PENDU_MAX], pk_[PENDU_MAX], ent[PENDU_MAX]; uint4 pT[PENDU_MAX]; uint32_t px1[PENDU_MAX];
#pragma unroll
        for (uint32_t e = 0; e < PENDU_MAX; ++e) if (e < nmax) {
            const bool on = e < npend;                                              // (a lane without entry e works on position 0 of row 0: valid addresses, nothing stored)
            pw0[e] = on ? my_pend_lds[3u * e] : 0u; px1[e] = on ? my_pend_lds[3u * e + 1u] : 0u; px2[e] = on ? my_pend_lds[3u * e + 2u] : 0u;
            pT[e] = *reinterpret_cast<const uint4*>(subs + ((size_t)(pw0[e] >> 10) * (uint32_t)B + (pw0[e] & 1023u)) * 4u);
        }
        constexpr uint32_t AB = RingGeo<QK>::ABITS;
#pragma unroll
        for (uint32_t e = 0; e < PENDU_MAX; ++e) if (e < nmax) {
            const uint32_t pos = pw0[e] & 1023u, pc = (pw0[e] >> 10) & 3u;
            pk_[e] = (px1[e] >= pT[e].x) + (px1[e] >= pT[e].y) + (px1[e] >= pT[e].z);
            ent[e] = tb.qual_alias[(size_t)((pc * 4u + pk_[e]) * (uint32_t)B + pos) * (QK + QK / 4) + (px2[e] >> (32u - AB))];
        }
        uint32_t sym[PENDU_MAX];
#pragma unroll
        for (uint32_t e = 0; e < PENDU_MAX; ++e) if (e < nmax) {
            const uint32_t pos = pw0[e] & 1023u, pc = (pw0[e] >> 10) & 3u, col = px2[e] >> (32u - AB);
            const uint32_t pick = ((px2[e] << AB) | (uint32_t)(QK - 1)) < ent[e] ? col : (ent[e] & (uint32_t)(QK - 1));   // alias_pick
            sym[e] = reinterpret_cast<const uint8_t*>(tb.qual_alias + (size_t)((pc * 4u + pk_[e]) * (uint32_t)B + pos) * (QK + QK / 4) + QK)[pick];
        }
        // (the streams' ends go out HERE: the loads above do not queue up behind their stores, and the patches below follow them)
        tails();
#pragma unroll
        for (uint32_t e = 0; e < PENDU_MAX; ++e) if (e < nmax && e < npend) {
            const uint32_t pos = pw0[e] & 1023u;
            wg_out[sec1 + a1 + pos] = (char)((0x54474341u >> (8u * pk_[e])) & 255u);
            wg_out[sec2 + a2 + pos] = (char)(33u + sym[e]);
        }
    } else
    for (uint32_t e = 0; e < npend; ++e) {                                         // deferred qualities of substituted bases
        uint2 pe;
        if (FROM_PAIRS) { pe.x = my_pend_lds[2u * e]; pe.y = my_pend_lds[2u * e + 1u]; } else pe = my_pend[e];
        const uint32_t pk = (pe.x >> 12) & 3u, pc = (pe.x >> 14) & 3u, qrow = (pc * 4u + pk) * (uint32_t)B + (pe.x >> 16);
        const uint32_t* __restrict__ arow = tb.qual_alias + (size_t)qrow * (QK + QK / 4);
        const uint32_t qv = alias_pick<QK>(arow, reinterpret_cast<const uint8_t*>(arow + QK), pe.y);
        if (FROM_PAIRS) wg_out[sec2 + a2 + (pe.x & 4095u)] = (char)(33u + qv);    // after the record's own stores (same lane: program order)
        else my_q[pe.x & 4095u] = (char)(33u + qv);
    }
    SCS_PHASE(6);
    if constexpr (UNI) {
        if (redo) {
            const int64_t dir = (pr.flags & 2u) ? -1 : 1; const uint32_t comp = pr.flags & 1u;
            const int64_t gb = rd == 0 ? pr.base + dir * (int64_t)pr.pos : pr.base + dir * (int64_t)(pr.pos + pr.isz - 1);
            const uint32_t gf = rd == 0 ? (comp | ((dir < 0) ? 2u : 0u)) : ((comp ^ 1u) | ((dir < 0) ? 0u : 2u));
            redo_read<QK>(g, gb, gf, spool.data, fpool.data, pr.e1, pr.e2, pr.k1, pr.pos, pr.isz, rd, n, (uint32_t)B, subs, subs_d, tb.qual_alias,
                          draw4(key, ST_READ, aux, uid, 1), wg_out + sec1 + a1 - sh1, wg_out + sec2 + a2 - sh1, D1 && !is_i1 ? del_pos : 0xFFFFu, D1 && is_i1 ? del_pos : 0xFFFFu);
        }
    }
    if (live && !FROM_PAIRS) {
        lens[r] = (uint32_t)n_out;
    }
    SCS_PHASE(7);
#ifdef SCS_PHASE_CLOCK
    if (UNI && CLS == 1 && tid == 0) { atomicAdd(&g_phase[15], 1ull); atomicAdd(&g_phase[12], __builtin_amdgcn_s_memtime() - ph_c0_); atomicAdd(&g_phase[13], wall_clock64() - ph_r0_); }
#endif
}

// (bid: the workgroup's index within ITS class' grid -- blockIdx.x of a launch of one class, or blockIdx.x less the grids of the
// classes in front of it in the merged launch below)
template <bool FROM_PAIRS, int QK, int CLS>
__global__ void __launch_bounds__(RB, 1024 / RB) k_reads(const uint8_t* __restrict__ g, DevErrPool spool, DevErrPool fpool, const PairRec* __restrict__ pairs,
                                              uint32_t np, int paired, const uint8_t* __restrict__ windows, const uint64_t* __restrict__ uids,
                                              const uint32_t* __restrict__ atts, const uint8_t* __restrict__ is_read1, uint32_t n_explicit,
                                              const DevTables tb, RngKey key, uint32_t slot, uint32_t n_slots_cap, uint32_t force_replay,
                                              const uint32_t* __restrict__ ev_hdr, const uint4* __restrict__ ev_dat,
                                              const uint64_t* __restrict__ off1, const uint64_t* __restrict__ off2, char* __restrict__ out1, char* __restrict__ out2,
                                              uint32_t amp_index_base, char* __restrict__ slot_b, char* __restrict__ slot_q, uint32_t* __restrict__ lens,
                                              uint32_t* __restrict__ flags, uint64_t cap1, uint64_t cap2,
                                              const uint32_t* __restrict__ list1, const uint32_t* __restrict__ list2, uint32_t nlist1, uint32_t nlist2) {
    reads_body<FROM_PAIRS, QK, CLS>(blockIdx.x, g, spool, fpool, pairs, np, paired, windows, uids, atts, is_read1, n_explicit, tb, key, slot, n_slots_cap, force_replay, ev_hdr, ev_dat,
                                    off1, off2, out1, out2, amp_index_base, slot_b, slot_q, lens, flags, cap1, cap2, list1, list2, nlist1, nlist2);
}
// The base pass of a batch as ONE launch: the workgroups of the general class first (the longest), then the one-event class,
// then the event-free class.  The three grids used to go to three streams; whether they really ran side by side depended on
// which hardware queues the process' streams had been given (the reads stage moved by +-4 % from process to process).  One
// grid leaves the mix to the workgroup dispatcher.  lists: {general, one-event, event-free} x {mate 1, mate 2}.
struct ReadLists { const uint32_t* l[3][2]; uint32_t n[3][2]; uint32_t grid[3]; };
template <int QK>
__global__ void __launch_bounds__(RB, 1024 / RB) k_reads_all(const uint8_t* __restrict__ g, const uint8_t* __restrict__ g2, DevErrPool spool, DevErrPool fpool, const PairRec* __restrict__ pairs,
                                                    uint32_t np, int paired, const DevTables tb, RngKey key, uint32_t slot, uint32_t n_slots_cap, uint32_t force_replay,
                                                    const uint32_t* __restrict__ ev_hdr, const uint4* __restrict__ ev_dat,
                                                    const uint64_t* __restrict__ off1, const uint64_t* __restrict__ off2, char* __restrict__ out1, char* __restrict__ out2,
                                                    uint32_t amp_index_base, uint32_t* __restrict__ flags, uint64_t cap1, uint64_t cap2, ReadLists rl) {
    const uint32_t b = blockIdx.x;
    if (b < rl.grid[0])
        reads_body<true, QK, 2>(b, g, spool, fpool, pairs, np, paired, nullptr, nullptr, nullptr, nullptr, 0u, tb, key, slot, n_slots_cap, force_replay, ev_hdr, ev_dat,
                                off1, off2, out1, out2, amp_index_base, nullptr, nullptr, nullptr, flags, cap1, cap2, rl.l[0][0], rl.l[0][1], rl.n[0][0], rl.n[0][1]);
    else if (b < rl.grid[0] + rl.grid[1])
        reads_body<true, QK, 3>(b - rl.grid[0], g, spool, fpool, pairs, np, paired, g2, nullptr, nullptr, nullptr, 0u, tb, key, slot, n_slots_cap, force_replay, ev_hdr, ev_dat,
                                off1, off2, out1, out2, amp_index_base, nullptr, nullptr, nullptr, flags, cap1, cap2, rl.l[1][0], rl.l[1][1], rl.n[1][0], rl.n[1][1]);
    else
        reads_body<true, QK, 1>(b - rl.grid[0] - rl.grid[1], g, spool, fpool, pairs, np, paired, g2, nullptr, nullptr, nullptr, 0u, tb, key, slot, n_slots_cap, force_replay, ev_hdr, ev_dat,
                                off1, off2, out1, out2, amp_index_base, nullptr, nullptr, nullptr, flags, cap1, cap2, rl.l[2][0], rl.l[2][1], rl.n[2][0], rl.n[2][1]);
}

void launch_plan_pairs(hipStream_t s, DevFrags fr, DevAmps semis, DevAmps fulls, uint32_t first, uint32_t n_fulls, uint32_t pair_lo, uint32_t pair_hi, const uint32_t* read_numbers,
                       const uint32_t* pair_off, SegMap gmap, DevTables tb, RngKey key, int paired, PairRec* pairs, unsigned long long* holes) {
    if (n_fulls == 0) return;
    hipLaunchKernelGGL(k_plan_pairs, dim3(cdiv(n_fulls, 256)), dim3(256), 0, s, fr, semis, fulls, first, n_fulls, pair_lo, pair_hi, read_numbers, pair_off, gmap, tb, key, paired, pairs, holes);
}
// bounds[b] = the amplicon that holds pair b * batch (the first i with pair_off[i + 1] > b * batch), b = 0 .. nb; bounds[nb] = ac
__global__ void k_batch_bounds(const uint32_t* __restrict__ pair_off, uint32_t ac, unsigned long long batch, uint32_t nb, uint32_t* __restrict__ bounds) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    if (b == nb) { bounds[b] = ac; return; }
    const unsigned long long p = (unsigned long long)b * batch;
    uint32_t lo = 0, hi = ac;                                                        // first i in [0, ac) with pair_off[i + 1] > p
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (pair_off[mid + 1] > p) hi = mid; else lo = mid + 1; }
    bounds[b] = lo;
}
void launch_batch_bounds(hipStream_t s, const uint32_t* pair_off, uint32_t ac, unsigned long long batch, uint32_t nb, uint32_t* bounds) {
    hipLaunchKernelGGL(k_batch_bounds, dim3(cdiv(nb + 1, 256)), dim3(256), 0, s, pair_off, ac, batch, nb, bounds);
}
void ReadsSide::release() {
    for (int k = 0; k < 2; ++k) { if (st[k]) (void)hipStreamDestroy(st[k]); if (join[k]) (void)hipEventDestroy(join[k]); st[k] = nullptr; join[k] = nullptr; }
    if (fork) (void)hipEventDestroy(fork);
    fork = nullptr;
}
size_t reads_lds_bytes(const DevTables& tb, bool uni) {
    const size_t ring = tb.qual_k == 16 ? RingGeo<16>::SLOTS * sizeof(RingBin<16>) : tb.qual_k == 64 ? RingGeo<64>::SLOTS * sizeof(RingBin<64>) : RingGeo<128>::SLOTS * sizeof(RingBin<128>);
    const size_t ring_u = tb.qual_k == 16 ? RingGeo<16>::SLOTS * sizeof(RingBinU<16>) : tb.qual_k == 64 ? RingGeo<64>::SLOTS * sizeof(RingBinU<64>) : RingGeo<128>::SLOTS * sizeof(RingBinU<128>);
    const size_t park = (size_t)RB * 19 * 4 + (64 + RB) * 4;                              // the prologue's parked records + sort counters (pair mode)
    if (uni) return std::max(park, ring_u + (size_t)RB * uni_row_bytes((uint32_t)tb.L) + 512);   // + the head rows
    return std::max(park, ring + (size_t)RB * EV_MAX * 2 + (size_t)RB * win_stride((uint32_t)tb.L));
}
template <bool FROM_PAIRS, int CLS, class... Args>
static void launch_reads_kernel(hipStream_t s, dim3 grid, const DevTables& tb, Args... args) {
    const size_t lds = reads_lds_bytes(tb, FROM_PAIRS && (CLS == 1 || CLS == 3));
    // > 64 KB of dynamic LDS needs the opt-in; the limit is raised to exactly what this profile needs (once per size:
    // the call sits on the host's critical path of a small job)
    static size_t opted_all[64][3] = {};                                           // (static per instantiation <FROM_PAIRS, CLS>)                                           // per device and instantiation (the attribute belongs to the device's code object)
    int dev = 0; (void)hipGetDevice(&dev);
    size_t* opted = opted_all[dev & 63];
#define SCS_LAUNCH_READS(QKV, SLOT) do { \
        if (opted[SLOT] != lds) { note_launch(hipFuncSetAttribute((const void*)k_reads<FROM_PAIRS, QKV, CLS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); opted[SLOT] = lds; } \
        hipLaunchKernelGGL((k_reads<FROM_PAIRS, QKV, CLS>), grid, dim3(RB), lds, s, args...); } while (0)
    if (tb.qual_k == 16) SCS_LAUNCH_READS(16, 0); else if (tb.qual_k == 64) SCS_LAUNCH_READS(64, 1); else SCS_LAUNCH_READS(128, 2);
#undef SCS_LAUNCH_READS
}
static uint32_t reads_force_replay() {                                             // tests: bit 0: every read with an indel takes the replay path;
    static const uint32_t v = (seam_env("SCS_EV_REPLAY") ? 1u : 0u) | (seam_env("SCS_TEST_REDO") ? 2u : 0u) | (seam_env("SCS_TEST_GENERAL") ? 4u : 0u) | (seam_env("SCS_TEST_NO_D1") ? 8u : 0u) | (seam_env("SCS_TEST_NO_I1") ? 16u : 0u);   // bit 1: every event-free read with a substitution is redone (redo_read)
    return v;
}
void launch_indels(hipStream_t s, const PairRec* pairs, uint32_t np, int paired, DevTables tb, RngKey key, uint32_t slot, uint32_t* ev_hdr, uint4* ev_dat,
                   uint32_t* sizes1, uint32_t* sizes2, uint32_t* d1f1, uint32_t* d1f2, uint32_t* flags) {
    if (np == 0) return;
    const uint32_t nreads = paired ? 2 * np : np;
    (void)slot;                                                                    // the FASTQ record takes whatever length the read has (header field: 16 bits)
    hipLaunchKernelGGL(k_indels, dim3(cdiv(nreads, 256)), dim3(256), 0, s, pairs, np, paired, tb, key, 65535u, reads_force_replay() | (tb.L > 1008 ? 4u : 0u) /* the uniform walk gathers a read with at most 64 lanes */, ev_hdr, ev_dat, sizes1, sizes2, d1f1, d1f2, flags);
}
// event-free reads and the rest as two launches over their lists (k_read_lists); the grid of a launch covers the longer of
// the two mates' lists
void launch_reads(hipStream_t s, const uint8_t* g, const uint32_t* g2, DevErrPool spool, DevErrPool fpool,
                  const PairRec* pairs, uint32_t np, uint32_t amp_index_base, DevTables tb, const DevTables* d_tb, RngKey key, int paired, uint32_t slot,
                  const uint32_t* ev_hdr, const uint4* ev_dat, const uint64_t* off1, const uint64_t* off2, char* out1, char* out2, uint32_t* flags,
                  uint64_t cap1, uint64_t cap2, const uint32_t* slist1, const uint32_t* slist2, const uint32_t* clist1, const uint32_t* clist2, uint32_t nc1, uint32_t nc2,
                  const uint32_t* dlist1, const uint32_t* dlist2, uint32_t nd1, uint32_t nd2, ReadsSide* side) {
    if (np == 0) return;
    (void)d_tb;
    static const bool shrink = seam_env("SCS_TEST_SHRINK_OUT") != nullptr;               // tests: provoke the record-bound guard
    if (shrink) { cap1 /= 2; cap2 /= 2; }
    const uint32_t ns1 = np - nc1 - nd1, ns2 = paired ? np - nc2 - nd2 : 0u;
    uint32_t gs = cdiv(std::max(ns1, ns2), RB), gd = cdiv(std::max(nd1, paired ? nd2 : 0u), RB);
    const uint32_t gc = cdiv(std::max(nc1, paired ? nc2 : 0u), RB);
    if (tb.L > 1008) { gs = 0; gd = 0; }                                           // reads this long all sit in the general list (launch_indels); what is left in the others are holes: nothing to write
    // The three class kernels write disjoint records: the two small ones go to side streams and run BESIDE the big one (each alone
    // leaves the chip half empty through its first and last wave of workgroups); the caller's stream waits for both.
    static const bool split_env = seam_env("SCS_READS_SPLIT") != nullptr, serial_env = seam_env("SCS_READS_SERIAL") != nullptr;
    if (!split_env && !serial_env) {
        // ONE launch for the three classes (k_reads_all); its LDS is the larger of the uniform walk's and the general variant's
        ReadLists rl{};
        rl.l[0][0] = clist1; rl.l[0][1] = clist2; rl.n[0][0] = nc1; rl.n[0][1] = paired ? nc2 : 0u; rl.grid[0] = paired ? 2 * gc : gc;
        rl.l[1][0] = dlist1; rl.l[1][1] = dlist2; rl.n[1][0] = nd1; rl.n[1][1] = paired ? nd2 : 0u; rl.grid[1] = paired ? 2 * gd : gd;
        rl.l[2][0] = slist1; rl.l[2][1] = slist2; rl.n[2][0] = ns1; rl.n[2][1] = ns2; rl.grid[2] = paired ? 2 * gs : gs;
        const uint32_t grid = rl.grid[0] + rl.grid[1] + rl.grid[2];
        if (!grid) return;
        const size_t lds = std::max(reads_lds_bytes(tb, true), reads_lds_bytes(tb, false));
        static size_t opted_all[64][3] = {};
        int dev = 0; (void)hipGetDevice(&dev);
        size_t* opted = opted_all[dev & 63];
        const uint32_t cap = (uint32_t)(paired ? 2ull * np : np);
#define SCS_LAUNCH_ALL(QKV, SLOT) do { \
            if (opted[SLOT] != lds) { note_launch(hipFuncSetAttribute((const void*)k_reads_all<QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); opted[SLOT] = lds; } \
            hipLaunchKernelGGL((k_reads_all<QKV>), dim3(grid), dim3(RB), lds, s, g, reinterpret_cast<const uint8_t*>(g2), spool, fpool, pairs, np, paired, tb, key, slot, cap, reads_force_replay(), \
                               ev_hdr, ev_dat, off1, off2, out1, out2, amp_index_base, flags, cap1, cap2, rl); } while (0)
        if (tb.qual_k == 16) SCS_LAUNCH_ALL(16, 0); else if (tb.qual_k == 64) SCS_LAUNCH_ALL(64, 1); else SCS_LAUNCH_ALL(128, 2);
#undef SCS_LAUNCH_ALL
        return;
    }
    // SCS_READS_SPLIT: the three classes as three launches on three streams (round 2's form); SCS_READS_SERIAL: one after the other
    // (side: the caller's two side streams and fork / join events -- they belong to its ctx, created on first use, destroyed with it)
    static const bool dummy_serial_env = false; (void)dummy_serial_env;
    const bool serial = serial_env || !side;
    ReadsSide none; ReadsSide& sd = side ? *side : none;
    if (!serial && !sd.fork) {
        note_launch(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) { note_launch(hipStreamCreateWithFlags(&sd.st[k], hipStreamNonBlocking)); note_launch(hipEventCreateWithFlags(&sd.join[k], hipEventDisableTiming)); }
    }
    hipStream_t s_main = s;
    hipStream_t s_c = serial ? s : sd.st[0], s_d = serial ? s : sd.st[1];
    if (!serial && (gc || gd)) { note_launch(hipEventRecord(sd.fork, s_main)); if (gc) note_launch(hipStreamWaitEvent(s_c, sd.fork, 0)); if (gd) note_launch(hipStreamWaitEvent(s_d, sd.fork, 0)); }
    if (gd) launch_reads_kernel<true, 3>(s_d, dim3(paired ? 2 * gd : gd), tb, g, spool, fpool, pairs, np, paired,
                              reinterpret_cast<const uint8_t*>(g2), (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint8_t*)nullptr, 0u, tb, key, slot,
                              (uint32_t)(paired ? 2ull * np : np), reads_force_replay(), ev_hdr, ev_dat, off1, off2, out1, out2, amp_index_base,
                              (char*)nullptr, (char*)nullptr, (uint32_t*)nullptr, flags, cap1, cap2, dlist1, dlist2, nd1, paired ? nd2 : 0u);
    if (gs) launch_reads_kernel<true, 1>(s, dim3(paired ? 2 * gs : gs), tb, g, spool, fpool, pairs, np, paired,
                              reinterpret_cast<const uint8_t*>(g2), (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint8_t*)nullptr, 0u, tb, key, slot,
                              (uint32_t)(paired ? 2ull * np : np), reads_force_replay(), ev_hdr, ev_dat, off1, off2, out1, out2, amp_index_base,
                              (char*)nullptr, (char*)nullptr, (uint32_t*)nullptr, flags, cap1, cap2, slist1, slist2, ns1, ns2);
    if (gc) launch_reads_kernel<true, 2>(s_c, dim3(paired ? 2 * gc : gc), tb, g, spool, fpool, pairs, np, paired,
                              (const uint8_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint8_t*)nullptr, 0u, tb, key, slot,
                              (uint32_t)(paired ? 2ull * np : np), reads_force_replay(), ev_hdr, ev_dat, off1, off2, out1, out2, amp_index_base,
                              (char*)nullptr, (char*)nullptr, (uint32_t*)nullptr, flags, cap1, cap2, clist1, clist2, nc1, paired ? nc2 : 0u);
    if (!serial) {
        if (gc) { note_launch(hipEventRecord(sd.join[0], s_c)); note_launch(hipStreamWaitEvent(s_main, sd.join[0], 0)); }
        if (gd) { note_launch(hipEventRecord(sd.join[1], s_d)); note_launch(hipStreamWaitEvent(s_main, sd.join[1], 0)); }
    }
}
// the batch's reads split by class (k_indels' flags cls, their exclusive scans cpos): ascending lists of pair indices
// (general class = bit 31 of the record size as k_indels left it, its list position = the scanned offset's bits above OFF_BITS; the
// one-event class = its flag array and that array's scan; the event-free class takes what is left)
__global__ void k_read_lists(uint32_t np, int paired, const uint32_t* __restrict__ sizes1, const uint64_t* __restrict__ off1, const uint32_t* __restrict__ d1f1,
                             const uint32_t* __restrict__ d1p1, const uint32_t* __restrict__ sizes2, const uint64_t* __restrict__ off2, const uint32_t* __restrict__ d1f2,
                             const uint32_t* __restrict__ d1p2, uint32_t* __restrict__ slist1, uint32_t* __restrict__ slist2, uint32_t* __restrict__ clist1,
                             uint32_t* __restrict__ clist2, uint32_t* __restrict__ dlist1, uint32_t* __restrict__ dlist2) {
    const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= np) return;
    { const uint32_t c = (uint32_t)(off1[pi] >> OFF_BITS), d = d1p1[pi]; if (sizes1[pi] >> 31) clist1[c] = pi; else if (d1f1[pi]) dlist1[d] = pi; else slist1[pi - c - d] = pi; }
    if (paired) { const uint32_t c = (uint32_t)(off2[pi] >> OFF_BITS), d = d1p2[pi]; if (sizes2[pi] >> 31) clist2[c] = pi; else if (d1f2[pi]) dlist2[d] = pi; else slist2[pi - c - d] = pi; }
}
void launch_read_lists(hipStream_t s, uint32_t np, int paired, const uint32_t* sizes1, const uint64_t* off1, const uint32_t* d1f1, uint32_t* d1p1,
                       const uint32_t* sizes2, const uint64_t* off2, const uint32_t* d1f2, uint32_t* d1p2,
                       uint32_t* slist1, uint32_t* slist2, uint32_t* clist1, uint32_t* clist2, uint32_t* dlist1, uint32_t* dlist2, void* temp, size_t temp_bytes) {
    if (np == 0) return;
    exclusive_scan_u32(s, d1f1, d1p1, np, temp, temp_bytes);
    if (paired) exclusive_scan_u32(s, d1f2, d1p2, np, temp, temp_bytes);
    hipLaunchKernelGGL(k_read_lists, dim3(cdiv(np, 256)), dim3(256), 0, s, np, paired, sizes1, off1, d1f1, d1p1, sizes2, off2, d1f2, d1p2, slist1, slist2, clist1, clist2, dlist1, dlist2);
}
void launch_predict_windows(hipStream_t s, const uint8_t* windows, uint32_t n_reads, const uint64_t* uids, const uint32_t* atts,
                            const uint8_t* is_read1, DevTables tb, const DevTables* d_tb, RngKey key, uint32_t slot, char* slot_b, char* slot_q, uint32_t* lens, uint32_t* flags) {
    if (n_reads == 0) return;
    (void)d_tb;
    DevErrPool none{};
    launch_reads_kernel<false, 0>(s, dim3(cdiv(n_reads, RB)), tb, (const uint8_t*)nullptr, none, none, (const PairRec*)nullptr, 0u, 0,
                               windows, uids, atts, is_read1, n_reads, tb, key, slot, n_reads, reads_force_replay(), (const uint32_t*)nullptr, (const uint4*)nullptr,
                               (const uint64_t*)nullptr, (const uint64_t*)nullptr, (char*)nullptr, (char*)nullptr, 0u, slot_b, slot_q, lens, flags, (uint64_t)0, (uint64_t)0,
                               (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, 0u);
}
// Checksum of a batch's FASTQ text where it lies in HBM (scs_set_batch_checksums): the text as little-endian 64-bit words w_i
// (the last one zero-padded), sum over i of fmix64(w_i + (i + 1) * 0x9E3779B97F4A7C15) mod 2^64 -- every word's position is mixed
// into its term, the sum is commutative, so the result does not depend on the order the waves' partial sums arrive in.
// HBM-bound: one read of the text, 16 bytes per lane and step.
__device__ __forceinline__ unsigned long long fmix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33; return x;
}
__global__ void __launch_bounds__(256) k_text_checksum(const unsigned long long* __restrict__ text, unsigned long long nbytes, unsigned long long* __restrict__ out) {
    const unsigned long long nw = nbytes >> 3, rem = nbytes & 7ull, stride = (unsigned long long)gridDim.x * blockDim.x * 2ull;
    unsigned long long acc = 0;
    for (unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 2ull; i < nw; i += stride) {
        if (i + 1 < nw) { const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(text + i); acc += fmix64(v.x + (i + 1) * 0x9E3779B97F4A7C15ull) + fmix64(v.y + (i + 2) * 0x9E3779B97F4A7C15ull); }
        else acc += fmix64(text[i] + (i + 1) * 0x9E3779B97F4A7C15ull);
    }
    if (rem && blockIdx.x == 0 && threadIdx.x == 0) {                               // the last, partial word
        const unsigned char* t = reinterpret_cast<const unsigned char*>(text + nw); unsigned long long w = 0;
        for (unsigned long long b = 0; b < rem; ++b) w |= (unsigned long long)t[b] << (8 * b);
        acc += fmix64(w + (nw + 1) * 0x9E3779B97F4A7C15ull);
    }
    acc = wave_sum_u64(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
void launch_text_checksum(hipStream_t s, const char* text, uint64_t nbytes, unsigned long long* out) {
    (void)hipMemsetAsync(out, 0, 8, s);
    if (nbytes == 0) return;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(4096, (nbytes / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_text_checksum, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const unsigned long long*>(text), (unsigned long long)nbytes, out);
}
void phase_clock_report_attach();
void phase_clock_report() {
#ifdef SCS_PHASE_CLOCK
    unsigned long long h[16] = {}, z[16] = {};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) != hipSuccess || !h[15]) return;
    static const char* nm[8] = {"lists, records, sort by sector phase", "window gather + error patch", "event words", "seed, name line, ring fill", "the walk", "tails", "deferred positions", "redo"};
    fprintf(stderr, "[phase clock] %llu workgroups of the uniform walk; mean time of thread 0 per phase (us):", h[15]);
    double tot = 0; for (int i = 0; i < 8; ++i) tot += (double)h[i];
    for (int i = 0; i < 8; ++i) fprintf(stderr, "  %s %.2f", nm[i], (double)h[i] / (double)h[15] / 100.0);
    fprintf(stderr, "  | total %.2f | shader clock over the workgroups' lives %.3f GHz\n", tot / (double)h[15] / 100.0, h[13] ? (double)h[12] / (double)h[13] * 0.1 : 0.0);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z);
    phase_clock_report_attach();
#endif
}
}  // namespace scs
