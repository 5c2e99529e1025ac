// scs_common.h -- definitions shared by host code and gfx950 kernels.
//
// Counter-RNG remapping (DESIGN.md "RNG remapping"): every random draw of the
// reference's hot path is keyed by logical ids instead of being pulled from a
// per-thread sequential stream (reference: lib/threadpool/ThreadPool.cpp:41-47,
// 203-212).  One Philox4x32-10 block = 4 draws:
//     counter = (idx, uid_lo, uid_hi, stage | aux << 8)     key = (seed_lo, seed_hi)
// The stage list and the meaning of idx/aux/word per draw site are below.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SCS_HD __host__ __device__ __forceinline__
#else
#define SCS_HD inline
#endif

namespace scs {

enum Stage : uint32_t {
    ST_FRAGSPLIT   = 1,   // uid = record index, idx = k-th fragment of the record, word 0     (Genome.cpp:760)
    ST_POISSON     = 2,   // uid = template uid, aux = kind | call<<1, draw t -> idx t/4 word t%4 (MyDefine.cpp:69-80)
    ST_ATTACH      = 3,   // uid = template uid, aux = kind | pass<<1, idx = primer: seeds the primer's xoshiro128++ try stream (spos, length per try)
    ST_ERR         = 4,   // uid = NEW amplicon uid, aux = kind; block 0: words 0,1 = 64-bit draw for the error COUNT
                          //   (binomial thresholds); blocks 1.. : candidate error positions, one word each   [REMAP of Fragment.cpp:100-104]
    ST_ERRALT      = 5,   // uid = NEW amplicon uid, aux = kind, idx = j | (a/4)<<16, word a%4  (Fragment.cpp:107-110)
    ST_WEIGHT      = 6,   // uid = full uid, idx = attempt, words 0,1                           (Profile.cpp:1503-1513)
    ST_ALLOC_TOP   = 7,   // uid = 0, idx = t, word 0                                           (MyDefine.cpp:242-245)
    ST_ALLOC_CHUNK = 8,   // uid = chunk, idx = t >> 2, word t & 3                                      (MyDefine.cpp:191-201)
    ST_PAIR        = 9,   // uid = full uid, idx = attempt, word0 insert size, word1 position   (Amplicon.cpp:483-491)
    ST_READ        = 10,  // uid = full uid, aux = rd | attempt<<1: block 0 seeds the read's xoshiro128++ stream A (insertion /
                          //   deletion tests, Profile.cpp:1556-1566), block 1 stream B (substitution, quality, random quality of
                          //   an N, Profile.cpp:1666-1692); both are consumed in the reference's order
    ST_INDEL_INS   = 11,  // retired: inserted bases are drawn from the read's stream B when emitted  (Profile.cpp:1560)
    ST_INDEL_LEN   = 12   // same aux, idx = j, word0 : insertion / deletion length             (Profile.cpp:1515-1521)
};

SCS_HD uint32_t stage_word(uint32_t stage, uint32_t aux) { return stage | (aux << 8); }

// lineage uids: order-free and shard-free
SCS_HD uint64_t semi_uid(uint64_t frag, uint32_t pass, uint32_t i) { return (frag << 23) | ((uint64_t)pass << 20) | i; }
SCS_HD uint64_t full_uid(uint64_t semi, uint32_t cyc, uint32_t i) { return (semi << 15) | ((uint64_t)cyc << 12) | i; }

struct U4 { uint32_t w[4]; };

SCS_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11)
SCS_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    U4 o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

struct RngKey { uint32_t k0, k1; };

// xoshiro128++ (Blackman & Vigna, public domain): per-read sequential streams seeded from Philox blocks.  All
// full-rate 32-bit ops (no multiplies), ~11 instructions per draw.
struct Xoshiro {
    uint32_t s0, s1, s2, s3;
    SCS_HD void seed(U4 w) { s0 = w.w[0]; s1 = w.w[1]; s2 = w.w[2]; s3 = w.w[3]; if (!(s0 | s1 | s2 | s3)) s0 = 1; }
    SCS_HD uint32_t next() {
        const uint32_t a = s0 + s3;
        const uint32_t result = ((a << 7) | (a >> 25)) + s0;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t; s3 = (s3 << 11) | (s3 >> 21);
        return result;
    }
    // [REMAP] ONE step, TWO words: a = the xoshiro128++ output (scrambler on s0, s3), b = the same scrambler on the other two
    // state words (s1, s2).  Stream B of a read (substitution / quality draws) advances one step per output position.
    // (Round 3 tried the "+" scrambler here -- s0 + s3, s1 + s2: 2 instead of 6 of the step's 13 instructions -- and measured
    // -1.5 % on k_reads in an alternating A/B: not worth a generator with weak low bits; "++" stays.)
    SCS_HD void next2(uint32_t& a, uint32_t& b) {
        const uint32_t p = s0 + s3, q = s1 + s2;
        a = ((p << 7) | (p >> 25)) + s0; b = ((q << 7) | (q >> 25)) + s1;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t; s3 = (s3 << 11) | (s3 >> 21);
    }
};

SCS_HD U4 draw4(RngKey key, uint32_t stage, uint32_t aux, uint64_t uid, uint32_t idx) {
    return philox4x32_10(idx, (uint32_t)uid, (uint32_t)(uid >> 32), stage_word(stage, aux), key.k0, key.k1);
}

// randomInteger(s,e) / truncating randomDouble(s,e) of ThreadPool.cpp:203-212 for a 32-bit draw x:
// trunc(s + (e-s) * x/2^32).  For (e-s) < 2^21 the double expression is exact, so this integer form
// returns the same value as the reference's double arithmetic.
SCS_HD uint32_t scale_draw(uint32_t x, uint32_t start, uint32_t span) {
    return start + (uint32_t)(((uint64_t)span * x) >> 32);
}

// [REMAP] attach tries (Fragment.cpp:73-95, Amplicon.cpp:176-198).  A try draws spos uniformly from [27, len) and the
// amplicon length from [amin, amax] and fails outright when spos + alen > len; those tries are i.i.d., so the number of
// them before a FITTING try is geometric and the fitting try is uniform over the feasible pairs.  The feasible set of a
// template of length len: the spos <= len - amax take every length (A positions x W lengths), then a triangle in which
// the position len - amin + 1 - d admits d lengths (d = 1 .. D).  attach_fit_count = its size N (< 2^32 for len < 2^21);
// attach_fit_decode maps r in [0, N) to the pair.
struct AttachFit { uint32_t A, W, D, N; };
SCS_HD AttachFit attach_fit_count(uint32_t len, uint32_t amin, uint32_t amax) {
    AttachFit f; f.W = amax - amin + 1u;
    f.A = len >= amax + 27u ? len - amax - 26u : 0u;                              // spos = 27 .. len - amax
    const uint32_t d1 = len - amin - 26u;                                         // positions 27 .. len - amin in all (len >= amin + 27)
    f.D = d1 - f.A;                                                               // the rest of them form the triangle: D <= W - 1
    f.N = f.A * f.W + f.D * (f.D + 1u) / 2u;
    return f;
}
SCS_HD void attach_fit_decode(const AttachFit& f, uint32_t len, uint32_t amin, uint32_t r, uint32_t& spos, uint32_t& alen) {
    const uint32_t full = f.A * f.W;
    if (r < full) { spos = 27u + r / f.W; alen = amin + r % f.W; return; }
    const uint32_t q = r - full;                                                  // q in [0, D(D+1)/2): row d holds d entries, rows 1, 2, .. D
    uint32_t d = (uint32_t)((1.0 + sqrt(1.0 + 8.0 * (double)q)) * 0.5);
    while (d > 1u && d * (d - 1u) / 2u > q) --d;                                  // exact whatever the rounding of the square root
    while (d * (d + 1u) / 2u <= q) ++d;
    alen = amin + (q - d * (d - 1u) / 2u);
    spos = len - amin + 1u - d;
}
// (Round 3 tried r / W as a multiplication by ceil(2^42 / W) and a single-precision estimate of the triangle row: k_attach<semi>
// unchanged, k_attach<frag> 21.1 -> 24.0 ms -- the kernel waits for its chain of dependent gathers per candidate (position bitmap,
// the 8 primer bases, the stock), not for these instructions.)
// number of non-fitting tries before the next fitting one, capped at 51 (> 50 tries kill the primer): u uniform in (0, 1),
// qfail = 1 - N / M.  P(gap >= g) = qfail^g, evaluated by repeated multiplication (the same IEEE products everywhere).
SCS_HD uint32_t attach_gap(double u, double qfail) {
    double acc = qfail; uint32_t g = 0;
    while (g < 51u && u < acc) { acc = acc * qfail; ++g; }
    return g;
}
// (Round 3 tried the gap by the binary digits of g -- q^32 .. q by repeated squaring, six compare-and-multiply steps, no loop:
// k_attach<semi> 58.9 -> 62.9 ms.  The loop's mean of 11 cheap rounds beats eleven fixed products with six double selects.)

// base codes: 0..3 = ACGT, 4 = N / anything else (MyDefine.cpp:352-367: complement of non-ACGT is 'N')
SCS_HD uint8_t comp_code(uint8_t c) { return c < 4 ? (uint8_t)(3 - c) : (uint8_t)4; }
SCS_HD bool is_gc(uint32_t c) { return c == 1 || c == 2; }

// packed amplicon record (reference: Amplicon::data[8], lib/amplicon/Amplicon.cpp:47-154)
//   pk0 = spos (17 bits) | len (11 bits) << 17          pk1 = gc (11 bits) | primers (12 bits) << 11
SCS_HD uint32_t pack_sl(uint32_t spos, uint32_t len) { return spos | (len << 17); }
SCS_HD uint32_t sl_spos(uint32_t p) { return p & 0x1FFFFu; }
SCS_HD uint32_t sl_len(uint32_t p) { return p >> 17; }

// amplification error entry (reference: AmpError, Amplicon.cpp:13-45): u16 = pos (11 bits) | alt (2 bits) << 11, 0 = empty
// an amplicon keeps up to 4 inline in a uint64; bit 63 set = overflow: bits 0..31 pool offset, 32..47 count
SCS_HD uint32_t err_pack(uint32_t pos, uint32_t alt) { return pos | (alt << 11); }
SCS_HD uint32_t err_pos(uint32_t e) { return e & 0x7FFu; }
SCS_HD uint32_t err_alt(uint32_t e) { return (e >> 11) & 3u; }
static const uint64_t ERR_OVERFLOW_BIT = 1ull << 63;

static const int BINOM_KMAX = 16;  // error-count thresholds per window length ([REMAP], scs_tables.cpp binom_table)
static const int NQ = 94;          // Phred chars 33..126 (Profile.cpp:172-173)
static const int NKMER = 84;       // 4 + 16 + 64 (Profile.cpp:69-123)

// k-mer row of the substitution table for the context (a, b, c); 5 = 'X' (before read start), 4 = N.
// -1 = not in the table -> base kept (Profile.cpp:1523-1530)
SCS_HD int kmer_index(uint32_t a, uint32_t b, uint32_t c) {
    if (c > 3) return -1;
    if (a == 5 && b == 5) return (int)c;
    if (a == 5 && b < 4) return (int)(4 + b * 4 + c);
    if (a < 4 && b < 4) return (int)(20 + a * 16 + b * 4 + c);
    return -1;
}

}  // namespace scs
