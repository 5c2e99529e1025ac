// scs_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the genreads hot path.
// Integer / byte work bounded by HBM and the per-lane Philox rate; no MFMA by design.
// Built with -ffp-contract=off: the few fp64 expressions must round exactly like the CPU oracle.
#include <utility>
#include <type_traits>
#include "scs_device.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace scs {
// Build-time diagnostic (-DSCS_PHASE_CLOCK): where a workgroup of the uniform walk spends its life.  Thread 0 reads the 100 MHz
// wall clock at the phase boundaries and adds the differences to g_phase[] (g_phase[15] counts the workgroups); the host prints and
// zeroes them (scs_phase_clock_report, called at the end of a yield when SCS_PHASE_CLOCK is set in the environment).
#ifdef SCS_PHASE_CLOCK
__device__ unsigned long long g_phase[16];
#define SCS_PHASE(i) do { if (UNI && CLS == 1 && tid == 0) { const unsigned long long now_ = wall_clock64(); if ((i) >= 0) atomicAdd(&g_phase[(i) < 0 ? 0 : (i)], now_ - ph_t_); ph_t_ = now_; } } while (0)
// the same for k_attach<semi>: wave-level sections of its loop (g_phase_att[15] counts the waves)
__device__ unsigned long long g_phase_att[8 * 256];                                // [section][workgroup & 255]: the waves' sums, spread over 256 slots
#define SCS_ATT(i) do { if (!FROM_FRAG) { const unsigned long long now_ = wall_clock64(); if (lane == __ffsll((long long)__ballot(1)) - 1) { s_att_acc[i] += now_ - s_att_t; s_att_t = wall_clock64(); } } } while (0)   /* (one wave per workgroup: mark and sums live in LDS, whichever lanes are active) */
#else
#define SCS_PHASE(i) do {} while (0)
#define SCS_ATT(i) do {} while (0)
#endif

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { uint32_t t = __shfl_up(v, d); if (lane >= d) v += t; }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ------------------------------------------------------------------------------------------------
// views (reference Amplicon::getSequence, lib/amplicon/Amplicon.cpp:255-382)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ View frag_view(uint64_t goff, uint32_t len, int strand) {
    View v;
    if (strand > 0) { v.base = (int64_t)goff + len - 1; v.dir = -1; v.comp = 1; }
    else { v.base = (int64_t)goff; v.dir = 1; v.comp = 0; }
    return v;
}
// template strand c(S) of a semi amplicon (s, l) made on a template with view fv
__device__ __forceinline__ View semi_tmpl_view(View fv, uint32_t s, uint32_t l) {
    View v; v.base = fv.base + (int64_t)fv.dir * (int64_t)(s + l - 1); v.dir = -fv.dir; v.comp = fv.comp ^ 1u; return v;
}
__device__ __forceinline__ View shift_view(View v, uint32_t s) { v.base += (int64_t)v.dir * (int64_t)s; return v; }
__device__ __forceinline__ uint32_t view_base(const uint8_t* __restrict__ g, View v, uint32_t i) {
    uint32_t c = g[v.base + (int64_t)v.dir * (int64_t)i];
    return v.comp ? (uint32_t)comp_code((uint8_t)c) : c;
}

// bases i .. i+7 of a view in one 8-byte load: byte k = base i+k (codes 0-3 ACGT, 4 = N)
__device__ __forceinline__ unsigned long long view_bases8(const uint8_t* __restrict__ g, View v, uint32_t i) {
    const int64_t a = v.base + (int64_t)v.dir * (int64_t)i;
    unsigned long long x;
    __builtin_memcpy(&x, g + (v.dir > 0 ? a : a - 7), 8);
    if (v.dir < 0) x = __builtin_bswap64(x);
    if (v.comp) x ^= 0x0303030303030303ull & ~(((x >> 2) & 0x0101010101010101ull) * 3ull);   // 3 - c for ACGT, N stays
    return x;
}

// iterate the error entries of an amplicon (inline u16 x4, or overflow list)
template <class F>
__device__ __forceinline__ void for_each_err(uint64_t e, const uint32_t* __restrict__ pool, F f) {
    if (e == 0) return;
    if (e & ERR_OVERFLOW_BIT) {
        const uint32_t off = (uint32_t)e, cnt = (uint32_t)(e >> 32) & 0xFFFFu;
        for (uint32_t i = 0; i < cnt; ++i) f(pool[off + i]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint32_t v = (uint32_t)(e >> (16 * k)) & 0xFFFFu; if (v) f(v); }
    }
}

// base t of the template strand c(S) of a semi: view + the semi's own substitutions
//   S'[j] = alt  =>  c(S)[l-1-j] = comp(alt)
__device__ __forceinline__ uint32_t semi_tmpl_base(const uint8_t* __restrict__ g, View stv, uint32_t l, uint64_t errs,
                                                   const uint32_t* __restrict__ pool, uint32_t t) {
    uint32_t c = view_base(g, stv, t);
    for_each_err(errs, pool, [&](uint32_t e) { if (l - 1 - err_pos(e) == t) c = 3u - err_alt(e); });
    return c;
}

// ------------------------------------------------------------------------------------------------
// table lookups: randIndx(cdf, ac) (lib/mydefine/MyDefine.cpp:274-282) on integer thresholds
// ------------------------------------------------------------------------------------------------
__device__ __noinline__ uint32_t rand_indx_slow(const double* __restrict__ cdf, uint32_t ac, uint32_t x) {
    const double r = 2.2204e-16 + (1 - 2.2204e-16) * ((double)x / 4294967296.0);
    for (uint32_t k = 0; k < ac; ++k) if (r <= cdf[k]) return k;
    return ac - 1;
}
// first k with x < T[k], else ac-1 (T non-decreasing)
__device__ __forceinline__ uint32_t rand_indx_thr(const uint32_t* __restrict__ T, const double* __restrict__ cdf, uint32_t ac, uint32_t x) {
    if (x == 0xFFFFFFFFu) return rand_indx_slow(cdf, ac, x);
    uint32_t lo = 0, hi = ac;                 // lower bound of "x < T[k]"
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (x < T[mid]) hi = mid; else lo = mid + 1; }
    return lo < ac ? lo : ac - 1;
}

// start of Malbac::amplify: createPrimers (4^8 primer types x `copies`, Malbac.cpp:204-234) + the run's device scalars
__global__ void k_amplify_init(int64_t* __restrict__ cnt, unsigned long long* __restrict__ cut, int64_t copies, uint32_t* __restrict__ delta, uint32_t* __restrict__ gdelta, uint32_t* __restrict__ flags,
                               unsigned long long* __restrict__ sums, unsigned long long nf_all, unsigned long long frag_len_all, unsigned long long total_primers,
                               uint32_t* __restrict__ pool_head_a, uint32_t* __restrict__ pool_head_b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536) { cnt[i] = copies; cut[i] = copies > 0 ? ~0ull : 0ull; delta[i] = 0; if (gdelta) gdelta[i] = 0; }
    if (i < SHARD_TAIL_WORDS && gdelta) gdelta[65536 + i] = 0;
    if (i < 16) sums[i] = i == DS_G_PRIMERS ? total_primers : i == DS_G_TOTALS || i == DS_G_NF ? nf_all : i == DS_G_TOTALS + 1 || i == DS_G_FRAG_LEN ? frag_len_all : i == DS_MIN_STOCK ? (unsigned long long)copies : 0ull;
    if (i == 0) { flags[0] = 0; if (pool_head_a) *pool_head_a = 0; if (pool_head_b) *pool_head_b = 0; }   // error overflow pools of the two amplicon stores
}
// end of a pass (every lane of the wave calls it): the stock loses what the pass took -- never more than there was (k_attach's
// cuts, exact_stock) --, the next pass's cuts (all of it / none of it), the smallest stock left (the host skips the
// over-demand check of a pass that cannot reach it)
__device__ __forceinline__ void stock_update(uint32_t i, int64_t* __restrict__ cnt, const uint32_t* __restrict__ taken, uint32_t* __restrict__ delta, unsigned long long* __restrict__ cut,
                                             unsigned long long* __restrict__ sums, uint32_t* __restrict__ flags) {
    unsigned long long left = ~0ull;
    if (i < 65536u) {
        int64_t c = cnt[i] - (int64_t)taken[i];
        if (c < 0) { atomicOr(flags, (uint32_t)FLAG_INTERNAL); c = 0; }
        cnt[i] = c; cut[i] = c > 0 ? ~0ull : 0ull; delta[i] = 0;
        if (c > 0) left = (unsigned long long)c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(left >> 32), d) << 32) | (uint32_t)__shfl_xor((int)left, d);
        left = o < left ? o : left;
    }
    if ((threadIdx.x & 63u) == 0 && left != ~0ull) atomicMin(&sums[DS_MIN_STOCK], left);
}
// sharded job.  What the shards owe each other besides what they took from the stock -- the semi amplicons a fragment pass
// made (count, total length) and the budgets the last setPrimers handed out -- rides on the pass's closing all-reduce, as
// 24-bit limbs in 32-bit words behind the 65536 counters: k_shard_tail writes this shard's share before the collective ...
__global__ void k_shard_tail(uint32_t* __restrict__ gdelta, const unsigned long long* __restrict__ sums, const uint32_t* __restrict__ new_semis, int with_budgets) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t* t = gdelta + 65536;
    const unsigned long long len = sums[DS_SEMI_LEN] - sums[DS_REPORTED_LEN];
    t[0] = new_semis ? *new_semis : 0u;
    t[1] = (uint32_t)(len & 0xFFFFFFull); t[2] = (uint32_t)((len >> 24) & 0xFFFFFFull); t[3] = (uint32_t)(len >> 48);
    for (int k = 0; k < 2; ++k) {
        const unsigned long long b = with_budgets ? sums[k] : 0ull;
        t[4 + 3 * k] = (uint32_t)(b & 0xFFFFFFull); t[5 + 3 * k] = (uint32_t)((b >> 24) & 0xFFFFFFull); t[6 + 3 * k] = (uint32_t)(b >> 48);
    }
}
// ... and the stock update after it folds the summed tail into the whole-job scalars that setPrimers reads
__global__ void k_primer_update_sharded(int64_t* __restrict__ cnt, uint32_t* __restrict__ gdelta, uint32_t* __restrict__ delta, unsigned long long* __restrict__ cut,
                                        unsigned long long* __restrict__ sums, uint32_t* __restrict__ flags, int with_budgets) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    stock_update(i, cnt, gdelta, delta, cut, sums, flags);
    if (i < 65536u) gdelta[i] = 0;
    if (i == 0) {
        uint32_t* t = gdelta + 65536;
        auto limbs = [&](int o) { return (unsigned long long)t[o] + ((unsigned long long)t[o + 1] << 24) + ((unsigned long long)t[o + 2] << 48); };
        sums[DS_G_SEMIS_N] += t[0]; sums[DS_G_SEMI_LEN] += limbs(1);
        sums[DS_G_PRIMERS] -= limbs(4) + limbs(7);
        sums[DS_G_TOTALS] = sums[DS_G_NF] + sums[DS_G_SEMIS_N]; sums[DS_G_TOTALS + 1] = sums[DS_G_FRAG_LEN] + sums[DS_G_SEMI_LEN];
        sums[DS_REPORTED_LEN] = sums[DS_SEMI_LEN];
        if (with_budgets) { sums[0] = 0; sums[1] = 0; }
        for (int k = 0; k < SHARD_TAIL_WORDS; ++k) t[k] = 0;
    }
}
__global__ void k_primer_update(int64_t* __restrict__ cnt, uint32_t* __restrict__ delta, unsigned long long* __restrict__ cut, unsigned long long* __restrict__ sums, uint32_t* __restrict__ flags) {
    stock_update(blockIdx.x * blockDim.x + threadIdx.x, cnt, delta, delta, cut, sums, flags);
}
// sharded job, a pass run again segment by segment (exact_stock): the stock after the segment its owner has just finished
// (gdelta: what the owner took, summed over the shards: everybody else sent zeros)
__global__ void k_stock_apply(int64_t* __restrict__ cnt, uint32_t* __restrict__ gdelta, uint32_t* __restrict__ delta, unsigned long long* __restrict__ cut, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    int64_t c = cnt[i] - (int64_t)gdelta[i];
    if (c < 0) { atomicOr(flags, (uint32_t)FLAG_INTERNAL); c = 0; }
    cnt[i] = c; cut[i] = c > 0 ? ~0ull : 0ull; delta[i] = 0; gdelta[i] = 0;
}

// ASCII -> base code, in place (0..3 = ACGT either case, 4 = anything else): Genome::getSubSequence's toupper
// (lib/genome/Genome.cpp:272-278) + getIndexOfBase (lib/mydefine/MyDefine.cpp:326-334).  16 bytes per thread.
__global__ void k_encode_bases(uint8_t* __restrict__ g, uint64_t n) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= n) return;
    auto code = [](uint32_t c) -> uint32_t { c &= 0xDFu; return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; };
    if (i + 16 <= n && ((uintptr_t)(g + i) & 15) == 0) {
        uint4 v = *reinterpret_cast<uint4*>(g + i); uint32_t* w = &v.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t x = w[k]; w[k] = code(x & 255u) | (code((x >> 8) & 255u) << 8) | (code((x >> 16) & 255u) << 16) | (code(x >> 24) << 24); }
        *reinterpret_cast<uint4*>(g + i) = v;
    } else for (uint64_t k = i; k < n && k < i + 16; ++k) g[k] = (uint8_t)code(g[k]);
}
// ------------------------------------------------------------------------------------------------
// FASTA parsed on the device (lib/fastahack/Fasta.cpp:45-215 index + 304-334 getSubSequence, Genome.cpp:176-195):
// the file's raw bytes arrive in chunks; a byte is a base iff its LINE is a sequence line (not a '>' header, not a ';'
// comment) and it is neither '\n' nor '\r'.  The line's kind is the kind of its first byte carried forward: an inclusive scan
// with "the right operand wins if it starts a line" (kinds 1 header, 2 comment, 3 sequence; 0 = not a line start).  Kept
// bytes are compacted behind the bases of the earlier chunks; headers (rare) are listed with their file offset and the
// number of bases before them, from which the host takes the names and the record lengths.
// ------------------------------------------------------------------------------------------------
struct FaKindOp { __host__ __device__ uint8_t operator()(uint8_t a, uint8_t b) const { return b ? b : a; } };
// st: [0] bases so far, [1] headers so far, [2] kind of the line open at the chunk's start (0 = the chunk starts a line)
__global__ void __launch_bounds__(256) k_fa_kind(const uint8_t* __restrict__ raw, uint32_t n, const unsigned long long* __restrict__ st, uint8_t* __restrict__ kind) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = raw[i];
    const bool start = i == 0 ? st[2] == 0 : raw[i - 1] == '\n';
    uint8_t k = start ? (b == '>' ? 1 : b == ';' ? 2 : 3) : 0;
    if (i == 0 && !start) k = (uint8_t)st[2];                                      // the line continues from the previous chunk
    kind[i] = k;
}
__global__ void __launch_bounds__(256) k_fa_keep(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ kind, uint32_t n, uint32_t* __restrict__ keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { keep[i] = 0; return; }
    const uint32_t b = raw[i];
    keep[i] = (kind[i] == 3 && b != '\n' && b != '\r') ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_fa_scatter(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ kind, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                                                    uint32_t n, unsigned long long chunk_off, const unsigned long long* __restrict__ st, uint8_t* __restrict__ out,
                                                    unsigned long long* __restrict__ hdr, uint32_t hdr_cap, unsigned long long* __restrict__ nhdr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long base = st[0];
    if (keep[i]) out[base + pos[i]] = raw[i];
    const bool start = i == 0 ? st[2] == 0 : raw[i - 1] == '\n';
    if (start && raw[i] == '>') {                                                  // a header: file offset, bases before it
        const unsigned long long k = atomicAdd(nhdr, 1ull);
        if (k < hdr_cap) { hdr[2 * k] = chunk_off + i; hdr[2 * k + 1] = base + pos[i]; }
    }
}
// closes a chunk: bases so far += the chunk's, and the kind of the line left open at its end
__global__ void k_fa_close(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ kind, const uint32_t* __restrict__ pos, uint32_t n, unsigned long long* __restrict__ st) {
    if (threadIdx.x || blockIdx.x) return;
    st[0] += pos[n];
    st[2] = raw[n - 1] == '\n' ? 0ull : (unsigned long long)kind[n - 1];
}
size_t fasta_chunk_temp_bytes(uint32_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::inclusive_scan(nullptr, a, (const uint8_t*)nullptr, (uint8_t*)nullptr, (size_t)n, FaKindOp());
    (void)rocprim::exclusive_scan(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n + 1, rocprim::plus<uint32_t>());
    return (a > b ? a : b) + 256;
}
void launch_fasta_chunk(hipStream_t s, const uint8_t* raw, uint32_t n, unsigned long long chunk_off, unsigned long long* st, uint8_t* kind, uint32_t* keep, uint32_t* pos,
                        uint8_t* out, unsigned long long* hdr, uint32_t hdr_cap, void* temp, size_t temp_bytes) {
    if (n == 0) return;
    const unsigned g = (unsigned)((n + 256) / 256 + 1);
    hipLaunchKernelGGL(k_fa_kind, dim3(g), dim3(256), 0, s, raw, n, st, kind);
    (void)rocprim::inclusive_scan(temp, temp_bytes, kind, kind, (size_t)n, FaKindOp(), s);
    hipLaunchKernelGGL(k_fa_keep, dim3(g), dim3(256), 0, s, raw, kind, n, keep);
    (void)rocprim::exclusive_scan(temp, temp_bytes, keep, pos, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s);
    hipLaunchKernelGGL(k_fa_scatter, dim3(g), dim3(256), 0, s, raw, kind, keep, pos, n, chunk_off, st, out, hdr, hdr_cap, st + 1);
    hipLaunchKernelGGL(k_fa_close, dim3(1), dim3(64), 0, s, raw, kind, pos, n, st);
}

// ------------------------------------------------------------------------------------------------
// simuvars on the data plane: the haplotype sequences that Genome::saveSequence / generateSegment
// (lib/genome/Genome.cpp:329-691) assemble with std::string edits are materialised here from the host's plan: the output is
// a concatenation of pieces, each a range of the reference (resident in HBM, as read from the FASTA) or of the literal
// pool (inserted sequences), upper-cased (the toupper of Genome.cpp:393,684); SNP / SNV alleles are written afterwards.
// A thread produces 16 consecutive output bytes: binary search for its first piece, then a walk.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sv_build(const uint8_t* __restrict__ ref, const uint8_t* __restrict__ lit, const SvPiece* __restrict__ pieces, uint32_t np,
                                                  uint8_t* __restrict__ out, uint64_t total) {
    const uint64_t o = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (o >= total) return;
    uint32_t lo = 0, hi = np;                                                      // last piece with dst <= o
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pieces[mid].dst <= o) lo = mid; else hi = mid; }
    SvPiece pc = pieces[lo]; uint32_t pi = lo;
    uint32_t w[4] = {0, 0, 0, 0};
    const uint32_t nb = (uint32_t)min((uint64_t)16, total - o);
    for (uint32_t b = 0; b < nb; ++b) {
        const uint64_t x = o + b;
        while (x >= pc.dst + pc.len) pc = pieces[++pi];                            // pieces cover the output exactly: never runs past np
        uint32_t c = (pc.lit ? lit : ref)[pc.src + (x - pc.dst)];
        if (c >= 'a' && c <= 'z') c -= 32u;
        w[b >> 2] |= c << (8u * (b & 3u));
    }
    if (nb == 16) *reinterpret_cast<uint4*>(out + o) = make_uint4(w[0], w[1], w[2], w[3]);
    else for (uint32_t b = 0; b < nb; ++b) out[o + b] = (uint8_t)(w[b >> 2] >> (8u * (b & 3u)));
}
__global__ void k_sv_subst(const SvSubst* __restrict__ subs, uint32_t n, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[subs[i].dst] = (uint8_t)subs[i].ch;
}
void launch_sv_build(hipStream_t s, const uint8_t* ref, const uint8_t* lit, const SvPiece* pieces, uint32_t np, const SvSubst* subs, uint32_t nsub, uint8_t* out, uint64_t total) {
    if (total && np) hipLaunchKernelGGL(k_sv_build, dim3((unsigned)((total + 16 * 256 - 1) / (16 * 256))), dim3(256), 0, s, ref, lit, pieces, np, out, total);
    if (nsub) hipLaunchKernelGGL(k_sv_subst, dim3((nsub + 255) / 256), dim3(256), 0, s, subs, nsub, out);
}
// A piece of a REGULAR FASTA record (every line but the last holds lb bases in lw bytes: what the .fai states) without its line
// ends: base j of the piece, which starts in column col0 of its line, lies at raw offset j + ((col0 + j) / lb) * (lw - lb).
__global__ void __launch_bounds__(256) k_fa_gather_regular(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst, uint64_t n, uint32_t col0, uint32_t lb, uint32_t lw) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = raw[j + ((uint64_t)col0 + j) / lb * (uint64_t)(lw - lb)];
}
void launch_fa_gather_regular(hipStream_t s, const uint8_t* raw, uint8_t* dst, uint64_t n, uint32_t col0, uint32_t lb, uint32_t lw) {
    if (n) hipLaunchKernelGGL(k_fa_gather_regular, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, raw, dst, n, col0, lb, lw);
}
void launch_encode_bases(hipStream_t s, uint8_t* g, uint64_t n) {
    if (n) hipLaunchKernelGGL(k_encode_bases, dim3((unsigned)((n + 16 * 256 - 1) / (16 * 256))), dim3(256), 0, s, g, n);
}

// ------------------------------------------------------------------------------------------------
// genome bit index: per 64-base word a G/C mask and an N mask plus running counts, so the GC count
// and the any-N test of ANY window are O(1) (countGC, lib/mydefine/MyDefine.cpp:434-452, without
// re-reading the 1-2 kb window per amplicon; GC-ness and N-ness are strand-invariant).
// ------------------------------------------------------------------------------------------------
// Also the genome with TWO BITS PER BASE (g2: base i in bits 2 (i & 15) of word i >> 4, a non-ACGT base as 0): the windows of
// reads that cannot see an N (k_reads' uniform walk) are gathered from it -- a quarter of the bytes, and already in the form
// the walk keeps them in LDS.
__global__ void k_genome_bits(const uint8_t* __restrict__ g, uint64_t n, uint64_t nwords, unsigned long long* __restrict__ gc_bits,
                              unsigned long long* __restrict__ n_bits, uint32_t* __restrict__ gc_cnt, uint32_t* __restrict__ n_cnt, uint32_t* __restrict__ g2) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwords) return;
    unsigned long long gm = 0, nm = 0; uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    if (w < nwords) {
        const uint64_t b0 = w * 64;
        for (int k = 0; k < 64; ++k) {
            const uint64_t i = b0 + k;
            if (i < n) {
                const uint32_t c = g[i]; gm |= (unsigned long long)is_gc(c) << k; nm |= (unsigned long long)(c > 3) << k;
                const uint32_t v = (c & 3u) << (2 * (k & 15));
                if (k < 16) p0 |= v; else if (k < 32) p1 |= v; else if (k < 48) p2 |= v; else p3 |= v;
            }
        }
    }
    gc_bits[w] = gm; n_bits[w] = nm; gc_cnt[w] = __popcll(gm); n_cnt[w] = __popcll(nm);
    reinterpret_cast<uint4*>(g2)[w] = make_uint4(p0, p1, p2, p3);
}
__device__ __forceinline__ uint64_t bit_rank(const unsigned long long* __restrict__ bits, const uint64_t* __restrict__ pref, uint64_t x) {
    const uint64_t w = x >> 6; const uint32_t r = (uint32_t)(x & 63);
    return pref[w] + (uint64_t)__popcll(bits[w] & ((1ull << r) - 1ull));
}

__global__ void k_frag_has_n(const uint64_t* __restrict__ goff, const uint32_t* __restrict__ len, uint32_t nf, DevGenomeIdx gx, uint8_t* __restrict__ has_n) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    has_n[f] = bit_rank(gx.n_bits, gx.n_pref, goff[f] + len[f]) != bit_rank(gx.n_bits, gx.n_pref, goff[f]) ? 1 : 0;
}
void launch_frag_has_n(hipStream_t s, const uint64_t* goff, const uint32_t* len, uint32_t nf, DevGenomeIdx gx, uint8_t* has_n) {
    if (nf) hipLaunchKernelGGL(k_frag_has_n, dim3((nf + 255) / 256), dim3(256), 0, s, goff, len, nf, gx, has_n);
}

// ------------------------------------------------------------------------------------------------
// K1b  new amplicon records: one thread per attached primer.  GC content of the window from the bit
//      index (+ the parent semi's substitutions), amplification errors as K ~ Binomial(l-8, ber) and K
//      distinct positions ([REMAP] of the per-base Bernoulli loop, Fragment.cpp:97-123 /
//      Amplicon.cpp:200-226), alt-base rejection draws, packed record written at its final
//      (reference -t 1 list) position.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t u4_word(const U4& d, uint32_t k) { return k == 0 ? d.w[0] : k == 1 ? d.w[1] : k == 2 ? d.w[2] : d.w[3]; }
template <bool FROM_FRAG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_errs(const uint8_t* __restrict__ g, DevGenomeIdx gx, DevFrags fr, DevAmps semis, DevErrPool spool,
                                              uint32_t n_slots, const uint32_t* __restrict__ slot_off, const uint32_t* __restrict__ slots,
                                              const uint32_t* __restrict__ slot_tmpl, const uint32_t* __restrict__ valid_off, uint32_t n_tmpl,
                                              DevAmps out, uint32_t out_base, DevErrPool pool, uint32_t* __restrict__ flags,
                                              const unsigned long long* __restrict__ binom, AmplifyParams p,
                                              int64_t* __restrict__ primer_cnt, uint32_t* __restrict__ primer_delta, unsigned long long* __restrict__ primer_cut,
                                              unsigned long long* __restrict__ sums, unsigned long long* __restrict__ semis_n) {
    __shared__ uint16_t s_item[4][256], s_res[4][256];                             // per wave: the errors of its amplicons (owner lane | index << 6), and what came back
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    // the pass epilogue rides along (unsharded job; a sharded one all-reduces what the shards took first and launches
    // k_primer_update_sharded): primer stock -= what this pass took, and the device-side semi amplicon count that the next
    // setPrimers reads
    if (primer_cnt && w < 65536u) stock_update(w, primer_cnt, primer_delta, primer_delta, primer_cut, sums, flags);   // (the grid has at least 256 workgroups then)
    if (FROM_FRAG && w == 0 && semis_n) *semis_n += valid_off[n_tmpl];
    // fragments: a thread per reserved slot (nearly all of them are used).  Semi amplicons: three quarters of the reserved
    // slots stay unused (most primers find no place on a 1-2 kb template), so k_expand_items has listed the template of every
    // amplicon actually made (slot_tmpl reused as that dense map) and the thread index IS the amplicon.
    // (no lane leaves before the wave has resolved its errors together: see below)
    uint32_t t = 0, i = 0;
    bool active;
    if (FROM_FRAG) {
        active = w < n_slots;
        if (active) { t = slot_tmpl[w]; active = t != 0xFFFFFFFFu; }              // reserved but unused slot (aborted template)
        if (active) i = w - slot_off[t];
    } else {
        active = w < valid_off[n_tmpl];
        if (active) { t = slot_tmpl[w]; i = w - valid_off[t]; }
    }
    const uint32_t kind = FROM_FRAG ? 0u : 1u;
    uint32_t n_fwd = 0, spos = 0, alen = 0, plen = 0, K = 0, ntr = 0; uint64_t perrs = 0, nuid = 0, P = 0; int gcn = 0;
    View tv{0, 1, 0}; U4 d0{};
    if (active) {
        n_fwd = valid_off[t] + i;
        const uint32_t sl = slots[slot_off[t] + i];
        spos = sl_spos(sl); alen = sl_len(sl);
        if (FROM_FRAG) { tv = frag_view(fr.goff[t], fr.len[t], fr.strand[t]); nuid = semi_uid(fr.gidx_base + t, p.pass, i); }
        else {
            const uint32_t f = semis.parent[t], psl = semis.sl[t];
            plen = sl_len(psl); perrs = semis.errs[t];
            tv = semi_tmpl_view(frag_view(fr.goff[f], fr.len[f], fr.strand[f]), sl_spos(psl), plen);
            nuid = full_uid(semis.uid[t], p.pass, i);
        }
        // ---- GC / N of the template window [spos, spos+alen)
        const int64_t first = tv.base + (int64_t)tv.dir * (int64_t)spos;
        const uint64_t ga = (uint64_t)(tv.dir > 0 ? first : first - (int64_t)(alen - 1)), gb = ga + alen;
        int gc = (int)(bit_rank(gx.gc_bits, gx.gc_pref, gb) - bit_rank(gx.gc_bits, gx.gc_pref, ga));
        // the N count costs four more scattered loads: skipped for the templates of a fragment without any N (nearly all)
        int nn = fr.has_n[FROM_FRAG ? t : semis.parent[t]] ? (int)(bit_rank(gx.n_bits, gx.n_pref, gb) - bit_rank(gx.n_bits, gx.n_pref, ga)) : 0;
        if (!FROM_FRAG) for_each_err(perrs, spool.data, [&](uint32_t e) {
            const uint32_t tp = plen - 1 - err_pos(e);
            if (tp >= spos && tp < spos + alen) {
                const uint32_t orig = g[tv.base + (int64_t)tv.dir * (int64_t)tp];
                if (orig > 3) --nn; else gc -= is_gc(orig) ? 1 : 0;
                gc += is_gc(err_alt(e)) ? 1 : 0;                                   // complement keeps GC-ness
            }
        });
        gcn = nn > 0 ? 0 : gc;                                                     // countGC: 0 if any N
        // ---- error count and positions
        ntr = alen - 8;
        d0 = draw4(p.key, ST_ERR, kind, nuid, 0);
        const unsigned long long x64 = ((unsigned long long)d0.w[0] << 32) | d0.w[1];
        const unsigned long long* __restrict__ Tn = binom + (size_t)(ntr - (p.amp_min - 8)) * BINOM_KMAX;
        while (K < (uint32_t)BINOM_KMAX && x64 >= Tn[K]) ++K;
        // the K positions, sorted: 16 bits each of one register for K <= 4; the 2 amplicons in 10 000 with more keep them (and
        // then their entries) in their slice of the overflow pool (two 16-entry register arrays cost the kernel a wave per SIMD)
        if (K && K <= 4) {
            uint32_t cnt = 0, q = 0; U4 d = d0;
            while (cnt < K) {
                if ((q & 3) == 0) d = draw4(p.key, ST_ERR, kind, nuid, 1 + (q >> 2));
                const uint32_t cand = 8 + scale_draw(u4_word(d, q & 3), 0, ntr); ++q;
                bool dup = false; uint32_t below = 0;
                for (uint32_t z = 0; z < cnt; ++z) { const uint32_t v = (uint32_t)(P >> (16 * z)) & 0xFFFFu; dup |= v == cand; below += v < cand ? 1u : 0u; }
                if (!dup) {                                                        // insert sorted
                    const uint64_t low = (1ull << (16 * below)) - 1ull;
                    P = (P & low) | ((uint64_t)cand << (16 * below)) | ((P & ~low) << 16);
                    ++cnt;
                }
            }
        }
    }
    // entry of the error at position j of an amplicon (template base through the parent's substitutions, alternative base by
    // rejection) and its GC change
    auto resolve = [&](View v, uint32_t pl, uint64_t pe, uint32_t sp, uint64_t uid, uint32_t j, int& dgc) {
        const uint32_t base = FROM_FRAG ? view_base(g, v, sp + j) : semi_tmpl_base(g, v, pl, pe, spool.data, sp + j);
        uint32_t alt, a = 0;
        do {                                                                       // do { n = rand } while (bases[n] == base)
            const U4 e = draw4(p.key, ST_ERRALT, kind, uid, j | ((a >> 2) << 16));
            alt = u4_word(e, a & 3) >> 30; ++a;                                    // trunc(4 * x / 2^32)
        } while (alt == base);
        dgc = (is_gc(alt) ? 1 : 0) - (is_gc(base) ? 1 : 0);
        return err_pack(j, alt);
    };
    // ---- the wave resolves its errors together: 0.51 per amplicon, but 60 % of the amplicons have none and a lane with three
    // kept the other 63 waiting three rounds.  Every error becomes an item (owner lane, index); a lane takes ONE item, fetches
    // the owner's view by shuffles, and hands the entry back through LDS.
    const uint32_t cnt = K <= 4 ? K : 0u;
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if ((int)lane >= d) incl += v; }
    const uint32_t pre = incl - cnt, total = __shfl(incl, 63);
    for (uint32_t z = 0; z < cnt; ++z) s_item[wv][pre + z] = (uint16_t)(lane | (z << 6));
    __builtin_amdgcn_wave_barrier();
    for (uint32_t b = 0; b < total; b += 64) {
        const uint32_t it = b + lane; const bool valid = it < total;
        const uint32_t e = valid ? s_item[wv][it] : 0u, o = e & 63u, z = e >> 6;
        View ov; uint64_t ope, ouid, oP;
        {
            const uint32_t lo = __shfl((uint32_t)(uint64_t)tv.base, o), hi = __shfl((uint32_t)((uint64_t)tv.base >> 32), o);
            ov.base = (int64_t)(((uint64_t)hi << 32) | lo);
            const uint32_t fl = __shfl((uint32_t)(tv.dir > 0 ? 1u : 0u) | (tv.comp << 1), o);
            ov.dir = (fl & 1u) ? 1 : -1; ov.comp = fl >> 1;
            ope = ((uint64_t)__shfl((uint32_t)(perrs >> 32), o) << 32) | __shfl((uint32_t)perrs, o);
            ouid = ((uint64_t)__shfl((uint32_t)(nuid >> 32), o) << 32) | __shfl((uint32_t)nuid, o);
            oP = ((uint64_t)__shfl((uint32_t)(P >> 32), o) << 32) | __shfl((uint32_t)P, o);
        }
        const uint32_t opl = __shfl(plen, o), osp = __shfl(spos, o);
        if (valid) {
            int dgc;
            const uint32_t ent = resolve(ov, opl, ope, osp, ouid, (uint32_t)(oP >> (16 * z)) & 0xFFFFu, dgc);
            s_res[wv][it] = (uint16_t)(ent | ((uint32_t)(dgc + 1) << 13));        // entry: 13 bits; GC change + 1 above
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (!active) return;
    uint64_t packed = 0;
    if (cnt) {
        for (uint32_t z = 0; z < cnt; ++z) { const uint32_t r = s_res[wv][pre + z]; packed |= (uint64_t)(r & 0x1FFFu) << (16 * z); gcn += (int)(r >> 13) - 1; }
    } else if (K) {
        const uint32_t off = atomicAdd(pool.head, K);
        if (off + K > pool.cap) atomicOr(flags, (uint32_t)FLAG_ERRPOOL);
        else {
            uint32_t* pos = pool.data + off;
            uint32_t c2 = 0, q = 0; U4 d = d0;
            while (c2 < K) {
                if ((q & 3) == 0) d = draw4(p.key, ST_ERR, kind, nuid, 1 + (q >> 2));
                const uint32_t cand = 8 + scale_draw(u4_word(d, q & 3), 0, ntr); ++q;
                bool dup = false;
                for (uint32_t z = 0; z < c2; ++z) dup |= pos[z] == cand;
                if (!dup) { uint32_t z = c2++; while (z > 0 && pos[z - 1] > cand) { pos[z] = pos[z - 1]; --z; } pos[z] = cand; }
            }
            for (uint32_t z = 0; z < K; ++z) { int dgc; pos[z] = resolve(tv, plen, perrs, spos, nuid, pos[z], dgc); gcn += dgc; }
            packed = ERR_OVERFLOW_BIT | ((uint64_t)K << 32) | off;
        }
    }
    if (gcn < 0) gcn = 0;                                                          // max(0, gcNum)
    const uint32_t n_new = valid_off[n_tmpl];
    const uint32_t dst = out_base + (n_new - 1 - n_fwd);                          // reversed within the pass: insertLinkList prepends
    out.parent[dst] = t; out.sl[dst] = pack_sl(spos, alen); out.gc[dst] = (uint16_t)gcn; out.primers[dst] = 0;
    out.uid[dst] = nuid; out.errs[dst] = packed;
}

// ------------------------------------------------------------------------------------------------
// deterministic log (same operation sequence as oracle/scs_oracle.cpp det_log; IEEE + - * / only)
// ------------------------------------------------------------------------------------------------
__device__ double det_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0) return __longlong_as_double(0x7ff8000000000000LL);
    if (x == 0) return __longlong_as_double(0xfff0000000000000LL);
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    int k = 0;
    if ((b >> 52) == 0) { x *= 18014398509481984.0; b = (unsigned long long)__double_as_longlong(x); k = -54; }
    if ((b >> 52) == 0x7ff) return x;
    k += (int)(b >> 52) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)b);
    if (m >= 1.4142135623730951) { m = m * 0.5; k += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp(x) for -256 <= x <= 0, the oracle's det_exp operation for operation (fdlibm's reduction and polynomial)
__device__ double det_exp(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x >= 0) return 1.0;
    if (x < -256.0) return 0.0;
    const int k = (int)(invln2 * x - 0.5);
    const double dk = (double)k;
    const double hi = x - dk * ln2_hi, lo = dk * ln2_lo, r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return __longlong_as_double(__double_as_longlong(y) + ((long long)k << 52));
}

// K2  weights: Amplicon::getWeightedLength (Amplicon.cpp:396-400) x Profile::getGCFactor (Profile.cpp:1503-1513)
// [REMAP] Marsaglia polar, keyed: attempt a = words 2(a&1), 2(a&1)+1 of Philox block a/2 of the amplicon's uid; the first
// attempt inside the unit disc whose value is not negative counts.  Four of five amplicons are served by attempt 0 and one in
// 15 needs an attempt of a later block (or drew a negative value): written as ONE loop, nearly every wave pays a second
// Philox block and two or three more logarithm / square root / division rounds for its few unlucky lanes.  So the
// workgroup's first pass takes the first accepted attempt of block 0 without branching, and the amplicons it leaves are
// queued in LDS and finished by the workgroup's first lanes, densely.
struct GcDraw { double y, r2; bool in; };
__device__ __forceinline__ GcDraw gc_attempt(uint32_t wx, uint32_t wy) {
    const double x = 2.0 * (((double)wx + 0.5) / 4294967296.0) - 1.0;
    const double y = 2.0 * (((double)wy + 0.5) / 4294967296.0) - 1.0;
    const double r2 = x * x + y * y;
    return GcDraw{y, r2, !(r2 > 1.0 || r2 == 0.0)};
}
__device__ __forceinline__ double gc_value(double mean, double sd, double y, double r2) {
    const double mult = __dsqrt_rn(-2.0 * det_log(r2) / r2);
    return mean + sd * (y * mult);
}
constexpr int WEIGHTS_BLOCK = 1024;
__global__ __launch_bounds__(WEIGHTS_BLOCK) void k_weights(DevAmps fulls, uint32_t n, DevTables tb, RngKey key, uint32_t frag_size, double* __restrict__ w) {
    __shared__ uint32_t s_retry[WEIGHTS_BLOCK];                                    // thread | first attempt still to try << 16
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const double scale = (double)(frag_size * frag_size), sd = tb.gc_std;
    if (i < n) {
        const uint32_t len = sl_len(fulls.sl[i]);
        const uint32_t gcp = 100u * fulls.gc[i] / len;
        if (gcp > 100) w[i] = 0.0 * (double)len / scale;
        else {
            const U4 d = draw4(key, ST_WEIGHT, 0, fulls.uid[i], 0);
            const GcDraw a0 = gc_attempt(d.w[0], d.w[1]), a1 = gc_attempt(d.w[2], d.w[3]);
            const bool any = a0.in || a1.in;
            double v = -1.0;
            if (any) v = gc_value(tb.gc_means[gcp], sd, a0.in ? a0.y : a1.y, a0.in ? a0.r2 : a1.r2);
            if (v < 0) s_retry[atomicAdd(&s_n, 1u)] = threadIdx.x | ((any && a0.in ? 1u : 2u) << 16);
            else w[i] = v * (double)len / scale;
        }
    }
    __syncthreads();
    const uint32_t nr = s_n;
    for (uint32_t k = threadIdx.x; k < nr; k += blockDim.x) {
        const uint32_t e = s_retry[k], j = blockIdx.x * blockDim.x + (e & 0xFFFFu);
        const uint32_t len = sl_len(fulls.sl[j]);
        const double mean = tb.gc_means[100u * fulls.gc[j] / len];
        const uint64_t uid = fulls.uid[j];
        uint32_t a = e >> 16;
        U4 d = draw4(key, ST_WEIGHT, 0, uid, a >> 1);
        double v;
        for (;; ++a) {
            if ((a & 1u) == 0 && a != (e >> 16)) d = draw4(key, ST_WEIGHT, 0, uid, a >> 1);
            const GcDraw t = (a & 1u) ? gc_attempt(d.w[2], d.w[3]) : gc_attempt(d.w[0], d.w[1]);
            if (!t.in) continue;
            v = gc_value(mean, sd, t.y, t.r2);
            if (v >= 0) break;
        }
        w[j] = v * (double)len / scale;
    }
}

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
    const uint32_t n_isz = (uint32_t)tb.n_isize;
    const bool isz_lds = n_isz <= 1024u;
    if (isz_lds) for (uint32_t k = threadIdx.x; k < n_isz; k += blockDim.x) s_isz[k] = tb.isize_t[k];
    __syncthreads();
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_fulls) return;
    const uint32_t i = first + j;
    int n = (int)read_numbers[i];
    if (n == 0) return;
    const uint32_t po = pair_off[i], want = pair_off[i + 1] - po;
    if (po >= pair_hi || po + want <= pair_lo) return;                               // none of its pairs lies in this batch
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
//       3  the same walk for reads whose only event is the deletion of one base (n' = L - 1: bins j L / (L - 1) = j);
//       2  everything else: the general loop described above.  (0: explicit-window mode, the general loop.)
//     The prologue is a chain of dependent loads (list entry -> pair record + offset -> window gather) at four workgroups
//     per CU: the ring's first groups and the event words are requested early, the records are parked in LDS for the
//     hand-out by sector phase, the gather issues all its loads before it uses one, and every barrier orders LDS only
//     (lds_barrier).  -DSCS_PHASE_CLOCK times the phases of a workgroup's life.
//     LDS per workgroup at L = 150: 14 KB ring + 19 KB rows (uniform walks) / 16 KB ring + 4 KB events + 19 KB windows
//     (general) -> 4 workgroups per CU.
// ------------------------------------------------------------------------------------------------
#define RB 256
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
// the uniform walk's bin: the 3-mers' KEEP intervals (lo, width) instead of their threshold triples (scs_pipeline.cpp ring_image_u)
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
                                       U4 seed, char* __restrict__ out_b, char* __restrict__ out_q, uint32_t del_pos) {
    // (del_pos: the one deleted base of a read of the one-deletion class, 0xFFFF for none: n - 1 positions, bins j * n / (n - 1) = j)
    Xoshiro xb; xb.seed(seed);
    uint32_t c0 = 5u, c1 = 5u;
    const int np = del_pos == 0xFFFFu ? n : n - 1;
    for (int tp = 0; tp < np; ++tp) {
        const int t = tp + ((uint32_t)tp >= del_pos ? 1 : 0);                      // window base of output position tp
        uint32_t c2 = g[(gf & 2u) ? gb - t : gb + t];
        if ((gf & 1u) && c2 < 4u) c2 = 3u - c2;
        for_each_err(e1, spool, [&](uint32_t e) {
            const int tt = k1 - (int)err_pos(e); const int k = rd ? (int)(pos + isz - 1) - tt : tt - (int)pos;
            if (k == t) c2 = rd ? err_alt(e) : 3u - err_alt(e);
        });
        for_each_err(e2, fpool, [&](uint32_t e) {
            const int tt = (int)err_pos(e); const int k = rd ? (int)(pos + isz - 1) - tt : tt - (int)pos;
            if (k == t) c2 = rd ? 3u - err_alt(e) : err_alt(e);
        });
        const int ki = kmer_index(c0, c1, c2);
        uint32_t xs, xq; xb.next2(xs, xq);
        uint32_t bc, qc;
        if (ki < 0 && c2 > 3u) { bc = 'N'; qc = 33u + scale_draw(xq, 0, 20); }
        else { const uint32_t kq = call_global_body<QK>(subs, subs_d, qalias, B, ki, c2, c2, (uint32_t)tp, xs, xq); bc = (0x54474341u >> (8u * (kq & 255u))) & 255u; qc = 33u + (kq >> 8); }
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
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nreads = paired ? 2 * np : np;
    if (r >= nreads) return;
    const uint32_t pi = paired ? r >> 1 : r, rd = paired ? (r & 1u) : 0u;
    const uint64_t uid = pairs[pi].uid; const uint32_t att = pairs[pi].att, isz = pairs[pi].isz, amp = pairs[pi].amp, has_n = pairs[pi].flags & 4u;
    uint32_t* sz = rd ? sizes2 : sizes1;
    uint32_t* d1f = rd ? d1f2 : d1f1;                                              // 1: the read's only event is the deletion of one base (k_reads' one-deletion walk)
    if (isz == 0) { ev_hdr[r] = 0; sz[pi] = 0; d1f[pi] = 0; return; }               // hole: the insert-size loop gave up (Amplicon.cpp:484-489)
    unsigned long long e_lo = 0, e_hi = 0;
    const IndelPass ip = indel_pass(tb, key, rd | (att << 1), uid, force_replay, slot, flags, [&](int i, uint32_t v) {
        if (i < 4) e_lo |= (unsigned long long)v << (16 * i); else e_hi |= (unsigned long long)v << (16 * (i - 4));
    });
    ev_hdr[r] = (uint32_t)ip.n_out | ((uint32_t)ip.nev << 16) | (ip.replay ? 1u << 24 : 0u) | (1u << 25);
    if (ip.nev > 0) ev_dat[r] = make_uint4((uint32_t)e_lo, (uint32_t)(e_lo >> 32), (uint32_t)e_hi, (uint32_t)(e_hi >> 32));   // (86 % of the reads have no event: nothing reads their slots)
    const uint32_t e0 = (uint32_t)e_lo & 0xFFFFu;
    const bool d1 = ip.nev == 1 && !ip.replay && !has_n && !(force_replay & 12u) && ev_del(e0) && ev_len(e0) == 1u && ip.n_out == tb.L - 1 && tb.bins == tb.L;
    d1f[pi] = d1 ? 1u : 0u;
    const uint32_t cls = ((ip.nev > 0 || ip.replay || has_n || (force_replay & 4u)) && !d1) ? 1u : 0u;   // the uniform walk takes ACGT-only windows without events
    // "@<ampIdx>#<fragCount>[/1|/2]\n" + seq + "\n+\n" + qual + "\n"   (Amplicon.cpp:459-466,497-504)
    sz[pi] = (ip.n_out == 0 ? 0u : 1u + dec_digits(amp) + 1u + dec_digits(att + 1) + (paired ? 2u : 0u) + 1u + 2u * (uint32_t)ip.n_out + 4u) | (cls << 31);   // bit 31: the class rides along into the offsets' scan
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
    uint32_t* s_head = reinterpret_cast<uint32_t*>(s_dyn + SLOTS * sizeof(Bin) + (size_t)RB * ROW);   // [64] UNI: threshold rows of the 1- and 2-mers
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
#ifdef SCS_PHASE_CLOCK
    unsigned long long ph_t_ = 0;
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
    const uint32_t head_w = UNI ? reinterpret_cast<const uint32_t*>(ring_img + (size_t)((B + 7) & ~7) * (sizeof(Bin) / 16))[tid & 63] : 0u;
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
    int nev = 0, n_out = 0; bool replay = false; uint32_t replay_first = 0, del_pos = 0xFFFFu;
    if (live) {
        if (FROM_PAIRS) {
            const uint32_t h = ev_h; const uint4 e = ev_e;
            n_out = (int)(h & 0xFFFFu); nev = SIMPLE ? 0 : (int)((h >> 16) & 0xFFu); replay = SIMPLE ? false : (h >> 24) & 1u;
            if (D1) del_pos = ev_pos(e.x & 0xFFFFu);                                 // its one event: the deleted base
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
    if (UNI && live && n_out > 0) {
        // The name line "@<amp>#<cnt>[/1|/2]\n" is WRITTEN INTO LDS FIRST, character by character at its place from the end -- into the
        // 28 bytes in front of my window that the walk's pending entries use later, the line's last character in byte 27 (h <= 25) --
        // and read back as seven words: no six-way choice per character, the digit loops are the only data-dependent part.  The words are
        // shifted to the record's alignment: the last s1 characters ride in the first dword of the bases, the rest ends on the aligned
        // address ta1 and goes out as whole dwords, then the <= 3 leading bytes.
        const uint32_t amp = amp_index_base + pr.amp, cnt = pr.att + 1u, d2 = dec_digits(cnt), h = rec_h, dbase = paired ? 3u : 1u, da = h - 2u - dbase - d2;
        const uint32_t o1 = rec_rel + h, o2 = o1 + (uint32_t)n_out + 3u;          // where the bases / the qualities start
        a1 = o1 & 31u; sec1 = o1 - a1; a2 = o2 & 31u; sec2 = o2 - a2;
        const uint32_t s1 = a1 & 3u; char* ta1 = wg_out + (o1 - s1);
        LdsU8* nb = (LdsU8*)my_pend_lds;
        nb[27] = (uint8_t)'\n';
        if (paired) { nb[26] = (uint8_t)(rd ? '2' : '1'); nb[25] = (uint8_t)'/'; }
        uint32_t at = 27u - dbase, v = cnt;                                        // byte of the next character to the left
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
#pragma unroll
    for (int u = 0; u < NPRE; ++u) { const int idx = tid + u * RB; if (idx < GE) ring16[idx] = ring0[u]; }
    if (UNI && tid < 64) s_head[tid] = head_w;
    lds_barrier();
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
        c0 = 0; c1 = 0;
        // D1 (CLS 3): the reads with exactly ONE indel event, the deletion of ONE base, run the same walk.  Their n' = L - 1 positions
        // fall into bins j * L / (L - 1) = j, one step of stream B each; only the source base differs: position j reads window
        // base j before the deleted base and j + 1 from it on (a shift by one two-bit field, the next dword kept beside the
        // current one).  NP: the positions of a read of this class.
        const int NP = D1 ? B - 1 : B;
        // SOFTWARE PIPELINE, one position deep.  The compiler's scheduler waits for an LDS read right where it issues it; here
        // every read gets a stage's worth of independent work before its use.  Step u runs, in this order,
        //   finish_a(u-1): thresholds and alias entry of the previous position are back -> its base k, its alias column -> issue the symbol read
        //   start(u):      this position's window base, stream-B step, table addresses -> issue the threshold and alias-entry reads
        //   finish_b(u-1): the symbol is back -> pending-quality bookkeeping, the output words
        // with the ring's refill (commit, barrier, prefetch) between finish_a and start: every ring read of a group is issued
        // before its wave arrives at the next group's barrier, as the refill's slot reuse assumes.
        uint32_t aLo = 0, aWid = 0, aE = 0, aX1 = 0, aX2 = 0, aC2 = 0, aRow = 0; const LdsU8* aQ = ring8;   // start -> finish_a
        uint32_t bC2 = 0, bX1 = 0, bX2 = 0, bSym = 0, bRow = 0; bool bKept = true;                          // finish_a -> finish_b
        // (FIRST: t0 == 0, as a compile-time constant -- a run-time test would put branches between the positions, and a branch around
        // the stores makes the compiler's s_waitcnt before the next ring commit cover them on every path: vmcnt(1) instead of vmcnt(5))
        auto start = [&](auto U, auto FIRST, int t0) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            constexpr bool first = decltype(FIRST)::value;
            uint32_t c2;
            if constexpr (D1) {
                if (u == 0) { wreg = first ? win32[0] : wnext; wnext = win32[(t0 >> 4) + 1]; if (!mine) { wreg = 0; wnext = 0; } }   // 16 bases + the 16 behind them
                const bool after = (uint32_t)(t0 + u) >= del_pos;                    // from the deleted base on: the next window base
                if (u < 15) c2 = __builtin_amdgcn_ubfe(wreg, 2u * u + (after ? 2u : 0u), 2u);
                else c2 = after ? (wnext & 3u) : (wreg >> 30);
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
            if constexpr (u < 2 && first) { rowi = u == 0 ? c2 : 4u + c1 * 4u + c2; kp8 = head8 + rowi * 8u; }   // the read's first two bases: 1-mer / 2-mer rows (table rows 0..19)
            const LdsU32* qrow = (const LdsU32*)(bin8 + c2 * (uint32_t)(QROW * 16));   // the diagonal row (c2, c2) as an alias row
            const u32x2_t kp = *(const LdsU2*)kp8;                                    // the draws that keep the base: lo <= x1 < lo + width
            aLo = kp.x; aWid = kp.y; aE = qrow[x2 >> (32u - Geo::ABITS)];
            aQ = (const LdsU8*)(qrow + QK); aX1 = x1; aX2 = x2; aC2 = c2; aRow = rowi;
            c0 = c1; c1 = c2;
            __builtin_amdgcn_sched_barrier(0);
        };
        auto finish_a = [&]() __attribute__((always_inline)) {
            bKept = aX1 - aLo < aWid;                                                 // k == c2 (never for the draw 0xFFFFFFFF)
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
            const uint32_t lastj = (uint32_t)n_out - 1u, m = lastj >> 4, nw = ((lastj & 15u) >> 2) + 1u, nv = (lastj & 3u) + 1u;
            const uint32_t s1 = a1 & 3u, s2 = a2 & 3u, nq = s2 + (uint32_t)n_out + 1u;
            bo_b.tail(wg_out, sec1, a1, m, nw, nv, 0x0A2B0Au, (s1 + (uint32_t)n_out + 3u - s2) >> 2, 0u);   // the bases' dwords end where the qualities' first one starts
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
                          draw4(key, ST_READ, aux, uid, 1), wg_out + sec1 + a1, wg_out + sec2 + a2, D1 ? del_pos : 0xFFFFu);
        }
    }
    if (live && !FROM_PAIRS) {
        lens[r] = (uint32_t)n_out;
    }
    SCS_PHASE(7);
#ifdef SCS_PHASE_CLOCK
    if (UNI && CLS == 1 && tid == 0) atomicAdd(&g_phase[15], 1ull);
#endif
}

// (bid: the workgroup's index within ITS class' grid -- blockIdx.x of a launch of one class, or blockIdx.x less the grids of the
// classes in front of it in the merged launch below)
template <bool FROM_PAIRS, int QK, int CLS>
__global__ void __launch_bounds__(RB, 4) k_reads(const uint8_t* __restrict__ g, DevErrPool spool, DevErrPool fpool, const PairRec* __restrict__ pairs,
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
// The base pass of a batch as ONE launch: the workgroups of the general class first (the longest), then the one-deletion class,
// then the event-free class.  The three grids used to go to three streams; whether they really ran side by side depended on
// which hardware queues the process' streams had been given (the reads stage moved by +-4 % from process to process).  One
// grid leaves the mix to the workgroup dispatcher.  lists: {general, one-deletion, event-free} x {mate 1, mate 2}.
struct ReadLists { const uint32_t* l[3][2]; uint32_t n[3][2]; uint32_t grid[3]; };
template <int QK>
__global__ void __launch_bounds__(RB, 4) k_reads_all(const uint8_t* __restrict__ g, const uint8_t* __restrict__ g2, DevErrPool spool, DevErrPool fpool, const PairRec* __restrict__ pairs,
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

__global__ void k_philox(const uint32_t* __restrict__ ctr, uint32_t n, RngKey key, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const U4 o = philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key.k0, key.k1);
    out[4 * i] = o.w[0]; out[4 * i + 1] = o.w[1]; out[4 * i + 2] = o.w[2]; out[4 * i + 3] = o.w[3];
}
__global__ void k_detlog(const double* __restrict__ x, uint32_t n, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] <= 0 ? det_exp(x[i]) : det_log(x[i]);   // the test entry serves both: arguments <= 0 go to det_exp
}

// ------------------------------------------------------------------------------------------------
// block-level helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
// every thread of the block must call this exactly once (contains __syncthreads)
__device__ __forceinline__ void block_add_u64(unsigned long long v, unsigned long long* __restrict__ dst) {
    __shared__ unsigned long long s_acc;
    if (threadIdx.x == 0) s_acc = 0;
    __syncthreads();
    v = wave_sum_u64(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_acc, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_acc) atomicAdd(dst, s_acc);
}

// ------------------------------------------------------------------------------------------------
// K0  primer budgets: Malbac::setPrimers (lib/malbac/Malbac.cpp:236-283) with poissRand
//     (lib/mydefine/MyDefine.cpp:69-80: Knuth, sum of logs of uniforms) -- one thread per template.
//     sums[0] += sum of k over fragments, sums[1] += sum of UNTRUNCATED k over semis (Malbac.cpp:282).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double poisson_lambda(const PoissonParams& p, uint32_t len) {
    const uint64_t template_num = p.totals ? p.totals[0] : p.nf + p.dev[DS_SEMIS_N], total_len = p.totals ? p.totals[1] : p.frag_len + p.dev[DS_SEMI_LEN];
    const uint64_t pool = p.total_primers_dev ? *p.total_primers_dev : p.total_primers;
    const unsigned long long expected = (unsigned long long)((double)pool * p.gamma * (double)template_num);
    return (double)expected * (1.0 * (double)len / (double)total_len);
}
// semis: lambda ~ 6 -> one thread per semi amplicon
// n_cap: the host's upper bound of the semi count (grid size); the count itself is read from the device scalars
__device__ __forceinline__ void poisson_semis_block(uint32_t block, DevAmps semis, uint32_t n_cap, const PoissonParams& p, uint32_t* __restrict__ budget_s,
                                                    unsigned long long* __restrict__ part) {
    const uint32_t i = block * blockDim.x + threadIdx.x;
    const uint32_t n_semis = (uint32_t)p.dev[DS_SEMIS_N];
    unsigned long long ks = 0;
    if (i >= n_semis && i <= n_cap) budget_s[i] = 0;                               // the scan runs over n_cap + 1 entries
    if (i < n_semis) {
        const double lambda = poisson_lambda(p, sl_len(semis.sl[i])), log2 = -lambda;
        const uint64_t tuid = semis.uid[i];
        const uint32_t aux = 1u | (p.call << 1);
        long x = -1; uint32_t n = 0; U4 d;
        if (lambda <= 256.0) {                                                     // [REMAP] product form: p *= u until p < exp(-lambda)
            const double L = det_exp(log2); double pr = 1.0;
            do {
                if ((n & 3) == 0) d = draw4(p.key, ST_POISSON, aux, tuid, n >> 2);
                pr = pr * ((double)d.w[n & 3] / 4294967296.0); ++n; ++x;
            } while (pr >= L);
        } else {
            double log1 = 0;
            do {
                if ((n & 3) == 0) d = draw4(p.key, ST_POISSON, aux, tuid, n >> 2);
                const double u = (double)d.w[n & 3] / 4294967296.0; ++n;
                log1 += det_log(u); ++x;
            } while (log1 >= log2);
        }
        budget_s[i] = (uint32_t)x & 0xFFFu; semis.primers[i] = (uint16_t)((uint32_t)x & 0xFFFu);      // 12-bit field (Amplicon.cpp:76-79)
        ks = (unsigned long long)x;
    }
    // the workgroup's sum goes to its own slot: half a million same-address atomics per call cost more than the draws
    __shared__ unsigned long long s_w[4];
    ks = wave_sum_u64(ks);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = ks;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// fragments: lambda in the hundreds to thousands -> one 256-thread workgroup per fragment.  A round = 1024 draws: every
// thread turns one Philox block into four logs (LDS, draw order); then the first wave adds them to log1 IN DRAW ORDER
// (the rounding of the serial loop), eight at a time with one exit test per eight.
__device__ __forceinline__ void poisson_frag_block(uint32_t t, DevFrags fr, const PoissonParams& p, uint32_t* __restrict__ budget_f, unsigned long long* __restrict__ part) {
    __shared__ double s_lg[1024];
    __shared__ int s_more;
    const int tid = threadIdx.x;
    const double log2 = -poisson_lambda(p, fr.len[t]);
    const uint64_t tuid = fr.gidx_base + t;
    const uint32_t aux = 0u | (p.call << 1);
    long x = -1; double log1 = 0;
    for (uint32_t c = 0;; ++c) {
        const U4 d = draw4(p.key, ST_POISSON, aux, tuid, c * 256u + (uint32_t)tid);   // draws 1024c + 4 tid .. + 3
#pragma unroll
        for (int j = 0; j < 4; ++j) s_lg[4 * tid + j] = det_log((double)d.w[j] / 4294967296.0);
        __syncthreads();
        if (tid < 64) {                                                            // uniform over the wave: every lane runs the same serial sum
            bool more = true;
            for (int i = 0; i < 1024 && more; i += 8) {
                double pre[8]; double acc = log1;
#pragma unroll
                for (int k = 0; k < 8; ++k) { acc += s_lg[i + k]; pre[k] = acc; }
                int stop = 8;
#pragma unroll
                for (int k = 7; k >= 0; --k) if (!(pre[k] >= log2)) stop = k;     // first draw that ends the loop
                if (stop < 8) { x += stop + 1; more = false; } else { x += 8; log1 = acc; }
            }
            if (tid == 0) s_more = more ? 1 : 0;
        }
        __syncthreads();
        if (!s_more) break;
    }
    if (tid == 0) { budget_f[t] = (uint32_t)(int)x; part[blockIdx.x] = (unsigned long long)x; }
}

// one launch for both template kinds: workgroups [0, nf) take a fragment each, the rest 256 semi amplicons each
__global__ void __launch_bounds__(256) k_poisson(DevFrags fr, DevAmps semis, uint32_t n_cap, PoissonParams p, uint32_t* __restrict__ budget_f,
                                                 uint32_t* __restrict__ budget_s, unsigned long long* __restrict__ part) {
    if (blockIdx.x < fr.n) poisson_frag_block(blockIdx.x, fr, p, budget_f, part);
    else poisson_semis_block(blockIdx.x - fr.n, semis, n_cap, p, budget_s, part);
}
// *dst += sum of a u64 array (per-workgroup partials).  A few dozen workgroups, one atomic each (integer sums: any order): as ONE
// workgroup this kernel waited 220 memory round trips in a row on the main stream (1.4 ms after every fragment pass)
__global__ void __launch_bounds__(1024) k_sum_u64_add(const unsigned long long* __restrict__ v, uint32_t n, unsigned long long* __restrict__ dst) {
    __shared__ unsigned long long s_p[16];
    unsigned long long a = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a += v[i];
    a = wave_sum_u64(a);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int k = 0; k < 16; ++k) t += s_p[k]; if (t) atomicAdd(dst, t); }
}
// sums[0] += budgets of the fragments (workgroups [0, nf)), sums[1] += budgets of the semi amplicons (the rest)
__global__ void __launch_bounds__(1024) k_poisson_sums(const unsigned long long* __restrict__ part, uint32_t nf, uint32_t nb, unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long s_p[2][16];
    unsigned long long a = 0, b = 0;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) { const unsigned long long v = part[i]; if (i < nf) a += v; else b += v; }
    a = wave_sum_u64(a); b = wave_sum_u64(b);
    if ((threadIdx.x & 63) == 0) { s_p[0][threadIdx.x >> 6] = a; s_p[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x < 2) { unsigned long long t = 0; for (int k = 0; k < 16; ++k) t += s_p[threadIdx.x][k]; sums[threadIdx.x] += t; }
}

// ------------------------------------------------------------------------------------------------
// a2  the primer stock, exactly (Malbac::updatePrimerCount, lib/malbac/Malbac.cpp:91-103: decrement if positive, under
//     a mutex; at -t 1 the attachments of a pass ask in list order: template by template, primer by primer).  A type is
//     used exactly `stock` times -- by the FIRST `stock` attachments in list order that ask for it.
//     An attachment's place in that order is its key (template index, primer index).  k_attach takes a type when
//     key <= cut[type].  A pass starts with cut = "all of the pass" for every type in stock (none for the others): if no
//     type was then taken more often than it has stock -- every pass of a job whose primers do not run out -- the pass IS
//     the sequential loop's result.  Otherwise (exact_stock in scs_pipeline.cpp) the over-demanded types get the key of
//     their stock-th attachment as cut (k_stock_collect, a sort, k_stock_pick) and the templates from the earliest such
//     key on are run again (k_attach with undo); what they now take elsewhere may move other cuts, so this repeats until
//     no type is over its stock and no cut type under it: at that fixed point every decision equals the sequential
//     loop's (induction over the keys), and each round extends the prefix of the list on which that holds.
// ------------------------------------------------------------------------------------------------
#define STOCK_KEY_BITS 46                                                           // (fragment < 2^26, primer < 2^20) or (semi < 2^32, primer < 2^12), + 1
template <bool FROM_FRAG> __device__ __forceinline__ unsigned long long attach_key(uint32_t t, uint32_t i) {
    return (((unsigned long long)t << (FROM_FRAG ? 20 : 12)) | i) + 1ull;         // never 0: cut 0 = nothing to be had
}
// one workgroup: which types were taken more often than they have stock (over), which cut types less (under: an earlier
// round's cut came too early -- it is lifted, and the pass is run again from where it lay); the over types numbered in type
// order, the start of each one's stretch in the sorted list of their attachments.  info: [0] over types, [1] their
// attachments, [2] under types, [3] first template to run again on account of the under types; [5] (k_stock_collect's cursor) and
// [6] (k_stock_pick's first template) are reset here
__global__ void __launch_bounds__(1024) k_stock_check(const int64_t* __restrict__ cnt, const uint32_t* __restrict__ taken, unsigned long long* __restrict__ cut, int key_shift,
                                                      uint32_t* __restrict__ eidx, uint32_t* __restrict__ etype, uint32_t* __restrict__ estart, unsigned long long* __restrict__ info) {
    __shared__ uint32_t s_n[1024], s_m[1024]; __shared__ uint32_t s_under, s_tmin;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_under = 0; s_tmin = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t n = 0, m = 0, under = 0, tmin = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < 64; ++k) {
        const uint32_t x = tid * 64 + k; const int64_t c = cnt[x]; const uint32_t d = taken[x];
        if ((int64_t)d > c) { ++n; m += d; }
        else { const unsigned long long q = cut[x]; if (q != 0ull && q != ~0ull && (int64_t)d < c) { ++under; tmin = min(tmin, (uint32_t)((q - 1ull) >> key_shift)); cut[x] = ~0ull; } }
    }
    s_n[tid] = n; s_m[tid] = m;
    if (under) { atomicAdd(&s_under, under); atomicMin(&s_tmin, tmin); }
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                                      // inclusive scans of both counts
        const uint32_t a = tid >= d ? s_n[tid - d] : 0u, b = tid >= d ? s_m[tid - d] : 0u;
        __syncthreads();
        s_n[tid] += a; s_m[tid] += b;
        __syncthreads();
    }
    uint32_t e = s_n[tid] - n, off = s_m[tid] - m;
    for (uint32_t k = 0; k < 64; ++k) {
        const uint32_t x = tid * 64 + k; const uint32_t d = taken[x];
        if ((int64_t)d > cnt[x]) { eidx[x] = e; etype[e] = x; estart[e] = off; ++e; off += d; } else eidx[x] = 0xFFFFFFFFu;
    }
    if (tid == 1023) { info[0] = s_n[1023]; info[1] = s_m[1023]; info[2] = s_under; info[3] = s_tmin; info[5] = 0; info[6] = s_tmin; }
}
// the attachments of the over-demanded types: (type's number, key) of every one the pass has made, in any order
template <bool FROM_FRAG, int G>
__global__ void __launch_bounds__(64) k_stock_collect(const uint8_t* __restrict__ g, DevFrags fr, DevAmps semis, DevErrPool spool, const uint32_t* __restrict__ slot_off,
                                                      const uint32_t* __restrict__ slots, const uint32_t* __restrict__ valid, const uint32_t* __restrict__ eidx,
                                                      unsigned long long* __restrict__ list, unsigned long long* __restrict__ info, uint32_t t_first, uint32_t t_end) {
    constexpr int TPB = 64 / G;
    const int lane = threadIdx.x, gi = lane / G, gl = lane % G;
    const uint32_t t = t_first + blockIdx.x * TPB + gi;
    uint32_t len = 0, n = 0, base_slot = 0; uint64_t errs = 0; View tv{0, 1, 0};
    if (t < t_end) {
        n = valid[t]; base_slot = slot_off[t];
        if (FROM_FRAG) { len = fr.len[t]; tv = frag_view(fr.goff[t], len, fr.strand[t]); }
        else {
            const uint32_t f = semis.parent[t], sl = semis.sl[t];
            len = sl_len(sl); errs = semis.errs[t];
            tv = semi_tmpl_view(frag_view(fr.goff[f], fr.len[f], fr.strand[f]), sl_spos(sl), len);
        }
    }
    uint32_t rounds = (n + G - 1) / G;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rounds = max(rounds, (uint32_t)__shfl_xor((int)rounds, d));
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t w = r * G + gl; uint32_t e = 0xFFFFFFFFu;
        if (w < n) {
            const uint32_t sp = sl_spos(slots[base_slot + w]);
            unsigned long long v8 = view_bases8(g, tv, sp);
            if (!FROM_FRAG) for_each_err(errs, spool.data, [&](uint32_t er) {
                const uint32_t k = len - 1u - err_pos(er) - sp;
                if (k < 8u) v8 = (v8 & ~(0xFFull << (8u * k))) | ((unsigned long long)(3u - err_alt(er)) << (8u * k));
            });
            uint32_t idx = 0;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) idx = (idx << 2) | ((uint32_t)(v8 >> (8u * k)) & 3u);
            e = eidx[idx];
        }
        const unsigned long long hit = __ballot(e != 0xFFFFFFFFu);
        if (hit) {
            unsigned long long base = 0;
            if (lane == __ffsll((long long)hit) - 1) base = atomicAdd(&info[5], (unsigned long long)__popcll(hit));
            base = ((unsigned long long)__shfl((int)(base >> 32), __ffsll((long long)hit) - 1) << 32) | (uint32_t)__shfl((int)base, __ffsll((long long)hit) - 1);
            if (e != 0xFFFFFFFFu) list[base + __popcll(hit & ((1ull << lane) - 1ull))] = ((unsigned long long)e << STOCK_KEY_BITS) | attach_key<FROM_FRAG>(t, w);
        }
    }
}
// an over-demanded type's cut = the key of its stock-th attachment in list order; info[6] = the first template any new cut lies in
__global__ void __launch_bounds__(256) k_stock_pick(const int64_t* __restrict__ cnt, const uint32_t* __restrict__ etype, const uint32_t* __restrict__ estart, uint32_t ne,
                                                    const unsigned long long* __restrict__ sorted, unsigned long long* __restrict__ cut, int key_shift, unsigned long long* __restrict__ info) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const uint32_t x = etype[e];
    const unsigned long long k = sorted[(size_t)estart[e] + (size_t)cnt[x] - 1] & ((1ull << STOCK_KEY_BITS) - 1ull);
    cut[x] = k;
    atomicMin(&info[6], (k - 1ull) >> key_shift);
}

// ------------------------------------------------------------------------------------------------
// K1a  attach: the primer loop of Fragment::amplify (lib/fragment/Fragment.cpp:73-95) and
//      Amplicon::amplify (lib/amplicon/Amplicon.cpp:176-198).  The reference's loop is sequential
//      in the primer index (posAttached[] and the >50-tries abort).  Here a group of G lanes owns
//      one template (G = 64: one wave per fragment, budgets of hundreds; G = 8 or 4: eight or
//      sixteen semi amplicons per wave, budgets of a few): the lanes evaluate G primers speculatively and commit
//      them in index order -- a primer commits only when every lower primer has; one whose
//      proposal hits a committed position moves on to its next try exactly as the sequential loop
//      would.  The result is identical to running the sequential loop.
// ------------------------------------------------------------------------------------------------
template <bool FROM_FRAG, int G>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(G == 4 ? 8 : 1, 8))) k_attach(const uint8_t* __restrict__ g, DevFrags fr, DevAmps semis, uint32_t n_semis, DevErrPool spool,
                                               const uint32_t* __restrict__ slot_off, uint32_t* __restrict__ slots, uint32_t* __restrict__ slot_tmpl,
                                               uint32_t* __restrict__ valid, const unsigned long long* __restrict__ primer_cut, uint32_t* __restrict__ primer_delta,
                                               unsigned long long* __restrict__ len_sum, AmplifyParams p, uint32_t t_first, uint32_t t_end, int undo,
                                               const unsigned long long* __restrict__ t_from) {
    constexpr int TPB = 64 / G;                              // templates per wave
    constexpr int WORDS = FROM_FRAG ? 4096 : 64;             // position bitmap: 131072 / 2048 positions (packed-record limits)
    __shared__ uint32_t s_bits[TPB * WORDS];
    const int lane = threadIdx.x, gi = lane / G, gl = lane % G;
    const unsigned long long gmask = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << (gi * G));
    // templates [t_first, t_end) of the pass's list -- of which a run-again (undo) only touches those from *t_from on
    const uint32_t t = t_first + blockIdx.x * TPB + gi;
    const uint32_t nt = t_from && (unsigned long long)t < *t_from ? 0u : t_end;
    uint32_t len = 0, budget = 0; uint64_t tuid = 0, errs = 0; View tv{0, 1, 0};
    if (t < nt) {
        if (FROM_FRAG) { len = fr.len[t]; budget = fr.primers[t]; tuid = fr.gidx_base + t; tv = frag_view(fr.goff[t], len, fr.strand[t]); }
        else {
            const uint32_t f = semis.parent[t], sl = semis.sl[t];
            len = sl_len(sl); budget = semis.primers[t]; tuid = semis.uid[t]; errs = semis.errs[t];
            tv = semi_tmpl_view(frag_view(fr.goff[f], fr.len[f], fr.strand[f]), sl_spos(sl), len);
        }
    }
    uint32_t* bits = s_bits + gi * WORDS;
    const uint32_t base_slot = t < nt ? slot_off[t] : 0, aux = (FROM_FRAG ? 0u : 1u) | (p.pass << 1);
    // the primer type under a position of my template: its 8 bases are contiguous in the genome -> ONE 8-byte load (reversed /
    // complemented in registers), then the semi's own substitutions are patched in (no load sits under a branch)
    auto primer_type = [&](uint32_t sp, bool& hasN) -> uint32_t {
        unsigned long long v8 = view_bases8(g, tv, sp);
        if (!FROM_FRAG) for_each_err(errs, spool.data, [&](uint32_t e) {
            const uint32_t k = len - 1u - err_pos(e) - sp;                           // template position of the error, relative to sp
            if (k < 8u) v8 = (v8 & ~(0xFFull << (8u * k))) | ((unsigned long long)(3u - err_alt(e)) << (8u * k));
        });
        hasN = (v8 & 0xFCFCFCFCFCFCFCFCull) != 0;
        uint32_t idx = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) idx = (idx << 2) | ((uint32_t)(v8 >> (8u * k)) & 3u);
        return idx;
    };
    if (undo && t < nt) {                                                           // a pass run again from here on (exact_stock below): first take back what my template took before
        const uint32_t old = valid[t];
        for (uint32_t w = gl; w < old; w += G) { bool hn; const uint32_t idx = primer_type(sl_spos(slots[base_slot + w]), hn); atomicSub(&primer_delta[idx], 1u); }
    }
    if (FROM_FRAG) {                                                               // (semi amplicons: k_expand_items lists the amplicons made instead)
        for (uint32_t w = gl; w < budget; w += G) slot_tmpl[base_slot + w] = 0xFFFFFFFFu;   // my template's slots start out unused (k_errs skips those)
        __threadfence_block();                                                     // ... before any commit below rewrites one of them
    }
    bool group_done = !(t < nt && len >= p.amp_min + 27 && budget > 0);
    if (!group_done) for (uint32_t w = gl; w < (len + 31) / 32; w += G) bits[w] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t v = 0, c0 = 0, i = 0, tries = 0, spos = 0, alen = 0, pidx = 0;
    bool fresh = true, unresolved = false, need = false, dead = false;
    const AttachFit fit = group_done ? AttachFit{0, 1, 0, 1} : attach_fit_count(len, p.amp_min, p.amp_max);
    const double qfail = 1.0 - (double)fit.N / ((double)(len > 27 ? len - 27 : 1) * (double)fit.W);   // P(a try does not fit)
    unsigned long long lsum = 0; Xoshiro xt{};                                     // [REMAP] primer i's try stream, seeded by Philox block i
#ifdef SCS_PHASE_CLOCK
    __shared__ unsigned long long s_att_t, s_att_acc[8];
    if (lane < 8) s_att_acc[lane] = 0;
    if (lane == 0) s_att_t = wall_clock64();
    __builtin_amdgcn_wave_barrier();
#endif
    SCS_ATT(0);
    while (__ballot(!group_done)) {
        if (!group_done) {
            if (fresh) { i = c0 + gl; unresolved = i < budget; need = unresolved; dead = false; tries = 0; fresh = false; if (unresolved) xt.seed(draw4(p.key, ST_ATTACH, aux, tuid, i)); }
            SCS_ATT(1);
            if (unresolved && !dead) {
                if (!need && ((bits[spos >> 5] >> (spos & 31)) & 1u)) need = true;       // a lower primer took this position meanwhile
                while (need) {
                    // (a) my stream's next try that fits the template and lands on a free position: draws and LDS only, so the
                    // lanes of the wave run it together without a memory wait per candidate ...
                    bool cand = false;
                    while (!cand) {
                        // [REMAP] the tries that do not fit the template are skipped in one step: their number is geometric ...
                        tries += attach_gap(((double)xt.next() + 0.5) / 4294967296.0, qfail) + 1u;
                        if (tries > 50) { dead = true; break; }
                        // ... and the try that fits is uniform over the feasible (position, length) pairs
                        const unsigned long long x64 = ((unsigned long long)xt.next() << 32) | xt.next();
                        attach_fit_decode(fit, len, p.amp_min, (uint32_t)__umul64hi(x64, (unsigned long long)fit.N), spos, alen);
                        if ((bits[spos >> 5] >> (spos & 31)) & 1u) continue;              // posAttached[spos]
                        cand = true;
                    }
                    if (dead) break;
                    SCS_ATT(2);
                    // (b) ... and only then its primer 8-mer and the stock, all candidates of the wave in one round of loads.
                    // updatePrimerCount (Malbac.cpp:91-103) hands a type out while its stock lasts, in the list order of the
                    // attachments: the type's CUT is the place of this pass's list -- (template, primer) -- up to which it is to
                    // be had (all of the pass, none of it, or, once the pass's demand is known to exceed the stock, the place of
                    // the attachment that takes the last copy: exact_stock below)
                    bool hasN; const uint32_t idx = primer_type(spos, hasN);
                    const bool nostock = hasN || attach_key<FROM_FRAG>(t, i) > primer_cut[idx];
                    SCS_ATT(3);
                    if (nostock) continue;                          // no stock (none for N 8-mers)
                    pidx = idx; need = false;
                }
            }
        }
        SCS_ATT(4);
        // blocked = a lower unresolved live lane of my group proposes the same position
        const bool live = !group_done && unresolved && !dead;
        const unsigned long long um = __ballot(live);
        bool blocked = false;
        if (G == 64) {
            for (unsigned long long m = um; m; m &= m - 1) {
                const int j = __ffsll((long long)m) - 1;
                const uint32_t sj = __shfl(spos, j);
                if (j < lane && live && sj == spos) blocked = true;
            }
        } else {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const uint32_t sj = __shfl(spos, gi * G + j);
                if (j < gl && live && ((um >> (gi * G + j)) & 1ull) && sj == spos) blocked = true;
            }
        }
        const unsigned long long bm = __ballot(live && blocked) & gmask, dm = __ballot(!group_done && unresolved && dead) & gmask;
        const int first_blocked = bm ? __ffsll((long long)bm) - 1 - gi * G : G, first_dead = dm ? __ffsll((long long)dm) - 1 - gi * G : G;
        const int commit_end = first_blocked < first_dead ? first_blocked : first_dead;
        SCS_ATT(5);
        if (live && gl < commit_end) {                                                   // commit, in primer order
            atomicOr(&bits[spos >> 5], 1u << (spos & 31));
            atomicAdd(&primer_delta[pidx], 1u);
            slots[base_slot + i] = pack_sl(spos, alen); if (FROM_FRAG) slot_tmpl[base_slot + i] = t;
            lsum += alen; unresolved = false;
        }
        __builtin_amdgcn_wave_barrier();
        if (!group_done) {
            if (first_dead <= first_blocked && first_dead < G) { group_done = true; v = c0 + (uint32_t)first_dead; }   // abandons the remaining primers
            else if ((__ballot(unresolved) & gmask) == 0) {                              // chunk finished
                c0 += G;
                if (c0 >= budget) { group_done = true; v = budget; } else fresh = true;
            }
        } else (void)__ballot(false);
        SCS_ATT(6);
    }
#ifdef SCS_PHASE_CLOCK
    __builtin_amdgcn_wave_barrier();
    if (!FROM_FRAG && lane < 8) atomicAdd(&g_phase_att[lane * 256 + (blockIdx.x & 255u)], lane == 7 ? 1ull : s_att_acc[lane]);
#endif
    if (FROM_FRAG) { lsum = wave_sum_u64(lsum); if (lane == 0 && t < nt) len_sum[t] = lsum; }   // per fragment; launch_frag_len_sum adds them up (no same-address atomics)
    if (gl == 0 && t < nt) valid[t] = v;
}

// ------------------------------------------------------------------------------------------------
// K3  read allocation: Malbac::setReadCounts (lib/malbac/Malbac.cpp:370-408) with randIndx_hp /
//     batchSampling (lib/mydefine/MyDefine.cpp:191-272), chunk = 1000 amplicons of the WHOLE JOB's list
//     (loadPerThread at -t 1).  [REMAP] every sum has a fixed shape (oracle: tree1000 / tree_sum / scan1000 /
//     scan_all) instead of the reference's serial order, so that it can be computed by 64 lanes and by shards:
//       tree1000: lane l adds v[l], v[l+64], ... in index order, then the shuffle-xor butterfly d = 32 .. 1
//       scan1000: rows of 64, Hillis-Steele inclusive scan inside a row (shuffle-up d = 1 .. 32), carry row to row
//     A shard works on the chunks in which it has at least one amplicon.  A chunk that lies inside one of its own
//     (locally contiguous) list segments is read in place; the few chunks that straddle a segment boundary -- at most two
//     per segment -- are materialised as dense rows (k_alloc_bgather) from the shard's own weights and from the first /
//     last 1000 weights of every segment of every shard (one small all-gather), so that every shard that shares a chunk
//     computes the same sums for it.  Nothing O(amplicons) is replicated or exchanged: the shards all-reduce only the
//     per-chunk partials (8 B per 1000 amplicons).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_xor_f64(double v, int d) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_xor((int)b, d), hi = __shfl_xor((int)(b >> 32), d);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_up((int)b, d), hi = __shfl_up((int)(b >> 32), d);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl((int)b, src), hi = __shfl((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_butterfly_f64(double acc) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc = acc + shfl_xor_f64(acc, d);
    return acc;
}
// where work chunk q of this shard lives: its whole-job chunk id, its row (in place in the local list, or a
// materialised boundary row), its length, and whether this shard owns it (holds its first amplicon)
struct ChunkRef { uint32_t c, n, local0; int brow; bool owner; };
__device__ __forceinline__ ChunkRef chunk_ref(const AllocPlan& pl, uint32_t q) {
    ChunkRef r;
    if (q < pl.n_interior) {
        uint32_t k = 0;
        while (k + 1 < pl.n_ranges && q >= pl.rng[k + 1].q0) ++k;                  // <= 40 ranges
        const AllocRange& g = pl.rng[k];
        r.c = g.c0 + (q - g.q0); r.local0 = g.local0 + (q - g.q0) * ALLOC_CHUNK; r.brow = -1; r.owner = true;
        const unsigned long long b = (unsigned long long)r.c * ALLOC_CHUNK;
        r.n = (uint32_t)min((unsigned long long)ALLOC_CHUNK, pl.total - b);
    } else {
        const uint32_t bi = q - pl.n_interior;
        r.c = pl.bchunk[bi].c; r.n = pl.bchunk[bi].n; r.local0 = 0; r.brow = (int)bi; r.owner = pl.bchunk[bi].owner != 0;
    }
    return r;
}
// this shard's first / last 1000 weights of each of its list segments, for the other shards' boundary rows
__global__ void __launch_bounds__(256) k_alloc_bpack(const double* __restrict__ w, AllocPlan pl, double* __restrict__ send) {
    const uint32_t slot = blockIdx.x / 2, half = blockIdx.x & 1u;                   // one block per (segment slot, first | last)
    const uint32_t n = pl.my_seg[slot].n, lo = pl.my_seg[slot].lo, m = min(n, ALLOC_CHUNK);
    for (uint32_t i = threadIdx.x; i < ALLOC_CHUNK; i += blockDim.x)
        send[((size_t)slot * 2 + half) * ALLOC_CHUNK + i] = i < m ? w[lo + (half ? n - m + i : i)] : 0.0;
}
// dense rows of the chunks that straddle a segment boundary: value and local index (-1: another shard's amplicon)
__global__ void __launch_bounds__(256) k_alloc_bgather(const double* __restrict__ w, AllocPlan pl, const double* __restrict__ gathered,
                                                       double* __restrict__ brow, int* __restrict__ bmap) {
    const uint32_t bi = blockIdx.x;
    const unsigned long long g0 = (unsigned long long)pl.bchunk[bi].c * ALLOC_CHUNK;
    for (uint32_t e = threadIdx.x; e < ALLOC_CHUNK; e += blockDim.x) {
        double v = 0; int mp = -1;
        if (e < pl.bchunk[bi].n) {
            const unsigned long long gi = g0 + e;
            uint32_t lo = 0, hi = pl.n_gseg;                                          // last segment with go <= gi
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pl.gseg[mid].go <= gi) lo = mid; else hi = mid; }
            const AllocGSeg sg = pl.gseg[lo];
            const uint32_t j = (uint32_t)(gi - sg.go);
            if (sg.owner == pl.rank) { mp = (int)(sg.lo + j); v = w[sg.lo + j]; }
            else {
                const double* src = gathered + ((size_t)sg.owner * ALLOC_SLOTS + sg.slot) * 2 * ALLOC_CHUNK;
                v = j < ALLOC_CHUNK ? src[j] : src[ALLOC_CHUNK + (j - (sg.n - ALLOC_CHUNK))];   // a boundary chunk touches only the first / last 1000 of a foreign segment
            }
        }
        brow[(size_t)bi * ALLOC_CHUNK + e] = v; bmap[(size_t)bi * ALLOC_CHUNK + e] = mp;
    }
}
// tree1000 over every work chunk: one wave per chunk (four per workgroup).  The owner writes the chunk's sum.
__global__ void __launch_bounds__(256) k_alloc_chunk_sum(const double* __restrict__ w, const double* __restrict__ brow, AllocPlan pl,
                                                         double* __restrict__ part) {
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= pl.n_interior + pl.n_boundary) return;
    const ChunkRef r = chunk_ref(pl, q);
    const double* __restrict__ v = r.brow < 0 ? w + r.local0 : brow + (size_t)r.brow * ALLOC_CHUNK;
    double acc = 0;
    for (uint32_t i = lane; i < r.n; i += WAVE) acc += v[i];
    acc = wave_butterfly_f64(acc);
    if (lane == 0 && r.owner) part[r.c] = acc;
}
// tree1000 over consecutive groups of 1000 of an array (the levels of tree_sum above the chunks)
__global__ void __launch_bounds__(256) k_tree1000(const double* __restrict__ in, uint32_t n, double* __restrict__ out) {
    const uint32_t gq = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t b = gq * ALLOC_CHUNK;
    if (b >= n) return;
    const uint32_t m = min(ALLOC_CHUNK, n - b);
    double acc = 0;
    for (uint32_t i = lane; i < m; i += WAVE) acc += in[b + i];
    acc = wave_butterfly_f64(acc);
    if (lane == 0) out[gq] = acc;
}
// wls.normalize(0) + floor (Malbac.cpp:384-390) + the chunk's total probability (MyDefine.cpp:218-224), one pass:
//   p = w / (2.2204e-16 + total) in place, readNumbers = trunc(p * reads) for this shard's amplicons,
//   tp[c] = tree1000(p) (owner), crn[q] = sum of the read numbers set by this chunk
__global__ void __launch_bounds__(256) k_alloc_norm(double* __restrict__ w, double* __restrict__ brow, const int* __restrict__ bmap, AllocPlan pl,
                                                    const double* __restrict__ total, unsigned long long reads, uint32_t* __restrict__ rn,
                                                    double* __restrict__ tp, uint32_t* __restrict__ crn) {
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= pl.n_interior + pl.n_boundary) return;
    const ChunkRef r = chunk_ref(pl, q);
    double* __restrict__ v = r.brow < 0 ? w + r.local0 : brow + (size_t)r.brow * ALLOC_CHUNK;
    const int* __restrict__ mp = r.brow < 0 ? nullptr : bmap + (size_t)r.brow * ALLOC_CHUNK;
    const double den = 2.2204e-16 + *total;
    double acc = 0; uint32_t cs = 0;
    for (uint32_t i = lane; i < r.n; i += WAVE) {
        const double p = v[i] / den;
        v[i] = p; acc += p;
        const uint32_t c = (uint32_t)(p * (double)(long long)reads);               // unsigned readCount = wls.get(0,i)*reads
        if (mp) { const int li = mp[i]; if (li >= 0) { rn[li] = c; cs += c; } }
        else { rn[r.local0 + i] = c; cs += c; }
    }
    acc = wave_butterfly_f64(acc);
    cs = wave_sum(cs);
    if (lane == 0) { if (r.owner) tp[r.c] = acc; crn[q] = cs; }
}
// sum of a u32 array into *dst (one workgroup; the arrays are per-chunk partials)
__global__ void __launch_bounds__(1024) k_sum_u32(const uint32_t* __restrict__ v, uint32_t n, unsigned long long* __restrict__ dst, int add) {
    __shared__ unsigned long long s_p[16];
    unsigned long long acc = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
    acc = wave_sum_u64(acc);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int k = 0; k < 16; ++k) t += s_p[k]; *dst = add ? *dst + t : t; }
}
// per-chunk quota of the residual reads: unsigned(tp * n) (MyDefine.cpp:225-227), for every chunk of the whole job
__global__ void __launch_bounds__(256) k_alloc_quota(const double* __restrict__ tp, uint32_t nch, unsigned long long reads, const unsigned long long* __restrict__ sum_rn,
                                                     uint32_t* __restrict__ quota) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    const unsigned long long nres = reads - *sum_rn;
    quota[c] = (uint32_t)(tp[c] * (double)nres);
}
// scan_all, level 0: scan1000 inside every group of 1000 of `in`; last[g] = the group's last entry
__global__ void __launch_bounds__(64) k_scan1000(const double* __restrict__ in, uint32_t n, double* __restrict__ out, double* __restrict__ last) {
    const uint32_t g = blockIdx.x, b = g * ALLOC_CHUNK, m = min(ALLOC_CHUNK, n - b), lane = threadIdx.x;
    double carry = 0;
    for (uint32_t r = 0; r < m; r += WAVE) {
        double s = r + lane < m ? in[b + r + lane] : 0.0;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const double t = shfl_up_f64(s, d); if ((int)lane >= d) s = s + t; }
        if (r + lane < m) out[b + r + lane] = carry + s;
        carry = carry + shfl_f64(s, 63);
    }
    if (lane == 0 && last) last[g] = carry;                                         // == out[b + m - 1]: zeros beyond m add nothing
}
// scan_all, fix-up: out[i] = pre[g-1] + out[i] for the groups g >= 1
__global__ void __launch_bounds__(256) k_scan_fix(double* __restrict__ out, uint32_t n, const double* __restrict__ pre) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i < ALLOC_CHUNK) return;
    out[i] = pre[i / ALLOC_CHUNK - 1] + out[i];
}
__device__ __forceinline__ uint32_t first_le(const double* __restrict__ cdf, uint32_t n, double r) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (r <= cdf[mid]) hi = mid; else lo = mid + 1; }
    return lo < n ? lo : n - 1;
}
__global__ void k_alloc_top_draws(const double* __restrict__ probs, uint32_t nch, unsigned long long reads, const unsigned long long* __restrict__ sum_rn,
                                  const unsigned long long* __restrict__ sum_quota, RngKey key, uint32_t* __restrict__ quota) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long n = reads - *sum_rn - *sum_quota;                           // leftover after the per-chunk quotas (< nch)
    if (t >= n) return;
    const U4 d = draw4(key, ST_ALLOC_TOP, 0, 0, t);
    const double r = 2.2204e-16 + (1 - 2.2204e-16) * ((double)d.w[0] / 4294967296.0);
    atomicAdd(&quota[first_le(probs, nch, r)], 1u);
}
// batchSampling (MyDefine.cpp:191-201) of one chunk: the chunk-local CDF (scan1000 of p / tp) lives in LDS only; the
// chunk's quota of draws is counted in LDS and added to this shard's read numbers in one coalesced pass
__global__ void __launch_bounds__(256) k_alloc_sample(const double* __restrict__ w, const double* __restrict__ brow, const int* __restrict__ bmap, AllocPlan pl,
                                                      const double* __restrict__ tp, const uint32_t* __restrict__ quota, RngKey key, uint32_t* __restrict__ rn) {
    constexpr int ROUNDS = (ALLOC_CHUNK + WAVE - 1) / WAVE, WAVES = 4, MINE = ROUNDS / WAVES;   // four waves per chunk: with one, its 12 KB of LDS left a CU 13 waves
    static_assert(ROUNDS % WAVES == 0, "rounds of scan1000 divide among the waves");
    __shared__ double s_cdf[ALLOC_CHUNK];
    __shared__ uint32_t s_cnt[ALLOC_CHUNK];
    __shared__ double s_tot[ROUNDS];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const ChunkRef r = chunk_ref(pl, q);
    const uint32_t nq = quota[r.c];
    if (nq == 0) return;
    const double* __restrict__ v = r.brow < 0 ? w + r.local0 : brow + (size_t)r.brow * ALLOC_CHUNK;
    const double t = tp[r.c];
    // scan1000 = a wave-wide scan inside every round of 64 + the running sum of the rounds' totals in front of it: the rounds are
    // scanned by the four waves side by side (all their loads in flight together), the carries added afterwards -- the same
    // additions in the same order as the one-wave form
    double pv[MINE];
#pragma unroll
    for (int m = 0; m < MINE; ++m) { const uint32_t i = (uint32_t)((wv + WAVES * m) * WAVE) + lane; pv[m] = i < r.n ? v[i] : 0.0; }
#pragma unroll
    for (int m = 0; m < MINE; ++m) {
        const uint32_t k = wv + WAVES * m, i = k * WAVE + lane;
        double s = i < r.n ? pv[m] / t : 0.0;                                          // p[i]/totalProb (MyDefine.cpp:224)
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const double u = shfl_up_f64(s, d); if ((int)lane >= d) s = s + u; }
        pv[m] = s;
        if (lane == 63) s_tot[k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MINE; ++m) {
        const uint32_t k = wv + WAVES * m, i = k * WAVE + lane;
        double carry = 0;
        for (uint32_t z = 0; z < k; ++z) carry = carry + s_tot[z];
        if (i < r.n) { s_cdf[i] = carry + pv[m]; s_cnt[i] = 0; }
    }
    __syncthreads();
    // [REMAP] draw k of the chunk = word k & 3 of Philox block k >> 2: a lane takes a whole block, and its four bisections run
    // interleaved (each is ten dependent LDS reads: four in flight instead of one)
    for (uint32_t j = tid; 4u * j < nq; j += blockDim.x) {
        const U4 d = draw4(key, ST_ALLOC_CHUNK, 0, r.c, j);
        double x[4]; uint32_t lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = 2.2204e-16 + (1 - 2.2204e-16) * ((double)d.w[i] / 4294967296.0); lo[i] = 0; hi[i] = r.n; }
        for (uint32_t span = r.n; span; span >>= 1) {                                  // ceil(log2(n + 1)) rounds settle every one (first_le)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lo[i] < hi[i]) { const uint32_t mid = (lo[i] + hi[i]) >> 1; if (x[i] <= s_cdf[mid]) hi[i] = mid; else lo[i] = mid + 1; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4u * j + (uint32_t)i < nq) atomicAdd(&s_cnt[lo[i] < r.n ? lo[i] : r.n - 1], 1u);
    }
    __syncthreads();
    for (uint32_t i = tid; i < r.n; i += blockDim.x) {
        const uint32_t c = s_cnt[i];
        if (!c) continue;
        if (r.brow < 0) rn[r.local0 + i] += c;
        else { const int li = bmap[(size_t)r.brow * ALLOC_CHUNK + i]; if (li >= 0) rn[li] += c; }
    }
}
// odd entries of each of this shard's list segments (from the local exclusive scan of the odd bits) -> the shard's slots
// of the whole-job table (laid out in list order: cycle, pass descending, shard)
__global__ void k_alloc_odd_counts(const uint32_t* __restrict__ odd_before, AllocPlan pl, unsigned long long* __restrict__ table) {
    const uint32_t slot = threadIdx.x;
    if (slot >= ALLOC_SLOTS) return;
    const uint32_t n = pl.my_seg[slot].n, lo = pl.my_seg[slot].lo;
    table[(size_t)pl.my_seg[slot].order] = n ? odd_before[lo + n] - odd_before[lo] : 0u;
}
// PE parity fix (Malbac.cpp:399-407): the j-th odd entry OF THE WHOLE JOB's list gets +1 for even j, -1 for odd j.
// table: odd counts of all segments in list order (summed over the shards); a thread finds its segment (<= 40), the odd
// entries before the segment and adds the local ones.
__global__ void __launch_bounds__(256) k_alloc_parity(uint32_t* __restrict__ rn, const uint32_t* __restrict__ odd_before, uint32_t ac, AllocPlan pl,
                                                      const unsigned long long* __restrict__ table) {
    __shared__ unsigned long long s_base[ALLOC_SLOTS];
    if (threadIdx.x < ALLOC_SLOTS) {
        unsigned long long b = 0;
        if (table) { const uint32_t ord = pl.my_seg[threadIdx.x].order; for (uint32_t k = 0; k < ord; ++k) b += table[k]; }
        s_base[threadIdx.x] = b;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ac) return;
    const uint32_t v = rn[i];
    if (!(v & 1u)) return;
    unsigned long long j = odd_before[i];
    if (table) {
        uint32_t k = 0;
        while (k + 1 < ALLOC_SLOTS && (pl.my_seg[k].n == 0 || i >= pl.my_seg[k].lo + pl.my_seg[k].n)) ++k;   // local segments are stored in slot order
        j = s_base[k] + (odd_before[i] - odd_before[pl.my_seg[k].lo]);
    }
    rn[i] = (j & 1ull) ? v - 1u : v + 1u;
}

struct Widen { __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; } };
// mailbox: collects scattered device scalars into one contiguous block of PINNED, DEVICE-MAPPED host memory, so the
// host reads them without a copy or a stream synchronisation: it spins on the sequence word that the post writes last
// (system-scope release).  A null source posts 0.
struct MailSrc { const void* p[16]; int w[16]; int dst[16]; int n; unsigned clear; };   // clear: bit i = zero source i after reading it
__global__ void k_mail(MailSrc m, unsigned long long* __restrict__ mail, unsigned long long seq) {
    const int i = threadIdx.x;
    if (i < m.n) {
        unsigned long long v = 0;
        if (m.p[i]) {
            if (m.w[i] == 8) { unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<void*>(m.p[i])); v = *q; if ((m.clear >> i) & 1u) *q = 0; }
            else { uint32_t* q = reinterpret_cast<uint32_t*>(const_cast<void*>(m.p[i])); v = *q; if ((m.clear >> i) & 1u) *q = 0; }
        }
        __hip_atomic_store(&mail[m.dst[i]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (i == 0 && seq) __hip_atomic_store(&mail[MAIL_SEQ_SLOT], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_mail(hipStream_t s, const void* const* srcs, const int* widths, const int* dsts, int n, unsigned clear, unsigned long long* mail, unsigned long long seq) {
    MailSrc m; m.n = n; m.clear = clear;
    for (int i = 0; i < 16; ++i) { m.p[i] = i < n ? srcs[i] : nullptr; m.w[i] = i < n ? widths[i] : 4; m.dst[i] = i < n ? dsts[i] : 0; }
    hipLaunchKernelGGL(k_mail, dim3(1), dim3(64), 0, s, m, mail, seq);
}
struct OddBit { __host__ __device__ uint32_t operator()(uint32_t v) const { return v & 1u; } };
struct HalfUp { __host__ __device__ uint32_t operator()(uint32_t v) const { return (v + 1u) >> 1; } };

// ------------------------------------------------------------------------------------------------
// launch wrappers.  Grids: >> 256 workgroups wherever the unit count allows; grid-stride kernels
// are capped at 256 CUs x 8 workgroups.
// ------------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
// launch errors are latched here and surfaced by the pipeline at its next check (take_launch_error)
static thread_local hipError_t g_launch_err = hipSuccess;
static inline void note_launch(hipError_t e) { if (e != hipSuccess && g_launch_err == hipSuccess) g_launch_err = e; }
hipError_t take_launch_error() { note_launch(hipGetLastError()); const hipError_t e = g_launch_err; g_launch_err = hipSuccess; return e; }

void launch_attach_frags(hipStream_t s, const uint8_t* g, DevFrags fr, const uint32_t* slot_off, uint32_t* slots, uint32_t* slot_tmpl,
                         uint32_t* valid, const unsigned long long* primer_cut, uint32_t* primer_delta, unsigned long long* len_part, AmplifyParams p,
                         uint32_t t_first, uint32_t t_end, int undo, const unsigned long long* t_from) {
    if (t_end <= t_first) return;
    DevAmps none{}; DevErrPool np{};
    hipLaunchKernelGGL((k_attach<true, 64>), dim3(t_end - t_first), dim3(64), 0, s, g, fr, none, 0u, np, slot_off, slots, slot_tmpl, valid, primer_cut, primer_delta, len_part, p, t_first, t_end, undo, t_from);
}
void launch_frag_len_sum(hipStream_t s, const unsigned long long* len_part, uint32_t nf, unsigned long long* len_sum) {
    if (nf) hipLaunchKernelGGL(k_sum_u64_add, dim3(std::min(cdiv(nf, 2048), 128u)), dim3(1024), 0, s, len_part, nf, len_sum);
}
void launch_poisson(hipStream_t s, DevFrags fr, DevAmps semis, uint32_t n_semis, PoissonParams p, uint32_t* budget_f, uint32_t* budget_s,
                    unsigned long long* sums, unsigned long long* part) {
    const uint32_t semi_blocks = n_semis ? cdiv((uint64_t)n_semis + 1, 256) : 0u;
    if (fr.n + semi_blocks) {
        hipLaunchKernelGGL(k_poisson, dim3(fr.n + semi_blocks), dim3(256), 0, s, fr, semis, n_semis, p, budget_f, budget_s, part);
        hipLaunchKernelGGL(k_poisson_sums, dim3(1), dim3(1024), 0, s, part, fr.n, fr.n + semi_blocks, sums);
    }
}
void launch_alloc_bpack(hipStream_t s, const double* w, const AllocPlan& pl, double* send) {
    hipLaunchKernelGGL(k_alloc_bpack, dim3(2 * ALLOC_SLOTS), dim3(256), 0, s, w, pl, send);
}
void launch_alloc_bgather(hipStream_t s, const double* w, const AllocPlan& pl, const double* gathered, double* brow, int* bmap) {
    if (pl.n_boundary) hipLaunchKernelGGL(k_alloc_bgather, dim3(pl.n_boundary), dim3(256), 0, s, w, pl, gathered, brow, bmap);
}
void launch_alloc_chunk_sum(hipStream_t s, const double* w, const double* brow, const AllocPlan& pl, double* part) {
    const uint32_t nq = pl.n_interior + pl.n_boundary;
    if (nq) hipLaunchKernelGGL(k_alloc_chunk_sum, dim3(cdiv(nq, 4)), dim3(256), 0, s, w, brow, pl, part);
}
// tree_sum above the chunk level: tree1000 over groups of 1000 until one value is left; *total receives it
void launch_tree_sum(hipStream_t s, const double* part, uint32_t nch, double* scratch, double* total) {
    if (nch == 0) { (void)hipMemsetAsync(total, 0, 8, s); return; }
    if (nch == 1) { (void)hipMemcpyAsync(total, part, 8, hipMemcpyDeviceToDevice, s); return; }   // a single chunk: its sum is the total
    const double* cur = part; uint32_t n = nch; double* nxt = scratch;
    while (n > 1) {
        const uint32_t ng = cdiv(n, ALLOC_CHUNK);
        double* dst = ng == 1 ? total : nxt;
        hipLaunchKernelGGL(k_tree1000, dim3(cdiv(ng, 4)), dim3(256), 0, s, cur, n, dst);
        cur = dst; nxt += ng; n = ng;
    }
}
void launch_alloc_norm(hipStream_t s, double* w, double* brow, const int* bmap, const AllocPlan& pl, const double* total, unsigned long long reads,
                       uint32_t* rn, double* tp, uint32_t* crn, unsigned long long* sum_rn) {
    const uint32_t nq = pl.n_interior + pl.n_boundary;
    if (nq) hipLaunchKernelGGL(k_alloc_norm, dim3(cdiv(nq, 4)), dim3(256), 0, s, w, brow, bmap, pl, total, reads, rn, tp, crn);
    hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(1024), 0, s, crn, nq, sum_rn, 0);
}
// scan_all: scan1000 inside groups of 1000, recursively over the groups' last entries, then the fix-up
static void scan_all_dev(hipStream_t s, const double* in, uint32_t n, double* out, double* scratch) {
    const uint32_t ng = cdiv(n, ALLOC_CHUNK);
    double* last = scratch; double* pre = scratch + ng;
    hipLaunchKernelGGL(k_scan1000, dim3(ng), dim3(64), 0, s, in, n, out, ng > 1 ? last : (double*)nullptr);
    if (ng == 1) return;
    scan_all_dev(s, last, ng, pre, scratch + 2 * (size_t)ng);
    hipLaunchKernelGGL(k_scan_fix, dim3(cdiv(n, 256)), dim3(256), 0, s, out, n, pre);
}
// per-chunk quotas of the residual reads, the CDF over the chunks and the leftover draws (MyDefine.cpp:225-245): every
// shard computes them for ALL chunks of the job from the all-reduced tp[] (8 B per 1000 amplicons)
void launch_alloc_quota(hipStream_t s, const double* tp, uint32_t nch, unsigned long long reads, const unsigned long long* sum_rn, unsigned long long* sum_quota,
                        uint32_t* quota, double* probs, double* scratch, RngKey key) {
    if (nch == 0) return;
    hipLaunchKernelGGL(k_alloc_quota, dim3(cdiv(nch, 256)), dim3(256), 0, s, tp, nch, reads, sum_rn, quota);
    hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(1024), 0, s, quota, nch, sum_quota, 0);
    scan_all_dev(s, tp, nch, probs, scratch);
    hipLaunchKernelGGL(k_alloc_top_draws, dim3(cdiv((uint64_t)nch + 1024, 256)), dim3(256), 0, s, probs, nch, reads, sum_rn, sum_quota, key, quota);
}
void launch_alloc_sample(hipStream_t s, const double* w, const double* brow, const int* bmap, const AllocPlan& pl, const double* tp, const uint32_t* quota, RngKey key, uint32_t* rn) {
    const uint32_t nq = pl.n_interior + pl.n_boundary;
    if (nq) hipLaunchKernelGGL(k_alloc_sample, dim3(nq), dim3(256), 0, s, w, brow, bmap, pl, tp, quota, key, rn);
}
void launch_alloc_odd_scan(hipStream_t s, const uint32_t* rn, uint32_t ac, uint32_t* odd_before, void* temp, size_t temp_bytes) {
    (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator(rn, OddBit()), odd_before, 0u, (size_t)ac + 1, rocprim::plus<uint32_t>(), s);
}
void launch_alloc_odd_counts(hipStream_t s, const uint32_t* odd_before, const AllocPlan& pl, unsigned long long* table) {
    hipLaunchKernelGGL(k_alloc_odd_counts, dim3(1), dim3(64), 0, s, odd_before, pl, table);
}
void launch_alloc_parity(hipStream_t s, uint32_t* rn, const uint32_t* odd_before, uint32_t ac, const AllocPlan& pl, const unsigned long long* table) {
    if (ac) hipLaunchKernelGGL(k_alloc_parity, dim3(cdiv(ac, 256)), dim3(256), 0, s, rn, odd_before, ac, pl, table);
}
// lanes per semi amplicon (budget ~ Poisson(6)): 4 keeps the lanes busiest when the grid fills the chip, 8 finishes a
// template in one round when the job is small and the pass is latency bound (measured: 15 vs 18 ms at 13 M semis,
// 0.35 vs 0.5 ms per step at 44 k)
static int attach_semi_group(uint32_t n_semis) {
    static const int forced = getenv("SCS_ATTACH_G") ? atoi(getenv("SCS_ATTACH_G")) : 0;   // tuning experiments
    return forced == 2 || forced == 4 || forced == 8 || forced == 16 ? forced : (n_semis >= (1u << 18) ? 4 : 8);
}
void launch_attach_semis(hipStream_t s, const uint8_t* g, DevFrags fr, DevAmps semis, uint32_t n_semis, DevErrPool spool,
                         const uint32_t* slot_off, uint32_t* slots, uint32_t* slot_tmpl, uint32_t* valid,
                         const unsigned long long* primer_cut, uint32_t* primer_delta, AmplifyParams p, uint32_t t_first, uint32_t t_end, int undo, const unsigned long long* t_from) {
    if (t_end <= t_first) return;
    const int G = attach_semi_group(n_semis); const uint32_t nt = t_end - t_first;
#define SCS_LAUNCH_ATTACH_SEMI(GG) hipLaunchKernelGGL((k_attach<false, GG>), dim3(cdiv(nt, 64 / GG)), dim3(64), 0, s, g, fr, semis, n_semis, spool, slot_off, slots, slot_tmpl, valid, \
                           primer_cut, primer_delta, (unsigned long long*)nullptr, p, t_first, t_end, undo, t_from)
    if (G == 2) SCS_LAUNCH_ATTACH_SEMI(2); else if (G == 16) SCS_LAUNCH_ATTACH_SEMI(16); else if (G == 8) SCS_LAUNCH_ATTACH_SEMI(8); else SCS_LAUNCH_ATTACH_SEMI(4);
#undef SCS_LAUNCH_ATTACH_SEMI
}
// exact primer stock (k_stock_* above; the loop is exact_stock in scs_pipeline.cpp)
void launch_stock_check(hipStream_t s, const int64_t* cnt, const uint32_t* taken, unsigned long long* cut, bool from_frag, uint32_t* eidx, uint32_t* etype, uint32_t* estart, unsigned long long* info) {
    hipLaunchKernelGGL(k_stock_check, dim3(1), dim3(1024), 0, s, cnt, taken, cut, from_frag ? 20 : 12, eidx, etype, estart, info);
}
void launch_stock_collect(hipStream_t s, const uint8_t* g, DevFrags fr, DevAmps semis, DevErrPool spool, bool from_frag, const uint32_t* slot_off, const uint32_t* slots,
                          const uint32_t* valid, const uint32_t* eidx, unsigned long long* list, unsigned long long* info, uint32_t t_first, uint32_t t_end) {
    if (t_end <= t_first) return;
    const uint32_t nt = t_end - t_first;
    if (from_frag) hipLaunchKernelGGL((k_stock_collect<true, 64>), dim3(nt), dim3(64), 0, s, g, fr, semis, spool, slot_off, slots, valid, eidx, list, info, t_first, t_end);
    else hipLaunchKernelGGL((k_stock_collect<false, 8>), dim3(cdiv(nt, 8)), dim3(64), 0, s, g, fr, semis, spool, slot_off, slots, valid, eidx, list, info, t_first, t_end);
}
size_t stock_sort_temp_bytes(size_t n) {
    size_t b = 0; (void)rocprim::radix_sort_keys(nullptr, b, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n, 0, STOCK_KEY_BITS + 16);
    return b + 256;
}
void launch_stock_sort(hipStream_t s, const unsigned long long* in, unsigned long long* out, size_t n, void* temp, size_t temp_bytes) {
    if (n) note_launch(rocprim::radix_sort_keys(temp, temp_bytes, in, out, n, 0, STOCK_KEY_BITS + 16, s));
}
void launch_stock_pick(hipStream_t s, const int64_t* cnt, const uint32_t* etype, const uint32_t* estart, uint32_t ne, const unsigned long long* sorted, unsigned long long* cut,
                       bool from_frag, unsigned long long* info) {
    if (ne) hipLaunchKernelGGL(k_stock_pick, dim3(cdiv(ne, 256)), dim3(256), 0, s, cnt, etype, estart, ne, sorted, cut, from_frag ? 20 : 12, info);
}
void launch_stock_apply(hipStream_t s, int64_t* cnt, uint32_t* gdelta, uint32_t* delta, unsigned long long* cut, uint32_t* flags) {
    hipLaunchKernelGGL(k_stock_apply, dim3(256), dim3(256), 0, s, cnt, gdelta, delta, cut, flags);
}
void launch_errs_frags(hipStream_t s, const uint8_t* g, DevGenomeIdx gx, DevFrags fr, uint32_t n_slots, const uint32_t* slot_off, const uint32_t* slots,
                       const uint32_t* slot_tmpl, const uint32_t* valid_off, DevAmps out, uint32_t out_base, DevErrPool pool, uint32_t* flags,
                       const unsigned long long* binom, AmplifyParams p, int64_t* primer_cnt, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* sums,
                       unsigned long long* semis_n) {
    if (n_slots == 0) return;
    DevAmps none{}; DevErrPool np{};
    // a riding stock update wants every primer type covered with one entry per thread: never fewer than 256 workgroups
    hipLaunchKernelGGL(k_errs<true>, dim3(primer_cnt ? std::max(cdiv(n_slots, 256), 256u) : cdiv(n_slots, 256)), dim3(256), 0, s, g, gx, fr, none, np, n_slots, slot_off, slots, slot_tmpl, valid_off, fr.n, out, out_base, pool, flags, binom, p,
                       primer_cnt, primer_delta, primer_cut, sums, semis_n);
}
// the template of every amplicon a semi pass made, in creation order (valid_off = exclusive scan of the per-template counts)
__global__ void k_expand_items(const uint32_t* __restrict__ valid_off, uint32_t n_tmpl, uint32_t* __restrict__ item_tmpl) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tmpl) return;
    const uint32_t b = valid_off[t], e = valid_off[t + 1];
    for (uint32_t k = b; k < e; ++k) item_tmpl[k] = t;
}
void launch_errs_semis(hipStream_t s, const uint8_t* g, DevGenomeIdx gx, DevFrags fr, DevAmps semis, uint32_t n_semis, DevErrPool spool, uint32_t n_slots,
                       const uint32_t* slot_off, const uint32_t* slots, const uint32_t* slot_tmpl, const uint32_t* valid_off,
                       DevAmps out, uint32_t out_base, DevErrPool pool, uint32_t* flags, const unsigned long long* binom, AmplifyParams p,
                       int64_t* primer_cnt, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* sums) {
    if (n_slots == 0) return;
    hipLaunchKernelGGL(k_expand_items, dim3(cdiv(n_semis, 256)), dim3(256), 0, s, valid_off, n_semis, const_cast<uint32_t*>(slot_tmpl));
    hipLaunchKernelGGL(k_errs<false>, dim3(primer_cnt ? std::max(cdiv(n_slots, 256), 256u) : cdiv(n_slots, 256)), dim3(256), 0, s, g, gx, fr, semis, spool, n_slots, slot_off, slots, slot_tmpl, valid_off, n_semis, out, out_base, pool, flags, binom, p,
                       primer_cnt, primer_delta, primer_cut, sums, (unsigned long long*)nullptr);
}
void launch_genome_bits(hipStream_t s, const uint8_t* g, uint64_t n, uint64_t nwords, unsigned long long* gc_bits, unsigned long long* n_bits,
                        uint32_t* gc_cnt, uint32_t* n_cnt, uint64_t* gc_pref, uint64_t* n_pref, void* temp, size_t temp_bytes, uint32_t* g2) {
    hipLaunchKernelGGL(k_genome_bits, dim3(cdiv(nwords + 1, 256)), dim3(256), 0, s, g, n, nwords, gc_bits, n_bits, gc_cnt, n_cnt, g2);
    (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator((const uint32_t*)gc_cnt, Widen()), gc_pref, (uint64_t)0, nwords + 1, rocprim::plus<uint64_t>(), s);
    (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator((const uint32_t*)n_cnt, Widen()), n_pref, (uint64_t)0, nwords + 1, rocprim::plus<uint64_t>(), s);
}
void launch_amplify_init(hipStream_t s, int64_t* primer_cnt, unsigned long long* primer_cut, int64_t copies, uint32_t* primer_delta, uint32_t* primer_gdelta, uint32_t* flags, unsigned long long* sums,
                         unsigned long long nf_all, unsigned long long frag_len_all, unsigned long long total_primers, uint32_t* pool_head_a, uint32_t* pool_head_b) {
    hipLaunchKernelGGL(k_amplify_init, dim3(256), dim3(256), 0, s, primer_cnt, primer_cut, copies, primer_delta, primer_gdelta, flags, sums, nf_all, frag_len_all, total_primers, pool_head_a, pool_head_b);
}
void launch_shard_tail(hipStream_t s, uint32_t* primer_gdelta, const unsigned long long* dsums, const uint32_t* new_semis, int with_budgets) {
    hipLaunchKernelGGL(k_shard_tail, dim3(1), dim3(64), 0, s, primer_gdelta, dsums, new_semis, with_budgets);
}
void launch_primer_update_sharded(hipStream_t s, int64_t* primer_cnt, uint32_t* primer_gdelta, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* dsums, uint32_t* flags, int with_budgets) {
    hipLaunchKernelGGL(k_primer_update_sharded, dim3(256), dim3(256), 0, s, primer_cnt, primer_gdelta, primer_delta, primer_cut, dsums, flags, with_budgets);
}
void launch_primer_update(hipStream_t s, int64_t* primer_cnt, uint32_t* primer_delta, unsigned long long* primer_cut, unsigned long long* dsums, uint32_t* flags) {
    hipLaunchKernelGGL(k_primer_update, dim3(256), dim3(256), 0, s, primer_cnt, primer_delta, primer_cut, dsums, flags);
}
void launch_weights(hipStream_t s, DevAmps fulls, uint32_t n, DevTables tb, RngKey key, uint32_t frag_size, double* w) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_weights, dim3(cdiv(n, (uint32_t)WEIGHTS_BLOCK)), dim3(WEIGHTS_BLOCK), 0, s, fulls, n, tb, key, frag_size, w);
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
// PE on one shard: the parity fix (Malbac.cpp:399-407) and the pair offsets in ONE scan.  The scanned value carries the
// odd entries so far (high word) beside the halves rn >> 1 (low word); the j-th odd entry becomes rn + 1 for even j and
// rn - 1 for odd j, so the pairs before entry i are the halves before it + the even j below its odd count -- the store
// of the scan's result at i writes pair_off[i] and the fixed rn[i] (12 bytes per amplicon instead of 28 over three passes)
using ScanCfg = rocprim::scan_config<256, 16, rocprim::block_load_method::block_load_transpose, rocprim::block_store_method::block_store_transpose, rocprim::block_scan_algorithm::using_warp_scan>;   // rocPRIM ships no tuned scan for gfx950: 4096 items per workgroup for the one scan over all amplicons (6.3 -> 4.6 ms; the reads stage's scans of 8 M entries beside other kernels are better off with the default's small tiles)
struct PackOddHalf { __host__ __device__ uint64_t operator()(uint32_t v) const { return ((uint64_t)(v & 1u) << 32) | (uint64_t)(v >> 1); } };
struct ParityOut {
    struct Ref {
        uint32_t* rn; uint32_t* pair_off; size_t ac, i;
        __device__ const Ref& operator=(uint64_t sum) const {
            const uint32_t odd = (uint32_t)(sum >> 32);
            pair_off[i] = (uint32_t)sum + ((odd + 1u) >> 1);
            if (i < ac) { const uint32_t v = rn[i]; if (v & 1u) rn[i] = (odd & 1u) ? v - 1u : v + 1u; }
            return *this;
        }
    };
    using iterator_category = std::random_access_iterator_tag; using value_type = uint64_t; using difference_type = std::ptrdiff_t;
    using pointer = void; using reference = Ref;
    uint32_t* rn; uint32_t* pair_off; size_t ac, at;
    __host__ __device__ Ref operator[](difference_type k) const { return Ref{rn, pair_off, ac, at + (size_t)k}; }
    __host__ __device__ Ref operator*() const { return Ref{rn, pair_off, ac, at}; }
    __host__ __device__ ParityOut operator+(difference_type k) const { return ParityOut{rn, pair_off, ac, at + (size_t)k}; }
    __host__ __device__ ParityOut operator-(difference_type k) const { return ParityOut{rn, pair_off, ac, at - (size_t)k}; }
    __host__ __device__ ParityOut& operator+=(difference_type k) { at += (size_t)k; return *this; }
    __host__ __device__ ParityOut& operator++() { ++at; return *this; }
    __host__ __device__ difference_type operator-(const ParityOut& o) const { return (difference_type)at - (difference_type)o.at; }
};
void launch_parity_pair_offsets(hipStream_t s, uint32_t* rn, uint32_t ac, uint32_t* pair_cnt_off, void* temp, size_t temp_bytes) {
    (void)rocprim::exclusive_scan<ScanCfg>(temp, temp_bytes, rocprim::make_transform_iterator((const uint32_t*)rn, PackOddHalf()), ParityOut{rn, pair_cnt_off, (size_t)ac, 0},
                                  (uint64_t)0, (size_t)ac + 1, rocprim::plus<uint64_t>(), s);
}
void launch_pair_offsets(hipStream_t s, const uint32_t* rn, uint32_t ac, int paired, uint32_t* pair_cnt_off, void* temp, size_t temp_bytes) {
    if (paired) (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator(rn, HalfUp()), pair_cnt_off, 0u, (size_t)ac + 1, rocprim::plus<uint32_t>(), s);
    else (void)rocprim::exclusive_scan(temp, temp_bytes, rn, pair_cnt_off, 0u, (size_t)ac + 1, rocprim::plus<uint32_t>(), s);
}
void ReadsSide::release() {
    for (int k = 0; k < 2; ++k) { if (st[k]) (void)hipStreamDestroy(st[k]); if (join[k]) (void)hipEventDestroy(join[k]); st[k] = nullptr; join[k] = nullptr; }
    if (fork) (void)hipEventDestroy(fork);
    fork = nullptr;
}
size_t reads_lds_bytes(const DevTables& tb, bool uni) {
    const size_t ring = tb.qual_k == 16 ? RingGeo<16>::SLOTS * sizeof(RingBin<16>) : tb.qual_k == 64 ? RingGeo<64>::SLOTS * sizeof(RingBin<64>) : RingGeo<128>::SLOTS * sizeof(RingBin<128>);
    const size_t ring_u = tb.qual_k == 16 ? RingGeo<16>::SLOTS * sizeof(RingBinU<16>) : tb.qual_k == 64 ? RingGeo<64>::SLOTS * sizeof(RingBinU<64>) : RingGeo<128>::SLOTS * sizeof(RingBinU<128>);
    const size_t park = (size_t)RB * 19 * 4 + 320 * 4;                              // the prologue's parked records + sort counters (pair mode)
    if (uni) return std::max(park, ring_u + (size_t)RB * uni_row_bytes((uint32_t)tb.L) + 256);   // + the head rows
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
    static const uint32_t v = (getenv("SCS_EV_REPLAY") ? 1u : 0u) | (getenv("SCS_TEST_REDO") ? 2u : 0u) | (getenv("SCS_TEST_GENERAL") ? 4u : 0u) | (getenv("SCS_TEST_NO_D1") ? 8u : 0u);   // bit 1: every event-free read with a substitution is redone (redo_read)
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
    static const bool shrink = getenv("SCS_TEST_SHRINK_OUT") != nullptr;               // tests: provoke the record-bound guard
    if (shrink) { cap1 /= 2; cap2 /= 2; }
    const uint32_t ns1 = np - nc1 - nd1, ns2 = paired ? np - nc2 - nd2 : 0u;
    uint32_t gs = cdiv(std::max(ns1, ns2), RB), gd = cdiv(std::max(nd1, paired ? nd2 : 0u), RB);
    const uint32_t gc = cdiv(std::max(nc1, paired ? nc2 : 0u), RB);
    if (tb.L > 1008) { gs = 0; gd = 0; }                                           // reads this long all sit in the general list (launch_indels); what is left in the others are holes: nothing to write
    // The three class kernels write disjoint records: the two small ones go to side streams and run BESIDE the big one (each alone
    // leaves the chip half empty through its first and last wave of workgroups); the caller's stream waits for both.
    static const bool split_env = getenv("SCS_READS_SPLIT") != nullptr, serial_env = getenv("SCS_READS_SERIAL") != nullptr;
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
// one-deletion class = its flag array and that array's scan; the event-free class takes what is left)
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
void launch_philox(hipStream_t s, const uint32_t* ctr, uint32_t n, RngKey key, uint32_t* out) {
    if (n) hipLaunchKernelGGL(k_philox, dim3(cdiv(n, 256)), dim3(256), 0, s, ctr, n, key, out);
}
void launch_detlog(hipStream_t s, const double* x, uint32_t n, double* out) {
    if (n) hipLaunchKernelGGL(k_detlog, dim3(cdiv(n, 256)), dim3(256), 0, s, x, n, out);
}

// ---- device-wide scans (rocPRIM; plumbing between the hand-written kernels) --------------------------
size_t scan_temp_bytes(size_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::exclusive_scan(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n + 1, rocprim::plus<uint32_t>());
    (void)rocprim::exclusive_scan(nullptr, b, rocprim::make_transform_iterator((const uint32_t*)nullptr, Widen()),
                                  (uint64_t*)nullptr, (uint64_t)0, n + 1, rocprim::plus<uint64_t>());
    size_t c3 = 0;
    (void)rocprim::exclusive_scan<ScanCfg>(nullptr, c3, rocprim::make_transform_iterator((const uint32_t*)nullptr, PackOddHalf()), ParityOut{nullptr, nullptr, 0, 0},
                                  (uint64_t)0, n + 1, rocprim::plus<uint64_t>());
    b = std::max(b, c3);
    return (a > b ? a : b) + 256;
}
// exclusive scan of up to two small arrays in ONE launch (one 1024-thread workgroup each; out gets n+1 entries).  The
// per-pass scans of a small job are launch-latency bound: rocPRIM's scan is two launches per array.
#define SMALL_SCAN_MAX (256u * 1024u)
__global__ void __launch_bounds__(1024) k_scan_small(const uint32_t* __restrict__ in0, uint32_t* __restrict__ out0, uint32_t n0,
                                                     const uint32_t* __restrict__ in1, uint32_t* __restrict__ out1, uint32_t n1) {
    const uint32_t* __restrict__ in = blockIdx.x ? in1 : in0; uint32_t* __restrict__ out = blockIdx.x ? out1 : out0; const uint32_t n = blockIdx.x ? n1 : n0;
    __shared__ uint32_t s_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    // wave w owns a contiguous segment (a multiple of 256 elements); tiles of 256 = one uint4 per lane, coalesced
    const uint32_t seg = (((n + 15u) / 16u) + 255u) & ~255u, lo = min(w * seg, n), hi = min(lo + seg, n);
    auto load4 = [&](uint32_t idx, uint32_t v[4]) {
        if (idx + 3u < n) { const uint4 q = *reinterpret_cast<const uint4*>(in + idx); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else for (uint32_t k = 0; k < 4; ++k) v[k] = idx + k < n ? in[idx + k] : 0u;
    };
    uint32_t sum = 0;
    for (uint32_t base = lo; base < hi; base += 256u) { uint32_t v[4]; load4(base + 4u * lane, v); sum += v[0] + v[1] + v[2] + v[3]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) s_tot[w] = sum;
    __syncthreads();
    uint32_t carry = 0, total = 0;
    for (uint32_t k = 0; k < 16; ++k) { const uint32_t t = s_tot[k]; if (k < w) carry += t; total += t; }
    for (uint32_t base = lo; base < hi; base += 256u) {
        const uint32_t idx = base + 4u * lane; uint32_t v[4]; load4(idx, v);
        const uint32_t t = v[0] + v[1] + v[2] + v[3];
        uint32_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
        uint32_t run = carry + inc - t;
        if (idx + 3u < n) { *reinterpret_cast<uint4*>(out + idx) = make_uint4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]); }
        else for (uint32_t k = 0; k < 4; ++k) { if (idx + k < n) out[idx + k] = run; run += v[k]; }
        carry += __shfl(inc, 63);
    }
    if (tid == 0) out[n] = total;
}
// NOTE: `in` must have n+1 readable entries (the last one is ignored by an exclusive scan but read).
void exclusive_scan_u32(hipStream_t s, const uint32_t* in, uint32_t* out, size_t n, void* temp, size_t temp_bytes) {
    if (n <= SMALL_SCAN_MAX) { hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, in, out, (uint32_t)n, in, out, 0u); return; }
    (void)rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), s);
}
// two independent scans (either may be empty: n == 0 still writes out[0] = 0)
void exclusive_scan_u32_pair(hipStream_t s, const uint32_t* in0, uint32_t* out0, size_t n0, const uint32_t* in1, uint32_t* out1, size_t n1, void* temp, size_t temp_bytes) {
    if (n0 <= SMALL_SCAN_MAX && n1 <= SMALL_SCAN_MAX && in1) { hipLaunchKernelGGL(k_scan_small, dim3(2), dim3(1024), 0, s, in0, out0, (uint32_t)n0, in1, out1, (uint32_t)n1); return; }
    exclusive_scan_u32(s, in0, out0, n0, temp, temp_bytes);
    if (in1) exclusive_scan_u32(s, in1, out1, n1, temp, temp_bytes);
}
// record sizes -> record offsets AND the class flags -> list positions in ONE scan: k_indels leaves the read's class in bit 31 of
// its record size; the scanned value carries the byte offset in its low 40 bits and the count of flagged reads above
struct SizeCls { __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)(v & 0x7FFFFFFFu) | ((uint64_t)(v >> 31) << OFF_BITS); } };
void exclusive_scan_sizes(hipStream_t s, const uint32_t* in, uint64_t* out, size_t n, void* temp, size_t temp_bytes) {
    (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator(in, SizeCls()), out, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), s);
}
void exclusive_scan_u32_to_u64(hipStream_t s, const uint32_t* in, uint64_t* out, size_t n, void* temp, size_t temp_bytes) {
    (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator(in, Widen()), out, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), s);
}

void phase_clock_report() {
#ifdef SCS_PHASE_CLOCK
    unsigned long long h[16] = {}, z[16] = {};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) != hipSuccess || !h[15]) return;
    static const char* nm[8] = {"lists, records, sort by sector phase", "window gather + error patch", "event words", "seed, name line, ring fill", "the walk", "tails", "deferred positions", "redo"};
    fprintf(stderr, "[phase clock] %llu workgroups of the uniform walk; mean time of thread 0 per phase (us):", h[15]);
    double tot = 0; for (int i = 0; i < 8; ++i) tot += (double)h[i];
    for (int i = 0; i < 8; ++i) fprintf(stderr, "  %s %.2f", nm[i], (double)h[i] / (double)h[15] / 100.0);
    fprintf(stderr, "  | total %.2f\n", tot / (double)h[15] / 100.0);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z);
    static unsigned long long ha[8 * 256], za[8 * 256];
    if (hipMemcpyFromSymbol(ha, HIP_SYMBOL(g_phase_att), sizeof ha) == hipSuccess) {
        for (int i = 0; i < 8; ++i) { h[i] = 0; for (int k = 0; k < 256; ++k) h[i] += ha[i * 256 + k]; }
        h[15] = h[7];
    } else h[15] = 0;
    if (h[15]) {
        static const char* an[7] = {"setup", "seeding (Philox)", "gap + decode + bitmap (the candidate)", "8-mer gather, patch, stock", "(loop tail)", "blocked test", "commit + bookkeeping"};
        fprintf(stderr, "[phase clock] k_attach<semi>: %llu waves; mean wave time per section (us):", h[15]);
        for (int i = 0; i < 7; ++i) fprintf(stderr, "  %s %.2f", an[i], (double)h[i] / (double)h[15] / 100.0);
        fprintf(stderr, "\n");
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_att), za, sizeof za);
    }
#endif
}
}  // namespace scs
