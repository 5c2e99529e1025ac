// scs_kernels_common.h -- device helpers shared by the kernel files (scs_k_*.hip): wave / block reductions, the index-map view of a
// template strand, the bit index's rank, the threshold lookups, det_log / det_exp.  Every kernel file is its own translation
// unit (no relocatable device code): what is here is inlined (or, for the two noinline routines, copied) into each.
#pragma once
namespace scs {
// Build-time diagnostic (-DSCS_PHASE_CLOCK): where a workgroup of the uniform walk spends its life.  Thread 0 reads the 100 MHz
// wall clock at the phase boundaries and adds the differences to g_phase[] (g_phase[15] counts the workgroups); the host prints and
// zeroes them (scs_phase_clock_report, called at the end of a yield when SCS_PHASE_CLOCK is set in the environment).
#ifdef SCS_PHASE_CLOCK
extern __device__ unsigned long long g_phase[16];              // (scs_k_reads.hip)
#define SCS_PHASE(i) do { if (UNI && CLS == 1 && tid == 0) { const unsigned long long now_ = wall_clock64(); if ((i) >= 0) atomicAdd(&g_phase[(i) < 0 ? 0 : (i)], now_ - ph_t_); ph_t_ = now_; } } while (0)
// the same for k_attach<semi>: wave-level sections of its loop (g_phase_att[15] counts the waves)
extern __device__ unsigned long long g_phase_att[8 * 256];                                // [section][workgroup & 255]: the waves' sums, spread over 256 slots
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
    if (ac == 0u) return 0u;                  // an empty table has no last entry (the hosts refuse such a lookup before any launch: do_yield)
    if (x == 0xFFFFFFFFu) return rand_indx_slow(cdf, ac, x);
    uint32_t lo = 0, hi = ac;                 // lower bound of "x < T[k]"
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (x < T[mid]) hi = mid; else lo = mid + 1; }
    return lo < ac ? lo : ac - 1;
}

__device__ __forceinline__ uint64_t bit_rank(const unsigned long long* __restrict__ bits, const uint64_t* __restrict__ pref, uint64_t x) {
    const uint64_t w = x >> 6; const uint32_t r = (uint32_t)(x & 63);
    return pref[w] + (uint64_t)__popcll(bits[w] & ((1ull << r) - 1ull));
}
__device__ __forceinline__ uint64_t bit_rank_pair(const ulonglong2* __restrict__ pair, uint64_t x) {   // the same from the interleaved index: one 16-byte load
    const ulonglong2 v = pair[x >> 6]; const uint32_t r = (uint32_t)(x & 63);
    return v.y + (uint64_t)__popcll(v.x & ((1ull << r) - 1ull));
}

__device__ __forceinline__ uint32_t u4_word(const U4& d, uint32_t k) { return k == 0 ? d.w[0] : k == 1 ? d.w[1] : k == 2 ? d.w[2] : d.w[3]; }
// ------------------------------------------------------------------------------------------------
// deterministic log (same operation sequence as oracle/scs_oracle.cpp det_log; IEEE + - * / only)
// ------------------------------------------------------------------------------------------------
static __device__ double det_log(double x) {
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
static __device__ double det_exp(double x) {
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


// (Round 4 tried an XCD-aware order of the workgroups for k_attach<semi> and k_errs -- every XCD a contiguous eighth of the amplicon
// list, so that its L2 holds the fragments of an eighth of the workgroups in flight: k_attach unchanged, k_errs<semi->full> 1.47 ->
// 1.92 ms per pass at 600 Mb.  With the hardware's round-robin the eight XCDs work on ONE stretch of the genome at a time and share its
// lines in the Infinity Cache; eight distant stretches at once cost more than the L2 hits bring.  profiles/r04_attach_ab_xcd.log)
static inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
// launch errors are latched (scs_k_misc.hip) and surfaced by the pipeline at its next check (take_launch_error)
void note_launch(hipError_t e);
}  // namespace scs
