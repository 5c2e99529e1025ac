// scs_k_allocate.hip -- gfx950 kernels of Malbac::setReadCounts: GC-bias weights, the chunked multinomial allocation over the whole
// job's amplicon list (fixed-shape sums and scans), the parity fix and the pair offsets.  fp64 with -ffp-contract=off.
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

struct OddBit { __host__ __device__ uint32_t operator()(uint32_t v) const { return v & 1u; } };
struct HalfUp { __host__ __device__ uint32_t operator()(uint32_t v) const { return (v + 1u) >> 1; } };

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
void launch_weights(hipStream_t s, DevAmps fulls, uint32_t n, DevTables tb, RngKey key, uint32_t frag_size, double* w) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_weights, dim3(cdiv(n, (uint32_t)WEIGHTS_BLOCK)), dim3(WEIGHTS_BLOCK), 0, s, fulls, n, tb, key, frag_size, w);
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
size_t parity_scan_temp_bytes(size_t n) {
    size_t c3 = 0;
    (void)rocprim::exclusive_scan<ScanCfg>(nullptr, c3, rocprim::make_transform_iterator((const uint32_t*)nullptr, PackOddHalf()), ParityOut{nullptr, nullptr, 0, 0},
                                  (uint64_t)0, n + 1, rocprim::plus<uint64_t>());
    return c3;
}
void launch_parity_pair_offsets(hipStream_t s, uint32_t* rn, uint32_t ac, uint32_t* pair_cnt_off, void* temp, size_t temp_bytes) {
    (void)rocprim::exclusive_scan<ScanCfg>(temp, temp_bytes, rocprim::make_transform_iterator((const uint32_t*)rn, PackOddHalf()), ParityOut{rn, pair_cnt_off, (size_t)ac, 0},
                                  (uint64_t)0, (size_t)ac + 1, rocprim::plus<uint64_t>(), s);
}
void launch_pair_offsets(hipStream_t s, const uint32_t* rn, uint32_t ac, int paired, uint32_t* pair_cnt_off, void* temp, size_t temp_bytes) {
    if (paired) (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator(rn, HalfUp()), pair_cnt_off, 0u, (size_t)ac + 1, rocprim::plus<uint32_t>(), s);
    else (void)rocprim::exclusive_scan(temp, temp_bytes, rn, pair_cnt_off, 0u, (size_t)ac + 1, rocprim::plus<uint32_t>(), s);
}
}  // namespace scs
