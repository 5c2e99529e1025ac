// scs_k_amplify.hip -- gfx950 (CDNA4, wave64) kernels of Malbac::amplify: primer budgets (setPrimers), primer attachment with the exact
// primer stock, the new amplicons' records.  Integer / gather work; built with -ffp-contract=off (the Poisson budgets' fp64 must
// round exactly like the CPU oracle).
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
__device__ unsigned long long g_phase_att[8 * 256];                                // [section][workgroup & 255]: the waves' sums, spread over 256 slots
#endif
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
    // (a thousand waves' atomics on one word cost the pass 13 us on a small job: only a wave that undercuts the word's last value speaks)
    if ((threadIdx.x & 63u) == 0 && left < __hip_atomic_load(&sums[DS_MIN_STOCK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&sums[DS_MIN_STOCK], left);
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

// ------------------------------------------------------------------------------------------------
// K1b  new amplicon records: one thread per attached primer.  GC content of the window from the bit
//      index (+ the parent semi's substitutions), amplification errors as K ~ Binomial(l-8, ber) and K
//      distinct positions ([REMAP] of the per-base Bernoulli loop, Fragment.cpp:97-123 /
//      Amplicon.cpp:200-226), alt-base rejection draws, packed record written at its final
//      (reference -t 1 list) position.
// ------------------------------------------------------------------------------------------------
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
        // the first four thresholds of the error count's row leave HERE, beside the template's loads (the row only needs the window's
        // length): read where they are used, behind the Philox block, they were one or two more round trips at the end of the chain
        const unsigned long long* __restrict__ Tn = binom + (size_t)(alen - 8 - (p.amp_min - 8)) * BINOM_KMAX;
        const ulonglong2 tn01 = reinterpret_cast<const ulonglong2*>(Tn)[0], tn23 = reinterpret_cast<const ulonglong2*>(Tn)[1];
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
        int gc = (int)(bit_rank_pair(gx.gc_pair, gb) - bit_rank_pair(gx.gc_pair, ga));
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
        K = (x64 >= tn01.x ? 1u : 0u) + (x64 >= tn01.y ? 1u : 0u) + (x64 >= tn23.x ? 1u : 0u) + (x64 >= tn23.y ? 1u : 0u);   // (the thresholds ascend: the count IS the first k with x64 < T[k])
        if (K == 4u) while (K < (uint32_t)BINOM_KMAX && x64 >= Tn[K]) ++K;
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
__device__ __forceinline__ void poisson_frag_block(uint32_t t, DevFrags fr, const PoissonParams& p, uint32_t* __restrict__ budget_f, unsigned long long* __restrict__ part, uint32_t* __restrict__ flags) {
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
    if (tid == 0) {
        budget_f[t] = (uint32_t)(int)x; part[blockIdx.x] = (unsigned long long)x;
        if ((unsigned long long)x >= (1ull << 20)) atomicOr(flags, (uint32_t)FLAG_KEYSPACE);   // attach_key packs a fragment's primer index into 20 bits
    }
}

// one launch for both template kinds: workgroups [0, nf) take a fragment each, the rest 256 semi amplicons each
__global__ void __launch_bounds__(256) k_poisson(DevFrags fr, DevAmps semis, uint32_t n_cap, PoissonParams p, uint32_t* __restrict__ budget_f,
                                                 uint32_t* __restrict__ budget_s, unsigned long long* __restrict__ part, uint32_t* __restrict__ flags) {
    if (blockIdx.x < fr.n) poisson_frag_block(blockIdx.x, fr, p, budget_f, part, flags);
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
//     the sequential loop's result.  Otherwise (exact_stock in scs_amplify.cpp) the over-demanded types get the key of
//     their stock-th attachment as cut (k_stock_collect, a sort, k_stock_pick) and the templates from the earliest such
//     key on are run again (k_attach with undo); what they now take elsewhere may move other cuts, so this repeats until
//     no type is over its stock and no cut type under it: at that fixed point every decision equals the sequential
//     loop's (induction over the keys), and each round extends the prefix of the list on which that holds.
// ------------------------------------------------------------------------------------------------
// (enforced: set_primers_launch refuses 2^26 local fragments, k_poisson flags a fragment budget of 2^20 -- FLAG_KEYSPACE --; a semi's budget is masked to 12 bits)
#define STOCK_KEY_BITS 46                                                           // (fragment < 2^26, primer < 2^20) or (semi < 2^32, primer < 2^12), + 1
template <bool FROM_FRAG> __device__ __forceinline__ unsigned long long attach_key(uint32_t t, uint32_t i) {
    return (((unsigned long long)t << (FROM_FRAG ? 20 : 12)) | i) + 1ull;         // never 0: cut 0 = nothing to be had
}
// which types were taken more often than they have stock (over), which cut types less (under: an earlier round's cut came too
// early -- it is lifted, and the pass is run again from where it lay).  info (zeroed by the launcher): [0] over types, [1] their
// attachments, [2] under types, [3] 2^32 - 1 - the first template to run again on account of the under types.  Nearly every call
// finds nothing: one thread per type, a handful of atomics.
__global__ void __launch_bounds__(256) k_stock_check(const int64_t* __restrict__ cnt, const uint32_t* __restrict__ taken, unsigned long long* __restrict__ cut, int key_shift,
                                                     unsigned long long* __restrict__ info) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 65536u) return;
    const int64_t c = cnt[x]; const uint32_t d = taken[x];
    if ((int64_t)d > c) { atomicAdd(&info[0], 1ull); atomicAdd(&info[1], (unsigned long long)d); }
    else { const unsigned long long q = cut[x]; if (q != 0ull && q != ~0ull && (int64_t)d < c) { atomicAdd(&info[2], 1ull); atomicMax(&info[3], 0xFFFFFFFFull - ((q - 1ull) >> key_shift)); cut[x] = ~0ull; } }
}
// one workgroup, only when the check found something: the over types numbered in type order, the start of each one's stretch in the
// sorted list of their attachments; [5] (k_stock_collect's cursor) and [6] (the first template to run again: the under types', then
// k_stock_pick's) are set here
__global__ void __launch_bounds__(1024) k_stock_list(const int64_t* __restrict__ cnt, const uint32_t* __restrict__ taken,
                                                     uint32_t* __restrict__ eidx, uint32_t* __restrict__ etype, uint32_t* __restrict__ estart, unsigned long long* __restrict__ info) {
    __shared__ uint32_t s_n[1024], s_m[1024];
    const uint32_t tid = threadIdx.x;
    uint32_t n = 0, m = 0;
    for (uint32_t k = 0; k < 64; ++k) { const uint32_t x = tid * 64 + k; const uint32_t d = taken[x]; if ((int64_t)d > cnt[x]) { ++n; m += d; } }
    s_n[tid] = n; s_m[tid] = m;
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
    if (tid == 0) { info[5] = 0; info[6] = 0xFFFFFFFFull - info[3]; }
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
// K1a' attach, semi amplicons, DENSE: one lane = one primer.  k_attach<false, 4> gives every semi amplicon four lanes and
//      walks its budget (Poisson, mean 6.5) four primers at a time: 27 of 64 lanes worked (profiles/r03n_bench_sq.csv).  Here the
//      pass's primers are numbered through -- item s = slot_off[t] + i, which is also where the attachment is recorded -- and a
//      wave takes the templates whose first item lies in [56 w, 56 w + 56) with ALL their primers, lane by lane (eight lanes of
//      slack for the last template's overhang; what does not fit, or more than the bitmap rows hold, goes to a further chunk of
//      the same wave).  The templates of a chunk are segments of the wave, delimited by head flags; the speculative evaluation
//      and the in-order commit are k_attach's, with the group masks read from the segment bounds.  k_attach_plan maps the
//      items to their templates (slot_tmpl, free until k_expand_items rewrites it) and the waves to their first template.
//      Measured (profiles/r04_attach_ab_dense_vs_groups.log, 600 Mb): 42 lanes per VALU instruction instead of 27 -- and the same
//      2.5 ms per pass: a wave's round costs what it costs whatever the number of lanes in it (1.09e9 against 1.13e9
//      wave-instructions per pass), and a round of 56 primers replaces 1.6 rounds of the sixteen groups' 104.  The per-template
//      table of attach_gap's thresholds (the gap by bisection instead of the loop the wave's slowest lane sets the length of)
//      took 38 % of the scalar and 5 % of the vector instructions out and gave 2.66 ms (its 3.5 KB of LDS cost a wave per
//      SIMD): the pass is priced by its fp64 / 32-bit-multiply instructions (Philox seeding, the 64-bit scaling of the draw,
//      the decode's division and square root), not by loop control.  Not kept.
// ------------------------------------------------------------------------------------------------
#define ATTACH_DENSE_STRIDE 56u
__global__ void __launch_bounds__(256) k_attach_plan(const uint32_t* __restrict__ slot_off, uint32_t nt, uint32_t n_slots, uint32_t n_waves, uint32_t* __restrict__ item_tmpl,
                                                     uint32_t* __restrict__ wave_first, uint32_t* __restrict__ valid) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nt) return;
    const uint32_t h = slot_off[t];                                                // (slot_off[nt] = n_slots)
    const uint32_t lo = t == 0 ? 0u : slot_off[t - 1] / ATTACH_DENSE_STRIDE + 1u;  // the waves w with slot_off[t - 1] < 56 w <= slot_off[t]: t is their first template
    const uint32_t hi = t == nt ? n_waves : min(h / ATTACH_DENSE_STRIDE, n_waves);
    for (uint32_t w = lo; w <= hi; ++w) wave_first[w] = t;
    if (t == nt) return;
    const uint32_t e = slot_off[t + 1];
    for (uint32_t k = h; k < e; ++k) item_tmpl[k] = t;
    if (e == h) valid[t] = 0;                                                      // no primer: nothing attached (no lane of the pass will say so)
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 8))) k_attach_dense(const uint8_t* __restrict__ g, DevFrags fr, DevAmps semis, DevErrPool spool,
                                               const uint32_t* __restrict__ slot_off, uint32_t* __restrict__ slots, const uint32_t* __restrict__ item_tmpl, const uint32_t* __restrict__ wave_first,
                                               uint32_t* __restrict__ valid, const unsigned long long* __restrict__ primer_cut, uint32_t* __restrict__ primer_delta,
                                               AmplifyParams p, uint32_t w_first, uint32_t n_waves, uint32_t row_words, int undo, const unsigned long long* __restrict__ t_from) {
    __shared__ uint32_t s_bits[1024];                                              // position bitmaps: 1024 / row_words rows, one per template of the chunk
    const uint32_t lane = threadIdx.x, w = w_first + blockIdx.x;
    if (w >= n_waves) return;
#ifdef SCS_PHASE_CLOCK
    constexpr bool FROM_FRAG = false;                                              // (SCS_ATT's switch)
    __shared__ unsigned long long s_att_t, s_att_acc[8];
    if (lane < 8) s_att_acc[lane] = 0;
    if (lane == 0) s_att_t = wall_clock64();
    __builtin_amdgcn_wave_barrier();
#endif
    const uint32_t t0 = wave_first[w], t1 = wave_first[w + 1];
    if (t1 <= t0) return;
    const uint32_t tf = t_from ? (uint32_t)min(*t_from, 0xFFFFFFFFull) : 0u;
    if (t1 <= tf) return;                                                          // a run-again touches the templates from *t_from on
    const uint32_t end = slot_off[t1], max_rows = 1024u / row_words, aux = 1u | (p.pass << 1);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t item0 = slot_off[t0];
    bool carry_open = false, carry_abort = false; uint32_t carry_v = 0;            // the template cut by the previous chunk's end: still open? aborted? its count so far
    while (item0 < end) {                                                          // (wave-uniform)
        const uint32_t item = item0 + lane; const bool has = item < end;
        uint32_t t = 0, i = 0, len = 0, budget = 0; uint64_t tuid = 0, errs = 0; View tv{0, 1, 0};
        if (has) {
            t = item_tmpl[item]; i = item - slot_off[t];
            const uint32_t f = semis.parent[t], sl = semis.sl[t];
            len = sl_len(sl); budget = semis.primers[t]; tuid = semis.uid[t]; errs = semis.errs[t];
            tv = semi_tmpl_view(frag_view(fr.goff[f], fr.len[f], fr.strand[f]), sl_spos(sl), len);
        }
        // segments: a lane is a head when its primer is its template's first -- or the chunk's first lane (a template cut by the
        // previous chunk).  Row = the segment's number in the chunk; the chunk ends before the first segment without a row.
        const unsigned long long hm0 = __ballot(has && (i == 0u || lane == 0u));
        const uint32_t row = (uint32_t)__popcll(hm0 & (lt_mask | (1ull << lane))) - 1u;
        const bool in = has && row < max_rows;
        const unsigned long long inm = __ballot(in), hm = hm0 & inm;
        const uint32_t n_in = (uint32_t)__popcll(inm);                               // the chunk's lanes are 0 .. n_in - 1
        const uint32_t seg_lo = 63u - (uint32_t)__builtin_clzll((hm & (lt_mask | (1ull << lane))) | 1ull);
        const unsigned long long above = hm & ~(lt_mask | (1ull << lane));
        const uint32_t seg_hi = above ? (uint32_t)__ffsll((long long)above) - 1u : n_in;   // one past my segment's last lane
        const unsigned long long gmask = in ? ((seg_hi >= 64u ? ~0ull : (1ull << seg_hi) - 1ull) & ~((1ull << seg_lo) - 1ull)) : 0ull;
        const bool cont = in && seg_lo == 0u && carry_open && __shfl((int)i, 0) != 0;   // my segment continues the template the last chunk cut
        const bool skip = in && t < tf;
        uint32_t* bits = s_bits + row * row_words;
        // the type under a position of my template (k_attach's primer_type)
        auto primer_type = [&](uint32_t sp, bool& hasN) -> uint32_t {
            unsigned long long v8 = view_bases8(g, tv, sp);
            for_each_err(errs, spool.data, [&](uint32_t e) {
                const uint32_t k = len - 1u - err_pos(e) - sp;
                if (k < 8u) v8 = (v8 & ~(0xFFull << (8u * k))) | ((unsigned long long)(3u - err_alt(e)) << (8u * k));
            });
            hasN = (v8 & 0xFCFCFCFCFCFCFCFCull) != 0;
            uint32_t idx = 0;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) idx = (idx << 2) | ((uint32_t)(v8 >> (8u * k)) & 3u);
            return idx;
        };
        if (undo && in && !skip && i < valid[t]) { bool hn; const uint32_t idx = primer_type(sl_spos(slots[item]), hn); atomicSub(&primer_delta[idx], 1u); }
        // bitmap rows: row 0 keeps the cut template's positions (moved there at the last chunk's end), the others start empty
        __builtin_amdgcn_wave_barrier();
        {   const uint32_t rows = (uint32_t)__popcll(hm), total = rows * row_words;
            const uint32_t keep = (carry_open && __shfl((int)i, 0) != 0) ? row_words : 0u;
            for (uint32_t k = lane; k < total; k += 64u) if (k >= keep) s_bits[k] = 0u;
        }
        __builtin_amdgcn_wave_barrier();
        bool unresolved = in && !skip && len >= p.amp_min + 27u && !(cont && carry_abort);
        bool need = unresolved, dead = false; uint32_t tries = 0, spos = 0, alen = 0, pidx = 0;
        const AttachFit fit = unresolved ? attach_fit_count(len, p.amp_min, p.amp_max) : AttachFit{0, 1, 0, 1};
        const double qfail = 1.0 - (double)fit.N / ((double)(len > 27 ? len - 27 : 1) * (double)fit.W);
        SCS_ATT(0);
        Xoshiro xt{}; if (unresolved) xt.seed(draw4(p.key, ST_ATTACH, aux, tuid, i));
        bool aborted = cont && carry_abort; uint32_t v_abort = cont ? carry_v : 0u; // my segment: abandoned at primer v_abort
        SCS_ATT(1);
        while (__ballot(unresolved)) {
#ifdef SCS_PHASE_CLOCK
            if (lane == 0) s_att_acc[4] += 100ull;                                  // (rounds, in the clock's unit: printed as a count)
#endif
            if (unresolved && !dead) {
                const uint32_t bp = spos - 27u;
                if (!need && ((bits[bp >> 5] >> (bp & 31u)) & 1u)) need = true;       // a lower primer took this position meanwhile
                while (need) {
                    bool cand = false;
                    while (!cand) {
                        tries += attach_gap(((double)xt.next() + 0.5) / 4294967296.0, qfail) + 1u;
                        if (tries > 50) { dead = true; break; }
                        const unsigned long long x64 = ((unsigned long long)xt.next() << 32) | xt.next();
                        attach_fit_decode(fit, len, p.amp_min, (uint32_t)__umul64hi(x64, (unsigned long long)fit.N), spos, alen);
                        const uint32_t b2 = spos - 27u;
                        if ((bits[b2 >> 5] >> (b2 & 31u)) & 1u) continue;              // posAttached[spos]
                        cand = true;
                    }
                    if (dead) break;
                    SCS_ATT(2);
                    bool hasN; const uint32_t idx = primer_type(spos, hasN);
                    const bool nostock_ = hasN || attach_key<false>(t, i) > primer_cut[idx];
                    SCS_ATT(3);
                    if (nostock_) continue;                                          // no stock (none for N 8-mers): the try counts, the next one follows
                    pidx = idx; need = false;
                }
            }
            // blocked = a lower unresolved live lane of my segment proposes the same position
            SCS_ATT(2);
            const bool live = unresolved && !dead;
            const unsigned long long um = __ballot(live);
            bool blocked = false;
            for (uint32_t d = 1; __ballot(in && lane >= seg_lo + d); ++d) {
                const uint32_t src = lane >= d ? lane - d : 0u;
                const uint32_t sj = (uint32_t)__shfl((int)spos, (int)src);
                if (live && lane >= seg_lo + d && ((um >> src) & 1ull) && sj == spos) blocked = true;
            }
            const unsigned long long bm = __ballot(live && blocked) & gmask, dm = __ballot(unresolved && dead) & gmask;
            const uint32_t first_blocked = bm ? (uint32_t)__ffsll((long long)bm) - 1u : 64u, first_dead = dm ? (uint32_t)__ffsll((long long)dm) - 1u : 64u;
            SCS_ATT(5);
            if (live && lane < first_blocked && lane < first_dead) {                  // commit, in primer order
                const uint32_t bp = spos - 27u;
                atomicOr(&bits[bp >> 5], 1u << (bp & 31u));
                atomicAdd(&primer_delta[pidx], 1u);
                slots[item] = pack_sl(spos, alen);
                unresolved = false;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t vi = (uint32_t)__shfl((int)i, (int)(first_dead < 64u ? first_dead : lane));
            if (first_dead < 64u && first_dead <= first_blocked) {                     // the template abandons its remaining primers (> 50 tries)
                if (in && !aborted) { aborted = true; v_abort = vi; }
                if (lane >= first_dead) unresolved = false;
            }
            SCS_ATT(6);
        }
        // what my template attached: its budget, or the primer it was abandoned at.  A template cut by the chunk's end waits.
        const uint32_t last = n_in - 1u;                                              // the chunk's last lane
        const uint32_t l_t = (uint32_t)__shfl((int)t, (int)last), l_i = (uint32_t)__shfl((int)i, (int)last), l_b = (uint32_t)__shfl((int)budget, (int)last);
        const bool open_next = l_i + 1u < l_b;                                         // the last segment's template has primers beyond the chunk
        const bool mine_open = in && open_next && t == l_t && seg_hi == n_in;
        const bool ok_len = len >= p.amp_min + 27u;
        if (in && !skip && lane == seg_lo && !mine_open) valid[t] = aborted ? v_abort : (ok_len ? budget : 0u);
        // carry the cut template over: abort state, and its bitmap row to row 0
        const uint32_t l_row = (uint32_t)__shfl((int)row, (int)last);
        const bool l_ab = __shfl((int)(aborted ? 1 : 0), (int)last) != 0; const uint32_t l_v = (uint32_t)__shfl((int)v_abort, (int)last);
        __builtin_amdgcn_wave_barrier();
        if (open_next && l_row != 0u) { for (uint32_t k = lane; k < row_words; k += 64u) s_bits[k] = s_bits[l_row * row_words + k]; }
        __builtin_amdgcn_wave_barrier();
        carry_open = open_next; carry_abort = l_ab; carry_v = l_v;
        item0 += n_in;
        SCS_ATT(0);
    }
#ifdef SCS_PHASE_CLOCK
    __builtin_amdgcn_wave_barrier();
    if (lane < 8) atomicAdd(&g_phase_att[lane * 256 + (blockIdx.x & 255u)], lane == 7 ? 1ull : s_att_acc[lane]);
#endif
}

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
                    unsigned long long* sums, unsigned long long* part, uint32_t* flags) {
    const uint32_t semi_blocks = n_semis ? cdiv((uint64_t)n_semis + 1, 256) : 0u;
    if (fr.n + semi_blocks) {
        hipLaunchKernelGGL(k_poisson, dim3(fr.n + semi_blocks), dim3(256), 0, s, fr, semis, n_semis, p, budget_f, budget_s, part, flags);
        hipLaunchKernelGGL(k_poisson_sums, dim3(1), dim3(1024), 0, s, part, fr.n, fr.n + semi_blocks, sums);
    }
}
// lanes per semi amplicon (budget ~ Poisson(6)): 4 keeps the lanes busiest when the grid fills the chip, 8 finishes a
// template in one round when the job is small and the pass is latency bound (measured: 15 vs 18 ms at 13 M semis,
// 0.35 vs 0.5 ms per step at 44 k)
static int attach_semi_group(uint32_t n_semis) {
    static const int forced = seam_env("SCS_ATTACH_G") ? atoi(seam_env("SCS_ATTACH_G")) : 0;   // tuning experiments
    return forced == 2 || forced == 4 || forced == 8 || forced == 16 ? forced : (n_semis >= (1u << 18) ? 4 : 8);
}
uint32_t attach_dense_waves(uint32_t n_slots) { return n_slots / ATTACH_DENSE_STRIDE + 1u; }
// once per pass: items -> templates (item_tmpl: n_slots words), waves -> their first template (wave_first: attach_dense_waves + 1 words)
void launch_attach_plan(hipStream_t s, const uint32_t* slot_off, uint32_t nt, uint32_t n_slots, uint32_t* item_tmpl, uint32_t* wave_first, uint32_t* valid) {
    hipLaunchKernelGGL(k_attach_plan, dim3(cdiv((uint64_t)nt + 1, 256)), dim3(256), 0, s, slot_off, nt, n_slots, attach_dense_waves(n_slots), item_tmpl, wave_first, valid);
}
void launch_attach_dense(hipStream_t s, const uint8_t* g, DevFrags fr, DevAmps semis, DevErrPool spool, const uint32_t* slot_off, uint32_t* slots, const uint32_t* item_tmpl,
                         const uint32_t* wave_first, uint32_t n_slots, uint32_t* valid, const unsigned long long* primer_cut, uint32_t* primer_delta, AmplifyParams p, int undo,
                         const unsigned long long* t_from) {
    // a bitmap row holds the positions 27 .. len - amp_min of a template (len <= amp_max)
    const uint32_t row_words = (p.amp_max > p.amp_min ? (p.amp_max - p.amp_min) / 32u : 0u) + 2u;
    hipLaunchKernelGGL(k_attach_dense, dim3(attach_dense_waves(n_slots)), dim3(64), 0, s, g, fr, semis, spool, slot_off, slots, item_tmpl, wave_first, valid, primer_cut, primer_delta, p, 0u,
                       attach_dense_waves(n_slots), std::min(row_words, 1024u), undo, t_from);
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
// exact primer stock (k_stock_* above; the loop is exact_stock in scs_amplify.cpp)
void launch_stock_check(hipStream_t s, const int64_t* cnt, const uint32_t* taken, unsigned long long* cut, bool from_frag, unsigned long long* info) {
    (void)hipMemsetAsync(info, 0, 64, s);
    hipLaunchKernelGGL(k_stock_check, dim3(256), dim3(256), 0, s, cnt, taken, cut, from_frag ? 20 : 12, info);
}
void launch_stock_list(hipStream_t s, const int64_t* cnt, const uint32_t* taken, uint32_t* eidx, uint32_t* etype, uint32_t* estart, unsigned long long* info) {
    hipLaunchKernelGGL(k_stock_list, dim3(1), dim3(1024), 0, s, cnt, taken, eidx, etype, estart, info);
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
void phase_clock_report_attach() {
#ifdef SCS_PHASE_CLOCK
    unsigned long long h[16] = {};
    static unsigned long long ha[8 * 256], za[8 * 256];
    if (hipMemcpyFromSymbol(ha, HIP_SYMBOL(g_phase_att), sizeof ha) == hipSuccess) {
        for (int i = 0; i < 8; ++i) { h[i] = 0; for (int k = 0; k < 256; ++k) h[i] += ha[i * 256 + k]; }
        h[15] = h[7];
    } else h[15] = 0;
    if (h[15]) {
        static const char* an[7] = {"setup + chunk end", "seeding (Philox)", "gap + decode + bitmap (the candidate)", "8-mer gather, patch, stock", "ROUNDS (a count)", "blocked test", "commit + bookkeeping"};
        fprintf(stderr, "[phase clock] k_attach<semi>: %llu waves; mean wave time per section (us):", h[15]);
        for (int i = 0; i < 7; ++i) fprintf(stderr, "  %s %.2f", an[i], (double)h[i] / (double)h[15] / 100.0);
        fprintf(stderr, "\n");
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_att), za, sizeof za);
    }
#endif
}
}  // namespace scs
