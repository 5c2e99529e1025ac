// scs_k_staging.hip -- gfx950 kernels of the input side: base codes, the FASTA parsed on the device, simuvars' haplotype builder,
// the genome's bit index and two-bit copy.  HBM-bound streaming work.
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
// *ragged is raised when a byte taken for a base is a line end or a header mark: the index's line geometry does not hold for this
// stretch (a longer and a shorter line that cancel out, a blank line) -- the caller then stages the whole file with the parser
__global__ void __launch_bounds__(256) k_fa_gather_regular(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst, uint64_t n, uint32_t col0, uint32_t lb, uint32_t lw, uint32_t* __restrict__ ragged) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint8_t b = raw[j + ((uint64_t)col0 + j) / lb * (uint64_t)(lw - lb)];
    dst[j] = b;
    if (b == '\n' || b == '\r' || b == '>') *ragged = 1u;
}
void launch_fa_gather_regular(hipStream_t s, const uint8_t* raw, uint8_t* dst, uint64_t n, uint32_t col0, uint32_t lb, uint32_t lw, uint32_t* ragged) {
    if (n) hipLaunchKernelGGL(k_fa_gather_regular, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, raw, dst, n, col0, lb, lw, ragged);
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
__global__ void k_frag_has_n(const uint64_t* __restrict__ goff, const uint32_t* __restrict__ len, uint32_t nf, DevGenomeIdx gx, uint8_t* __restrict__ has_n) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    has_n[f] = bit_rank(gx.n_bits, gx.n_pref, goff[f] + len[f]) != bit_rank(gx.n_bits, gx.n_pref, goff[f]) ? 1 : 0;
}
void launch_frag_has_n(hipStream_t s, const uint64_t* goff, const uint32_t* len, uint32_t nf, DevGenomeIdx gx, uint8_t* has_n) {
    if (nf) hipLaunchKernelGGL(k_frag_has_n, dim3((nf + 255) / 256), dim3(256), 0, s, goff, len, nf, gx, has_n);
}

struct Widen { __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; } };
__global__ void __launch_bounds__(256) k_genome_pair(const unsigned long long* __restrict__ bits, const uint64_t* __restrict__ pref, uint64_t nwords, ulonglong2* __restrict__ pair) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w <= nwords) pair[w] = make_ulonglong2(bits[w], pref[w]);
}
void launch_genome_bits(hipStream_t s, const uint8_t* g, uint64_t n, uint64_t nwords, unsigned long long* gc_bits, unsigned long long* n_bits,
                        uint32_t* gc_cnt, uint32_t* n_cnt, uint64_t* gc_pref, uint64_t* n_pref, void* temp, size_t temp_bytes, uint32_t* g2, ulonglong2* gc_pair) {
    hipLaunchKernelGGL(k_genome_bits, dim3(cdiv(nwords + 1, 256)), dim3(256), 0, s, g, n, nwords, gc_bits, n_bits, gc_cnt, n_cnt, g2);
    (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator((const uint32_t*)gc_cnt, Widen()), gc_pref, (uint64_t)0, nwords + 1, rocprim::plus<uint64_t>(), s);
    (void)rocprim::exclusive_scan(temp, temp_bytes, rocprim::make_transform_iterator((const uint32_t*)n_cnt, Widen()), n_pref, (uint64_t)0, nwords + 1, rocprim::plus<uint64_t>(), s);
    hipLaunchKernelGGL(k_genome_pair, dim3(cdiv(nwords + 1, 256)), dim3(256), 0, s, gc_bits, gc_pref, nwords, gc_pair);
}
}  // namespace scs
