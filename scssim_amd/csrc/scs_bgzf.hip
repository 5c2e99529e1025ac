// scs_bgzf.hip -- BGZF (blocked gzip, the format `bgzip` writes and htslib / samtools / bwa read) made ON the GPU from a batch's
// FASTQ text where it lies in HBM, so that 3-4x fewer bytes cross PCIe and reach the file system (the sink is
// the wall of a `genreads` job by 5x (PCIe) to 30x (one writer per file); the reference writes plain text only, SeqWriter.cpp).
//
// One workgroup per BGZF block of BGZF_IN input bytes, two kernels with a prefix sum of the block sizes between them:
//   k_bgzf_plan : byte histogram (LDS, four copies per wave against same-address conflicts) -> Huffman code lengths of the
//                 literals + end-of-block (two-queue merge over the rank-sorted symbols, limited to 15 bits the way zlib's
//                 gen_bitlen does) -> the exact size of the block (header + one dynamic-Huffman deflate block of literals)
//   k_bgzf_emit : canonical codes from the lengths, the deflate header (code lengths run-length coded with a fixed, complete
//                 code-length code), every thread encodes its 252 input bytes at the bit offset a block-wide prefix sum of the
//                 chunks' bit counts gives it (LDS, atomicOr), CRC-32 of the input (a chunk per thread, combined by carry-less
//                 multiplication with x^(8 * bytes behind the chunk) mod P), trailer; the block is assembled in LDS and copied
//                 to its final byte offset.
// FASTQ text has no long repeats worth an LZ77 search (bases and qualities are fresh draws; only the 12-byte names repeat), so
// the deflate stream is literals only: 2 bits per base, the qualities at their entropy.  A block that would not fit the LDS
// staging (text this never happens to) is written as a stored deflate block instead.
// The checker is zlib: tests inflate the files with Python's gzip / zlib and compare with the oracle's text.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#include "scs_bgzf.h"

namespace scs {

namespace {

constexpr uint32_t NSYM = 257;                            // literals 0..255 + end-of-block (256)
constexpr uint32_t CHUNK = BGZF_IN / 256;                 // input bytes per thread of the emit kernel: 252
static_assert(BGZF_IN % 256 == 0 && CHUNK % 4 == 0, "a whole number of dwords per thread");

// the fixed code-length code: 13 symbols of 4 bits, 6 of 5 bits (complete: 13/16 + 6/32 = 1); canonical codes in symbol order
// (constexpr tables: usable from host and device code alike -- the host emulation below runs the same functions)
static constexpr uint8_t kClLen[19] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 4, 4};
static constexpr uint8_t kClCode[19] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 26, 27, 28, 29, 30, 31, 11, 12};
static constexpr uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__host__ __device__ inline uint32_t rev_bits(uint32_t v, uint32_t n) { return __builtin_bitreverse32(v) >> (32u - n); }

// CRC-32 (reflected, polynomial 0xEDB88320): a(x) * b(x) mod P in the reflected representation (x^0 = 0x80000000)
__host__ __device__ inline uint32_t crc_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) { if (a & m) p ^= b; b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}

struct BitSink {                                          // thread-private writer into a byte buffer the thread owns alone (the header)
    uint8_t* out; uint64_t acc; uint32_t n, pos;          // pos: bytes written
    __host__ __device__ void put(uint32_t v, uint32_t bits) { acc |= (uint64_t)v << n; n += bits; while (n >= 8) { out[pos++] = (uint8_t)acc; acc >>= 8; n -= 8; } }
};
struct BitCount { uint32_t bits; __host__ __device__ void put(uint32_t, uint32_t b) { bits += b; } };

// the code lengths of the 257 literal/length symbols + one distance code of length 0, run-length coded (RFC 1951 3.2.7)
template <class W>
__host__ __device__ void write_lengths(const uint8_t* len, W& w) {
    auto sym = [&](uint32_t s, uint32_t extra, uint32_t ebits) { w.put(rev_bits(kClCode[s], kClLen[s]), kClLen[s]); if (ebits) w.put(extra, ebits); };
    const uint32_t total = NSYM + 1;
    for (uint32_t i = 0; i < total;) {
        const uint32_t v = i < NSYM ? len[i] : 0u;
        uint32_t run = 1; while (i + run < total && (i + run < NSYM ? len[i + run] : 0u) == v) ++run;
        i += run;
        if (v == 0) {
            while (run >= 11) { const uint32_t r = run < 138u ? run : 138u; sym(18, r - 11u, 7); run -= r; }
            if (run >= 3) { sym(17, run - 3u, 3); run = 0; }
            while (run--) sym(0, 0, 0);
        } else {
            sym(v, 0, 0); --run;
            while (run >= 3) { const uint32_t r = run < 6u ? run : 6u; sym(16, r - 3u, 2); run -= r; }
            while (run--) sym(v, 0, 0);
        }
    }
}
template <class W>
__host__ __device__ void write_deflate_header(const uint8_t* len, W& w) {
    w.put(1, 1); w.put(2, 2);                             // BFINAL, BTYPE = dynamic Huffman
    w.put(0, 5); w.put(0, 5); w.put(15, 4);               // HLIT = 257, HDIST = 1, HCLEN = 19
    for (int k = 0; k < 19; ++k) w.put(kClLen[kClOrder[k]], 3);
    write_lengths(len, w);
}

// Huffman code lengths (at most 15 bits) of the m used symbols order[0..m) (ascending frequency); w / par: 2 m scratch entries
// returns whether the lengths form a complete prefix code (Kraft sum exactly 1; the callers store the block otherwise)
__host__ __device__ inline bool huff_lengths(const uint32_t* freq, const uint16_t* order, uint32_t m, uint32_t* w, uint16_t* par, uint8_t* len) {
    // the tree by the two-queue method: leaves 0..m-1 in ascending weight, internal nodes m.. in creation (= ascending) order
    for (uint32_t i = 0; i < m; ++i) w[i] = freq[order[i]];
    uint32_t li = 0, ni = m, nn = m;
    for (uint32_t k = 0; k + 1 < m; ++k) {
        uint32_t pick[2];
        for (int q = 0; q < 2; ++q) { if (li < m && (ni >= nn || w[li] <= w[ni])) pick[q] = li++; else pick[q] = ni++; }
        w[nn] = w[pick[0]] + w[pick[1]]; par[pick[0]] = (uint16_t)nn; par[pick[1]] = (uint16_t)nn; ++nn;
    }
    // depths from the root down (node nn - 1); bl_count with the 15-bit limit, overflow repaired as zlib's gen_bitlen does
    uint32_t bl[17]; for (int i = 0; i < 17; ++i) bl[i] = 0;
    w[nn - 1] = 0;                                        // the weights are spent: the array now holds depths
    int overflow = 0;
    // (zlib counts EVERY node deeper than the limit, internal ones included: a subtree of L leaves hanging at depth 15 needs L - 1
    // repair rounds, not L / 2 -- counting the leaves alone left the code oversubscribed for trees deeper than 17)
    for (int i = (int)nn - 2; i >= 0; --i) { uint32_t d = w[par[i]] + 1u; if (d > 15u) { d = 15u; ++overflow; } w[i] = d; if ((uint32_t)i < m) ++bl[d]; }
    if (m == 1) bl[1] = 1;                                // (a lone symbol would still need one bit; the callers always have two)
    while (overflow > 0) { uint32_t bits = 14; while (bl[bits] == 0) --bits; --bl[bits]; bl[bits + 1] += 2; --bl[15]; overflow -= 2; }
    uint32_t idx = 0;                                     // the longest codes go to the rarest symbols
    for (uint32_t bits = 15; bits >= 1; --bits) for (uint32_t c = bl[bits]; c; --c) len[order[idx++]] = (uint8_t)bits;
    uint32_t kraft = 0; for (uint32_t bits = 1; bits <= 15; ++bits) kraft += bl[bits] << (15u - bits);
    return kraft == (1u << 15) || m == 1;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ plan
__global__ void __launch_bounds__(256) k_bgzf_plan(const uint8_t* __restrict__ text, uint64_t nbytes, uint8_t* __restrict__ plans, uint32_t* __restrict__ sizes) {
    __shared__ uint32_t s_hist[4][4][256];
    __shared__ uint32_t s_freq[NSYM + 3];
    __shared__ uint16_t s_order[NSYM + 3];                // symbols by ascending (frequency, symbol); unused ones are not in it
    __shared__ uint32_t s_w[2 * NSYM]; __shared__ uint16_t s_par[2 * NSYM];
    __shared__ uint8_t s_len[BGZF_PLAN_BYTES];
    __shared__ uint32_t s_used;
    const uint32_t tid = threadIdx.x, wv = tid >> 6;
    const uint64_t b0 = (uint64_t)blockIdx.x * BGZF_IN;
    const uint32_t n = (uint32_t)(nbytes - b0 < BGZF_IN ? nbytes - b0 : BGZF_IN);
    for (uint32_t i = tid; i < 4 * 4 * 256; i += 256) (&s_hist[0][0][0])[i] = 0;
    if (tid == 0) s_used = 0;
    __syncthreads();
    {   // histogram: coalesced 16-byte loads, four copies per wave (lane & 3) so that the few hot symbols of a text do not serialise a wave
        uint32_t* h = s_hist[wv][tid & 3];
        const uint4* t16 = reinterpret_cast<const uint4*>(text + b0);
        const uint32_t n16 = n >> 4;
        for (uint32_t i = tid; i < n16; i += 256) {
            const uint4 v = t16[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t x = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; atomicAdd(&h[x & 255u], 1u); atomicAdd(&h[(x >> 8) & 255u], 1u); atomicAdd(&h[(x >> 16) & 255u], 1u); atomicAdd(&h[x >> 24], 1u); }
        }
        for (uint32_t i = (n16 << 4) + tid; i < n; i += 256) atomicAdd(&h[text[b0 + i]], 1u);
    }
    __syncthreads();
    { uint32_t f = 0; for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) f += s_hist[a][c][tid]; s_freq[tid] = f; if (tid == 0) s_freq[256] = 1; }
    __syncthreads();
    // rank of every used symbol among the used ones (ties by symbol): thread t ranks symbol t (thread 0 also the end-of-block symbol)
    for (uint32_t s = tid; s < NSYM; s += 256) {
        const uint32_t f = s_freq[s];
        if (f) { uint32_t r = 0; for (uint32_t o = 0; o < NSYM; ++o) { const uint32_t g = s_freq[o]; r += (g && (g < f || (g == f && o < s))) ? 1u : 0u; } s_order[r] = (uint16_t)s; atomicAdd(&s_used, 1u); }
    }
    for (uint32_t i = tid; i < BGZF_PLAN_BYTES; i += 256) s_len[i] = 0;
    __syncthreads();
    if (tid == 0) {
        const uint32_t m = s_used;                        // >= 2: a literal and the end-of-block symbol
        const bool complete = huff_lengths(s_freq, s_order, m, s_w, s_par, s_len);
        // exact size: BGZF header 18 + deflate (3 header bits ... + data + end-of-block) + CRC32 + ISIZE
        BitCount bc{0}; write_deflate_header(s_len, bc);
        uint64_t bits = bc.bits;
        for (uint32_t i = 0; i < m; ++i) { const uint32_t s = s_order[i]; bits += (uint64_t)s_freq[s] * s_len[s]; }
        uint32_t cbytes = (uint32_t)((bits + 7) >> 3), stored = 0;
        if (!complete || cbytes > BGZF_LDS_OUT || cbytes >= n + 5u) { cbytes = n + 5u; stored = 1; }   // stored deflate block: 1 + LEN + NLEN + the bytes
        s_len[BGZF_PLAN_BYTES - 1] = (uint8_t)stored;
        sizes[blockIdx.x] = 18u + cbytes + 8u;
    }
    __syncthreads();
    for (uint32_t i = tid; i < BGZF_PLAN_BYTES / 4; i += 256) reinterpret_cast<uint32_t*>(plans + (size_t)blockIdx.x * BGZF_PLAN_BYTES)[i] = reinterpret_cast<const uint32_t*>(s_len)[i];
}

// ------------------------------------------------------------------------------------------------ emit
__global__ void __launch_bounds__(256) k_bgzf_emit(const uint8_t* __restrict__ text, uint64_t nbytes, const uint8_t* __restrict__ plans, const uint32_t* __restrict__ sizes,
                                                   const uint32_t* __restrict__ offs, const uint32_t* __restrict__ crc_tab, const uint32_t* __restrict__ crc_pow,
                                                   uint8_t* __restrict__ zout, uint64_t zbase) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint32_t* s_out = reinterpret_cast<uint32_t*>(s_dyn);                         // [(4 + 18 + BGZF_LDS_OUT + 8 + 4) / 4] the block as it will lie in memory, behind `pad` bytes
    uint32_t* s_code = s_out + (BGZF_LDS_OUT + 48) / 4;                          // [257] code (bit-reversed) | length << 16
    uint32_t* s_crc = s_code + 260;                                               // [256] CRC table
    uint32_t* s_scan = s_crc + 256;                                               // [8]
    uint8_t* s_len = reinterpret_cast<uint8_t*>(s_scan + 8);                      // [BGZF_PLAN_BYTES]
    __shared__ uint32_t s_hdr_bits, s_crc_acc;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t b0 = (uint64_t)blockIdx.x * BGZF_IN;
    const uint32_t n = (uint32_t)(nbytes - b0 < BGZF_IN ? nbytes - b0 : BGZF_IN);
    const uint32_t total = sizes[blockIdx.x];
    const uint64_t dst = zbase + offs[blockIdx.x];
    const uint32_t pad = (uint32_t)(dst & 3u);                                    // LDS byte i + pad <-> memory byte dst + i: dwords line up
    for (uint32_t i = tid; i < BGZF_PLAN_BYTES / 4; i += 256) reinterpret_cast<uint32_t*>(s_len)[i] = reinterpret_cast<const uint32_t*>(plans + (size_t)blockIdx.x * BGZF_PLAN_BYTES)[i];
    s_crc[tid] = crc_tab[tid];
    for (uint32_t i = tid; i < (BGZF_LDS_OUT + 48) / 4; i += 256) s_out[i] = 0;
    if (tid == 0) s_crc_acc = 0;
    __syncthreads();
    const bool stored = s_len[BGZF_PLAN_BYTES - 1] != 0;
    uint8_t* out8 = reinterpret_cast<uint8_t*>(s_out) + pad;
    // ---- CRC-32 of the input: thread t takes the t-th chunk FROM THE END (so that the bytes behind a chunk are a multiple of
    // CHUNK whatever n is: the factor x^(8 * CHUNK * t) comes from a table), standard CRC of the chunk, then the combination
    // crc(A || B) = crc(A) * x^(8 |B|) + crc(B) is linear: the XOR of the shifted chunk CRCs
    {
        const int64_t hi = (int64_t)n - (int64_t)CHUNK * tid, lo = hi - (int64_t)CHUNK > 0 ? hi - (int64_t)CHUNK : 0;
        uint32_t c = 0;
        if (hi > 0) {
            c = 0xFFFFFFFFu;
            const uint8_t* p = text + b0;
            if (((lo | hi) & 3) == 0) for (int64_t i = lo; i < hi; i += 4) { uint32_t v = *reinterpret_cast<const uint32_t*>(p + i); for (int k = 0; k < 4; ++k) { c = s_crc[(c ^ v) & 255u] ^ (c >> 8); v >>= 8; } }
            else for (int64_t i = lo; i < hi; ++i) c = s_crc[(c ^ p[i]) & 255u] ^ (c >> 8);
            c = ~c;
            c = crc_mul(crc_pow[tid], c);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c ^= __shfl_xor((int)c, d);
        if (lane == 0) atomicXor(&s_crc_acc, c);
    }
    if (stored) {
        // a stored deflate block: 01 LEN NLEN + the bytes (never taken by text; kept so that no input can overrun the LDS staging).
        // Written straight to memory, byte by byte.
        uint8_t* o = zout + dst;
        __syncthreads();
        if (tid == 0) {
            const uint8_t h[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, (uint8_t)((total - 1u) & 255u), (uint8_t)((total - 1u) >> 8)};
            for (int i = 0; i < 18; ++i) o[i] = h[i];
            o[18] = 1; o[19] = (uint8_t)(n & 255u); o[20] = (uint8_t)(n >> 8); o[21] = (uint8_t)(~n & 255u); o[22] = (uint8_t)((~n >> 8) & 255u);
            const uint32_t crc = s_crc_acc;
            for (int i = 0; i < 4; ++i) { o[23 + n + i] = (uint8_t)(crc >> (8 * i)); o[27 + n + i] = (uint8_t)(n >> (8 * i)); }
        }
        for (uint32_t i = tid; i < n; i += 256) o[23 + i] = text[b0 + i];
        return;
    }
    // ---- canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first stream
    {
        if (tid == 0) {
            uint32_t next[16], bl[16]; for (int i = 0; i < 16; ++i) bl[i] = 0;
            for (uint32_t s = 0; s < NSYM; ++s) ++bl[s_len[s]];
            bl[0] = 0; uint32_t code = 0; next[0] = 0;
            for (int b = 1; b < 16; ++b) { code = (code + bl[b - 1]) << 1; next[b] = code; }
            for (uint32_t s = 0; s < NSYM; ++s) { const uint32_t l = s_len[s]; s_code[s] = l ? (rev_bits(next[l]++, l) | (l << 16)) : 0u; }
            // the BGZF header and the deflate header; the data starts at bit s_hdr_bits of the deflate stream
            const uint8_t h[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, (uint8_t)((total - 1u) & 255u), (uint8_t)((total - 1u) >> 8)};
            for (int i = 0; i < 18; ++i) out8[i] = h[i];
            BitSink bs{out8 + 18, 0, 0, 0}; write_deflate_header(s_len, bs);
            s_hdr_bits = bs.pos * 8u + bs.n;
            if (bs.n) out8[18 + bs.pos] = (uint8_t)bs.acc;                        // the open byte: the data's first bits are OR-ed into it
        }
    }
    __syncthreads();
    // ---- every thread: the bit count of its chunk [t CHUNK, (t + 1) CHUNK), a block-wide exclusive prefix sum, then its codes at that offset
    const uint32_t c_lo = tid * CHUNK < n ? tid * CHUNK : n, c_hi = (tid + 1) * CHUNK < n ? (tid + 1) * CHUNK : n;
    const uint8_t* src = text + b0;
    uint32_t my_bits = 0;
    for (uint32_t i = c_lo; i + 4 <= c_hi; i += 4) { const uint32_t v = *reinterpret_cast<const uint32_t*>(src + i); my_bits += (s_code[v & 255u] >> 16) + (s_code[(v >> 8) & 255u] >> 16) + (s_code[(v >> 16) & 255u] >> 16) + (s_code[v >> 24] >> 16); }
    for (uint32_t i = c_lo + ((c_hi - c_lo) & ~3u); i < c_hi; ++i) my_bits += s_code[src[i]] >> 16;
    uint32_t incl = my_bits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += o; }
    if (lane == 63) s_scan[wv] = incl;
    __syncthreads();
    uint32_t base = 0; for (uint32_t w = 0; w < wv; ++w) base += s_scan[w];
    const uint32_t data_bits = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    // bit position inside the LDS image: (pad + 18) bytes, the header's bits, my prefix
    uint64_t pos = (uint64_t)(pad + 18u) * 8u + s_hdr_bits + base + (incl - my_bits);
    {
        uint64_t acc = 0; uint32_t fill = (uint32_t)(pos & 31u); uint32_t w = (uint32_t)(pos >> 5);
        auto put = [&](uint32_t e) { acc |= (uint64_t)(e & 0xFFFFu) << fill; fill += e >> 16; if (fill >= 32u) { atomicOr(&s_out[w++], (uint32_t)acc); acc >>= 32; fill -= 32u; } };
        for (uint32_t i = c_lo; i + 4 <= c_hi; i += 4) { const uint32_t v = *reinterpret_cast<const uint32_t*>(src + i); put(s_code[v & 255u]); put(s_code[(v >> 8) & 255u]); put(s_code[(v >> 16) & 255u]); put(s_code[v >> 24]); }
        for (uint32_t i = c_lo + ((c_hi - c_lo) & ~3u); i < c_hi; ++i) put(s_code[src[i]]);
        if (tid == 255) put(s_code[256]);                                         // end of block (thread 255's chunk is the last, possibly empty, one)
        if (fill) atomicOr(&s_out[w], (uint32_t)acc);
    }
    __syncthreads();
    if (tid == 0) {                                                              // trailer: CRC32, ISIZE behind the deflate data
        const uint32_t cbytes = (s_hdr_bits + data_bits + (s_code[256] >> 16) + 7u) >> 3;
        uint8_t* t8 = out8 + 18 + cbytes; const uint32_t crc = s_crc_acc;
        for (int i = 0; i < 4; ++i) { t8[i] = (uint8_t)(crc >> (8 * i)); t8[4 + i] = (uint8_t)(n >> (8 * i)); }
    }
    __syncthreads();
    // ---- LDS -> memory: the first and last partial dwords byte by byte, whole dwords between
    {
        uint8_t* o = zout + (dst - pad);                                          // dword aligned
        const uint32_t end = pad + total, w_lo = (pad + 3u) >> 2, w_hi = end >> 2;
        const uint8_t* img = reinterpret_cast<const uint8_t*>(s_out);
        if (tid == 0) { for (uint32_t i = pad; i < (w_lo << 2) && i < end; ++i) o[i] = img[i]; for (uint32_t i = (w_hi << 2) > pad ? (w_hi << 2) : pad; i < end && w_hi >= w_lo; ++i) o[i] = img[i]; }
        for (uint32_t w = w_lo + tid; w < w_hi; w += 256) reinterpret_cast<uint32_t*>(o)[w] = s_out[w];
    }
}

// ------------------------------------------------------------------------------------------------ host
static uint32_t crc_x2n(uint32_t bytes);
// Host emulation of the two kernels, "thread" by "thread", over the same functions and the same arithmetic (chunked CRC combined by
// carry-less multiplication, chunk bit offsets by prefix sums, the stored fallback): the CPU suite inflates its output with zlib
// (scs_bgzf_probe).  A test seam -- the product compresses on the device only.
void bgzf_compress_host(const uint8_t* text, uint64_t nbytes, uint32_t lds_out_cap, std::vector<uint8_t>& out) {
    uint32_t crc_tab[256], crc_pow[256]; bgzf_host_tables(crc_tab, crc_pow);
    out.clear();
    for (uint64_t b0 = 0; b0 < nbytes; b0 += BGZF_IN) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(BGZF_IN, nbytes - b0); const uint8_t* src = text + b0;
        // plan
        std::vector<uint32_t> freq(NSYM + 3, 0), w(2 * NSYM); std::vector<uint16_t> order, par(2 * NSYM); uint8_t len[BGZF_PLAN_BYTES] = {0};
        for (uint32_t i = 0; i < n; ++i) ++freq[src[i]];
        freq[256] = 1;
        for (uint32_t s = 0; s < NSYM; ++s) if (freq[s]) order.push_back((uint16_t)s);
        std::stable_sort(order.begin(), order.end(), [&](uint16_t a, uint16_t b) { return freq[a] < freq[b]; });
        const bool complete = huff_lengths(freq.data(), order.data(), (uint32_t)order.size(), w.data(), par.data(), len);
        BitCount bc{0}; write_deflate_header(len, bc);
        uint64_t bits = bc.bits; for (uint16_t s : order) bits += (uint64_t)freq[s] * len[s];
        uint32_t cbytes = (uint32_t)((bits + 7) >> 3); bool stored = false;
        if (!complete || cbytes > lds_out_cap || cbytes >= n + 5u) { cbytes = n + 5u; stored = true; }
        const uint32_t total = 18u + cbytes + 8u;
        // emit
        std::vector<uint8_t> img(total + 8, 0);
        const uint8_t h[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, (uint8_t)((total - 1u) & 255u), (uint8_t)((total - 1u) >> 8)};
        std::copy(h, h + 18, img.begin());
        uint32_t crc = 0;
        for (uint32_t t = 0; t < 256; ++t) {                                       // chunk t from the end
            const int64_t hi = (int64_t)n - (int64_t)CHUNK * t, lo = std::max<int64_t>(hi - (int64_t)CHUNK, 0);
            if (hi <= 0) continue;
            uint32_t c = 0xFFFFFFFFu; for (int64_t i = lo; i < hi; ++i) c = crc_tab[(c ^ src[i]) & 255u] ^ (c >> 8);
            crc ^= crc_mul(crc_pow[t], ~c);
        }
        if (stored) {
            img[18] = 1; img[19] = (uint8_t)(n & 255u); img[20] = (uint8_t)(n >> 8); img[21] = (uint8_t)(~n & 255u); img[22] = (uint8_t)((~n >> 8) & 255u);
            std::copy(src, src + n, img.begin() + 23);
            for (int i = 0; i < 4; ++i) { img[23 + n + i] = (uint8_t)(crc >> (8 * i)); img[27 + n + i] = (uint8_t)(n >> (8 * i)); }
        } else {
            uint32_t next[16], bl[16] = {0}, code[NSYM];
            for (uint32_t s = 0; s < NSYM; ++s) ++bl[len[s]];
            bl[0] = 0; uint32_t cd = 0; next[0] = 0;
            for (int b = 1; b < 16; ++b) { cd = (cd + bl[b - 1]) << 1; next[b] = cd; }
            for (uint32_t s = 0; s < NSYM; ++s) { const uint32_t l = len[s]; code[s] = l ? (rev_bits(next[l]++, l) | (l << 16)) : 0u; }
            BitSink bs{img.data() + 18, 0, 0, 0}; write_deflate_header(len, bs);
            const uint32_t hdr_bits = bs.pos * 8u + bs.n;
            if (bs.n) img[18 + bs.pos] = (uint8_t)bs.acc;
            std::vector<uint32_t> words((total + 11) / 4, 0);                        // the image as dwords, OR-ed into like the LDS image
            uint64_t pos = 18ull * 8 + hdr_bits;
            for (uint32_t t = 0; t < 256; ++t) {
                const uint32_t c_lo = std::min(t * CHUNK, n), c_hi = std::min((t + 1) * CHUNK, n);
                uint64_t acc = 0; uint32_t fill = (uint32_t)(pos & 31u), wi = (uint32_t)(pos >> 5);
                auto put = [&](uint32_t e) { acc |= (uint64_t)(e & 0xFFFFu) << fill; fill += e >> 16; pos += e >> 16; if (fill >= 32u) { words[wi++] |= (uint32_t)acc; acc >>= 32; fill -= 32u; } };
                for (uint32_t i = c_lo; i < c_hi; ++i) put(code[src[i]]);
                if (t == 255) put(code[256]);
                if (fill) words[wi] |= (uint32_t)acc;
            }
            for (size_t i = 0; i < total; ++i) img[i] |= (uint8_t)(words[i >> 2] >> (8 * (i & 3)));
            const uint32_t cb = (uint32_t)((pos - 18ull * 8 + 7) >> 3);
            for (int i = 0; i < 4; ++i) { img[18 + cb + i] = (uint8_t)(crc >> (8 * i)); img[22 + cb + i] = (uint8_t)(n >> (8 * i)); }
        }
        out.insert(out.end(), img.begin(), img.begin() + total);
    }
}
static uint32_t crc_x2n(uint32_t bytes) {                                          // x^(8 * bytes) mod P (reflected)
    uint32_t p = 0x80000000u, sq = 0x00800000u;                                    // x^0, x^8
    for (uint32_t k = bytes; k; k >>= 1) { if (k & 1u) p = crc_mul(sq, p); sq = crc_mul(sq, sq); }
    return p;
}
void bgzf_host_tables(uint32_t* crc_tab256, uint32_t* crc_pow256) {
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1; crc_tab256[i] = c; }
    for (uint32_t t = 0; t < 256; ++t) crc_pow256[t] = crc_x2n(CHUNK * t);
}
size_t bgzf_emit_lds_bytes() { return (BGZF_LDS_OUT + 48) + 260 * 4 + 256 * 4 + 8 * 4 + BGZF_PLAN_BYTES; }

void launch_bgzf_plan(hipStream_t s, const char* text, uint64_t nbytes, uint8_t* plans, uint32_t* sizes) {
    const uint32_t nblk = (uint32_t)((nbytes + BGZF_IN - 1) / BGZF_IN);
    if (nblk) hipLaunchKernelGGL(k_bgzf_plan, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(text), nbytes, plans, sizes);
}
void launch_bgzf_emit(hipStream_t s, const char* text, uint64_t nbytes, const uint8_t* plans, const uint32_t* sizes, const uint32_t* offs, const uint32_t* crc_tab, const uint32_t* crc_pow,
                      char* zout, uint64_t zbase) {
    const uint32_t nblk = (uint32_t)((nbytes + BGZF_IN - 1) / BGZF_IN);
    if (!nblk) return;
    static bool opted[64] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (!opted[dev & 63]) { (void)hipFuncSetAttribute((const void*)k_bgzf_emit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bgzf_emit_lds_bytes()); opted[dev & 63] = true; }
    hipLaunchKernelGGL(k_bgzf_emit, dim3(nblk), dim3(256), bgzf_emit_lds_bytes(), s, reinterpret_cast<const uint8_t*>(text), nbytes, plans, sizes, offs, crc_tab, crc_pow,
                       reinterpret_cast<uint8_t*>(zout), zbase);
}

}  // namespace scs
