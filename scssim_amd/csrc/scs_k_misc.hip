// scs_k_misc.hip -- plumbing between the hand-written kernels: the mailbox (device scalars -> pinned host words), device-wide scans
// (rocPRIM), the latch for launch errors, and the two kernel-level test entries (Philox words, det_log bits).
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
static thread_local hipError_t g_launch_err = hipSuccess;
void note_launch(hipError_t e) { if (e != hipSuccess && g_launch_err == hipSuccess) g_launch_err = e; }
hipError_t take_launch_error() { note_launch(hipGetLastError()); const hipError_t e = g_launch_err; g_launch_err = hipSuccess; return e; }
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
    b = std::max(b, parity_scan_temp_bytes(n));                                   // (scs_k_allocate.hip: the parity fix + pair offsets scan)
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

}  // namespace scs
