// How fast are the two Random123 block functions on gfx950?  Philox4x32-10 is built on 32 x 32 -> 64-bit multiplies (v_mul_hi_u32 +
// v_mul_lo_u32: quarter rate on CDNA), Threefry4x32 on add / rotate / xor (full rate).  Blocks per second, every lane busy.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/rng_rate.hip -o /tmp/rng_rate && /tmp/rng_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }
template <int ROUNDS>
__device__ __forceinline__ void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0], hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
template <int ROUNDS>
__device__ __forceinline__ void threefry(uint32_t x[4], const uint32_t k[4]) {
    const uint32_t ks[5] = {k[0], k[1], k[2], k[3], 0x1BD11BDAu ^ k[0] ^ k[1] ^ k[2] ^ k[3]};
    constexpr int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] += ks[i];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r & 1) { x[0] += x[3]; x[3] = rotl(x[3], R[r & 7][0]) ^ x[0]; x[2] += x[1]; x[1] = rotl(x[1], R[r & 7][1]) ^ x[2]; }
        else       { x[0] += x[1]; x[1] = rotl(x[1], R[r & 7][0]) ^ x[0]; x[2] += x[3]; x[3] = rotl(x[3], R[r & 7][1]) ^ x[2]; }
        if ((r & 3) == 3) { const int s = r / 4 + 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] += ks[(s + i) % 5];
            x[3] += (uint32_t)s; }
    }
}
template <int WHICH, int ROUNDS>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
    uint32_t c[4] = {threadIdx.x, blockIdx.x, 3u, 4u}; const uint32_t key[4] = {0x12345678u, 0x9ABCDEF0u, 0u, 0u};
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t x[4] = {c[0] + (uint32_t)i, c[1], c[2], c[3]};
        if (WHICH == 0) philox<ROUNDS>(x, key[0], key[1]); else threefry<ROUNDS>(x, key);
        acc ^= x[0] ^ x[1] ^ x[2] ^ x[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int WHICH, int ROUNDS> void run(const char* name, uint32_t* d) {
    const int blocks = 256 * 32, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<WHICH, ROUNDS>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a); hipLaunchKernelGGL((k<WHICH, ROUNDS>), dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-22s %7.2f ms  %.3e blocks/s\n", name, ms, (double)blocks * 256 * iters / (ms * 1e-3));
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 32 * 256 * 4);
    run<0, 10>("Philox4x32-10", d); run<0, 7>("Philox4x32-7", d); run<1, 20>("Threefry4x32-20", d); run<1, 13>("Threefry4x32-13", d); run<1, 12>("Threefry4x32-12", d);
    // known answers (Random123 kat_vectors): threefry4x32 20 rounds, counter = key = 0
    return 0;
}
