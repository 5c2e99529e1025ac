// valu_peak.hip -- measures the integer VALU issue rate of gfx950 that the roofline's "VALU issue fraction" is priced
// against (bench.py roofline.valu.issue_frac).  Every lane runs K independent chains of the
// full-rate 32-bit ops the hot kernels are made of (v_add_u32, v_xor_b32, v_alignbit/v_lshl_or); the grid puts 1, 2, 4
// or 8 waves on every SIMD.  Prints wave-instructions per cycle per SIMD at the nominal 2.4 GHz and the measured time.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS>
__global__ void __launch_bounds__(256) k_valu(unsigned* out, int iters, unsigned seed) {
    unsigned x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = seed + threadIdx.x * 977u + c * 131u + blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {                                          // 4 VALU per chain step: add, rotate (alignbit), xor, add
            unsigned a = x[c] + 0x9E3779B9u;
            a = (a << 7) | (a >> 25);
            a ^= x[(c + 1) % CHAINS];
            x[c] = a + (unsigned)i;
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r ^= x[c];
    if (r == 0x12345678u) out[0] = r;                                               // keeps the work alive, (almost) never stores
}

template <int CHAINS>
static void run(unsigned* d, int waves_per_simd) {
    const int cus = 256, iters = 20000;
    const int blocks = cus * waves_per_simd;                                        // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_valu<CHAINS><<<blocks, 256>>>(d, 100, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k_valu<CHAINS><<<blocks, 256>>>(d, iters, 2u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double insts = (double)blocks * 4.0 * iters * CHAINS * 4.0;                // wave-instructions (loop overhead not counted)
    const double per_simd_per_cycle = insts / (ms * 1e-3) / (cus * 4.0) / 2.4e9;
    printf("chains %d  waves/SIMD %d  %.3f ms  %.3e wave-insts/s  %.3f wave-insts per cycle per SIMD (2.4 GHz nominal) -> %.2f cycles per wave-inst\n",
           CHAINS, waves_per_simd, ms, insts / (ms * 1e-3), per_simd_per_cycle, 1.0 / per_simd_per_cycle);
}

int main() {
    unsigned* d; (void)hipMalloc(&d, 256);
    for (int w : {1, 2, 4, 8}) run<1>(d, w);
    for (int w : {1, 2, 4, 8}) run<4>(d, w);
    for (int w : {1, 2, 4, 8}) run<8>(d, w);
    (void)hipFree(d);
    return 0;
}
