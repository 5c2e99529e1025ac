// clock_probe.hip -- which clock does an MI355X hold under which kind of work?  (MI355X_MICROARCH.md, 'DVFS give-back' item 6: the in-kernel
// clock is delta s_memtime / delta s_memrealtime x 100 MHz.)  Bodies: a sleeping wave, integer VALU chains, VALU + random LDS reads, VALU +
// scattered 32-byte sector stores (k_reads' pattern: every lane its own 320-byte record), all three together.  1024 workgroups of 256 threads
// (4 per CU, one wave per SIMD each -> 4 waves per SIMD, k_reads' occupancy), back to back for about two seconds per body; thread 0 of every
// workgroup stamps both clocks around its loop; printed: the mean clock and the work rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o clock_probe && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long g_clk[2];

template <int BODY>
__global__ void __launch_bounds__(256, 4) k_body(unsigned* out, uint4* recs, int iters, unsigned seed) {
    __shared__ unsigned s_tab[8192];                                                 // 32 KB: four workgroups per CU fit
    const unsigned tid = threadIdx.x;
    for (unsigned k = tid; k < 8192; k += 256) s_tab[k] = k * 2654435761u + seed;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned x0 = seed + tid * 977u + blockIdx.x, x1 = x0 ^ 0x85EBCA6Bu, x2 = x0 + 0xC2B2AE35u, x3 = ~x0;
    uint4* rec = recs + ((size_t)blockIdx.x * 256 + tid) * 20;                        // my 320-byte record
    for (int i = 0; i < iters; ++i) {
        if (BODY == 0) { __builtin_amdgcn_s_sleep(64); continue; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {                                                // 32 VALU of the full-rate integer kind per step
            x0 = ((x0 << 7) | (x0 >> 25)) + x1; x1 ^= x2 + 0x9E3779B9u; x2 = ((x2 << 13) | (x2 >> 19)) ^ x3; x3 += x0;
            if (BODY == 2 || BODY == 4) { x1 += s_tab[(x0 >> 3) & 8191u]; x3 ^= s_tab[(x2 >> 5) & 8191u]; }   // two random LDS reads per 4 VALU steps
        }
        if (BODY == 3 || BODY == 4) {                                                // a 32-byte sector of my record per step (two dwordx4 stores)
            const int sct = i % 10;
            rec[2 * sct] = make_uint4(x0, x1, x2, x3); rec[2 * sct + 1] = make_uint4(x3, x2, x1, x0);
        }
    }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345678u) out[0] = x0;
    if (tid == 0) { atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - c0); atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - r0); }
}

template <int BODY>
static void run(const char* what, unsigned* d, uint4* recs, int iters) {
    unsigned long long z[2] = {0, 0}, h[2];
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_body<BODY><<<1024, 256>>>(d, recs, iters, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k_body<BODY><<<1024, 256>>>(d, recs, iters, 1u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float one = 0; (void)hipEventElapsedTime(&one, a, b);
    const int reps = one > 0 ? (int)(2000.0f / one) + 1 : 1;                          // about two seconds of it back to back, the last launches stamped
    for (int r = 0; r < reps; ++r) k_body<BODY><<<1024, 256>>>(d, recs, iters, 2u + r);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof z);
    (void)hipEventRecord(a);
    for (int r = 0; r < 8; ++r) k_body<BODY><<<1024, 256>>>(d, recs, iters, 100u + r);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b); ms /= 8.0f;
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof h);
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    const double steps = 1024.0 * 4.0 * iters;                                        // wave-steps per launch
    printf("%-46s %8.3f ms per launch  clock %.3f GHz  %.3e wave-steps/s  (%.1f shader cycles per wave-step per SIMD)\n", what, ms, ghz, steps / (ms * 1e-3),
           ghz * 1e9 * 1024.0 / (steps / (ms * 1e-3)));
}

int main() {
    unsigned* d; (void)hipMalloc(&d, 256);
    uint4* recs; (void)hipMalloc(&recs, (size_t)1024 * 256 * 320);
    run<0>("asleep (s_sleep 64 per step)", d, recs, 20000);
    run<1>("32 integer VALU per step", d, recs, 20000);
    run<2>("32 VALU + 4 random LDS reads per step", d, recs, 20000);
    run<3>("32 VALU + one 32-byte sector store per step", d, recs, 20000);
    run<4>("32 VALU + 4 LDS reads + a sector store", d, recs, 20000);
    run<1>("32 integer VALU per step (again)", d, recs, 20000);
    (void)hipFree(d); (void)hipFree(recs);
    return 0;
}
