// What does ONE vector-memory wave-instruction cost the CU's memory path (TA -> TCP -> TD), by access pattern?  k_reads' counters
// (tools/reads_mem_diag.sh) show the path busy most of the launch; this probe prices its instructions: 16 waves per CU (k_reads' occupancy),
// every wave loops over a 4 KB region of its own (L2-resident, beyond the CU's 32 KB L1 in sum), NI instructions per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/tcp_rate.hip -o /tmp/tcp_rate && /tmp/tcp_rate
// Prints shader cycles per wave-instruction and CU (s_memtime of the waves, their mean life / instructions per CU), and the same per lane.
// Under rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN2_sum it calibrates those counters (one mode per run: argv[1]).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int NI = 4096, UN = 8;                   // instructions per wave, independent ones per iteration
constexpr uint32_t REG = 4096;                     // bytes of a wave's region
// modes: 0 coalesced dword load, 1 coalesced dwordx4 load, 2 scattered dword load (a lane = a 64-byte line), 3 scattered dwordx4 load,
// 4 coalesced dwordx4 store, 5 scattered dwordx4 store, 6 scattered sector (two dwordx4 stores back to back to one 32-byte sector),
// 7 scattered dword store, 8 scattered dwordx2 load, 9 coalesced dwordx2 load, 10 dword load with stride 56 B (the pair records),
// 11 scattered byte store, 12 quad-coalesced dwordx4 store (4 lanes = 64 contiguous bytes, 16 lines per instruction)
template <int MODE>
__global__ void __launch_bounds__(256) k_probe(char* __restrict__ buf, unsigned long long* __restrict__ cyc, uint32_t* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    char* r = buf + (size_t)wave * REG;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < NI / UN; ++it) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t k = (uint32_t)(it * UN + u);
            if (MODE == 0) acc ^= *reinterpret_cast<const volatile uint32_t*>(r + ((lane * 4u + k * 256u) & (REG - 1)));
            else if (MODE == 1) { const u32x4 v = *reinterpret_cast<const volatile u32x4*>(r + ((lane * 16u + k * 1024u) & (REG - 1))); acc ^= v.x ^ v.w; }
            else if (MODE == 2) acc ^= *reinterpret_cast<const volatile uint32_t*>(r + lane * 64u + ((k * 4u) & 63u));
            else if (MODE == 3) { const u32x4 v = *reinterpret_cast<const volatile u32x4*>(r + lane * 64u + ((k * 16u) & 63u)); acc ^= v.x ^ v.w; }
            else if (MODE == 4) *reinterpret_cast<volatile u32x4*>(r + ((lane * 16u + k * 1024u) & (REG - 1))) = u32x4{k, lane, k, lane};
            else if (MODE == 5) *reinterpret_cast<volatile u32x4*>(r + lane * 64u + ((k * 16u) & 63u)) = u32x4{k, lane, k, lane};
            else if (MODE == 6) *reinterpret_cast<volatile u32x4*>(r + lane * 64u + ((k >> 1) & 1u) * 32u + (k & 1u) * 16u) = u32x4{k, lane, k, lane};
            else if (MODE == 7) *reinterpret_cast<volatile uint32_t*>(r + lane * 64u + ((k * 4u) & 63u)) = k;
            else if (MODE == 8) { const u32x2 v = *reinterpret_cast<const volatile u32x2*>(r + lane * 64u + ((k * 8u) & 63u)); acc ^= v.x ^ v.y; }
            else if (MODE == 9) { const u32x2 v = *reinterpret_cast<const volatile u32x2*>(r + ((lane * 8u + k * 512u) & (REG - 1))); acc ^= v.x ^ v.y; }
            else if (MODE == 10) acc ^= *reinterpret_cast<const volatile uint32_t*>(r + ((lane * 56u + (k % 14u) * 4u) & (REG - 1)));
            else if (MODE == 11) *reinterpret_cast<volatile uint8_t*>(r + lane * 64u + (k & 63u)) = (uint8_t)k;
            else if (MODE == 12) *reinterpret_cast<volatile u32x4*>(r + (lane >> 2) * 256u + (lane & 3u) * 16u + ((k * 64u) & 255u)) = u32x4{k, lane, k, lane};
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicAdd(cyc, t1 - t0);
    if (acc == 0x12345u) *sink = acc;
}
template <int M> static void launch(uint32_t grid, char* buf, unsigned long long* cyc, uint32_t* sink) { hipLaunchKernelGGL(k_probe<M>, dim3(grid), dim3(256), 0, 0, buf, cyc, sink); }
int main(int argc, char** argv) {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const uint32_t cus = (uint32_t)pr.multiProcessorCount, wg_per_cu = 4, grid = cus * wg_per_cu, waves = grid * 4;
    char* buf; CK(hipMalloc(&buf, (size_t)waves * REG + 4096)); CK(hipMemset(buf, 1, (size_t)waves * REG + 4096));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 8)); uint32_t* sink; CK(hipMalloc(&sink, 4));
    static const char* names[13] = {"coalesced dword load", "coalesced dwordx4 load", "scattered dword load (lane = line)", "scattered dwordx4 load", "coalesced dwordx4 store",
        "scattered dwordx4 store", "scattered sector (2 x dwordx4)", "scattered dword store", "scattered dwordx2 load", "coalesced dwordx2 load", "dword load, stride 56 B", "scattered byte store", "quad-coalesced dwordx4 store"};
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int m = 0; m < 13; ++m) {
        if (only >= 0 && m != only) continue;
        float best = 1e9f; unsigned long long c = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cyc, 0, 8));
            CK(hipEventRecord(a));
            switch (m) { case 0: launch<0>(grid, buf, cyc, sink); break; case 1: launch<1>(grid, buf, cyc, sink); break; case 2: launch<2>(grid, buf, cyc, sink); break; case 3: launch<3>(grid, buf, cyc, sink); break;
                         case 4: launch<4>(grid, buf, cyc, sink); break; case 5: launch<5>(grid, buf, cyc, sink); break; case 6: launch<6>(grid, buf, cyc, sink); break; case 7: launch<7>(grid, buf, cyc, sink); break;
                         case 8: launch<8>(grid, buf, cyc, sink); break; case 9: launch<9>(grid, buf, cyc, sink); break; case 10: launch<10>(grid, buf, cyc, sink); break; case 11: launch<11>(grid, buf, cyc, sink); break; default: launch<12>(grid, buf, cyc, sink); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) { best = ms; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); }
        }
        const double life = (double)c / waves;                                       // mean shader cycles of a wave's loop
        const double per_cu = life / ((double)NI * wg_per_cu * 4);                    // cycles per wave-instruction and CU (16 waves share the path)
        printf("mode %2d  %-38s %8.3f ms  wave life %9.0f cycles  %6.2f cycles per wave-instruction and CU  (%5.3f per lane)  %6.1f G lane-accesses/s\n",
               m, names[m], best, life, per_cu, per_cu / 64.0, (double)waves * NI * 64 / best / 1e6);
    }
    return 0;
}
