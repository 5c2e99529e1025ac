// How fast can a wave64 kernel WRITE text records the way k_reads does -- every lane its own ~320-byte record, 16 bytes per store
// instruction, a 32-byte sector as two stores back to back, 16 "positions" of other work between sectors -- and how fast when four
// neighbouring lanes write 64 contiguous bytes of ONE record per instruction (a 4x4 transposition of the quad's pieces)?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr uint32_t REC = 320;                      // bytes per record (10 sectors)
// mode 0: lane = record, per step one 32-byte sector (two dwordx4 stores); mode 1: a quad writes 64 contiguous bytes of one record per
// instruction, 4 instructions per 4 records and 64 bytes (the same bytes, transposed); mode 2: lane = record, one dwordx4 per step
// (16-byte pieces: the text path without the half-sector pairing); mode 3: fully coalesced stream (the roof)
template <int MODE>
__global__ void __launch_bounds__(256) k_store(char* __restrict__ out, uint32_t nrec, uint32_t spin) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    uint32_t x = gid * 2654435761u + 1u;
    if (MODE == 3) {
        const size_t total = (size_t)nrec * REC / 16;
        for (size_t i = gid; i < total; i += (size_t)gridDim.x * 256) { for (uint32_t s = 0; s < spin / 4; ++s) __builtin_amdgcn_s_sleep(8); x += 7u; reinterpret_cast<uint4*>(out)[i] = make_uint4(x, x + 1, x + 2, x + 3); }
        return;
    }
    if (gid >= nrec) return;
    if (MODE == 0 || MODE == 2) {
        char* r = out + (size_t)gid * REC;
        for (uint32_t sec = 0; sec < REC / 32; ++sec) {
            for (uint32_t s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8); x += 7u;   // the walk between two sectors: spin x 512 cycles in which the wave issues nothing
            if (MODE == 0) { reinterpret_cast<uint4*>(r + 32 * sec)[0] = make_uint4(x, x + 1, x + 2, x + 3); reinterpret_cast<uint4*>(r + 32 * sec)[1] = make_uint4(x + 4, x + 5, x + 6, x + 7); }
            else { reinterpret_cast<uint4*>(r + 32 * sec)[0] = make_uint4(x, x + 1, x + 2, x + 3); for (uint32_t s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8); x += 3u; reinterpret_cast<uint4*>(r + 32 * sec)[1] = make_uint4(x + 4, x + 5, x + 6, x + 7); }
        }
    } else {
        const uint32_t q0 = gid & ~3u, li = lane & 3u;                               // my quad's first record, my place in it
        for (uint32_t blk = 0; blk < REC / 64; ++blk) {
            for (uint32_t s = 0; s < 2 * spin; ++s) __builtin_amdgcn_s_sleep(8); x += 5u;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {                                      // instruction k: the quad writes 64 bytes of record q0 + k
                char* r = out + (size_t)(q0 + k) * REC + 64 * blk;
                reinterpret_cast<uint4*>(r)[li] = make_uint4(x + k, x + 1, x + 2, x + 3);
            }
        }
    }
}
int main() {
    const uint32_t nrec = 16u << 20;                                                // 16 M records x 320 B = 5.4 GB
    char* out; CK(hipMalloc(&out, (size_t)nrec * REC + 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (uint32_t spin : {0u, 2u, 8u, 20u}) for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            const uint32_t grid = mode == 3 ? 4096u : nrec / 256;
            if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(grid), dim3(256), 0, 0, out, nrec, spin);
            else if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(grid), dim3(256), 0, 0, out, nrec, spin);
            else if (mode == 2) hipLaunchKernelGGL(k_store<2>, dim3(grid), dim3(256), 0, 0, out, nrec, spin);
            else hipLaunchKernelGGL(k_store<3>, dim3(grid), dim3(256), 0, 0, out, nrec, spin);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
        }
        printf("spin %4u  mode %d (%s): %.3f ms  %.2f TB/s\n", spin, mode, mode == 0 ? "lane = record, 32-byte sectors" : mode == 1 ? "quad = 64 contiguous bytes" : mode == 2 ? "lane = record, 16-byte pieces" : "coalesced stream", best, (double)nrec * REC / best / 1e9);
    }
    return 0;
}
