// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of k_reads, against byte
// counts known by construction (MI355X_MICROARCH.md: FETCH_SIZE halves wide coalesced reads and is uncalibrated for other
// widths: "calibrate on a known byte count in your own access pattern").  Four kernels over a 4 GB buffer (far beyond the
// 256 MB Infinity Cache), every byte touched at most once:
//   k_wide   : 16 bytes per lane, coalesced (the ring images, the pair records' 16-byte pieces)
//   k_gather : 11 consecutive lanes read 11 consecutive dwords at a random 4-byte-aligned place (the window gather of the uniform walk:
//              a read's 150 bases at two bits each + the funnel's extra word); 44 bytes used per group
//   k_dword  : one dword per lane at a random place (k_attach's primer gathers are 8 bytes: the same sector economics)
//   k_store32: every lane stores one whole 32-byte sector (two dwordx4) at a random sector (k_reads' sector stores)
// Run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); the program prints the true bytes per kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_wide(const uint4* __restrict__ p, uint64_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
// group of 11 lanes (5 groups per wave, lanes 55..63 idle): group q reads dwords [base_q, base_q + 11), base_q a permutation-like hash
// of q over the buffer in units of 64 bytes (so that groups never share a sector), + a random dword phase inside 20 bytes
__global__ void k_gather(const uint32_t* __restrict__ p, uint64_t n_groups, uint64_t slots64, uint32_t* __restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, j = lane % 11u, r5 = lane / 11u;
    const uint64_t q = (t >> 6) * 5u + r5;
    uint32_t acc = 0;
    if (r5 < 5u && q < n_groups) {
        const uint64_t slot = ((uint64_t)mix((uint32_t)q) * 2654435761ull + q * 40503ull) % slots64;      // (collisions are rare and only lower the true count slightly)
        const uint32_t ph = mix((uint32_t)q ^ 0xABCDu) % 5u;                                                // the 44 bytes start at dword 0..4 of the 64-byte slot
        acc = p[slot * 16u + ph + j];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_dword(const uint32_t* __restrict__ p, uint64_t n, uint64_t slots64, uint32_t* __restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    if (t < n) { const uint64_t slot = ((uint64_t)mix((uint32_t)t) * 2654435761ull + t * 40503ull) % slots64; acc = p[slot * 16u + (mix((uint32_t)t ^ 0x5555u) & 15u)]; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_store32(uint4* __restrict__ p, uint64_t n, uint64_t sectors) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { const uint64_t sec = ((uint64_t)mix((uint32_t)t) * 2654435761ull + t * 40503ull) % sectors; p[sec * 2] = make_uint4((uint32_t)t, 1, 2, 3); p[sec * 2 + 1] = make_uint4(4, 5, 6, (uint32_t)t); }
}

int main() {
    const uint64_t bytes = 4ull << 30;
    void* buf = nullptr; uint32_t* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes); (void)hipDeviceSynchronize();
    const uint64_t n16 = bytes / 16, slots64 = bytes / 64, groups = 1ull << 24, nd = 1ull << 25, ns = 1ull << 25;
    hipLaunchKernelGGL(k_wide, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)buf, n16, sink);
    hipLaunchKernelGGL(k_gather, dim3((uint32_t)((groups / 5 + 3) / 4 + 1)), dim3(256), 0, 0, (const uint32_t*)buf, groups, slots64, sink);
    hipLaunchKernelGGL(k_dword, dim3((uint32_t)(nd / 256)), dim3(256), 0, 0, (const uint32_t*)buf, nd, slots64, sink);
    hipLaunchKernelGGL(k_store32, dim3((uint32_t)(ns / 256)), dim3(256), 0, 0, (uint4*)buf, ns, bytes / 32);
    (void)hipDeviceSynchronize();
    printf("k_wide   : %llu bytes read, all used (16 B per lane)\n", (unsigned long long)bytes);
    printf("k_gather : %llu groups x 44 bytes used = %llu; x 64 bytes (two 32-byte sectors, or one 64-byte request) = %llu\n", (unsigned long long)groups, (unsigned long long)(groups * 44), (unsigned long long)(groups * 64));
    printf("k_dword  : %llu loads x 4 bytes used = %llu; x 32 bytes per sector = %llu; x 64 = %llu\n", (unsigned long long)nd, (unsigned long long)(nd * 4), (unsigned long long)(nd * 32), (unsigned long long)(nd * 64));
    printf("k_store32: %llu sectors x 32 bytes written = %llu\n", (unsigned long long)ns, (unsigned long long)(ns * 32));
    return 0;
}
