// scs_bgzf.h -- BGZF blocks made on the GPU from FASTQ text in HBM (scs_bgzf.hip).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include <hip/hip_runtime.h>

namespace scs {

#define BGZF_IN 64512u            // input bytes per BGZF block (a block's worst case -- a stored deflate block -- stays below the format's 64 KB)
#define BGZF_LDS_OUT 40960u       // deflate bytes a block may take in the emit kernel's LDS image; above it the block is stored
#define BGZF_PLAN_BYTES 288u      // per block: the 257 code lengths, padding, [287] = 1: stored

// upper bound of the BGZF bytes of nbytes of text (every block stored)
static inline uint64_t bgzf_bound(uint64_t nbytes) { return nbytes + ((nbytes + BGZF_IN - 1) / BGZF_IN) * 31u + 64u; }
static inline uint32_t bgzf_blocks(uint64_t nbytes) { return (uint32_t)((nbytes + BGZF_IN - 1) / BGZF_IN); }

// CRC-32 byte table [256] and x^(8 * 252 * t) mod P for t = 0..255 (uploaded once per ctx)
void bgzf_host_tables(uint32_t* crc_tab256, uint32_t* crc_pow256);
// host emulation of the kernels (test seam: scs_bgzf_probe); lds_out_cap: BGZF_LDS_OUT, or less to force stored blocks
void bgzf_compress_host(const uint8_t* text, uint64_t nbytes, uint32_t lds_out_cap, std::vector<uint8_t>& out);
// per block: code lengths -> plans[b * BGZF_PLAN_BYTES ..], exact block size -> sizes[b]
void launch_bgzf_plan(hipStream_t s, const char* text, uint64_t nbytes, uint8_t* plans, uint32_t* sizes);
// block b -> zout[zbase + offs[b] ..) (offs = exclusive prefix sum of sizes); text 16-byte aligned, zout 4-byte aligned
void launch_bgzf_emit(hipStream_t s, const char* text, uint64_t nbytes, const uint8_t* plans, const uint32_t* sizes, const uint32_t* offs,
                      const uint32_t* crc_tab, const uint32_t* crc_pow, char* zout, uint64_t zbase);

}  // namespace scs
