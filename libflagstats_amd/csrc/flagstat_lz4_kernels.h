// flagstat_lz4_kernels.h -- device side of the GPU LZ4 block decoder (flagstat_lz4_kernels.hip), as the host
// orchestration (flagstat_gpu_decode.hip) sees it: plain C++, no device code, so the orchestration also builds against the
// test-only HIP stand-in (tests/hoststub) and runs under ThreadSanitizer.
#ifndef FLAGSTAT_LZ4_KERNELS_H_
#define FLAGSTAT_LZ4_KERNELS_H_

#include <hip/hip_runtime.h>

#include <stdint.h>

namespace fsk {

struct GpuBlock {
    uint64_t src_off;  // payload offset in the compressed image
    uint64_t dst_off;  // offset in the decoded buffer (multiple of 16)
    uint32_t src_len;
    uint32_t dst_len;
};

// which decode kernel: the workgroup pipeline (default) or r03's one-wave-per-block kernel (kept for A/B and as the
// yardstick the new one is measured against; knob "lz4_gpu_kernel")
enum { LZ4K_WORKGROUP = 0, LZ4K_WAVE = 1, LZ4K_WAVE_RING16 = 2 };

constexpr int kLz4WgEmitters = 3, kLz4WgScanners = 3;  // waves per role of the workgroup kernel (profile output divides by them)
constexpr int kLz4TallyWords = 32;  // unsigned long long words of the tally the kernels add to (see flagstat_lz4_kernels.hip)

}  // namespace fsk

extern "C" {
// Decode `nblocks` LZ4 blocks on `stream`: block i reads comp[blocks[i].src_off ..+src_len) and writes
// out[blocks[i].dst_off ..+dst_len); status[i] = 0 or an error code (the block then holds garbage); tally[0] += sequences,
// tally[1] += matches read back from global memory (wave kernels only); prof != 0: per-phase cycle counters in tally[2..].
// `comp` must be readable for 64 bytes past the last block's payload.  All pointers are device pointers.
hipError_t fsk_lz4_decode(int kernel, const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out,
                          uint32_t* status, unsigned long long* tally, int prof, hipStream_t stream);
// workgroups of `kernel` one CU holds at once (occupancy query; 0 on failure)
int fsk_lz4_blocks_per_cu(int kernel);
}

#endif  // FLAGSTAT_LZ4_KERNELS_H_
