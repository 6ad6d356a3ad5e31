// flagstat_zstd_kernels.h -- device side of the GPU Zstandard frame decoder (flagstat_zstd_kernels.hip), as the host
// orchestration (flagstat_gpu_decode.hip) sees it: plain C++, no device code, so the orchestration also builds against the
// test-only HIP stand-in (tests/hoststub) and runs under ThreadSanitizer.
#ifndef FLAGSTAT_ZSTD_KERNELS_H_
#define FLAGSTAT_ZSTD_KERNELS_H_

#include <hip/hip_runtime.h>

#include <stdint.h>

#include "flagstat_lz4_kernels.h"  // fsk::GpuBlock: one payload = one frame here

namespace fsk {

constexpr uint32_t kZstdMaxBlocks = 256;        // Zstandard blocks per frame the scratch of a first pass holds at most (more: status kZstdTooManyBlocks)
constexpr uint32_t kZstdMaxBlocksRetry = 4096;  // ... and of the second pass the host makes over frames that ran out (fsk_zstd_decode_ex)
constexpr uint32_t kZstdMaxFrameBytes = 1u << 26;  // decoded bytes per frame it takes
constexpr int kZstdTallyWords = 32;             // unsigned long long words of the tally the kernels add to
// Waves per role of the execution kernel: 4 emit + 3 scan + 1 copy = EIGHT waves, two per SIMD, so that a CU holds as many
// workgroups as its LDS allows (two of 80 KB when this was measured, three of 47 KB now: see the ring below).  With ten waves (4 + 5 + 1, what this kernel shipped with first) a workgroup puts 3 + 3 + 2 + 2
// waves on the four SIMDs and the second workgroup's three do not fit beside them at 95 VGPRs: a CU then held ONE frame
// (tests/perf/zstd_occupancy.sh: 256 / 512 / 768 / 1024 frames took 1.25 / 2.49 / 3.69 / 4.87 ms; eight waves: 1.44 / 1.76 ms for
// 256 / 512).  Measurement builds override the split: make HIPFLAGS+=-DFLAGSTAT_ZSTD_EMITTERS=.. -DFLAGSTAT_ZSTD_SCANNERS=..
#ifndef FLAGSTAT_ZSTD_EMITTERS
#define FLAGSTAT_ZSTD_SHIPPED_SPLIT 1
#define FLAGSTAT_ZSTD_EMITTERS 4
#define FLAGSTAT_ZSTD_SCANNERS 3
#endif
constexpr int kZstdEmitters = FLAGSTAT_ZSTD_EMITTERS, kZstdScanners = FLAGSTAT_ZSTD_SCANNERS;  // waves per role of the execution kernel (profile output divides by them)

// Bytes of output the ring keeps for near matches: 32 KiB, not the 64 KiB it could.  What limits this kernel is LDS: a second
// workgroup on a CU costs 22 % of its time and a third little more, and with 32 + 4 KiB of ring a workgroup takes 47 KB = THREE a CU
// (24 waves: 80 VGPRs each, which costs four spilled registers); the matches between 32 and 64 KiB back take the far path
// (flushed output, global loads) that those beyond 64 KiB take anyway.  2^31 flags 39.9 -> 36.8 ms at level 1, 43.3 -> 39.3 at level
// 19 (profiles/r04/zstd_window.log).  Measurement builds: -DFLAGSTAT_ZSTD_WINDOW=65536 -DFLAGSTAT_ZSTD_EXEC_WAVES_PER_SIMD=4.
#ifndef FLAGSTAT_ZSTD_WINDOW
#define FLAGSTAT_ZSTD_WINDOW 32768
#define FLAGSTAT_ZSTD_EXEC_WAVES_PER_SIMD 6
#endif
constexpr uint32_t kZstdWindow = FLAGSTAT_ZSTD_WINDOW;

// Status codes of a frame.  1..63: the frame is damaged; from 64: valid Zstandard this decoder does not take (skippable or
// concatenated frames, dictionaries, content checksums, more than kZstdMaxBlocks blocks, frames above kZstdMaxFrameBytes) --
// the host decodes such a file with libzstd.
enum {
    kZstdOk = 0,
    kZstdBadHeader = 1,        // frame header damaged / content size differs from the declared block size
    kZstdBadBlock = 2,         // block header: reserved type, size past the payload
    kZstdBadLiterals = 3,      // literals section: sizes past the block, stream table, Huffman stream not consumed exactly
    kZstdBadHuffman = 4,       // Huffman tree description
    kZstdBadSequences = 5,     // sequences section header / FSE table description
    kZstdBadBitstream = 6,     // sequence bit stream not consumed exactly
    kZstdBadOffset = 7,        // a match reaches before the frame, a repeat offset of zero
    kZstdBadSize = 8,          // decoded size differs from the declared one, a block decodes to more than 128 KiB
    kZstdNoTable = 9,          // repeat mode / treeless literals without an earlier table
    kZstdStuck = 10,           // a wait inside the workgroup ran out (a logic error, not the data)
    kZstdUnsupported = 64,     // not a plain Zstandard frame (skippable frame, bad magic)
    kZstdDictionary = 65,
    kZstdChecksum = 66,
    kZstdTooManyBlocks = 67,
    kZstdTrailingData = 68,    // bytes behind the frame (a second frame)
    kZstdTooLarge = 69,
};

}  // namespace fsk

extern "C" {
// Bytes of device scratch a launch over `nframes` frames of at most `max_dst_len` decoded bytes each needs (sequence
// records, literals and batch checkpoints between the two kernels).
uint64_t fsk_zstd_scratch_bytes(uint32_t max_dst_len, uint32_t nframes);
// Decode `nblocks` Zstandard frames on `stream`: frame i reads comp[blocks[i].src_off ..+src_len) and writes
// out[blocks[i].dst_off ..+dst_len); status[i] = 0 or one of the codes above (the frame's output then holds garbage).
// Two kernels: entropy decode (Huffman literals, FSE sequences -> records in `scratch`), then sequence execution.
// tally[0] += sequence records, tally[1] += matches read back from global memory; prof != 0: cycle counters --
// [5] batches and [6] groups of the emitters, [8..11] scanners (cycles, waiting, doubling rounds, chunks with pointers inside),
// [12..14] copier (cycles, waiting, chunks), [15] / [16] emitters (cycles, waiting), [17..19] prepare (cycles, literals, tables),
// [20..22] chain (wave cycles, waves, steps), [23..25] records (workgroup cycles, relaxation rounds, batches).
// `comp` must be readable for 64 bytes past the last payload; `scratch` holds fsk_zstd_scratch_bytes(max_dst_len, nblocks).
hipError_t fsk_zstd_decode(const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out, uint32_t* status,
                           unsigned long long* tally, void* scratch, uint64_t scratch_bytes, uint32_t max_dst_len, int prof,
                           hipStream_t stream);
// The same with room for at least `min_blocks` Zstandard blocks per frame (capped at kZstdMaxBlocksRetry): the second pass over
// the frames a first pass answered with kZstdTooManyBlocks (windows of a few KiB, streaming compressors that flush often).
uint64_t fsk_zstd_scratch_bytes_ex(uint32_t max_dst_len, uint32_t nframes, uint32_t min_blocks);
hipError_t fsk_zstd_decode_ex(const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out, uint32_t* status,
                              unsigned long long* tally, void* scratch, uint64_t scratch_bytes, uint32_t max_dst_len, uint32_t min_blocks, int prof,
                              hipStream_t stream);
// waves per role of the execution kernel in this build
void fsk_zstd_role_waves(int* emitters, int* scanners);
// workgroups of the execution kernel one CU holds at once (the runtime's occupancy query, which counts registers and LDS per CU
// but not how a workgroup's waves fall on the four SIMDs: tests/perf/zstd_occupancy.sh measures what really fits; 0 on failure)
int fsk_zstd_frames_per_cu(void);
}

#endif  // FLAGSTAT_ZSTD_KERNELS_H_
