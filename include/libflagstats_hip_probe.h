/*
 * libflagstats_hip_probe.h -- the MEASUREMENT side of libflagstats_hip.so: timing helpers, read-bandwidth probes, the clock
 * probe, the GPU block decoder called directly with its own statistics, and the full reference of the knobs behind
 * FLAGSTATS_hip_set / FLAGSTATS_hip_get.  Nothing a drop-in user of the reference needs: that is libflagstats_hip.h.
 * Exported by the same library (tests/test_host_logic.py checks both headers against it).
 */
#ifndef LIBFLAGSTATS_HIP_PROBE_H_
#define LIBFLAGSTATS_HIP_PROBE_H_

#include "libflagstats_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement: `reps` back-to-back launches of the hot path (K1; + K2 if knob "epilogue" is 0) between two hipEvents on
 * the library's stream, after `warmup` untimed ones.  *ms_total = elapsed ms of
 * the timed region; out[32] += counters of ONE pass.  Returns 0 on success. */
int FLAGSTATS_hip_time_device_u16(const uint16_t* d_array, uint64_t n, int warmup, int reps, float* ms_total,
                                  uint64_t* out);
/* The same over ROTATING slices: launch i counts d_array[slot_i * stride_flags, + n) with slot_i = (i * 7919) % slots,
 * so an array that fits the 256 MiB Infinity Cache is never re-read from it (slots * stride_flags flags must be
 * allocated; stride_flags >= n, even).  *ms_total = elapsed ms of the `reps` timed launches; out[32] += the counters
 * of ALL timed launches.  Returns 0 on success. */
int FLAGSTATS_hip_time_device_u16_rotating(const uint16_t* d_array, uint64_t n, uint64_t stride_flags, uint32_t slots,
                                           int warmup, int reps, float* ms_total, uint64_t* out);

/* shader clock the chip sustains WHILE `launches` back-to-back K1 launches over d_array[0..n) run: a one-wave probe per XCD
 * on a second stream compares the shader-clock counter with the constant 100 MHz reference counter.  K1 is ~65 % VALU-busy
 * at one wave per SIMD, so a chip that holds a lower clock under this load (power, temperature) is slower on the SAME
 * kernel: bench.py reports the number next to the roofline fraction (roofline.sclk_mhz).  Returns 0 on success. */
int FLAGSTATS_hip_sclk_under_load(const uint16_t* d_array, uint64_t n, int launches, double* sclk_mhz);


/* read-only bandwidth probe, no flagstat arithmetic (the analogue of the reference's memcpy baseline,
 * linux/instrumented_benchmark.cpp:456-544), in the fastest read pattern found on the chip (24 KiB in flight per CU:
 * 384-thread workgroups x 4 vectors per lane; profiles/r03/read_probe_sweep.log): `reps` sweeps of d_buf[0..bytes)
 * (16-B aligned) between two hipEvents; nt = non-temporal loads. */
int FLAGSTATS_hip_read_probe(const void* d_buf, uint64_t bytes, int nt, int warmup, int reps, float* ms_total);

/* the fastest pattern with the load's cache-policy bits spelled out (tools/policy_probe.py): policy 0 plain, 1 nt, 2 sc1,
 * 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt, 6 sc0, 7 sc0 nt */
int FLAGSTATS_hip_read_probe_policy(const void* d_buf, uint64_t bytes, int policy, int warmup, int reps, float* ms_total);

/* parameterised variant for access-pattern sweeps (tools/probe_sweep.py): mode 0 grid-stride /
 * 1 block-contiguous; unroll 2|4|8|16 vectors of 16 B per lane per step; threads per workgroup. */
int FLAGSTATS_hip_read_probe2(const void* d_buf, uint64_t bytes, int mode, int unroll, uint32_t threads, uint32_t grid,
                              int nt, int warmup, int reps, float* ms_total);


/* The GPU LZ4 decoder called directly, with its own statistics (flagstat_gpu_decode.hip): the compressed image goes over PCIe
 * in pieces, one wave decodes one block through LDS as soon as its piece has landed, K1 counts the decoded buffer.
 * Synchronous; out[32] += counters.  This is what FLAGSTATS_hip_blockfile* / blockimage_lz4 run for large LZ4 files
 * (knob "lz4_decoder"); measurements: profiles/r03/gpu_lz4_4GiB.log, lz4_decoder_sweep.log. */
typedef struct FLAGSTATS_gpu_lz4_stats {
    uint64_t n_blocks, n_flags, bad_blocks, compressed_bytes, decoded_bytes;
    double h2d_ms, decode_ms, count_ms;            /* stream-event times of the three phases */
    uint64_t sequences, far_matches;               /* LZ4 sequences decoded; matches that reached behind the LDS ring */
    uint64_t ring_kib;                             /* LDS ring per wave (env FLAGSTATS_HIP_GPU_LZ4_RING = 8 | 16) */
    uint64_t chunks;                               /* pieces the image went over PCIe in (env FLAGSTATS_HIP_GPU_LZ4_CHUNKS) */
    double pipeline_ms;                            /* first copy .. counters done; with chunks > 1, h2d_ms = all copies and
                                                      decode_ms = the decode time left exposed after the last copy */
    uint64_t uncompressed_bytes, readers;          /* sum of the blocks' declared sizes; file mode: parallel preads used */
    double wall_s;                                 /* whole call: index, allocations, pipeline, results */
    uint64_t segments;                             /* files larger than the device can hold go through in several segments
                                                      (compressed + decoded bytes of one are resident together); the ms
                                                      fields and `chunks` are sums over them */
} FLAGSTATS_gpu_lz4_stats;
int FLAGSTATS_hip_blockimage_lz4_gpu(const void* image, uint64_t bytes, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats);

/* ---- knob reference: FLAGSTATS_hip_set(key, value) / FLAGSTATS_hip_get(key) of libflagstats_hip.h ----
 * (also env FLAGSTATS_HIP_BLOCKS_PER_CU / _VARIANT / _FUSE / _EPILOGUE / _CHUNK_FLAGS / _ON_ERROR / _NUMA /
 * _GROUP_MIN_GRID / _FENCE_FREE_EVENTS).  key =
 *   "blocks_per_cu"  workgroups per CU of K1's grid (default 1)
 *   "variant"        K1 schedule.  Shipped: 71 (default since r03: non-temporal loads, rolling re-issue at a distance of 6
 *                    vectors = 24 KiB in flight per CU, each wave a contiguous 8 KiB of a step), 25 (r01-r02 default: rolling
 *                    over a whole step = 32 KiB in flight, waves interleaved at 1 KiB) and 9 (plain loop).  Up to 31 the
 *                    number is a bit set (bit0 non-temporal loads, bit1 chain depth 7, bit2 register prefetch, bit3
 *                    interleaved waves, bit4 rolling re-issue), larger numbers are labels; everything that lost a sweep
 *                    (incl. 153 = dynamic schedule with its "dyn_*" policy keys, 29 = two waves per SIMD, 41 = LDS-DMA
 *                    ring) exists only in a `make TUNING=1` build
 *   "epilogue"       accumulate (+=) forms into device memory: 1 (default) = K1's workgroups add their totals to the
 *                    counters with atomics, ONE launch per call, any number of streams may share a counter array;
 *                    0 = partials + K2 (then one counter array must be targeted from one stream at a time).
 *                    The store forms and counters in pinned host memory always use K2.
 *   "fuse"           tuning build only (the r01 experiment that lost): 1 = the last-arriving workgroup of K1 finalises
 *   "chunk_flags"    flags per H2D chunk of the host-pointer entries (default 32 Mi = 64 MiB); for the chunk pipeline (block files on
 *                    host threads, raw files, large pageable arrays) the LARGEST chunk: it aims for 16 MiB, a larger block has its own
 *   "staged_min_flags" pageable host arrays of at least this many flags go through the chunk pipeline (default 2^27; 0 = never)
 *   "on_error"       reference-shaped entry points on failure: 1 abort() after the message (default), 0 return non-zero
 *   "numa"           1 (default): pinned buffers and block-decoder threads are placed on the GPU's host NUMA node
 *   "group_min_grid" K1's atomic epilogue goes through the workspace's 8 per-XCD copies (8 x 2 contended adds on the
 *                    caller's counters per launch instead of one pair per workgroup) from this many workgroups on
 *                    (default 64; 0 = any grid), as long as a workgroup has at most "group_max_steps" steps
 *   "group_max_steps" (default 40 = arrays up to ~320 MiB on 256 CUs; the two forms measure equal at 48 steps, and beyond
 *                    that the workgroups finish too far apart for the contention to matter: one level is 0.3-0.8 % faster).  Read-only key
 *                    "last_k1_two_level": 1 if the most recent K1 launch took the two-level form
 *   "small_flags"    host-pointer calls of up to this many flags (default 1048576, maximum 4194304) are copied by the CPU --
 *                    no copy call -- into the engine's input buffer: fine-grained device memory written through the PCIe BAR
 *                    (knob "small_bar", default 1, needs a large-BAR device; read-only key "small_in_is_device" says which), else
 *                    pinned host memory that K1 reads in place.  Larger single-chunk calls use an asynchronous H2D copy into
 *                    device staging.  0 = always stage
 *   "poll"           1 (default): single-chunk host-pointer calls poll the {value, sequence} pairs the last kernel writes to
 *                    pinned host memory instead of synchronising the stream; 0 = hipStreamSynchronize
 *   "epoch_stagger"  1 (default): the four waves of a K1 workgroup fold their bit-sliced counters (every 255 steps)
 *                    at different steps, so HBM never idles for it chip-wide; 0 = all at the same step (r02)
 *   "fence_free_events" FLAGSTATS_hip_stream_wait_stream / the overlapped all-reduce: 1 = ordering events without the
 *                    system-scope fence (default 0)
 *   "lz4_decoder"    LZ4 block files (FLAGSTATS_hip_blockfile*, blockimage_lz4): 0 = decode on host threads into pinned
 *                    chunks (decoded flags cross PCIe), 1 = decode on the GPU (the compressed bytes cross PCIe, one
 *                    workgroup per block), 2 (default) = by size: on the GPU for files of at least "lz4_gpu_min_bytes"
 *                    (default 64 MiB compressed) and for smaller ones that DECODE to at least 2.5 x that (160 MiB of flags:
 *                    the two decoders cross at 120-200 MB decoded whatever the codec and level, which is 27-95 MiB of
 *                    file; profiles/r05/decoder_crossover.log), on the host below -- and on the host whatever the size when
 *                    the blocks hardly compress (decoded bytes < 1.25 x the file's; Zstandard: 1.9 x): the host pipeline is
 *                    PCIe-bound on such a file and the GPU decoders' literal paths are their slow ones
 *                    (profiles/r04/incompressible_blockfiles.log).  env FLAGSTATS_HIP_LZ4_DECODER / FLAGSTATS_HIP_LZ4_GPU_MIN_BYTES
 *   "zstd_decoder"   Zstandard block files (blockfile_zstd, blockimage_zstd, blockfile): 0 = libzstd on host threads, 1 = decode
 *                    on the GPU (four kernels: Huffman literals + FSE tables, the serial walk of the FSE states, sequence
 *                    records, execution; a frame the decoder does not take -- dictionary, content checksum, concatenated
 *                    or skippable frames, damage -- fails the call with its status code), 2 (default) = by size, like
 *                    "lz4_decoder": on the GPU for files of at least "zstd_gpu_min_bytes" (default 64 MiB) or that decode
 *                    to at least 2.5 x that, and a file with a frame the GPU decoder does not take is
 *                    decoded by libzstd on the host threads instead.  env FLAGSTATS_HIP_ZSTD_DECODER / FLAGSTATS_HIP_ZSTD_GPU_MIN_BYTES
 *   "lz4_gpu_keep_bytes" device memory the GPU LZ4 / Zstandard decoder may keep between calls: its two buffers -- a segment's
 *                    compressed and decoded bytes -- are reused by the next file (allocating them right after freeing
 *                    them was measured to stall ~0.5 s on the driver wiping the freed memory).  Default ~0 = automatic:
 *                    what the last call needed, at most a quarter of the device; in every mode they are released once
 *                    eight calls of other entry points of the engine have passed, by a failed call and by
 *                    FLAGSTATS_hip_shutdown; 0 = free after every call.  Read-only "lz4_gpu_kept_bytes": held now
 *   "lz4_gpu_kernel" 0 (default) = the workgroup decode kernel (eight waves per block, 64 KiB LZ4 window in LDS),
 *                    1 = r03's one wave per block (kept as the yardstick)
 * Read-only keys of FLAGSTATS_hip_get: "grid" (K1 workgroups), "numa_node" (of the default device),
 * "host_chunks" / "host_overlapped" (last multi-chunk host-pointer call on the default engine: chunks
 * submitted / chunks handed over while the previous chunk's copy + kernel were still in flight).
 * Returns 0 on success. */

#ifdef __cplusplus
}
#endif
#endif
