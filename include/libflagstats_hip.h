/*
 * libflagstats_hip.h -- C ABI of libflagstats_hip.so, the MI355X (gfx950) engine behind libflagstats' flagstat entry points.
 * Plain C: pointers and sizes, no HIP or torch types (streams travel as void*).  Every symbol below is exported by
 * libflagstats_amd/libflagstats_hip.so (tests/test_host_logic.py checks that).  Measurement entries (read probes, timing
 * helpers, the clock probe, the GPU decoder called directly) and the full knob reference: libflagstats_hip_probe.h.
 */
#ifndef LIBFLAGSTATS_HIP_H_
#define LIBFLAGSTATS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ================= the drop-in: same names, argument meaning and return as the reference =================
 *
 * Contract (all entry points of this library): counters follow the reference's FLAGSTAT_scalar exactly
 * (libflagstats.h:118-142, :170-176): 32 slots, [0..15] pass-QC, [16..31] fail-QC, slot index = FLAGSTAT_*_OFF
 * (libflagstats.h:69-112); the 19 live slots are {2,6,7,8,10,11,12,13,14} and {18,22..30}; the other slots are never
 * written.  Counters are ACCUMULATED (+=), never zeroed by the callee (`++f[...]`; python/libflagstats.pyx:19 zeroes in
 * the caller).  There is NO CPU fallback: if the GPU path cannot run the call fails loudly -- a message on stderr and a
 * non-zero return; the three reference-shaped entry points (FLAGSTATS_u16, FLAGSTAT_hip, STORM_pospopcnt_u16), whose
 * reference callers ignore the return value (python/libflagstats.pyx:22, benchmark/flagstats.cpp:329), abort() after the
 * message instead of handing back silently-zero counters (knob "on_error" / FLAGSTATS_HIP_ON_ERROR=return restores the
 * plain non-zero return).  Any thread may call; the caller's current HIP device is restored before returning. */

/* replaces: typedef at libflagstats.h:2970 */
typedef int (*FLAGSTATS_func)(const uint16_t*, uint32_t, uint32_t*);

/* replaces: `static uint64_t FLAGSTATS_u16(const uint16_t* array, uint32_t n_len, uint32_t* flags)` libflagstats.h:3024-3070
 * (callers: python/libflagstats.pyx:22).  `array` is a HOST pointer (any 2-byte alignment); flags[32] += counters.
 * Returns 0 like the reference; non-zero only if the GPU path failed. */
uint64_t FLAGSTATS_u16(const uint16_t* array, uint32_t n_len, uint32_t* flags);

/* replaces: `static FLAGSTATS_func FLAGSTATS_get_function(uint32_t n_len)` libflagstats.h:2976-3022 (callers:
 * benchmark/flagstats.cpp:328,450,665).  One kernel family, no host kernels: returns &FLAGSTAT_hip for every n; the
 * length-aware rule (:2999-3021) stays in the reference's dispatcher, which INTEGRATION.md section B extends with this
 * library as its first branch (n >= FLAGSTATS_HIP_MIN_LEN). */
FLAGSTATS_func FLAGSTATS_get_function(uint32_t n_len);

/* the kernel itself, shaped like every FLAGSTAT_<impl> of the reference (FLAGSTAT_avx512 libflagstats.h:1644,
 * FLAGSTAT_scalar :170): host pointer in, flags[32] += counters, returns 0 on success. */
int FLAGSTAT_hip(const uint16_t* array, uint32_t len, uint32_t* flags);

/* replaces: `static int STORM_pospopcnt_u16(const uint16_t* data, size_t len, uint32_t* out)` python/libalgebra.h:3496-3551
 * (SURVEY section 8 f4): out[16] is ZEROED first (:3497), then out[j] = number of words with bit j set.  HOST pointer. */
int STORM_pospopcnt_u16(const uint16_t* data, size_t len, uint32_t* out);

/* ================= extensions the reference's uint32 ABI cannot express ================= */

/* 64-bit length, 64-bit counters, HOST pointer: out[32] += counters.  Streams the array through two device buffers (the
 * H2D copy of chunk k+1 overlaps the kernel on chunk k; truly asynchronous when `array` is page-locked, e.g. from
 * FLAGSTATS_hip_host_alloc; large pageable arrays are copied into page-locked chunks by worker threads). */
int FLAGSTATS_u16_x64(const uint16_t* array, uint64_t n, uint64_t* out);

/* DEVICE-resident array (any 2-byte alignment), DEVICE counters: d_out[32] (uint64, device memory) += counters,
 * asynchronously on `stream` (a hipStream_t passed as void*; NULL = HIP's null stream).  One kernel launch; the adds are
 * atomic, so launches on several streams may share d_out.  d_array may also be page-locked host memory: it is then read in
 * place over PCIe, with no staging copy. */
int FLAGSTATS_hip_device_u16(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream);
/* same, but d_out[32] = counters (all 32 slots written, never-written slots as 0): one query per call with no zeroing
 * launch in front; used by the multi-GPU step before its all-reduce. */
int FLAGSTATS_hip_device_u16_store(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream);
/* DEVICE-resident array, HOST counters: out[32] += counters; synchronous. */
int FLAGSTATS_hip_device_u16_sync(const uint16_t* d_array, uint64_t n, uint64_t* out);

/* Superset forms: the same 19 counters PLUS, counted by the same kernel,
 *   slot 0 / slot 16  primary paired reads, pass-QC / fail-QC  = samtools' n_pair_all[w] (benchmark/flagstats.cpp:58)
 *   slot 9            pass-QC reads = n - slot 25   (the reference's SIMD kernels' "QC adjust", libflagstats.h:1843)
 * -- what the reference's SIMD kernels leave in those slots for their SIMD-covered prefix, here for every flag.
 * With them the whole samtools flagstat report (benchmark/flagstats.cpp:577-588) follows from the 32 slots. */
int FLAGSTATS_u16_x64_superset(const uint16_t* array, uint64_t n, uint64_t* out);                          /* host array */
int FLAGSTATS_hip_device_u16_superset(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream); /* async */
int FLAGSTATS_hip_device_u16_superset_sync(const uint16_t* d_array, uint64_t n, uint64_t* out);

/* 64-bit positional popcount in this library's convention: out[16] += bit counts (host array / device array) */
int FLAGSTATS_hip_pospopcnt_u16_x64(const uint16_t* array, uint64_t n, uint64_t* out);
int FLAGSTATS_hip_device_pospopcnt_u16(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream);

/* ================= context =================
 * State: one engine per device (streams, staging, workspaces), created on first use.  Entry points that take a DEVICE
 * pointer run on the device that pointer lives on; entry points without one use the default device (FLAGSTATS_hip_init,
 * else env FLAGSTATS_HIP_DEVICE, else 0). */
int FLAGSTATS_hip_available(void);          /* 1 if a gfx950-capable device can be used */
int FLAGSTATS_hip_device_count(void);       /* HIP devices visible to this process (0 if none) */
int FLAGSTATS_hip_init(int device);         /* optional: selects the default device (lazily FLAGSTATS_HIP_DEVICE or 0) */
void FLAGSTATS_hip_shutdown(void);          /* releases every engine (close sessions and contexts first) */
const char* FLAGSTATS_hip_last_error(void); /* text of the last failure on this thread ("" if none) */
int FLAGSTATS_hip_device_id(void);          /* device the context is bound to, -1 before init */
int FLAGSTATS_hip_compute_units(void);      /* CU count of that device, -1 before init */
/* fork(): unlike the reference's pure function (libflagstats.h:3024-3070) this library is an engine -- a HIP context,
 * streams, helper threads -- that does not exist in a child fork()ed after its first use.  Every entry point refuses such a
 * child before touching any GPU state: a message naming the fork on stderr and in FLAGSTATS_hip_last_error, a non-zero /
 * NULL return (the reference-shaped entry points then abort() per "on_error"); release-type entries (shutdown, *_free,
 * *_destroy, stream_close) do nothing there.  Use the "spawn" start method, or make the first call after the fork.
 * FLAGSTATS_hip_forked: 1 in such a child, else 0 (touches nothing, claims nothing). */
int FLAGSTATS_hip_forked(void);

/* explicit contexts: a private engine (own streams, staging and workspaces) on `device`; several may exist per device and
 * are fully independent of each other and of the default engines -- one per caller thread gives concurrent host-array calls. */
typedef struct FLAGSTATS_hip_ctx FLAGSTATS_hip_ctx;
FLAGSTATS_hip_ctx* FLAGSTATS_hip_ctx_create(int device);   /* device < 0: the default device; NULL on failure */
void FLAGSTATS_hip_ctx_destroy(FLAGSTATS_hip_ctx* ctx);
int FLAGSTATS_hip_ctx_device(const FLAGSTATS_hip_ctx* ctx);
int FLAGSTATS_hip_ctx_u16_x64(FLAGSTATS_hip_ctx* ctx, const uint16_t* array, uint64_t n, uint64_t* out); /* as FLAGSTATS_u16_x64 */
int FLAGSTATS_hip_ctx_device_u16_sync(FLAGSTATS_hip_ctx* ctx, const uint16_t* d_array, uint64_t n, uint64_t* out);

/* Knobs (also env FLAGSTATS_HIP_<KEY IN CAPITALS>); the ones a caller may want -- the full list, with what each was measured
 * to do, is in libflagstats_hip_probe.h:
 *   "on_error"      reference-shaped entry points on failure: 1 abort() after the message (default), 0 return non-zero
 *   "chunk_flags"   flags per H2D chunk of the host-pointer entries (default 32 Mi = 64 MiB)
 *   "small_flags"   host-pointer calls up to this many flags take the latency path (default 1 Mi; 0 = always stage)
 *   "lz4_decoder" / "zstd_decoder"   block files: 0 host threads, 1 on the GPU, 2 (default) by size
 *   "numa"          1 (default): pinned buffers and worker threads are placed on the GPU's host NUMA node
 * Returns 0 on success. */
int FLAGSTATS_hip_set(const char* key, uint64_t value);
uint64_t FLAGSTATS_hip_get(const char* key);

/* ================= multi-GPU: contiguous shards, counters summed; no other exchange =================
 * rank's shard of n flags over `world` ranks: [begin, end), remainder to the last rank */
void FLAGSTATS_hip_shard_range(uint64_t n, int rank, int world, uint64_t* begin, uint64_t* end);
/* one process, `ndev` devices, HOST array: shard i goes through a private engine on devices[i] (NULL = devices 0..ndev-1; a
 * device may repeat) over that device's PCIe link, one host thread per shard; out[32] += the host-side sum. */
int FLAGSTATS_hip_multi_u16_x64(const uint16_t* array, uint64_t n, const int* devices, int ndev, uint64_t* out);
/* one process, DEVICE-resident shards (d_arrays[i] holds n[i] flags on whichever device it was allocated on): counted where
 * they live, all devices concurrently; out[32] += host-side sum. */
int FLAGSTATS_hip_multi_device_u16(const uint16_t* const* d_arrays, const uint64_t* n, int nshards, uint64_t* out);
/* one process per device (RCCL over xGMI; librccl.so.1 is bound at first use, env FLAGSTATS_HIP_RCCL overrides its path).
 * Communicators travel as void* (an ncclComm_t made here or by the caller):
 *   rank 0: FLAGSTATS_hip_comm_unique_id(id) -> ship the 128 bytes to every rank by any means
 *   all:    comm = FLAGSTATS_hip_comm_init_rank(id, nranks, rank, device)
 *   query:  FLAGSTATS_hip_device_u16_allreduce(d_shard, n, d_out, comm, stream)
 *           = K1 + K2 storing this shard's counters into d_out[32], then ONE
 *             ncclAllReduce(d_out, d_out, 32, ncclUint64, ncclSum) on the same stream. */
int FLAGSTATS_hip_comm_unique_id(void* id128);
void* FLAGSTATS_hip_comm_init_rank(const void* id128, int nranks, int rank, int device); /* NULL on failure */
int FLAGSTATS_hip_comm_destroy(void* comm);
int FLAGSTATS_hip_comm_count(void* comm);   /* ranks RCCL sees in the communicator (ncclCommCount); < 0 on failure */
/* which RCCL is bound: path of the shared object that holds ncclAllReduce (dladdr; "" if unknown) and ncclGetVersion's number
 * (-1 if unknown); binds RCCL if nothing has yet; non-zero when RCCL cannot be loaded */
int FLAGSTATS_hip_comm_library(char* path, uint64_t cap, int* version);
int FLAGSTATS_hip_allreduce_counters(uint64_t* d_counters, void* comm, void* stream);
int FLAGSTATS_hip_device_u16_allreduce(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* comm, void* stream);
/* the same query with the collective OFF the launch stream: K1 + K2 (store) on `stream`, the all-reduce on `comm_stream`,
 * ordered behind the kernels by an event without timing, so it overlaps whatever `stream` runs next.  d_out may be
 * rewritten once `comm_stream` has passed the all-reduce: FLAGSTATS_hip_stream_wait_stream(stream, comm_stream, device) makes
 * `stream` wait on the device (cheap, e.g. once per ring of counter buffers); a host sync of comm_stream works too. */
int FLAGSTATS_hip_device_u16_allreduce_overlapped(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* comm, void* stream,
                                                  void* comm_stream);
int FLAGSTATS_hip_stream_wait_stream(void* waiter, void* on, int device);

/* ================= memory helpers for callers without a HIP runtime of their own ================= */
void* FLAGSTATS_hip_host_alloc(size_t bytes);   /* pinned host memory */
void FLAGSTATS_hip_host_free(void* p);
void* FLAGSTATS_hip_device_alloc(size_t bytes); /* device memory on the default device */
void* FLAGSTATS_hip_device_alloc_on(int device, size_t bytes);
void FLAGSTATS_hip_device_free(void* p);
int FLAGSTATS_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int FLAGSTATS_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int FLAGSTATS_hip_synchronize(void);

/* ================= inputs =================
 * synthetic inputs on device (counterpart of benchmark/generate.cpp:8-14): d_array[k] = flag(kind, seed, mask,
 * first_index + k), k in [0, n).  kind 0 uniform (& mask), 1 NA12878-like (mask bit0 = +eps), 2 ramp.  Asynchronous. */
int FLAGSTATS_hip_generate_u16(uint16_t* d_array, uint64_t n, int kind, uint64_t seed, uint32_t mask,
                               uint64_t first_index, void* stream);
/* FLAG text -> uint16 array (counterpart of benchmark/utility.cpp:9-16, the reference's `samtools view FILE | cut -f 2 |
 * utility > FLAGS.bin` step): one decimal FLAG per line, std::getline + atoi rules (a final unterminated line counts; an
 * empty or non-numeric line is 0; "99\r" is 99; the int is truncated to 16 bits).  Host code.  Returns the number of values
 * written to out[0..cap), or < 0 if `cap` is too small; FLAGSTATS_text_count_lines gives the exact count beforehand. */
int64_t FLAGSTATS_text_to_u16(const char* text, uint64_t len, uint16_t* out, uint64_t cap);
uint64_t FLAGSTATS_text_count_lines(const char* text, uint64_t len);

/* ================= block files: the reference's `bench decompress -d` / `-D` callers =================
 * File format written by benchmark/flagstats.cpp:119-138 and read at :311-316: a sequence of
 *   int32 uncompressed_size, int32 compressed_size, <raw LZ4 block>   (little-endian; not LZ4 frames).
 * The LZ4 path decodes blocks on `threads` host threads (<= 0: up to 24) into pinned chunk buffers, overlapped with the H2D
 * copy and K1 of earlier chunks -- or, for large files (knob "lz4_decoder"), sends the file over PCIe as it is and decodes the
 * blocks on the GPU; out[32] += counters of every flag (a block contributes uncompressed_size >> 1 flags, as
 * benchmark/flagstats.cpp:323). */
typedef struct FLAGSTATS_blockfile_stats {
    uint64_t n_flags, n_blocks, compressed_bytes, uncompressed_bytes;
    double wall_s, index_s, setup_s, decode_cpu_s; /* setup_s: index + buffers; decode_cpu_s: sum over threads */
    double wait_decode_s, wait_copy_s;             /* orchestrator: waiting for decoders / for H2D copies */
    int32_t threads, chunks;
    int32_t gpu_decode, reserved;                  /* 1: the blocks were decoded on the GPU: threads = parallel file readers
                                                      (0 in image mode), chunks = pieces copied, decode_cpu_s = 0,
                                                      wait_copy_s = copies, wait_decode_s = decode left exposed after the last copy */
} FLAGSTATS_blockfile_stats;
int FLAGSTATS_hip_blockfile_lz4(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats);
int FLAGSTATS_hip_blockimage_lz4(const void* image, uint64_t bytes, int threads, uint64_t* out,
                                 FLAGSTATS_blockfile_stats* stats); /* same, file already in memory */
/* Zstandard block files (.zst: benchmark/flagstats.cpp:192-226 writer, :636-682 reader): same block header, payload = one
 * Zstandard frame.  libzstd stays the third-party dependency it is in the reference; it is resolved at run time
 * (libzstd.so.1, or env FLAGSTATS_HIP_ZSTD_LIB) when the host has to decode a .zst file, and the call fails loudly without
 * it.  Large files are decoded on the GPU instead (knob "zstd_decoder"; stats->gpu_decode = 1), which needs libzstd only for
 * files with frames the GPU decoder does not take. */
int FLAGSTATS_hip_blockfile_zstd(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats);
int FLAGSTATS_hip_blockimage_zstd(const void* image, uint64_t bytes, int threads, uint64_t* out,
                                  FLAGSTATS_blockfile_stats* stats);
int FLAGSTATS_hip_zstd_available(void);   /* 1 if libzstd could be loaded */
/* codec by extension as the reference's check_file_extension (benchmark/flagstats.cpp:828-839): .lz4 | .zst */
int FLAGSTATS_hip_blockfile(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats);
/* raw uint16 file (benchmark/flagstats.cpp:415-468, `-D`): the same threaded pipeline without a codec -- workers pread
 * 1 MiB slices straight into the pinned chunks (env FLAGSTATS_HIP_RAW_IO=mmap: mmap + FLAGSTATS_u16_x64) */
int FLAGSTATS_hip_file_raw(const char* path, uint64_t* out, FLAGSTATS_blockfile_stats* stats);
/* A host array in ordinary (pageable) memory through the same chunk pipeline: `threads` workers (0 = automatic) copy 1 MiB
 * slices into the engine's page-locked chunks, which cross PCIe behind them; out[32] += counters.  FLAGSTATS_u16 /
 * FLAGSTATS_u16_x64 do this by themselves for pageable arrays of at least `staged_min_flags` flags (default 2^27); exported
 * for callers that know their arrays are in small pages and want it from 64 MiB (profiles/r05/pageable_c.log). */
int FLAGSTATS_hip_host_staged_u16(const uint16_t* array, uint64_t n, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats);
/* the file entries with SUPERSET counters (slots 0 / 16 = n_pair_all, slot 9 = pass-QC reads): what the samtools report of
 * `bench decompress -s` (block file) / `-S` (raw file) needs, benchmark/flagstats.cpp:577-588 */
int FLAGSTATS_hip_blockfile_superset(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats);
int FLAGSTATS_hip_file_raw_superset(const char* path, uint64_t* out, FLAGSTATS_blockfile_stats* stats);
/* the host LZ4 *block* decoder of the pipeline above (replaces the reference's call to liblz4's LZ4_decompress_safe,
 * benchmark/flagstats.cpp:316): returns decoded bytes, < 0 on malformed input */
int64_t FLAGSTATS_lz4_block_decode(const void* src, uint64_t srclen, void* dst, uint64_t dstcap);

/* ================= streaming sessions =================
 * For callers that keep their own per-block loop and accumulate into one counter array read after the loop, as
 * benchmark/flagstats.cpp:304,311-342 does.  The caller decodes straight into pinned memory handed out by `acquire` (zero
 * copy) and `commit`s; copies and kernels run behind it; `finish` waits and adds the counters of everything committed since
 * the last finish.
 *   FLAGSTATS_hip_stream* s = FLAGSTATS_hip_stream_open();
 *   for each block: uint16_t* p = FLAGSTATS_hip_stream_acquire(s, N);  decode N flags into p;
 *                   FLAGSTATS_hip_stream_commit(s, N);
 *   FLAGSTATS_hip_stream_finish(s, counters);  FLAGSTATS_hip_stream_close(s);
 * `push` = acquire + memcpy + commit for callers that cannot decode in place.  A block may not exceed the chunk size (knob
 * "chunk_flags").  The pointer from `acquire` is valid until `commit`. */
typedef struct FLAGSTATS_hip_stream FLAGSTATS_hip_stream;
FLAGSTATS_hip_stream* FLAGSTATS_hip_stream_open(void);                       /* NULL on failure */
uint16_t* FLAGSTATS_hip_stream_acquire(FLAGSTATS_hip_stream* s, uint64_t n); /* room for n flags; NULL on failure */
int FLAGSTATS_hip_stream_commit(FLAGSTATS_hip_stream* s, uint64_t n);        /* n <= the acquired size */
int FLAGSTATS_hip_stream_push(FLAGSTATS_hip_stream* s, const uint16_t* array, uint64_t n);
int FLAGSTATS_hip_stream_finish(FLAGSTATS_hip_stream* s, uint64_t* out);     /* out[32] += counters; session reusable */
uint64_t FLAGSTATS_hip_stream_flags(const FLAGSTATS_hip_stream* s);          /* flags committed since the last finish */
void FLAGSTATS_hip_stream_close(FLAGSTATS_hip_stream* s);

#ifdef __cplusplus
}
#endif
#endif
