// flagstat_blocks.hip -- SURVEY.md section 8 row f1: the caller-side step BEFORE the hot path.
//
// Reads the reference's block files and feeds the decoded FLAG stream to K1/K2:
//   * format (benchmark/flagstats.cpp:119-138 writer, :288-332 reader): a sequence of
//       int32 uncompressed_size, int32 compressed_size, <raw LZ4 block>      (native little-endian)
//     -- NOT the LZ4 frame format.  The reference writes 1,024,000-byte blocks
//     (512,000 flags) and, when the input size is a multiple of that, a trailing empty block.
//   * reference pipeline (:311-332): read -> LZ4_decompress_safe -> FLAGSTATS_get_function(N)(...)
//     strictly serially, one thread, ~80 % of the time in I/O + decode (README.md:27-29).
//   * here: index pass over the headers, then N host threads decode blocks straight into their
//     final place of a pinned chunk buffer (block sizes are known from the headers, so there is
//     no ordering between threads), while the previous chunk is on its way over PCIe
//     (hipMemcpyAsync) and the one before is being counted by K1/K2.  3 pinned chunk buffers,
//     2 device buffers, 2 streams.
//
// LZ4 itself is a third-party dependency of the reference's bench (liblz4, found at build time
// via LZ4_PATH, Makefile:28-39; not vendored).  The decoder below is written from the published
// LZ4 *block* format (token = 4 bits literal length | 4 bits match length - 4, 255-continued
// length bytes, 2-byte little-endian offset, last sequence literals-only) and is checked in
// tests/ against the image's liblz4.so.1 (1.9.3) acting as oracle.
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/libflagstats_hip_probe.h"
#include "flagstat_engine.h"
#include "lz4_block_decode.h"

namespace {

using fslz4::lz4_block_decode;

// ---------------------------------------------------------------- block codecs
// The reference's bench writes the same int32,int32 block header around raw LZ4 blocks (.lz4) or Zstandard
// frames (.zst; benchmark/flagstats.cpp:192-226 writer, :636-682 reader calling ZSTD_decompress).  LZ4 is
// decoded by this library's own decoder.  Zstandard is the reference's third-party dependency (libzstd,
// found at build time via ZSTD_PATH, Makefile:28-39) and stays one here: libzstd.so.1 is resolved at run
// time, only when a .zst file is actually opened (as RCCL is in flagstat_multi.hip); without it the call
// fails loudly -- there is no second decoder to fall back to.
typedef int64_t (*block_decode_fn)(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

struct ZstdApi {
    void* handle = nullptr;
    size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
};
ZstdApi g_zstd;
std::once_flag g_zstd_once;
char g_zstd_why[256] = {0};

bool zstd_load()
{
    std::call_once(g_zstd_once, [] {
        const char* env = std::getenv("FLAGSTATS_HIP_ZSTD_LIB");
        const char* names[] = {env && *env ? env : nullptr, "libzstd.so.1", "libzstd.so"};
        for (const char* nm : names) {
            if (!nm) continue;
            g_zstd.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (g_zstd.handle) break;
        }
        if (!g_zstd.handle) {
            std::snprintf(g_zstd_why, sizeof g_zstd_why, "cannot load libzstd.so.1 (needed for .zst block files): %s", dlerror());
            return;
        }
        g_zstd.decompress = reinterpret_cast<decltype(g_zstd.decompress)>(dlsym(g_zstd.handle, "ZSTD_decompress"));
        g_zstd.is_error = reinterpret_cast<decltype(g_zstd.is_error)>(dlsym(g_zstd.handle, "ZSTD_isError"));
        if (!g_zstd.decompress || !g_zstd.is_error)
            std::snprintf(g_zstd_why, sizeof g_zstd_why, "libzstd lacks ZSTD_decompress / ZSTD_isError");
    });
    return g_zstd.decompress && g_zstd.is_error;
}

int64_t zstd_block_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    const size_t r = g_zstd.decompress(dst, cap, src, n);
    return g_zstd.is_error(r) ? -1 : static_cast<int64_t>(r);
}

struct BlockRef {
    const uint8_t* src;   // payload in memory (image mode) or nullptr (file mode: pread at file_off)
    uint64_t file_off;
    uint32_t csize, usize;
    uint64_t dst_off;  // byte offset inside its chunk buffer (16-byte aligned)
    uint32_t chunk;
};

struct ChunkRef {
    size_t b0, b1;      // blocks [b0, b1)
    uint64_t bytes;     // padded bytes used in the chunk buffer (multiple of 16)
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Parse the block headers of a file image.  Every block's decoded bytes get a 16-byte aligned
// slot in a chunk buffer; the gap up to the next slot is zero-filled, and a zero flag counts
// nothing, so a whole chunk is counted as ONE array.
// Chunks: a chunk is closed when the next block would take it over `soft_cap` (the first two chunks: a quarter and a half of it
// when `ramp` -- nothing overlaps the first chunk's decode); a block larger than that has a chunk of its own, up to `hard_cap`.
int index_blocks(const uint8_t* img, int fd, uint64_t bytes, uint64_t soft_cap, uint64_t hard_cap, bool ramp, std::vector<BlockRef>& blocks,
                 std::vector<ChunkRef>& chunks, uint64_t& uncompressed, int codec)
{
    uint64_t pos = 0;
    uncompressed = 0;
    ChunkRef cur{0, 0, 0};
    while (pos < bytes) {
        if (bytes - pos < 8) return fsint::fail_text("block file: truncated block header");
        int32_t us, cs;
        uint8_t hdr[8];
        if (img) {
            std::memcpy(hdr, img + pos, 8);
        } else if (pread(fd, hdr, 8, static_cast<off_t>(pos)) != 8) {
            return fsint::fail_text("block file: cannot read block header");
        }
        std::memcpy(&us, hdr, 4);
        std::memcpy(&cs, hdr + 4, 4);
        pos += 8;
        if (us < 0 || cs < 0) return fsint::fail_text("block file: negative size in block header");
        if (static_cast<uint64_t>(cs) > bytes - pos) return fsint::fail_text("block file: block payload runs past end of file");
        if (!fsint::block_sizes_plausible(codec, static_cast<uint64_t>(us), static_cast<uint64_t>(cs)))
            return fsint::fail_text("block file: a block header declares more decoded bytes than a payload of its size can hold");
        const uint64_t padded = (static_cast<uint64_t>(us) + 15) & ~15ull;
        if (padded > hard_cap) return fsint::fail_text("block file: block larger than the chunk buffer");
        const uint64_t cap_now = ramp && chunks.size() < 2 ? soft_cap >> (2 - chunks.size()) : soft_cap;
        if (cur.bytes && cur.bytes + padded > cap_now) {
            cur.b1 = blocks.size();
            chunks.push_back(cur);
            cur = ChunkRef{blocks.size(), 0, 0};
        }
        blocks.push_back(BlockRef{img ? img + pos : nullptr, pos, static_cast<uint32_t>(cs), static_cast<uint32_t>(us),
                                  cur.bytes, static_cast<uint32_t>(chunks.size())});
        cur.bytes += padded;
        uncompressed += static_cast<uint64_t>(us);
        pos += static_cast<uint64_t>(cs);
    }
    cur.b1 = blocks.size();
    if (cur.b1 > cur.b0) chunks.push_back(cur);
    return 0;
}

}  // namespace

namespace fsint {
// CPUs of a host NUMA node ("/sys/devices/system/node/nodeN/cpulist", e.g. "0-63,128-191")
bool node_cpuset(int node, cpu_set_t* set)
{
    CPU_ZERO(set);
    if (node < 0) return false;
    char path[96];
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = std::fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const bool have = std::fgets(buf, sizeof buf, f) != nullptr;
    std::fclose(f);
    if (!have) return false;
    int count = 0;
    for (char* p = buf; *p && *p != '\n';) {
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            CPU_SET(static_cast<int>(c), set);
            ++count;
        }
        if (*p == ',') ++p;
    }
    if (count == 0) return false;
    // never widen what the process was given (taskset, a container's cpuset): pin to node CPUs INSIDE the current mask,
    // and leave the threads alone if the two do not overlap
    cpu_set_t mine;
    if (sched_getaffinity(0, sizeof mine, &mine) == 0) {
        CPU_AND(set, set, &mine);
        if (CPU_COUNT(set) == 0) return false;
    }
    return true;
}

}  // namespace fsint

namespace {
using fsint::node_cpuset;

// CPUs this process may use per the cgroup v2 CPU controller ("<quota> <period>" or "max <period>"); 0 = no limit known
double cgroup_cpu_quota()
{
    FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r");
    if (!f) return 0;
    char q[32] = {0};
    long period = 0;
    const int got = std::fscanf(f, "%31s %ld", q, &period);
    std::fclose(f);
    if (got != 2 || period <= 0 || !std::strcmp(q, "max")) return 0;
    return std::strtod(q, nullptr) / static_cast<double>(period);
}

struct Pipe {
    std::mutex m;
    std::condition_variable cv_workers;  // "a chunk buffer was released" (orchestrator -> decoders)
    std::condition_variable cv_main;     // "a chunk is fully decoded" (last decoder of the chunk -> orchestrator)
    size_t released = 0;                 // chunks [0, released) may be decoded
    std::vector<size_t> done;            // blocks decoded per chunk
    std::vector<std::atomic<size_t>> next;  // next block to grab per chunk
    bool failed = false;
    double decode_cpu_s = 0;
    explicit Pipe(size_t nchunks) : done(nchunks, 0), next(nchunks) {}
};

constexpr int kPinned = 3;

// img != nullptr: the whole file image is in memory.  img == nullptr: file mode -- every worker
// preads the compressed payload of its block into a private buffer.  (mmap-ing the file instead
// makes all decode threads fault on one address space; the contention grows with the thread
// count and was measured to cost more than the extra copy.)
// where the bytes come from and what they are
struct Source {
    const uint8_t* img = nullptr;              // whole file image in memory (image mode), else nullptr
    int fd = -1;                               // file mode: every worker preads its block
    uint64_t bytes = 0;
    const uint8_t* map = nullptr;              // file mode, opt-in: read-only mapping the payloads are decoded out of
    block_decode_fn decode = lz4_block_decode; // block codec (ignored for raw)
    bool raw = false;                          // headerless uint16 file: 1 MiB slices pread straight into the chunks
    bool superset = false;                     // also count slots 0 / 16 (n_pair_all) and 9 (pass-QC reads)
};

int run_pipeline(fsint::Engine& eng, const Source& in, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* st)
{
    const uint8_t* const img = in.img;
    const int fd = in.fd;
    const uint64_t bytes = in.bytes;
    const uint8_t* const map = in.map;
    const block_decode_fn decode = in.decode;
    const bool raw = in.raw, superset = in.superset;
    const double t0 = now_s();
    // Chunk sizes.  The knob "chunk_flags" (default 64 MiB) is the LARGEST chunk the pipeline makes -- a block up to that size has a
    // chunk of its own; what it aims for is 16 MiB, and a quarter and a half of that for the first two chunks: nothing overlaps the
    // first chunk's decode and nothing the last chunk's copy, and with 64 MiB chunks a file that decodes to 200 MB was hardly
    // pipelined at all (LZ4-fast, 2^25 flags: 2.9 -> 2.1 ms, 2^26: 4.2 -> 3.3; Zstandard 2^26: 7.1 -> 6.3; the raw file of the
    // README's size 32-34 -> 29.6-30.2 ms; large files: the same; profiles/r05/pipeline_chunk_sizes.log).
    // env FLAGSTATS_HIP_PIPE_CHUNK_MIB / FLAGSTATS_HIP_PIPE_RAMP=0: measurement knobs.
    uint64_t hard_cap = (fsint::chunk_bytes() + 15) & ~15ull;
    if (hard_cap < (4ull << 20)) hard_cap = 4ull << 20;
    uint64_t chunk_cap = 16ull << 20;
    if (const char* ck = std::getenv("FLAGSTATS_HIP_PIPE_CHUNK_MIB"))
        if (std::atoi(ck) > 0) chunk_cap = static_cast<uint64_t>(std::atoi(ck)) << 20;
    if (chunk_cap > hard_cap) chunk_cap = hard_cap;
    const char* rk = std::getenv("FLAGSTATS_HIP_PIPE_RAMP");
    const bool ramp = !(rk && std::atoi(rk) == 0);
    std::vector<BlockRef> blocks;
    std::vector<ChunkRef> chunks;
    uint64_t uncompressed = 0;
    int rc = 0;
    if (raw) {
        // raw uint16 file (`bench decompress -D`, benchmark/flagstats.cpp:415-468): no headers, no codec --
        // the "blocks" are 1 MiB slices that the workers pread straight into their place of the pinned
        // chunk (many readers in parallel: one thread copying out of the page cache is the bottleneck
        // of a plain mmap + hipMemcpy, 25 GB/s against the 57 GB/s the bus carries)
        const uint64_t slice = 1ull << 20;
        ChunkRef cur{0, 0, 0};
        for (uint64_t pos = 0; pos < bytes; pos += slice) {
            const uint64_t len = bytes - pos < slice ? bytes - pos : slice;
            const uint64_t padded = (len + 15) & ~15ull;
            const uint64_t cap_now = ramp && chunks.size() < 2 ? (chunk_cap >> (2 - chunks.size())) : chunk_cap;
            if (cur.bytes + padded > cap_now && cur.bytes) {
                cur.b1 = blocks.size();
                chunks.push_back(cur);
                cur = ChunkRef{blocks.size(), 0, 0};
            }
            blocks.push_back(BlockRef{nullptr, pos, static_cast<uint32_t>(len), static_cast<uint32_t>(len), cur.bytes,
                                      static_cast<uint32_t>(chunks.size())});
            cur.bytes += padded;
        }
        cur.b1 = blocks.size();
        if (cur.b1 > cur.b0) chunks.push_back(cur);
        uncompressed = bytes;
    } else {
        rc = index_blocks(img, fd, bytes, chunk_cap, hard_cap, ramp, blocks, chunks, uncompressed, in.decode == lz4_block_decode ? 0 : 1);
    }
    if (rc) return rc;
    if (map)  // file mode with a mapping: headers were pread (no fault per block), payloads are decoded in place
        for (BlockRef& b : blocks) b.src = map + b.file_off;
    const double t_index = now_s() - t0;
    uint64_t n_flags = 0;
    for (const BlockRef& b : blocks) n_flags += b.usize >> 1;  // as benchmark/flagstats.cpp:323
    if (threads <= 0) {
        threads = static_cast<int>(std::thread::hardware_concurrency());
        if (threads > 24) threads = 24;  // ~PCIe-bound from 16-24 decoders on 2x EPYC 9575F (profiles/r01, r02)
        // a container's CPU quota (cgroup v2 cpu.max) is what the decoders really get: more runnable
        // threads than that are throttled, not run (the GPU boxes hand a 1-GPU job 16 of 256 CPUs)
        const double quota = cgroup_cpu_quota();
        if (quota > 0 && threads > static_cast<int>(quota * 1.25)) threads = static_cast<int>(quota * 1.25);
        if (threads < 1) threads = 1;
    }
    if (static_cast<size_t>(threads) > blocks.size() && !blocks.empty()) threads = static_cast<int>(blocks.size());

    // the engine's host pipeline (staging buffers, two streams, counters, pinned chunks) is one resource
    std::lock_guard<std::mutex> lk(eng.mu);
    if (fsint::engine_alive(eng)) return -1;   // (a call racing FLAGSTATS_hip_shutdown: the engine's streams and buffers are gone)
    fsint::lz4_gpu_other_use(eng);  // (the GPU LZ4 decoder's kept buffers go after eight calls that did not use it)
    fsint::DeviceGuard guard(eng.device);
    if (!guard.ok()) return -1;
    rc = fsint::engine_second(eng);   // both streams are used below
    if (rc) return rc;
    uint64_t buf_cap = chunk_cap;   // what the chunk buffers must hold: the largest chunk (a large block has one of its own)
    for (const ChunkRef& c : chunks) buf_cap = c.bytes > buf_cap ? c.bytes : buf_cap;
    buf_cap = (buf_cap + 15) & ~15ull;
    if (!chunks.empty()) {
        for (int s = 0; s < 2 && !rc; ++s) rc = fsint::stage_reserve(eng, s, buf_cap / 2);
        if (rc) return rc;
    }
    uint8_t* pinned[kPinned] = {nullptr, nullptr, nullptr};
    hipEvent_t copied[kPinned];
    const int npin = chunks.size() < static_cast<size_t>(kPinned) ? static_cast<int>(chunks.size()) : kPinned;
    if (npin) {
        void* bufs[3];
        rc = fsint::pinned_reserve(eng, buf_cap, bufs);  // allocated once per process: hipHostMalloc costs ~10 ms / 64 MiB
        if (rc) return rc;
        for (int i = 0; i < kPinned; ++i) pinned[i] = static_cast<uint8_t*>(bufs[i]);
    }
    for (int i = 0; i < npin; ++i) {
        // blocking sync: the orchestrator sleeps in hipEventSynchronize instead of spinning on a CPU
        // the decoders could use (it waits ~70 % of the wall time once the pipeline is PCIe-bound)
        hipError_t e = hipEventCreateWithFlags(&copied[i], hipEventDisableTiming | hipEventBlockingSync);
        if (e != hipSuccess) return fsint::fail_hip("hipEventCreate", e);
    }
    const double t_setup = now_s() - t0;
    for (int s = 0; s < 2; ++s) {
        hipError_t e = hipMemsetAsync(eng.d_out[s], 0, 32 * sizeof(uint64_t), eng.stream[s]);
        if (e != hipSuccess) return fsint::fail_hip("hipMemsetAsync", e);
    }

    Pipe pipe(chunks.size());
    for (auto& a : pipe.next) a.store(0);
    // decoders run on the CPUs of the GPU's own NUMA node, next to the pinned chunk buffers
    // (fsint::pinned_reserve places those there): decoded bytes then cross no socket link on their
    // way to the PCIe root.  Knob "numa" / FLAGSTATS_HIP_NUMA=0 turns both off.
    cpu_set_t node_cpus;
    const bool pin_threads = fsint::knobs().numa.load() && node_cpuset(eng.numa_node, &node_cpus);
    auto worker = [&]() {
        if (pin_threads) (void)pthread_setaffinity_np(pthread_self(), sizeof node_cpus, &node_cpus);
        double busy = 0;
        std::vector<uint8_t> local;  // file mode: this thread's copy of the compressed payload
        for (size_t c = 0; c < chunks.size(); ++c) {
            {
                std::unique_lock<std::mutex> ul(pipe.m);
                pipe.cv_workers.wait(ul, [&] { return pipe.released > c || pipe.failed; });
                if (pipe.failed) break;
            }
            uint8_t* base = pinned[c % kPinned];
            size_t mine = 0;
            bool bad = false;
            const double w0 = now_s();
            for (;;) {
                const size_t b = chunks[c].b0 + pipe.next[c].fetch_add(1);
                if (b >= chunks[c].b1) break;
                const BlockRef& br = blocks[b];
                uint8_t* dst = base + br.dst_off;
                const uint64_t padded = (static_cast<uint64_t>(br.usize) + 15) & ~15ull;
                const uint8_t* src = br.src;
                if (src && map && br.csize) {
                    // map this block's pages of the page cache in ONE call (Linux >= 5.14) instead of one
                    // fault per 16 pages; where the kernel does not know the advice, faults do the same lazily
                    const uintptr_t a0 = reinterpret_cast<uintptr_t>(src) & ~static_cast<uintptr_t>(4095);
                    const uintptr_t a1 = (reinterpret_cast<uintptr_t>(src) + br.csize + 4095) & ~static_cast<uintptr_t>(4095);
                    (void)madvise(reinterpret_cast<void*>(a0), a1 - a0, 22 /* MADV_POPULATE_READ */);
                }
                if (raw && img) {
                    // (a pageable array in memory: the workers' copies into the page-locked chunk replace the runtime's own
                    // pin-as-you-go copy, which moves 30-49 GB/s out of 4 KiB pages it has not seen before: count_host_shared)
                    std::memcpy(dst, img + br.file_off, br.usize);
                    const uint64_t keep = br.usize & ~1ull;
                    std::memset(dst + keep, 0, padded - keep);
                    ++mine;
                    continue;
                }
                if (raw) {
                    size_t have = 0;
                    while (have < br.usize) {
                        const ssize_t r = pread(fd, dst + have, br.usize - have, static_cast<off_t>(br.file_off + have));
                        if (r <= 0) break;
                        have += static_cast<size_t>(r);
                    }
                    if (have != br.usize) {
                        bad = true;
                        break;
                    }
                    const uint64_t keep = br.usize & ~1ull;   // a trailing odd byte of the file is dropped (:450 `read >> 1`)
                    std::memset(dst + keep, 0, padded - keep);
                    ++mine;
                    continue;
                }
                if (!src) {
                    if (local.size() < br.csize) local.resize(br.csize + (br.csize >> 2) + 64);
                    size_t have = 0;
                    while (have < br.csize) {
                        const ssize_t r = pread(fd, local.data() + have, br.csize - have, static_cast<off_t>(br.file_off + have));
                        if (r <= 0) break;
                        have += static_cast<size_t>(r);
                    }
                    if (have != br.csize) {
                        bad = true;
                        break;
                    }
                    src = local.data();
                }
                const int64_t got = decode(src, br.csize, dst, br.usize);
                if (got != static_cast<int64_t>(br.usize)) {
                    bad = true;
                    break;
                }
                // an odd trailing byte is dropped like the reference's N = size >> 1; pad with zero flags
                uint64_t keep = br.usize & ~1ull;
                std::memset(dst + keep, 0, padded - keep);
                ++mine;
            }
            busy += now_s() - w0;
            std::lock_guard<std::mutex> g(pipe.m);
            pipe.done[c] += mine;
            if (bad) {
                pipe.failed = true;
                pipe.cv_workers.notify_all();
            }
            // one wake-up per chunk, not one per decoder (a shared condvar made this O(threads^2))
            if (bad || pipe.done[c] == chunks[c].b1 - chunks[c].b0) pipe.cv_main.notify_one();
            if (bad) break;
        }
        std::lock_guard<std::mutex> g(pipe.m);
        pipe.decode_cpu_s += busy;
    };
    const double t_a = now_s();
    // the engine's worker pool (threads made on first need, parked between calls): making and joining them per call was ~0.3 ms
    fsint::WorkerPool& pool = fsint::engine_pool(eng);
    const bool pooled = threads > 0 && !chunks.empty();
    if (pooled && !pool.start(threads, [&worker](int) { worker(); })) {
        for (int i = 0; i < npin; ++i) (void)hipEventDestroy(copied[i]);
        return -1;
    }
    const double t_b = now_s();

    auto release = [&](size_t upto) {
        std::lock_guard<std::mutex> g(pipe.m);
        if (upto > pipe.released) pipe.released = upto;
        pipe.cv_workers.notify_all();
    };
    release(static_cast<size_t>(npin));
    int err = 0;
    double wait_decode = 0, wait_copy = 0;
    for (size_t c = 0; c < chunks.size() && !err; ++c) {
        {
            const double w0 = now_s();
            std::unique_lock<std::mutex> ul(pipe.m);
            pipe.cv_main.wait(ul, [&] { return pipe.done[c] == chunks[c].b1 - chunks[c].b0 || pipe.failed; });
            if (pipe.failed) {
                err = fsint::fail_text(raw ? "raw file: short read" : "block file: a block failed to decode to its declared size");
                break;
            }
            wait_decode += now_s() - w0;
        }
        const int sl = static_cast<int>(c & 1);
        const int pb = static_cast<int>(c % kPinned);
        hipError_t e = hipMemcpyAsync(eng.stage[sl], pinned[pb], chunks[c].bytes, hipMemcpyHostToDevice, eng.stream[sl]);
        if (e == hipSuccess) e = hipEventRecord(copied[pb], eng.stream[sl]);
        if (e != hipSuccess) {
            err = fsint::fail_hip("hipMemcpyAsync(chunk)", e);
            break;
        }
        err = fsint::count_device_async(eng, eng.stage[sl], chunks[c].bytes / 2, eng.d_out[sl], eng.stream[sl], eng.ws[sl],
                                        fsint::OP_FLAGSTAT | (superset ? fsint::OP_SUPERSET : 0));
        if (err) break;
        if (c >= 1 && (c - 1) + kPinned < chunks.size()) {
            // The pinned buffer of chunk c-1 is reusable once ITS copy has left the host.  Waiting for
            // it only now -- with copy(c) already queued behind it -- keeps the copy engine busy
            // back to back; waiting for copy(c) itself here left it idle between chunks.
            const double w1 = now_s();
            e = hipEventSynchronize(copied[(c - 1) % kPinned]);
            wait_copy += now_s() - w1;
            if (e != hipSuccess) {
                err = fsint::fail_hip("hipEventSynchronize", e);
                break;
            }
            release((c - 1) + kPinned + 1);
        }
    }
    if (err) {
        std::lock_guard<std::mutex> g(pipe.m);
        pipe.failed = true;
        pipe.cv_workers.notify_all();
    }
    const double t_c = now_s();
    if (pooled) pool.wait();
    const double t_d = now_s();
    if (!err) {
        for (int s = 0; s < 2 && !err; ++s) {
            hipError_t e = hipMemcpyAsync(eng.h_out + 32 * s, eng.d_out[s], 32 * sizeof(uint64_t),
                                          hipMemcpyDeviceToHost, eng.stream[s]);
            if (e != hipSuccess) err = fsint::fail_hip("hipMemcpyAsync(counters)", e);
        }
    }
    for (int s = 0; s < 2; ++s) {
        hipError_t e = hipStreamSynchronize(eng.stream[s]);
        if (e != hipSuccess && !err) err = fsint::fail_hip("hipStreamSynchronize", e);
    }
    for (int i = 0; i < npin; ++i) (void)hipEventDestroy(copied[i]);
    if (getenv("FLAGSTATS_HIP_TRACE"))
        fprintf(stderr, "blocks: workers started %.4f loop %.4f workers done %.4f sync %.4f\n", t_b - t_a, t_c - t_b, t_d - t_c, now_s() - t_d);
    if (err) return err;
    if (superset) {
        // slot 9 = pass-QC reads = flags - fail-QC reads is taken per launch over the whole chunk, and a chunk
        // carries zero flags between its blocks (16-byte slots, dropped odd bytes): those are not reads
        uint64_t counted = 0;
        for (const ChunkRef& c : chunks) counted += c.bytes / 2;
        eng.h_out[9] -= counted - n_flags;
    }
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < 32; ++k) out[k] += eng.h_out[32 * s + k];
    if (st) {
        st->n_flags = n_flags;
        st->n_blocks = blocks.size();
        st->compressed_bytes = bytes;
        st->uncompressed_bytes = uncompressed;
        st->wall_s = now_s() - t0;
        st->index_s = t_index;
        st->setup_s = t_setup;
        st->wait_decode_s = wait_decode;
        st->wait_copy_s = wait_copy;
        st->decode_cpu_s = pipe.decode_cpu_s;
        st->threads = threads;
        st->chunks = static_cast<int32_t>(chunks.size());
        st->gpu_decode = 0;
        st->reserved = 0;
    }
    return 0;
}

struct Mapped {
    const uint8_t* p = nullptr;
    uint64_t bytes = 0;
    int fd = -1;
    ~Mapped()
    {
        if (p && bytes) munmap(const_cast<uint8_t*>(p), bytes);
        if (fd >= 0) close(fd);
    }
};

int map_file(const char* path, Mapped& m)
{
    if (!path) return fsint::fail_text("NULL path");
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) return fsint::fail_text("cannot open file");
    struct stat sb;
    if (fstat(m.fd, &sb) != 0) return fsint::fail_text("cannot stat file");
    m.bytes = static_cast<uint64_t>(sb.st_size);
    if (m.bytes == 0) return 0;
    void* p = mmap(nullptr, m.bytes, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (p == MAP_FAILED) {
        m.bytes = 0;
        return fsint::fail_text("cannot mmap file");
    }
    (void)madvise(p, m.bytes, MADV_SEQUENTIAL);
    m.p = static_cast<const uint8_t*>(p);
    return 0;
}

}  // namespace

extern "C" {

int64_t FLAGSTATS_lz4_block_decode(const void* src, uint64_t srclen, void* dst, uint64_t dstcap)
{
    if (!src || (!dst && dstcap)) return -1;
    return lz4_block_decode(static_cast<const uint8_t*>(src), srclen, static_cast<uint8_t*>(dst), dstcap);
}

}  // extern "C"

namespace {

// codec: 0 = raw LZ4 blocks, 1 = Zstandard frames
int pick_decoder(int codec, block_decode_fn* fn)
{
    if (codec == 0) {
        *fn = lz4_block_decode;
        return 0;
    }
    if (!zstd_load()) return fsint::fail_text(g_zstd_why[0] ? g_zstd_why : "libzstd is not available");
    *fn = zstd_block_decode;
    return 0;
}

// Block files: knob "lz4_decoder" / "zstd_decoder" -- 0 host threads, 1 GPU, 2 (default) by size: GPU for files of at least
// "lz4_gpu_min_bytes" / "zstd_gpu_min_bytes" compressed bytes, and for smaller ones that DECODE to at least 2.5 x that (known after
// the GPU path's index pass, which hands the file back otherwise: lz4_gpu_run) -- what a file costs either decoder goes with its
// decoded size: the two cross at 120-200 MB decoded for LZ4-fast, LZ4-HC-9 and Zstandard 1 / 3 / 19 alike, which is 27-95 MiB of
// file (profiles/r05/decoder_crossover.log).  Files under a quarter of the knob are not even indexed twice.
bool decode_on_gpu(int codec, uint64_t bytes)
{
    const int mode = codec == 0 ? fsint::knobs().lz4_decoder.load() : fsint::knobs().zstd_decoder.load();
    const uint64_t from = codec == 0 ? fsint::knobs().lz4_gpu_min_bytes.load() : fsint::knobs().zstd_gpu_min_bytes.load();
    return mode == 1 || (mode == 2 && bytes >= from / 4);
}

// > 0: the GPU decoder did not take the file and nothing was counted -- the caller decodes it on host threads
int run_gpu_lz4(fsint::Engine& eng, int codec, const uint8_t* img, int fd, uint64_t bytes, int threads, bool superset, uint64_t* out,
                FLAGSTATS_blockfile_stats* st)
{
    fsint::Lz4GpuSource src;
    src.img = img;
    src.fd = fd;
    src.bytes = bytes;
    src.superset = superset;
    src.threads = threads;
    src.codec = codec;
    src.by_size = (codec == 0 ? fsint::knobs().lz4_decoder.load() : fsint::knobs().zstd_decoder.load()) == 2;
    // the index pass and the size rules come first, WITHOUT the engine's lock: a 16-64 MiB file that ends up on the host threads
    // is handed back here and never makes concurrent small callers wait for its header reads
    fsint::GpuFileIndexPtr index;
    const int irc = fsint::lz4_gpu_index(src, index);
    if (irc) return irc;
    std::lock_guard<std::mutex> lk(eng.mu);
    if (fsint::engine_alive(eng)) return -1;
    fsint::DeviceGuard guard(eng.device);
    if (!guard.ok()) return -1;
    FLAGSTATS_gpu_lz4_stats g;
    const int rc = fsint::lz4_gpu_run(eng, src, *index, out, &g);
    const bool forced = (codec == 0 ? fsint::knobs().lz4_decoder.load() : fsint::knobs().zstd_decoder.load()) == 1;
    if (rc == fsint::kLz4GpuNoMemory && forced)
        return fsint::fail_text("GPU block decoder: the device cannot hold the file's compressed and decoded bytes");
    if (rc == fsint::kGpuDecodeRejected && forced) return -1;  // (the message names the frame and the code)
    if (rc) return rc;  // (> 0 with the decoder chosen by size: the caller takes the host-thread pipeline)
    if (st) {
        *st = FLAGSTATS_blockfile_stats{};
        st->n_flags = g.n_flags;
        st->n_blocks = g.n_blocks;
        st->compressed_bytes = bytes;
        st->uncompressed_bytes = g.uncompressed_bytes;
        st->wall_s = g.wall_s;
        st->wait_copy_s = g.h2d_ms * 1e-3;
        st->wait_decode_s = g.decode_ms * 1e-3;
        st->threads = static_cast<int32_t>(g.readers);
        st->chunks = static_cast<int32_t>(g.chunks);
        st->gpu_decode = 1;
    }
    return 0;
}

int blockimage(const void* image, uint64_t bytes, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats, int codec,
               bool superset = false)
{
    if (fsint::process_guard("FLAGSTATS_hip_blockimage_*")) return -1;
    if (!out) return fsint::fail_text("NULL out");
    if (!image && bytes) return fsint::fail_text("NULL image");
    block_decode_fn fn = nullptr;
    const int prc = pick_decoder(codec, &fn);  // (libzstd missing: only fatal if the host has to decode)
    if (prc && !decode_on_gpu(codec, bytes)) return prc;
    static const uint8_t empty = 0;
    fsint::Engine* eng = fsint::default_engine();
    if (!eng) return -1;
    if (decode_on_gpu(codec, bytes)) {
        const int rc = run_gpu_lz4(*eng, codec, image ? static_cast<const uint8_t*>(image) : &empty, -1, bytes, threads, superset, out, stats);
        if (rc != fsint::kLz4GpuNoMemory && rc != fsint::kGpuDecodeRejected) return rc;
    }
    if (prc) return prc;
    Source in;
    in.img = image ? static_cast<const uint8_t*>(image) : &empty;
    in.bytes = bytes;
    in.decode = fn;
    in.superset = superset;
    return run_pipeline(*eng, in, threads, out, stats);
}

int blockfile(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats, int codec, bool superset = false)
{
    if (fsint::process_guard("FLAGSTATS_hip_blockfile*")) return -1;
    if (!out) return fsint::fail_text("NULL out");
    if (!path) return fsint::fail_text("NULL path");
    block_decode_fn fn = nullptr;
    const int prc = pick_decoder(codec, &fn);  // (libzstd missing: only fatal if the host has to decode)
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fsint::fail_text("cannot open file");
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        return fsint::fail_text("cannot stat file");
    }
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    // (POSIX_FADV_WILLNEED here -- so that a file that is not in the page cache starts coming in while the first call of a process
    // still creates its engine -- was measured and does not pay: one-shot processes without an explicit init, file evicted before
    // every sample: 204 ms against 178 for the README-size HC-9 file, 333 against 319 for the raw file; the advice walks / queues the
    // whole file on the calling thread first: profiles/r05/cold_start_evicted_lazy_init.log)
    // File mode: every worker preads the compressed payload of its block into a private buffer (30 % of
    // the decoders' CPU time).  FLAGSTATS_HIP_BLOCK_IO=mmap decodes straight out of a read-only mapping
    // instead (each worker populates its own block's pages with one madvise): built and measured SLOWER
    // on the page cache of the r02 boxes -- populating and tearing down 490 k PTEs costs what the copy
    // costs (19-22 vs 24-26 Gflags/s, profiles/r02/blockfile_lz4_4GiB.log) -- so it is opt-in.
    const uint64_t bytes = static_cast<uint64_t>(sb.st_size);
    if (prc && !decode_on_gpu(codec, bytes)) {
        close(fd);
        return prc;
    }
    if (decode_on_gpu(codec, bytes)) {
        fsint::Engine* eng = fsint::default_engine();
        const int rc = eng ? run_gpu_lz4(*eng, codec, nullptr, fd, bytes, threads, superset, out, stats) : -1;
        if (rc != fsint::kLz4GpuNoMemory && rc != fsint::kGpuDecodeRejected) {
            close(fd);
            return rc;
        }
    }
    if (prc) {
        close(fd);
        return prc;
    }
    const char* io = std::getenv("FLAGSTATS_HIP_BLOCK_IO");
    void* map = MAP_FAILED;
    if (bytes && io && !std::strcmp(io, "mmap")) map = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    fsint::Engine* eng = fsint::default_engine();
    Source in;
    in.fd = fd;
    in.bytes = bytes;
    in.map = map == MAP_FAILED ? nullptr : static_cast<const uint8_t*>(map);
    in.decode = fn;
    in.superset = superset;
    const int rc = eng ? run_pipeline(*eng, in, threads, out, stats) : -1;
    if (map != MAP_FAILED) munmap(map, bytes);
    close(fd);
    return rc;
}

}  // namespace

extern "C" {

int FLAGSTATS_hip_blockimage_lz4(const void* image, uint64_t bytes, int threads, uint64_t* out,
                                 FLAGSTATS_blockfile_stats* stats)
{
    return blockimage(image, bytes, threads, out, stats, 0);
}

int FLAGSTATS_hip_blockfile_lz4(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats)
{
    return blockfile(path, threads, out, stats, 0);
}

int FLAGSTATS_hip_blockimage_zstd(const void* image, uint64_t bytes, int threads, uint64_t* out,
                                  FLAGSTATS_blockfile_stats* stats)
{
    return blockimage(image, bytes, threads, out, stats, 1);
}

int FLAGSTATS_hip_blockfile_zstd(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats)
{
    return blockfile(path, threads, out, stats, 1);
}

// codec by file extension, as the reference's check_file_extension (benchmark/flagstats.cpp:828-839):
// ".zst" -> Zstandard, ".lz4" -> LZ4; anything else is refused (the reference prints and exits).
int FLAGSTATS_hip_blockfile(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats)
{
    if (!path) return fsint::fail_text("NULL path");
    const char* dot = std::strrchr(path, '.');
    if (dot && !std::strcmp(dot + 1, "zst")) return blockfile(path, threads, out, stats, 1);
    if (dot && !std::strcmp(dot + 1, "lz4")) return blockfile(path, threads, out, stats, 0);
    return fsint::fail_text("block file: unknown extension (expected .lz4 or .zst)");
}

// the same with SUPERSET counters (slots 0 / 16 = primary paired reads, slot 9 = pass-QC reads): everything the
// samtools report of `bench decompress -s` / `-S` needs (benchmark/flagstats.cpp:577-588)
int FLAGSTATS_hip_blockfile_superset(const char* path, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats)
{
    if (!path) return fsint::fail_text("NULL path");
    const char* dot = std::strrchr(path, '.');
    if (dot && !std::strcmp(dot + 1, "zst")) return blockfile(path, threads, out, stats, 1, true);
    if (dot && !std::strcmp(dot + 1, "lz4")) return blockfile(path, threads, out, stats, 0, true);
    return fsint::fail_text("block file: unknown extension (expected .lz4 or .zst)");
}

int FLAGSTATS_hip_zstd_available(void) { return zstd_load() ? 1 : 0; }

}  // extern "C"

namespace {
// a host array through the chunk pipeline (1 MiB slices copied by the workers)
int host_staged(fsint::Engine& eng, const uint16_t* array, uint64_t n, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats, bool superset)
{
    if (fsint::process_guard("FLAGSTATS_hip_host_staged_u16")) return -1;
    if (!out) return fsint::fail_text("NULL out");
    if (n && !array) return fsint::fail_text("NULL array with n > 0");
    Source in;
    in.img = reinterpret_cast<const uint8_t*>(array);
    in.bytes = n * 2;
    in.raw = true;
    in.superset = superset;
    return run_pipeline(eng, in, threads, out, stats);
}

int file_raw(const char* path, uint64_t* out, FLAGSTATS_blockfile_stats* stats, bool superset)
{
    if (fsint::process_guard("FLAGSTATS_hip_file_raw*")) return -1;
    if (!out) return fsint::fail_text("NULL out");
    if (!path) return fsint::fail_text("NULL path");
    const char* io = std::getenv("FLAGSTATS_HIP_RAW_IO");
    if (io && !std::strcmp(io, "mmap")) {
        // the r01 form: mmap + the host-array path (one thread faults and copies: ~25 GB/s)
        const double t0 = now_s();
        Mapped m;
        int rc = map_file(path, m);
        if (rc) return rc;
        const uint64_t n = m.bytes / 2;  // a trailing odd byte is dropped, as `read >> 1` at benchmark/flagstats.cpp:450
        fsint::Engine* eng = fsint::default_engine();
        if (!eng) return -1;
        rc = fsint::count_host(*eng, reinterpret_cast<const uint16_t*>(m.p), n, out,
                               fsint::OP_FLAGSTAT | (superset ? fsint::OP_SUPERSET : 0));
        if (rc) return rc;
        if (stats) {
            std::memset(stats, 0, sizeof *stats);
            stats->n_flags = n;
            stats->compressed_bytes = stats->uncompressed_bytes = m.bytes;
            stats->wall_s = now_s() - t0;
        }
        return 0;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fsint::fail_text("cannot open file");
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        return fsint::fail_text("cannot stat file");
    }
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    // (POSIX_FADV_WILLNEED here -- so that a file that is not in the page cache starts coming in while the first call of a process
    // still creates its engine -- was measured and does not pay: one-shot processes without an explicit init, file evicted before
    // every sample: 204 ms against 178 for the README-size HC-9 file, 333 against 319 for the raw file; the advice walks / queues the
    // whole file on the calling thread first: profiles/r05/cold_start_evicted_lazy_init.log)
    fsint::Engine* eng = fsint::default_engine();
    Source in;
    in.fd = fd;
    in.bytes = static_cast<uint64_t>(sb.st_size);
    in.raw = true;
    in.superset = superset;
    const int rc = eng ? run_pipeline(*eng, in, 0, out, stats) : -1;
    close(fd);
    return rc;
}
}  // namespace

namespace fsint {
int count_host_staged(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, bool superset, int threads)
{
    return host_staged(e, h, n, threads, out, nullptr, superset);
}
}  // namespace fsint

extern "C" {

int FLAGSTATS_hip_host_staged_u16(const uint16_t* array, uint64_t n, int threads, uint64_t* out, FLAGSTATS_blockfile_stats* stats)
{
    FS_ENTRY();
    fsint::Engine* eng = fsint::default_engine();
    return eng ? host_staged(*eng, array, n, threads, out, stats, false) : -1;
}

int FLAGSTATS_hip_file_raw(const char* path, uint64_t* out, FLAGSTATS_blockfile_stats* stats)
{
    return file_raw(path, out, stats, false);
}

int FLAGSTATS_hip_file_raw_superset(const char* path, uint64_t* out, FLAGSTATS_blockfile_stats* stats)
{
    return file_raw(path, out, stats, true);
}

}  // extern "C"
