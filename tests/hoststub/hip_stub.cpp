// TEST-ONLY HIP runtime stand-in (see hip/hip_runtime.h next to this file): device memory = host memory, one FIFO worker
// thread per stream, events as generation counters, and the kernels' launchers (fsk_*) replaced by closures that compute
// the same counters with the ORACLE's scalar rule on the stream's thread.  Purpose: run the product's host code -- engine
// registry and locks, the block pipeline's decoder threads / condition variables / pinned-buffer rotation, sessions, the
// multi-device entry -- under ThreadSanitizer.  A buffer that the host re-uses before the "GPU" work on it has been
// waited for is a data race here, and TSan reports it.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

#include "../../include/libflagstats_hip_probe.h"
#include "../../libflagstats_amd/csrc/flagstat_engine.h"
#include "../../libflagstats_amd/csrc/flagstat_kernels.h"

extern "C" {
#include "../../oracle/flagstat_oracle.h"
}

// ------------------------------------------------------------------ streams
struct StubStream {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t submitted = 0, done = 0;
    bool stop = false;
    int device = 0;

    void run()
    {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                fn = std::move(q.front());
                q.pop_front();
            }
            fn();
            {
                std::lock_guard<std::mutex> lk(mu);
                ++done;
            }
            cv.notify_all();
        }
    }
};

struct StubEvent {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t issued = 0, completed = 0;
    std::chrono::steady_clock::time_point when;
};

namespace {

thread_local int t_device = 0;
std::mutex g_mu;                       // stream registry + allocation map
std::shared_mutex g_life;              // a device-wide sync walks the streams (shared); a stream is deleted under it (exclusive):
                                       // without it a hipFree on one thread touched a stream another thread was destroying
                                       // (two sessions closing at once: one ThreadSanitizer report in ~20 runs)
std::vector<StubStream*> g_streams;
std::map<int, StubStream*> g_null;     // per-device NULL stream

struct Alloc {
    size_t bytes;
    hipMemoryType type;
    int device;
};
std::map<uintptr_t, Alloc> g_allocs;

int device_count()
{
    const char* s = std::getenv("FLAGSTATS_STUB_DEVICES");
    return s && *s ? std::atoi(s) : 2;
}

StubStream* make_stream(int device)
{
    StubStream* s = new StubStream();
    s->device = device;
    s->th = std::thread([s] { s->run(); });
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.push_back(s);
    return s;
}

StubStream* resolve(hipStream_t s)
{
    if (s) return s;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_null.find(t_device);
        if (it != g_null.end()) return it->second;
    }
    StubStream* n = make_stream(t_device);
    std::lock_guard<std::mutex> lk(g_mu);
    auto ins = g_null.emplace(t_device, n);
    return ins.first->second;
}

void enqueue(hipStream_t hs, std::function<void()> fn)
{
    StubStream* s = resolve(hs);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->q.push_back(std::move(fn));
        ++s->submitted;
    }
    s->cv.notify_all();
}

void sync_stream(StubStream* s)
{
    std::unique_lock<std::mutex> lk(s->mu);
    const uint64_t target = s->submitted;
    s->cv.wait(lk, [&] { return s->done >= target; });
}

void remember(void* p, size_t bytes, hipMemoryType type)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs[reinterpret_cast<uintptr_t>(p)] = Alloc{bytes, type, t_device};
}

}  // namespace

// ------------------------------------------------------------------ devices
hipError_t hipGetDeviceCount(int* n)
{
    *n = device_count();
    return *n > 0 ? hipSuccess : hipErrorNoDevice;
}
hipError_t hipGetDevice(int* d)
{
    *d = t_device;
    return hipSuccess;
}
hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= device_count()) return hipErrorInvalidValue;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d)
{
    if (d < 0 || d >= device_count()) return hipErrorInvalidValue;
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 8;
    return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char* buf, int len, int d)
{
    std::snprintf(buf, static_cast<size_t>(len), "ffff:ff:%02x.0", d);
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void)
{
    std::shared_lock<std::shared_mutex> life(g_life);
    std::vector<StubStream*> all;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        all = g_streams;
    }
    for (StubStream* s : all)
        if (s->device == t_device) sync_stream(s);
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "stub error"; }

// ------------------------------------------------------------------ memory
hipError_t hipMalloc(void** p, size_t bytes)
{
    *p = std::aligned_alloc(256, (bytes + 255) & ~static_cast<size_t>(255));
    if (!*p) return hipErrorInvalidValue;
    remember(*p, bytes, hipMemoryTypeDevice);
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t, int)
{
    *value = 1;  // "large BAR": the engines then write small inputs straight into (stub) device memory
    return hipSuccess;
}
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipFree(void* p)
{
    if (!p) return hipSuccess;
    (void)hipDeviceSynchronize();  // like the real one: waits for the device's work
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_allocs.erase(reinterpret_cast<uintptr_t>(p));
    }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned)
{
    *p = std::aligned_alloc(4096, (bytes + 4095) & ~static_cast<size_t>(4095));
    if (!*p) return hipErrorInvalidValue;
    remember(*p, bytes, hipMemoryTypeHost);
    return hipSuccess;
}
hipError_t hipHostRegister(void* p, size_t bytes, unsigned)
{
    if (!p || !bytes) return hipErrorInvalidValue;
    remember(p, bytes, hipMemoryTypeHost);
    return hipSuccess;
}
hipError_t hipHostUnregister(void* p)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_allocs.erase(reinterpret_cast<uintptr_t>(p)) ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes)
{
    if (free_bytes) *free_bytes = 8ull << 30;
    if (total_bytes) *total_bytes = 16ull << 30;
    return hipSuccess;
}

hipError_t hipHostFree(void* p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_allocs.erase(reinterpret_cast<uintptr_t>(p));
    }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned)
{
    *dp = hp;
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const uintptr_t x = reinterpret_cast<uintptr_t>(p);
    auto it = g_allocs.upper_bound(x);
    if (it == g_allocs.begin()) return hipErrorInvalidValue;
    --it;
    if (x >= it->first + it->second.bytes) return hipErrorInvalidValue;
    a->type = it->second.type;
    a->device = it->second.device;
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind)
{
    std::memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    if (kind == hipMemcpyHostToDevice) {
        bool pinned = false;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, src) == hipSuccess) pinned = true;
        if (!pinned) {
            // pageable source: the real runtime stages it before the call returns
            std::vector<unsigned char>* tmp = new std::vector<unsigned char>(static_cast<const unsigned char*>(src),
                                                                              static_cast<const unsigned char*>(src) + bytes);
            enqueue(s, [dst, tmp] {
                std::memcpy(dst, tmp->data(), tmp->size());
                delete tmp;
            });
            return hipSuccess;
        }
    }
    enqueue(s, [dst, src, bytes] { std::memcpy(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemset(void* dst, int value, size_t bytes)
{
    // NULL-stream semantics of the real thing: asynchronous, NOT ordered against non-blocking streams
    enqueue(nullptr, [dst, value, bytes] { std::memset(dst, value, bytes); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s)
{
    enqueue(s, [dst, value, bytes] { std::memset(dst, value, bytes); });
    return hipSuccess;
}

// ------------------------------------------------------------------ streams and events
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned)
{
    *s = make_stream(t_device);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s) return hipErrorInvalidValue;
    sync_stream(s);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    s->th.join();
    std::unique_lock<std::shared_mutex> life(g_life);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_streams.size(); ++i)
            if (g_streams[i] == s) {
                g_streams.erase(g_streams.begin() + static_cast<long>(i));
                break;
            }
    }
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    sync_stream(resolve(s));
    return hipSuccess;
}
hipError_t hipStreamGetDevice(hipStream_t s, hipDevice_t* d)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (StubStream* k : g_streams)
        if (k == s) {
            *d = s->device;
            return hipSuccess;
        }
    return hipErrorInvalidValue;
}
hipError_t hipThreadExchangeStreamCaptureMode(hipStreamCaptureMode*) { return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t* e)
{
    *e = new StubEvent();
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    // the real runtime keeps a destroyed event alive until the work that refers to it has run
    {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv.wait(lk, [&] { return e->completed >= e->issued; });
    }
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    uint64_t gen;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        gen = ++e->issued;
    }
    enqueue(s, [e, gen] {
        {
            std::lock_guard<std::mutex> lk(e->mu);
            if (e->completed < gen) e->completed = gen;
            e->when = std::chrono::steady_clock::now();
        }
        e->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    uint64_t gen;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        gen = e->issued;  // a wait binds to the record that was current when it was queued
    }
    enqueue(s, [e, gen] {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv.wait(lk, [&] { return e->completed >= gen; });
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    std::unique_lock<std::mutex> lk(e->mu);
    const uint64_t gen = e->issued;
    e->cv.wait(lk, [&] { return e->completed >= gen; });
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e)
{
    std::lock_guard<std::mutex> lk(e->mu);
    return e->completed >= e->issued ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    std::lock_guard<std::mutex> la(a->mu);
    std::lock_guard<std::mutex> lb(b->mu);
    *ms = std::chrono::duration<float, std::milli>(b->when - a->when).count();
    return hipSuccess;
}

// ------------------------------------------------------------------ the kernels' launchers, computed by the oracle
extern "C" {

void fsk_warm(void) {}
size_t fsk_partials_bytes(uint32_t grid) { return static_cast<size_t>(grid) * fsk::kInternal * sizeof(uint64_t) + 8192; }

hipError_t fsk_launch(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, uint64_t* d_partials, uint32_t*,
                      uint64_t* d_out32, hipStream_t stream, uint64_t* signal_word, uint64_t signal_value)
{
    if (n == 0) return hipSuccess;
    if (!grid || !d_array || !d_partials || !d_out32) return hipErrorInvalidValue;
    const bool store = (variant >> 8) & 1, superset = (variant >> 10) & 1, direct = (variant >> 11) & 1;
    if (signal_word && !store) return hipErrorInvalidValue;
    enqueue(stream, [=] {
        uint64_t c[32];
        std::memset(c, 0, sizeof c);
        oracle_flagstat_u16(d_array, n, c);
        if (superset) {
            for (uint64_t i = 0; i < n; ++i) {
                const uint16_t x = d_array[i];
                const bool fail = x & 0x200;
                if ((x & 1) && !(x & 0x100) && !(x & 0x800)) ++c[fail ? 16 : 0];
                if (!fail) ++c[9];
            }
        }
        if (!direct) d_partials[0] = n;  // the K2 forms write the stream's workspace
        if (signal_word) {
            for (int t = 0; t < 32; ++t) {
                __atomic_store_n(&signal_word[2 * t], c[t], __ATOMIC_RELAXED);
                __atomic_store_n(&signal_word[2 * t + 1], signal_value, __ATOMIC_RELEASE);
            }
        } else if (store) {
            for (int t = 0; t < 32; ++t) d_out32[t] = c[t];
        } else if (direct) {
            for (int t = 0; t < 32; ++t)
                if (c[t]) __atomic_fetch_add(&d_out32[t], c[t], __ATOMIC_RELAXED);  // K1's atomic epilogue
        } else {
            for (int t = 0; t < 32; ++t)
                if (c[t]) d_out32[t] += c[t];                                       // K2's plain +=
        }
    });
    return hipSuccess;
}

hipError_t fsk_launch_pospopcnt(const uint16_t* d_array, uint64_t n, uint32_t grid, uint64_t* d_partials, uint64_t* d_out16,
                                hipStream_t stream, int direct)
{
    if (n == 0) return hipSuccess;
    if (!grid || !d_array || !d_partials || !d_out16) return hipErrorInvalidValue;
    enqueue(stream, [=] {
        uint64_t c[16];
        std::memset(c, 0, sizeof c);
        oracle_pospopcnt_u16(d_array, n, c);
        for (int t = 0; t < 16; ++t) {
            if (direct)
                __atomic_fetch_add(&d_out16[t], c[t], __ATOMIC_RELAXED);
            else
                d_out16[t] += c[t];
        }
    });
    return hipSuccess;
}

hipError_t fsk_generate(uint16_t* d_array, uint64_t n, int kind, uint64_t seed, uint32_t mask, uint64_t first_index, hipStream_t stream)
{
    enqueue(stream, [=] { oracle_generate_u16(kind, seed, mask, first_index, n, d_array); });
    return hipSuccess;
}

hipError_t fsk_read_probe_policy(const void*, uint64_t, int, uint32_t, uint32_t*, hipStream_t) { return hipSuccess; }
hipError_t fsk_read_probe2(const void*, uint64_t, int, int, uint32_t, uint32_t, int, uint32_t*, hipStream_t) { return hipSuccess; }
hipError_t fsk_clock_probe(uint64_t* d_out, uint32_t grid, uint64_t ticks, hipStream_t stream)
{
    enqueue(stream, [=] {
        for (uint32_t b = 0; b < grid; ++b) {
            d_out[2 * b] = ticks * 21;
            d_out[2 * b + 1] = ticks;
        }
    });
    return hipSuccess;
}
int fsk_variant_supported(int variant) { return (variant & 255) == 9 || (variant & 255) == 25; }
void fsk_set_anatomy(int) {}
int fsk_tuning_build(void) { return 0; }
void fsk_set_dyn(uint32_t, uint32_t, uint32_t, uint32_t) {}
void fsk_set_dyn_queues(uint32_t) {}
void fsk_set_group_min_grid(uint32_t) {}
void fsk_set_group_max_steps(uint64_t) {}
int fsk_last_mode(void) { return 0; }
void fsk_set_epoch_stagger(int) {}

}  // extern "C"

// The GPU LZ4 decoder's KERNEL is device code; its host side (flagstat_gpu_decode.hip: pieces, reader pool, span recycling,
// cached buffers) is part of this build.  The stand-in "kernel" decodes the launch's blocks on the stream's worker thread with
// the PRODUCT's own host decoder (lz4_block_decode.h), so copies, decode and counting touch the same buffers in the same
// order as on the device.
#include "../../libflagstats_amd/csrc/flagstat_lz4_kernels.h"
#include "../../libflagstats_amd/csrc/lz4_block_decode.h"

extern "C" hipError_t fsk_lz4_decode(int kernel, const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out,
                                     uint32_t* status, unsigned long long* tally, int, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (!comp || !blocks || !out || !status || !tally || kernel < 0 || kernel > 2) return hipErrorInvalidValue;
    enqueue(stream, [=] {
        for (uint32_t i = 0; i < nblocks; ++i) {
            const fsk::GpuBlock b = blocks[i];
            std::vector<uint8_t> tmp(b.dst_len + 64);
            const int64_t got = fslz4::lz4_block_decode(comp + b.src_off, b.src_len, tmp.data(), b.dst_len);
            if (got == static_cast<int64_t>(b.dst_len)) {
                std::memcpy(out + b.dst_off, tmp.data(), b.dst_len & ~1u);  // (an odd trailing byte is dropped, like the kernel)
                status[i] = 0;
            } else {
                status[i] = 5;
            }
            __atomic_fetch_add(&tally[0], 1ull, __ATOMIC_RELAXED);
        }
    });
    return hipSuccess;
}
extern "C" int fsk_lz4_blocks_per_cu(int) { return 2; }

// ---- the GPU Zstandard decoder's launcher: the stand-in "kernels" are the image's libzstd (what the product's host
// pipeline calls too, resolved at run time like there), on the stream's worker thread
#include <dlfcn.h>

#include "../../libflagstats_amd/csrc/flagstat_zstd_kernels.h"

// (like the kernels: a first pass has table slots for 4 blocks per 128 KiB + 8, a second pass -- _ex with min_blocks -- for more;
// a frame with more Zstandard blocks than that answers kZstdTooManyBlocks, so the host's second pass runs in the sanitizer builds)
static uint32_t stub_blk_cap(uint32_t max_dst_len, uint32_t min_blocks)
{
    uint32_t cap = 4u * ((max_dst_len + 131071u) / 131072u) + 8u;
    if (cap > fsk::kZstdMaxBlocks) cap = fsk::kZstdMaxBlocks;
    if (min_blocks > cap) cap = min_blocks < fsk::kZstdMaxBlocksRetry ? min_blocks : fsk::kZstdMaxBlocksRetry;
    return cap;
}
static uint32_t stub_count_zstd_blocks(const uint8_t* f, uint32_t n)
{
    if (n < 6 || f[0] != 0x28 || f[1] != 0xB5 || f[2] != 0x2F || f[3] != 0xFD) return 0;
    const uint32_t fhd = f[4], single = (fhd >> 5) & 1u, fcs_flag = fhd >> 6, did = fhd & 3u;
    uint32_t p = 5u + (single ? 0u : 1u) + (did == 3u ? 4u : did) + (fcs_flag == 0u ? single : (1u << fcs_flag));
    uint32_t count = 0;
    while (p + 3u <= n) {
        const uint32_t bh = f[p] | (f[p + 1] << 8) | (static_cast<uint32_t>(f[p + 2]) << 16);
        ++count;
        p += 3u + (((bh >> 1) & 3u) == 1u ? 1u : (bh >> 3));
        if (bh & 1u) break;
    }
    return count;
}
extern "C" uint64_t fsk_zstd_scratch_bytes_ex(uint32_t max_dst_len, uint32_t nframes, uint32_t min_blocks)
{
    return 4096 + (static_cast<uint64_t>(max_dst_len) / 64 + 64 + stub_blk_cap(max_dst_len, min_blocks)) * nframes;
}
extern "C" uint64_t fsk_zstd_scratch_bytes(uint32_t max_dst_len, uint32_t nframes) { return fsk_zstd_scratch_bytes_ex(max_dst_len, nframes, 0u); }
extern "C" hipError_t fsk_zstd_decode(const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out, uint32_t* status,
                                      unsigned long long* tally, void* scratch, uint64_t scratch_bytes, uint32_t max_dst_len, int prof, hipStream_t stream)
{
    return fsk_zstd_decode_ex(comp, blocks, nblocks, out, status, tally, scratch, scratch_bytes, max_dst_len, 0u, prof, stream);
}
extern "C" hipError_t fsk_zstd_decode_ex(const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out, uint32_t* status,
                                         unsigned long long* tally, void* scratch, uint64_t scratch_bytes, uint32_t max_dst_len, uint32_t min_blocks, int,
                                         hipStream_t stream)
{
    const uint32_t blk_cap = stub_blk_cap(max_dst_len, min_blocks);
    if (nblocks == 0) return hipSuccess;
    if (!comp || !blocks || !out || !status || !tally || !scratch || scratch_bytes < fsk_zstd_scratch_bytes_ex(max_dst_len, nblocks, min_blocks)) return hipErrorInvalidValue;
    typedef size_t (*decompress_fn)(void*, size_t, const void*, size_t);
    typedef unsigned (*is_error_fn)(size_t);
    static void* handle = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    static decompress_fn decompress = handle ? reinterpret_cast<decompress_fn>(dlsym(handle, "ZSTD_decompress")) : nullptr;
    static is_error_fn is_error = handle ? reinterpret_cast<is_error_fn>(dlsym(handle, "ZSTD_isError")) : nullptr;
    if (!decompress || !is_error) return hipErrorInvalidValue;
    enqueue(stream, [=] {
        uint8_t* sc = static_cast<uint8_t*>(scratch);
        for (uint32_t i = 0; i < nblocks; ++i) {
            const fsk::GpuBlock b = blocks[i];
            if (stub_count_zstd_blocks(comp + b.src_off, b.src_len) > blk_cap) {
                status[i] = fsk::kZstdTooManyBlocks;
                continue;
            }
            std::vector<uint8_t> tmp(b.dst_len + 64);
            const size_t got = decompress(tmp.data(), b.dst_len, comp + b.src_off, b.src_len);
            sc[i & 4095u] = static_cast<uint8_t>(i);  // (the scratch is written by the launch: a use-after-release shows under TSan)
            if (!is_error(got) && got == b.dst_len) {
                std::memcpy(out + b.dst_off, tmp.data(), b.dst_len & ~1u);
                status[i] = 0;
            } else {
                status[i] = 5;
            }
            __atomic_fetch_add(&tally[0], 1ull, __ATOMIC_RELAXED);
        }
    });
    return hipSuccess;
}
extern "C" int fsk_zstd_frames_per_cu(void) { return 2; }
