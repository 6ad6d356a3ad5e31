// flagstat_session.hip -- streaming sessions for callers that keep their own per-block loop.
//
// The reference's block reader is such a caller (benchmark/flagstats.cpp:311-332): per block it
// decompresses into a buffer and calls the flagstat kernel, accumulating into one counters[32]
// that it prints after the loop.  Through the synchronous drop-in entry every such call costs a
// PCIe round trip (66 us per 512,000-flag block).  A session removes the wait: the caller obtains
// pinned memory (`acquire`), decodes straight into it (zero copy), `commit`s, and goes on to its
// next block while the H2D copy and K1/K2 run behind it; `finish` returns the accumulated
// counters -- the same accumulate-then-read contract, with the counting off the caller's thread.
//
// Blocks are packed into pinned chunk buffers (16-byte aligned slots, zero-filled gaps: a zero
// flag counts nothing, so a chunk is counted as ONE array); 3 pinned chunks, 2 device chunks,
// 2 streams, K1 workspaces and device counters, all private to the session: sessions of different
// caller threads (and the engine's own host entry points) overlap freely; a session's calls are
// serialised by its own mutex only.
#include <cstring>
#include <mutex>

#include "../../include/libflagstats_hip.h"
#include "flagstat_engine.h"

struct FLAGSTATS_hip_stream {
    fsint::Engine* eng = nullptr;
    std::mutex mu;
    hipStream_t st[2] = {nullptr, nullptr};
    fsint::Workspace ws[2];
    uint8_t* pinned[3] = {nullptr, nullptr, nullptr};
    fsint::RegisteredHost pinned_reg[3];   // what they are made of (registered huge pages: opening a session does not wait for hipHostMalloc)
    hipEvent_t copied[3];
    bool have_event[3] = {false, false, false};
    bool in_flight[3] = {false, false, false};
    uint16_t* dstage[2] = {nullptr, nullptr};
    uint64_t* d_out[2] = {nullptr, nullptr};
    uint64_t* h_out = nullptr;  // pinned 2 x 32
    uint64_t cap = 0;           // bytes per chunk
    int cur = 0;                // pinned chunk being filled
    uint64_t used = 0;          // bytes used in it
    uint64_t acquired = 0;      // flags handed out by the pending acquire
    uint64_t submitted = 0;     // chunks submitted so far
    uint64_t flags = 0;         // flags committed since the last finish
};

namespace {

int submit(FLAGSTATS_hip_stream* s)
{
    if (s->used == 0) return 0;
    const int sl = static_cast<int>(s->submitted & 1);
    hipError_t e = hipMemcpyAsync(s->dstage[sl], s->pinned[s->cur], s->used, hipMemcpyHostToDevice, s->st[sl]);
    if (e == hipSuccess) e = hipEventRecord(s->copied[s->cur], s->st[sl]);
    if (e != hipSuccess) return fsint::fail_hip("session: hipMemcpyAsync", e);
    s->in_flight[s->cur] = true;
    int rc = fsint::count_device_async(*s->eng, s->dstage[sl], s->used / 2, s->d_out[sl], s->st[sl], s->ws[sl]);
    if (rc) return rc;
    ++s->submitted;
    s->cur = (s->cur + 1) % 3;
    s->used = 0;
    if (s->in_flight[s->cur]) {  // the chunk we are about to refill: its copy must have left the host
        e = hipEventSynchronize(s->copied[s->cur]);
        if (e != hipSuccess) return fsint::fail_hip("session: hipEventSynchronize", e);
        s->in_flight[s->cur] = false;
    }
    return 0;
}

}  // namespace

extern "C" {

FLAGSTATS_hip_stream* FLAGSTATS_hip_stream_open(void)
{
    FS_ENTRY_PTR();
    fsint::Engine* eng = fsint::default_engine();
    if (!eng) return nullptr;
    fsint::DeviceGuard guard(eng->device);
    if (!guard.ok()) return nullptr;
    FLAGSTATS_hip_stream* s = new FLAGSTATS_hip_stream();
    s->eng = eng;
    fsint::engine_retain(eng);  // the session only needs the engine's device / geometry; a FLAGSTATS_hip_shutdown while it is
                                // open releases the engine's own buffers but leaves the object to the session
    s->cap = (fsint::chunk_bytes() + 15) & ~15ull;
    if (s->cap < (1ull << 20)) s->cap = 1ull << 20;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 3 && e == hipSuccess; ++i) {
        s->pinned_reg[i] = fsint::host_alloc_registered(s->cap, eng->numa_node);
        s->pinned[i] = static_cast<uint8_t*>(s->pinned_reg[i].ptr);
        if (!s->pinned[i]) {
            FLAGSTATS_hip_stream_close(s);
            return nullptr;
        }
        e = hipEventCreateWithFlags(&s->copied[i], hipEventDisableTiming);
        if (e == hipSuccess) s->have_event[i] = true;
    }
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&s->st[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->dstage[i]), s->cap);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_out[i]), 32 * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemsetAsync(s->d_out[i], 0, 32 * sizeof(uint64_t), s->st[i]);  // ordered on ITS stream
    }
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s->h_out), 2 * 32 * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        fsint::fail_hip("FLAGSTATS_hip_stream_open", e);
        FLAGSTATS_hip_stream_close(s);
        return nullptr;
    }
    return s;
}

uint16_t* FLAGSTATS_hip_stream_acquire(FLAGSTATS_hip_stream* s, uint64_t n)
{
    FS_ENTRY_PTR();
    if (!s) return nullptr;
    std::lock_guard<std::mutex> lk(s->mu);
    const uint64_t padded = (2 * n + 15) & ~15ull;
    if (padded > s->cap) {
        fsint::fail_text("session: block larger than the chunk size (knob chunk_flags)");
        return nullptr;
    }
    if (s->used + padded > s->cap) {
        fsint::DeviceGuard guard(s->eng->device);
        if (!guard.ok() || submit(s)) return nullptr;
    }
    s->acquired = n;
    return reinterpret_cast<uint16_t*>(s->pinned[s->cur] + s->used);
}

int FLAGSTATS_hip_stream_commit(FLAGSTATS_hip_stream* s, uint64_t n)
{
    FS_ENTRY();
    if (!s) return fsint::fail_text("NULL session");
    std::lock_guard<std::mutex> lk(s->mu);
    if (n > s->acquired) return fsint::fail_text("session: commit exceeds the acquired size");
    const uint64_t padded = (2 * n + 15) & ~15ull;
    std::memset(s->pinned[s->cur] + s->used + 2 * n, 0, padded - 2 * n);
    s->used += padded;
    s->flags += n;
    s->acquired = 0;
    return 0;
}

int FLAGSTATS_hip_stream_push(FLAGSTATS_hip_stream* s, const uint16_t* array, uint64_t n)
{
    FS_ENTRY();
    if (!s) return fsint::fail_text("NULL session");
    if (n && !array) return fsint::fail_text("NULL array with n > 0");
    const uint64_t maxn = (s->cap / 2) & ~7ull;
    for (uint64_t done = 0; done < n;) {
        const uint64_t c = (n - done < maxn) ? n - done : maxn;
        uint16_t* p = FLAGSTATS_hip_stream_acquire(s, c);
        if (!p) return -1;
        std::memcpy(p, array + done, 2 * c);
        const int rc = FLAGSTATS_hip_stream_commit(s, c);
        if (rc) return rc;
        done += c;
    }
    return 0;
}

int FLAGSTATS_hip_stream_finish(FLAGSTATS_hip_stream* s, uint64_t* out)
{
    FS_ENTRY();
    if (!s || !out) return fsint::fail_text("NULL session or out");
    std::lock_guard<std::mutex> lk(s->mu);
    fsint::DeviceGuard guard(s->eng->device);
    if (!guard.ok()) return -1;
    int rc = submit(s);
    if (rc) return rc;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipMemcpyAsync(s->h_out + 32 * i, s->d_out[i], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s->st[i]);
        if (e == hipSuccess) e = hipMemsetAsync(s->d_out[i], 0, 32 * sizeof(uint64_t), s->st[i]);
        if (e != hipSuccess) return fsint::fail_hip("session: finish", e);
    }
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipStreamSynchronize(s->st[i]);
        if (e != hipSuccess) return fsint::fail_hip("session: hipStreamSynchronize", e);
    }
    for (int i = 0; i < 3; ++i) s->in_flight[i] = false;
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 32; ++k) out[k] += s->h_out[32 * i + k];
    s->flags = 0;
    return 0;
}

uint64_t FLAGSTATS_hip_stream_flags(const FLAGSTATS_hip_stream* s) { return s ? s->flags : 0; }

void FLAGSTATS_hip_stream_close(FLAGSTATS_hip_stream* s)
{
    FS_ENTRY_RELEASE();
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        fsint::DeviceGuard guard(s->eng->device);
        for (int i = 0; i < 2; ++i)
            if (s->st[i]) (void)hipStreamSynchronize(s->st[i]);
        for (int i = 0; i < 3; ++i) {
            if (s->pinned[i]) fsint::host_free_registered(s->pinned_reg[i]);
            if (s->have_event[i]) (void)hipEventDestroy(s->copied[i]);
        }
        for (int i = 0; i < 2; ++i) {
            if (s->dstage[i]) (void)hipFree(s->dstage[i]);
            if (s->d_out[i]) (void)hipFree(s->d_out[i]);
            if (s->ws[i].partials) (void)hipFree(s->ws[i].partials);
            if (s->st[i]) (void)hipStreamDestroy(s->st[i]);
        }
        if (s->h_out) (void)hipHostFree(s->h_out);
        s->h_out = nullptr;
    }
    fsint::engine_release(s->eng);
    delete s;
}

}  // extern "C"
