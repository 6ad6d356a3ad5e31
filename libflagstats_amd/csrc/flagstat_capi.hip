// flagstat_capi.hip -- the C-ABI boundary (include/libflagstats_hip.h): a thin host
// shim (hipMalloc / hipMemcpyAsync / kernel launches) around the kernels in
// flagstat_kernels.hip.  Replaces the dispatch layer of the reference,
// libflagstats.h:2967-3070 (FLAGSTATS_func, FLAGSTATS_get_function,
// FLAGSTATS_u16).  No CPU compute path exists here: failures are loud.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "../../include/libflagstats_hip.h"
#include "flagstat_kernels.h"

namespace {

thread_local std::string g_err;

struct Workspace {
    uint64_t* partials = nullptr;  // [grid][19]
    uint32_t grid_cap = 0;
};

struct Ctx {
    bool ready = false;
    int device = -1;
    int cus = 0;
    uint32_t blocks_per_cu = 0;  // 0 = auto
    int variant = 25;                    // DEPTH 8, nt loads, interleaved waves, rolling re-issue (tools/tune.py)
    int fuse = 0;                        // 1: K1 finalises itself (last-arriving workgroup), no K2 launch
    uint64_t chunk_flags = 32ull << 20;  // host streaming chunk: 32 Mi flags = 64 MiB
    hipStream_t stream[2] = {nullptr, nullptr};
    Workspace ws[2];
    uint64_t* d_out[2] = {nullptr, nullptr};   // device uint64[32] per slot
    uint16_t* stage[2] = {nullptr, nullptr};   // device staging for host arrays
    uint64_t stage_flags = 0;
    uint64_t* h_out = nullptr;                 // pinned 2 x 32
    std::map<void*, Workspace> user_ws;        // workspaces for caller-owned streams
    void* pinned[3] = {nullptr, nullptr, nullptr};  // block-file chunk buffers (flagstat_blocks.hip), kept across calls
    uint64_t pinned_bytes = 0;
};

Ctx g;
std::recursive_mutex g_mu;

int fail(const char* what, hipError_t e)
{
    char buf[512];
    std::snprintf(buf, sizeof buf, "libflagstats_hip: %s failed: %s (%d)", what, hipGetErrorString(e), (int)e);
    g_err = buf;
    std::fprintf(stderr, "%s\n", buf);
    return (int)e ? (int)e : -1;
}

int fail_msg(const char* msg)
{
    g_err = std::string("libflagstats_hip: ") + msg;
    std::fprintf(stderr, "%s\n", g_err.c_str());
    return -1;
}

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return fail(#expr, e_);  \
    } while (0)

uint64_t env_u64(const char* name, uint64_t dflt)
{
    const char* s = std::getenv(name);
    if (!s || !*s) return dflt;
    return std::strtoull(s, nullptr, 0);
}

uint32_t grid_for(uint64_t n)
{
    (void)n;
    uint32_t bpc = g.blocks_per_cu ? g.blocks_per_cu : 1;
    return (uint32_t)g.cus * bpc;
}

int ensure_ws(Workspace& w, uint32_t grid)
{
    if (w.grid_cap >= grid) return 0;
    if (w.partials) HIP_TRY(hipFree(w.partials));
    w.partials = nullptr;
    w.grid_cap = 0;
    HIP_TRY(hipMalloc(&w.partials, fsk_partials_bytes(grid)));
    HIP_TRY(hipMemset(w.partials, 0, fsk_partials_bytes(grid)));  // the ticket word must start at 0
    w.grid_cap = grid;
    return 0;
}

int init_locked(int device)
{
    if (g.ready) {
        if (device >= 0 && device != g.device) return fail_msg("already initialised on another device");
        return 0;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail("hipGetDeviceCount (no usable GPU)", e == hipSuccess ? hipErrorNoDevice : e);
    if (device < 0) device = (int)env_u64("FLAGSTATS_HIP_DEVICE", 0);
    if (device >= count) return fail_msg("device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "device %d is %s; this library carries gfx950 (MI355X) code only", device,
                      prop.gcnArchName);
        return fail_msg(buf);
    }
    g.device = device;
    g.cus = prop.multiProcessorCount;
    g.blocks_per_cu = (uint32_t)env_u64("FLAGSTATS_HIP_BLOCKS_PER_CU", g.blocks_per_cu);
    g.variant = (int)env_u64("FLAGSTATS_HIP_VARIANT", (uint64_t)g.variant);
    g.chunk_flags = env_u64("FLAGSTATS_HIP_CHUNK_FLAGS", g.chunk_flags);
    g.fuse = (int)env_u64("FLAGSTATS_HIP_FUSE", (uint64_t)g.fuse);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&g.stream[i], hipStreamNonBlocking));
        HIP_TRY(hipMalloc(&g.d_out[i], 32 * sizeof(uint64_t)));
    }
    HIP_TRY(hipHostMalloc(&g.h_out, 2 * 32 * sizeof(uint64_t), hipHostMallocDefault));
    g.ready = true;
    return 0;
}

int ensure_init()
{
    return g.ready ? 0 : init_locked(-1);
}

// make sure the calling thread targets the context's device
int bind()
{
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(g.device));
    return 0;
}

enum { OP_FLAGSTAT = 0, OP_POSPOPCNT = 1, OP_FLAGSTAT_STORE = 2 };

int count_device_async(const uint16_t* d_array, uint64_t n, uint64_t* d_out, hipStream_t s, Workspace& w,
                       int op = OP_FLAGSTAT)
{
    if (n == 0) {
        if (op == OP_FLAGSTAT_STORE) HIP_TRY(hipMemsetAsync(d_out, 0, 32 * sizeof(uint64_t), s));
        return 0;
    }
    if (!d_array) return fail_msg("NULL array with n > 0");
    if (reinterpret_cast<uintptr_t>(d_array) & 1u) return fail_msg("array must be 2-byte aligned");
    const uint32_t grid = grid_for(n);
    int rc = ensure_ws(w, grid);
    if (rc) return rc;
    if (op == OP_POSPOPCNT)
        HIP_TRY(fsk_launch_pospopcnt(d_array, n, grid, w.partials, d_out, s));
    else
        HIP_TRY(fsk_launch(d_array, n, grid, g.variant | (op == OP_FLAGSTAT_STORE ? 256 : 0) | (g.fuse ? 512 : 0), w.partials,
                           reinterpret_cast<uint32_t*>(w.partials + (size_t)w.grid_cap * fsk::kInternal), d_out, s));
    return 0;
}

int ensure_stage(uint64_t flags)
{
    if (g.stage_flags >= flags) return 0;
    for (int i = 0; i < 2; ++i) {
        if (g.stage[i]) HIP_TRY(hipFree(g.stage[i]));
        g.stage[i] = nullptr;
    }
    g.stage_flags = 0;
    for (int i = 0; i < 2; ++i) HIP_TRY(hipMalloc(&g.stage[i], flags * sizeof(uint16_t)));
    g.stage_flags = flags;
    return 0;
}

// host array -> counters: double-buffered H2D + K1/K2 per chunk on two streams
int count_host(const uint16_t* h, uint64_t n, uint64_t* out, int op = OP_FLAGSTAT)
{
    const int nout = (op == OP_POSPOPCNT) ? 16 : 32;
    if (n == 0) return 0;
    if (!h) return fail_msg("NULL array with n > 0");
    const uint64_t chunk = g.chunk_flags < 8 ? 8 : g.chunk_flags;
    int rc = ensure_stage(n < chunk ? n : chunk);
    if (rc) return rc;
    const int slots = (n > chunk) ? 2 : 1;
    if (slots == 1 && op == OP_FLAGSTAT) {
        // latency path (what an unmodified per-block caller of the reference hits, e.g. 512,000 flags
        // per call, benchmark/flagstats.cpp:328-329): one copy, K1, and K2 STORING straight into the
        // pinned host result buffer -- no counter memset, no D2H copy
        uint64_t* h_out_dev = nullptr;
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_out_dev), g.h_out, 0));
        HIP_TRY(hipMemcpyAsync(g.stage[0], h, n * sizeof(uint16_t), hipMemcpyHostToDevice, g.stream[0]));
        rc = count_device_async(g.stage[0], n, h_out_dev, g.stream[0], g.ws[0], OP_FLAGSTAT_STORE);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(g.stream[0]));
        for (int s = 0; s < 32; ++s) out[s] += g.h_out[s];
        return 0;
    }
    for (int i = 0; i < slots; ++i) HIP_TRY(hipMemsetAsync(g.d_out[i], 0, 32 * sizeof(uint64_t), g.stream[i]));
    uint64_t done = 0;
    for (uint64_t k = 0; done < n; ++k) {
        const int sl = (int)(k & 1);
        const uint64_t c = (n - done < chunk) ? n - done : chunk;
        // same stream per slot: the copy into stage[sl] is ordered after the
        // kernel that last read it
        HIP_TRY(hipMemcpyAsync(g.stage[sl], h + done, c * sizeof(uint16_t), hipMemcpyHostToDevice, g.stream[sl]));
        rc = count_device_async(g.stage[sl], c, g.d_out[sl], g.stream[sl], g.ws[sl], op);
        if (rc) return rc;
        done += c;
    }
    for (int i = 0; i < slots; ++i)
        HIP_TRY(hipMemcpyAsync(g.h_out + 32 * i, g.d_out[i], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream[i]));
    for (int i = 0; i < slots; ++i) HIP_TRY(hipStreamSynchronize(g.stream[i]));
    for (int i = 0; i < slots; ++i)
        for (int s = 0; s < nout; ++s) out[s] += g.h_out[32 * i + s];
    return 0;
}

}  // namespace

// internal surface for flagstat_blocks.hip (block-file pipeline), declared in flagstat_ctx.h
namespace fsint {
std::recursive_mutex& mutex() { return g_mu; }
int bind_ctx() { return bind(); }
int fail_text(const char* msg) { return fail_msg(msg); }
int fail_hip(const char* what, hipError_t e) { return fail(what, e); }
int stage_reserve(uint64_t flags) { return ensure_stage(flags); }
uint16_t* stage_buf(int slot) { return g.stage[slot]; }
hipStream_t stream(int slot) { return g.stream[slot]; }
uint64_t* dev_out(int slot) { return g.d_out[slot]; }
uint64_t* host_out() { return g.h_out; }
int count_async(const uint16_t* d, uint64_t n, int slot) { return count_device_async(d, n, g.d_out[slot], g.stream[slot], g.ws[slot]); }
int count_async_to(const uint16_t* d, uint64_t n, uint64_t* d_out, int slot) { return count_device_async(d, n, d_out, g.stream[slot], g.ws[slot]); }
int count_host_array(const uint16_t* h, uint64_t n, uint64_t* out) { return count_host(h, n, out); }
uint64_t chunk_bytes() { return g.chunk_flags * 2; }
// three pinned host buffers of >= bytes each, allocated once and reused by later calls
int pinned_reserve(uint64_t bytes, void* bufs[3])
{
    if (g.pinned_bytes < bytes) {
        for (int i = 0; i < 3; ++i) {
            if (g.pinned[i]) (void)hipHostFree(g.pinned[i]);
            g.pinned[i] = nullptr;
        }
        g.pinned_bytes = 0;
        for (int i = 0; i < 3; ++i) {
            hipError_t e = hipHostMalloc(&g.pinned[i], bytes, hipHostMallocDefault);
            if (e != hipSuccess) return fail("hipHostMalloc(chunk buffer)", e);
        }
        g.pinned_bytes = bytes;
    }
    for (int i = 0; i < 3; ++i) bufs[i] = g.pinned[i];
    return 0;
}
}  // namespace fsint

extern "C" {

int FLAGSTATS_hip_available(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g.ready) return 1;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return 0;
    hipDeviceProp_t prop;
    const int dev = (int)env_u64("FLAGSTATS_HIP_DEVICE", 0);
    if (dev >= count || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int FLAGSTATS_hip_init(int device)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return init_locked(device);
}

void FLAGSTATS_hip_shutdown(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) {
        if (g.ws[i].partials) (void)hipFree(g.ws[i].partials);
        if (g.d_out[i]) (void)hipFree(g.d_out[i]);
        if (g.stage[i]) (void)hipFree(g.stage[i]);
        if (g.stream[i]) (void)hipStreamDestroy(g.stream[i]);
    }
    for (auto& kv : g.user_ws)
        if (kv.second.partials) (void)hipFree(kv.second.partials);
    if (g.h_out) (void)hipHostFree(g.h_out);
    for (int i = 0; i < 3; ++i)
        if (g.pinned[i]) (void)hipHostFree(g.pinned[i]);
    const uint32_t bpc = g.blocks_per_cu;
    const int variant = g.variant;
    const int fuse = g.fuse;
    const uint64_t chunk = g.chunk_flags;
    g = Ctx();
    g.blocks_per_cu = bpc;
    g.variant = variant;
    g.fuse = fuse;
    g.chunk_flags = chunk;
}

const char* FLAGSTATS_hip_last_error(void) { return g_err.c_str(); }

int FLAGSTATS_hip_device_id(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return g.ready ? g.device : -1;
}

int FLAGSTATS_hip_compute_units(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return g.ready ? g.cus : -1;
}

int FLAGSTATS_hip_set(const char* key, uint64_t value)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!key) return fail_msg("NULL key");
    if (!std::strcmp(key, "blocks_per_cu")) {
        if (value > 16) return fail_msg("blocks_per_cu must be 0 (auto) .. 16");
        g.blocks_per_cu = (uint32_t)value;
    } else if (!std::strcmp(key, "variant")) {
        if (value > 127) return fail_msg("variant must be 0..127");
        g.variant = (int)value;
    } else if (!std::strcmp(key, "fuse")) {
        if (value > 1) return fail_msg("fuse must be 0 or 1");
        g.fuse = (int)value;
    } else if (!std::strcmp(key, "chunk_flags")) {
        if (value < 8) return fail_msg("chunk_flags must be >= 8");
        g.chunk_flags = value;
    } else {
        return fail_msg("unknown key");
    }
    return 0;
}

uint64_t FLAGSTATS_hip_get(const char* key)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!key) return 0;
    if (!std::strcmp(key, "blocks_per_cu")) return g.blocks_per_cu ? g.blocks_per_cu : 1;
    if (!std::strcmp(key, "variant")) return (uint64_t)g.variant;
    if (!std::strcmp(key, "chunk_flags")) return g.chunk_flags;
    if (!std::strcmp(key, "fuse")) return (uint64_t)g.fuse;
    if (!std::strcmp(key, "grid")) return g.ready ? grid_for(0) : 0;
    return 0;
}

int FLAGSTATS_u16_x64(const uint16_t* array, uint64_t n, uint64_t* out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out) return fail_msg("NULL out");
    int rc = bind();
    if (rc) return rc;
    return count_host(array, n, out);
}

int FLAGSTAT_hip(const uint16_t* array, uint32_t len, uint32_t* flags)
{
    if (!flags) return fail_msg("NULL flags");
    uint64_t wide[32];
    std::memset(wide, 0, sizeof wide);
    const int rc = FLAGSTATS_u16_x64(array, len, wide);
    if (rc) return rc;
    for (int i = 0; i < 32; ++i) flags[i] += (uint32_t)wide[i];
    return 0;
}

uint64_t FLAGSTATS_u16(const uint16_t* array, uint32_t n_len, uint32_t* flags)
{
    // libflagstats.h:3024-3070 forwards the kernel's int as uint64_t
    return (uint64_t)(int64_t)FLAGSTAT_hip(array, n_len, flags);
}

FLAGSTATS_func FLAGSTATS_get_function(uint32_t n_len)
{
    (void)n_len;
    return &FLAGSTAT_hip;
}

int FLAGSTATS_hip_device_u16(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!d_out) return fail_msg("NULL d_out");
    int rc = bind();
    if (rc) return rc;
    // `stream` is used as given: NULL is HIP's null stream (what torch's default stream is)
    return count_device_async(d_array, n, d_out, (hipStream_t)stream, g.user_ws[stream]);
}

int FLAGSTATS_hip_device_u16_store(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!d_out) return fail_msg("NULL d_out");
    int rc = bind();
    if (rc) return rc;
    return count_device_async(d_array, n, d_out, (hipStream_t)stream, g.user_ws[stream], OP_FLAGSTAT_STORE);
}

int FLAGSTATS_hip_device_u16_sync(const uint16_t* d_array, uint64_t n, uint64_t* out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out) return fail_msg("NULL out");
    int rc = bind();
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(g.d_out[0], 0, 32 * sizeof(uint64_t), g.stream[0]));
    rc = count_device_async(d_array, n, g.d_out[0], g.stream[0], g.ws[0]);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(g.h_out, g.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, g.stream[0]));
    HIP_TRY(hipStreamSynchronize(g.stream[0]));
    for (int s = 0; s < 32; ++s) out[s] += g.h_out[s];
    return 0;
}

void* FLAGSTATS_hip_host_alloc(size_t bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (bind()) return nullptr;
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        fail("hipHostMalloc", e);
        return nullptr;
    }
    return p;
}

void FLAGSTATS_hip_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

void* FLAGSTATS_hip_device_alloc(size_t bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (bind()) return nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        fail("hipMalloc", e);
        return nullptr;
    }
    return p;
}

void FLAGSTATS_hip_device_free(void* p)
{
    if (p) (void)hipFree(p);
}

int FLAGSTATS_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = bind();
    if (rc) return rc;
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return 0;
}

int FLAGSTATS_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = bind();
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int FLAGSTATS_hip_synchronize(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = bind();
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int FLAGSTATS_hip_generate_u16(uint16_t* d_array, uint64_t n, int kind, uint64_t seed, uint32_t mask,
                               uint64_t first_index, void* stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = bind();
    if (rc) return rc;
    HIP_TRY(fsk_generate(d_array, n, kind, seed, mask, first_index, (hipStream_t)stream));
    return 0;
}

int FLAGSTATS_hip_time_device_u16(const uint16_t* d_array, uint64_t n, int warmup, int reps, float* ms_total,
                                  uint64_t* out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!ms_total || reps < 1 || warmup < 0) return fail_msg("bad timing arguments");
    int rc = bind();
    if (rc) return rc;
    hipStream_t s = g.stream[0];
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < warmup; ++i) {
        rc = count_device_async(d_array, n, g.d_out[0], s, g.ws[0]);
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(g.d_out[0], 0, 32 * sizeof(uint64_t), s));
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) {
        rc = count_device_async(d_array, n, g.d_out[0], s, g.ws[0]);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipMemcpyAsync(g.h_out, g.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(ms_total, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (out)
        for (int k = 0; k < 32; ++k) out[k] += g.h_out[k] / (uint64_t)reps;
    return 0;
}

/* ---- row f4: plain positional popcount (python/libalgebra.h:3496-3551) ---- */
int FLAGSTATS_hip_pospopcnt_u16_x64(const uint16_t* array, uint64_t n, uint64_t* out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out) return fail_msg("NULL out");
    int rc = bind();
    if (rc) return rc;
    return count_host(array, n, out, OP_POSPOPCNT);
}

int STORM_pospopcnt_u16(const uint16_t* data, size_t len, uint32_t* out)
{
    if (!out) return fail_msg("NULL out");
    std::memset(out, 0, 16 * sizeof(uint32_t));  // the reference zeroes out[] first (:3497)
    uint64_t wide[16];
    std::memset(wide, 0, sizeof wide);
    const int rc = FLAGSTATS_hip_pospopcnt_u16_x64(data, len, wide);
    if (rc) return rc;
    for (int i = 0; i < 16; ++i) out[i] = (uint32_t)wide[i];
    return 0;
}

int FLAGSTATS_hip_device_pospopcnt_u16(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!d_out) return fail_msg("NULL d_out");
    int rc = bind();
    if (rc) return rc;
    return count_device_async(d_array, n, d_out, (hipStream_t)stream, g.user_ws[stream], OP_POSPOPCNT);
}

int FLAGSTATS_hip_read_probe(const void* d_buf, uint64_t bytes, int nt, int warmup, int reps, float* ms_total)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!ms_total || reps < 1 || warmup < 0) return fail_msg("bad timing arguments");
    int rc = bind();
    if (rc) return rc;
    hipStream_t s = g.stream[0];
    const uint32_t grid = grid_for(0);
    rc = ensure_ws(g.ws[0], grid);  // reuse the partials buffer as the (never written) sink
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < warmup; ++i) HIP_TRY(fsk_read_probe(d_buf, bytes, grid, nt, (uint32_t*)g.ws[0].partials, s));
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) HIP_TRY(fsk_read_probe(d_buf, bytes, grid, nt, (uint32_t*)g.ws[0].partials, s));
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(ms_total, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

extern "C" hipError_t fsk_read_probe2(const void* d_buf, uint64_t bytes, int mode, int unroll, uint32_t threads,
                                      uint32_t grid, int nt, uint32_t* d_sink, hipStream_t stream);

int FLAGSTATS_hip_read_probe2(const void* d_buf, uint64_t bytes, int mode, int unroll, uint32_t threads, uint32_t grid,
                              int nt, int warmup, int reps, float* ms_total)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!ms_total || reps < 1 || warmup < 0) return fail_msg("bad timing arguments");
    int rc = bind();
    if (rc) return rc;
    hipStream_t s = g.stream[0];
    rc = ensure_ws(g.ws[0], grid_for(0));
    if (rc) return rc;
    uint32_t* sink = (uint32_t*)g.ws[0].partials;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < warmup; ++i) HIP_TRY(fsk_read_probe2(d_buf, bytes, mode, unroll, threads, grid, nt, sink, s));
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) HIP_TRY(fsk_read_probe2(d_buf, bytes, mode, unroll, threads, grid, nt, sink, s));
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(ms_total, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

}  // extern "C"
