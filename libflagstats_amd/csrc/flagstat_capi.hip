// flagstat_capi.hip -- the C-ABI boundary (include/libflagstats_hip.h): a thin host shim
// (hipMalloc / hipMemcpyAsync / kernel launches) around the kernels in flagstat_kernels.hip.
// Replaces the dispatch layer of the reference, libflagstats.h:2967-3070 (FLAGSTATS_func,
// FLAGSTATS_get_function, FLAGSTATS_u16).  No CPU compute path exists here: failures are loud.
// State lives in per-device engines (flagstat_engine.h); this file only routes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/libflagstats_hip_probe.h"
#include "flagstat_engine.h"
#include "flagstat_kernels.h"

using fsint::DeviceGuard;
using fsint::Engine;
using fsint::fail_hip;
using fsint::fail_text;

struct FLAGSTATS_hip_ctx {
    Engine* engine;
};

namespace {

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return fail_hip(#expr, e_);  \
    } while (0)

// The reference's entry points have no error channel and its callers ignore the return value
// (python/libflagstats.pyx:22, benchmark/flagstats.cpp:329): a failed GPU call must not turn into
// silently-zero counters there.  Policy knob "on_error" (env FLAGSTATS_HIP_ON_ERROR=abort|return):
// 1 (default) = the message is followed by abort(); 0 = return non-zero (for callers that check).
int legacy_result(int rc, const char* entry)
{
    if (rc != 0 && fsint::knobs().on_error.load()) {
        std::fprintf(stderr,
                     "libflagstats_hip: %s cannot return an error to a caller of the reference API, and there is no CPU "
                     "fallback: aborting (FLAGSTATS_HIP_ON_ERROR=return makes it return non-zero instead)\n",
                     entry);
        std::fflush(stderr);
        std::abort();
    }
    return rc;
}

struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~EventPair()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    int create()
    {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        return 0;
    }
};

// device-resident array -> host counters through an engine's own stream; takes e.mu
int count_device_sync(Engine& e, const uint16_t* d_array, uint64_t n, uint64_t* out, int op = fsint::OP_FLAGSTAT)
{
    std::lock_guard<std::mutex> lk(e.mu);
    if (fsint::engine_alive(e)) return -1;
    DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    HIP_TRY(hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), e.stream[0]));
    int rc = fsint::count_device_async(e, d_array, n, e.d_out[0], e.stream[0], e.ws[0], op);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(e.h_out, e.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, e.stream[0]));
    HIP_TRY(hipStreamSynchronize(e.stream[0]));
    for (int s = 0; s < 32; ++s) out[s] += e.h_out[s];
    return 0;
}

Engine* engine_of_array(const uint16_t* d_array, uint64_t n)
{
    if (n == 0 || !d_array) return fsint::default_engine();
    int dev = -1;
    if (fsint::device_of_pointer(d_array, "d_array", &dev)) return nullptr;
    return fsint::engine_for_device(dev);
}

}  // namespace

extern "C" {

int FLAGSTATS_hip_available(void)
{
    if (fsint::process_guard(__func__)) return 0;
    if (fsint::default_device() >= 0) return 1;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    hipDeviceProp_t prop;
    const char* s = std::getenv("FLAGSTATS_HIP_DEVICE");
    const int dev = (s && *s) ? std::atoi(s) : 0;
    if (dev < 0 || dev >= count || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int FLAGSTATS_hip_device_count(void)
{
    if (fsint::process_guard(__func__)) return 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

int FLAGSTATS_hip_init(int device)
{
    FS_ENTRY();
    return fsint::select_default_device(device);
}

void FLAGSTATS_hip_shutdown(void)
{
    FS_ENTRY_RELEASE();
    fsint::multi_forget();
    fsint::shutdown_all();
}

const char* FLAGSTATS_hip_last_error(void) { return fsint::last_error_text(); }

int FLAGSTATS_hip_device_id(void) { return fsint::process_forked() ? -1 : fsint::default_device(); }

/* 1 in a child fork()ed after the library was first used (every other entry point refuses such a process), else 0; never
 * claims the library for the calling process and touches no GPU state */
int FLAGSTATS_hip_forked(void) { return fsint::process_forked() ? 1 : 0; }

int FLAGSTATS_hip_compute_units(void)
{
    FS_ENTRY();
    if (fsint::default_device() < 0) return -1;
    Engine* e = fsint::default_engine();
    return e ? e->cus : -1;
}

int FLAGSTATS_hip_set(const char* key, uint64_t value)
{
    FS_ENTRY();
    if (!key) return fail_text("NULL key");
    fsint::Knobs& k = fsint::knobs();
    if (!std::strcmp(key, "blocks_per_cu")) {
        if (value > 16) return fail_text("blocks_per_cu must be 0 (auto) .. 16");
        k.blocks_per_cu = static_cast<uint32_t>(value);
    } else if (!std::strcmp(key, "variant")) {
        if (value > 255) return fail_text("variant must be 0..255");
        if (!fsk_variant_supported(static_cast<int>(value)))
            return fail_text("this build carries K1 schedules 9, 25 and 71 only (the sweeps' losers need make TUNING=1)");
        k.variant = static_cast<int>(value);
    } else if (!std::strcmp(key, "fuse")) {
        if (value > 1) return fail_text("fuse must be 0 or 1");
        if (value && !fsk_tuning_build())
            return fail_text("the ticket-fused finalise lost to K1 + K2 and to the atomic epilogue; it is carried by the tuning build only (make TUNING=1)");
        k.fuse = static_cast<int>(value);
    } else if (!std::strcmp(key, "anatomy")) {
        if (!fsk_tuning_build()) return fail_text("anatomy is a tuning-build knob (make TUNING=1)");
        fsk_set_anatomy(static_cast<int>(value));
    } else if (!std::strcmp(key, "epilogue")) {
        if (value > 1) return fail_text("epilogue must be 0 (partials + K2) or 1 (atomic adds from K1)");
        k.epilogue = static_cast<int>(value);
    } else if (!std::strcmp(key, "small_flags")) {
        k.small_flags = value;
    } else if (!std::strcmp(key, "small_bar")) {
        if (value > 1) return fail_text("small_bar must be 0 or 1");
        k.small_bar = static_cast<int>(value);   // takes effect for engines created afterwards
    } else if (!std::strcmp(key, "poll")) {
        if (value > 1) return fail_text("poll must be 0 or 1");
        k.poll = static_cast<int>(value);
    } else if (!std::strcmp(key, "epoch_stagger")) {
        if (value > 1) return fail_text("epoch_stagger must be 0 or 1");
        k.epoch_stagger = static_cast<int>(value);
        fsk_set_epoch_stagger(k.epoch_stagger.load());
    } else if (!std::strcmp(key, "group_min_grid")) {
        if (value > 0xFFFFFFFFull) return fail_text("group_min_grid must fit 32 bits");
        k.group_min_grid = static_cast<uint32_t>(value);
        fsk_set_group_min_grid(k.group_min_grid.load());
    } else if (!std::strcmp(key, "group_max_steps")) {
        k.group_max_steps = value;
        fsk_set_group_max_steps(value);
    } else if (!std::strcmp(key, "dyn_lg_queues")) {
        if (value > 4) return fail_text("dyn_lg_queues must be 0..4");
        k.dyn_lgq = static_cast<uint32_t>(value);
        if (fsk_tuning_build()) fsk_set_dyn_queues(k.dyn_lgq.load());
    } else if (!std::strcmp(key, "dyn_first_pct") || !std::strcmp(key, "dyn_div") || !std::strcmp(key, "dyn_cmax") ||
               !std::strcmp(key, "dyn_min_steps")) {
        // K1's dynamic schedule (variant bit 7, flagstat_kernels.h DynSched)
        std::atomic<uint32_t>& slot = key[4] == 'f' ? k.dyn_first_pct : key[4] == 'd' ? k.dyn_div : key[4] == 'c' ? k.dyn_cmax : k.dyn_min_steps;
        if ((key[4] == 'f' && value > 100) || (key[4] == 'd' && (value < 1 || value > 64)) || (key[4] == 'c' && (value < 1 || value > 65535)) ||
            value > 0xFFFFFFFFull)
            return fail_text("dyn_first_pct 0..100, dyn_div 1..64, dyn_cmax 1..65535");
        slot = static_cast<uint32_t>(value);
        if (fsk_tuning_build()) fsk_set_dyn(k.dyn_first_pct.load(), k.dyn_div.load(), k.dyn_cmax.load(), k.dyn_min_steps.load());
    } else if (!std::strcmp(key, "chunk_flags")) {
        if (value < 8) return fail_text("chunk_flags must be >= 8");
        k.chunk_flags = value;
    } else if (!std::strcmp(key, "on_error")) {
        if (value > 1) return fail_text("on_error must be 0 (return) or 1 (abort)");
        k.on_error = static_cast<int>(value);
    } else if (!std::strcmp(key, "fence_free_events")) {
        if (value > 1) return fail_text("fence_free_events must be 0 or 1");
        k.fence_free_events = static_cast<int>(value);
    } else if (!std::strcmp(key, "lz4_decoder")) {
        if (value > 2) return fail_text("lz4_decoder must be 0 (host threads), 1 (GPU) or 2 (by file size)");
        k.lz4_decoder = static_cast<int>(value);
    } else if (!std::strcmp(key, "zstd_decoder")) {
        if (value > 2) return fail_text("zstd_decoder must be 0 (libzstd on host threads), 1 (GPU) or 2 (by file size)");
        k.zstd_decoder = static_cast<int>(value);
    } else if (!std::strcmp(key, "zstd_gpu_min_bytes")) {
        k.zstd_gpu_min_bytes = value;
    } else if (!std::strcmp(key, "lz4_gpu_kernel")) {
        if (value > 1) return fail_text("lz4_gpu_kernel must be 0 (workgroup pipeline) or 1 (one wave per block)");
        k.lz4_gpu_kernel = static_cast<int>(value);
    } else if (!std::strcmp(key, "lz4_gpu_min_bytes")) {
        k.lz4_gpu_min_bytes = value;
    } else if (!std::strcmp(key, "staged_min_flags")) {
        k.staged_min_flags = value;
    } else if (!std::strcmp(key, "lz4_gpu_keep_bytes")) {
        k.lz4_gpu_keep_bytes = value;
    } else if (!std::strcmp(key, "numa")) {
        if (value > 1) return fail_text("numa must be 0 or 1");
        k.numa = static_cast<int>(value);
    } else {
        return fail_text("unknown key");
    }
    return 0;
}

uint64_t FLAGSTATS_hip_get(const char* key)
{
    if (fsint::process_guard(__func__)) return 0;
    if (!key) return 0;
    fsint::Knobs& k = fsint::knobs();
    if (!std::strcmp(key, "blocks_per_cu")) return k.blocks_per_cu ? k.blocks_per_cu.load() : 1;
    if (!std::strcmp(key, "variant")) return static_cast<uint64_t>(k.variant.load());
    if (!std::strcmp(key, "chunk_flags")) return k.chunk_flags.load();
    if (!std::strcmp(key, "dyn_first_pct")) return k.dyn_first_pct.load();
    if (!std::strcmp(key, "dyn_div")) return k.dyn_div.load();
    if (!std::strcmp(key, "dyn_cmax")) return k.dyn_cmax.load();
    if (!std::strcmp(key, "dyn_min_steps")) return k.dyn_min_steps.load();
    if (!std::strcmp(key, "dyn_lg_queues")) return k.dyn_lgq.load();
    if (!std::strcmp(key, "group_min_grid")) return k.group_min_grid.load();
    if (!std::strcmp(key, "group_max_steps")) return k.group_max_steps.load();
    if (!std::strcmp(key, "last_k1_two_level")) return (fsk_last_mode() >> 3) & 1;
    if (!std::strcmp(key, "small_flags")) return k.small_flags.load();
    if (!std::strcmp(key, "small_bar")) return static_cast<uint64_t>(k.small_bar.load());
    if (!std::strcmp(key, "small_in_is_device")) {
        if (fsint::default_device() < 0) return 0;
        Engine* e = fsint::default_engine();
        return e && e->small_bar_in ? 1 : 0;
    }
    if (!std::strcmp(key, "poll")) return static_cast<uint64_t>(k.poll.load());
    if (!std::strcmp(key, "epoch_stagger")) return static_cast<uint64_t>(k.epoch_stagger.load());
    if (!std::strcmp(key, "fuse")) return static_cast<uint64_t>(k.fuse.load());
    if (!std::strcmp(key, "epilogue")) return static_cast<uint64_t>(k.epilogue.load());
    if (!std::strcmp(key, "tuning_build")) return static_cast<uint64_t>(fsk_tuning_build());
    if (!std::strcmp(key, "on_error")) return static_cast<uint64_t>(k.on_error.load());
    if (!std::strcmp(key, "numa")) return static_cast<uint64_t>(k.numa.load());
    if (!std::strcmp(key, "lz4_decoder")) return static_cast<uint64_t>(k.lz4_decoder.load());
    if (!std::strcmp(key, "zstd_decoder")) return static_cast<uint64_t>(k.zstd_decoder.load());
    if (!std::strcmp(key, "zstd_gpu_min_bytes")) return k.zstd_gpu_min_bytes.load();
    if (!std::strcmp(key, "lz4_gpu_kernel")) return static_cast<uint64_t>(k.lz4_gpu_kernel.load());
    if (!std::strcmp(key, "lz4_gpu_min_bytes")) return k.lz4_gpu_min_bytes.load();
    if (!std::strcmp(key, "staged_min_flags")) return k.staged_min_flags.load();
    if (!std::strcmp(key, "staged_calls")) return k.staged_calls.load();
    if (!std::strcmp(key, "lz4_gpu_keep_bytes")) return k.lz4_gpu_keep_bytes.load();
    if (!std::strcmp(key, "lz4_gpu_kept_bytes")) {
        if (fsint::default_device() < 0) return 0;
        Engine* e = fsint::default_engine();
        if (!e) return 0;
        std::lock_guard<std::mutex> lk(e->mu);
        uint64_t held = e->lz4_cap[0] + e->lz4_cap[1];
        for (uint64_t c : e->zstd_scratch_cap) held += c;
        return held;
    }
    if (!std::strcmp(key, "fence_free_events")) return static_cast<uint64_t>(k.fence_free_events.load());
    if (!std::strcmp(key, "grid")) {
        if (fsint::default_device() < 0) return 0;
        Engine* e = fsint::default_engine();
        return e ? fsint::grid_for(*e) : 0;
    }
    if (!std::strcmp(key, "host_chunks") || !std::strcmp(key, "host_overlapped")) {
        if (fsint::default_device() < 0) return 0;
        Engine* e = fsint::default_engine();
        if (!e) return 0;
        std::lock_guard<std::mutex> lk(e->mu);
        return key[5] == 'c' ? e->host_chunks : e->host_overlapped;
    }
    if (!std::strcmp(key, "numa_node")) {
        if (fsint::default_device() < 0) return static_cast<uint64_t>(-1);
        Engine* e = fsint::default_engine();
        return e ? static_cast<uint64_t>(static_cast<int64_t>(e->numa_node)) : static_cast<uint64_t>(-1);
    }
    return 0;
}

int FLAGSTATS_u16_x64(const uint16_t* array, uint64_t n, uint64_t* out)
{
    FS_ENTRY();
    if (!out) return fail_text("NULL out");
    Engine* e = fsint::default_engine();
    if (!e) return -1;
    return fsint::count_host_shared(*e, array, n, out);
}

int FLAGSTATS_u16_x64_superset(const uint16_t* array, uint64_t n, uint64_t* out)
{
    FS_ENTRY();
    if (!out) return fail_text("NULL out");
    Engine* e = fsint::default_engine();
    if (!e) return -1;
    return fsint::count_host_shared(*e, array, n, out, fsint::OP_FLAGSTAT | fsint::OP_SUPERSET);
}

static int flagstat_hip_u32(const uint16_t* array, uint32_t len, uint32_t* flags)
{
    if (!flags) return fail_text("NULL flags");
    uint64_t wide[32];
    std::memset(wide, 0, sizeof wide);
    const int rc = FLAGSTATS_u16_x64(array, len, wide);
    if (rc) return rc;
    for (int i = 0; i < 32; ++i) flags[i] += static_cast<uint32_t>(wide[i]);
    return 0;
}

int FLAGSTAT_hip(const uint16_t* array, uint32_t len, uint32_t* flags)
{
    return legacy_result(flagstat_hip_u32(array, len, flags), "FLAGSTAT_hip");
}

uint64_t FLAGSTATS_u16(const uint16_t* array, uint32_t n_len, uint32_t* flags)
{
    // libflagstats.h:3024-3070 forwards the kernel's int as uint64_t
    return static_cast<uint64_t>(static_cast<int64_t>(legacy_result(flagstat_hip_u32(array, n_len, flags), "FLAGSTATS_u16")));
}

FLAGSTATS_func FLAGSTATS_get_function(uint32_t n_len)
{
    // One kernel family, no host kernels: every length gets the GPU path.  The length-aware rule of
    // libflagstats.h:2999-3021 (small n stays on the host's own SIMD kernels) belongs to the reference's
    // dispatcher; INTEGRATION.md section B adds this library as its first branch, threshold
    // FLAGSTATS_HIP_MIN_LEN (tests/test_reference_patch.py builds and exercises that patch).
    (void)n_len;
    return &FLAGSTAT_hip;
}

/* ---- explicit contexts ---- */
FLAGSTATS_hip_ctx* FLAGSTATS_hip_ctx_create(int device)
{
    FS_ENTRY_PTR();
    Engine* e = fsint::engine_create(device);
    if (!e) return nullptr;
    FLAGSTATS_hip_ctx* c = new FLAGSTATS_hip_ctx;
    c->engine = e;
    return c;
}

void FLAGSTATS_hip_ctx_destroy(FLAGSTATS_hip_ctx* ctx)
{
    FS_ENTRY_RELEASE();
    if (!ctx) return;
    fsint::engine_destroy(ctx->engine);  // after a FLAGSTATS_hip_shutdown: only drops the (dead) object
    delete ctx;
}

int FLAGSTATS_hip_ctx_device(const FLAGSTATS_hip_ctx* ctx) { return (ctx && !fsint::process_forked() && !ctx->engine->dead.load()) ? ctx->engine->device : -1; }

int FLAGSTATS_hip_ctx_u16_x64(FLAGSTATS_hip_ctx* ctx, const uint16_t* array, uint64_t n, uint64_t* out)
{
    FS_ENTRY();
    if (!ctx || !out) return fail_text("NULL context or out");
    return fsint::count_host(*ctx->engine, array, n, out);
}

int FLAGSTATS_hip_ctx_device_u16_sync(FLAGSTATS_hip_ctx* ctx, const uint16_t* d_array, uint64_t n, uint64_t* out)
{
    FS_ENTRY();
    if (!ctx || !out) return fail_text("NULL context or out");
    if (n) {
        int dev = -1;
        int rc = fsint::device_of_pointer(d_array, "d_array", &dev);
        if (rc) return rc;
        if (dev != ctx->engine->device) return fail_text("d_array does not live on the context's device");
    }
    return count_device_sync(*ctx->engine, d_array, n, out);
}

/* ---- device-resident arrays ---- */
int FLAGSTATS_hip_device_u16(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream)
{
    FS_ENTRY();
    // `stream` is used as given: NULL is HIP's null stream (what torch's default stream is)
    return fsint::count_on_user_stream(d_array, n, d_out, stream, fsint::OP_FLAGSTAT);
}

int FLAGSTATS_hip_device_u16_store(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream)
{
    FS_ENTRY();
    return fsint::count_on_user_stream(d_array, n, d_out, stream, fsint::OP_FLAGSTAT_STORE);
}

int FLAGSTATS_hip_device_u16_sync(const uint16_t* d_array, uint64_t n, uint64_t* out)
{
    FS_ENTRY();
    if (!out) return fail_text("NULL out");
    Engine* e = engine_of_array(d_array, n);
    if (!e) return -1;
    return count_device_sync(*e, d_array, n, out);
}

int FLAGSTATS_hip_device_u16_superset(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream)
{
    FS_ENTRY();
    return fsint::count_on_user_stream(d_array, n, d_out, stream, fsint::OP_FLAGSTAT | fsint::OP_SUPERSET);
}

int FLAGSTATS_hip_device_u16_superset_sync(const uint16_t* d_array, uint64_t n, uint64_t* out)
{
    FS_ENTRY();
    if (!out) return fail_text("NULL out");
    Engine* e = engine_of_array(d_array, n);
    if (!e) return -1;
    return count_device_sync(*e, d_array, n, out, fsint::OP_FLAGSTAT | fsint::OP_SUPERSET);
}

/* ---- memory helpers ---- */
void* FLAGSTATS_hip_host_alloc(size_t bytes)
{
    FS_ENTRY_PTR();
    Engine* e = fsint::default_engine();
    if (!e) return nullptr;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return nullptr;
    return fsint::host_alloc_on_node(bytes, e->numa_node);
}

void FLAGSTATS_hip_host_free(void* p)
{
    FS_ENTRY_RELEASE();
    if (p) (void)hipHostFree(p);
}

void* FLAGSTATS_hip_device_alloc(size_t bytes)
{
    FS_ENTRY_PTR();
    Engine* e = fsint::default_engine();
    if (!e) return nullptr;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return nullptr;
    void* p = nullptr;
    hipError_t err = hipMalloc(&p, bytes ? bytes : 1);
    if (err != hipSuccess) {
        fail_hip("hipMalloc", err);
        return nullptr;
    }
    return p;
}

void* FLAGSTATS_hip_device_alloc_on(int device, size_t bytes)
{
    FS_ENTRY_PTR();
    Engine* e = fsint::engine_for_device(device);
    if (!e) return nullptr;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return nullptr;
    void* p = nullptr;
    hipError_t err = hipMalloc(&p, bytes ? bytes : 1);
    if (err != hipSuccess) {
        fail_hip("hipMalloc", err);
        return nullptr;
    }
    return p;
}

void FLAGSTATS_hip_device_free(void* p)
{
    FS_ENTRY_RELEASE();
    if (p) (void)hipFree(p);
}

int FLAGSTATS_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes)
{
    FS_ENTRY();
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return 0;
}

int FLAGSTATS_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes)
{
    FS_ENTRY();
    HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int FLAGSTATS_hip_synchronize(void)
{
    FS_ENTRY();
    Engine* e = fsint::default_engine();
    if (!e) return -1;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return -1;
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int FLAGSTATS_hip_generate_u16(uint16_t* d_array, uint64_t n, int kind, uint64_t seed, uint32_t mask,
                               uint64_t first_index, void* stream)
{
    FS_ENTRY();
    if (n == 0) return 0;
    int dev = -1;
    int rc = fsint::device_of_pointer(d_array, "d_array", &dev);
    if (rc) return rc;
    Engine* e = fsint::engine_for_device(dev);
    if (!e) return -1;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return -1;
    rc = fsint::check_stream_device(static_cast<hipStream_t>(stream), e->device);
    if (rc) return rc;
    HIP_TRY(fsk_generate(d_array, n, kind, seed, mask, first_index, static_cast<hipStream_t>(stream)));
    return 0;
}

int FLAGSTATS_hip_time_device_u16(const uint16_t* d_array, uint64_t n, int warmup, int reps, float* ms_total,
                                  uint64_t* out)
{
    FS_ENTRY();
    if (!ms_total || reps < 1 || warmup < 0) return fail_text("bad timing arguments");
    Engine* ep = engine_of_array(d_array, n);
    if (!ep) return -1;
    Engine& e = *ep;
    std::lock_guard<std::mutex> lk(e.mu);
    DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    hipStream_t s = e.stream[0];
    EventPair ev;
    int rc = ev.create();
    if (rc) return rc;
    for (int i = 0; i < warmup; ++i) {
        rc = fsint::count_device_async(e, d_array, n, e.d_out[0], s, e.ws[0]);
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), s));
    HIP_TRY(hipEventRecord(ev.e0, s));
    for (int i = 0; i < reps; ++i) {
        rc = fsint::count_device_async(e, d_array, n, e.d_out[0], s, e.ws[0]);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(ev.e1, s));
    HIP_TRY(hipMemcpyAsync(e.h_out, e.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(ms_total, ev.e0, ev.e1));
    if (out)
        for (int k = 0; k < 32; ++k) out[k] += e.h_out[k] / static_cast<uint64_t>(reps);
    return 0;
}

int FLAGSTATS_hip_time_device_u16_rotating(const uint16_t* d_array, uint64_t n, uint64_t stride_flags, uint32_t slots,
                                           int warmup, int reps, float* ms_total, uint64_t* out)
{
    FS_ENTRY();
    if (!ms_total || reps < 1 || warmup < 0 || slots < 1 || stride_flags < n || (stride_flags & 1)) return fail_text("bad timing arguments");
    Engine* ep = engine_of_array(d_array, n);
    if (!ep) return -1;
    Engine& e = *ep;
    std::lock_guard<std::mutex> lk(e.mu);
    if (fsint::engine_alive(e)) return -1;
    DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    hipStream_t s = e.stream[0];
    EventPair ev;
    int rc = ev.create();
    if (rc) return rc;
    auto slice = [&](int i) { return d_array + (static_cast<uint64_t>(i) * 7919u % slots) * stride_flags; };
    for (int i = 0; i < warmup; ++i) {
        rc = fsint::count_device_async(e, slice(reps + i), n, e.d_out[0], s, e.ws[0]);
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), s));
    HIP_TRY(hipEventRecord(ev.e0, s));
    for (int i = 0; i < reps; ++i) {
        rc = fsint::count_device_async(e, slice(i), n, e.d_out[0], s, e.ws[0]);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(ev.e1, s));
    HIP_TRY(hipMemcpyAsync(e.h_out, e.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(ms_total, ev.e0, ev.e1));
    if (out)
        for (int k = 0; k < 32; ++k) out[k] += e.h_out[k];
    return 0;
}

int FLAGSTATS_hip_sclk_under_load(const uint16_t* d_array, uint64_t n, int launches, double* sclk_mhz)
{
    FS_ENTRY();
    if (!sclk_mhz || launches < 1 || launches > 10000) return fail_text("bad arguments");
    Engine* ep = engine_of_array(d_array, n);
    if (!ep) return -1;
    Engine& e = *ep;
    std::lock_guard<std::mutex> lk(e.mu);
    if (fsint::engine_alive(e)) return -1;
    DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    // time `launches` K1 launches first (untimed warm-up included), then spin the probe for ~80 % of that while the same
    // launches run again on the other stream
    EventPair ev;
    int rc = ev.create();
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), e.stream[0]));
    HIP_TRY(hipEventRecord(ev.e0, e.stream[0]));
    for (int i = 0; i < launches; ++i) {
        rc = fsint::count_device_async(e, d_array, n, e.d_out[0], e.stream[0], e.ws[0]);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(ev.e1, e.stream[0]));
    HIP_TRY(hipStreamSynchronize(e.stream[0]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    uint64_t ticks = static_cast<uint64_t>(ms * 1e-3 * 0.8 * 1e8);  // 100 MHz reference
    if (ticks < 1000) ticks = 1000;
    if (ticks > 100000000ull) ticks = 100000000ull;
    constexpr uint32_t kProbes = 8;
    rc = fsint::engine_second(e);
    if (rc) return rc;
    uint64_t* d_probe = e.d_out[1];  // 4 KiB
    HIP_TRY(hipMemsetAsync(d_probe, 0, kProbes * 16, e.stream[1]));
    for (int i = 0; i < launches; ++i) {
        rc = fsint::count_device_async(e, d_array, n, e.d_out[0], e.stream[0], e.ws[0]);
        if (rc) return rc;
        if (i == 0) HIP_TRY(fsk_clock_probe(d_probe, kProbes, ticks, e.stream[1]));  // starts once the first launch is queued
    }
    uint64_t h[2 * kProbes];
    HIP_TRY(hipMemcpyAsync(e.h_out, d_probe, sizeof h, hipMemcpyDeviceToHost, e.stream[1]));
    HIP_TRY(hipStreamSynchronize(e.stream[1]));
    HIP_TRY(hipStreamSynchronize(e.stream[0]));
    std::memcpy(h, e.h_out, sizeof h);
    double cyc = 0, ref = 0;
    for (uint32_t b = 0; b < kProbes; ++b) {
        cyc += static_cast<double>(h[2 * b]);
        ref += static_cast<double>(h[2 * b + 1]);
    }
    if (ref <= 0) return fail_text("the clock probe did not run");
    *sclk_mhz = cyc / ref * 100.0;
    return 0;
}

/* ---- row f4: plain positional popcount (python/libalgebra.h:3496-3551) ---- */
int FLAGSTATS_hip_pospopcnt_u16_x64(const uint16_t* array, uint64_t n, uint64_t* out)
{
    FS_ENTRY();
    if (!out) return fail_text("NULL out");
    Engine* e = fsint::default_engine();
    if (!e) return -1;
    return fsint::count_host_shared(*e, array, n, out, fsint::OP_POSPOPCNT);
}

int STORM_pospopcnt_u16(const uint16_t* data, size_t len, uint32_t* out)
{
    if (!out) return legacy_result(fail_text("NULL out"), "STORM_pospopcnt_u16");
    std::memset(out, 0, 16 * sizeof(uint32_t));  // the reference zeroes out[] first (:3497)
    uint64_t wide[16];
    std::memset(wide, 0, sizeof wide);
    const int rc = FLAGSTATS_hip_pospopcnt_u16_x64(data, len, wide);
    if (rc) return legacy_result(rc, "STORM_pospopcnt_u16");
    for (int i = 0; i < 16; ++i) out[i] = static_cast<uint32_t>(wide[i]);
    return 0;
}

int FLAGSTATS_hip_device_pospopcnt_u16(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream)
{
    FS_ENTRY();
    return fsint::count_on_user_stream(d_array, n, d_out, stream, fsint::OP_POSPOPCNT);
}

/* ---- read-only bandwidth probes (measurement) ---- */
extern "C" hipError_t fsk_read_probe2(const void* d_buf, uint64_t bytes, int mode, int unroll, uint32_t threads,
                                      uint32_t grid, int nt, uint32_t* d_sink, hipStream_t stream);

static int probe_common(const void* d_buf, int warmup, int reps, float* ms_total,
                        hipError_t (*launch)(const void*, uint64_t, uint32_t, uint32_t*, hipStream_t, const int*),
                        uint64_t bytes, uint32_t grid_override, const int* params)
{
    FS_ENTRY();
    if (!ms_total || reps < 1 || warmup < 0) return fail_text("bad timing arguments");
    int dev = -1;
    int rc = fsint::device_of_pointer(d_buf, "d_buf", &dev);
    if (rc) return rc;
    Engine* ep = fsint::engine_for_device(dev);
    if (!ep) return -1;
    Engine& e = *ep;
    std::lock_guard<std::mutex> lk(e.mu);
    DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    hipStream_t s = e.stream[0];
    const uint32_t grid = grid_override ? grid_override : fsint::grid_for(e);
    rc = fsint::ensure_ws(e.ws[0], fsint::grid_for(e), s);  // reuse the partials buffer as the (never written) sink
    if (rc) return rc;
    uint32_t* sink = reinterpret_cast<uint32_t*>(e.ws[0].partials);
    EventPair ev;
    rc = ev.create();
    if (rc) return rc;
    for (int i = 0; i < warmup; ++i) HIP_TRY(launch(d_buf, bytes, grid, sink, s, params));
    HIP_TRY(hipEventRecord(ev.e0, s));
    for (int i = 0; i < reps; ++i) HIP_TRY(launch(d_buf, bytes, grid, sink, s, params));
    HIP_TRY(hipEventRecord(ev.e1, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventElapsedTime(ms_total, ev.e0, ev.e1));
    return 0;
}

int FLAGSTATS_hip_read_probe(const void* d_buf, uint64_t bytes, int nt, int warmup, int reps, float* ms_total)
{
    // the fastest read-only pattern found on this chip (profiles/r03/read_probe_sweep.log): one 384-thread workgroup per
    // CU, 4 vectors per lane per step = 24 KiB in flight per CU, grid-stride (r01-r02 used 256 x 8 = 32 KiB: 1.5-2 % less)
    const int params[4] = {0, 4, 384, nt};
    return probe_common(
        d_buf, warmup, reps, ms_total,
        [](const void* b, uint64_t n, uint32_t g, uint32_t* sink, hipStream_t s, const int* p) {
            return fsk_read_probe2(b, n, p[0], p[1], static_cast<uint32_t>(p[2]), g, p[3], sink, s);
        },
        bytes, 0, params);
}

extern "C" hipError_t fsk_read_probe_policy(const void* d_buf, uint64_t bytes, int policy, uint32_t grid, uint32_t* d_sink, hipStream_t s);

int FLAGSTATS_hip_read_probe_policy(const void* d_buf, uint64_t bytes, int policy, int warmup, int reps, float* ms_total)
{
    const int params[1] = {policy};
    return probe_common(
        d_buf, warmup, reps, ms_total,
        [](const void* b, uint64_t n, uint32_t g, uint32_t* sink, hipStream_t s, const int* p) {
            return fsk_read_probe_policy(b, n, p[0], g, sink, s);
        },
        bytes, 0, params);
}

int FLAGSTATS_hip_read_probe2(const void* d_buf, uint64_t bytes, int mode, int unroll, uint32_t threads, uint32_t grid,
                              int nt, int warmup, int reps, float* ms_total)
{
    const int params[4] = {mode, unroll, static_cast<int>(threads), nt};
    if (grid == 0) return fail_text("grid must be > 0");
    return probe_common(
        d_buf, warmup, reps, ms_total,
        [](const void* b, uint64_t n, uint32_t g, uint32_t* sink, hipStream_t s, const int* p) {
            return fsk_read_probe2(b, n, p[0], p[1], static_cast<uint32_t>(p[2]), g, p[3], sink, s);
        },
        bytes, grid, params);
}

}  // extern "C"
