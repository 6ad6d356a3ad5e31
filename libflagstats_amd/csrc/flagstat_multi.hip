// flagstat_multi.hip -- the multi-GPU step in C: every flag is independent
// and the result is a sum of 32 integers, so
//   * one process, N devices: contiguous shards of a host array go over N PCIe links through N engines
//     (one host thread per engine), or N device-resident shards are counted where they live; the N x 256
//     bytes of counters are added on the host (section 8(e): "hipMemcpy 8 x 256 B to host and add");
//   * one process per device (the launch shape of bench.py --gpus N): each rank counts its shard and the
//     ranks exchange ONE ncclAllReduce(uint64[32], ncclSum) over xGMI.
// The reference has no multi-device path.  RCCL is resolved at run time (dlopen of
// librccl.so.1, so a process that already carries RCCL -- e.g. through torch -- shares that copy, and a
// process that never calls the collective entry points needs no RCCL at all); communicators travel
// through the C ABI as void*.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/libflagstats_hip.h"
#include "flagstat_engine.h"

using fsint::DeviceGuard;
using fsint::Engine;
using fsint::fail_hip;
using fsint::fail_text;

namespace {

// ---------------------------------------------------------------- RCCL, bound at first use
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_why[256] = "";

void load_rccl()
{
    const char* env = std::getenv("FLAGSTATS_HIP_RCCL");
    const char* names[] = {env && *env ? env : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (!nm) continue;
        g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        std::snprintf(g_rccl_why, sizeof g_rccl_why, "cannot load RCCL (librccl.so.1): %s", dlerror());
        return;
    }
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(g_rccl.handle, name);
        if (!p && !g_rccl_why[0]) std::snprintf(g_rccl_why, sizeof g_rccl_why, "RCCL lacks symbol %s", name);
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(sym("ncclCommCount"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.CommCount && g_rccl.AllReduce &&
                g_rccl.GetErrorString;
}

const Rccl* rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) {
        fail_text(g_rccl_why[0] ? g_rccl_why : "RCCL is not usable");
        return nullptr;
    }
    return &g_rccl;
}

int fail_nccl(const Rccl* r, const char* what, ncclResult_t e)
{
    char buf[320];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%d)", what, r->GetErrorString(e), static_cast<int>(e));
    return fail_text(buf);
}

// ---------------------------------------------------------------- one process, several devices
// Private engines of the multi-device host entry, one per shard position (so a device may appear twice
// in `devices`: two independent engines on it -- how a 1-GPU box exercises the path).
std::mutex g_multi_mu;
std::vector<Engine*> g_multi;

Engine* multi_engine(size_t slot, int device)
{
    if (g_multi.size() <= slot) g_multi.resize(slot + 1, nullptr);
    if (g_multi[slot] && g_multi[slot]->device != device) {
        fsint::engine_destroy(g_multi[slot]);
        g_multi[slot] = nullptr;
    }
    if (!g_multi[slot]) g_multi[slot] = fsint::engine_create(device);
    return g_multi[slot];
}

}  // namespace

namespace fsint {
// called by FLAGSTATS_hip_shutdown before the engines go away
void multi_forget()
{
    std::lock_guard<std::mutex> lk(g_multi_mu);
    for (Engine* e : g_multi) fsint::engine_destroy(e);  // private engines of the multi-device entry
    g_multi.clear();
}
}  // namespace fsint

extern "C" {

void FLAGSTATS_hip_shard_range(uint64_t n, int rank, int world, uint64_t* begin, uint64_t* end)
{
    // contiguous equal ranges, remainder to the last rank
    const uint64_t per = world > 0 ? n / static_cast<uint64_t>(world) : n;
    const uint64_t b = per * static_cast<uint64_t>(rank);
    if (begin) *begin = b;
    if (end) *end = (rank == world - 1) ? n : b + per;
}

int FLAGSTATS_hip_multi_u16_x64(const uint16_t* array, uint64_t n, const int* devices, int ndev, uint64_t* out)
{
    FS_ENTRY();
    if (!out) return fail_text("NULL out");
    if (ndev < 1 || ndev > 64) return fail_text("ndev must be 1..64");
    if (n && !array) return fail_text("NULL array with n > 0");
    std::lock_guard<std::mutex> lk(g_multi_mu);
    std::vector<Engine*> eng(static_cast<size_t>(ndev), nullptr);
    for (int i = 0; i < ndev; ++i) {
        eng[i] = multi_engine(static_cast<size_t>(i), devices ? devices[i] : i);
        if (!eng[i]) return -1;
    }
    std::vector<uint64_t> part(static_cast<size_t>(ndev) * 32, 0);
    std::vector<int> rcs(static_cast<size_t>(ndev), 0);
    std::vector<std::string> errs(static_cast<size_t>(ndev));
    auto work = [&](int i) {
        uint64_t b = 0, e = 0;
        FLAGSTATS_hip_shard_range(n, i, ndev, &b, &e);
        rcs[i] = fsint::count_host(*eng[i], array + b, e - b, &part[static_cast<size_t>(i) * 32]);
        if (rcs[i]) errs[i] = fsint::last_error_text();  // the error text is thread-local: carry it over
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < ndev; ++i) pool.emplace_back(work, i);
    work(0);
    for (auto& t : pool) t.join();
    for (int i = 0; i < ndev; ++i)
        if (rcs[i]) return fsint::fail_again(errs[i].c_str(), rcs[i]);
    for (int i = 0; i < ndev; ++i)
        for (int k = 0; k < 32; ++k) out[k] += part[static_cast<size_t>(i) * 32 + k];
    return 0;
}

int FLAGSTATS_hip_multi_device_u16(const uint16_t* const* d_arrays, const uint64_t* n, int nshards, uint64_t* out)
{
    FS_ENTRY();
    if (!out) return fail_text("NULL out");
    if (nshards < 0 || (nshards && (!d_arrays || !n))) return fail_text("bad shard list");
    // every shard is counted where it lives, on its device's default engine; engines are locked in
    // device order for the whole call (shards of one device run back to back on that engine's stream)
    std::vector<Engine*> owner(static_cast<size_t>(nshards), nullptr);
    std::vector<Engine*> uniq;
    for (int i = 0; i < nshards; ++i) {
        if (n[i] == 0) continue;
        int dev = -1;
        int rc = fsint::device_of_pointer(d_arrays[i], "d_arrays[i]", &dev);
        if (rc) return rc;
        owner[i] = fsint::engine_for_device(dev);
        if (!owner[i]) return -1;
        if (std::find(uniq.begin(), uniq.end(), owner[i]) == uniq.end()) uniq.push_back(owner[i]);
    }
    std::sort(uniq.begin(), uniq.end(), [](const Engine* a, const Engine* b) { return a->device < b->device; });
    std::vector<std::unique_lock<std::mutex>> locks;
    for (Engine* e : uniq) locks.emplace_back(e->mu);
    int rc = 0;
    for (Engine* e : uniq) {
        DeviceGuard guard(e->device);
        if (!guard.ok()) return -1;
        hipError_t err = hipMemsetAsync(e->d_out[0], 0, 32 * sizeof(uint64_t), e->stream[0]);
        if (err != hipSuccess) return fail_hip("hipMemsetAsync", err);
        for (int i = 0; i < nshards && !rc; ++i)
            if (owner[i] == e) rc = fsint::count_device_async(*e, d_arrays[i], n[i], e->d_out[0], e->stream[0], e->ws[0]);
        if (rc) break;
        err = hipMemcpyAsync(e->h_out, e->d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream[0]);
        if (err != hipSuccess) return fail_hip("hipMemcpyAsync(counters)", err);
    }
    // all devices are running by now: wait for each and add its 256 bytes
    for (Engine* e : uniq) {
        DeviceGuard guard(e->device);
        if (!guard.ok()) {
            if (!rc) rc = -1;
            continue;
        }
        hipError_t err = hipStreamSynchronize(e->stream[0]);
        if (err != hipSuccess && !rc) rc = fail_hip("hipStreamSynchronize", err);
    }
    if (rc) return rc;
    for (Engine* e : uniq)
        for (int k = 0; k < 32; ++k) out[k] += e->h_out[k];
    return 0;
}

/* ---- one process per device: RCCL ---- */
int FLAGSTATS_hip_comm_unique_id(void* id128)
{
    FS_ENTRY();
    if (!id128) return fail_text("NULL id buffer");
    const Rccl* r = rccl();
    if (!r) return -1;
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return fail_nccl(r, "ncclGetUniqueId", e);
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes in the ABI this entry point documents");
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

void* FLAGSTATS_hip_comm_init_rank(const void* id128, int nranks, int rank, int device)
{
    FS_ENTRY_PTR();
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        fail_text("bad communicator arguments");
        return nullptr;
    }
    const Rccl* r = rccl();
    if (!r) return nullptr;
    Engine* e = fsint::engine_for_device(device);  // also validates the device (gfx950) and makes it the default if none yet
    if (!e) return nullptr;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return nullptr;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t err = r->CommInitRank(&comm, nranks, id, rank);
    if (err != ncclSuccess) {
        fail_nccl(r, "ncclCommInitRank", err);
        return nullptr;
    }
    return comm;
}

int FLAGSTATS_hip_comm_destroy(void* comm)
{
    FS_ENTRY();
    if (!comm) return 0;
    const Rccl* r = rccl();
    if (!r) return -1;
    const ncclResult_t e = r->CommDestroy(static_cast<ncclComm_t>(comm));
    return e == ncclSuccess ? 0 : fail_nccl(r, "ncclCommDestroy", e);
}

int FLAGSTATS_hip_comm_count(void* comm)
{
    FS_ENTRY();
    if (!comm) return fail_text("NULL communicator");
    const Rccl* r = rccl();
    if (!r) return -1;
    int n = -1;
    const ncclResult_t e = r->CommCount(static_cast<ncclComm_t>(comm), &n);
    if (e != ncclSuccess) return fail_nccl(r, "ncclCommCount", e);
    return n;
}

// Which RCCL carried the collective: the path of the shared object that holds the bound ncclAllReduce (dladdr -- torch's
// bundled copy or /opt/rocm's) and its version (ncclGetVersion).  Goes into the N > 1 bench line, so that a scaling curve says
// what it was measured with.
int FLAGSTATS_hip_comm_library(char* path, uint64_t cap, int* version)
{
    FS_ENTRY();
    const Rccl* r = rccl();
    if (!r) return -1;
    if (path && cap) {
        path[0] = 0;
        Dl_info info;
        std::memset(&info, 0, sizeof info);
        if (dladdr(reinterpret_cast<const void*>(r->AllReduce), &info) && info.dli_fname) std::snprintf(path, cap, "%s", info.dli_fname);
    }
    if (version) {
        *version = -1;
        typedef ncclResult_t (*getversion_fn)(int*);
        if (getversion_fn gv = reinterpret_cast<getversion_fn>(dlsym(r->handle, "ncclGetVersion"))) {
            int v = -1;
            if (gv(&v) == ncclSuccess) *version = v;
        }
    }
    return 0;
}

int FLAGSTATS_hip_allreduce_counters(uint64_t* d_counters, void* comm, void* stream)
{
    FS_ENTRY();
    if (!d_counters || !comm) return fail_text("NULL counters or communicator");
    const Rccl* r = rccl();
    if (!r) return -1;
    int dev = -1;
    int rc = fsint::device_of_pointer(d_counters, "d_counters", &dev);
    if (rc) return rc;
    DeviceGuard guard(dev);
    if (!guard.ok()) return -1;
    rc = fsint::check_stream_device(static_cast<hipStream_t>(stream), dev);
    if (rc) return rc;
    // the path's only exchange step: 256 bytes, in place, exact (integer sum in any order)
    const ncclResult_t e = r->AllReduce(d_counters, d_counters, 32, ncclUint64, ncclSum, static_cast<ncclComm_t>(comm),
                                        static_cast<hipStream_t>(stream));
    return e == ncclSuccess ? 0 : fail_nccl(r, "ncclAllReduce", e);
}

int FLAGSTATS_hip_stream_wait_stream(void* waiter, void* on, int device)
{
    FS_ENTRY();
    Engine* e = fsint::engine_for_device(device);
    if (!e) return -1;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return -1;
    int rc = fsint::check_stream_device(static_cast<hipStream_t>(waiter), e->device);
    if (!rc) rc = fsint::check_stream_device(static_cast<hipStream_t>(on), e->device);
    if (rc) return rc;
    return fsint::stream_wait_stream(*e, static_cast<hipStream_t>(waiter), static_cast<hipStream_t>(on));
}

int FLAGSTATS_hip_device_u16_allreduce_overlapped(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* comm, void* stream,
                                                  void* comm_stream)
{
    FS_ENTRY();
    int rc = FLAGSTATS_hip_device_u16_store(d_array, n, d_out, stream);  // K1 + K2 on the launch stream
    if (rc) return rc;
    int dev = -1;
    rc = fsint::device_of_pointer(d_out, "d_out", &dev);
    if (rc) return rc;
    Engine* e = fsint::engine_for_device(dev);
    if (!e) return -1;
    {
        DeviceGuard guard(e->device);
        if (!guard.ok()) return -1;
        rc = fsint::check_stream_device(static_cast<hipStream_t>(comm_stream), e->device);
        if (rc) return rc;
        // the collective waits for the kernels on the device; the launch stream is not held up and pays no cache flush
        rc = fsint::stream_wait_stream(*e, static_cast<hipStream_t>(comm_stream), static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    return FLAGSTATS_hip_allreduce_counters(d_out, comm, comm_stream);
}

int FLAGSTATS_hip_device_u16_allreduce(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* comm, void* stream)
{
    FS_ENTRY();
    int rc = FLAGSTATS_hip_device_u16_store(d_array, n, d_out, stream);  // K1 + K2, d_out = this shard's counters
    if (rc) return rc;
    return FLAGSTATS_hip_allreduce_counters(d_out, comm, stream);
}

}  // extern "C"
