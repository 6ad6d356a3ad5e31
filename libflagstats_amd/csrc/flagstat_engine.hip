// flagstat_engine.hip -- per-device engine contexts: registry, device guard, error channel,
// workspaces, and the two host-side data paths every entry point is built from
// (device-resident array -> K1+K2 on a stream; host array -> double-buffered H2D + K1+K2).
// Replaces the state-free dispatch of the reference (libflagstats.h:2976-3070); see flagstat_engine.h.
#include "flagstat_engine.h"

#include <emmintrin.h>
#include <smmintrin.h>
#include <sys/syscall.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "flagstat_kernels.h"

namespace fsint {

namespace {

constexpr size_t kHostOutBytes = 1024;             // words 0..63: 2 x 32 counters; words 64..127: the small-call path's 32 {value, seq} pairs
constexpr size_t kSmallInBytes = 8ull << 20;       // input buffer of the small-call path: up to 4 Mi flags (knob small_flags picks the threshold)
constexpr uint64_t kSmallPinnedFlags = 1ull << 17; // up to here the pinned buffer (read in place) beats the BAR-written device buffer

thread_local std::string g_err;

// fork(): the reference's entry point is a pure function (libflagstats.h:3024-3070) and works in a forked child; this one is
// an engine -- a HIP context, streams, helper threads, mutexes, a polling protocol with sequence numbers -- none of which
// exists in a child.  The pid that first used the library owns it (process_guard); a child forked after that is marked by
// the atfork handler (and recognised by its pid where no handler ran: vfork, a raw clone) and every entry point refuses it
// before it touches a mutex, a stream or a HIP call.  Nothing is freed in the child: the memory is the parent's.
std::atomic<pid_t> g_owner_pid{0};
std::atomic<bool> g_forked{false};
void mark_forked_child() { g_forked.store(true, std::memory_order_relaxed); }

std::mutex g_reg_mu;                               // registry only; never held across GPU work
std::vector<Engine*> g_default;                    // index = device id; the registry holds one reference each
std::vector<Engine*> g_private;                    // engine_create()d, for shutdown_all
std::atomic<int> g_default_device{-1};
Knobs g_knobs;
std::once_flag g_env_once;

// The small-call path's hand-over: the last kernel of a call stores 32 {value, sequence number} pairs, each with ONE
// aligned 16-byte store; each is read back here with ONE aligned 16-byte load (atomic on every x86 with AVX), so a
// value is never paired with another call's sequence number and no fence or second word orders them.  Returns 0 when
// all 32 slots of call `seq` have arrived, 1 after `spins` rounds without them.
// (no_sanitize: the protocol is between the bus and the CPU, not between threads; the host-stub build's TSan run would
// otherwise see 16-byte loads racing with the stub's 8-byte atomic stores.)
__attribute__((no_sanitize("thread"))) int poll_pairs(const uint64_t* pairs64, uint64_t seq, uint64_t got[32], uint32_t spins)
{
    const __m128i* pairs = reinterpret_cast<const __m128i*>(pairs64);
    uint32_t have = 0;  // bit t: slot t of this call has arrived
    for (uint32_t spin = 0; spin < spins; ++spin) {
        for (int t = 0; t < 32; ++t) {
            if (have & (1u << t)) continue;
            const __m128i v = _mm_load_si128(pairs + t);
            if (static_cast<uint64_t>(_mm_extract_epi64(v, 1)) == seq) {
                got[t] = static_cast<uint64_t>(_mm_cvtsi128_si64(v));
                have |= 1u << t;
            }
        }
        if (have == 0xFFFFFFFFu) return 0;
        _mm_pause();
    }
    return 1;
}

uint64_t env_u64(const char* name, uint64_t dflt)
{
    const char* s = std::getenv(name);
    if (!s || !*s) return dflt;
    return std::strtoull(s, nullptr, 0);
}

void read_env_knobs()
{
    std::call_once(g_env_once, [] {
        g_knobs.blocks_per_cu = static_cast<uint32_t>(env_u64("FLAGSTATS_HIP_BLOCKS_PER_CU", g_knobs.blocks_per_cu));
        g_knobs.variant = static_cast<int>(env_u64("FLAGSTATS_HIP_VARIANT", static_cast<uint64_t>(g_knobs.variant)));
        g_knobs.chunk_flags = env_u64("FLAGSTATS_HIP_CHUNK_FLAGS", g_knobs.chunk_flags);
        g_knobs.dyn_first_pct = static_cast<uint32_t>(env_u64("FLAGSTATS_HIP_DYN_FIRST_PCT", g_knobs.dyn_first_pct));
        g_knobs.dyn_div = static_cast<uint32_t>(env_u64("FLAGSTATS_HIP_DYN_DIV", g_knobs.dyn_div));
        g_knobs.dyn_cmax = static_cast<uint32_t>(env_u64("FLAGSTATS_HIP_DYN_CMAX", g_knobs.dyn_cmax));
        g_knobs.dyn_min_steps = static_cast<uint32_t>(env_u64("FLAGSTATS_HIP_DYN_MIN_STEPS", g_knobs.dyn_min_steps));
        g_knobs.small_flags = env_u64("FLAGSTATS_HIP_SMALL_FLAGS", g_knobs.small_flags);
        g_knobs.small_bar = static_cast<int>(env_u64("FLAGSTATS_HIP_SMALL_BAR", static_cast<uint64_t>(g_knobs.small_bar)));
        g_knobs.poll = static_cast<int>(env_u64("FLAGSTATS_HIP_POLL", static_cast<uint64_t>(g_knobs.poll)));
        g_knobs.epoch_stagger = static_cast<int>(env_u64("FLAGSTATS_HIP_EPOCH_STAGGER", static_cast<uint64_t>(g_knobs.epoch_stagger)));
        fsk_set_epoch_stagger(g_knobs.epoch_stagger.load());
        g_knobs.group_min_grid = static_cast<uint32_t>(env_u64("FLAGSTATS_HIP_GROUP_MIN_GRID", g_knobs.group_min_grid));
        fsk_set_group_min_grid(g_knobs.group_min_grid.load());
        g_knobs.group_max_steps = env_u64("FLAGSTATS_HIP_GROUP_MAX_STEPS", g_knobs.group_max_steps);
        fsk_set_group_max_steps(g_knobs.group_max_steps.load());
        g_knobs.dyn_lgq = static_cast<uint32_t>(env_u64("FLAGSTATS_HIP_DYN_LG_QUEUES", g_knobs.dyn_lgq));
        if (fsk_tuning_build()) {   // (the dynamic schedule exists in the measurement build only)
            fsk_set_dyn_queues(g_knobs.dyn_lgq.load());
            fsk_set_dyn(g_knobs.dyn_first_pct.load(), g_knobs.dyn_div.load(), g_knobs.dyn_cmax.load(), g_knobs.dyn_min_steps.load());
        }
        g_knobs.fuse = fsk_tuning_build() ? static_cast<int>(env_u64("FLAGSTATS_HIP_FUSE", static_cast<uint64_t>(g_knobs.fuse))) : 0;
        g_knobs.epilogue = static_cast<int>(env_u64("FLAGSTATS_HIP_EPILOGUE", static_cast<uint64_t>(g_knobs.epilogue)));
        g_knobs.numa = static_cast<int>(env_u64("FLAGSTATS_HIP_NUMA", static_cast<uint64_t>(g_knobs.numa)));
        g_knobs.fence_free_events = static_cast<int>(env_u64("FLAGSTATS_HIP_FENCE_FREE_EVENTS", static_cast<uint64_t>(g_knobs.fence_free_events)));
        g_knobs.lz4_decoder = static_cast<int>(env_u64("FLAGSTATS_HIP_LZ4_DECODER", static_cast<uint64_t>(g_knobs.lz4_decoder)));
        g_knobs.lz4_gpu_kernel = static_cast<int>(env_u64("FLAGSTATS_HIP_LZ4_GPU_KERNEL", static_cast<uint64_t>(g_knobs.lz4_gpu_kernel)));
        g_knobs.lz4_gpu_min_bytes = env_u64("FLAGSTATS_HIP_LZ4_GPU_MIN_BYTES", g_knobs.lz4_gpu_min_bytes);
        g_knobs.staged_min_flags = env_u64("FLAGSTATS_HIP_STAGED_MIN_FLAGS", g_knobs.staged_min_flags);
        g_knobs.zstd_decoder = static_cast<int>(env_u64("FLAGSTATS_HIP_ZSTD_DECODER", static_cast<uint64_t>(g_knobs.zstd_decoder)));
        if (g_knobs.zstd_decoder > 2) g_knobs.zstd_decoder = 2;
        g_knobs.zstd_gpu_min_bytes = env_u64("FLAGSTATS_HIP_ZSTD_GPU_MIN_BYTES", g_knobs.zstd_gpu_min_bytes);
        g_knobs.lz4_gpu_keep_bytes = env_u64("FLAGSTATS_HIP_LZ4_GPU_KEEP_BYTES", g_knobs.lz4_gpu_keep_bytes);
        const char* oe = std::getenv("FLAGSTATS_HIP_ON_ERROR");
        if (oe && *oe) g_knobs.on_error = (!std::strcmp(oe, "return") || !std::strcmp(oe, "0")) ? 0 : 1;
    });
}

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return fail_hip(#expr, e_);  \
    } while (0)

// host NUMA node of a device, from sysfs via its PCI bus id; -1 if unknown
int numa_node_of_device(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return -1;
    for (char* p = bus; *p; ++p)
        if (*p >= 'A' && *p <= 'F') *p = static_cast<char>(*p - 'A' + 'a');
    char path[160];
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = std::fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

void join_second(Engine& e)
{
    e.second_hurry.store(true, std::memory_order_release);
    std::lock_guard<std::mutex> lk(e.second_mu);
    if (e.second_maker.joinable()) e.second_maker.join();
}

// A one-shot process may return from main while an engine's helper thread is still inside the runtime (making the second
// stream): the runtime's own exit handlers must not run under it.  Registered after the runtime's first call, so it runs
// before the runtime's handlers (atexit is last-in, first-out).
void join_helpers_at_exit();
std::once_flag g_atexit_once;

void release_engine_resources(Engine& e)
{
    join_second(e);
    e.pool.reset();
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) {
        if (e.ws[i].partials) (void)hipFree(e.ws[i].partials);
        if (e.d_out[i]) (void)hipFree(e.d_out[i]);
        if (e.stage[i]) (void)hipFree(e.stage[i]);
        if (e.stream[i]) (void)hipStreamDestroy(e.stream[i]);
    }
    lz4_gpu_release(e, true);
    for (auto& kv : e.user_ws)
        if (kv.second.partials) (void)hipFree(kv.second.partials);
    e.user_ws.clear();
    for (hipEvent_t& ev : e.order_ev) {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
    }
    if (e.h_out) (void)hipHostFree(e.h_out);
    if (e.small_in) (void)hipHostFree(e.small_in);
    if (e.small_bar_in) (void)hipFree(e.small_bar_in);
    for (int i = 0; i < 2; ++i)
        if (e.chunk_done[i]) (void)hipEventDestroy(e.chunk_done[i]);
    for (int i = 0; i < 3; ++i)
        if (e.pinned[i]) host_free_registered(e.pinned_reg[i]);
}

// build an engine on `device`; the caller holds no engine lock (the engine is not published yet)
int engine_setup(Engine& e, int device)
{
    // env FLAGSTATS_HIP_INIT_TIMES=1: where the creation of an engine spends its time (tests/perf/cold_start.py: what a one-shot
    // process pays before its first call -- the first HIP call of a process initialises the runtime)
    const char* tk = std::getenv("FLAGSTATS_HIP_INIT_TIMES");
    const bool timed = tk && std::atoi(tk) != 0;
    using clk = std::chrono::steady_clock;
    clk::time_point t_last = clk::now();
    double t_phase[6] = {0, 0, 0, 0, 0, 0};
    auto lap = [&](int k) {
        if (!timed) return;
        const clk::time_point now = clk::now();
        t_phase[k] += std::chrono::duration<double, std::milli>(now - t_last).count();
        t_last = now;
    };
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    lap(0);
    if (err != hipSuccess || count <= 0)
        return fail_hip("hipGetDeviceCount (no usable GPU)", err == hipSuccess ? hipErrorNoDevice : err);
    if (device < 0 || device >= count) return fail_text("device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        char buf[384];
        std::snprintf(buf, sizeof buf, "device %d is %s; this library carries gfx950 (MI355X) code only", device,
                      prop.gcnArchName);
        return fail_text(buf);
    }
    lap(1);
    DeviceGuard guard(device);
    if (!guard.ok()) return -1;
    lap(2);
    e.device = device;
    e.cus = prop.multiProcessorCount;
    e.numa_node = numa_node_of_device(device);
    {
        // A stream costs 7-20 ms to make (a hardware queue; profiles/r05/file_h2d.log, cold_start_final_tree.log) and streams are made
        // one after the other whatever thread asks.  The second stream, its counters and -- ONCE per process -- the runtime's
        // first-use warm-up are made by a helper thread that this function does not wait for (engine_second joins it before
        // anything touches stream[1] / d_out[1]): a process that only makes small FLAGSTATS_u16 calls, which poll their result out
        // of pinned memory on stream[0] and use neither a fill nor a copy, gets its counters ~17 ms earlier.
        // The warm-up: a fill and a small copy from pageable memory (the runtime loads its own fill / copy kernels at their first
        // use, ~12 ms) and copies both ways out of / into page-locked memory (the DMA engines' queues: the first such copy of a
        // process takes 13-16 ms; 256 KiB: small ones go through a copy kernel, not through the DMA engines).  Side engines,
        // explicit contexts and the engines of further devices gain nothing from a second warm-up and skip it.
        static std::atomic<bool> g_warmed{false};
        const bool warm = !g_warmed.exchange(true);
        Engine* ep = &e;
        auto make_second = [ep, device, warm] {
            Engine& e = *ep;
            // A few milliseconds of head start for the caller: making a stream holds a lock of the runtime for 8-20 ms, and a
            // first small call that arrives meanwhile waits for it with its launch (first call 8.7 ms instead of 1.0:
            // profiles/r06/cold_start.log).  Whoever needs the second stream cuts the wait short (engine_second).
            // (polled in half-millisecond naps: a timed condition-variable wait is pthread_cond_clockwait, which this image's
            // ThreadSanitizer does not know and reports as a mutex held twice)
            for (int nap = 0; nap < 8 && !e.second_hurry.load(std::memory_order_acquire); ++nap) usleep(500);
            hipError_t err1 = hipSetDevice(device);
            if (err1 == hipSuccess) err1 = hipStreamCreateWithFlags(&e.stream[1], hipStreamNonBlocking);
            if (err1 == hipSuccess) err1 = hipMalloc(&e.d_out[1], 4096);
            if (err1 == hipSuccess && warm) {
                constexpr size_t kWarm = 256u << 10;
                void *pinned_warm = nullptr, *device_warm = nullptr;
                uint64_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                err1 = hipMemsetAsync(e.d_out[1], 0, 4096, e.stream[1]);
                if (err1 == hipSuccess) err1 = hipMemcpyAsync(e.d_out[1], zeros, sizeof zeros, hipMemcpyHostToDevice, e.stream[1]);
                if (err1 == hipSuccess) err1 = hipStreamSynchronize(e.stream[1]);   // (zeros[] is this thread's stack)
                if (err1 == hipSuccess) err1 = hipHostMalloc(&pinned_warm, kWarm, hipHostMallocDefault);
                if (err1 == hipSuccess) err1 = hipMalloc(&device_warm, kWarm);
                if (err1 == hipSuccess) err1 = hipMemcpyAsync(device_warm, pinned_warm, kWarm, hipMemcpyHostToDevice, e.stream[1]);
                if (err1 == hipSuccess) err1 = hipMemcpyAsync(pinned_warm, device_warm, kWarm, hipMemcpyDeviceToHost, e.stream[1]);
                if (err1 == hipSuccess) err1 = hipStreamSynchronize(e.stream[1]);
                if (pinned_warm) (void)hipHostFree(pinned_warm);
                if (device_warm) (void)hipFree(device_warm);
            }
            e.second_err = err1;
        };
        const clk::time_point t_s0 = clk::now();
        hipError_t err0 = hipStreamCreateWithFlags(&e.stream[0], hipStreamNonBlocking);
        const clk::time_point t_s1 = clk::now();
        // (after the first stream: the helper's stream would otherwise be made first and this thread would wait for it)
        std::call_once(g_atexit_once, [] { (void)std::atexit(join_helpers_at_exit); });
        const char* lz = std::getenv("FLAGSTATS_HIP_EAGER_SECOND");   // (A/B: 1 = wait for the helper here, as r05 did)
        bool inline_second = false;
        try {
            e.second_maker = std::thread(make_second);
        } catch (const std::system_error&) {
            inline_second = true;   // (a pids / RLIMIT_NPROC limit: the work is done here instead)
        }
        if (err0 == hipSuccess) err0 = hipMalloc(&e.d_out[0], 4096);  // uint64[32] (+ room for the tuning build's 8-copy epilogue experiment)
        for (int i = 0; i < 2 && err0 == hipSuccess; ++i) err0 = hipEventCreateWithFlags(&e.chunk_done[i], hipEventDisableTiming);
        // (d_out[0] needs no zeroing here: every path that accumulates into it zeroes it first, stream-ordered)
        const clk::time_point t_s2 = clk::now();
        if (inline_second) make_second();
        if (lz && std::atoi(lz) != 0) join_second(e);
        if (timed)
            std::fprintf(stderr, "engine creation, streams (ms): first stream %.2f | counters + events %.2f | second stream%s: %s %.2f\n",
                         std::chrono::duration<double, std::milli>(t_s1 - t_s0).count(), std::chrono::duration<double, std::milli>(t_s2 - t_s1).count(),
                         warm ? " + the runtime's first fill and copies" : "", (lz && std::atoi(lz) != 0) || inline_second ? "waited for" : "left to its helper thread, not waited for",
                         std::chrono::duration<double, std::milli>(clk::now() - t_s2).count());
        if (err0 != hipSuccess) return fail_hip("engine creation: stream / counters / events", err0);
    }
    lap(3);
    HIP_TRY(hipHostMalloc(&e.h_out, kHostOutBytes, hipHostMallocDefault));
    std::memset(e.h_out, 0, kHostOutBytes);
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&e.h_out_dev), e.h_out, 0));
    // Input buffers of the small-call path.  (1) Pinned host memory that K1 reads in place over PCIe: lowest latency for a
    // few KiB (n = 1,000: 12.5 us).  (2) Fine-grained DEVICE memory that the CPU writes straight through the PCIe BAR
    // (posted, write-combined: 43 GB/s measured, tests/perf/bar_write_probe.py, against ~20 GB/s for the CPU's copy into
    // pinned memory plus the kernel's PCIe read): the data crosses the bus once, pushed, and K1 reads it from HBM --
    // 512,000 flags: 40-46 us instead of 51, 1 Mi flags: 65-72 instead of 106.  Without a large BAR, (1) serves both.
    int large_bar = 0;
    if (g_knobs.small_bar.load() && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) == hipSuccess && large_bar) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, kSmallInBytes, hipDeviceMallocFinegrained) == hipSuccess) e.small_bar_in = static_cast<uint16_t*>(p);
    }
    (void)hipGetLastError();
    lap(4);
    e.small_pinned_bytes = e.small_bar_in ? kSmallPinnedFlags * sizeof(uint16_t) : kSmallInBytes;
    HIP_TRY(hipHostMalloc(&e.small_in, e.small_pinned_bytes, hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&e.small_in_dev), e.small_in, 0));
    lap(5);
    if (timed)
        std::fprintf(stderr, "engine creation (ms): hipGetDeviceCount = runtime initialisation %.2f | device properties %.2f | hipSetDevice %.2f | two streams, counters, "
                             "events, NUMA node %.2f | pinned result buffer + fine-grained BAR input %.2f | pinned small-call input %.2f\n",
                     t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
    return 0;
}

}  // namespace

namespace {
void join_helpers_at_exit()
{
    if (g_forked.load(std::memory_order_relaxed)) return;
    std::vector<Engine*> all;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (Engine* e : g_default)
            if (e) all.push_back(e);
        for (Engine* e : g_private) all.push_back(e);
    }
    for (Engine* e : all) join_second(*e);
}
}  // namespace

int engine_second(Engine& e)
{
    join_second(e);
    if (e.second_err != hipSuccess) return fail_hip("engine creation: second stream (made by the helper thread)", e.second_err);
    if (!e.stream[1] || !e.d_out[1]) return fail_text("engine creation: the second stream was never made");
    return 0;
}

WorkerPool& engine_pool(Engine& e)
{
    if (!e.pool) e.pool.reset(new WorkerPool());
    return *e.pool;
}

WorkerPool::~WorkerPool()
{
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_job_.notify_all();
    for (std::thread& t : threads_) t.join();
}

void WorkerPool::loop(int id, uint64_t born)
{
    // `born` = the generation current when start() made this thread: the job of the NEXT generation is its first, however late
    // the thread gets here (reading generation_ now instead would skip that job if start() has bumped it already)
    uint64_t seen = born;
    std::unique_lock<std::mutex> ul(m_);
    for (;;) {
        cv_job_.wait(ul, [&] { return stop_ || (generation_ != seen && id < want_); });
        if (stop_) return;
        seen = generation_;
        ul.unlock();
        job_(id);
        ul.lock();
        if (--running_ == 0) cv_done_.notify_all();
    }
}

bool WorkerPool::start(int n, std::function<void(int)> fn)
{
    if (n < 1) return true;
    uint64_t born = 0;
    {
        std::unique_lock<std::mutex> ul(m_);
        cv_done_.wait(ul, [&] { return running_ == 0; });   // (one job at a time; the owner normally called wait() already)
        born = generation_;
    }
    try {
        while (static_cast<int>(threads_.size()) < n) {
            const int id = static_cast<int>(threads_.size());
            threads_.emplace_back([this, id, born] { loop(id, born); });
        }
    } catch (const std::system_error& ex) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "cannot start worker threads (%d of %d made): %s", static_cast<int>(threads_.size()), n, ex.what());
        fail_text(buf);
        return false;
    }
    {
        std::lock_guard<std::mutex> lk(m_);
        job_ = std::move(fn);
        want_ = n;
        running_ = n;
        ++generation_;
    }
    cv_job_.notify_all();
    return true;
}

void WorkerPool::wait()
{
    {
        std::unique_lock<std::mutex> ul(m_);
        cv_done_.wait(ul, [&] { return running_ == 0; });
        want_ = 0;
    }
    // env FLAGSTATS_HIP_POOL=0 (A/B, tests/perf/small_file_phases.py): threads made and joined per call, as until r05
    static const bool per_call = [] {
        const char* k = std::getenv("FLAGSTATS_HIP_POOL");
        return k && std::atoi(k) == 0;
    }();
    if (per_call) {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_job_.notify_all();
        for (std::thread& t : threads_) t.join();
        threads_.clear();
        stop_ = false;
    }
}

Knobs& knobs()
{
    if (!g_forked.load(std::memory_order_relaxed)) read_env_knobs();   // (a forked child inherits what its parent read)
    return g_knobs;
}

bool process_forked()
{
    if (g_forked.load(std::memory_order_relaxed)) return true;
    const pid_t owner = g_owner_pid.load(std::memory_order_relaxed);
    if (owner != 0 && owner != getpid()) {
        g_forked.store(true, std::memory_order_relaxed);   // forked without the handler having run
        return true;
    }
    return false;
}

int process_guard(const char* entry, bool quiet)
{
    if (!g_forked.load(std::memory_order_relaxed)) {
        const pid_t me = getpid();
        pid_t owner = g_owner_pid.load(std::memory_order_acquire);
        if (owner == me) return 0;
        if (owner == 0) {
            if (g_owner_pid.compare_exchange_strong(owner, me, std::memory_order_acq_rel)) {
                (void)pthread_atfork(nullptr, nullptr, mark_forked_child);
                read_env_knobs();   // so that a child never has to (on_error is part of how it is refused)
                return 0;
            }
            if (owner == me) return 0;   // another thread of this process was first
        }
        g_forked.store(true, std::memory_order_relaxed);
    }
    if (quiet) return -1;
    char buf[640];
    std::snprintf(buf, sizeof buf,
                  "%s was called in a process (pid %d) that was fork()ed from the one that uses the GPU engine (pid %d). A HIP context, "
                  "its streams and this library's worker threads do not exist in a forked child, so nothing was counted and nothing "
                  "was touched. Start worker processes with the \"spawn\" start method (Python: multiprocessing.get_context(\"spawn\")), "
                  "or make the library's first call after the fork.",
                  entry ? entry : "libflagstats_hip", static_cast<int>(getpid()), static_cast<int>(g_owner_pid.load(std::memory_order_relaxed)));
    return fail_text(buf);
}

int fail_hip(const char* what, hipError_t e)
{
    char buf[512];
    std::snprintf(buf, sizeof buf, "libflagstats_hip: %s failed: %s (%d)", what, hipGetErrorString(e), static_cast<int>(e));
    g_err = buf;
    std::fprintf(stderr, "%s\n", buf);
    return static_cast<int>(e) ? static_cast<int>(e) : -1;
}

int fail_text(const char* msg)
{
    g_err = std::string("libflagstats_hip: ") + msg;
    std::fprintf(stderr, "%s\n", g_err.c_str());
    return -1;
}

int fail_again(const char* full_text, int rc)
{
    g_err = (full_text && *full_text) ? full_text : "libflagstats_hip: a worker thread failed";
    return rc ? rc : -1;
}

const char* last_error_text() { return g_err.c_str(); }

DeviceGuard::DeviceGuard(int device)
{
    hipError_t e = hipGetDevice(&prev_);
    if (e != hipSuccess) {
        fail_hip("hipGetDevice", e);
        return;
    }
    if (prev_ != device) {
        e = hipSetDevice(device);
        if (e != hipSuccess) {
            fail_hip("hipSetDevice", e);
            return;
        }
        switched_ = true;
    }
    ok_ = true;
}

DeviceGuard::~DeviceGuard()
{
    if (switched_) (void)hipSetDevice(prev_);
}

int default_device() { return g_default_device.load(); }

Engine* engine_for_device(int device)
{
    if (process_guard("libflagstats_hip (engine lookup)")) return nullptr;
    read_env_knobs();
    if (device < 0) {
        device = g_default_device.load();
        if (device < 0) device = static_cast<int>(env_u64("FLAGSTATS_HIP_DEVICE", 0));
    }
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (device < static_cast<int>(g_default.size()) && g_default[device]) {
            if (g_default_device.load() < 0) g_default_device = device;
            return g_default[device];
        }
    }
    // build it outside the registry lock (streams, allocations: GPU work), then publish; if another thread was faster,
    // its engine wins and this one is given back
    std::unique_ptr<Engine> e(new Engine());
    if (engine_setup(*e, device)) {
        if (e->device >= 0) {
            DeviceGuard guard(e->device);
            release_engine_resources(*e);
        }
        return nullptr;
    }
    Engine* loser = nullptr;
    Engine* winner = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (static_cast<int>(g_default.size()) <= device) g_default.resize(device + 1, nullptr);
        if (g_default[device]) {
            loser = e.release();
        } else {
            e->refs = 1;
            g_default[device] = e.release();
        }
        winner = g_default[device];
        if (g_default_device.load() < 0) g_default_device = device;
    }
    if (loser) {
        DeviceGuard guard(loser->device);
        release_engine_resources(*loser);
        delete loser;
    }
    return winner;
}

Engine* default_engine() { return engine_for_device(-1); }

int select_default_device(int device)
{
    read_env_knobs();
    if (device < 0) return default_engine() ? 0 : -1;
    // the entry points that take no device (host-pointer calls, allocators, sessions) follow the latest
    // selection; device pointers are always served by the engine of the device they live on
    Engine* e = engine_for_device(device);
    if (!e) return -1;
    g_default_device = device;
    return 0;
}

Engine* engine_create(int device)
{
    if (process_guard("libflagstats_hip (engine creation)")) return nullptr;
    read_env_knobs();
    if (device < 0) {
        device = g_default_device.load();
        if (device < 0) device = static_cast<int>(env_u64("FLAGSTATS_HIP_DEVICE", 0));
    }
    std::unique_ptr<Engine> e(new Engine());
    if (engine_setup(*e, device)) {
        if (e->device >= 0) {
            DeviceGuard guard(e->device);
            release_engine_resources(*e);
        }
        return nullptr;
    }
    e->refs = 1;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_private.push_back(e.get());
    return e.release();
}

void engine_retain(Engine* e)
{
    if (e) e->refs.fetch_add(1);
}

void engine_release(Engine* e)
{
    if (e && e->refs.fetch_sub(1) == 1) delete e;  // resources went with the shutdown / destroy that marked it dead
}

int engine_alive(const Engine& e)
{
    if (!e.dead.load()) return 0;
    return fail_text("this engine was released by FLAGSTATS_hip_shutdown (or its context was destroyed); open a new session / context");
}

namespace {
// release an engine's GPU resources exactly once and mark it dead; the object stays until its last reference goes
void retire_engine(Engine* e)
{
    bool was = false;
    if (!e->dead.compare_exchange_strong(was, true)) return;
    {
        // its side engines (count_host_shared) go with it; a caller inside one finishes first (retire takes the engine's lock)
        std::lock_guard<std::mutex> mk(e->side_mu);
        for (auto& slot : e->side) {
            Engine* side = slot.exchange(nullptr);
            if (side) engine_destroy(side);
        }
    }
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard guard(e->device);
    release_engine_resources(*e);
    e->stream[0] = e->stream[1] = nullptr;
    e->d_out[0] = e->d_out[1] = nullptr;
    e->stage[0] = e->stage[1] = nullptr;
    e->stage_flags[0] = e->stage_flags[1] = 0;
    e->ws[0] = e->ws[1] = Workspace();
    e->h_out = e->h_out_dev = nullptr;
    e->small_in = e->small_in_dev = nullptr;
    e->chunk_done[0] = e->chunk_done[1] = nullptr;
    e->pinned[0] = e->pinned[1] = e->pinned[2] = nullptr;
    e->pinned_reg[0] = e->pinned_reg[1] = e->pinned_reg[2] = RegisteredHost{};
    e->pinned_bytes = 0;
}
}  // namespace

void engine_destroy(Engine* e)
{
    if (!e) return;  // (private engines only; default engines are released by shutdown_all)
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (size_t i = 0; i < g_private.size(); ++i)
            if (g_private[i] == e) {
                g_private.erase(g_private.begin() + static_cast<long>(i));
                break;
            }
    }
    retire_engine(e);   // no-op if a shutdown got there first
    engine_release(e);  // the creator's reference
}

void shutdown_all()
{
    if (process_forked()) return;   // the engines are the parent's
    std::vector<Engine*> defaults, privates;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        defaults.swap(g_default);
        privates.swap(g_private);
        g_default_device = -1;
    }
    for (Engine* e : privates) retire_engine(e);  // their creators' handles (ctx, multi, a default engine's side engines) drop the reference
    for (Engine* e : defaults) {                  // (after the private ones: a default engine lets go of its side engines here)
        if (!e) continue;
        retire_engine(e);
        engine_release(e);  // the registry's reference; sessions still holding one keep the (dead) object
    }
}

int device_of_pointer(const void* p, const char* what, int* device, bool* plain_device_memory)
{
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof attr);
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    char buf[192];
    if (e != hipSuccess) {
        (void)hipGetLastError();
        std::snprintf(buf, sizeof buf, "%s (%p) is not a device pointer known to this HIP runtime", what, p);
        return fail_text(buf);
    }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged && attr.type != hipMemoryTypeHost) {
        std::snprintf(buf, sizeof buf, "%s (%p) is not device-accessible memory", what, p);
        return fail_text(buf);
    }
    *device = attr.device;
    if (plain_device_memory) *plain_device_memory = (attr.type == hipMemoryTypeDevice);
    return 0;
}

int check_stream_device(hipStream_t s, int device)
{
    if (s == nullptr) return 0;  // the null stream of the device made current by the guard
    hipDevice_t d = -1;
    const hipError_t e = hipStreamGetDevice(s, &d);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail_hip("hipStreamGetDevice(caller stream)", e);
    }
    if (static_cast<int>(d) != device) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "the stream belongs to device %d but the array lives on device %d", static_cast<int>(d),
                      device);
        return fail_text(buf);
    }
    return 0;
}

int stream_wait_stream(Engine& e, hipStream_t waiter, hipStream_t on)
{
    if (waiter == on) return 0;
    // Record and wait while holding the lock: the ring has 16 slots, and a slot re-recorded by another caller between
    // this caller's record and its wait would bind the wait to the wrong record (two cheap runtime calls).
    std::lock_guard<std::mutex> lk(e.user_mu);
    const bool fence_free = g_knobs.fence_free_events.load() != 0;
    if (fence_free != e.order_ev_fence_free) {  // the knob changed: events of the other flavour are dropped
        for (hipEvent_t& ev : e.order_ev) {
            if (ev) (void)hipEventDestroy(ev);
            ev = nullptr;
        }
        e.order_ev_fence_free = fence_free;
    }
    hipEvent_t& slot = e.order_ev[e.order_next++ % Engine::kOrderEvents];
    if (!slot) {
        // default: an ordinary ordering event.  Knob "fence_free_events" = 1: without the system-scope fence -- enough to
        // order two streams of ONE device, ~10 us cheaper per record on the launch stream, but only ever exercised at
        // world size 1 (profiles/r02/dist_step_overhead.log): opt-in until it has met a second GPU.
        if (!fence_free || hipEventCreateWithFlags(&slot, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
            if (fence_free) (void)hipGetLastError();
            slot = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&slot, hipEventDisableTiming));
        }
    }
    // re-recording an event that an earlier hipStreamWaitEvent still refers to is fine: a wait binds to the
    // record that was current when it was queued
    HIP_TRY(hipEventRecord(slot, on));
    HIP_TRY(hipStreamWaitEvent(waiter, slot, 0));
    return 0;
}

uint32_t grid_for(const Engine& e)
{
    const uint32_t bpc = g_knobs.blocks_per_cu.load();
    return static_cast<uint32_t>(e.cus) * (bpc ? bpc : 1);
}

int ensure_ws(Workspace& w, uint32_t grid, hipStream_t s)
{
    if (w.grid_cap >= grid) return 0;
    if (w.partials) HIP_TRY(hipFree(w.partials));  // (grid knob raised) hipFree waits for the launches that still use it
    w.partials = nullptr;
    w.grid_cap = 0;
    // a first call may arrive while its stream is being captured into a graph: an allocation is legal there only
    // in the relaxed capture mode (it is not a stream operation; the memset below becomes a node of the graph)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    const hipError_t err = hipMalloc(&w.partials, fsk_partials_bytes(grid));
    if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (err != hipSuccess) return fail_hip("hipMalloc(workspace)", err);
    // The block behind the partials (K1's schedule counter, the tuning build's ticket) must be zero before the
    // first launch.  Stream-ordered on the LAUNCHING stream: no device-wide wait, other streams keep running.
    // (r02 used hipMemset + two hipDeviceSynchronize here: the NULL-stream memset is not ordered against the
    // engines' non-blocking streams, and the waits stalled every other stream on a caller's first call.)
    // While `s` is being captured into a graph a memset on it would become a NODE of that graph: the block would be zero
    // only once the graph has run, and a plain launch on the stream before its first replay (or after an abandoned
    // capture) would meet the fresh allocation's bytes in the epilogue's tickets and copies.  Zero it now instead, on a
    // stream of its own, and wait (a first call under capture, once per stream).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    if (cap == hipStreamCaptureStatusActive) {
        hipStreamCaptureMode relaxed = hipStreamCaptureModeRelaxed;
        const bool sw = hipThreadExchangeStreamCaptureMode(&relaxed) == hipSuccess;
        hipStream_t side = nullptr;
        hipError_t e1 = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        if (e1 == hipSuccess) e1 = hipMemsetAsync(w.partials, 0, fsk_partials_bytes(grid), side);
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(side);
        if (side) (void)hipStreamDestroy(side);
        if (sw) (void)hipThreadExchangeStreamCaptureMode(&relaxed);
        if (e1 != hipSuccess) return fail_hip("workspace zeroing beside a stream capture", e1);
    } else {
        HIP_TRY(hipMemsetAsync(w.partials, 0, fsk_partials_bytes(grid), s));
    }
    w.grid_cap = grid;
    return 0;
}

int count_device_async(Engine& e, const uint16_t* d_array, uint64_t n, uint64_t* d_out, hipStream_t s, Workspace& w, int op,
                       uint64_t* signal_word, uint64_t signal_value)
{
    const int base = op & OP_BASE_MASK;
    if (n == 0) {
        if (base == OP_FLAGSTAT_STORE) HIP_TRY(hipMemsetAsync(d_out, 0, 32 * sizeof(uint64_t), s));
        return 0;
    }
    if (!d_array) return fail_text("NULL array with n > 0");
    if (reinterpret_cast<uintptr_t>(d_array) & 1u) return fail_text("array must be 2-byte aligned");
    const uint32_t grid = grid_for(e);
    int rc = ensure_ws(w, grid, s);
    if (rc) return rc;
    if (base == OP_POSPOPCNT) {
        HIP_TRY(fsk_launch_pospopcnt(d_array, n, grid, w.partials, d_out, s, (!(op & OP_HOST_OUT) && g_knobs.epilogue.load()) ? 1 : 0));
    } else {
        // accumulate into plain device memory: K1 alone, its workgroups add their totals to d_out with atomics
        const bool direct = base == OP_FLAGSTAT && !(op & OP_HOST_OUT) && g_knobs.epilogue.load() && !g_knobs.fuse.load();
        const int variant = g_knobs.variant.load() | (base == OP_FLAGSTAT_STORE ? 256 : 0) |
                            ((g_knobs.fuse.load() && !direct) ? 512 : 0) | ((op & OP_SUPERSET) ? 1024 : 0) | (direct ? 2048 : 0);
        HIP_TRY(fsk_launch(d_array, n, grid, variant, w.partials,
                           reinterpret_cast<uint32_t*>(w.partials + static_cast<size_t>(w.grid_cap) * fsk::kInternal), d_out, s,
                           signal_word, signal_value));
    }
    return 0;
}

int count_on_user_stream(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream, int op)
{
    if (!d_out) return fail_text("NULL d_out");
    int dev_out = -1, dev_in = -1;
    bool out_plain = false;
    int rc = device_of_pointer(d_out, "d_out", &dev_out, &out_plain);
    if (rc) return rc;
    if (!out_plain) op |= OP_HOST_OUT;  // pinned-host / managed counters: K2 writes them, no atomics over the bus
    if (n) {
        if (!d_array) return fail_text("NULL array with n > 0");
        rc = device_of_pointer(d_array, "d_array", &dev_in);
        if (rc) return rc;
        if (dev_in != dev_out) {
            char buf[160];
            std::snprintf(buf, sizeof buf, "d_array lives on device %d but d_out on device %d", dev_in, dev_out);
            return fail_text(buf);
        }
    }
    Engine* e = engine_for_device(dev_out);
    if (!e) return -1;
    DeviceGuard guard(e->device);
    if (!guard.ok()) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = check_stream_device(s, e->device);
    if (rc) return rc;
    // one workspace per caller stream: launches on a stream are ordered, so they may share it.
    // Bounded, most recently used first; an evicted stream's workspace is freed after a sync.
    constexpr size_t kMaxUserStreams = 64;
    std::lock_guard<std::mutex> lk(e->user_mu);
    auto it = e->user_ws.begin();
    for (; it != e->user_ws.end(); ++it)
        if (it->first == stream) break;
    if (it == e->user_ws.end()) {
        if (e->user_ws.size() >= kMaxUserStreams) {
            Workspace& old = e->user_ws.back().second;
            if (old.partials) {
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipFree(old.partials));
            }
            e->user_ws.pop_back();
        }
        e->user_ws.emplace_front(stream, Workspace());
        it = e->user_ws.begin();
    } else if (it != e->user_ws.begin()) {
        e->user_ws.splice(e->user_ws.begin(), e->user_ws, it);
        it = e->user_ws.begin();
    }
    return count_device_async(*e, d_array, n, d_out, s, it->second, op);
}

int stage_reserve(Engine& e, int slot, uint64_t flags)
{
    if (e.stage_flags[slot] >= flags) return 0;
    // geometric growth up to the streaming chunk: a per-block caller with growing block sizes must not
    // pay a (device-synchronising) hipFree + hipMalloc on every call
    uint64_t want = e.stage_flags[slot] ? e.stage_flags[slot] : (1ull << 16);
    while (want < flags) want *= 2;
    const uint64_t chunk = g_knobs.chunk_flags.load() < 8 ? 8 : g_knobs.chunk_flags.load();
    if (want > chunk && flags <= chunk) want = chunk;
    if (e.stage[slot]) {
        HIP_TRY(hipStreamSynchronize(e.stream[slot]));
        HIP_TRY(hipFree(e.stage[slot]));
    }
    e.stage[slot] = nullptr;
    e.stage_flags[slot] = 0;
    HIP_TRY(hipMalloc(&e.stage[slot], want * sizeof(uint16_t)));
    e.stage_flags[slot] = want;
    return 0;
}

uint64_t chunk_bytes() { return g_knobs.chunk_flags.load() * 2; }

void* host_alloc_on_node(size_t bytes, int numa_node)
{
    void* p = nullptr;
    bool bound = false;
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
    // MPOL_PREFERRED (1) for the calling thread while the pinned pages are created; the caller's own policy (numactl
    // --interleave, say) is read first and put back afterwards
    int old_mode = 0;
    unsigned long old_mask[16] = {0};
    bool have_old = false;
    if (numa_node >= 0 && numa_node < 64 && g_knobs.numa.load()) {
        have_old = syscall(SYS_get_mempolicy, &old_mode, old_mask, sizeof(old_mask) * 8, nullptr, 0ul) == 0;
        if (!have_old) old_mode = 0;  // cannot read it (sandboxed): bind anyway and put MPOL_DEFAULT back, as r02 did
        unsigned long mask = 1ul << numa_node;
        bound = syscall(SYS_set_mempolicy, 1, &mask, sizeof(mask) * 8 + 1) == 0;
    }
#endif
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, bound ? hipHostMallocNumaUser : hipHostMallocDefault);
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
    if (bound) (void)syscall(SYS_set_mempolicy, old_mode, old_mode ? old_mask : nullptr, old_mode ? sizeof(old_mask) * 8 : 0ul);
#endif
    if (e != hipSuccess) {
        fail_hip("hipHostMalloc", e);
        return nullptr;
    }
    return p;
}

thread_local double g_reg_times[4] = {0, 0, 0, 0};   // last host_alloc_registered: mmap + madvise, first touch, hipHostRegister, fallback (ms; tests/perf)

RegisteredHost host_alloc_registered(size_t bytes, int numa_node, bool touch_and_register)
{
    RegisteredHost r;
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    auto since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    g_reg_times[0] = g_reg_times[1] = g_reg_times[2] = g_reg_times[3] = 0;
    if (!bytes) bytes = 1;
    const char* k = std::getenv("FLAGSTATS_HIP_HOST_ALLOC");   // (A/B: "malloc" = always hipHostMalloc)
    const size_t huge = 2u << 20;
    const size_t len = (bytes + huge - 1) / huge * huge;
    void* map = (k && !std::strcmp(k, "malloc")) ? MAP_FAILED : mmap(nullptr, len + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (map != MAP_FAILED) {
        uint8_t* const base = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(map) + huge - 1) & ~static_cast<uintptr_t>(huge - 1));
#ifdef MADV_HUGEPAGE
        (void)madvise(base, len, MADV_HUGEPAGE);   // (refused where huge pages are off: 4 KiB pages then, still 5x quicker than hipHostMalloc)
#endif
        g_reg_times[0] = since(t0);
        if (!touch_and_register) {
            r.ptr = base;
            r.map = map;
            r.map_bytes = len + huge;
            r.len = len;
            r.registered = false;
            return r;
        }
        const clk::time_point t1 = clk::now();
        // first touch = placement: by threads on the node's CPUs, a slice each
        cpu_set_t cpus;
        const bool bind = g_knobs.numa.load() && node_cpuset(numa_node, &cpus);
        const int nthr = len >= (16u << 20) ? 8 : 1;
        auto touch = [&](int t) {
            if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof cpus, &cpus);
            const size_t share = (len / huge + static_cast<size_t>(nthr) - 1) / static_cast<size_t>(nthr) * huge;
            const size_t lo = share * static_cast<size_t>(t);
            if (lo < len) std::memset(base + lo, 0, lo + share < len ? share : len - lo);
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t) pool.emplace_back(touch, t);   // (also for one thread: the CALLER's affinity stays as it is)
        for (std::thread& t : pool) t.join();
        g_reg_times[1] = since(t1);
        const clk::time_point t2 = clk::now();
        const hipError_t reg = hipHostRegister(base, len, hipHostRegisterDefault);
        g_reg_times[2] = since(t2);
        if (reg == hipSuccess) {
            r.ptr = base;
            r.map = map;
            r.map_bytes = len + huge;
            r.len = len;
            r.registered = true;
            return r;
        }
        (void)hipGetLastError();
        munmap(map, len + huge);
    }
    const clk::time_point t3 = clk::now();
    r.ptr = host_alloc_on_node(bytes, numa_node);
    r.len = bytes;
    r.registered = r.ptr != nullptr;
    g_reg_times[3] = since(t3);
    return r;
}

bool host_register_late(RegisteredHost& r)
{
    if (r.registered || !r.map || r.refused) return r.registered;
    using clk = std::chrono::steady_clock;
    const clk::time_point t2 = clk::now();
    const hipError_t reg = hipHostRegister(r.ptr, r.len, hipHostRegisterDefault);
    g_reg_times[2] = std::chrono::duration<double, std::milli>(clk::now() - t2).count();
    if (reg != hipSuccess) {
        (void)hipGetLastError();
        r.refused = true;   // not asked again: every attempt walks and tries to pin the whole mapping on the calling thread
        return false;
    }
    r.registered = true;
    return true;
}

void host_free_registered(RegisteredHost& r)
{
    if (r.map) {
        if (r.registered) (void)hipHostUnregister(r.ptr);
        munmap(r.map, r.map_bytes);
    } else if (r.ptr) {
        (void)hipHostFree(r.ptr);
    }
    r = RegisteredHost{};
}

int pinned_reserve(Engine& e, uint64_t bytes, void* bufs[3])
{
    if (e.pinned_bytes < bytes) {
        for (int i = 0; i < 3; ++i) {
            if (e.pinned[i]) host_free_registered(e.pinned_reg[i]);
            e.pinned[i] = nullptr;
        }
        e.pinned_bytes = 0;
        // (registered huge pages: three 64 MiB buffers in ~3 ms where hipHostMalloc takes 40-100 -- what the first block-file or
        // raw-file call of a process used to wait for)
        for (int i = 0; i < 3; ++i) {
            e.pinned_reg[i] = host_alloc_registered(bytes, e.numa_node);
            e.pinned[i] = e.pinned_reg[i].ptr;
            if (!e.pinned[i]) return -1;
        }
        e.pinned_bytes = bytes;
    }
    for (int i = 0; i < 3; ++i) bufs[i] = e.pinned[i];
    return 0;
}

// host array -> counters: double-buffered H2D + K1/K2 per chunk on the engine's two streams
// The reference's entry points are reentrant and lock-free (libflagstats.h:2980-2997): caller threads that meet on the
// DEFAULT engine must not queue behind each other.  A call takes the default engine if it is free; if another thread is
// inside it, one of up to kSideEngines side engines of the same device (private streams, staging and result buffers, made
// on first need, released with the default engine); only when all are busy does it wait for the default engine.
static int count_host_locked(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, int op);

// true if the HIP runtime does not know `p`: ordinary (pageable) host memory -- neither page-locked by hipHostMalloc /
// hipHostRegister nor device or managed memory
static bool pageable_host(const void* p)
{
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof attr);
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return e == hipErrorInvalidValue;
    }
    return attr.type == hipMemoryTypeUnregistered;
}

int count_host_shared(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, int op)
{
    if (n == 0) return 0;
    if (!h) return fail_text("NULL array with n > 0");
    // Large arrays in pageable memory: hipMemcpyAsync out of such memory makes the runtime pin it as it goes -- 53 GB/s out of
    // transparent huge pages (numpy's arrays), 30-49 GB/s, box by box and run by run, out of the 4 KiB pages of a plain malloc (a
    // 4 GiB array: 91 ms at best, 102 in the median; 256 MiB: 7.4 at best, 9.4 in the median).  Eight worker threads copying 1 MiB
    // slices into the engine's page-locked chunks (4, 8, then 16 MiB), which go over PCIe behind them, move 51-57 GB/s whatever the
    // pages are (76 ms; 5.3 ms; profiles/r05/pageable_c.log, pageable_c_ramp.log).  Below 256 MiB the pipeline's head costs
    // huge-page arrays more than the steadier rate gives back (128 MiB: +11 %, 256 MiB: +3 %, 512 MiB: -2 %), so the rule starts
    // there (knob "staged_min_flags"); FLAGSTATS_hip_host_staged_u16 is the same thing for callers who know their pages are small
    // (from 64 MiB: -20 to -50 %).
    {
        const uint64_t staged_min = g_knobs.staged_min_flags.load();
        if (staged_min && n >= staged_min && (op & OP_BASE_MASK) == OP_FLAGSTAT && pageable_host(h)) {
            g_knobs.staged_calls.fetch_add(1, std::memory_order_relaxed);
            return count_host_staged(e, h, n, out, (op & OP_SUPERSET) != 0, 8);
        }
    }
    {
        std::unique_lock<std::mutex> lk(e.mu, std::try_to_lock);
        if (lk.owns_lock()) return count_host_locked(e, h, n, out, op);
    }
    for (int i = 0; i < Engine::kSideEngines; ++i) {
        // The pointer is read AND retained under side_mu: retire_engine (FLAGSTATS_hip_shutdown from another thread) empties the
        // slots under the same lock and destroys the side engines, whose object must outlive a caller that has picked it up
        // but not yet locked it -- the reference keeps the (then dead) object, engine_alive refuses the call.
        Engine* side = nullptr;
        {
            std::lock_guard<std::mutex> mk(e.side_mu);
            if (e.dead.load()) break;
            side = e.side[i].load(std::memory_order_relaxed);
            if (!side) {
                side = engine_create(e.device);
                if (!side) break;  // (out of memory for another set of buffers: wait for the default engine instead)
                e.side[i].store(side, std::memory_order_release);
            }
            engine_retain(side);
        }
        int rc = 0;
        bool ran = false;
        {
            std::unique_lock<std::mutex> lk(side->mu, std::try_to_lock);
            if (lk.owns_lock()) {
                rc = count_host_locked(*side, h, n, out, op);
                ran = true;
            }
        }
        engine_release(side);
        if (ran) return rc;
    }
    std::lock_guard<std::mutex> lk(e.mu);
    return count_host_locked(e, h, n, out, op);
}

int count_host(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, int op)
{
    if (n == 0) return 0;
    if (!h) return fail_text("NULL array with n > 0");
    std::lock_guard<std::mutex> lk(e.mu);
    return count_host_locked(e, h, n, out, op);
}

static int count_host_locked(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, int op)
{
    const int nout = ((op & OP_BASE_MASK) == OP_POSPOPCNT) ? 16 : 32;
    if (engine_alive(e)) return -1;
    DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    lz4_gpu_other_use(e);
    const uint64_t chunk = g_knobs.chunk_flags.load() < 8 ? 8 : g_knobs.chunk_flags.load();
    const int slots = (n > chunk) ? 2 : 1;
    int rc = 0;
    const bool small_in_place = slots == 1 && (op & OP_BASE_MASK) == OP_FLAGSTAT && n <= g_knobs.small_flags.load() &&
                                n * sizeof(uint16_t) <= kSmallInBytes;
    if (slots == 2) rc = engine_second(e);
    for (int i = 0; i < slots && !rc && !small_in_place; ++i) rc = stage_reserve(e, i, n < chunk ? n : chunk);
    if (rc) return rc;
    if (slots == 1 && (op & OP_BASE_MASK) == OP_FLAGSTAT) {
        // Latency path (what an unmodified per-block or per-read caller of the reference hits, e.g. 512,000 flags per
        // call, benchmark/flagstats.cpp:328-329).  One launch sequence whose last kernel STORES the 32 slots straight
        // into the pinned host result buffer (no counter memset, no D2H copy) and then writes this call's sequence
        // number next to them; the host polls that word instead of synchronising the stream (a synchronous stream
        // round trip is ~20 us of the r02 path's 23).  Input: up to `small_flags` flags are copied by the CPU into a
        // pinned buffer that K1 reads in place over PCIe (no copy call, no DMA set-up: a copy call alone is 10-12 us);
        // above that (measured: between 1 Mi and 2 Mi flags, profiles/r03/small_calls.log, small_calls_pinned_input.log) the runtime's
        // asynchronous H2D copy into device staging is faster than the CPU's memcpy.
        const bool in_place = n <= g_knobs.small_flags.load() && n * sizeof(uint16_t) <= kSmallInBytes;
        const bool through_bar = in_place && e.small_bar_in && n > kSmallPinnedFlags;
        const bool poll = g_knobs.poll.load() != 0;
        const uint16_t* src = e.stage[0];
        if (through_bar) {
            std::memcpy(e.small_bar_in, h, n * sizeof(uint16_t));
            _mm_sfence();  // the write-combined stores leave the CPU before the doorbell of the launch
            src = e.small_bar_in;
        } else if (in_place) {
            std::memcpy(e.small_in, h, n * sizeof(uint16_t));
            src = e.small_in_dev;
        } else {
            HIP_TRY(hipMemcpyAsync(e.stage[0], h, n * sizeof(uint16_t), hipMemcpyHostToDevice, e.stream[0]));
        }
        const uint64_t seq = ++e.small_seq;
        // polled: the last kernel writes 32 {value, seq} pairs (16-byte stores) to h_out[64..127]; else plain slots to h_out[0..31]
        rc = count_device_async(e, src, n, e.h_out_dev, e.stream[0], e.ws[0], OP_FLAGSTAT_STORE | (op & OP_SUPERSET),
                                poll ? e.h_out_dev + 64 : nullptr, seq);
        if (rc) return rc;
        if (!poll) {
            HIP_TRY(hipStreamSynchronize(e.stream[0]));
            for (int s = 0; s < 32; ++s) out[s] += e.h_out[s];
            return 0;
        }
        uint64_t got[32];
        int prc = poll_pairs(e.h_out + 64, seq, got, 200000u);
        if (prc == 1) {
            // ~2 ms of polling (a busy GPU, a debugger): let the stream wait take over -- it also reports launch failures
            HIP_TRY(hipStreamSynchronize(e.stream[0]));
            e.small_since_sync = 0;
            prc = poll_pairs(e.h_out + 64, seq, got, 100u);
            if (prc) return fail_text("the kernels finished without writing this call's result pairs");
        }
        if (++e.small_since_sync >= 256) {
            // (every 256th polled call also waits on the stream, so the runtime retires its finished commands)
            HIP_TRY(hipStreamSynchronize(e.stream[0]));
            e.small_since_sync = 0;
        }
        for (int s = 0; s < 32; ++s) out[s] += got[s];
        return 0;
    }
    for (int i = 0; i < slots; ++i) HIP_TRY(hipMemsetAsync(e.d_out[i], 0, 32 * sizeof(uint64_t), e.stream[i]));
    uint64_t done = 0;
    e.host_chunks = e.host_overlapped = 0;
    for (uint64_t k = 0; done < n; ++k) {
        const int sl = static_cast<int>(k & 1);
        const uint64_t c = (n - done < chunk) ? n - done : chunk;
        // was the previous chunk (other slot, other stream) still in flight when this one is handed over?
        // (pinned host memory: yes, the copies are asynchronous; pageable: the runtime stages them itself)
        if (k > 0 && hipEventQuery(e.chunk_done[sl ^ 1]) == hipErrorNotReady) ++e.host_overlapped;
        (void)hipGetLastError();
        // same stream per slot: the copy into stage[sl] is ordered after the kernel that last read it
        HIP_TRY(hipMemcpyAsync(e.stage[sl], h + done, c * sizeof(uint16_t), hipMemcpyHostToDevice, e.stream[sl]));
        rc = count_device_async(e, e.stage[sl], c, e.d_out[sl], e.stream[sl], e.ws[sl], op);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e.chunk_done[sl], e.stream[sl]));
        ++e.host_chunks;
        done += c;
    }
    for (int i = 0; i < slots; ++i)
        HIP_TRY(hipMemcpyAsync(e.h_out + 32 * i, e.d_out[i], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, e.stream[i]));
    for (int i = 0; i < slots; ++i) HIP_TRY(hipStreamSynchronize(e.stream[i]));
    for (int i = 0; i < slots; ++i)
        for (int s = 0; s < nout; ++s) out[s] += e.h_out[32 * i + s];
    return 0;
}

}  // namespace fsint
