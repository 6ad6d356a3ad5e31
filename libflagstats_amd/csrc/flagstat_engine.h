// flagstat_engine.h -- internal: per-device engine contexts behind the C-ABI of libflagstats_hip.so.
//
// The reference keeps no state but a cached cpuid (libflagstats.h:2980-2997) and is reentrant.
// Here the state is an *engine* per device: two streams, device staging for host arrays, K1
// workspaces, device + pinned counters.  Engines are independent of each other (own mutex), so
//   * one process can drive several GPUs (FLAGSTATS_hip_multi_*), and a caller's device pointer is
//     served by the engine of the device that pointer lives on -- never by "the" global context;
//   * streaming sessions and caller-stream launches own their streams / workspaces and take no
//     engine-wide lock on the data path.
// Every entry point restores the calling thread's current HIP device before it returns.
#ifndef FLAGSTAT_ENGINE_H_
#define FLAGSTAT_ENGINE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <list>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

struct FLAGSTATS_gpu_lz4_stats;

namespace fsint {

struct Workspace {
    uint64_t* partials = nullptr;  // [fsk::kInternal][grid] + ticket block (fsk_partials_bytes)
    uint32_t grid_cap = 0;
};

// process-wide tuning knobs (FLAGSTATS_hip_set / env FLAGSTATS_HIP_*); they survive a shutdown
struct Knobs {
    std::atomic<uint32_t> blocks_per_cu{0};       // 0 = auto (1)
    std::atomic<int> variant{71};                 // K1 schedule, see flagstat_kernels.hip (71: rolling at distance 6, r03)
    std::atomic<uint32_t> dyn_first_pct{75}, dyn_div{4}, dyn_cmax{32}, dyn_min_steps{32}, dyn_lgq{3};  // dynamic schedule (variant bit 7)
    std::atomic<uint64_t> small_flags{1ull << 20}; // host arrays up to this many flags: copied into the pinned input buffer and read
                                                  // in place (no H2D copy call); 0 = always stage through device memory
    std::atomic<int> small_bar{1};                // 1: small-call input goes into device memory through the BAR when the device has a
                                                  // large BAR (read at engine creation); 0: always pinned host memory
    std::atomic<int> poll{1};                     // single-launch host calls wait by polling a completion word the kernel writes
    std::atomic<int> epoch_stagger{1};            // K1: waves of a workgroup flush their epochs at different steps
    std::atomic<uint32_t> group_min_grid{64};     // K1's atomic epilogue goes through per-XCD copies from this many workgroups on
    std::atomic<uint64_t> group_max_steps{40};    // ... and only up to this many steps per workgroup (they finish together)
    std::atomic<int> fuse{0};                     // 1: K1 finalises itself (last-arriving workgroup), no K2 launch
    std::atomic<int> epilogue{1};                 // accumulate form into device memory: 1 = K1 adds its workgroup totals to
                                                  // out[] with atomics (one launch), 0 = partials + K2
    std::atomic<uint64_t> chunk_flags{32ull << 20};  // host streaming chunk: 32 Mi flags = 64 MiB
    std::atomic<int> on_error{1};                 // legacy uint32 entry points: 1 = abort after the message, 0 = return non-zero
    std::atomic<int> lz4_decoder{2};              // LZ4 block files: 0 = decode on host threads, 1 = on the GPU, 2 = by size
    std::atomic<int> lz4_gpu_kernel{0};           // GPU LZ4 decode kernel: 0 = workgroup pipeline (r04), 1 = one wave per block (r03)
    std::atomic<uint64_t> lz4_gpu_min_bytes{64ull << 20};  // lz4_decoder 2: GPU decode for files of at least this many bytes
    std::atomic<uint64_t> staged_min_flags{1ull << 27};    // host arrays in PAGEABLE memory of at least this many flags go through the
                                                           // engine's page-locked chunks, copied there by worker threads (0 = never)
    std::atomic<uint64_t> staged_calls{0};                 // (read-only from outside: how many calls that rule has sent there)
    std::atomic<uint64_t> lz4_gpu_keep_bytes{~0ull};  // device bytes the GPU LZ4 decoder may keep between calls; ~0 = automatic: what the
                                                  // last call needed, at most a quarter of the device, until 8 other calls have passed
    std::atomic<int> zstd_decoder{2};             // Zstandard block files: 0 = libzstd on host threads, 1 = decode on the GPU, 2 = by size
    std::atomic<uint64_t> zstd_gpu_min_bytes{64ull << 20};  // zstd_decoder 2: GPU decode for files of at least this many bytes
    std::atomic<int> fence_free_events{0};        // stream_wait_stream: 1 = ordering events without the system-scope fence (opt-in)
    std::atomic<int> numa{1};                     // block pipeline: 1 = pinned chunks + decoders on the GPU's NUMA node
};
Knobs& knobs();

// base operation | OP_SUPERSET (flagstat only: also fill slots 0/16 = primary paired reads, slot 9 = pass-QC reads)
// | OP_HOST_OUT (d_out is pinned-host or managed memory: counters are written by K2, never by device atomics)
enum { OP_FLAGSTAT = 0, OP_POSPOPCNT = 1, OP_FLAGSTAT_STORE = 2, OP_BASE_MASK = 3, OP_SUPERSET = 4, OP_HOST_OUT = 8 };

// page-locked host memory from host_alloc_registered (below)
struct RegisteredHost {
    void* ptr = nullptr;       // what the caller uses (2 MiB aligned when mapped here)
    void* map = nullptr;       // the mapping behind it (nullptr: `ptr` came from hipHostMalloc)
    size_t map_bytes = 0;
    size_t len = 0;            // bytes registered (or to be registered) from ptr
    bool registered = false;   // false after host_alloc_registered(..., false) until host_register_late
    bool refused = false;      // host_register_late was refused once: not asked again (the memory stays usable, pageable)
};

// Worker threads kept with an engine (the host-thread block pipeline's decoders, the GPU decoders' file readers): making and
// joining 8-24 threads per call cost ~0.3 ms, a quarter of a 2-17-block file's whole call (tests/perf/small_file_phases.py).
// Threads are made on first need and park on a condition variable between jobs.  One job at a time: the owner holds the
// engine's lock from start() to wait().
class WorkerPool {
  public:
    WorkerPool() = default;
    ~WorkerPool();
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;
    // fn(t) runs once for every t in [0, n) on n pool threads.  Returns false (nothing started, error recorded) when the
    // threads cannot be made.  fn must stay valid until wait() has returned.
    bool start(int n, std::function<void(int)> fn);
    void wait();   // until every fn(t) of the job has returned
    int threads() const { return static_cast<int>(threads_.size()); }

  private:
    void loop(int id, uint64_t born);
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    std::vector<std::thread> threads_;
    std::function<void(int)> job_;
    uint64_t generation_ = 0;   // bumped by start(): a parked thread with id < want_ runs the job of a generation once
    int want_ = 0, running_ = 0;
    bool stop_ = false;
};

struct Engine {
    // Handles that outlive a FLAGSTATS_hip_shutdown (sessions, explicit contexts) keep the OBJECT alive through `refs`; the
    // shutdown releases the engine's GPU resources and marks it `dead`, after which every operation on it fails loudly
    // and the last handle to let go deletes it (engine_release).
    std::atomic<int> refs{0};
    std::atomic<bool> dead{false};
    int device = -1;
    int cus = 0;
    int numa_node = -1;                            // host NUMA node closest to the device (-1 unknown)
    std::mutex mu;                                 // guards everything below up to `user_mu`
    // stream[1], d_out[1] and the runtime's first-use warm-up are made by a helper thread that engine creation does NOT wait
    // for: engine_second() (below) joins it before anything touches them.  A process that only makes small FLAGSTATS_u16
    // calls never waits (a stream costs 8-20 ms to make and streams are made one after the other whatever thread asks).
    std::mutex second_mu;                          // guards the join only
    std::thread second_maker;
    hipError_t second_err = hipSuccess;            // written by the helper, read after the join
    std::atomic<bool> second_hurry{false};         // cuts the helper's head start for a first small call short (engine_setup, engine_second)
    std::unique_ptr<WorkerPool> pool;              // made on first need (run_pipeline, file-mode readers); e.mu held
    hipStream_t stream[2] = {nullptr, nullptr};
    Workspace ws[2];
    uint64_t* d_out[2] = {nullptr, nullptr};       // device uint64[32] per slot
    uint16_t* stage[2] = {nullptr, nullptr};       // device staging for host arrays
    uint64_t stage_flags[2] = {0, 0};
    uint64_t* h_out = nullptr;                     // pinned 2 x 32 (also mapped into the device); words 64..127 = the small-call path's {value, seq} pairs
    uint64_t* h_out_dev = nullptr;                 // the same buffer as the device sees it
    uint16_t* small_in = nullptr;                  // small-call input, pinned host memory that K1 reads in place
    uint16_t* small_in_dev = nullptr;              // ... as K1 reads it
    size_t small_pinned_bytes = 0;
    uint16_t* small_bar_in = nullptr;              // small-call input above 2^17 flags: fine-grained DEVICE memory the CPU writes
                                                   // through the PCIe BAR (nullptr: no large BAR, the pinned buffer serves all sizes)
    uint64_t small_seq = 0;                        // sequence number of the last small call (what its kernel stores next to each slot)
    uint32_t small_since_sync = 0;                 // polled calls since the stream was last synchronised
    hipEvent_t chunk_done[2] = {nullptr, nullptr}; // host streaming: chunk in slot i has been counted
    uint64_t host_chunks = 0;                      // last multi-chunk host call: chunks submitted ...
    uint64_t host_overlapped = 0;                  // ... and how many were submitted while the previous one was still in flight
    void* pinned[3] = {nullptr, nullptr, nullptr}; // block-file chunk buffers, kept across calls
    RegisteredHost pinned_reg[3];                  // ... and what they are made of (host_alloc_registered)
    uint8_t* lz4_buf[2] = {nullptr, nullptr};      // GPU LZ4 decoder: compressed / decoded bytes of a segment, kept across calls
    uint64_t lz4_cap[2] = {0, 0};                  // (knob "lz4_gpu_keep_bytes"; released after kLz4IdleCalls other calls)
#ifndef FLAGSTAT_DECODE_STREAMS   // (measurement builds: more decode streams)
#define FLAGSTAT_DECODE_STREAMS 2
#endif
    static constexpr int kLz4Streams = FLAGSTAT_DECODE_STREAMS, kLz4MaxPieces = 64, kLz4IdleCalls = 8;
    hipStream_t lz4_stream[kLz4Streams] = {};      // ... its decode streams, events and small device buffers, made on first use
    hipEvent_t lz4_ev[5] = {};                     // start, copies queued, decoded + counted (timed); index on the device; decoded, before K1 (timed)
    static constexpr int kLz4MaxSpans = 8;          // file mode: pinned spans the readers fill ahead of the copies (a ring)
    hipEvent_t lz4_landed[kLz4MaxPieces] = {}, lz4_joined[kLz4Streams] = {}, lz4_pin_free[kLz4MaxSpans] = {};
    uint8_t* lz4_pin = nullptr;                    // ... ONE page-locked allocation holding the ring (on the GPU's NUMA node)
    uint64_t lz4_pin_bytes = 0;
    RegisteredHost lz4_pin_reg;                    // (what it is made of: host_alloc_registered)
    RegisteredHost lz4_index_host;                 // page-locked staging of a segment's block index (a copy out of pageable memory costs
    uint64_t lz4_index_host_cap = 0;               //  the first call of a process 11 ms: the runtime sets its own staging up)
    uint8_t* zstd_scratch[kLz4Streams] = {};       // GPU Zstandard decoder: records / literals / checkpoints of a piece, per decode stream
    uint64_t zstd_scratch_cap[kLz4Streams] = {};   // (kept and released with the two large buffers)
    void* lz4_index = nullptr;                     // blocks + status + tally of a segment
    uint64_t lz4_index_cap = 0;
    bool lz4_ready = false;
    uint32_t lz4_idle = 0;                         // calls of other entry points since the decoder last ran
    uint64_t pinned_bytes = 0;
    static constexpr int kOrderEvents = 16;        // device-scope ordering events (no timing, no system fence), used round robin
    hipEvent_t order_ev[kOrderEvents] = {};
    unsigned order_next = 0;                       // guarded by user_mu
    bool order_ev_fence_free = false;              // flavour of the events currently in order_ev[] (guarded by user_mu)
    static constexpr int kSideEngines = 3;         // default engines only: where concurrent caller threads go (count_host_shared)
    std::atomic<Engine*> side[kSideEngines] = {};
    std::mutex side_mu;                            // creation / release of the side engines
    std::mutex user_mu;                            // guards user_ws and the ordering events
    std::list<std::pair<void*, Workspace>> user_ws;  // caller-owned streams, most recently used first (bounded)
};

// RAII: make `device` current for the calling thread and put the previous device back afterwards
class DeviceGuard {
  public:
    explicit DeviceGuard(int device);
    ~DeviceGuard();
    bool ok() const { return ok_; }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;

  private:
    int prev_ = -1;
    bool switched_ = false;
    bool ok_ = false;
};

// First statement of every C entry that touches engine or HIP state.  0: this process owns the library (the first caller's
// process claims it).  Non-zero in a child fork()ed after that: the error (naming the fork and the remedy) is recorded and
// printed -- unless `quiet`, for the release-type entries a child's interpreter may run on objects it inherited -- and no
// mutex, stream or HIP call has been touched.
int process_guard(const char* entry, bool quiet = false);
bool process_forked();                             // true in such a child (never claims)
#define FS_ENTRY() do { if (fsint::process_guard(__func__)) return -1; } while (0)
#define FS_ENTRY_PTR() do { if (fsint::process_guard(__func__)) return nullptr; } while (0)
#define FS_ENTRY_RELEASE() do { if (fsint::process_guard(__func__, true)) return; } while (0)

int fail_text(const char* msg);                    // records (thread-local) + prints, returns -1
int fail_hip(const char* what, hipError_t e);      // same with the HIP error text, returns non-zero
int fail_again(const char* full_text, int rc);     // re-records a message another thread already printed
const char* last_error_text();
void multi_forget();                               // flagstat_multi.hip: drop cached engine pointers (shutdown)

// default engine of `device` (created on first use); device < 0: the process default device
// (FLAGSTATS_hip_init / env FLAGSTATS_HIP_DEVICE / 0).  nullptr on failure (error recorded).
Engine* engine_for_device(int device);
Engine* default_engine();
// a private engine (own streams and buffers) on `device`; release with engine_destroy
Engine* engine_create(int device);               // refs = 1 (the creator's handle)
void engine_destroy(Engine* e);                  // releases the GPU resources now, the object with the last reference
void engine_retain(Engine* e);
void engine_release(Engine* e);
int engine_alive(const Engine& e);               // 0, or a recorded error if the engine was shut down
void shutdown_all();
int default_device();                              // -1 before the first successful init
int select_default_device(int device);             // FLAGSTATS_hip_init

// which device a device pointer lives on; fails loudly for host / unknown pointers
int device_of_pointer(const void* p, const char* what, int* device, bool* plain_device_memory = nullptr);
// a caller's stream must belong to `device` (NULL = that device's null stream)
int check_stream_device(hipStream_t s, int device);
// `waiter` waits (on the device; the host does not) for everything queued on `on` so far.  Knob "fence_free_events":
// the event carries no system-scope fence (a plain hipEventRecord costs the launch stream ~10 us of cache write-back
// per record here); off by default.
int stream_wait_stream(Engine& e, hipStream_t waiter, hipStream_t on);

// stream[1] / d_out[1] exist when this has returned 0 (joins the creation's helper thread; cheap afterwards).  Callers hold
// e.mu, or own the engine outright (creation, retirement).
int engine_second(Engine& e);
WorkerPool& engine_pool(Engine& e);                // e.mu held

uint32_t grid_for(const Engine& e);
int ensure_ws(Workspace& w, uint32_t grid, hipStream_t s);  // zeroed (stream-ordered on s) at creation
// K1 + K2 on `s`: d_out += (or =, OP_FLAGSTAT_STORE) counters of d_array[0..n).  Device must be current.
int count_device_async(Engine& e, const uint16_t* d_array, uint64_t n, uint64_t* d_out, hipStream_t s, Workspace& w,
                       int op = OP_FLAGSTAT, uint64_t* signal_word = nullptr, uint64_t signal_value = 0);
// the same for a caller-owned stream: validates devices, finds / creates the stream's workspace
int count_on_user_stream(const uint16_t* d_array, uint64_t n, uint64_t* d_out, void* stream, int op);
// host array -> counters through the engine's two-stream pipeline; takes e.mu
int count_host(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, int op = OP_FLAGSTAT);
// the same for the DEFAULT engine's reference-shaped entry points: concurrent caller threads do not queue behind each
// other, they spread over the engine and up to Engine::kSideEngines side engines of its device
int count_host_shared(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, int op = OP_FLAGSTAT);
// flagstat_blocks.hip: a host array through the block pipeline's page-locked chunks (worker threads copy 1 MiB slices; takes e.mu)
int count_host_staged(Engine& e, const uint16_t* h, uint64_t n, uint64_t* out, bool superset, int threads);

int stage_reserve(Engine& e, int slot, uint64_t flags);   // e.mu held
int pinned_reserve(Engine& e, uint64_t bytes, void* bufs[3]);  // e.mu held
void* host_alloc_on_node(size_t bytes, int numa_node);   // pinned, pages placed on `numa_node` when >= 0
// Page-locked memory the quick way: anonymous memory backed by transparent huge pages, first touched by threads bound to
// `numa_node`'s CPUs (so it lies there), then hipHostRegister-ed -- 48 MiB in 0.8 ms where hipHostMalloc takes 6-40 ms (it locks
// and maps 4 KiB pages one by one: 0.2-0.5 ms per MiB; profiles/r05/file_h2d.log), copies out of it at the link's rate.
// Falls back to host_alloc_on_node.  Release with host_free_registered (the pair keeps no table: the owner remembers `bytes`).
// touch_and_register = false: only the mapping is made (0.03 ms) -- the owner's own threads touch it as they fill it (their
// first writes place the pages) and host_register_late page-locks it before the first copy out of it.
RegisteredHost host_alloc_registered(size_t bytes, int numa_node, bool touch_and_register = true);   // ptr == nullptr on failure (error recorded)
bool host_register_late(RegisteredHost& r);   // false: the runtime refused; the memory stays usable as ordinary (pageable) memory
void host_free_registered(RegisteredHost& r);
extern thread_local double g_reg_times[4];   // the last host_alloc_registered's phases in ms (diagnostics)
uint64_t chunk_bytes();
// CPUs of host NUMA node `node` inside the calling process's own affinity mask (flagstat_blocks.hip); false: unknown / none
bool node_cpuset(int node, cpu_set_t* set);

// A block header that declares more decoded bytes than ANY payload of its size can decode to is refused by both index passes
// before a buffer is sized from it (a damaged 60-byte file must not make the reader allocate and zero-fill 2 GiB): an LZ4
// block grows by at most 255 output bytes per input byte (a match-length byte), a Zstandard frame by at most 128 KiB per
// 4-byte RLE block.  (The reference decodes such a block into its fixed 1,089,536-byte buffer with the DECLARED size as the
// capacity, benchmark/flagstats.cpp:296-316: undefined behaviour there, a loud error here.)
inline bool block_sizes_plausible(int codec, uint64_t decoded, uint64_t payload)
{
    return codec == 0 ? decoded <= payload * 255u + 64u : decoded <= (payload / 4u + 1u) * 131072u;
}

// LZ4 block file decoded on the GPU (flagstat_gpu_decode.hip).  img != nullptr: whole file image in memory; else fd: file mode.
struct Lz4GpuSource {
    const uint8_t* img = nullptr;
    int fd = -1;
    uint64_t bytes = 0;
    bool superset = false;
    int threads = 0;        // file mode: parallel preads (<= 0: up to 16)
    int codec = 0;          // payloads: 0 = LZ4 blocks, 1 = Zstandard frames
    bool by_size = false;   // the decoder was chosen by the size rule: a file of flags that hardly compress is handed back (kGpuDecodeRejected)
};
// The index pass over the block headers, apart from the run: plain host code that touches no engine state, so a caller makes
// it BEFORE taking e.mu (file mode: one pread per header; a mid-size file that the size rules hand back to the host threads
// never holds the lock that concurrent small FLAGSTATS_u16 callers wait for).
// Returns 0 (index made), kGpuDecodeRejected (decoder chosen by size and the rules say host threads) or < 0 (malformed: error recorded).
struct GpuFileIndex;
struct GpuFileIndexDeleter {
    void operator()(GpuFileIndex* p) const;
};
using GpuFileIndexPtr = std::unique_ptr<GpuFileIndex, GpuFileIndexDeleter>;
int lz4_gpu_index(const Lz4GpuSource& in, GpuFileIndexPtr& index);
int lz4_gpu_run(Engine& e, const Lz4GpuSource& in, const GpuFileIndex& index, uint64_t* out, struct ::FLAGSTATS_gpu_lz4_stats* stats);  // e.mu held, device current
// > 0 from lz4_gpu_run: the device could not hold the decoder's buffers (nothing was counted; the caller may take the host pipeline)
constexpr int kLz4GpuNoMemory = 77;
// > 0 from lz4_gpu_run: the GPU decoder did not take the file -- a Zstandard frame it does not handle or finds damaged, or
// (either codec, decoder chosen by size) flags that hardly compress; nothing was counted; the caller decodes the file on
// the host threads
constexpr int kGpuDecodeRejected = 78;
void lz4_gpu_release(Engine& e, bool all);         // e.mu held: the two large buffers (all: streams, events, index too)
void lz4_gpu_other_use(Engine& e);                 // e.mu held: another entry point ran (the idle rule of the kept buffers)

}  // namespace fsint

#endif
