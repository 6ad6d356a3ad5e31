// flagstat_gpu_decode.hip -- host side of the block decode ON the GPU for both codecs of the reference's block files (row f1;
// kernels: flagstat_lz4_kernels.hip for raw LZ4 blocks, flagstat_zstd_kernels.hip for Zstandard frames; behind the block-file
// entries for large files, knobs "lz4_decoder" / "zstd_decoder").  Plain host code: no device code in this file, so it also
// builds against the test-only HIP stand-in (tests/hoststub) and runs under ThreadSanitizer and Address/UB sanitizers.
// (Until r04 this file was flagstat_lz4_gpu.hip; the internal names lz4_gpu_* stayed.)
//
// The reference's block reader decodes every block with liblz4's LZ4_decompress_safe on the host
// (benchmark/flagstats.cpp:311-316); the host pipeline of this repo (flagstat_blocks.hip) does the same on N host threads
// and is PCIe-bound on the DECODED bytes (27 Gflags/s).  Here the file crosses PCIe as it is (2.1-3.4x fewer bytes) and is
// decoded on the device:
//   * the file goes over in PIECES of whole blocks on the engine's copy stream (a piece = 1/16 of the file, 256..512
//     blocks); every piece's blocks are decoded by their own launch as soon as the piece has landed, on one of TWO decode
//     streams.  Two, and 256 blocks at least: the runtime multiplexes a process's streams onto four hardware queues, of
//     which the decode streams get two (a rocprofv3 timeline of sixteen 128-block pieces on four streams shows launches on
//     two queues only, two in flight at any time, half of the 512 blocks the chip holds: 31 ms for 2^30 flags, where
//     256-block pieces take 2x ms, profiles/r04/lz4_timeline_*.txt); a launch lasts at least one block's 2.5-3.5 ms, so
//     what is exposed behind the last copy is one piece's decode;
//   * ONE K1 pass over the decoded buffer at the end (0.15 ms per GiB; counting each piece behind its decode changed
//     nothing measurable at large sizes and costs K1 launches whose 168-VGPR waves wait for decode workgroups to leave);
//   * file mode reads with a pool of parallel preads into the engine's three pinned chunk buffers, one span ahead of the
//     copies;
//   * streams, events and the small device buffers are made once per engine; the two large buffers (compressed, decoded)
//     stay with the engine between calls and are released when eight other calls have passed without the decoder
//     (knob "lz4_gpu_keep_bytes"), on every failure, and by FLAGSTATS_hip_shutdown;
//   * compressed and decoded bytes of a run are resident together, so a file larger than a third of the free device
//     memory (or 16 GiB of flags) goes through in several segments.
// Measurements: profiles/r04/lz4_*.log, profiles/r05/; DESIGN.md section 6, development notes in HISTORY.md.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include "../../include/libflagstats_hip_probe.h"
#include "flagstat_engine.h"
#include "flagstat_kernels.h"
#include "flagstat_lz4_kernels.h"
#include "flagstat_zstd_kernels.h"

namespace fsint {

// Host-side phases of a call (env FLAGSTATS_HIP_GPU_DECODE_TIMES=1 prints them on stderr: tests/perf/cold_start.py reads
// what the FIRST call of a process spends where)
namespace {
struct PhaseClock {
    using clk = std::chrono::steady_clock;
    bool on = false;
    clk::time_point last;
    double index = 0, meminfo = 0, streams = 0, alloc_comp = 0, alloc_out = 0, alloc_out_own = 0, alloc_small = 0, alloc_scratch = 0, preset = 0, pinned = 0, queue = 0, wait = 0;
    void start()
    {
        const char* k = std::getenv("FLAGSTATS_HIP_GPU_DECODE_TIMES");
        on = k && std::atoi(k) != 0;
        if (on) last = clk::now();
    }
    void lap(double& into)
    {
        if (!on) return;
        const clk::time_point now = clk::now();
        into += std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
    }
};
}  // namespace

static void lz4_gpu_free_ring(Engine& e)
{
    host_free_registered(e.lz4_pin_reg);
    e.lz4_pin = nullptr;
    e.lz4_pin_bytes = 0;
}

void lz4_gpu_release(Engine& e, bool all)
{
    for (int i = 0; i < 2; ++i) {
        uint8_t* p = e.lz4_buf[i];
        e.lz4_buf[i] = nullptr;
        e.lz4_cap[i] = 0;
        if (p) (void)hipFree(p);
    }
    for (int i = 0; i < Engine::kLz4Streams; ++i) {
        uint8_t* p = e.zstd_scratch[i];
        e.zstd_scratch[i] = nullptr;
        e.zstd_scratch_cap[i] = 0;
        if (p) (void)hipFree(p);
    }
    if (!all) return;
    for (int i = 0; i < Engine::kLz4Streams; ++i) {
        hipStream_t& x = e.lz4_stream[i];
        if (x && i > 0) (void)hipStreamDestroy(x);   // ([0] is the engine's own second stream, borrowed)
        x = nullptr;
    }
    auto drop = [](hipEvent_t& x) {
        if (x) (void)hipEventDestroy(x);
        x = nullptr;
    };
    for (hipEvent_t& x : e.lz4_ev) drop(x);
    for (hipEvent_t& x : e.lz4_landed) drop(x);
    for (hipEvent_t& x : e.lz4_joined) drop(x);
    for (hipEvent_t& x : e.lz4_pin_free) drop(x);
    if (e.lz4_index) (void)hipFree(e.lz4_index);
    e.lz4_index = nullptr;
    e.lz4_index_cap = 0;
    lz4_gpu_free_ring(e);
    host_free_registered(e.lz4_index_host);
    e.lz4_index_host_cap = 0;
    e.lz4_ready = false;
}

void lz4_gpu_other_use(Engine& e)
{
    if (!e.lz4_buf[0] && !e.lz4_buf[1] && !e.zstd_scratch[0]) return;
    if (++e.lz4_idle >= static_cast<uint32_t>(Engine::kLz4IdleCalls)) lz4_gpu_release(e, false);
}

// Streams and events: once per engine.  The 75 events take 0.03 ms; the two decode streams 15-20 ms (a hardware queue each:
// tests/perf/micro/file_h2d.hip, profiles/r05/file_h2d.log), which the FIRST call of a process would wait for -- so the streams
// are made on a helper thread beside the buffers' allocation, the pinned ring and the first reads, and the thread is joined
// before the first decode launch (lz4_gpu_segment).
static int lz4_gpu_events(Engine& e)
{
    hipError_t err = hipSuccess;
    for (hipEvent_t& x : e.lz4_ev)
        if (err == hipSuccess && !x) err = hipEventCreate(&x);
    for (hipEvent_t& x : e.lz4_landed)
        if (err == hipSuccess && !x) err = hipEventCreateWithFlags(&x, hipEventDisableTiming);
    for (hipEvent_t& x : e.lz4_joined)
        if (err == hipSuccess && !x) err = hipEventCreateWithFlags(&x, hipEventDisableTiming);
    for (hipEvent_t& x : e.lz4_pin_free)
        if (err == hipSuccess && !x) err = hipEventCreateWithFlags(&x, hipEventDisableTiming | hipEventBlockingSync);
    if (err != hipSuccess) {
        lz4_gpu_release(e, true);
        return fail_hip("GPU block decoder: events", err);
    }
    return 0;
}

static int lz4_gpu_streams(Engine& e, int codec)   // (runs on a helper thread: makes the engine's device current itself)
{
    DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    // The first decode stream is the engine's own second stream (idle while the decoder has the engine: every call ends with a
    // stream wait), so ONE stream is made here: a stream costs 8-10 ms, and the first call of a process waited 12 ms for two.
    hipError_t err = hipSuccess;
    if (engine_second(e)) return -1;   // (joins the engine creation's helper thread: a first call of a process may get here before it is done)
    e.lz4_stream[0] = e.stream[1];
    for (int i = 1; i < Engine::kLz4Streams; ++i)
        if (err == hipSuccess && !e.lz4_stream[i]) err = hipStreamCreateWithFlags(&e.lz4_stream[i], hipStreamNonBlocking);
    if (err != hipSuccess) return fail_hip("GPU block decoder: streams", err);
    // ... and the code objects the call is about to launch from are loaded here instead of inside the first launches (the
    // occupancy queries load the decode kernels' translation unit, fsk_warm K1's)
    const char* wk = std::getenv("FLAGSTATS_HIP_GPU_WARM");   // (A/B: 0 = leave the loads to the first launches)
    const int warm = wk ? std::atoi(wk) : 3;
    if (warm & 1) {
        if (codec == 1)
            (void)fsk_zstd_frames_per_cu();
        else
            (void)fsk_lz4_blocks_per_cu(knobs().lz4_gpu_kernel.load() == 0 ? fsk::LZ4K_WORKGROUP : fsk::LZ4K_WAVE);
    }
    if (warm & 2) fsk_warm();
    return 0;
}

// One segment: file bytes [file_lo, file_lo + bytes) hold `blocks` (offsets relative to the segment), dpos decoded bytes.
// e.mu must be held.  Returns kLz4GpuNoMemory when the device cannot hold the buffers (nothing has been queued then).
static int lz4_gpu_segment(Engine& e, const Lz4GpuSource& in, uint64_t file_lo, const std::vector<fsk::GpuBlock>& blocks,
                           uint64_t bytes, uint64_t dpos, uint64_t n_flags, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats, PhaseClock& pc_,
                           bool wave_kernel)
{
    const uint8_t* img = in.img ? in.img + file_lo : nullptr;
    int rc = lz4_gpu_events(e);
    if (rc) return rc;
    hipStream_t s = e.stream[0];
    constexpr uint32_t nstreams = Engine::kLz4Streams;
    // first use on this engine: the decode streams are made beside everything up to the first decode launch
    std::thread maker;
    int maker_rc = 0;
    std::string maker_err;
    struct Joiner {
        std::thread& t;
        ~Joiner()
        {
            if (t.joinable()) t.join();
        }
    } joiner{maker};
    if (!e.lz4_ready) {
        auto make = [&] {
            maker_rc = lz4_gpu_streams(e, in.codec);
            if (maker_rc) maker_err = last_error_text();
        };
        try {
            maker = std::thread(make);
        } catch (const std::system_error&) {
            make();   // (a pids / RLIMIT_NPROC limit: made here, in line)
        }
    }
    // The decoded buffer is not needed before the first decode launch, and on some hosts a hipMalloc of a gigabyte and more takes
    // 30 ms where it takes 0.03 on others (profiles/r05/cold_start_exit.log: 1.6 GB, every fresh process of that box): when it
    // has to be made it is made on a thread of its own, beside the index upload, the first reads and the first copies.
    std::thread out_maker;
    uint8_t* d_out = nullptr;   // (known once join_streams / join_out has returned 0)
    int out_rc = 0;   // 0, kLz4GpuNoMemory, or -1 with out_err
    std::string out_err;
    Joiner out_joiner{out_maker};
    bool streams_wait_index = false;   // the decode streams still have to be told to wait for the index (event 3)
    auto join_out = [&]() -> int {
        if (out_maker.joinable()) {
            pc_.lap(pc_.queue);
            out_maker.join();
            pc_.lap(pc_.alloc_out);   // (only what the call WAITED for it)
        }
        if (out_rc) {
            const int r = out_rc;
            out_rc = 0;
            return r == kLz4GpuNoMemory ? r : fail_again(out_err.c_str(), r);
        }
        d_out = e.lz4_buf[1];
        return 0;
    };
    auto join_streams = [&]() -> int {
        const int orc = join_out();
        if (orc) return orc;
        if (maker.joinable()) {
            pc_.lap(pc_.queue);
            maker.join();
            pc_.lap(pc_.streams);   // (only what the call WAITED for the helper thread)
            if (maker_rc) {
                lz4_gpu_release(e, true);
                return fail_again(maker_err.c_str(), maker_rc);
            }
            e.lz4_ready = true;
        }
        if (streams_wait_index) {
            streams_wait_index = false;
            for (uint32_t i = 0; i < nstreams; ++i) {
                const hipError_t e_ = hipStreamWaitEvent(e.lz4_stream[i], e.lz4_ev[3], 0);
                if (e_ != hipSuccess) return fail_hip("hipStreamWaitEvent(index on the device)", e_);
            }
        }
        return 0;
    };
    // everything queued so far must be over before an error leaves (the buffers may be released by the caller)
    auto settle = [&] {
        if (maker.joinable()) maker.join();
        if (out_maker.joinable()) out_maker.join();
        (void)hipStreamSynchronize(s);
        for (hipStream_t x : e.lz4_stream)
            if (x) (void)hipStreamSynchronize(x);
    };
#define LZG_TRY(expr)                            \
    do {                                         \
        hipError_t e_ = (expr);                  \
        if (e_ != hipSuccess) {                  \
            rc = fail_hip(#expr, e_);            \
            settle();                            \
            return rc;                           \
        }                                        \
    } while (0)
    // the two large buffers belong to the engine and are reused by the next segment / file (knob "lz4_gpu_keep_bytes")
    const uint64_t want[2] = {bytes + 64, dpos + 16};
    // (capacities grow in steps of 64 MiB; env FLAGSTATS_HIP_GPU_BUFFER_GRAIN: the sanitizer builds ask for exact sizes, so that
    // an access behind what a segment needs is a report instead of a hit in the slack)
    uint64_t grain = 64ull << 20;
    if (const char* gk = std::getenv("FLAGSTATS_HIP_GPU_BUFFER_GRAIN"))
        if (std::strtoull(gk, nullptr, 0) >= 16) grain = std::strtoull(gk, nullptr, 0);
    auto grow = [&e, grain](int i, uint64_t bytes_wanted, std::string& err_text) -> int {   // 0, kLz4GpuNoMemory, -1 (err_text)
        uint8_t* old = e.lz4_buf[i];
        e.lz4_buf[i] = nullptr;
        e.lz4_cap[i] = 0;
        if (old) {
            const hipError_t f_ = hipFree(old);
            if (f_ != hipSuccess) {
                fail_hip("hipFree(decoder buffer)", f_);
                err_text = last_error_text();
                return -1;
            }
        }
        const uint64_t cap = (bytes_wanted + grain - 1) / grain * grain;
        // (env FLAGSTATS_HIP_GPU_OUT_CAP: tests make requests for the DECODED buffer above it fail, as a full device would)
        const char* ock = i == 1 ? std::getenv("FLAGSTATS_HIP_GPU_OUT_CAP") : nullptr;
        const hipError_t e_ = ock && cap > std::strtoull(ock, nullptr, 0) ? hipErrorOutOfMemory : hipMalloc(&e.lz4_buf[i], cap);
        if (e_ != hipSuccess) {
            (void)hipGetLastError();
            e.lz4_buf[i] = nullptr;
            return kLz4GpuNoMemory;
        }
        e.lz4_cap[i] = cap;
        return 0;
    };
    if (e.lz4_cap[1] < want[1]) {
        auto make_out = [&] {
            DeviceGuard g2(e.device);
            if (!g2.ok()) {
                out_err = last_error_text();
                out_rc = -1;
                return;
            }
            const PhaseClock::clk::time_point t0 = PhaseClock::clk::now();
            out_rc = grow(1, want[1], out_err);
            pc_.alloc_out_own += std::chrono::duration<double, std::milli>(PhaseClock::clk::now() - t0).count();   // (read after the join)
        };
        try {
            out_maker = std::thread(make_out);
        } catch (const std::system_error&) {
            make_out();
        }
    }
    if (e.lz4_cap[0] < want[0]) {
        std::string text;
        rc = grow(0, want[0], text);
        if (rc) {
            settle();
            return rc;   // (kLz4GpuNoMemory, or the error of a hipFree: message set)
        }
        pc_.lap(pc_.alloc_comp);
    }
    uint8_t* const d_comp = e.lz4_buf[0];
    // blocks | status | tally in one small allocation that grows with the largest segment seen
    const uint64_t off_status = (blocks.size() * sizeof(fsk::GpuBlock) + 255) & ~255ull;
    const uint64_t off_tally = (off_status + blocks.size() * sizeof(uint32_t) + 255) & ~255ull;
    const uint64_t index_bytes = off_tally + fsk::kLz4TallyWords * 8;
    if (e.lz4_index_cap < index_bytes) {
        void* old = e.lz4_index;
        e.lz4_index = nullptr;
        e.lz4_index_cap = 0;
        if (old) LZG_TRY(hipFree(old));
        const uint64_t cap = (index_bytes + 65535) & ~65535ull;
        const hipError_t e_ = hipMalloc(&e.lz4_index, cap);
        if (e_ != hipSuccess) {
            (void)hipGetLastError();
            e.lz4_index = nullptr;
            return kLz4GpuNoMemory;
        }
        e.lz4_index_cap = cap;
        pc_.lap(pc_.alloc_small);
    }
    fsk::GpuBlock* const d_blocks = static_cast<fsk::GpuBlock*>(e.lz4_index);
    uint32_t* const d_status = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(e.lz4_index) + off_status);
    unsigned long long* const d_tally = reinterpret_cast<unsigned long long*>(static_cast<uint8_t*>(e.lz4_index) + off_tally);
    // only the slack between blocks (16-byte slots) and behind dropped odd bytes needs zero flags: blocks of the reference's
    // writer are whole multiples of 16 bytes, so this is normally nothing at all
    bool ragged = false;
    for (const fsk::GpuBlock& b : blocks) ragged = ragged || (b.dst_len & 15u);
    double t_pre[4] = {0, 0, 0, 0};
    auto pre_lap = [&](int k) {
        if (!pc_.on) return;
        const PhaseClock::clk::time_point now = PhaseClock::clk::now();
        t_pre[k] = std::chrono::duration<double, std::milli>(now - pc_.last).count();
        pc_.preset += t_pre[k];
        pc_.last = now;
    };
    pre_lap(0);
    LZG_TRY(hipEventRecord(e.lz4_ev[0], s));
    pre_lap(1);
    if (ragged) {
        rc = join_out();
        if (rc) {
            settle();
            return rc;
        }
        LZG_TRY(hipMemsetAsync(d_out, 0, dpos + 16, s));
    }
    LZG_TRY(hipMemsetAsync(d_status, 0xFF, blocks.size() * sizeof(uint32_t), s));
    LZG_TRY(hipMemsetAsync(d_tally, 0, fsk::kLz4TallyWords * 8, s));
    LZG_TRY(hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), s));
    pre_lap(2);
    {
        // The index goes up out of a small page-locked buffer of the engine's: a copy out of pageable memory made the first call of a
        // process wait 11.5 ms for the runtime's own staging set-up (profiles/r05/cold_start.log); the engine's streams are idle
        // between calls (every call ends with a stream wait), so the buffer is free to be rewritten here.
        const uint64_t nbytes = blocks.size() * sizeof(fsk::GpuBlock);
        if (e.lz4_index_host_cap < nbytes) {
            host_free_registered(e.lz4_index_host);
            e.lz4_index_host_cap = 0;
            e.lz4_index_host = host_alloc_registered((nbytes + (1u << 20) - 1) & ~static_cast<uint64_t>((1u << 20) - 1), e.numa_node);
            if (e.lz4_index_host.ptr) e.lz4_index_host_cap = (nbytes + (1u << 20) - 1) & ~static_cast<uint64_t>((1u << 20) - 1);
        }
        const void* src = blocks.data();
        if (e.lz4_index_host_cap >= nbytes) {
            std::memcpy(e.lz4_index_host.ptr, blocks.data(), nbytes);
            src = e.lz4_index_host.ptr;
        }
        LZG_TRY(hipMemcpyAsync(d_blocks, src, nbytes, hipMemcpyHostToDevice, s));
    }
    pre_lap(3);
    if (pc_.on) std::fprintf(stderr, "gpu decode, presets: before %.2f ms, hipEventRecord %.2f, three hipMemsetAsync %.2f, hipMemcpyAsync of the index (%zu bytes) %.2f\n", t_pre[0], t_pre[1], t_pre[2], blocks.size() * sizeof(fsk::GpuBlock), t_pre[3]);
    // env FLAGSTATS_HIP_GPU_LZ4_RING = 8 (default) | 16: KiB of recent output per wave in LDS (r03's kernel only)
    const char* rk = std::getenv("FLAGSTATS_HIP_GPU_LZ4_RING");
    const bool big_ring = rk && std::atoi(rk) == 16;
    // knob "lz4_gpu_kernel": 0 = the workgroup pipeline (eight waves per block, 64 KiB window in LDS), 1 = r03's wave per block
    const bool zstd = in.codec == 1;
    // (wave_kernel: a block beyond what the workgroup kernel's records can address -- 16 MiB of payload, 256 MiB decoded; the
    // reference's writer makes 1,024,000-byte blocks -- sends the whole file to the wave-per-block kernel, which has no such limit)
    const int kernel = zstd || (knobs().lz4_gpu_kernel.load() == 0 && !wave_kernel) ? fsk::LZ4K_WORKGROUP : (big_ring ? fsk::LZ4K_WAVE_RING16 : fsk::LZ4K_WAVE);
    const char* pk = std::getenv("FLAGSTATS_HIP_GPU_LZ4_PROFILE");  // tuning: per-phase wave cycles on stderr
    const bool prof = pk && std::atoi(pk) != 0;
    // Pieces.  The workgroup kernel holds 512 blocks at a time, a block takes 2.5-3.5 ms whatever else runs, and TWO
    // decode launches run at a time (two hardware queues): pieces of 1/16 of the file, but not under 256 blocks (two
    // launches then fill the chip) nor over 512.  The decode time left exposed behind the last copy is one piece's.
    // r03's kernel wants few large ones (a launch lasts 20-30 ms however small): one per 1024 blocks, at most 4.
    // env FLAGSTATS_HIP_GPU_LZ4_CHUNKS overrides.
    const char* ck = std::getenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS");
    uint32_t npieces;
    if (ck) {
        npieces = static_cast<uint32_t>(std::atoi(ck));
    } else if (zstd) {
        // Zstandard: an equal share is at most 1024 frames, two pieces at least (the first is half a share, the others up to
        // 1536 frames then: what the chain kernel holds at a time) -- a frame takes ~6 ms through the four kernels whatever
        // else runs, so few large launches, two of them in flight; the stash, records, literals and checkpoints between the
        // kernels take 8.5 MB of scratch per frame and decode stream
        // (r05, profiles/r05/zstd_pieces_by_size.log: up to ~2400 frames a share of 800 is the better cut -- 1611 frames, the README's
        // file: three pieces 16.5 ms, two 17.9; 2098 frames: three 21.0, two 25.1, four 22.3; 1049 frames: two 13.5, three 14.6 --
        // what is exposed behind the last copy is the last piece's way through the four kernels, and a chain launch lasts 3 ms
        // however few frames it holds; 4195 frames: four or five pieces 35.9-36.1, six 37.2)
        const uint64_t share = blocks.size() <= 2400 ? 800 : 1024;
        npieces = static_cast<uint32_t>((blocks.size() + share - 1) / share);
        if (npieces < 2) npieces = 2;
    } else if (kernel == fsk::LZ4K_WORKGROUP) {
        uint64_t per = blocks.size() / 16;
        per = per < 256 ? 256 : (per > 512 ? 512 : per);
        npieces = static_cast<uint32_t>((blocks.size() + per - 1) / per);
    } else {
        npieces = static_cast<uint32_t>((blocks.size() + 1023) / 1024);
        if (npieces > 4) npieces = 4;
    }
    if (npieces < 1) npieces = 1;
    if (npieces > static_cast<uint32_t>(Engine::kLz4MaxPieces)) npieces = Engine::kLz4MaxPieces;
    if (npieces > blocks.size()) npieces = static_cast<uint32_t>(blocks.size());
    pc_.lap(pc_.preset);
    // File mode: a ring of page-locked spans, filled by parallel preads ahead of the copies.  ONE allocation on the GPU's NUMA
    // node, made on first use and kept with the engine: 4 spans of 16 MiB -- r04 took the host pipeline's three 64 MiB chunk
    // buffers, whose page-locking cost the first call of a process 40-53 ms (0.2-0.25 ms per MiB; 48 MiB in one call: 7.6 ms;
    // profiles/r05/file_h2d.log, cold_start_before.log); small spans also shorten what nothing overlaps: the first span's
    // read and the last span's copy.  env FLAGSTATS_HIP_GPU_SPAN_MIB / FLAGSTATS_HIP_GPU_SPANS override (A/B, tests); a chunk
    // size below 16 MiB (knob "chunk_flags") makes the spans that small (tests: many spans in a small file).
    uint8_t* pinned[Engine::kLz4MaxSpans] = {};
    hipEvent_t* const pin_free = e.lz4_pin_free;
    uint64_t span_cap = 0;
    int readers = 0, nspans_ring = 0;
    if (!img) {
        const char* sk = std::getenv("FLAGSTATS_HIP_GPU_SPAN_MIB");
        span_cap = sk && std::atoi(sk) > 0 ? static_cast<uint64_t>(std::atoi(sk)) << 20 : 16ull << 20;
        if (!sk && chunk_bytes() < span_cap) span_cap = (chunk_bytes() + 4095) & ~4095ull;
        if (span_cap < (1ull << 20)) span_cap = 1ull << 20;
        const char* nk = std::getenv("FLAGSTATS_HIP_GPU_SPANS");
        nspans_ring = nk ? std::atoi(nk) : 4;
        if (nspans_ring < 2) nspans_ring = 2;
        if (nspans_ring > Engine::kLz4MaxSpans) nspans_ring = Engine::kLz4MaxSpans;
        const uint64_t ring_bytes = span_cap * static_cast<uint64_t>(nspans_ring);
        if (e.lz4_pin_bytes < ring_bytes) {
            if (e.lz4_pin) {
                LZG_TRY(hipStreamSynchronize(s));
                lz4_gpu_free_ring(e);
            }
            // Page-locked the quick way (host_alloc_registered: huge pages touched on the GPU's node, then registered): r04's three
            // hipHostMalloc-ed 64 MiB buffers cost the first call of a process 40-53 ms, this ring 1 ms.
            // ... and not even that here: the mapping is made now, the READERS touch it as they fill it (parallel, on the GPU's node:
            // that places the pages) and it is page-locked just before the first copy out of it (ring_register below).
            e.lz4_pin_reg = host_alloc_registered(ring_bytes, e.numa_node, false);
            if (!e.lz4_pin_reg.ptr) {
                settle();
                return -1;
            }
            e.lz4_pin = static_cast<uint8_t*>(e.lz4_pin_reg.ptr);
            e.lz4_pin_bytes = ring_bytes;
        }
        for (int i = 0; i < nspans_ring; ++i) pinned[i] = e.lz4_pin + span_cap * static_cast<uint64_t>(i);
        pc_.lap(pc_.pinned);
        readers = in.threads > 0 ? in.threads : static_cast<int>(std::thread::hardware_concurrency());
        if (readers > 16) readers = 16;
        if (readers < 1) readers = 1;
    }
    LZG_TRY(hipEventRecord(e.lz4_ev[3], s));  // index on the device, status and counters preset
    streams_wait_index = true;                // (told to the decode streams when they exist: join_streams)
    // pieces: blocks [first, last) = segment bytes [lo, hi), split by compressed bytes
    struct Piece {
        uint64_t first, last, lo, hi;
    };
    std::vector<Piece> pieces;
    // Zstandard: the first piece is half an equal share -- the decode starts when it has landed, and its four kernels take
    // 5 ms before the second stream has anything to do (2^31 flags 43.0 -> 40.5 ms, the README-size file 21.3 -> 19.1 ms;
    // LZ4, whose pieces are a sixteenth of the file, does not gain: profiles/r04/first_piece.log).  env FLAGSTATS_HIP_GPU_FIRST_PIECE
    // = per cent of an equal share overrides (0 / 100: equal pieces).
    const char* fpk = std::getenv("FLAGSTATS_HIP_GPU_FIRST_PIECE");
    int first_pct = fpk ? std::atoi(fpk) : (zstd ? 50 : 0);
    if (first_pct <= 0 || first_pct >= 100) first_pct = 100;
    // ... and the LAST piece may be a fraction of a share too (env FLAGSTATS_HIP_GPU_LAST_PIECE, per cent): what is left exposed
    // behind the last copy is the last piece's way through the kernels
    const char* lpk = std::getenv("FLAGSTATS_HIP_GPU_LAST_PIECE");
    int last_pct = lpk ? std::atoi(lpk) : 100;
    if (last_pct <= 0 || last_pct >= 100) last_pct = 100;
    auto cut_pieces = [&] {
        pieces.clear();
        // the first piece is first_pct of an equal share; the others share what is left, the last one weighing last_pct of a middle one
        const uint64_t t0 = npieces > 1 ? bytes / npieces * static_cast<uint64_t>(first_pct) / 100u : bytes;
        const uint64_t wsum = npieces > 2 ? 100u * (npieces - 2u) + static_cast<uint64_t>(last_pct) : 100u;
        uint64_t wrun = 0;
        for (uint64_t first = 0; pieces.size() < npieces && first < blocks.size();) {
            const uint64_t c = pieces.size();
            uint64_t target = t0;
            if (c > 0) {
                wrun += (c + 1 == npieces && npieces > 2) ? static_cast<uint64_t>(last_pct) : 100u;
                target = t0 + static_cast<uint64_t>(static_cast<unsigned __int128>(bytes - t0) * wrun / wsum);
            }
            uint64_t last = first + 1;
            while (last < blocks.size() && (c + 1 == npieces || blocks[last].src_off + blocks[last].src_len <= target)) ++last;
            pieces.push_back(Piece{first, last, blocks[first].src_off - 8, blocks[last - 1].src_off + blocks[last - 1].src_len});
            first = last;
        }
    };
    cut_pieces();
    // Zstandard: scratch between the four kernels (8.5 MB per 1,024,000-byte frame in flight), one per decode stream, sized for the
    // largest piece.  A device that cannot hold it gets more and smaller pieces (half the frames per launch each time) before the
    // call gives up: the scratch is what a busy device runs out of first, and smaller launches only cost some overlap.
    uint32_t z_max_dst = 0;
    if (zstd) {
        for (const fsk::GpuBlock& b : blocks) z_max_dst = b.dst_len > z_max_dst ? b.dst_len : z_max_dst;
        if (z_max_dst > fsk::kZstdMaxFrameBytes) {
            settle();
            if (knobs().zstd_decoder.load() == 1) fail_text("GPU Zstandard decoder: a block decodes to more than it takes (64 MiB)");
            return kGpuDecodeRejected;
        }
        for (;;) {
            // every decode stream's scratch holds the largest piece THAT stream gets (piece c runs on stream c % nstreams): with the
            // first piece half a share, the two differ by a third
            uint64_t most[nstreams] = {};
            uint64_t most_all = 0;
            for (size_t c = 0; c < pieces.size(); ++c) {
                const uint64_t k = pieces[c].last - pieces[c].first;
                most[c % nstreams] = k > most[c % nstreams] ? k : most[c % nstreams];
                most_all = k > most_all ? k : most_all;
            }
            bool ok = true;
            for (uint32_t i = 0; i < nstreams && ok; ++i) {
                if (!most[i]) continue;
                const uint64_t need = fsk_zstd_scratch_bytes(z_max_dst, static_cast<uint32_t>(most[i]));
                if (e.zstd_scratch_cap[i] >= need) continue;
                uint8_t* old = e.zstd_scratch[i];
                e.zstd_scratch[i] = nullptr;
                e.zstd_scratch_cap[i] = 0;
                if (old) LZG_TRY(hipFree(old));
                const uint64_t sgrain = grain < (16ull << 20) ? grain : (16ull << 20);
                const uint64_t cap = (need + sgrain - 1) / sgrain * sgrain;
                // (env FLAGSTATS_HIP_GPU_SCRATCH_CAP: tests make requests above it fail, as a busy device would)
                const char* sck = std::getenv("FLAGSTATS_HIP_GPU_SCRATCH_CAP");
                const hipError_t e_ = sck && cap > std::strtoull(sck, nullptr, 0) ? hipErrorOutOfMemory : hipMalloc(&e.zstd_scratch[i], cap);
                if (e_ != hipSuccess) {
                    (void)hipGetLastError();
                    e.zstd_scratch[i] = nullptr;
                    ok = false;
                } else {
                    e.zstd_scratch_cap[i] = cap;
                }
            }
            if (ok) break;
            if (most_all <= 1 || npieces >= static_cast<uint32_t>(Engine::kLz4MaxPieces) || npieces >= blocks.size()) {
                settle();
                return kLz4GpuNoMemory;
            }
            npieces = npieces * 2 > static_cast<uint32_t>(Engine::kLz4MaxPieces) ? static_cast<uint32_t>(Engine::kLz4MaxPieces) : npieces * 2;
            if (npieces > blocks.size()) npieces = static_cast<uint32_t>(blocks.size());
            cut_pieces();
        }
        pc_.lap(pc_.alloc_scratch);
    }
    const uint32_t pieces_done = static_cast<uint32_t>(pieces.size());
    // piece c has been queued on the copy stream: decode its blocks behind it
    auto launch_piece = [&](uint32_t c) -> int {
        const int jrc = join_streams();
        if (jrc) return jrc;
        hipError_t e_ = hipEventRecord(e.lz4_landed[c], s);
        hipStream_t ds = e.lz4_stream[c % nstreams];
        if (e_ == hipSuccess) e_ = hipStreamWaitEvent(ds, e.lz4_landed[c], 0);
        if (e_ != hipSuccess) return fail_hip("hipEventRecord / hipStreamWaitEvent(piece landed)", e_);
        const Piece& pc = pieces[c];
        if (zstd)
            e_ = fsk_zstd_decode(d_comp, d_blocks + pc.first, static_cast<uint32_t>(pc.last - pc.first), d_out, d_status + pc.first, d_tally,
                                 e.zstd_scratch[c % nstreams], e.zstd_scratch_cap[c % nstreams], z_max_dst, prof ? 1 : 0, ds);
        else
            e_ = fsk_lz4_decode(kernel, d_comp, d_blocks + pc.first, static_cast<uint32_t>(pc.last - pc.first), d_out, d_status + pc.first,
                                d_tally, prof ? 1 : 0, ds);
        return e_ == hipSuccess ? 0 : fail_hip(zstd ? "Zstandard decode kernel launch" : "LZ4 decode kernel launch", e_);
    };
    if (img) {
        for (uint32_t c = 0; c < pieces.size() && !rc; ++c) {
            hipError_t e_ = hipMemcpyAsync(d_comp + pieces[c].lo, img + pieces[c].lo, pieces[c].hi - pieces[c].lo, hipMemcpyHostToDevice, s);
            rc = e_ == hipSuccess ? launch_piece(c) : fail_hip("hipMemcpyAsync(block image piece)", e_);
        }
    } else {
        // File mode.  The pieces are cut into spans of one ring slot; a pool of `readers` threads preads the released spans
        // (every thread its share of each) while this thread queues the copies behind them: span i may be read as soon as the
        // copy that last used its slot (span i - ring) has left the host, so the readers run up to ring - 1 spans ahead of the
        // copy queue.  (Threads started per span cost a third of the read time: r03, 83 -> 7x ms.)
        struct Span {
            uint64_t at, len;
            int ends_piece;  // piece that is complete once this span is queued, or -1
        };
        std::vector<Span> spans;
        for (uint32_t c = 0; c < pieces.size(); ++c)
            for (uint64_t at = pieces[c].lo; at < pieces[c].hi; at += span_cap) {
                const uint64_t len = pieces[c].hi - at < span_cap ? pieces[c].hi - at : span_cap;
                spans.push_back(Span{at, len, at + len == pieces[c].hi ? static_cast<int>(c) : -1});
            }
        const size_t ring = static_cast<size_t>(nspans_ring);
        std::mutex m;
        std::condition_variable cv_work, cv_done;
        size_t released = 0;                       // spans [0, released) may be read
        std::vector<int> done(spans.size(), 0);    // reader threads finished per span
        std::vector<std::atomic<uint64_t>> slice_next(spans.size());   // next slice of a span to be read
        for (auto& a : slice_next) a.store(0, std::memory_order_relaxed);
        bool stop = false, failed = false;
        // (the pinned spans live on the GPU's host NUMA node: the readers run there too, like the host pipeline's decoders)
        cpu_set_t node_cpus;
        const bool pin_threads = knobs().numa.load() && node_cpuset(e.numa_node, &node_cpus);
        auto reader = [&](int) {
            if (pin_threads) (void)pthread_setaffinity_np(pthread_self(), sizeof node_cpus, &node_cpus);
            for (size_t i = 0; i < spans.size(); ++i) {
                {
                    std::unique_lock<std::mutex> ul(m);
                    cv_work.wait(ul, [&] { return released > i || stop; });
                    if (stop) return;
                }
                // A span is read in slices of 256 KiB that the readers TAKE (an atomic counter per span), not in one fixed share each:
                // a reader that is descheduled or sits on a slower core then holds up one slice, not a sixteenth of the span (the
                // box's CPUs are shared with other jobs: with fixed shares the file-mode rounds of a soak spread over 33-44 ms where
                // image mode, which no CPU touches, stays within 1 %; profiles/r05/lz4hc9_gpu_soak_gpu_rounds_only.log)
                const Span& sp = spans[i];
                constexpr uint64_t kSlice = 256u << 10;
                const uint64_t nslices = (sp.len + kSlice - 1) / kSlice;
                bool ok = true;
                uint8_t* base = pinned[i % ring];
                for (;;) {
                    const uint64_t k = slice_next[i].fetch_add(1, std::memory_order_relaxed);
                    if (k >= nslices) break;
                    uint64_t o = k * kSlice;
                    const uint64_t end = o + kSlice < sp.len ? o + kSlice : sp.len;
                    while (o < end) {
                        const ssize_t r = pread(in.fd, base + o, end - o, static_cast<off_t>(file_lo + sp.at + o));
                        if (r <= 0) {
                            ok = false;
                            break;
                        }
                        o += static_cast<uint64_t>(r);
                    }
                    if (!ok) break;
                }
                std::lock_guard<std::mutex> g(m);
                if (!ok) failed = true;
                if (++done[i] == readers) cv_done.notify_one();
            }
        };
        // (the engine's worker pool: the sixteen readers are parked threads, not made and joined per call)
        WorkerPool& pool = engine_pool(e);
        const bool pooled = readers > 0 && !spans.empty();
        if (pooled && !pool.start(readers, reader)) {
            settle();
            return -1;
        }
        auto queue_span = [&](size_t i) -> int {
            {
                std::unique_lock<std::mutex> ul(m);
                cv_done.wait(ul, [&] { return done[i] == readers; });
                if (failed) return fail_text("block file: short read");
            }
            if (!e.lz4_pin_reg.registered && !e.lz4_pin_reg.refused) {
                // (first use of the ring: the spans released so far have been touched by the readers -- on the GPU's node; the
                // registration covers the WHOLE ring, so pages of spans no reader has reached yet are faulted in here, on this
                // thread's node.  A refusal leaves ordinary memory, out of which the copies still work, through the runtime's
                // staging, and is not asked for again: RegisteredHost::refused)
                DeviceGuard g2(e.device);
                const bool locked = g2.ok() && host_register_late(e.lz4_pin_reg);
                if (pc_.on) std::fprintf(stderr, "gpu decode, pinned ring of %llu MiB: mapped, touched by the readers' first reads, hipHostRegister %.2f ms%s\n",
                                         static_cast<unsigned long long>(e.lz4_pin_bytes >> 20), g_reg_times[2], locked ? "" : " (refused)");
            }
            hipError_t e_ = hipMemcpyAsync(d_comp + spans[i].at, pinned[i % ring], spans[i].len, hipMemcpyHostToDevice, s);
            if (e_ == hipSuccess) e_ = hipEventRecord(pin_free[i % ring], s);
            if (e_ != hipSuccess) return fail_hip("hipMemcpyAsync(block file span)", e_);
            return spans[i].ends_piece >= 0 ? launch_piece(static_cast<uint32_t>(spans[i].ends_piece)) : 0;
        };
        size_t next_release = 0;
        for (size_t q = 0; q < spans.size() && !rc; ++q) {
            // Release what the ring allows before waiting for span q's readers: spans up to q + ring - 2 -- the slot of span
            // q + ring - 1 is span q - 1's, whose copy has only just been queued: waiting for THAT here would leave the copy
            // engine idle until span q is queued; waiting for span q - 2's copy leaves it one copy to work on.
            const size_t ahead = ring > 2 ? q + ring - 1 : q + 1;
            while (!rc && next_release < spans.size() && next_release < ahead) {
                if (next_release >= ring) {
                    // (the copy of span next_release - ring, queued at an earlier q, has left this slot)
                    const hipError_t e_ = hipEventSynchronize(pin_free[next_release % ring]);
                    if (e_ != hipSuccess) rc = fail_hip("hipEventSynchronize(pinned span)", e_);
                }
                if (!rc) {
                    {
                        std::lock_guard<std::mutex> g(m);
                        released = ++next_release;
                    }
                    cv_work.notify_all();
                }
            }
            if (!rc) rc = queue_span(q);
        }
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cv_work.notify_all();
        if (pooled) pool.wait();
    }
    if (rc) {
        settle();
        return rc;
    }
    rc = join_streams();
    if (rc) {
        settle();
        return rc;
    }
    pc_.lap(pc_.queue);
    LZG_TRY(hipEventRecord(e.lz4_ev[1], s));  // every piece has landed
    for (uint32_t i = 0; i < nstreams; ++i) {
        LZG_TRY(hipEventRecord(e.lz4_joined[i], e.lz4_stream[i]));
        LZG_TRY(hipStreamWaitEvent(s, e.lz4_joined[i], 0));
    }
    // One K1 pass over the whole decoded buffer (0.15 ms per GiB).
    LZG_TRY(hipEventRecord(e.lz4_ev[4], s));  // ... and is decoded
    rc = count_device_async(e, reinterpret_cast<const uint16_t*>(d_out), dpos / 2, e.d_out[0], s, e.ws[0],
                            OP_FLAGSTAT | (in.superset ? OP_SUPERSET : 0));
    if (rc) {
        settle();
        return rc;
    }
    LZG_TRY(hipEventRecord(e.lz4_ev[2], s));  // ... is decoded and counted
    LZG_TRY(hipMemcpyAsync(e.h_out, e.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    unsigned long long tally[fsk::kLz4TallyWords] = {0};
    LZG_TRY(hipMemcpyAsync(tally, d_tally, sizeof tally, hipMemcpyDeviceToHost, s));
    std::vector<uint32_t> st(blocks.size());
    LZG_TRY(hipMemcpyAsync(st.data(), d_status, st.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LZG_TRY(hipStreamSynchronize(s));
    pc_.lap(pc_.wait);
    uint64_t bad = 0, first_bad = 0;
    uint32_t first_code = 0;
    auto tally_bad = [&] {
        bad = first_bad = 0;
        first_code = 0;
        for (size_t i = 0; i < st.size(); ++i)
            if (st[i] != 0) {
                if (!bad) {
                    first_bad = i;
                    first_code = st[i];
                }
                ++bad;
            }
    };
    tally_bad();
    // Zstandard frames that ran out of block slots (kZstdTooManyBlocks: windows of a few KiB, compressors that flush often -- a
    // first pass reserves tables for 4 blocks per 128 KiB): when nothing else is wrong, those frames alone go through the four kernels
    // once more with room for a block per KiB, and K1 counts the buffer again.  Rare, so it costs the common file nothing.
    if (zstd && bad) {
        std::vector<uint32_t> again;
        for (size_t i = 0; i < st.size(); ++i)
            if (st[i] == fsk::kZstdTooManyBlocks) again.push_back(static_cast<uint32_t>(i));
        if (again.size() == bad) {
            const uint32_t min_blocks = z_max_dst / 1024u + 16u;
            const uint32_t per = 64;   // frames a launch: 1 MB frames with 1 KiB blocks take 12 MB of scratch each
            const uint64_t need = fsk_zstd_scratch_bytes_ex(z_max_dst, per, min_blocks);
            std::vector<fsk::GpuBlock> hb(per);
            void *d_rb = nullptr, *d_rs = nullptr, *d_rscratch = nullptr;
            hipError_t e_ = hipMalloc(&d_rb, per * sizeof(fsk::GpuBlock));
            if (e_ == hipSuccess) e_ = hipMalloc(&d_rs, per * sizeof(uint32_t));
            if (e_ == hipSuccess) e_ = hipMalloc(&d_rscratch, need);
            std::vector<uint32_t> rs(per);
            for (size_t at = 0; at < again.size() && e_ == hipSuccess; at += per) {
                const uint32_t k = static_cast<uint32_t>(again.size() - at < per ? again.size() - at : per);
                for (uint32_t i = 0; i < k; ++i) hb[i] = blocks[again[at + i]];
                e_ = hipMemcpyAsync(d_rb, hb.data(), k * sizeof(fsk::GpuBlock), hipMemcpyHostToDevice, s);
                if (e_ == hipSuccess) e_ = hipMemsetAsync(d_rs, 0xFF, k * sizeof(uint32_t), s);
                if (e_ == hipSuccess)
                    e_ = fsk_zstd_decode_ex(d_comp, static_cast<const fsk::GpuBlock*>(d_rb), k, d_out, static_cast<uint32_t*>(d_rs), d_tally, d_rscratch, need, z_max_dst,
                                            min_blocks, 0, s);
                if (e_ == hipSuccess) e_ = hipMemcpyAsync(rs.data(), d_rs, k * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
                if (e_ == hipSuccess) e_ = hipStreamSynchronize(s);   // (hb / rs are reused by the next round)
                for (uint32_t i = 0; i < k && e_ == hipSuccess; ++i) st[again[at + i]] = rs[i];
            }
            if (e_ == hipSuccess) e_ = hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), s);
            if (e_ == hipSuccess) {
                rc = count_device_async(e, reinterpret_cast<const uint16_t*>(d_out), dpos / 2, e.d_out[0], s, e.ws[0], OP_FLAGSTAT | (in.superset ? OP_SUPERSET : 0));
                if (!rc) e_ = hipMemcpyAsync(e.h_out, e.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
                if (!rc && e_ == hipSuccess) e_ = hipStreamSynchronize(s);
            }
            (void)hipStreamSynchronize(s);
            if (d_rb) (void)hipFree(d_rb);
            if (d_rs) (void)hipFree(d_rs);
            if (d_rscratch) (void)hipFree(d_rscratch);
            if (e_ != hipSuccess && e_ != hipErrorOutOfMemory) return fail_hip("GPU Zstandard decoder: second pass over frames of many blocks", e_);
            (void)hipGetLastError();
            if (rc) return rc;
            if (e_ == hipSuccess) tally_bad();   // (out of memory for the second pass: the frames stay "not taken")
        }
    }
    float h2d = 0, dec = 0, cnt = 0, pipe = 0;
    LZG_TRY(hipEventElapsedTime(&h2d, e.lz4_ev[0], e.lz4_ev[1]));
    LZG_TRY(hipEventElapsedTime(&dec, e.lz4_ev[1], e.lz4_ev[4]));
    LZG_TRY(hipEventElapsedTime(&cnt, e.lz4_ev[4], e.lz4_ev[2]));
    LZG_TRY(hipEventElapsedTime(&pipe, e.lz4_ev[0], e.lz4_ev[2]));
    if (prof && zstd) {
        const double nb = static_cast<double>(blocks.size());
        const int ne = fsk::kZstdEmitters, ns = fsk::kZstdScanners;
        std::fprintf(stderr, "zstd gpu profile: %d frame(s) per CU fit (execution kernel); cycles per frame: prepare %.3g (literals %.3g, tables %.3g) | "
                             "chain %.3g per wave of two frames, %.0f steps a frame | records %.3g (%.1f relaxation rounds a batch) | "
                             "emit %.3g x %d (waiting %.1f %%), scan %.3g x %d (waiting %.1f %%), copy %.3g (waiting %.1f %%) | per frame: %.0f records, %.0f far matches, "
                             "%.0f batches in %.0f groups, %.0f chunks, %.1f %% with pointers inside (%.2f doubling rounds each)\n",
                     fsk_zstd_frames_per_cu(), tally[17] / nb, tally[18] / nb, tally[19] / nb, tally[20] / (tally[21] + 1e-9), tally[22] / nb, tally[23] / nb,
                     tally[24] / (tally[25] + 1e-9), tally[15] / nb / ne, ne, 100.0 * tally[16] / (tally[15] + 1e-9), tally[8] / nb / ns, ns,
                     100.0 * tally[9] / (tally[8] + 1e-9), tally[12] / nb, 100.0 * tally[13] / (tally[12] + 1e-9), tally[0] / nb, tally[1] / nb, tally[5] / nb,
                     tally[6] / nb, tally[14] / nb, 100.0 * tally[11] / (tally[14] + 1e-9), tally[11] ? static_cast<double>(tally[10]) / tally[11] : 0.0);
    } else if (prof) {
        std::fprintf(stderr, "lz4 gpu profile: kernel %d, %d block(s) per CU fit\n", kernel, fsk_lz4_blocks_per_cu(kernel));
        if (kernel == fsk::LZ4K_WORKGROUP) {
            const double nb = static_cast<double>(blocks.size());
            const int ne = fsk::kLz4WgEmitters, ns = fsk::kLz4WgScanners;
            std::fprintf(stderr, "lz4 gpu profile (workgroup pipeline), cycles per block and wave: walk %.3g (waiting %.1f %%), emit %.3g x %d (waiting %.1f %%), "
                                 "scan %.3g x %d (waiting %.1f %%), copy %.3g (waiting %.1f %%) | per block: %.0f tiles, %.0f windows (%.1f sequences each), "
                                 "%.0f scalar sequences, %.0f chunks, %.1f %% with pointers inside (%.2f doubling rounds each)\n",
                         tally[2] / nb, 100.0 * tally[3] / (tally[2] + 1e-9), tally[15] / nb / ne, ne, 100.0 * tally[16] / (tally[15] + 1e-9),
                         tally[8] / nb / ns, ns, 100.0 * tally[9] / (tally[8] + 1e-9), tally[12] / nb, 100.0 * tally[13] / (tally[12] + 1e-9), tally[7] / nb,
                         tally[5] / nb, tally[5] ? static_cast<double>(tally[0] - tally[6]) / tally[5] : 0.0,
                         tally[6] / nb, tally[14] / nb, 100.0 * tally[11] / (tally[14] + 1e-9), tally[11] ? static_cast<double>(tally[10]) / tally[11] : 0.0);
            if (tally[7])
                std::fprintf(stderr, "lz4 gpu profile, walker cycles per tile: exits %.0f, chain %.0f, members %.0f, records (incl. waiting for the emitters) %.0f\n",
                             static_cast<double>(tally[17]) / tally[7], static_cast<double>(tally[18]) / tally[7], static_cast<double>(tally[19]) / tally[7],
                             static_cast<double>(tally[20]) / tally[7]);
        } else {
            const double tot = static_cast<double>(tally[10]) + 1e-9;
            std::fprintf(stderr, "lz4 gpu profile (wave per block, ring 8 KiB): wave cycles %.3g | copy %.1f %% far %.1f %% lit %.1f %% slow %.1f %% flush %.1f %% cover %.1f %% parse %.1f %% | "
                                 "%llu batches (%.1f seq each), %llu passes + %llu alone, %llu short-literal + %llu slow sequences, %llu far\n",
                         tot, 100 * tally[2] / tot, 100 * tally[3] / tot, 100 * tally[11] / tot, 100 * tally[4] / tot, 100 * tally[5] / tot, 100 * tally[6] / tot,
                         100 * tally[7] / tot, tally[8], tally[8] ? static_cast<double>(tally[0] - tally[9] - tally[12]) / tally[8] : 0.0, tally[13], tally[14], tally[12], tally[9], tally[1]);
        }
    }
    stats->bad_blocks += bad;
    stats->h2d_ms += h2d;
    stats->decode_ms += dec;
    stats->count_ms += cnt;  // the ONE K1 pass over the decoded buffer that ends the pipeline (0.15 ms per GiB)
    stats->sequences += tally[0];
    stats->far_matches += tally[1];
    stats->ring_kib = zstd ? (fsk::kZstdWindow + 4096u) / 1024u : (kernel == fsk::LZ4K_WORKGROUP ? 66 : (kernel == fsk::LZ4K_WAVE ? 8 : 16));
    stats->chunks += pieces_done;
    stats->pipeline_ms += pipe;
    stats->readers = static_cast<uint64_t>(readers);
    if (!bad) {
        if (in.superset) e.h_out[9] -= dpos / 2 - n_flags;  // zero flags in the slack between blocks are not reads (see run_pipeline)
        for (int k = 0; k < 32; ++k) out[k] += e.h_out[k];
    }
    if (bad) {
        // (Zstandard with the decoder chosen by size: the caller decodes the file with libzstd, whose verdict -- and message --
        // counts; nothing of this run has reached the caller's counters)
        if (zstd && knobs().zstd_decoder.load() != 1) return kGpuDecodeRejected;
        char buf[192];
        std::snprintf(buf, sizeof buf, "block file: %llu block(s) failed to decode to their declared size (GPU %s decoder; first: block %llu, code %u)",
                      static_cast<unsigned long long>(bad), zstd ? "Zstandard" : "LZ4", static_cast<unsigned long long>(first_bad), first_code);
        const int rc_ = fail_text(buf);
        return zstd ? kGpuDecodeRejected : rc_;
    }
    return 0;
#undef LZG_TRY
}

struct GpuFileIndex {
    std::vector<fsk::GpuBlock> blocks;
    uint64_t dpos = 0, n_flags = 0, usum = 0;
    uint32_t max_src = 0, max_dst = 0;
    double index_ms = 0;
};

void GpuFileIndexDeleter::operator()(GpuFileIndex* p) const { delete p; }

int lz4_gpu_index(const Lz4GpuSource& in, GpuFileIndexPtr& index)
{
    const auto t0 = std::chrono::steady_clock::now();
    index.reset(new GpuFileIndex());
    GpuFileIndex& ix = *index;
    const uint8_t* img = in.img;
    const uint64_t bytes = in.bytes;
    // index: int32 uncompressed size, int32 compressed size, payload (benchmark/flagstats.cpp:119-138)
    std::vector<fsk::GpuBlock>& blocks = ix.blocks;
    uint64_t pos = 0;
    while (pos < bytes) {
        if (bytes - pos < 8) return fail_text("block file: truncated block header");
        int32_t us, cs;
        uint8_t hdr[8];
        if (img) {
            std::memcpy(hdr, img + pos, 8);
        } else if (pread(in.fd, hdr, 8, static_cast<off_t>(pos)) != 8) {
            return fail_text("block file: cannot read block header");
        }
        std::memcpy(&us, hdr, 4);
        std::memcpy(&cs, hdr + 4, 4);
        if (us < 0 || cs < 0) return fail_text("block file: negative size in block header");
        if (static_cast<uint64_t>(cs) > bytes - pos - 8) return fail_text("block file: block payload runs past end of file");
        if (!block_sizes_plausible(in.codec, static_cast<uint64_t>(us), static_cast<uint64_t>(cs)))
            return fail_text("block file: a block header declares more decoded bytes than a payload of its size can hold");
        blocks.push_back(fsk::GpuBlock{pos + 8, ix.dpos, static_cast<uint32_t>(cs), static_cast<uint32_t>(us)});
        ix.n_flags += static_cast<uint64_t>(us) >> 1;  // as benchmark/flagstats.cpp:323
        ix.usum += static_cast<uint64_t>(us);
        ix.dpos += (static_cast<uint64_t>(us) + 15) & ~15ull;
        pos += 8 + static_cast<uint64_t>(cs);
        ix.max_src = static_cast<uint32_t>(cs) > ix.max_src ? static_cast<uint32_t>(cs) : ix.max_src;
        ix.max_dst = static_cast<uint32_t>(us) > ix.max_dst ? static_cast<uint32_t>(us) : ix.max_dst;
    }
    ix.index_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // Flags that hardly compress: the host-thread pipeline then moves about as many bytes over PCIe as the GPU decoder and has
    // next to nothing to decode (raw blocks, literal runs), while the GPU decoders' literal paths are their slow ones
    // (profiles/r04/incompressible_blockfiles.log: 11 ms against 14 (LZ4) / 22-29 (Zstandard) for 512 MiB).  With the
    // decoder chosen by size such a file goes to the host threads: decoded bytes below 1.25 x (LZ4) / 1.9 x (Zstandard) the file's
    // (where the two decoders cross on blocks with a growing share of noise: profiles/r04/ratio_sweep.log).
    if (!blocks.empty()) {
        const uint64_t pct = in.codec == 1 ? 190 : 125;
        if (in.by_size && ix.usum * 100 < bytes * pct) return kGpuDecodeRejected;
        // ... and the size rule proper (flagstat_blocks.hip, decode_on_gpu): a file below the knob's compressed size is taken if
        // it decodes to at least 2.5 x the knob (default: 64 MiB of file or 160 MiB of flags)
        const uint64_t from = in.codec == 0 ? knobs().lz4_gpu_min_bytes.load() : knobs().zstd_gpu_min_bytes.load();
        if (in.by_size && bytes < from && ix.usum * 2 < from * 5) return kGpuDecodeRejected;
    }
    return 0;
}

int lz4_gpu_run(Engine& e, const Lz4GpuSource& in, const GpuFileIndex& ix, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats)
{
    const auto t_start = std::chrono::steady_clock::now();
    PhaseClock pc;
    pc.start();
    pc.index = ix.index_ms;   // (made before the engine's lock was taken: lz4_gpu_index)
    // The engine's second stream is this decoder's first decode stream.  In a process whose engine has only just been made its
    // helper thread may still be at it: wait for it HERE, with nothing else going on -- joined later, from the stream-making
    // thread below, the helper's runtime calls ran beside this thread's allocations and index upload and slowed each other down
    // (one-shot HC-9 file: 132 ms start -> counters against 128; profiles/r06/cold_start.log, cold_start_eager_second_stream.log)
    {
        const int src = engine_second(e);
        if (src) return src;
    }
    pc.lap(pc.streams);
    const uint64_t bytes = in.bytes;
    const std::vector<fsk::GpuBlock>& blocks = ix.blocks;
    const uint64_t dpos = ix.dpos, n_flags = ix.n_flags, usum = ix.usum;
    const uint32_t max_src = ix.max_src, max_dst = ix.max_dst;
    FLAGSTATS_gpu_lz4_stats local;
    if (!stats) stats = &local;
    {
        *stats = FLAGSTATS_gpu_lz4_stats{};
        stats->n_blocks = blocks.size();
        stats->n_flags = n_flags;
        stats->compressed_bytes = bytes;
        stats->decoded_bytes = dpos;
        stats->uncompressed_bytes = usum;
    }
    if (blocks.empty()) return 0;
    // What the kernels cannot address is known from the index -- a capability limit, not damage.  LZ4: the workgroup kernel's
    // records carry 24-bit input and 28-bit output positions (status 10 if it met such a block); a file with a block of 16 MiB of
    // payload or 256 MiB decoded goes to the wave-per-block kernel, which has no such limit.
    const bool wave_kernel = in.codec == 0 && (max_src >= (1u << 24) || max_dst >= (1u << 28));
    // Segments: compressed and decoded bytes of a segment are resident on the device together, so a file larger than the
    // card can hold goes through in several of them, one after the other (each with its own pieces, decode launches and
    // counting pass).  Default: a third of the free device memory -- for Zstandard, of what is free beside the scratch of
    // two launches (8.5 MB per frame in flight; when that alone is more than half the device's free memory the pieces shrink
    // instead, see lz4_gpu_segment) -- and at most 16 GiB of decoded bytes; env FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES overrides (tests).
    uint64_t seg_cap = 16ull << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            uint64_t avail = free_b;
            // (what the engine already holds for the decoder is reused, not needed on top)
            avail += e.lz4_cap[0] + e.lz4_cap[1];
            for (uint64_t c : e.zstd_scratch_cap) avail += c;
            if (in.codec == 1 && max_dst <= fsk::kZstdMaxFrameBytes) {
                const uint64_t frames = blocks.size() < 1536 ? blocks.size() : 1536;
                uint64_t scratch = static_cast<uint64_t>(Engine::kLz4Streams) * fsk_zstd_scratch_bytes(max_dst, static_cast<uint32_t>(frames));
                if (scratch > avail / 2) scratch = avail / 2;
                avail -= scratch;
            }
            if (avail / 3 < seg_cap) seg_cap = avail / 3;
        } else {
            (void)hipGetLastError();
        }
        const char* sb = std::getenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES");
        if (sb && *sb) seg_cap = std::strtoull(sb, nullptr, 0);
    }
    pc.lap(pc.meminfo);
    // The counters of all segments are collected here and reach the caller's out[] only when the last segment has succeeded: a
    // segment the decoder does not take (a Zstandard frame with a checksum in the middle of a large file) or cannot hold then
    // still hands the WHOLE file to the host-thread pipeline, and a failed call never leaves partial sums behind.
    uint64_t acc[32] = {0};
    std::vector<fsk::GpuBlock> seg;
    for (size_t b0 = 0; b0 < blocks.size();) {
        size_t b1 = b0;
        uint64_t dsz = 0;
        while (b1 < blocks.size()) {
            const uint64_t padded = (static_cast<uint64_t>(blocks[b1].dst_len) + 15) & ~15ull;
            if (b1 > b0 && dsz + padded > seg_cap) break;
            dsz += padded;
            ++b1;
        }
        const uint64_t file_lo = blocks[b0].src_off - 8, file_hi = blocks[b1 - 1].src_off + blocks[b1 - 1].src_len;
        const uint64_t d0 = blocks[b0].dst_off;
        seg.assign(blocks.begin() + static_cast<std::ptrdiff_t>(b0), blocks.begin() + static_cast<std::ptrdiff_t>(b1));
        uint64_t seg_flags = 0;
        for (fsk::GpuBlock& g : seg) {
            g.src_off -= file_lo;
            g.dst_off -= d0;
            seg_flags += static_cast<uint64_t>(g.dst_len) >> 1;
        }
        const int rc = lz4_gpu_segment(e, in, file_lo, seg, file_hi - file_lo, dsz, seg_flags, acc, stats, pc, wave_kernel);
        if (rc) {
            // nothing of a failed run stays on the device, nothing of it has reached out[]: the two "not taken" codes mean the same
            // in the first segment and in a later one
            lz4_gpu_release(e, false);
            return rc;
        }
        ++stats->segments;
        b0 = b1;
    }
    for (int k = 0; k < 32; ++k) out[k] += acc[k];
    // what stays with the engine for the next call: knob "lz4_gpu_keep_bytes" (~0 = automatic: at most a quarter of the device)
    e.lz4_idle = 0;
    {
        uint64_t keep = knobs().lz4_gpu_keep_bytes.load();
        if (keep == ~0ull) {
            size_t free_b = 0, total_b = 0;
            keep = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? total_b / 4 : 0;
        }
        uint64_t held = e.lz4_cap[0] + e.lz4_cap[1];
        for (uint64_t c : e.zstd_scratch_cap) held += c;
        if (held > keep) lz4_gpu_release(e, false);
    }
    stats->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() + ix.index_ms * 1e-3;
    if (pc.on) {
        double keep_rule = 0;
        pc.lap(keep_rule);
        std::fprintf(stderr, "gpu decode, host side (ms): index %.2f | device memory query %.2f | streams + events %.2f | hipMalloc compressed %.2f, decoded %.2f waited for (%.2f on its own thread), "
                             "index %.2f, scratch %.2f | presets + index upload queued %.2f | pinned ring %.2f | reading + queueing the pieces %.2f | waiting for the device %.2f | keep rule %.2f | "
                             "call %.2f (stream events: copies %.2f, decode behind the last copy %.2f, K1 %.2f)\n",
                     pc.index, pc.meminfo, pc.streams, pc.alloc_comp, pc.alloc_out, pc.alloc_out_own, pc.alloc_small, pc.alloc_scratch, pc.preset, pc.pinned, pc.queue, pc.wait, keep_rule,
                     stats->wall_s * 1e3, stats->h2d_ms, stats->decode_ms, stats->count_ms);
    }
    return 0;
}

}  // namespace fsint

extern "C" int FLAGSTATS_hip_blockimage_lz4_gpu(const void* image, uint64_t bytes, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats)
{
    FS_ENTRY();
    if (!out) return fsint::fail_text("NULL out");
    if (!image && bytes) return fsint::fail_text("NULL image");
    fsint::Engine* ep = fsint::default_engine();
    if (!ep) return -1;
    fsint::Engine& e = *ep;
    static const uint8_t empty = 0;
    fsint::Lz4GpuSource src;
    src.img = image ? static_cast<const uint8_t*>(image) : &empty;
    src.bytes = bytes;
    fsint::GpuFileIndexPtr index;
    int rc = fsint::lz4_gpu_index(src, index);
    if (rc) return rc;   // (< 0: malformed; by_size is off here, so no size rule applies)
    std::lock_guard<std::mutex> lk(e.mu);
    if (fsint::engine_alive(e)) return -1;
    fsint::DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    rc = fsint::lz4_gpu_run(e, src, *index, out, stats);
    // (this entry has no host pipeline behind it: the "not taken" codes are errors here)
    if (rc == fsint::kLz4GpuNoMemory) return fsint::fail_text("GPU block decoder: the device cannot hold the file's compressed and decoded bytes");
    if (rc == fsint::kGpuDecodeRejected) return fsint::fail_text("GPU block decoder: the file is not one the decoder takes");
    return rc;
}
