// flagstat_kernels.h -- internal interface between the HIP kernels and the C-ABI shim.
#ifndef FLAGSTAT_KERNELS_H_
#define FLAGSTAT_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fsk {

constexpr int kThreads = 256;                       // 4 waves per workgroup
constexpr int kUnroll = 8;                          // 16-B vectors per lane per step
constexpr int kVecPerStep = kThreads * kUnroll;     // 2048 vectors = 32 KiB = 16384 flags
constexpr int kGroupTicketWord = 288;               // workspace block (u64 words): 8 group tickets, 16 words apart
constexpr int kGroupCopyWord = 512;                 // ... and 8 copies of the 32 slots (flagstat_count_core.h: grouped_epilogue)
constexpr int kInternal = 21;                       // 19 live counters (libflagstats.h:118-142) + primary-paired reads x {pass, fail}

// K1's dynamic schedule (STAGE 4): round 0 = c0 grid-stride steps per workgroup; the rest is cut into 2^lgq queues,
// each handed out in contiguous chunks of remaining / (workgroups per queue * div) steps (inv = 2^32 / that divisor),
// clamped to [1, cmax].  block = the workspace's 4 KiB block: u64 word 8 counts retired workgroups, word 16 * (i + 1)
// is queue i's counter (the last workgroup to retire zeroes them).  c0 * grid >= the number of full steps: fully
// static, block untouched.
struct DynSched {
    uint64_t* block;
    uint32_t c0;
    uint32_t inv;
    uint32_t cmax;
    uint32_t lgq;
};

// result pairs in pinned host memory for a polling host thread: pairs[2 * t] = slot t, pairs[2 * t + 1] = `value`, the
// sequence number that marks THIS call's result, both written by one 16-byte store (nullptr: none; out[] is used)
struct HostSignal {
    uint64_t* pairs;
    uint64_t value;
};

struct CountArgs {
    const void* a0;        // 16-B aligned-down base of the array
    uint64_t lo, hi;       // caller's flags occupy positions [lo, hi) of that grid
    uint64_t nsteps;       // ceil(ceil(hi/8) / kVecPerStep)
    uint64_t fast_begin;   // steps in [fast_begin, fast_end) are fully in range
    uint64_t fast_end;
    uint32_t grid;
    uint64_t* partials;    // [kInternal][grid]
    uint32_t* ticket;      // non-null: fused finalise by the last-arriving workgroup (must be 0 at launch)
    uint64_t* out;         // device uint64[32]
    DynSched dyn;
    HostSignal sig;
    int mode;              // bit 0: out = counters instead of +=; bit 1: superset slots (0/16 n_pair_all, 9 pass-QC reads);
                           // bit 2: direct epilogue -- every workgroup adds its totals to out[] with atomics, no K2;
                           // bit 3 (with bit 2): through the workspace's per-XCD copies (grouped_epilogue);
                           // bit 4: the waves of a workgroup start their first epoch at different counts;
                           // bit 5: latency form -- a grid of one workgroup stores all 32 slots and signals `sig`, no K2
};

}  // namespace fsk

extern "C" {
// bytes of workspace K1 needs for `grid` workgroups
size_t fsk_partials_bytes(uint32_t grid);
void fsk_warm(void);                      // loads K1 / K2's code object without a launch (a process's first call, off its critical path)
// K1 + K2 on `stream`: d_out32[32] += counters of d_array[0..n).  Asynchronous.
// variant bits 0-7: K1 schedule; bit 8: store instead of accumulate; bit 9: fused finalise (needs d_ticket);
// bit 10: superset slots; bit 11: direct atomic epilogue (accumulate form only; K1 alone, no K2)
// signal_word != nullptr (store form only): the kernel that finishes the counters writes them as 32 {value,
// signal_value} pairs of 16 bytes to signal_word[0..63] (pinned host memory) instead of d_out32 -- a host thread may
// poll the pairs instead of synchronising the stream.
hipError_t fsk_launch(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, uint64_t* d_partials,
                      uint32_t* d_ticket, uint64_t* d_out32, hipStream_t stream, uint64_t* signal_word = nullptr,
                      uint64_t signal_value = 0);
int fsk_variant_supported(int variant);   // K1 schedule compiled into this build?
// measurement build only (defined in flagstat_kernels_tuning.hip, `make tuning`; NULL in the product library -- callers
// check fsk_tuning_build() first).  Dynamic schedule policy (variant bit 7): first_pct = share of the steps in the static
// round 0 (0..100), div and cmax as in DynSched; arrays of fewer than min_steps_per_wg full steps per workgroup stay fully static
void fsk_set_dyn(uint32_t first_pct, uint32_t div, uint32_t cmax, uint32_t min_steps_per_wg) __attribute__((weak));
void fsk_set_dyn_queues(uint32_t lg_queues) __attribute__((weak));   // 2^lg_queues (<= 16) counters
void fsk_set_anatomy(int bits) __attribute__((weak));                // skip parts of K1 to time the rest (results wrong)
// what the product launcher shares with the measurement build's: the epilogue policy knobs, the last mode word, K2
void fsk_launch_policy(int* stagger, uint32_t* group_min_grid, uint64_t* group_max_steps);
void fsk_note_mode(int mode);
hipError_t fsk_launch_finalize(uint64_t* d_partials, uint32_t grid, uint64_t* d_out32, int mode, uint64_t n, fsk::HostSignal sig,
                               hipStream_t stream);
// K1's direct epilogue adds to per-XCD copies first when the grid has at least this many workgroups (0: always)
void fsk_set_group_min_grid(uint32_t min_grid);
// ... and at most this many steps per workgroup (default 40); beyond that workgroups finish apart and add straight to out[]
void fsk_set_group_max_steps(uint64_t max_steps_per_workgroup);
int fsk_last_mode(void);                  // K1 mode word of the most recent fsk_launch (bit 3 = two-level epilogue); tests
void fsk_set_epoch_stagger(int on);       // 1 (default): wave w of a workgroup starts its first epoch at step count 64 * w
int fsk_tuning_build(void);               // 1: the measurement build (make tuning / make TUNING=1): flagstat_kernels_tuning.hip is linked
// positional popcount (flagstat_pospopcnt.hip): d_out16[16] += bit counts.  d_partials as for fsk_launch.
// direct != 0: the count kernel adds its workgroup totals to d_out16 with atomics (device memory only), no finalize launch.
hipError_t fsk_launch_pospopcnt(const uint16_t* d_array, uint64_t n, uint32_t grid, uint64_t* d_partials,
                                uint64_t* d_out16, hipStream_t stream, int direct);
// shader clock under load (measurement only): d_out[2 * b] = shader-clock cycles, d_out[2 * b + 1] = 100 MHz reference ticks
// that workgroup b (one wave) spent spinning; ticks <= 1e8 (1 s)
hipError_t fsk_clock_probe(uint64_t* d_out, uint32_t grid, uint64_t ticks, hipStream_t stream);
// on-device input makers (flagstat_generate.hip)
hipError_t fsk_generate(uint16_t* d_array, uint64_t n, int kind, uint64_t seed, uint32_t mask, uint64_t first_index,
                        hipStream_t stream);
}

#endif
