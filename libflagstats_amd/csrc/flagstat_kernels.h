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
constexpr int kInternal = 21;                       // 19 live counters (libflagstats.h:118-142) + primary-paired reads x {pass, fail}

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
    int mode;              // bit 0: out = counters instead of +=; bit 1: superset slots (0/16 n_pair_all, 9 pass-QC reads);
                           // bit 2: direct epilogue -- every workgroup adds its totals to out[] with atomics, no K2
};

}  // namespace fsk

extern "C" {
// bytes of workspace K1 needs for `grid` workgroups
size_t fsk_partials_bytes(uint32_t grid);
// K1 + K2 on `stream`: d_out32[32] += counters of d_array[0..n).  Asynchronous.
// variant bits 0-6: K1 schedule; bit 8: store instead of accumulate; bit 9: fused finalise (needs d_ticket);
// bit 10: superset slots; bit 11: direct atomic epilogue (accumulate form only; K1 alone, no K2)
hipError_t fsk_launch(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, uint64_t* d_partials,
                      uint32_t* d_ticket, uint64_t* d_out32, hipStream_t stream);
int fsk_variant_supported(int variant);   // K1 schedule compiled into this build?
void fsk_set_anatomy(int bits);           // tuning builds only: skip parts of K1 to time the rest (results wrong)
int fsk_tuning_build(void);               // 1: built with -DFLAGSTAT_TUNING_VARIANTS (make TUNING=1)
// positional popcount (flagstat_pospopcnt.hip): d_out16[16] += bit counts.  d_partials as for fsk_launch.
// direct != 0: the count kernel adds its workgroup totals to d_out16 with atomics (device memory only), no finalize launch.
hipError_t fsk_launch_pospopcnt(const uint16_t* d_array, uint64_t n, uint32_t grid, uint64_t* d_partials,
                                uint64_t* d_out16, hipStream_t stream, int direct);
// read-only bandwidth probe (measurement only)
hipError_t fsk_read_probe(const void* d_buf, uint64_t bytes, uint32_t grid, int nt, uint32_t* d_sink, hipStream_t stream);
// on-device input makers (flagstat_generate.hip)
hipError_t fsk_generate(uint16_t* d_array, uint64_t n, int kind, uint64_t seed, uint32_t mask, uint64_t first_index,
                        hipStream_t stream);
}

#endif
