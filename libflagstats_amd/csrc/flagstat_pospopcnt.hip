// flagstat_pospopcnt.hip -- SURVEY.md section 8 row f4: plain 16-bit positional popcount.
//
// Counterpart of libalgebra's `STORM_pospopcnt_u16(const uint16_t* data, size_t len, uint32_t* out)`
// (python/libalgebra.h:3496-3551; scalar statement :566-574): out[j] = number of words with bit j
// set.  It is K1 without the flagstat front end: every packed dword (2 words) is already a plane
// of 32 one-bit columns, so the 4 dwords of a 16-byte vector feed the carry-save tree directly
// (2 x v_bitop3_b32 per CSA, 31 CSAs per 32 dwords) and the weight-32 carry enters the same
// scalar-steered binary-counter chain.  Same load schedule as K1 (non-temporal 16 B/lane loads,
// waves interleaved at 1 KiB, rolling re-issue, one workgroup per CU), same zero-fill treatment
// of ragged heads and tails (a zero word has no bits).  HBM-bound: 2 bytes per word.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flagstat_device.h"
#include "flagstat_kernels.h"

namespace fsk {

constexpr int kPosDepth = 8;

struct PosLane {
    uint32_t p1, p2, p4, p8, p16;            // planes of weight 1..16
    uint32_t A[kPosDepth], B[kPosDepth];     // chain level j: weight 32 << j (accumulator, pending)
    uint32_t acc[16];
};

template <int J>
__device__ __forceinline__ void pos_chain_push(PosLane& s, uint32_t blk, uint32_t c)
{
    if constexpr (J < kPosDepth) {
        if ((blk & (1u << J)) == 0) {
            s.B[J] = c;
        } else {
            uint32_t n;
            csa(n, s.A[J], s.A[J], s.B[J], c);
            s.B[J] = 0;
            pos_chain_push<J + 1>(s, blk, n);
        }
    }
}

// Load schedule as K1's default (flagstat_kernels.hip, variant 71): each wave owns a contiguous 8 KiB of a step, and
// vector u's registers are re-issued -- as soon as it has been copied out -- for vector u + 6 of the same step (`cur`)
// or u - 2 of the lane's next step (`next`): 6 loads = 6 KiB per wave, 24 KiB per CU in flight, the chip's fastest
// read pattern (profiles/r03/read_probe_sweep.log, rolling_distance_sweep.log).
constexpr int kPosRD = 6;    // rolling distance in vectors
constexpr int kPosUS = 64;   // vectors between a lane's consecutive loads (wave-contiguous layout)

// 8 vectors = 32 dwords = 64 words per lane.  ROLL: copy out, re-issue into the same registers.
template <bool ROLL, bool HAS_NEXT = true>
__device__ __forceinline__ void pos_step(PosLane& s, uint4 (&v)[kUnroll], uint32_t blk, const uint4* __restrict__ next,
                                         const uint4* __restrict__ cur = nullptr)
{
    uint32_t c16[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t c8[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t c4[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int u = h * 4 + q * 2 + k;
                uint4 x = v[u];
                if constexpr (ROLL) {
                    __builtin_amdgcn_sched_barrier(0);
                    x = copy_out(x);
                    if (u + kPosRD < kUnroll)
                        v[u + kPosRD] = load_vec<true>(cur + (u + kPosRD) * kPosUS);
                    else if constexpr (HAS_NEXT)
                        v[u + kPosRD - kUnroll] = load_vec<true>(next + (u + kPosRD - kUnroll) * kPosUS);
                    __builtin_amdgcn_sched_barrier(0);
                }
                uint32_t a, b;
                csa(a, s.p1, s.p1, x.x, x.y);
                csa(b, s.p1, s.p1, x.z, x.w);
                csa(c4[k], s.p2, s.p2, a, b);
            }
            csa(c8[q], s.p4, s.p4, c4[0], c4[1]);
        }
        csa(c16[h], s.p8, s.p8, c8[0], c8[1]);
    }
    uint32_t c32;
    csa(c32, s.p16, s.p16, c16[0], c16[1]);
    pos_chain_push<0>(s, blk, c32);
}

__device__ __forceinline__ void pos_flush(PosLane& s)
{
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const uint32_t mk = 0x00010001u << c;  // bit c of both packed words
        uint32_t a = 0;
#pragma unroll
        for (int j = kPosDepth - 1; j >= 0; --j) {
            a = hstep(a, s.A[j], mk, true);
            a = hstep(a, s.B[j], mk, false);
        }
        a = hstep(a, s.p16, mk, true);
        a = hstep(a, s.p8, mk, true);
        a = hstep(a, s.p4, mk, true);
        a = hstep(a, s.p2, mk, true);
        a = hstep(a, s.p1, mk, true);
        s.acc[c] += a;
    }
    s.p1 = s.p2 = s.p4 = s.p8 = s.p16 = 0;
#pragma unroll
    for (int j = 0; j < kPosDepth; ++j) s.A[j] = s.B[j] = 0;
}

template <bool ROLL, bool HAS_NEXT = true>
__device__ __forceinline__ void pos_step_and_count(PosLane& s, uint4 (&v)[kUnroll], uint32_t& blk, const uint4* next,
                                                   const uint4* cur = nullptr)
{
    blk = __builtin_amdgcn_readfirstlane(blk);  // wave-uniform: scalar branches in the chain
    pos_step<ROLL, HAS_NEXT>(s, v, blk, next, cur);
    ++blk;
    if (blk == (1u << kPosDepth) - 1u) {
        pos_flush(s);
        blk = 0;
    }
}

__global__ __launch_bounds__(kThreads) void pospopcnt_count(const uint4* __restrict__ a0, uint64_t lo, uint64_t hi,
                                                            uint64_t nsteps, uint64_t fast_begin, uint64_t fast_end,
                                                            uint64_t* __restrict__ partials, uint64_t* __restrict__ out)
{
    PosLane s;
    s.p1 = s.p2 = s.p4 = s.p8 = s.p16 = 0;
#pragma unroll
    for (int j = 0; j < kPosDepth; ++j) s.A[j] = s.B[j] = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s.acc[c] = 0;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lane_off = static_cast<uint64_t>(wave) * (64 * kUnroll) + lane;  // vector of (wave, u, lane): wave*512 + u*64 + lane
    const uint64_t G = gridDim.x;
    uint32_t blk = 0;

    auto edge_step = [&](uint64_t st) {
        uint4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = load_guarded(a0, st * kVecPerStep + lane_off + u * kPosUS, lo, hi);
        pos_step_and_count<false>(s, v, blk, nullptr);
    };
    if (fast_begin != 0 && blockIdx.x == 0) edge_step(0);
    if (nsteps > fast_end && nsteps - 1 >= fast_begin && (nsteps - 1) % G == blockIdx.x) edge_step(nsteps - 1);

    uint64_t st = blockIdx.x;
    if (st < fast_begin) st += G;
    if (st < fast_end) {
        uint4 v[kUnroll];
        const uint4* p = a0 + st * kVecPerStep + lane_off;
#pragma unroll
        for (int u = 0; u < kPosRD; ++u) {  // the first 6 vectors; the rest is issued as they are consumed
            v[u] = load_vec<true>(p + u * kPosUS);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; st + G < fast_end; st += G) {
            const uint4* pn = p + G * kVecPerStep;
            pos_step_and_count<true, true>(s, v, blk, pn, p);
            p = pn;
        }
        pos_step_and_count<true, false>(s, v, blk, nullptr, p);
    }
    pos_flush(s);

    __shared__ uint32_t red[kThreads / 64][16];
    uint32_t wsum[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) wsum[c] = wave_sum_lane63(s.acc[c]);
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < 16; ++c) red[wave][c] = wsum[c];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        uint64_t sum = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) sum += red[w][threadIdx.x];
        if (out) {
            // direct epilogue, as K1's: this workgroup's 16 totals go to out[16] with relaxed atomics, no finalize launch
            if (sum) (void)__hip_atomic_fetch_add(&out[threadIdx.x], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            partials[static_cast<uint64_t>(threadIdx.x) * gridDim.x + blockIdx.x] = sum;  // [bit][block]
        }
    }
}

// out[16] += column sums of partials[16][nblocks]; one wave per bit position
__global__ __launch_bounds__(1024) void pospopcnt_finalize(const uint64_t* __restrict__ partials, uint32_t nblocks,
                                                           uint64_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u, c = threadIdx.x >> 6;
    uint64_t x = 0;
    for (uint32_t b = lane; b < nblocks; b += 64) x += partials[static_cast<uint64_t>(c) * nblocks + b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    if (lane == 0 && x) out[c] += x;
}

}  // namespace fsk

// d_out16[16] += positional popcounts of d_array[0..n).  Asynchronous.  d_partials: >= grid*19*8 bytes.
extern "C" hipError_t fsk_launch_pospopcnt(const uint16_t* d_array, uint64_t n, uint32_t grid, uint64_t* d_partials,
                                           uint64_t* d_out16, hipStream_t stream, int direct)
{
    if (n == 0) return hipSuccess;
    if (grid == 0 || d_array == nullptr || d_partials == nullptr || d_out16 == nullptr) return hipErrorInvalidValue;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_array);
    if (addr & 1u) return hipErrorInvalidValue;
    const uintptr_t base = addr & ~static_cast<uintptr_t>(15);
    const uint64_t lo = (addr - base) / 2, hi = lo + n;
    const uint64_t nvec = (hi + 7) / 8;
    const uint64_t nsteps = (nvec + fsk::kVecPerStep - 1) / fsk::kVecPerStep;
    const uint64_t fast_begin = (lo == 0) ? 0 : 1;
    uint64_t fast_end = (hi / 8) / fsk::kVecPerStep;
    if (fast_end < fast_begin) fast_end = fast_begin;
    if (static_cast<uint64_t>(grid) > nsteps) grid = static_cast<uint32_t>(nsteps);
    hipLaunchKernelGGL(fsk::pospopcnt_count, dim3(grid), dim3(fsk::kThreads), 0, stream,
                       reinterpret_cast<const uint4*>(base), lo, hi, nsteps, fast_begin, fast_end, d_partials,
                       direct ? d_out16 : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || direct) return e;
    hipLaunchKernelGGL(fsk::pospopcnt_finalize, dim3(1), dim3(1024), 0, stream, d_partials, grid, d_out16);
    return hipGetLastError();
}
