// flagstat_kernels.hip -- the flagstat hot path as hand-written HIP for gfx950 (CDNA4): K1, K2 and their launcher.
//
// What it replaces (reference, /root/reference): the FLAGSTAT_* kernels of libflagstats.h -- mask-select / propagate-carry
// front end (:1695-1704 and the LUT forms :1908-1935, :2474-2489), the Harley-Seal carry-save tree (:1706-1767,
// python/libalgebra.h:2311-2319), the counter flush (:1769-1840) and the scalar tail (:1649-1651) -- with FLAGSTAT_scalar's
// exact 19-counter semantics (:118-142), which of the reference's SIMD variants only _avx512_improved3 (:2321-2644) has.
// Nothing here is a translation of those x86 kernels; the design is for wave64 / VALU / HBM3E:
//
//  K1 flagstat_count   grid-stride over 32 KiB "steps" (one workgroup of 4 waves per CU); each lane loads 8 x 16 B per step
//                      (global_load_dwordx4 nt, 1 KiB per wave-instruction, fully coalesced) straight to VGPRs -- a read-once
//                      stream gains nothing from an LDS round trip -- with six of them in flight at any time: a vector's
//                      registers are re-issued for the vector six places on as soon as it has been split out (24 KiB in
//                      flight per CU is what this chip reads fastest: profiles/r03/rolling_distance_sweep.log).
//                      Per 16 B (8 flags), flagstat_count_core.h:
//                        * v_perm_b32 splits 2 dwords (4 flags) into a dword of low bytes L and one of high bytes H
//                          (byte-planar: every later op works on 4 flags at once);
//                        * three v_perm_b32 LUTs evaluate the flagstat decision tree: T = 8 counter bits per flag (any QC),
//                          F = T under fail-QC, S = one-hot(QC-only, DUP-only, both) + the primary-paired indicator by QC class;
//                        * T / F / S dwords enter bit-sliced carry-save counters: one CSA = 2 x v_bitop3_b32 (0x96, 0xE8).
//                      16 inputs per step collapse through a Harley-Seal tree; the weight-16 carry enters a binary-counter
//                      chain of (accumulator, pending) plane pairs driven by the SCALAR step count, so the amortised cost
//                      stays 1 CSA per input at any depth and the cross-lane transpose happens once per epoch of 255 steps,
//                      not per block as on x86 (:1751).  Epoch flush: carry-propagate the chain into 12 binary planes, then
//                      v_and_b32 + v_dot4_u32_u8 per (plane, counter) into 21 u32 lane counters.  Kernel end: DPP wave sums
//                      + LDS, then EITHER the workgroup adds its totals -- mapped to the reference's slots -- to out[32] with
//                      device atomics (the += contract's default: one launch; many workgroups go through 8 per-XCD copies,
//                      grouped_epilogue), OR it writes per-workgroup uint64[21] partials for K2 (store form, host counters).
//  K2 flagstat_finalize sums the partials of one launch, maps the 21 internal counters to the reference's 32 slots and
//                      stores / adds them.
//
// Zero flags contribute to no counter (KAT x=0), so ragged heads / tails and idle lanes are handled by zero-filling: no
// host-side tail, no scalar fallback.
//
// This file carries the three schedules the library ships: 71 (default: rolling re-issue at a distance of 6 vectors, each
// wave a contiguous 8 KiB of a step), 25 (the r01-r02 default: rolling over a whole step, waves interleaved at 1 KiB) and 9
// (the plain loop both are measured against).  Everything else that was ever built and measured -- the LDS-DMA ring, the
// dynamic scheduler, two waves per SIMD, the distance and grouping sweeps, the fused finalise, timeline stamps -- lives in
// flagstat_kernels_tuning.hip and is compiled by `make tuning` only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "flagstat_count_core.h"

namespace fsk {

// One step: 8 vectors of 16 B per lane = 64 flags -> 16 T, 16 F, 16 S inputs.
// STAGE 0: the vectors are in v[].
// STAGE 1 (rolling over a whole step): as soon as vector u has been split out of its registers, the same registers are
// re-issued for vector u of the lane's NEXT step (`next`, stride USTRIDE vectors), so a wave keeps 8 loads in flight through
// the whole step without a second register buffer.
// STAGE 9 (rolling at a distance of 6 vectors, the default): vector u's registers are re-issued for vector u + 6 of the same
// step (`cur`), or u - 2 of the next one (`next`, if HAS_NEXT): 6 loads = 24 KiB per CU in flight.
template <int DEPTH, int STAGE, bool NT, int USTRIDE, bool HAS_NEXT = true>
__device__ __forceinline__ void step(Lane<DEPTH>& s, uint4 (&v)[kUnroll], uint32_t blk, const uint4* __restrict__ next,
                                     const uint4* __restrict__ cur = nullptr)
{
    static_assert(STAGE == 0 || STAGE == 1 || STAGE == 9, "the product carries three schedules; the others: flagstat_kernels_tuning.hip");
    uint32_t t8a = 0, t8b = 0, f8a = 0, f8b = 0, s8a = 0, s8b = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t t4a = 0, t4b = 0, f4a = 0, f4b = 0, s4a = 0, s4b = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // two vectors -> 4 T/F/S inputs
            uint32_t T[4], F[4], S[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                uint32_t L0, H0, L1, H1;
                const int uu = half * 4 + q * 2 + k;  // a constant after unrolling
                if constexpr (STAGE == 9) {
                    constexpr int RD = 6;
                    __builtin_amdgcn_sched_barrier(0);
                    split_out(v[uu], L0, H0, L1, H1);
                    if (uu + RD < 8)
                        v[uu + RD] = load_vec<NT>(cur + (uu + RD) * USTRIDE);
                    else if constexpr (HAS_NEXT)
                        v[uu + RD - 8] = load_vec<NT>(next + (uu + RD - 8) * USTRIDE);
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (STAGE == 1) {
                    // Split the vector out of its registers HERE (a load may land at any time, so the registers it
                    // targets must be dead first), then re-issue into the same registers.  The asm keeps hipcc from
                    // turning the reads into loop-top PHI moves (which wait for all 8 loads), the sched_barriers from
                    // sinking the loads below the arithmetic.
                    __builtin_amdgcn_sched_barrier(0);
                    split_out(v[uu], L0, H0, L1, H1);
                    v[uu] = load_vec<NT>(next + uu * USTRIDE);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    const uint4 x = v[uu];
                    L0 = perm(x.y, x.x, 0x06040200u);
                    H0 = perm(x.y, x.x, 0x07050301u);
                    L1 = perm(x.w, x.z, 0x06040200u);
                    H1 = perm(x.w, x.z, 0x07050301u);
                }
                uint32_t qa, qb, ka, kb;
                front4(L0, H0, T[2 * k], qa, ka);
                front4(L1, H1, T[2 * k + 1], qb, kb);
                // fail-QC byte masks
                F[2 * k] = T[2 * k] & perm(0u, 0xFF00FF00u, qa);
                F[2 * k + 1] = T[2 * k + 1] & perm(0u, 0xFF00FF00u, qb);
                // S byte: LUT over (qcfail, dup) = one-hot {QC only, DUP only, both} in bits 0-2 plus a
                // QC-class template in bits 6 (pass) / 7 (fail), which survives only for primary paired
                // reads (bits 6,7 of the keep-mask).  lut & (keep | 0x3f) is ONE v_bitop3_b32.
                S[2 * k] = perm(0u, 0x84428140u, qa) & (ka | 0x3F3F3F3Fu);
                S[2 * k + 1] = perm(0u, 0x84428140u, qb) & (kb | 0x3F3F3F3Fu);
            }
            uint32_t t2a, t2b, f2a, f2b, s2a, s2b;
            csa(t2a, s.t1, s.t1, T[0], T[1]);
            csa(t2b, s.t1, s.t1, T[2], T[3]);
            csa(f2a, s.f1, s.f1, F[0], F[1]);
            csa(f2b, s.f1, s.f1, F[2], F[3]);
            csa(s2a, s.s1, s.s1, S[0], S[1]);
            csa(s2b, s.s1, s.s1, S[2], S[3]);
            csa(q ? t4b : t4a, s.t2, s.t2, t2a, t2b);
            csa(q ? f4b : f4a, s.f2, s.f2, f2a, f2b);
            csa(q ? s4b : s4a, s.s2, s.s2, s2a, s2b);
        }
        csa(half ? t8b : t8a, s.t4, s.t4, t4a, t4b);
        csa(half ? f8b : f8a, s.f4, s.f4, f4a, f4b);
        csa(half ? s8b : s8a, s.s4, s.s4, s4a, s4b);
    }
    uint32_t ct, cf, cs;
    csa(ct, s.t8, s.t8, t8a, t8b);  // weight-16 carries
    csa(cf, s.f8, s.f8, f8a, f8b);
    csa(cs, s.s8, s.s8, s8a, s8b);
    chain_push<0, DEPTH>(s, blk, ct, cf, cs);
}

template <int DEPTH, int STAGE = 0, bool NT = false, int USTRIDE = 64, bool HAS_NEXT = true>
__device__ __forceinline__ void step_and_count(Lane<DEPTH>& s, uint4 (&v)[kUnroll], uint32_t& blk,
                                               const uint4* __restrict__ next = nullptr, const uint4* __restrict__ cur = nullptr)
{
    // blk is the same in every lane; hipcc keeps it in a VGPR and branches through the exec mask (v_and, v_cmp,
    // s_and_saveexec per chain level) unless told so
    blk = __builtin_amdgcn_readfirstlane(blk);
    step<DEPTH, STAGE, NT, USTRIDE, HAS_NEXT>(s, v, blk, next, cur);
    ++blk;
    if (blk == (1u << DEPTH) - 1u) {
        flush(s, (1u << DEPTH) - 1u);
        blk = 0;
    }
}

// ------------------------------------------------------------------ K1
template <int DEPTH, bool NT, bool PREFETCH, bool INTERLEAVE, int STAGE>
__global__ __launch_bounds__(kThreads) void flagstat_count(const uint4* __restrict__ a0, uint64_t lo, uint64_t hi,
                                                           uint64_t nsteps, uint64_t fast_begin, uint64_t fast_end,
                                                           uint64_t* __restrict__ partials, uint32_t* ticket,
                                                           uint64_t* out, int mode, DynSched dyn, HostSignal sig)
{
    static_assert(!PREFETCH, "the two-register-buffer loop lives in flagstat_kernels_tuning.hip");
    Lane<DEPTH> s;
    lane_init(s);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // vector of (wave, u, lane) within a step: wave*512 + u*64 + lane, or u*256 + wave*64 + lane
    constexpr int T = kThreads;              // worker threads of a workgroup
    constexpr int VPS = T * kUnroll;         // vectors per step
    constexpr int US = INTERLEAVE ? T : 64;
    const uint64_t lane_off = INTERLEAVE ? static_cast<uint64_t>(threadIdx.x)
                                         : static_cast<uint64_t>(wave) * (64 * kUnroll) + lane;
    const uint64_t G = gridDim.x;
    // Steps pushed in the current epoch.  An epoch ends at 255 with a flush (~1 us of pure VALU work); started at 0 in
    // every wave, all 1024 waves of the chip would flush at the same step and HBM would idle meanwhile.  Mode bit 4
    // starts wave w of a workgroup at 64 * w: its first epoch is that much shorter, so at any time at most one wave of
    // a CU is flushing while the other three keep their loads in flight.  (Any start is arithmetically fine: the chain
    // levels are adders; a level whose pending plane is empty while its bit of blk is set just adds a zero.)
    uint32_t blk = (mode & 16) ? (wave & 3u) * 64u : 0u;

    constexpr bool ROLL = (STAGE != 0);
    if constexpr (ROLL) {
        // ragged edge steps (at most the first and the last of the whole array) go through the
        // guarded loader, outside the pipelined loop
        if (fast_begin != 0 && blockIdx.x == 0 && wave < T / 64) {
            uint4 v[kUnroll];
            load_step<NT, US, VPS>(v, a0, 0, lane_off, lo, hi, fast_begin, fast_end);
            step_and_count(s, v, blk);
        }
        if (nsteps > fast_end && nsteps - 1 >= fast_begin && (nsteps - 1) % G == blockIdx.x && wave < T / 64) {
            uint4 v[kUnroll];
            load_step<NT, US, VPS>(v, a0, nsteps - 1, lane_off, lo, hi, fast_begin, fast_end);
            step_and_count(s, v, blk);
        }
        // first fully in-range step of this workgroup
        uint64_t st = blockIdx.x;
        if (st < fast_begin) st += G;  // fast_begin is 0 or 1
        if constexpr (STAGE == 1) {
            if (st < fast_end) {
                uint4 v[kUnroll];
                const uint4* p = a0 + st * kVecPerStep + lane_off;
                // issue order = consumption order, so the loop-top wait can be vmcnt(7), not vmcnt(0)
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    v[u] = load_vec<NT>(p + u * US);
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (; st + G < fast_end; st += G) {
                    p += G * kVecPerStep;
                    step_and_count<DEPTH, STAGE, NT, US>(s, v, blk, p);
                }
                step_and_count(s, v, blk);
            }
        } else {
            constexpr int RD = 6;
            if (st < fast_end) {
                uint4 v[kUnroll];
                const uint4* p = a0 + st * VPS + lane_off;
#pragma unroll
                for (int u = 0; u < RD; ++u) {  // the first RD vectors; the rest is issued as they are consumed
                    v[u] = load_vec<NT>(p + u * US);
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (; st + G < fast_end; st += G) {
                    const uint4* pn = p + G * VPS;
                    step_and_count<DEPTH, STAGE, NT, US, true>(s, v, blk, pn, p);
                    p = pn;
                }
                step_and_count<DEPTH, STAGE, NT, US, false>(s, v, blk, nullptr, p);
            }
        }
    } else {
        for (uint64_t st = blockIdx.x; st < nsteps; st += G) {
            uint4 v[kUnroll];
            load_step<NT, US>(v, a0, st, lane_off, lo, hi, fast_begin, fast_end);
            step_and_count(s, v, blk);
        }
    }
    flush(s, blk);

    // wave sums on the VALU (DPP), then the 4 waves through LDS
    constexpr int kWaves = kThreads / 64;
    __shared__ uint32_t red[kWaves][kInternal];
    uint32_t wsum[kInternal];
#pragma unroll
    for (int c = 0; c < kInternal; ++c) wsum[c] = wave_sum_lane63(s.acc[c]);
    if (lane == 63 && wave < kWaves) {
#pragma unroll
        for (int c = 0; c < kInternal; ++c) red[wave][c] = wsum[c];
    }
    __syncthreads();
    uint64_t sum = 0;
    if (threadIdx.x < kInternal) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sum += red[w][threadIdx.x];
    }
    if (mode & 32) {
        // Latency form (a grid of ONE workgroup, result pairs in pinned host memory): this workgroup's totals ARE the
        // result, so it stores all 32 slots itself ("=" form) -- no partials, no K2 launch -- each with the call's
        // sequence number for the host thread polling them.
        __shared__ uint64_t one_tot[32];
        if (threadIdx.x < kInternal) one_tot[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x < 32) store_pair(sig, slot_value(one_tot, mode, hi - lo));
        return;
    }
    if (mode & 4) {
        // Direct epilogue (accumulate contract only): this workgroup maps ITS 21 totals to the
        // reference's slots and adds them to out[32] with relaxed agent-scope atomics (no return, no
        // fence, no ticket, no K2 launch).  Integer sums are exact in any order; pass-QC = T - F holds
        // per workgroup because F counts a subset of T.  Kernel end is the only ordering anyone needs.
        __shared__ uint64_t wg_tot[32];
        if (threadIdx.x < kInternal) wg_tot[threadIdx.x] = sum;
        __syncthreads();
        if (mode & 8)
            grouped_epilogue(wg_tot, out, mode, blockIdx.x == 0 ? hi - lo : 0, dyn.block);
        else
            finalize_slots<true>(wg_tot, out, mode, blockIdx.x == 0 ? hi - lo : 0);
        return;
    }
    if (threadIdx.x < kInternal) {
        // [counter][block]; write-through (sc1) so the finalising workgroup -- on whichever XCD -- sees it
        __hip_atomic_store(&partials[static_cast<uint64_t>(threadIdx.x) * gridDim.x + blockIdx.x], sum, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    (void)ticket;  // two-kernel form: K2 (flagstat_finalize) sums the partials
}

// ------------------------------------------------------------------ K2
// One workgroup of 16 waves: wave w sums column w (and w+16) of partials[21][nblocks]
// with coalesced 8-byte loads, then 32 threads map the 21 internal counters to the
// reference's 32 slots (index = FLAGSTAT_*_OFF, libflagstats.h:69-112; +16 for fail-QC)
// and ADD into out[32].  Slots the scalar rule never writes get nothing added.
constexpr int kFinalizeThreads = 1024;

__global__ __launch_bounds__(kFinalizeThreads) void flagstat_finalize(const uint64_t* __restrict__ partials,
                                                                      uint32_t nblocks, uint64_t* __restrict__ out,
                                                                      int mode, uint64_t n_flags, HostSignal sig)
{
    __shared__ uint64_t tot[32];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t c = wave; c < kInternal; c += kFinalizeThreads / 64) {
        uint64_t x = 0;
        for (uint32_t b = lane; b < nblocks; b += 64) x += partials[static_cast<uint64_t>(c) * nblocks + b];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        if (lane == 0) tot[c] = x;
    }
    __syncthreads();
    if (sig.pairs != nullptr) {
        if (threadIdx.x < 32) store_pair(sig, slot_value(tot, mode, n_flags));  // "=" form for a polling host thread
        return;
    }
    finalize_slots(tot, out, mode, n_flags);
}

}  // namespace fsk

// ------------------------------------------------------------------ launchers
// The measurement build (`make tuning`) links flagstat_kernels_tuning.hip, which defines these; the product library does not,
// and the weak references stay null.
extern "C" hipError_t fsk_tuning_launch(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, uint64_t* d_partials,
                                        uint32_t* d_ticket, uint64_t* d_out32, hipStream_t stream, uint64_t* signal_word,
                                        uint64_t signal_value) __attribute__((weak));
extern "C" int fsk_tuning_variant_supported(int variant) __attribute__((weak));

// partials[21][grid] followed by an 8 KiB block (u64 words): kGroupTicketWord + 16 * g the ticket of epilogue group g,
// kGroupCopyWord + 32 * g that group's copy of the 32 slots (the words below 288 belong to the measurement build's schedules).
// The block must be zero before the first launch and is left zero by every launch.
// Loads this translation unit's code object (K1, K2) without launching anything: the first launch of a process otherwise pays
// for it in line.  The GPU decoders' first call runs it on its helper thread (flagstat_gpu_decode.hip).
extern "C" void fsk_warm(void)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&fsk::flagstat_finalize)) != hipSuccess) (void)hipGetLastError();
}

extern "C" size_t fsk_partials_bytes(uint32_t grid) { return static_cast<size_t>(grid) * fsk::kInternal * sizeof(uint64_t) + 8192; }

static std::atomic<int> g_epoch_stagger{1};          // K1 mode bit 4: waves of a workgroup end their epochs at different steps

extern "C" void fsk_set_epoch_stagger(int on) { g_epoch_stagger = on ? 1 : 0; }

// The two-level (per-XCD copies) epilogue pays where the workgroups of a launch finish together: many of them, few steps
// each.  With more steps per workgroup the finish times drift apart by more than the adds take and the one-level form is
// 0.5-0.8 % faster (profiles/r04/ab_two_level_epilogue.log, ab_two_level_epilogue_crossover.log).
static std::atomic<uint32_t> g_group_min_grid{64};   // grids below this add straight to out[] (one level)
static std::atomic<uint64_t> g_group_max_steps{40};  // ... and so do launches with more steps per workgroup (40: <= 320 MiB on 256 CUs;
                                                     // the forms cross at 48, profiles/r04/ab_two_level_epilogue_crossover.log)

extern "C" void fsk_set_group_min_grid(uint32_t min_grid) { g_group_min_grid = min_grid; }
extern "C" void fsk_set_group_max_steps(uint64_t max_steps) { g_group_max_steps = max_steps; }

static std::atomic<int> g_last_mode{0};  // K1 mode word of the most recent launch (tests: which epilogue form ran)
extern "C" int fsk_last_mode(void) { return g_last_mode.load(); }

extern "C" void fsk_launch_policy(int* stagger, uint32_t* group_min_grid, uint64_t* group_max_steps)
{
    *stagger = g_epoch_stagger.load();
    *group_min_grid = g_group_min_grid.load();
    *group_max_steps = g_group_max_steps.load();
}
extern "C" void fsk_note_mode(int mode) { g_last_mode.store(mode, std::memory_order_relaxed); }

extern "C" hipError_t fsk_launch_finalize(uint64_t* d_partials, uint32_t grid, uint64_t* d_out32, int mode, uint64_t n, fsk::HostSignal sig,
                                          hipStream_t stream)
{
    hipLaunchKernelGGL(fsk::flagstat_finalize, dim3(1), dim3(fsk::kFinalizeThreads), 0, stream, d_partials, grid, d_out32, mode, n, sig);
    return hipGetLastError();
}

template <int DEPTH, bool NT, bool PREFETCH, bool INTERLEAVE, int STAGE = 0>
static hipError_t launch_count_t(const fsk::CountArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL((fsk::flagstat_count<DEPTH, NT, PREFETCH, INTERLEAVE, STAGE>), dim3(a.grid), dim3(fsk::kThreads), 0, stream,
                       reinterpret_cast<const uint4*>(a.a0), a.lo, a.hi, a.nsteps, a.fast_begin, a.fast_end, a.partials,
                       a.ticket, a.out, a.mode, a.dyn, a.sig);
    return hipGetLastError();
}

// Host-side geometry: everything the kernel assumes is derived here from
// (pointer, n) so operand shapes and the grid cannot disagree.
extern "C" hipError_t fsk_launch(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, uint64_t* d_partials,
                                 uint32_t* d_ticket, uint64_t* d_out32, hipStream_t stream, uint64_t* signal_word,
                                 uint64_t signal_value)
{
    if (fsk_tuning_launch)  // measurement build: its launcher takes every launch (instruments on the default schedule too)
        return fsk_tuning_launch(d_array, n, grid, variant, d_partials, d_ticket, d_out32, stream, signal_word, signal_value);
    if (n == 0) return hipSuccess;
    if (grid == 0 || d_array == nullptr || d_partials == nullptr || d_out32 == nullptr) return hipErrorInvalidValue;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_array);
    if (addr & 1u) return hipErrorInvalidValue;  // uint16_t* must be 2-byte aligned
    fsk::CountArgs a;
    const uintptr_t base = addr & ~static_cast<uintptr_t>(15);
    a.a0 = reinterpret_cast<const void*>(base);
    a.lo = (addr - base) / 2;
    a.hi = a.lo + n;
    const uint64_t nvec = (a.hi + 7) / 8;
    const uint64_t vps = fsk::kVecPerStep;
    a.nsteps = (nvec + vps - 1) / vps;
    // steps whose vectors are all fully inside [lo, hi)
    a.fast_begin = (a.lo == 0) ? 0 : 1;
    a.fast_end = (a.hi / 8) / vps;
    if (a.fast_end < a.fast_begin) a.fast_end = a.fast_begin;
    if (static_cast<uint64_t>(grid) > a.nsteps) grid = static_cast<uint32_t>(a.nsteps);
    a.grid = grid;
    a.partials = d_partials;
    // bit 8: store instead of accumulate; bit 10: superset slots; bit 11: direct (atomic) epilogue, no K2
    a.mode = ((variant >> 8) & 1) | (((variant >> 10) & 1) << 1) | (((variant >> 11) & 1) << 2);
    if ((a.mode & 4) && (a.mode & 1)) return hipErrorInvalidValue;  // accumulate form only
    if ((variant >> 9) & 1) return hipErrorInvalidValue;  // the ticket-fused finalise exists in the measurement build only
    a.ticket = nullptr;
    a.out = d_out32;
    a.dyn = fsk::DynSched{reinterpret_cast<uint64_t*>(d_ticket), 0xFFFFFFFFu, 0, 1, 0};  // (.block = the workspace's block: grouped_epilogue)
    if (g_epoch_stagger.load()) a.mode |= 16;
    // store form with a completion word: a grid of one workgroup stores its totals itself (K1's latency form, no K2);
    // larger grids go through K2, which signals after its stores
    a.sig = fsk::HostSignal{signal_word, signal_value};
    if (signal_word && !(a.mode & 1)) return hipErrorInvalidValue;  // pairs carry "=" results only
    if (signal_word && (a.mode & 1) && !(a.mode & 4) && grid == 1) a.mode |= 32;
    // direct epilogue: many workgroups add to per-XCD copies first (grouped_epilogue); few add straight to out[]
    if ((a.mode & 4) && grid >= g_group_min_grid.load() && (a.nsteps + grid - 1) / grid <= g_group_max_steps.load() && d_ticket != nullptr)
        a.mode |= 8;
    g_last_mode.store(a.mode, std::memory_order_relaxed);
    hipError_t e;
    // variant bits 0-7 name the K1 schedule (bit 0 non-temporal loads, bit 3 waves interleaved at 1 KiB within a step, bit 4
    // rolling re-issue of the load registers; 71 is a plain label)
    switch (variant & 255) {
    case 9: e = launch_count_t<8, true, false, true>(a, stream); break;        // plain loop
    case 25: e = launch_count_t<8, true, false, true, 1>(a, stream); break;    // r01-r02 default: rolling over a whole step
    // default since r03: rolling at a distance of 6 vectors (24 KiB in flight per CU instead of 32), each wave a
    // contiguous 8 KiB of the step -- +2.3-2.8 % at 8 GiB, +4 % at 1 GiB (profiles/r03/rolling_distance_sweep.log)
    case 71: e = launch_count_t<8, true, false, false, 9>(a, stream); break;
    default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    if (a.mode & (4 | 32)) return hipSuccess;  // K1 finalised by itself
    return fsk_launch_finalize(d_partials, grid, d_out32, a.mode, n, a.sig, stream);
}

// which K1 schedules this build of the library carries (bits 0-7 of `variant`)
extern "C" int fsk_variant_supported(int variant)
{
    switch (variant & 255) {
    case 9:
    case 25:
    case 71: return 1;
    default: return fsk_tuning_variant_supported ? fsk_tuning_variant_supported(variant) : 0;
    }
}

extern "C" int fsk_tuning_build(void) { return fsk_tuning_launch ? 1 : 0; }
