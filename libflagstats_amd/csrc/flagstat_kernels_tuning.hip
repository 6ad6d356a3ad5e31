// flagstat_kernels_tuning.hip -- MEASUREMENT BUILD ONLY (`make tuning` -> libflagstats_hip_tuning.so; not part of
// libflagstats_hip.so).  Every K1 schedule that was built, measured and lost a sweep, kept runnable as evidence:
//   0 / 1 / 13 / 27  plain loop without / with non-temporal loads, two register buffers, chain depth 7       (r01, tools/tune.py)
//   41               staging through a per-wave LDS ring filled by LDS-DMA (north_star's "stage into LDS")   (r01, r03)
//   89               rolling re-issue at distance 2 (two register buffers, 64 KiB in flight per CU)           (r02)
//   153              guided self-scheduling through device counters, grabbed by a fifth wave                  (r03, tools/dyn_sweep.py)
//   29               two waves per SIMD, half a step in flight each                                           (r03)
//   17 / 57 / 61 / 63 / 65 / 67 / 69 / 73 / 75 / 77 / 79 / 81   the rolling-distance and re-issue-grouping sweeps (r03)
// plus the three shipped schedules (9, 25, 71) with the instruments the product kernel does not carry: workgroup timeline
// stamps (tools/timeline.py), launch-anatomy switches that skip parts of the kernel (tools/launch_anatomy.py), the 8-copy
// epilogue experiment and the ticket-fused finalise (r01).  When this file is linked, fsk_launch (flagstat_kernels.hip) hands
// EVERY launch to fsk_tuning_launch below, so the instruments cover the default schedule too; the kernels here are
// fskt::flagstat_count<...>, never the product's fsk::flagstat_count.  The arithmetic -- front end, chain, flush, slot map,
// epilogues -- is the product's own (flagstat_count_core.h); only the way a step's vectors are loaded differs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "flagstat_count_core.h"

namespace fskt {
using namespace fsk;

// One step: 8 vectors of 16 B per lane = 64 flags -> 16 T, 16 F, 16 S inputs.
// ROLL: as soon as vector u has been copied out of its registers, the same registers are
// re-issued for vector u of the lane's NEXT step (`next`, stride USTRIDE vectors), so a wave
// keeps ~8 loads in flight through the whole step without a second register buffer.
// LDS staging (STAGE == 2): where a wave's ring of 16 x 1 KiB slots lives
struct LdsStage {
    const uint4* lane;   // this lane's 16 bytes of slot 0 of the wave's ring (slot i: lane + i*64)
    uint32_t slot0;      // LDS byte address of the first slot this step reads (wave-uniform)
    uint32_t first;      // index of that slot in the ring (0 or 8)
};

// one 1 KiB LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [m0, m0 + 1 KiB)
template <bool NT>
__device__ __forceinline__ void lds_dma16(const uint4* gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// STAGE 0: vectors are in v[].  1 (rolling registers): see below.  2 (LDS ring, north_star's
// "stage into LDS"): the vector is read from the wave's LDS slot (ds_read_b128) once the LDS-DMA
// that filled it has landed -- 15 younger DMAs are always in flight behind it, hence vmcnt(15) --
// and the slot is immediately re-targeted by the DMA for the lane's step after next.
// STAGE 5 (the two-waves-per-SIMD experiment: 8 waves per workgroup, each with HALF a step in flight): vector u's
// registers are re-issued for vector (u + 4) & 7 -- the second half of THIS step (`cur`) for u < 4, the first half of
// the lane's next step (`next`, if HAS_NEXT) for u >= 4 -- so a wave keeps 4 loads (4 KiB) in flight and the 8 waves of
// a CU together the same 32 KiB as the 4 waves of the default schedule.
template <int DEPTH, int STAGE, bool NT, int USTRIDE, bool HAS_NEXT = true>
__device__ __forceinline__ void step(Lane<DEPTH>& s, uint4 (&v)[kUnroll], uint32_t blk, const uint4* __restrict__ next,
                                     LdsStage lds = LdsStage{nullptr, 0, 0}, const uint4* __restrict__ cur = nullptr)
{
    constexpr bool ROLL = (STAGE == 1 || STAGE == 6 || STAGE == 7 || STAGE == 8);
    // measurement only (tuning variants 61 / 63): re-issue the loads in groups of RG instead of one by one -- RG vectors
    // are split out of their registers, then their RG loads go out back to back
    constexpr int RG = STAGE == 7 ? 2 : (STAGE == 8 ? 4 : 1);
    uint32_t PL[kUnroll][4];
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
                if constexpr (STAGE == 2) {
                    const int u = half * 4 + q * 2 + k;  // a constant after unrolling
                    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                    const uint4 x = lds.lane[(lds.first + u) * 64];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot has been read: it may be refilled
                    lds_dma16<NT>(next + u * USTRIDE, lds.slot0 + u * 1024);
                    L0 = perm(x.y, x.x, 0x06040200u);
                    H0 = perm(x.y, x.x, 0x07050301u);
                    L1 = perm(x.w, x.z, 0x06040200u);
                    H1 = perm(x.w, x.z, 0x07050301u);
                } else if constexpr (STAGE == 5 || STAGE == 9 || STAGE == 10 || STAGE == 11 || STAGE == 12) {
                    // rolling at a distance of RD < 8 vectors: vector u's registers are re-issued for vector u + RD of
                    // the same step, or u + RD - 8 of the next one (STAGE 5: RD = 4 with 8 waves; 9 / 10 / 11:
                    // measurement only, RD = 6 / 7 / 5 with 4 waves = 24 / 28 / 20 KiB in flight per CU)
                    constexpr int RD = STAGE == 5 ? 4 : ((STAGE == 9 || STAGE == 12) ? 6 : (STAGE == 10 ? 7 : 5));
                    const int uu = half * 4 + q * 2 + k;  // a constant after unrolling
                    __builtin_amdgcn_sched_barrier(0);
                    split_out(v[uu], L0, H0, L1, H1);
                    if (uu + RD < 8)
                        v[uu + RD] = load_vec<NT>(cur + (uu + RD) * USTRIDE);
                    else if constexpr (HAS_NEXT)
                        v[uu + RD - 8] = load_vec<NT>(next + (uu + RD - 8) * USTRIDE);
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (ROLL) {
                    // Split the vector out of its registers HERE (a load may land at any time, so the registers it
                    // targets must be dead first), then re-issue into the same registers.  The asm keeps hipcc from
                    // turning the reads into loop-top PHI moves (which wait for all 8 loads), the sched_barriers from
                    // sinking the loads below the arithmetic.
                    const int uu = half * 4 + q * 2 + k;  // a constant after unrolling
                    if (uu % RG == 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < RG; ++r) split_out(v[uu + r], PL[uu + r][0], PL[uu + r][1], PL[uu + r][2], PL[uu + r][3]);
#pragma unroll
                        for (int r = 0; r < RG; ++r) v[uu + r] = load_vec<NT>(next + (uu + r) * USTRIDE);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    L0 = PL[uu][0];
                    H0 = PL[uu][1];
                    L1 = PL[uu][2];
                    H1 = PL[uu][3];
                } else {
                    const uint4 x = v[half * 4 + q * 2 + k];
                    L0 = perm(x.y, x.x, 0x06040200u);
                    H0 = perm(x.y, x.x, 0x07050301u);
                    L1 = perm(x.w, x.z, 0x06040200u);
                    H1 = perm(x.w, x.z, 0x07050301u);
                }
                if constexpr (STAGE == 6 || STAGE == 12) {
                    // measurement only (tuning variants 57 / 73): K1's exact load schedule with the arithmetic reduced to
                    // one XOR per dword -- what the rolling re-issue reads when the VALU does nothing else
                    T[2 * k] = L0 ^ H0;
                    T[2 * k + 1] = L1 ^ H1;
                    F[2 * k] = F[2 * k + 1] = S[2 * k] = S[2 * k + 1] = 0;
                    continue;
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
            if constexpr (STAGE == 6 || STAGE == 12) {
                s.t1 ^= T[0] ^ T[1] ^ T[2] ^ T[3];
                continue;
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
    if constexpr (STAGE == 6 || STAGE == 12) return;
    uint32_t ct, cf, cs;
    csa(ct, s.t8, s.t8, t8a, t8b);  // weight-16 carries
    csa(cf, s.f8, s.f8, f8a, f8b);
    csa(cs, s.s8, s.s8, s8a, s8b);
    chain_push<0, DEPTH>(s, blk, ct, cf, cs);
}

template <int DEPTH, int STAGE = 0, bool NT = false, int USTRIDE = 64, bool HAS_NEXT = true>
__device__ __forceinline__ void step_and_count(Lane<DEPTH>& s, uint4 (&v)[kUnroll], uint32_t& blk,
                                               const uint4* __restrict__ next = nullptr, LdsStage lds = LdsStage{nullptr, 0, 0},
                                               const uint4* __restrict__ cur = nullptr)
{
    // blk is the same in every lane; hipcc keeps it in a VGPR and branches through the exec mask (v_and, v_cmp,
    // s_and_saveexec per chain level) unless told so
    blk = __builtin_amdgcn_readfirstlane(blk);
    step<DEPTH, STAGE, NT, USTRIDE, HAS_NEXT>(s, v, blk, next, lds, cur);
    ++blk;
    if (blk == (1u << DEPTH) - 1u) {
        flush(s, (1u << DEPTH) - 1u);
        blk = 0;
    }
}

// measurement only (tuning variants 79 / 81): the distance-RD rolling loop as a function, so that the waves of a workgroup
// can run different distances (22 or 26 KiB in flight per CU)
template <int DEPTH, int RSTAGE, bool NT, int US, int VPS>
__device__ __forceinline__ void roll_partial(Lane<DEPTH>& s, uint32_t& blk, const uint4* __restrict__ a0, uint64_t st, uint64_t G,
                                             uint64_t fast_end, uint64_t lane_off)
{
    constexpr int RD = RSTAGE == 9 ? 6 : (RSTAGE == 10 ? 7 : 5);
    if (st >= fast_end) return;
    uint4 v[kUnroll];
    const uint4* p = a0 + st * VPS + lane_off;
#pragma unroll
    for (int u = 0; u < RD; ++u) {
        v[u] = load_vec<NT>(p + u * US);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; st + G < fast_end; st += G) {
        const uint4* pn = p + G * VPS;
        step_and_count<DEPTH, RSTAGE, NT, US, true>(s, v, blk, pn, LdsStage{nullptr, 0, 0}, p);
        p = pn;
    }
    step_and_count<DEPTH, RSTAGE, NT, US, false>(s, v, blk, nullptr, LdsStage{nullptr, 0, 0}, p);
}

// PREFETCH = false: load 8 x 16 B, wait, compute; latency is hidden by the other
// waves of the SIMD only.  PREFETCH = true: two register buffers, the loads of
// step k+1 are in flight while step k is computed (one more 8 KiB per wave in
// flight, +32 VGPRs).
template <int DEPTH, bool NT, bool PREFETCH, bool INTERLEAVE, int STAGE>
__global__ __launch_bounds__(STAGE == 4 ? kThreads + 64 : (STAGE == 5 ? 2 * kThreads : kThreads)) void flagstat_count(const uint4* __restrict__ a0, uint64_t lo, uint64_t hi,
                                                           uint64_t nsteps, uint64_t fast_begin, uint64_t fast_end,
                                                           uint64_t* __restrict__ partials, uint32_t* ticket,
                                                           uint64_t* out, int mode, DynSched dyn, HostSignal sig)
{
    // workgroup timeline (tools/timeline.py; mode bit 11): wave 0 stamps the 100 MHz wall clock at entry, after its
    // first step, after its last step, after the final flush, after the workgroup reduction and at exit
    uint64_t tl[6] = {0, 0, 0, 0, 0, 0};
#define FSK_TL(i) do { if (mode & 2048) tl[i] = wall_clock64(); } while (0)
#define FSK_TL_ONCE(i) do { if ((mode & 2048) && tl[i] == 0) tl[i] = wall_clock64(); } while (0)
    FSK_TL(0);
    Lane<DEPTH> s;
    lane_init(s);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // vector of (wave, u, lane) within a step: wave*512 + u*64 + lane, or u*256 + wave*64 + lane
    constexpr int T = (STAGE == 5) ? 2 * kThreads : kThreads;  // worker threads of a workgroup
    constexpr int VPS = T * kUnroll;                            // vectors per step
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
        if constexpr (STAGE == 1 || STAGE == 6 || STAGE == 7 || STAGE == 8) {
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
                    FSK_TL_ONCE(1);
                }
                step_and_count(s, v, blk);
                FSK_TL_ONCE(1);
            }
        } else if constexpr (STAGE == 5 || STAGE == 9 || STAGE == 10 || STAGE == 11 || STAGE == 12) {
            constexpr int RD = STAGE == 5 ? 4 : ((STAGE == 9 || STAGE == 12) ? 6 : (STAGE == 10 ? 7 : 5));
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
                    step_and_count<DEPTH, STAGE, NT, US, true>(s, v, blk, pn, LdsStage{nullptr, 0, 0}, p);
                    FSK_TL_ONCE(1);
                    p = pn;
                }
                step_and_count<DEPTH, STAGE, NT, US, false>(s, v, blk, nullptr, LdsStage{nullptr, 0, 0}, p);
                FSK_TL_ONCE(1);
            }
        } else if constexpr (STAGE == 13 || STAGE == 14) {
            // waves 0 and 2 at distance 6, waves 1 and 3 at 7 (13: 26 KiB per CU) or 5 (14: 22 KiB per CU)
            if (wave & 1u)
                roll_partial<DEPTH, STAGE == 13 ? 10 : 11, NT, US, VPS>(s, blk, a0, st, G, fast_end, lane_off);
            else
                roll_partial<DEPTH, 9, NT, US, VPS>(s, blk, a0, st, G, fast_end, lane_off);
        } else if constexpr (STAGE == 4) {
            // Rolling registers + GUIDED SELF-SCHEDULING of the fully in-range steps (q-space [0, N), step = fast_begin + q).
            // Why: the XCDs do not read HBM equally fast, and which one is slow changes from launch to launch
            // (tools/timeline.py, profiles/r03/timeline_*.log: with the static grid-stride split the last workgroup
            // finishes 3-11 % after the median one while the early finishers' share of the bandwidth goes unused).
            // Round 0 is static and grid-stride like STAGE 1 (workgroup b: q = b, b+G, ... c0 steps, no atomics, no
            // barrier); the rest is handed out in contiguous chunks through ONE device counter, chunk size =
            // remaining / (G * div) clamped to [1, cmax], so the grabs get finer towards the end and every
            // workgroup stops within about one step of the others.
            // The grabs are made by a FIFTH wave (threads 256..319) that does nothing else: a returning atomic in a
            // worker wave either drains that wave's 8 loads in flight (the compiler waits for the result with
            // vmcnt(0), at the loop header of every step once the value is carried around the loop) or, issued behind
            // the compiler's back, has its result register copied while the atomic is still in flight (both seen in
            // the ISA).  The scheduler wave stays two chunks ahead: slot k & 3 of an LDS ring holds chunk k (start,
            // length), written before barrier k-1; at a chunk boundary every wave of the workgroup meets in ONE
            // s_barrier (LDS-only wait on the worker side: their loads in flight are not drained) and the workers
            // read slot k.  The counters reset themselves: the scheduler retires with an add to a "retired" word
            // after its last grab has returned, and the last workgroup to retire zeroes every word (launches sharing
            // a workspace are stream-ordered).
            const uint64_t N = fast_end - fast_begin;
            const uint64_t c0 = dyn.c0;
            const uint64_t D0 = G * c0;
            const bool dynamic = N > D0;  // grid-uniform
            __shared__ uint64_t tix[4];
            if (wave == kThreads / 64) {
                if (dynamic) {
                    // The dynamic region [D0, N) is cut into Q = 2^lgq equal queues, each with its own counter on its
                    // own cache line: same-address atomics retire at ~10 ns each, and 256 workgroups asking once per
                    // step (one step of the whole chip = 4.6 ns) would be bound by that.  Workgroup b is served by
                    // queue (b / 8) % Q: blockIdx.x % 8 is the XCD, so every queue serves the same mix of fast and
                    // slow XCDs, the queues drain at the same rate, and a workgroup whose queue is empty is done.
                    const uint32_t lgq = dyn.lgq;
                    const uint64_t M = N - D0;
                    const uint32_t qi = (blockIdx.x >> 3) & ((1u << lgq) - 1u);
                    const uint64_t qb = D0 + ((M * qi) >> lgq), qe = D0 + ((M * (qi + 1)) >> lgq);
                    uint64_t* ctr = dyn.block + 16 * (qi + 1);
                    uint64_t start[4] = {0, 0, 0, 0};  // what slot k & 3 holds (uniform: read back through readfirstlane)
                    uint64_t pos = qb;                 // latest known position of the queue
                    auto grab = [&](uint32_t k) {
                        const uint64_t rem = pos < qe ? qe - pos : 0;
                        const uint32_t r32 = rem > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<uint32_t>(rem);
                        uint32_t c = __umulhi(r32, dyn.inv);  // rem / (workgroups per queue * div)
                        c = c < 1u ? 1u : c;
                        c = c > dyn.cmax ? dyn.cmax : c;
                        uint64_t r = 0;
                        if (lane == 0) r = __hip_atomic_fetch_add(ctr, static_cast<uint64_t>(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t rl = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(r));
                        const uint32_t rh = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(r >> 32));
                        uint64_t st = qb + ((static_cast<uint64_t>(rh) << 32) | rl);
                        uint64_t len = c;
                        if (st >= qe) {
                            st = N;  // the queue is empty: tells the workers (and this wave) to stop
                            len = 0;
                        } else if (len > qe - st) {
                            len = qe - st;
                        }
                        if (lane == 0) tix[k & 3u] = st | (len << 48);
                        start[k & 3u] = st;
                        pos = st + c;
                    };
                    grab(1);
                    grab(2);
                    for (uint32_t k = 1;; ++k) {
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier k: slot k is readable
                        if (start[k & 3u] >= N) break;
                        grab(k + 2);
                    }
                    if (lane == 0) {
                        const uint64_t d = __hip_atomic_fetch_add(&dyn.block[8], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (d == G - 1) {
                            for (uint32_t i = 0; i < (1u << lgq); ++i)
                                __hip_atomic_store(&dyn.block[16 * (i + 1)], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&dyn.block[8], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            } else {
                uint64_t q = blockIdx.x;
                uint64_t left = q < N ? (N - q + G - 1) / G : 0;  // steps of round 0 for this workgroup
                if (left > c0) left = c0;
                uint64_t stride = G;
                if (left) {
                    uint4 v[kUnroll];
                    const uint4* p = a0 + (fast_begin + q) * kVecPerStep + lane_off;
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        v[u] = load_vec<NT>(p + u * US);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    uint32_t k = 1;
                    for (;;) {
                        uint64_t nq;
                        if (left > 1) {
                            nq = q + stride;
                            --left;
                        } else {
                            if (!dynamic) break;
                            asm volatile("s_barrier" ::: "memory");  // barrier k
                            const uint64_t t = tix[k & 3u];
                            ++k;
                            const uint32_t tl_ = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(t));
                            const uint32_t th_ = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(t >> 32));
                            const uint64_t ns = (static_cast<uint64_t>(th_ & 0xFFFFu) << 32) | tl_;
                            if (ns >= N) break;
                            nq = ns;
                            left = th_ >> 16;
                            if (left > N - ns) left = N - ns;
                            stride = 1;
                        }
                        p = a0 + (fast_begin + nq) * kVecPerStep + lane_off;
                        step_and_count<DEPTH, 1, NT, US>(s, v, blk, p);
                        FSK_TL_ONCE(1);
                        q = nq;
                    }
                    step_and_count(s, v, blk);
                    FSK_TL_ONCE(1);
                }
            }
        } else if constexpr (STAGE == 3) {
            // Rolling at distance 2: two register buffers, each vector's re-issue targets the lane's
            // step AFTER NEXT, so 16 loads (16 KiB per wave, 64 KiB per CU at one workgroup per CU) are
            // in flight at all times -- the deepest configuration of the read-probe sweep.
            if (st < fast_end) {
                uint4 va[kUnroll], vb[kUnroll];
                auto ptr = [&](uint64_t sx) { return a0 + sx * kVecPerStep + lane_off; };
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    va[u] = load_vec<NT>(ptr(st) + u * US);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (st + G >= fast_end) {
                    step_and_count(s, va, blk);  // a single step for this workgroup
                } else {
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        vb[u] = load_vec<NT>(ptr(st + G) + u * US);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // steady state: buffer A holds step st, B holds st+G, both re-issue two steps ahead;
                    // exactly 16 loads are outstanding at every wait (vmcnt(15)), on every path
                    for (; st + 3 * G < fast_end; st += 2 * G) {
                        step_and_count<DEPTH, 1, NT, US>(s, va, blk, ptr(st + 2 * G));
                        step_and_count<DEPTH, 1, NT, US>(s, vb, blk, ptr(st + 3 * G));
                    }
                    // tail: 2 or 3 steps left (st, st+G and perhaps st+2G)
                    if (st + 2 * G < fast_end) {
                        step_and_count<DEPTH, 1, NT, US>(s, va, blk, ptr(st + 2 * G));
                        step_and_count(s, vb, blk);
                        step_and_count(s, va, blk);
                    } else {
                        step_and_count(s, va, blk);
                        step_and_count(s, vb, blk);
                    }
                }
            }
        } else {
            // LDS ring: 16 slots of 1 KiB per wave (two steps); slot (k & 1) * 8 + u holds vector u of
            // the wave's k-th step.  DMAs for steps that do not exist re-read the current step (their
            // data is never consumed) so that exactly 15 DMAs are younger than the one being waited for.
            __shared__ uint4 ring[kThreads / 64][16][64];
            if (st < fast_end) {
                const uint4* lane_ptr = &ring[wave][0][lane];
                const uint32_t wave_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&ring[wave][0][0]));
                auto ptr = [&](uint64_t sx) { return a0 + (sx < fast_end ? sx : st) * kVecPerStep + lane_off; };
                uint4 dummy[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) lds_dma16<NT>(ptr(st) + u * US, wave_base + u * 1024);
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) lds_dma16<NT>(ptr(st + G) + u * US, wave_base + (8 + u) * 1024);
                uint32_t par = 0;
                for (; st < fast_end; st += G) {
                    const uint4* p2 = ptr(st + 2 * G);
                    step_and_count<DEPTH, 2, NT, US>(s, dummy, blk, p2, LdsStage{lane_ptr, wave_base + par * 8192, par * 8});
                    par ^= 1u;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the DMAs nobody consumes
            }
        }
    } else if constexpr (!PREFETCH) {
        for (uint64_t st = blockIdx.x; st < nsteps; st += G) {
            uint4 v[kUnroll];
            load_step<NT, US>(v, a0, st, lane_off, lo, hi, fast_begin, fast_end);
            step_and_count(s, v, blk);
        }
    } else {
        uint4 va[kUnroll], vb[kUnroll];
        uint64_t st = blockIdx.x;
        if (st < nsteps) load_step<NT, US>(va, a0, st, lane_off, lo, hi, fast_begin, fast_end);
        while (st < nsteps) {
            if (st + G < nsteps) load_step<NT, US>(vb, a0, st + G, lane_off, lo, hi, fast_begin, fast_end);
            step_and_count(s, va, blk);
            st += G;
            if (st >= nsteps) break;
            if (st + G < nsteps) load_step<NT, US>(va, a0, st + G, lane_off, lo, hi, fast_begin, fast_end);
            step_and_count(s, vb, blk);
            st += G;
        }
    }
    FSK_TL(2);
    // launch anatomy (tools/launch_anatomy.py; timing only, results are wrong): bit 9 skips the final
    // flush, bit 10 everything after it
    if (!(mode & 512)) flush(s, blk);
    if (mode & 1024) {
        uint32_t x = 0;
#pragma unroll
        for (int c = 0; c < kInternal; ++c) x ^= s.acc[c];
        if (x == 0x9E3779B9u) out[0] = x;
        return;
    }
    FSK_TL(3);

    // wave sums on the VALU (DPP), then the 4 waves through LDS
    constexpr int kWaves = (STAGE == 5 ? 2 * kThreads : kThreads) / 64;  // waves that count
    __shared__ uint32_t red[kWaves][kInternal];
    uint32_t wsum[kInternal];
#pragma unroll
    for (int c = 0; c < kInternal; ++c) wsum[c] = wave_sum_lane63(s.acc[c]);
    if (lane == 63 && wave < kWaves) {  // (the dynamic schedule's scheduler wave counts nothing)
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
        FSK_TL(4);
        // anatomy bit 3 (timing only, results land in 8 copies): workgroup b adds to copy b % 8 -- its XCD's own -- of
        // out[8][32]: how much of the epilogue is the same two cache lines bouncing between the 8 L2s?
        if (mode & 4096) out += 32 * (blockIdx.x & 7u);
        if (mode & 8)
            grouped_epilogue(wg_tot, out, mode, blockIdx.x == 0 ? hi - lo : 0, dyn.block);
        else
            finalize_slots<true>(wg_tot, out, mode, blockIdx.x == 0 ? hi - lo : 0);
        if ((mode & 2048) && threadIdx.x == 0) {
            uint64_t* row = reinterpret_cast<uint64_t*>(ticket) + static_cast<uint64_t>(blockIdx.x) * 8;
            for (int i = 0; i < 5; ++i) row[i] = tl[i];
            row[5] = wall_clock64();
            row[6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // XCC_ID
            row[7] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID
        }
        return;
    }
    if (threadIdx.x < kInternal) {
        // [counter][block]; write-through (sc1) so the finalising workgroup -- on whichever XCD -- sees it
        __hip_atomic_store(&partials[static_cast<uint64_t>(threadIdx.x) * gridDim.x + blockIdx.x], sum, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ticket == nullptr) return;  // two-kernel form: K2 (flagstat_finalize) sums the partials

    // Fused form through a last-arriver ticket (r01; measured slower than K1 + K2 at every size,
    // profiles/r01/fuse_ab.log, and superseded by the fence-free direct epilogue above; kept in the tuning
    // build as evidence): the workgroup that draws the last ticket finalises.  Hand-off per the CDNA
    // guide's counter recipe: every storing wave drains its stores, workgroup barrier, ONE lane:
    // agent-scope release -> drain -> relaxed agent ticket add; the last arriver: agent-scope acquire ->
    // drain -> barrier -> sc1 (atomic) loads of the partials.  `ticket` is zero before the first launch
    // and reset here for the next one (launches sharing a workspace are stream-ordered).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ uint32_t is_last;
    __shared__ uint64_t tot[32];
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = (t == gridDim.x - 1) ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    for (uint32_t c = wave; c < kInternal; c += kThreads / 64) {
        uint64_t x = 0;
        for (uint32_t b = lane; b < gridDim.x; b += 64)
            x += __hip_atomic_load(&partials[static_cast<uint64_t>(c) * gridDim.x + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        if (lane == 0) tot[c] = x;
    }
    __syncthreads();
    finalize_slots(tot, out, mode, hi - lo);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ read probe
// Measurement only ("fraction of a measured read-only probe
// kernel", the analogue of the reference's memcpy baseline,
// linux/instrumented_benchmark.cpp:456-544): the same load pattern as K1 with
// the arithmetic reduced to one XOR per dword.
template <bool NT>
__global__ __launch_bounds__(kThreads) void flagstat_read_probe(const uint4* __restrict__ a0, uint64_t nsteps,
                                                                uint32_t* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t lane_off = static_cast<uint64_t>(wave) * (64 * kUnroll) + lane;
    uint32_t acc = 0;
    for (uint64_t st = blockIdx.x; st < nsteps; st += gridDim.x) {
        uint4 v[kUnroll];
        const uint64_t j0 = st * kVecPerStep + lane_off;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = load_vec<NT>(a0 + j0 + u * 64);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;  // practically never; keeps the loads alive
}

}  // namespace fskt

// ------------------------------------------------------------------ launcher of the measurement build
static std::atomic<uint32_t> g_dyn_first_pct{75}, g_dyn_div{4}, g_dyn_cmax{32}, g_dyn_min_steps{32}, g_dyn_lgq{3};

extern "C" void fsk_set_dyn_queues(uint32_t lg_queues) { g_dyn_lgq = lg_queues > 4 ? 4 : lg_queues; }

extern "C" void fsk_set_dyn(uint32_t first_pct, uint32_t div, uint32_t cmax, uint32_t min_steps_per_wg)
{
    g_dyn_first_pct = first_pct > 100 ? 100 : first_pct;
    g_dyn_div = div < 1 ? 1 : (div > 64 ? 64 : div);
    g_dyn_cmax = cmax < 1 ? 1 : (cmax > 65535 ? 65535 : cmax);
    g_dyn_min_steps = min_steps_per_wg;
}

static int g_anatomy = 0;  // bit 0: no steps, bit 1: no final flush, bit 2: nothing after the flush (timing only)
static uint64_t* g_timeline = nullptr;  // fsk_timeline_run: device rows [grid][8] the direct-epilogue K1 stamps

template <int DEPTH, bool NT, bool PREFETCH, bool INTERLEAVE, int STAGE = 0>
static hipError_t launch_count_t(const fsk::CountArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL((fskt::flagstat_count<DEPTH, NT, PREFETCH, INTERLEAVE, STAGE>), dim3(a.grid),
                       dim3(STAGE == 4 ? fsk::kThreads + 64 : (STAGE == 5 ? 2 * fsk::kThreads : fsk::kThreads)), 0, stream,
                       reinterpret_cast<const uint4*>(a.a0), a.lo, a.hi, a.nsteps, a.fast_begin, a.fast_end, a.partials,
                       a.ticket, a.out, a.mode, a.dyn, a.sig);
    return hipGetLastError();
}

// fsk_launch (flagstat_kernels.hip) hands every launch here when this file is linked: the product's launcher with the
// instruments and schedules of the measurement build.  Same geometry, same mode rules (the epilogue policy is read from the
// product's knobs: fsk_launch_policy).
extern "C" hipError_t fsk_tuning_launch(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, uint64_t* d_partials,
                                        uint32_t* d_ticket, uint64_t* d_out32, hipStream_t stream, uint64_t* signal_word,
                                        uint64_t signal_value)
{
    int stagger = 1;
    uint32_t group_min_grid = 64;
    uint64_t group_max_steps = 40;
    fsk_launch_policy(&stagger, &group_min_grid, &group_max_steps);
    if (n == 0) return hipSuccess;
    if (grid == 0 || d_array == nullptr || d_partials == nullptr || d_out32 == nullptr) return hipErrorInvalidValue;
    if ((variant & 128) && d_ticket == nullptr) return hipErrorInvalidValue;  // the dynamic schedule's counter block
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_array);
    if (addr & 1u) return hipErrorInvalidValue;  // uint16_t* must be 2-byte aligned
    fsk::CountArgs a;
    const uintptr_t base = addr & ~static_cast<uintptr_t>(15);
    a.a0 = reinterpret_cast<const void*>(base);
    a.lo = (addr - base) / 2;
    a.hi = a.lo + n;
    const uint64_t nvec = (a.hi + 7) / 8;
    const uint64_t vps = ((variant & 255) == 29 ? 2 : 1) * fsk::kVecPerStep;  // schedule 29: 512-thread workgroups, 64 KiB steps
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
    if ((a.mode & 4) && ((a.mode & 1) || ((variant >> 9) & 1))) return hipErrorInvalidValue;  // accumulate form only
    a.mode |= (g_anatomy & 6) << 8;
    if ((g_anatomy & 8) && (a.mode & 4)) a.mode |= 4096;
    if (g_anatomy & 1) a.nsteps = a.fast_begin = a.fast_end = 0;  // no steps at all: launch + epilogue only
    a.ticket = ((variant >> 9) & 1) ? d_ticket : nullptr;  // bit 9: fused finalise inside K1
    if (g_timeline && (a.mode & 4)) {
        a.mode |= 2048;
        a.ticket = reinterpret_cast<uint32_t*>(g_timeline);
    }
    a.out = d_out32;
    a.dyn = fsk::DynSched{reinterpret_cast<uint64_t*>(d_ticket), 0xFFFFFFFFu, 0, 1, 0};
    // direct epilogue: many workgroups add to per-XCD copies first (grouped_epilogue); few add straight to out[]
    if (stagger) a.mode |= 16;
    // store form with a completion word: a grid of one workgroup stores its totals itself (K1's latency form, no K2);
    // larger grids go through K2, which signals after its stores
    a.sig = fsk::HostSignal{signal_word, signal_value};
    if (signal_word && !(a.mode & 1)) return hipErrorInvalidValue;  // pairs carry "=" results only
    if (signal_word && (a.mode & 1) && !(a.mode & 4) && grid == 1 && !((variant >> 9) & 1)) a.mode |= 32;
    if ((a.mode & 4) && grid >= group_min_grid && (a.nsteps + grid - 1) / grid <= group_max_steps &&
        d_ticket != nullptr && !(a.mode & 4096))
        a.mode |= 8;
    if (variant & 128) {
        // round 0 takes first_pct of the full steps; too few steps per workgroup to be worth balancing: all of them
        const uint64_t full = a.fast_end - a.fast_begin;
        const uint64_t per_wg = (full + grid - 1) / grid;
        uint64_t c0 = per_wg;
        if (per_wg >= g_dyn_min_steps.load() && per_wg >= 2) {
            c0 = full * g_dyn_first_pct.load() / (100ull * grid);
            if (c0 < 1) c0 = 1;
        }
        if (c0 > 0xFFFFFFFFull) c0 = 0xFFFFFFFFull;
        uint32_t lgq = g_dyn_lgq.load();
        while (lgq && (grid >> (3 + lgq)) == 0) --lgq;  // every queue serves at least 8 workgroups
        uint64_t per_queue = (static_cast<uint64_t>(grid) >> lgq) * g_dyn_div.load();
        if (per_queue < 2) per_queue = 2;                 // keeps 2^32 / per_queue inside 32 bits
        a.dyn.block = reinterpret_cast<uint64_t*>(d_ticket);
        a.dyn.c0 = static_cast<uint32_t>(c0);
        a.dyn.inv = static_cast<uint32_t>((1ull << 32) / per_queue);
        a.dyn.cmax = g_dyn_cmax.load();
        a.dyn.lgq = lgq;
    }
    if (((variant >> 9) & 1) && d_ticket == nullptr) return hipErrorInvalidValue;
    fsk_note_mode(a.mode);
    hipError_t e;
    // variant bits: 1 = non-temporal loads, 2 = chain depth 7 (else 8), 4 = register prefetch,
    // 8 = waves interleaved at 1 KiB within a step, 16 = rolling re-issue of load registers,
    // 32 = staging through a per-wave LDS ring filled by LDS-DMA, 64 = rolling at distance 2 (two buffers).
    // Numbers from 57 on are plain labels.  The shipped library carries the default schedule (71), the r02 default (25)
    // and the plain loop they are measured against (9).
    // The schedules that lost the r01 sweeps stay in the source as evidence and are compiled only into a
    // tuning build (make TUNING=1 -> -DFLAGSTAT_TUNING_VARIANTS; tools/tune.py, profiles/r01/tune_*.log).
    switch (variant & 255) {
    case 9: e = launch_count_t<8, true, false, true>(a, stream); break;
    case 25: e = launch_count_t<8, true, false, true, 1>(a, stream); break;   // r01-r02 default: rolling over a whole step
    // default since r03: rolling at a distance of 6 vectors (24 KiB in flight per CU instead of 32), each wave a
    // contiguous 8 KiB of the step -- +2.3-2.8 % at 8 GiB, +4 % at 1 GiB (profiles/r03/rolling_distance_sweep.log)
    case 71: e = launch_count_t<8, true, false, false, 9>(a, stream); break;
    // bit 7: 25 + guided self-scheduling.  Balances the XCDs to within 2 us of each other and is NOT faster (HBM, not the
    // split between XCDs, sets the time: profiles/r03/dyn_sweep_*.log, timeline_153_dynamic.log) -- evidence, tuning build only
    case 153: e = launch_count_t<8, true, false, true, 4>(a, stream); break;
    // 25 with TWO waves per SIMD: 512-thread workgroups, every wave rolls over half a step (4 loads in flight), same
    // 32 KiB in flight per CU -- the VERDICT r02 experiment on K1's VALU headroom (profiles/r03/two_waves_per_simd_and_lds_ring.log)
    case 29: e = launch_count_t<8, true, false, true, 5>(a, stream); break;
    // measurement only: 25's load schedule with (almost) no arithmetic (57), and 25 with each wave owning a contiguous
    // 8 KiB of the step instead of the 1 KiB interleave (17) -- where the last 1.5 % to the read probe is
    case 57: e = launch_count_t<8, true, false, true, 6>(a, stream); break;
    case 17: e = launch_count_t<8, true, false, false, 1>(a, stream); break;
    case 61: e = launch_count_t<8, true, false, true, 7>(a, stream); break;  // loads re-issued in pairs
    case 63: e = launch_count_t<8, true, false, true, 8>(a, stream); break;  // ... in fours
    case 65: e = launch_count_t<8, true, false, true, 11>(a, stream); break; // rolling distance 5 vectors (20 KiB per CU in flight)
    case 67: e = launch_count_t<8, true, false, true, 9>(a, stream); break;  // 6 (24 KiB)
    case 69: e = launch_count_t<8, true, false, true, 10>(a, stream); break; // 7 (28 KiB)
    case 73: e = launch_count_t<8, true, false, true, 12>(a, stream); break; // 6, (almost) no arithmetic: the schedule's own ceiling
    case 75: e = launch_count_t<8, true, false, false, 11>(a, stream); break; // 5, contiguous
    case 79: e = launch_count_t<8, true, false, false, 13>(a, stream); break; // 6 / 7 by wave parity (26 KiB), contiguous
    case 81: e = launch_count_t<8, true, false, false, 14>(a, stream); break; // 6 / 5 by wave parity (22 KiB), contiguous
    case 77: e = launch_count_t<8, true, false, false, 10>(a, stream); break; // 7, contiguous
    case 0: e = launch_count_t<8, false, false, false>(a, stream); break;
    case 1: e = launch_count_t<8, true, false, false>(a, stream); break;
    case 13: e = launch_count_t<8, true, true, true>(a, stream); break;
    case 27: e = launch_count_t<7, true, false, true, 1>(a, stream); break;
    case 41: e = launch_count_t<8, true, false, true, 2>(a, stream); break;  // bit 5: LDS-DMA ring instead of registers
    case 89: e = launch_count_t<8, true, false, true, 3>(a, stream); break;  // bit 6: rolling registers at distance 2
    default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    if (a.ticket || (a.mode & (4 | 32))) return hipSuccess;  // K1 finalised by itself
    return fsk_launch_finalize(d_partials, grid, d_out32, a.mode, n, a.sig, stream);   // the product's K2
}

extern "C" void fsk_set_anatomy(int bits) { g_anatomy = bits; }

// Workgroup timeline of ONE direct-epilogue K1 launch (after `warm` untimed ones): h_rows[grid][8] =
// wall-clock stamps (100 MHz) at entry / first step done / last step done / flushed / reduced / exit, XCC_ID, HW_ID.
// Returns the grid used, 0 on failure.  Measurement only (tools/timeline.py).
extern "C" uint32_t fsk_timeline_run(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, int warm,
                                     uint64_t* h_rows, uint32_t cap_rows)
{
    if (grid == 0 || grid > cap_rows) return 0;
    uint64_t *rows = nullptr, *partials = nullptr, *out = nullptr;
    hipStream_t s = nullptr;
    uint32_t used = 0;
    if (hipMalloc(&rows, static_cast<size_t>(grid) * 64) == hipSuccess && hipMalloc(&partials, fsk_partials_bytes(grid)) == hipSuccess &&
        hipMalloc(&out, 256) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
        hipMemsetAsync(out, 0, 256, s) == hipSuccess && hipMemsetAsync(rows, 0, static_cast<size_t>(grid) * 64, s) == hipSuccess &&
        hipMemsetAsync(partials, 0, fsk_partials_bytes(grid), s) == hipSuccess) {
        bool ok = true;
        uint32_t* block = reinterpret_cast<uint32_t*>(partials + static_cast<size_t>(grid) * fsk::kInternal);
        for (int i = 0; i < warm && ok; ++i)
            ok = fsk_tuning_launch(d_array, n, grid, (variant & 255) | 2048, partials, block, out, s, nullptr, 0) == hipSuccess;
        g_timeline = rows;
        ok = ok && fsk_tuning_launch(d_array, n, grid, (variant & 255) | 2048, partials, block, out, s, nullptr, 0) == hipSuccess;
        g_timeline = nullptr;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        const uint64_t nvec = (n + 7 + 7) / 8;  // upper bound on the steps (a ragged head adds at most one)
        used = grid;
        const uint64_t nsteps = (nvec + fsk::kVecPerStep - 1) / fsk::kVecPerStep;
        if (nsteps < used) used = static_cast<uint32_t>(nsteps);
        ok = ok && hipMemcpy(h_rows, rows, static_cast<size_t>(used) * 64, hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok) used = 0;
    }
    if (s) (void)hipStreamDestroy(s);
    if (rows) (void)hipFree(rows);
    if (partials) (void)hipFree(partials);
    if (out) (void)hipFree(out);
    return used;
}

// which K1 schedules the measurement build carries (bits 0-6 of `variant`)
extern "C" int fsk_tuning_variant_supported(int variant)
{
    switch (variant & 255) {
    case 9:
    case 25:
    case 71: return 1;
    case 17:
    case 29:
    case 57:
    case 61:
    case 63:
    case 65:
    case 67:
    case 69:
    case 73:
    case 75:
    case 77:
    case 79:
    case 81:
    case 153:
    case 0:
    case 1:
    case 13:
    case 27:
    case 41:
    case 89: return 1;
    default: return 0;
    }
}

// read-only bandwidth probe over the first floor(bytes / 32 KiB) steps of a 16-B aligned buffer
extern "C" hipError_t fsk_read_probe(const void* d_buf, uint64_t bytes, uint32_t grid, int nt, uint32_t* d_sink,
                                     hipStream_t stream)
{
    if ((reinterpret_cast<uintptr_t>(d_buf) & 15u) || grid == 0) return hipErrorInvalidValue;
    const uint64_t nsteps = bytes / (16ull * fsk::kVecPerStep);
    if (nsteps == 0) return hipSuccess;
    if (nt)
        hipLaunchKernelGGL(fskt::flagstat_read_probe<true>, dim3(grid), dim3(fsk::kThreads), 0, stream,
                           reinterpret_cast<const uint4*>(d_buf), nsteps, d_sink);
    else
        hipLaunchKernelGGL(fskt::flagstat_read_probe<false>, dim3(grid), dim3(fsk::kThreads), 0, stream,
                           reinterpret_cast<const uint4*>(d_buf), nsteps, d_sink);
    return hipGetLastError();
}
