// flagstat_kernels.hip -- the flagstat hot path as hand-written HIP for gfx950 (CDNA4).
//
// What it replaces (reference, /root/reference): the FLAGSTAT_* kernels of
// libflagstats.h -- mask-select / propagate-carry front end (:1695-1704 and the
// LUT forms :1908-1935, :2474-2489), the Harley-Seal carry-save tree (:1706-1767,
// python/libalgebra.h:2311-2319), the counter flush (:1769-1840) and the scalar
// tail (:1649-1651) -- with FLAGSTAT_scalar's exact 19-counter semantics
// (:118-142), which of the reference's SIMD variants only _avx512_improved3
// (:2321-2644) has.  Nothing here is a translation of those x86 kernels; the
// design is for wave64 / VALU / HBM3E:
//
//  K1 flagstat_count   grid-stride over 32 KiB "steps"; each lane loads 8 x 16 B
//                      (global_load_dwordx4, 1 KiB per wave-instruction, fully
//                      coalesced), straight to VGPRs (read-once stream: an LDS
//                      round trip buys nothing).  Per 16 B (8 flags):
//                        * v_perm_b32 splits 2 dwords (4 flags) into a dword of
//                          low bytes L and a dword of high bytes H  (byte-planar:
//                          every later op works on 4 flags at once);
//                        * three v_perm_b32 LUTs evaluate the flagstat decision
//                          tree: (proper,unmap,munmap) -> n_pair_good/n_sgltn/
//                          n_pair_map bits; (secondary,paired,supplementary) ->
//                          category keep-mask; (qcfail,dup) -> fail mask + one-hot;
//                        * result: T = 8 counter bits per flag (any QC),
//                          F = T & fail-QC mask, S = one-hot(QC-only, DUP-only, both) + the
//                          primary-paired indicator split by QC class (n_pair_all of the samtools
//                          loop, benchmark/flagstats.cpp:58 -- not among the scalar rule's slots,
//                          reported only through the superset entry points);
//                        * T/F/S dwords go into bit-sliced carry-save counters:
//                          one CSA = 2 x v_bitop3_b32 (0x96 sum, 0xE8 majority).
//                      16 T-inputs per step collapse through a Harley-Seal tree;
//                      the weight-16 carry enters a binary-counter chain of
//                      (accumulator, pending) plane pairs driven by the *scalar*
//                      step count, so the amortised cost stays 1 CSA per input at
//                      any depth and the cross-lane transpose happens once per
//                      epoch (2^DEPTH-1 steps), not per block as on x86 (:1751).
//                      Epoch flush: v_bcnt_u32_b32 per (plane, counter) into 19
//                      u32 lane counters; kernel end: wave butterfly + LDS ->
//                      per-block uint64[19] partials.
//  K2 flagstat_finalize sums partials on device, maps the 19 internal counters
//                      to the reference's 32 slots and ADDS into out[32]
//                      (accumulate contract).
//
// Zero flags contribute to no counter (KAT x=0), so ragged heads/tails and idle
// lanes are handled by zero-filling: no host-side tail, no scalar fallback.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flagstat_device.h"
#include "flagstat_kernels.h"

namespace fsk {

// Front end for 4 flags held in two dwords (xa = flags 0,1; xb = flags 2,3).
//
// Output byte layout (one byte per flag, bit -> internal counter index):
//   bit0 secondary            bit1 n_pair_good (proper & !unmap & pp)
//   bit2 unmapped             bit3 supplementary & !secondary
//   bit4 n_sgltn  (munmap & !unmap & pp)     bit5 n_pair_map (!munmap & !unmap & pp)
//   bit6 read1 & pp           bit7 read2 & pp
// where pp = paired & !secondary & !supplementary  (libflagstats.h:129-131).
// selq: per byte (qcfail | dup<<1), feeds the QC/DUP LUTs.  keep: the category keep-mask, whose
// bits 6 and 7 are set exactly for primary paired reads (pp).
__device__ __forceinline__ void front4(uint32_t xa, uint32_t xb, uint32_t& T, uint32_t& selq, uint32_t& keep)
{
    const uint32_t L = perm(xb, xa, 0x06040200u);  // FLAG bits 0..7  of 4 flags
    const uint32_t H = perm(xb, xa, 0x07050301u);  // FLAG bits 8..15 of 4 flags

    // LUT 1: idx = (proper, unmap, munmap) = L bits 1..3.  Entry = derived bits
    // at 1/4/5, plus constant ones at bits 0 and 3 (so the AND below passes the
    // secondary / supplementary bits that LUT 2 supplies).
    const uint32_t sel1 = (L >> 1) & 0x07070707u;
    const uint32_t abc = perm(0x09091B19u, 0x09092B29u, sel1);
    // raw read1/read2 (6,7) and unmapped (2) from L, derived bits from the LUT
    const uint32_t m = (L & 0xC4C4C4C4u) | abc;

    // LUT 2: idx = (secondary, paired, supplementary).  Entry = keep-mask:
    //   secondary         -> 0x01        supplementary only -> 0x08
    //   primary paired    -> 0xF2        none of them       -> 0x00
    // always | 0x04 so the unconditional UNMAP bit survives.
    uint32_t idx = (H & 0x01010101u);
    idx = ((H >> 1) & 0x04040404u) | idx;
    idx = ((L << 1) & 0x02020202u) | idx;
    keep = perm(0x050C050Cu, 0x05F60504u, idx);
    T = m & keep;

    selq = (H >> 1) & 0x03030303u;  // bit0 = QCFAIL, bit1 = DUP
}

// ------------------------------------------------------------------ lane state
// Bit-sliced counters of one lane.  Streams, each one byte per flag (4 flags per dword):
// T (8 counters), F (= T under fail-QC), S (bits 0-2 one-hot QC-only / DUP-only / both,
// bit 6 = pp & pass-QC, bit 7 = pp & fail-QC).
template <int DEPTH>
struct Lane {
    uint32_t t1, t2, t4, t8;   // T planes of weight 1,2,4,8
    uint32_t f1, f2, f4, f8;
    uint32_t s1, s2, s4, s8;
    uint32_t tA[DEPTH], tB[DEPTH];  // chain level j: weight 16<<j  (accumulator, pending)
    uint32_t fA[DEPTH], fB[DEPTH];
    uint32_t sA[DEPTH], sB[DEPTH];
    uint32_t acc[kInternal];        // flushed lane counters
};

template <int DEPTH>
__device__ __forceinline__ void lane_init(Lane<DEPTH>& s)
{
    s.t1 = s.t2 = s.t4 = s.t8 = 0;
    s.f1 = s.f2 = s.f4 = s.f8 = 0;
    s.s1 = s.s2 = s.s4 = s.s8 = 0;
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) s.tA[j] = s.tB[j] = s.fA[j] = s.fB[j] = s.sA[j] = s.sB[j] = 0;
#pragma unroll
    for (int c = 0; c < kInternal; ++c) s.acc[c] = 0;
}

// Binary-counter chain.  `blk` (steps pushed so far in this epoch) is wave-
// uniform, so the branches are scalar.  Level j: bit j of blk clear -> park the
// carry in the pending plane; set -> CSA(accumulator, pending, carry) and
// ripple the new carry up.  Epochs end at 2^DEPTH-1 steps, so the top level
// never carries out.
template <int J, int DEPTH>
__device__ __forceinline__ void chain_push(Lane<DEPTH>& s, uint32_t blk, uint32_t ct, uint32_t cf, uint32_t cs)
{
    if constexpr (J < DEPTH) {
        if ((blk & (1u << J)) == 0) {
            s.tB[J] = ct;
            s.fB[J] = cf;
            s.sB[J] = cs;
        } else {
            uint32_t nt, nf, ns;
            csa(nt, s.tA[J], s.tA[J], s.tB[J], ct);
            csa(nf, s.fA[J], s.fA[J], s.fB[J], cf);
            csa(ns, s.sA[J], s.sA[J], s.sB[J], cs);
            s.tB[J] = 0;
            s.fB[J] = 0;
            s.sB[J] = 0;
            chain_push<J + 1, DEPTH>(s, blk, nt, nf, ns);
        }
    }
}

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
template <int DEPTH, int STAGE, bool NT, int USTRIDE>
__device__ __forceinline__ void step(Lane<DEPTH>& s, uint4 (&v)[kUnroll], uint32_t blk, const uint4* __restrict__ next,
                                     LdsStage lds = LdsStage{nullptr, 0, 0})
{
    constexpr bool ROLL = (STAGE == 1);
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
                uint4 x;
                if constexpr (STAGE == 2) {
                    const int u = half * 4 + q * 2 + k;  // a constant after unrolling
                    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                    x = lds.lane[(lds.first + u) * 64];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot has been read: it may be refilled
                    lds_dma16<NT>(next + u * USTRIDE, lds.slot0 + u * 1024);
                } else {
                    x = v[half * 4 + q * 2 + k];
                }
                if constexpr (ROLL) {
                    // Copy the vector out with real v_movs HERE (a load may land at any time, so the
                    // registers it targets must be dead first), then re-issue into the same registers.
                    // The asm keeps hipcc from turning the copies into loop-top PHI moves (which wait
                    // for all 8 loads), the sched_barriers from sinking the loads below the arithmetic.
                    __builtin_amdgcn_sched_barrier(0);
                    x = copy_out(x);
                    v[half * 4 + q * 2 + k] = load_vec<NT>(next + (half * 4 + q * 2 + k) * USTRIDE);
                    __builtin_amdgcn_sched_barrier(0);
                }
                uint32_t qa, qb, ka, kb;
                front4(x.x, x.y, T[2 * k], qa, ka);
                front4(x.z, x.w, T[2 * k + 1], qb, kb);
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

// Flush: fold every plane into the 21 u32 lane counters and clear them.  `pushed` = steps pushed
// since the last flush (wave-uniform): chain level j can hold a carry only after 2^j steps, so a
// short run (a mid-size array leaves each workgroup a few dozen steps) skips the empty upper levels
// with scalar branches -- the flush is ~1300 VALU ops at full depth, paid once per kernel, and is
// part of the fixed cost that keeps sub-GiB launches off the roofline.
template <int DEPTH>
__device__ __forceinline__ void flush(Lane<DEPTH>& s, uint32_t pushed)
{
    constexpr int NS = kInternal - 16;
    uint32_t at[8], af[8], as[NS];
#pragma unroll
    for (int c = 0; c < 8; ++c) at[c] = af[c] = 0;
#pragma unroll
    for (int c = 0; c < NS; ++c) as[c] = 0;
#pragma unroll
    for (int j = DEPTH - 1; j >= 0; --j) {
        if (pushed >= (1u << j)) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t mk = 0x01010101u << c;
                at[c] = hstep(at[c], s.tA[j], mk, true);
                at[c] = hstep(at[c], s.tB[j], mk, false);
                af[c] = hstep(af[c], s.fA[j], mk, true);
                af[c] = hstep(af[c], s.fB[j], mk, false);
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                const uint32_t mk = 0x01010101u << (c < 3 ? c : c + 3);  // S bits 0,1,2 and 6,7
                as[c] = hstep(as[c], s.sA[j], mk, true);
                as[c] = hstep(as[c], s.sB[j], mk, false);
            }
            s.tA[j] = s.tB[j] = s.fA[j] = s.fB[j] = s.sA[j] = s.sB[j] = 0;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t mk = 0x01010101u << c;
        at[c] = hstep(at[c], s.t8, mk, true);
        at[c] = hstep(at[c], s.t4, mk, true);
        at[c] = hstep(at[c], s.t2, mk, true);
        at[c] = hstep(at[c], s.t1, mk, true);
        af[c] = hstep(af[c], s.f8, mk, true);
        af[c] = hstep(af[c], s.f4, mk, true);
        af[c] = hstep(af[c], s.f2, mk, true);
        af[c] = hstep(af[c], s.f1, mk, true);
        s.acc[c] += at[c];
        s.acc[8 + c] += af[c];
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const uint32_t mk = 0x01010101u << (c < 3 ? c : c + 3);
        as[c] = hstep(as[c], s.s8, mk, true);
        as[c] = hstep(as[c], s.s4, mk, true);
        as[c] = hstep(as[c], s.s2, mk, true);
        as[c] = hstep(as[c], s.s1, mk, true);
        s.acc[16 + c] += as[c];
    }
    s.t1 = s.t2 = s.t4 = s.t8 = 0;
    s.f1 = s.f2 = s.f4 = s.f8 = 0;
    s.s1 = s.s2 = s.s4 = s.s8 = 0;
}

// Map the 21 internal totals to the reference's 32 slots (index = FLAGSTAT_*_OFF,
// libflagstats.h:69-112; +16 for fail-QC) and add them to / store them in out[32].
// Called by the first 32 threads of a workgroup after tot[] is complete.
// mode bit 0: store instead of accumulate.  mode bit 1: superset -- additionally slots 0 / 16 =
// primary paired reads by QC class (samtools' n_pair_all, benchmark/flagstats.cpp:58; the slot the
// reference's SIMD kernels fill with the same quantity for their SIMD-covered prefix) and
// slot 9 = pass-QC reads (the "QC adjust" libflagstats.h:1843 of those kernels: len - fail-QC reads).
// Without bit 1 the 32 slots are exactly FLAGSTAT_scalar's (libflagstats.h:118-142).
// ATOMIC (K1's direct epilogue): tot[] are ONE workgroup's totals, added with relaxed agent-scope
// atomics -- any number of launches, on any streams, may target the same out[32].
template <bool ATOMIC = false>
__device__ __forceinline__ void finalize_slots(const uint64_t* tot, uint64_t* __restrict__ out, int mode, uint64_t n_flags)
{
    if (threadIdx.x < 32) {
        // reference slot -> internal T index (secondary, n_pair_good, unmap, supplementary,
        // n_sgltn, n_pair_map, read1, read2), -1 = slot has no T/F counter
        const int t_of_slot[16] = {-1, -1, 2, -1, -1, -1, 6, 7, 0, -1, -1, 3, 1, 4, 5, -1};
        const uint32_t slot = threadIdx.x & 15u;
        const bool fail = threadIdx.x >= 16;
        uint64_t add = 0;
        const int t = t_of_slot[slot];
        if (t >= 0) add = fail ? tot[8 + t] : tot[t] - tot[8 + t];  // pass-QC = all - fail
        if (slot == 10) add = fail ? tot[18] : tot[17];              // DUP: fail / pass
        if (slot == 9 && fail) add = tot[16] + tot[18];              // fail-QC read count (slot 25)
        if (mode & 2) {
            if (slot == 0) add = fail ? tot[20] : tot[19];
            // ATOMIC: n_flags is the launch's flag count in workgroup 0 and 0 elsewhere; the partial
            // sums wrap modulo 2^64 and the total over all workgroups is len - fail-QC reads
            if (slot == 9 && !fail) add = n_flags - (tot[16] + tot[18]);
        }
        if constexpr (ATOMIC) {
            if (add) (void)__hip_atomic_fetch_add(&out[threadIdx.x], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (mode & 1)
                out[threadIdx.x] = add;        // "=" form: all 32 slots written, dead slots as 0
            else if (add)
                out[threadIdx.x] += add;       // reference contract: accumulate, never touch dead slots
        }
    }
}

// ------------------------------------------------------------------ K1
// USTRIDE = vectors between a lane's consecutive loads: 64 -> each wave owns a contiguous
// 8 KiB of the step; 256 -> the 4 waves interleave at 1 KiB (each load instruction of the
// workgroup covers a contiguous 4 KiB).
template <bool NT, int USTRIDE>
__device__ __forceinline__ void load_step(uint4 (&v)[kUnroll], const uint4* __restrict__ a0, uint64_t st, uint64_t lane_off,
                                          uint64_t lo, uint64_t hi, uint64_t fast_begin, uint64_t fast_end)
{
    const uint64_t j0 = st * kVecPerStep + lane_off;
    if (st >= fast_begin && st < fast_end) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = load_vec<NT>(a0 + j0 + u * USTRIDE);
    } else {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = load_guarded(a0, j0 + u * USTRIDE, lo, hi);
    }
}

template <int DEPTH, int STAGE = 0, bool NT = false, int USTRIDE = 64>
__device__ __forceinline__ void step_and_count(Lane<DEPTH>& s, uint4 (&v)[kUnroll], uint32_t& blk,
                                               const uint4* __restrict__ next = nullptr, LdsStage lds = LdsStage{nullptr, 0, 0})
{
    step<DEPTH, STAGE, NT, USTRIDE>(s, v, blk, next, lds);
    ++blk;
    if (blk == (1u << DEPTH) - 1u) {
        flush(s, (1u << DEPTH) - 1u);
        blk = 0;
    }
}

// PREFETCH = false: load 8 x 16 B, wait, compute; latency is hidden by the other
// waves of the SIMD only.  PREFETCH = true: two register buffers, the loads of
// step k+1 are in flight while step k is computed (one more 8 KiB per wave in
// flight, +32 VGPRs).
template <int DEPTH, bool NT, bool PREFETCH, bool INTERLEAVE, int STAGE>
__global__ __launch_bounds__(kThreads) void flagstat_count(const uint4* __restrict__ a0, uint64_t lo, uint64_t hi,
                                                           uint64_t nsteps, uint64_t fast_begin, uint64_t fast_end,
                                                           uint64_t* __restrict__ partials, uint32_t* ticket,
                                                           uint64_t* out, int mode)
{
    Lane<DEPTH> s;
    lane_init(s);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // vector of (wave, u, lane) within a step: wave*512 + u*64 + lane, or u*256 + wave*64 + lane
    constexpr int US = INTERLEAVE ? kThreads : 64;
    const uint64_t lane_off = INTERLEAVE ? static_cast<uint64_t>(threadIdx.x)
                                         : static_cast<uint64_t>(wave) * (64 * kUnroll) + lane;
    const uint64_t G = gridDim.x;
    uint32_t blk = 0;

    constexpr bool ROLL = (STAGE != 0);
    if constexpr (ROLL) {
        // ragged edge steps (at most the first and the last of the whole array) go through the
        // guarded loader, outside the pipelined loop
        if (fast_begin != 0 && blockIdx.x == 0) {
            uint4 v[kUnroll];
            load_step<NT, US>(v, a0, 0, lane_off, lo, hi, fast_begin, fast_end);
            step_and_count(s, v, blk);
        }
        if (nsteps > fast_end && nsteps - 1 >= fast_begin && (nsteps - 1) % G == blockIdx.x) {
            uint4 v[kUnroll];
            load_step<NT, US>(v, a0, nsteps - 1, lane_off, lo, hi, fast_begin, fast_end);
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
                    step_and_count<DEPTH, 1, NT, US>(s, v, blk, p);
                }
                step_and_count(s, v, blk);
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
#ifdef FLAGSTAT_TUNING_VARIANTS
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
#else
    flush(s, blk);
#endif

    // wave sums on the VALU (DPP), then the 4 waves through LDS
    __shared__ uint32_t red[kThreads / 64][kInternal];
    uint32_t wsum[kInternal];
#pragma unroll
    for (int c = 0; c < kInternal; ++c) wsum[c] = wave_sum_lane63(s.acc[c]);
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < kInternal; ++c) red[wave][c] = wsum[c];
    }
    __syncthreads();
    uint64_t sum = 0;
    if (threadIdx.x < kInternal) {
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) sum += red[w][threadIdx.x];
    }
    if (mode & 4) {
        // Direct epilogue (accumulate contract only): this workgroup maps ITS 21 totals to the
        // reference's slots and adds them to out[32] with relaxed agent-scope atomics (no return, no
        // fence, no ticket, no K2 launch).  Integer sums are exact in any order; pass-QC = T - F holds
        // per workgroup because F counts a subset of T.  Kernel end is the only ordering anyone needs.
        __shared__ uint64_t wg_tot[32];
        if (threadIdx.x < kInternal) wg_tot[threadIdx.x] = sum;
        __syncthreads();
        finalize_slots<true>(wg_tot, out, mode, blockIdx.x == 0 ? hi - lo : 0);
        return;
    }
    if (threadIdx.x < kInternal) {
        // [counter][block]; write-through (sc1) so the finalising workgroup -- on whichever XCD -- sees it
        __hip_atomic_store(&partials[static_cast<uint64_t>(threadIdx.x) * gridDim.x + blockIdx.x], sum, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef FLAGSTAT_TUNING_VARIANTS
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
#else
    (void)ticket;  // two-kernel form: K2 (flagstat_finalize) sums the partials
#endif
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

// ------------------------------------------------------------------ K2
// One workgroup of 16 waves: wave w sums column w (and w+16) of partials[19][nblocks]
// with coalesced 8-byte loads, then 32 threads map the 19 internal counters to the
// reference's 32 slots (index = FLAGSTAT_*_OFF, libflagstats.h:69-112; +16 for fail-QC)
// and ADD into out[32].  Slots the scalar rule never writes get nothing added.
constexpr int kFinalizeThreads = 1024;

__global__ __launch_bounds__(kFinalizeThreads) void flagstat_finalize(const uint64_t* __restrict__ partials,
                                                                      uint32_t nblocks, uint64_t* __restrict__ out,
                                                                      int mode, uint64_t n_flags)
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
    finalize_slots(tot, out, mode, n_flags);
}

}  // namespace fsk

// ------------------------------------------------------------------ launchers
// partials[19][grid] followed by a 256-byte block holding the hand-off ticket
extern "C" size_t fsk_partials_bytes(uint32_t grid) { return static_cast<size_t>(grid) * fsk::kInternal * sizeof(uint64_t) + 256; }

#ifdef FLAGSTAT_TUNING_VARIANTS
static int g_anatomy = 0;  // bit 0: no steps, bit 1: no final flush, bit 2: nothing after the flush (timing only)
#endif

template <int DEPTH, bool NT, bool PREFETCH, bool INTERLEAVE, int STAGE = 0>
static hipError_t launch_count_t(const fsk::CountArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL((fsk::flagstat_count<DEPTH, NT, PREFETCH, INTERLEAVE, STAGE>), dim3(a.grid), dim3(fsk::kThreads), 0, stream,
                       reinterpret_cast<const uint4*>(a.a0), a.lo, a.hi, a.nsteps, a.fast_begin, a.fast_end, a.partials,
                       a.ticket, a.out, a.mode);
    return hipGetLastError();
}

// Host-side geometry: everything the kernel assumes is derived here from
// (pointer, n) so operand shapes and the grid cannot disagree.
extern "C" hipError_t fsk_launch(const uint16_t* d_array, uint64_t n, uint32_t grid, int variant, uint64_t* d_partials,
                                 uint32_t* d_ticket, uint64_t* d_out32, hipStream_t stream)
{
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
    a.nsteps = (nvec + fsk::kVecPerStep - 1) / fsk::kVecPerStep;
    // steps whose 2048 vectors are all fully inside [lo, hi)
    a.fast_begin = (a.lo == 0) ? 0 : 1;
    a.fast_end = (a.hi / 8) / fsk::kVecPerStep;
    if (a.fast_end < a.fast_begin) a.fast_end = a.fast_begin;
    if (static_cast<uint64_t>(grid) > a.nsteps) grid = static_cast<uint32_t>(a.nsteps);
    a.grid = grid;
    a.partials = d_partials;
    // bit 8: store instead of accumulate; bit 10: superset slots; bit 11: direct (atomic) epilogue, no K2
    a.mode = ((variant >> 8) & 1) | (((variant >> 10) & 1) << 1) | (((variant >> 11) & 1) << 2);
    if ((a.mode & 4) && ((a.mode & 1) || ((variant >> 9) & 1))) return hipErrorInvalidValue;  // accumulate form only
#ifdef FLAGSTAT_TUNING_VARIANTS
    a.mode |= (g_anatomy & 6) << 8;
    if (g_anatomy & 1) a.nsteps = a.fast_begin = a.fast_end = 0;  // no steps at all: launch + epilogue only
#endif
#ifndef FLAGSTAT_TUNING_VARIANTS
    if ((variant >> 9) & 1) return hipErrorInvalidValue;  // the ticket-fused finalise exists in the tuning build only
#endif
    a.ticket = ((variant >> 9) & 1) ? d_ticket : nullptr;  // bit 9: fused finalise inside K1
    a.out = d_out32;
    if (((variant >> 9) & 1) && d_ticket == nullptr) return hipErrorInvalidValue;
    hipError_t e;
    // variant bits: 1 = non-temporal loads, 2 = chain depth 7 (else 8), 4 = register prefetch,
    // 8 = waves interleaved at 1 KiB within a step, 16 = rolling re-issue of load registers,
    // 32 = staging through a per-wave LDS ring filled by LDS-DMA, 64 = rolling at distance 2 (two buffers).
    // The shipped library carries the default schedule (25) and the plain loop it is measured against (9).
    // The schedules that lost the r01 sweeps stay in the source as evidence and are compiled only into a
    // tuning build (make TUNING=1 -> -DFLAGSTAT_TUNING_VARIANTS; tools/tune.py, profiles/r01/tune_*.log).
    switch (variant & 127) {
    case 9: e = launch_count_t<8, true, false, true>(a, stream); break;
    case 25: e = launch_count_t<8, true, false, true, 1>(a, stream); break;
#ifdef FLAGSTAT_TUNING_VARIANTS
    case 0: e = launch_count_t<8, false, false, false>(a, stream); break;
    case 1: e = launch_count_t<8, true, false, false>(a, stream); break;
    case 13: e = launch_count_t<8, true, true, true>(a, stream); break;
    case 27: e = launch_count_t<7, true, false, true, 1>(a, stream); break;
    case 41: e = launch_count_t<8, true, false, true, 2>(a, stream); break;  // bit 5: LDS-DMA ring instead of registers
    case 89: e = launch_count_t<8, true, false, true, 3>(a, stream); break;  // bit 6: rolling registers at distance 2
#endif
    default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    if (a.ticket || (a.mode & 4)) return hipSuccess;  // K1 finalised by itself
    hipLaunchKernelGGL(fsk::flagstat_finalize, dim3(1), dim3(fsk::kFinalizeThreads), 0, stream, d_partials, grid, d_out32,
                       a.mode, n);
    return hipGetLastError();
}

#ifdef FLAGSTAT_TUNING_VARIANTS
extern "C" void fsk_set_anatomy(int bits) { g_anatomy = bits; }
#else
extern "C" void fsk_set_anatomy(int) {}
#endif

// which K1 schedules this build of the library carries (bits 0-6 of `variant`)
extern "C" int fsk_variant_supported(int variant)
{
    switch (variant & 127) {
    case 9:
    case 25: return 1;
#ifdef FLAGSTAT_TUNING_VARIANTS
    case 0:
    case 1:
    case 13:
    case 27:
    case 41:
    case 89: return 1;
#endif
    default: return 0;
    }
}

extern "C" int fsk_tuning_build(void)
{
#ifdef FLAGSTAT_TUNING_VARIANTS
    return 1;
#else
    return 0;
#endif
}

// read-only bandwidth probe over the first floor(bytes / 32 KiB) steps of a 16-B aligned buffer
extern "C" hipError_t fsk_read_probe(const void* d_buf, uint64_t bytes, uint32_t grid, int nt, uint32_t* d_sink,
                                     hipStream_t stream)
{
    if ((reinterpret_cast<uintptr_t>(d_buf) & 15u) || grid == 0) return hipErrorInvalidValue;
    const uint64_t nsteps = bytes / (16ull * fsk::kVecPerStep);
    if (nsteps == 0) return hipSuccess;
    if (nt)
        hipLaunchKernelGGL(fsk::flagstat_read_probe<true>, dim3(grid), dim3(fsk::kThreads), 0, stream,
                           reinterpret_cast<const uint4*>(d_buf), nsteps, d_sink);
    else
        hipLaunchKernelGGL(fsk::flagstat_read_probe<false>, dim3(grid), dim3(fsk::kThreads), 0, stream,
                           reinterpret_cast<const uint4*>(d_buf), nsteps, d_sink);
    return hipGetLastError();
}
