// flagstat_count_core.h -- the building blocks of K1 flagstat_count (flagstat_kernels.hip): the byte-planar front end, a
// lane's bit-sliced counters with their binary-counter chain, the epoch flush, the map from the 21 internal counters to the
// reference's 32 slots and the two epilogue forms.  Device code only.  Shared by the product kernel and by the measurement
// build's schedules (flagstat_kernels_tuning.hip, `make tuning`), which differ in HOW a step's vectors are loaded, not in what
// is done with them.
#ifndef FLAGSTAT_COUNT_CORE_H_
#define FLAGSTAT_COUNT_CORE_H_

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
__device__ __forceinline__ void front4(uint32_t L, uint32_t H, uint32_t& T, uint32_t& selq, uint32_t& keep)
{
    // L = FLAG bits 0..7 of 4 flags, H = FLAG bits 8..15 (byte-planar: split4 / perm of the two loaded dwords)
    // LUT 1: idx = (proper, unmap, munmap) = L bits 1..3.  Entry = derived bits
    // at 1/4/5, plus constant ones at bits 0 and 3 (so the AND below passes the
    // secondary / supplementary bits that LUT 2 supplies).
    const uint32_t sel1 = (L >> 1) & 0x07070707u;
    const uint32_t abc = perm(0x09091B19u, 0x09092B29u, sel1);
    // raw read1/read2 (6,7) and unmapped (2) from L, derived bits from the LUT: one v_bfi-shaped v_bitop3_b32
    const uint32_t m = __builtin_amdgcn_bitop3_b32(L, abc, 0xC4C4C4C4u, 0xE4);  // mask ? L : abc

    // LUT 2: idx = (secondary, paired, supplementary).  Entry = keep-mask:
    //   secondary         -> 0x01        supplementary only -> 0x08
    //   primary paired    -> 0xF2        none of them       -> 0x00
    // always | 0x04 so the unconditional UNMAP bit survives.
    // (4 ops through v_lshl_or_b32 / v_and_or_b32; left to itself hipcc builds it from 5)
    const uint32_t h1 = H >> 1;  // shared with selq
    const uint32_t idx = and_or(h1, 0x04040404u, lshl1_or(L & 0x01010101u, H & 0x01010101u));
    keep = perm(0x050C050Cu, 0x05F60504u, idx);
    T = m & keep;

    selq = h1 & 0x03030303u;  // bit0 = QCFAIL, bit1 = DUP
}

// Split a just-loaded vector (8 flags) into its four byte planes with real v_perm_b32s AT THIS POINT of the instruction
// stream: the registers it was loaded into are dead afterwards and are re-targeted by the load of the lane's next step
// (K1's ROLL path: a load may land at any moment).  This is the r02 copy_out (4 v_movs per vector) with the copies done
// by the front end's own first instructions -- 32 VALU ops per step fewer.
__device__ __forceinline__ void split_out(const uint4& o, uint32_t& L0, uint32_t& H0, uint32_t& L1, uint32_t& H1)
{
    const uint32_t lo = 0x06040200u, hi = 0x07050301u;
    asm volatile("v_perm_b32 %0, %5, %4, %8\n\tv_perm_b32 %1, %5, %4, %9\n\tv_perm_b32 %2, %7, %6, %8\n\tv_perm_b32 %3, %7, %6, %9"
                 : "=&v"(L0), "=&v"(H0), "=&v"(L1), "=&v"(H1)
                 : "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w), "s"(lo), "s"(hi));
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

// Flush: fold every plane into the 21 u32 lane counters and clear them.  `pushed` = steps pushed since the last
// flush (wave-uniform): chain level j can hold something only after 2^j steps, so a short run (a mid-size array
// leaves each workgroup a few dozen steps) skips the empty upper levels with scalar branches.  The flush is paid once
// per epoch and once at the end of the kernel, where nothing hides it (every wave of the chip flushes at the same
// time with HBM idle: tools/timeline.py), so it is built to be short:
//   1. carry-propagate: level j holds TWO planes of weight 16 << j (accumulator, pending); one CSA per level with the
//      carry from below turns the chain into plain binary -- ONE plane per weight, 12 planes instead of 20 (a column
//      holds at most 16 * 255 + 15 = 4095, so nothing is carried out of the top level);
//   2. per (plane, counter) two VALU ops: v_and_b32 selects the counter's bit in each of the 4 flag bytes and
//      v_dot4_u32_u8 adds the 4 bytes times the plane's weight to the counter's accumulator -- the multiply-add does
//      the weighting, there is no Horner doubling (r02: and + shift + bcnt = 3 ops on 20 planes, 1260 ops; now 594).
//      A selected byte is 0 or 2^c, so counter c accumulates 2^c times its count (shifted out at the end); weights
//      256..2048 enter with weights 1..8 before the accumulator is shifted left by 8.
__device__ __forceinline__ uint32_t wdot(uint32_t acc, uint32_t plane, uint32_t mask, uint32_t weight_bytes)
{
    return __builtin_amdgcn_udot4(plane & mask, weight_bytes, acc, false);
}

template <int DEPTH>
__device__ __forceinline__ void flush(Lane<DEPTH>& s, uint32_t pushed)
{
    static_assert(DEPTH >= 5 && DEPTH <= 8, "weights 16 << j are split at j = 4 (256) and must stay below 4096");
    constexpr int NS = kInternal - 16;
    // 1. chain -> binary: tA[j] becomes THE plane of weight 16 << j
    {
        uint32_t ct = 0, cf = 0, cs = 0;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            if (pushed >> j) {
                csa(ct, s.tA[j], s.tA[j], s.tB[j], ct);
                csa(cf, s.fA[j], s.fA[j], s.fB[j], cf);
                csa(cs, s.sA[j], s.sA[j], s.sB[j], cs);
                s.tB[j] = s.fB[j] = s.sB[j] = 0;
            }
        }
    }
    // 2. planes -> counters
    uint32_t at[8], af[8], as[NS];
#pragma unroll
    for (int c = 0; c < 8; ++c) at[c] = af[c] = 0;
#pragma unroll
    for (int c = 0; c < NS; ++c) as[c] = 0;
    auto fold = [&](uint32_t pt, uint32_t pf, uint32_t ps, uint32_t w) {
        const uint32_t wb = w * 0x01010101u;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            at[c] = wdot(at[c], pt, 0x01010101u << c, wb);
            af[c] = wdot(af[c], pf, 0x01010101u << c, wb);
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) as[c] = wdot(as[c], ps, 0x01010101u << (c < 3 ? c : c + 3), wb);  // S bits 0,1,2 and 6,7
    };
#pragma unroll
    for (int j = DEPTH - 1; j >= 4; --j) {  // weights 256 << (j - 4), entered as 1 << (j - 4)
        if (pushed >> j) {
            fold(s.tA[j], s.fA[j], s.sA[j], 1u << (j - 4));
            s.tA[j] = s.fA[j] = s.sA[j] = 0;
        }
    }
    if (pushed >> 4) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            at[c] <<= 8;
            af[c] <<= 8;
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) as[c] <<= 8;
    }
#pragma unroll
    for (int j = 3; j >= 0; --j) {  // weights 16 << j
        if (pushed >> j) {
            fold(s.tA[j], s.fA[j], s.sA[j], 16u << j);
            s.tA[j] = s.fA[j] = s.sA[j] = 0;
        }
    }
    fold(s.t8, s.f8, s.s8, 8u);
    fold(s.t4, s.f4, s.s4, 4u);
    fold(s.t2, s.f2, s.s2, 2u);
    fold(s.t1, s.f1, s.s1, 1u);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        s.acc[c] += at[c] >> c;
        s.acc[8 + c] += af[c] >> c;
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) s.acc[16 + c] += as[c] >> (c < 3 ? c : c + 3);
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
// Result hand-over to a host thread that polls instead of waiting on the stream (the small-call path; a synchronous
// stream round trip is ~20 us here, profiles/r02/latency_breakdown.log).  Thread t < 32 stores {slot value, sequence
// number of the call} as ONE 16-byte store into pairs[t] (pinned host memory): a slot is complete when its sequence
// field matches, so no fence, no separate "done" word and no second bus round trip order the two (an aligned 16-byte
// store is a single write on the bus, and the host reads it back with one aligned 16-byte load).
__device__ __forceinline__ void store_pair(const HostSignal& sig, uint64_t value)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    v.x = static_cast<uint32_t>(value);
    v.y = static_cast<uint32_t>(value >> 32);
    v.z = static_cast<uint32_t>(sig.value);
    v.w = static_cast<uint32_t>(sig.value >> 32);
    // sc0 sc1 = system scope: written through to the host now, not at the kernel's end
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(reinterpret_cast<u32x4*>(sig.pairs) + threadIdx.x), "v"(v) : "memory");
}

// what thread t < 32 contributes to slot t
__device__ __forceinline__ uint64_t slot_value(const uint64_t* tot, int mode, uint64_t n_flags)
{
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
    return add;
}

template <bool ATOMIC = false>
__device__ __forceinline__ void finalize_slots(const uint64_t* tot, uint64_t* __restrict__ out, int mode, uint64_t n_flags)
{
    if (threadIdx.x < 32) {
        const uint64_t add = slot_value(tot, mode, n_flags);
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

// K1's direct epilogue for grids of many workgroups.  256 workgroups adding to the caller's two cache lines at the
// same moment is what a mid-size launch ends with, and those lines then bounce between the 8 XCDs' L2s once per
// request: ~10 ns each, 3.4-4.6 us of every launch with HBM idle (profiles/r03/launch_anatomy_*.log: the same adds
// spread over 8 per-XCD copies cost nothing measurable).  So the adds go in two levels: workgroup b adds its slots to
// copy b % 8 of the workspace (blockIdx.x % 8 is the XCD a workgroup runs on, so a copy's lines stay in ONE L2), waits
// until those adds have been performed (vmcnt counts an atomic until the L2 has done it), then draws a ticket of its
// group; the group's last workgroup swaps the copy's 32 words for zero -- which leaves the workspace ready for the
// next launch -- and adds them to the caller's out[32]: 8 x 2 contended requests per launch instead of 256 x 2.
// Every access to the copies and tickets is a device-scope atomic, so the result does not depend on the
// blockIdx -> XCD mapping, only the speed does.
__device__ __forceinline__ void grouped_epilogue(const uint64_t* tot, uint64_t* __restrict__ out, int mode, uint64_t n_flags,
                                                 uint64_t* __restrict__ block)
{
    if (threadIdx.x >= 64) return;  // wave 0 (every lane of it, so the scalar branches below are wave-uniform)
    const uint32_t g = blockIdx.x & 7u;
    const uint32_t members = (gridDim.x - g + 7u) >> 3;  // workgroups b < gridDim.x with b % 8 == g
    uint64_t* copy = block + kGroupCopyWord + 32 * g;
    uint64_t* ticket = block + kGroupTicketWord + 16 * g;
    const uint64_t add = threadIdx.x < 32 ? slot_value(tot, mode, n_flags) : 0;
    if (add) (void)__hip_atomic_fetch_add(&copy[threadIdx.x], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's adds are in the copy
    uint32_t t = 0;
    if (threadIdx.x == 0) t = static_cast<uint32_t>(__hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    t = __builtin_amdgcn_readfirstlane(t);
    if (t != members - 1) return;
    // last of the group: every member's adds were performed before its ticket add, and all of those before this one
    if (threadIdx.x < 32) {
        const uint64_t v = __hip_atomic_exchange(&copy[threadIdx.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) (void)__hip_atomic_fetch_add(&out[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ K1
// USTRIDE = vectors between a lane's consecutive loads: 64 -> each wave owns a contiguous
// 8 KiB of the step; 256 -> the 4 waves interleave at 1 KiB (each load instruction of the
// workgroup covers a contiguous 4 KiB).
template <bool NT, int USTRIDE, int VPS = kVecPerStep>
__device__ __forceinline__ void load_step(uint4 (&v)[kUnroll], const uint4* __restrict__ a0, uint64_t st, uint64_t lane_off,
                                          uint64_t lo, uint64_t hi, uint64_t fast_begin, uint64_t fast_end)
{
    const uint64_t j0 = st * VPS + lane_off;
    if (st >= fast_begin && st < fast_end) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = load_vec<NT>(a0 + j0 + u * USTRIDE);
    } else {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = load_guarded(a0, j0 + u * USTRIDE, lo, hi);
    }
}

}  // namespace fsk

#endif
