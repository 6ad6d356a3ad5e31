// flagstat_lz4_kernels.hip -- LZ4 block decode ON the GPU (row f1: the reference decodes every block with liblz4's
// LZ4_decompress_safe on the host, benchmark/flagstats.cpp:311-316).  Device code only; the host orchestration is
// flagstat_gpu_decode.hip.
//
// Two kernels:
//
// lz4_decode_wg (r04, default) -- ONE WORKGROUP OF EIGHT WAVES PER BLOCK, the whole 64 KiB LZ4 window in LDS.
//   An LZ4 block is a serial chain twice over: the position of token k + 1 depends on token k, and a match may read what
//   the match before it wrote.  r03's kernel walked both chains with one wave, one to four sequences per LDS round trip
//   (216 cycles per sequence, 19-32 ms per block).  Here the two chains are taken apart, each is walked in units the
//   hardware is wide enough for, and the stages run as a pipeline of waves:
//     wave 0, WALK: which input bytes are tokens.  One lane per 32-byte segment of a 2 KiB tile: backward over its
//       segment every lane computes where a chain entering at position e = 0..11 leaves it (a sliding window of 5-bit
//       "exits" in three registers: no memory, no dependence on where the chain is); the real chain through the tile is
//       then 64 scalar table look-ups, straight-line; forward again every lane marks the positions reachable from its
//       segment's entry -- the tokens -- and adds up their output bytes; a prefix sum over the lanes gives every 64-byte
//       window its output position.  One record per window goes to the emitters.
//     waves 1-3, EMIT: lane = byte position of a window; a DPP prefix sum places every sequence; ONE 32-bit marker (the
//       offset) goes to the match's first output byte, a literal run gets a "literal" marker and its bytes go straight
//       into the output ring.  No copies, no dependence on decoded data.
//     waves 4-6, SCAN: per 256-byte chunk of OUTPUT (4 bytes a lane) a DPP "last marker" scan turns the markers into one
//       source pointer per byte; pointers that land inside the chunk itself are chased to their roots by pointer
//       doubling on packed byte indices (ds_bpermute, data-independent: a byte -> byte pointer composes without the
//       bytes), so that every byte ends up with the ring index of a byte that is FINAL before the chunk starts.
//     wave 7, COPY: per chunk one 16-byte read of four final sources, four byte gathers from the ring, one aligned
//       4-byte write: 256 output bytes per LDS round trip, whatever the sequences were; every KiB the ring goes to
//       global memory with 16-byte stores.
//   The ring holds 64 KiB + 2 KiB: LZ4 offsets reach at most 65,535 bytes back, so NO match ever reads global memory
//   (r03: 9 % / 25 % of the sequences of an LZ4-fast / HC-9 flag stream went behind its 8 KiB ring, a global round trip
//   each) and the 2 KiB are what the emitters may write ahead of the copier.  The waves hand over through a few words in
//   LDS (positions reached, a 16-entry record queue) polled with s_sleep back-off; every LDS operation of a wave executes
//   in order, so "data, then position" needs no fence.  80 KB of LDS per block: two blocks per CU, 512 in flight -- a
//   block takes 2-3.5 ms instead of 19-32.  What shaped the code: a taken branch costs a wave ~50 cycles and a scalar
//   instruction ~12 (sixteen waves share a CU's scalar unit) where a straight-line VALU instruction costs 5.5, so the hot
//   paths are straight-line (selects, idle lanes storing to a scratch word) and lane-parallel wherever a serial scalar
//   walk could be turned into one.  Every index is masked, clamped or checked; a malformed block sets its status word and
//   the eight waves leave through the same barrier; every wait is bounded.
//   tests/test_lz4_walker_model.py restates the walker's algorithm in Python against liblz4's token chains.
//
// lz4_decode_wave (r03) -- one wave per block, 8 KiB ring; kept as the yardstick (knob "lz4_gpu_kernel" = 1).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "flagstat_lz4_kernels.h"
#include "flagstat_wgpipe.h"

namespace fsk {

// ------------------------------------------------------------------------------------------------ lz4_decode_wg
// waves of a workgroup: 0 walk (token chain), then kWgEmit emitters (records r mod kWgEmit), kWgScan scanners (chunks k mod
// kWgScan), one copier
constexpr uint32_t kWgEmit = kLz4WgEmitters, kWgScan = kLz4WgScanners;
constexpr uint32_t kWgThreads = 64u * (1u + kWgEmit + kWgScan + 1u);
constexpr uint32_t kWgNR = 67584;                  // output ring: 64 KiB window + 2 KiB of write-ahead (66 x 1024, 264 x 256)
constexpr uint32_t kWgChunk = 256;                 // output bytes per scan / copy step (4 per lane)
constexpr uint32_t kWgAhead = kWgNR - 65536 - kWgChunk;  // the emitters' position may lead the copier's by this much
constexpr uint32_t kWgMR = 1024;                   // marker slots: the emitters may lead the scanners by this many output bytes
constexpr uint32_t kWgK = 512;                     // final-source slots: the scanners may lead the copier by this many
constexpr uint32_t kWgInw = 6144, kWgInPad = 96;   // input ring (6 x 1 KiB) + mirror of its first bytes behind its end
constexpr uint32_t kWgSeg = 32, kWgTileSegs = 64;  // the walker's tile: 64 segments of 32 input bytes, one lane each
constexpr uint32_t kWgQ = 16;                      // records the walker may lead the emitters by (each <= 96 input bytes)
constexpr uint32_t kWgSpan = 384;                  // most output bytes the emitters write before publishing (more: in batches)
constexpr uint32_t kWgFlush = 1024;
enum { REC_WINDOW = 0, REC_SEQ = 1, REC_END = 2 };
static_assert(kWgThreads == 512 && kWgEmit == 3 && kWgScan == 3, "two waves per SIMD; three positions and one more word per 16-byte poll");
static_assert(kWgNR % kWgFlush == 0 && kWgNR % kWgChunk == 0 && kWgMR % kWgChunk == 0 && kWgK % kWgChunk == 0, "grids");
static_assert(kWgSpan + kWgChunk <= kWgMR && kWgSpan + kWgChunk + kWgK <= kWgAhead + kWgChunk, "no cyclic wait");
static_assert(kWgSeg * kWgTileSegs + kWgInPad + 1023u + kWgQ * 96u < kWgInw - 1024u, "the walker cannot overwrite input an emitter still reads");

struct __attribute__((aligned(16))) WgLds {
    static constexpr uint32_t kNR = kWgNR, kMR = kWgMR, kK = kWgK, kChunk = kWgChunk, kFlush = kWgFlush, kScan = kWgScan;
    static constexpr bool kPublishFlush = false;
    uint8_t ring[kWgNR];
    uint32_t mark[kWgMR];          // per output byte: 0, the offset of the match that starts there, or kMarkLiteral
    uint32_t fsrc[kWgK];           // per output byte: ring index of the byte it is a copy of (final before its chunk starts)
    uint8_t inw[kWgInw + kWgInPad + 16];
    uint4 q[kWgQ];                 // walker -> emitters: x = kind << 28 | output position, then WINDOW: y = input position, z / w = member lanes;
                                   // SEQ: y = position of the literals | their count (<= 64) << 24, z = offset, w = match length (0: no match)
                                   // (input positions as indices into inw)
    uint32_t scratch[64];               // where the emitters' idle lanes write (straight-line stores instead of branches)
    uint32_t q_tail[kWgEmit], q_head;   // per emitter the next record it will take; records pushed (16 bytes: one read)
    uint32_t s_clr[kWgScan], d_op;      // per scanner: start of the next chunk it will clear; copier: end of the last chunk copied (16 bytes)
    uint32_t s_done[kWgScan];           // per scanner: end of the last chunk whose final sources it has written
    uint32_t p_walk;                    // output position of the first record not pushed yet (written after q_head)
    uint32_t c_ready, s_carry[8], err;
};

// (hand-over primitives, wave scans, the SCAN and COPY stages: flagstat_wgpipe.h)

// ---- wave 0: the token chain.  Stages the input and finds which byte positions are tokens -- with ONE LANE PER 32-BYTE
// SEGMENT of a 2 KiB tile, so that all 64 lanes work on every step (a lane per byte position, as the emitters sit, leaves
// two lanes in three idle and pays ~100 scalar instructions per 64 bytes):
//   (1) backward over the 32 positions of its segment, every lane computes where a chain that enters the segment at
//       position e = 0..19 leaves it ("exit": the position in the NEXT segment it lands on), from a sliding 18-entry
//       window of 5-bit exits held in three registers -- no memory, no dependence on where the chain really is;
//   (2) the real chain through the tile is then 64 scalar table lookups (entry of segment s + 1 = exit table of s at the
//       entry of s);
//   (3) forward over the 32 positions again, every lane marks the tokens reachable from its segment's entry: the members.
// Two segments make one 64-byte WINDOW record for the emitters.  A token whose literal length continues in further bytes
// (or whose match length needs more than one further byte) ends the tile there and goes through the scalar code.
template <bool PROF>
__device__ void lz4wg_walk(WgLds& L, const uint8_t* __restrict__ src, const uint32_t iend, const uint32_t oend,
                           const uint32_t lane, unsigned long long* __restrict__ tally)
{
    uint32_t ip = 0, ip_r = 0, in_hi = 0, in_hi_r = 0, err = 0, nrec = 0, tail_seen = 0, op = 0;
    unsigned long long t_wait = 0, n_win = 0, n_seq = 0, n_tile = 0, t_dp = 0, t_chain = 0, t_reach = 0, t_push = 0, t_mark = 0;
    auto tock = [&](unsigned long long& acc) {
        if (PROF) {
            const unsigned long long now = __builtin_readcyclecounter();
            acc += now - t_mark;
            t_mark = now;
        }
    };
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    // input: a ring of six KiB filled one KiB at a time, the next TWO KiB always in flight in registers.  What a commit
    // overwrites lies > 4 KiB behind the staged end, i.e. > 1.9 KiB behind ip, and the emitters are at most kWgQ records
    // (of <= 96 input bytes) behind.  The ring's first 96 bytes are mirrored behind its end: no read wraps.
    // (loads unconditional, the offset clamped instead: a load under a condition made the compiler wait for it on the
    // spot.  At most 16 bytes from iend on are read: the image is padded by 64)
    uint4 pend_a = *reinterpret_cast<const uint4*>(src + umin(lane * 16u, iend));
    uint4 pend_b = *reinterpret_cast<const uint4*>(src + umin(1024u + lane * 16u, iend));
    auto cover = [&](uint32_t need) {
        while (in_hi < iend && in_hi < ip + need) {
            *reinterpret_cast<uint4*>(&L.inw[in_hi_r + lane * 16u]) = pend_a;
            if (in_hi_r == 0u && lane < kWgInPad / 16u) *reinterpret_cast<uint4*>(&L.inw[kWgInw + lane * 16u]) = pend_a;
            in_hi += 1024u;
            in_hi_r = in_hi_r + 1024u == kWgInw ? 0u : in_hi_r + 1024u;
            pend_a = pend_b;
            pend_b = *reinterpret_cast<const uint4*>(src + umin(in_hi + 1024u + lane * 16u, iend));
        }
    };
    auto advance = [&](uint32_t n) {
        ip += n;
        ip_r += n;
        while (ip_r >= kWgInw) ip_r -= kWgInw;
    };
    auto inb = [&]() -> uint32_t { return __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(L.inw[ip_r])); };
    auto have = [&](uint32_t n) -> bool {
        if (nrec + n <= tail_seen + kWgQ) return true;
        return wg_wait_timed<PROF>(L, t_wait, [&] {
            const uint4 t = wg_ld4(L.q_tail);
            tail_seen = umin3(t.x, t.y, t.z);
            return nrec + n <= tail_seen + kWgQ;
        });
    };
    auto push = [&](uint32_t kind, uint32_t at, uint32_t y, uint32_t z, uint32_t w) -> bool {
        if (!have(1u)) return false;
        if (lane == 0u) L.q[nrec & (kWgQ - 1u)] = make_uint4((kind << 28) | at, y, z, w);
        ++nrec;
        wg_st(&L.q_head, nrec);
        wg_st(&L.p_walk, kind == REC_SEQ ? at + (y >> 24) + w : at);  // (where this record's output ends)
        return true;
    };
    // every queued record has been taken (before the walker runs far ahead of input that a record still refers to)
    auto drain = [&]() -> bool {
        return wg_wait_timed<PROF>(L, t_wait, [&] {
            const uint4 t = wg_ld4(L.q_tail);
            return umin3(t.x, t.y, t.z) >= nrec;
        });
    };
    // one sequence of any shape at ip, scalar: its literals in pieces of <= 64 bytes (by reference), the last piece with
    // the match (offset and length by value).  No record makes the walker advance more than 96 input bytes.
    auto slow_sequence = [&]() {
        ++n_seq;
        cover(96u);
        const uint32_t token = inb();
        advance(1u);
        uint32_t ll = token >> 4;
        if (ll == 15u) {
            uint32_t e, n = 0;
            do {
                cover(96u);
                if (ip >= iend) { err = 1; return; }
                if (++n == 8u && !drain()) { err = 8; return; }
                e = inb();
                advance(1u);
                ll += e;
                if (ll > oend) { err = 2; return; }
            } while (e == 255u);
        }
        if (ll > iend - ip || ll > oend - op) { err = 2; return; }
        while (ll > 64u) {
            cover(96u);
            if (!push(REC_SEQ, op, ip_r | (64u << 24), 0u, 0u)) { err = 8; return; }
            advance(64u);
            op += 64u;
            ll -= 64u;
        }
        cover(96u);
        const uint32_t lit_at = ip_r;
        advance(ll);
        if (ip >= iend) {  // the last sequence has no match
            if (!push(REC_SEQ, op, lit_at | (ll << 24), 0u, 0u)) err = 8;
            op += ll;
            return;
        }
        if (ip + 2u > iend) { err = 3; return; }
        uint32_t off = inb();
        advance(1u);
        off |= inb() << 8;
        advance(1u);
        if (off == 0u) { err = 5; return; }
        uint32_t ml = token & 15u;
        if (ml == 15u) {
            // (the literals of this sequence go first: the length bytes may be many)
            if (ll && !push(REC_SEQ, op, lit_at | (ll << 24), 0u, 0u)) { err = 8; return; }
            op += ll;
            ll = 0;
            uint32_t e, n = 0;
            do {
                cover(96u);
                if (ip >= iend) { err = 4; return; }
                if (++n == 8u && !drain()) { err = 8; return; }
                e = inb();
                advance(1u);
                ml += e;
                if (ml > oend) { err = 5; return; }
            } while (e == 255u);
        }
        ml += 4u;
        if (ll + ml > oend - op) { err = 5; return; }
        if (!push(REC_SEQ, op, lit_at | (ll << 24), off, ml)) err = 8;
        op += ll + ml;
    };

    if (iend >= (1u << 24) || oend >= (1u << 28)) err = 10;  // (records carry 24-bit input and 28-bit output positions; the format's blocks are 1,024,000 bytes)
    while (!err) {
        // segments whose every position, were it a token of a sequence with all its lengths in the token (<= 19 bytes),
        // would have its offset inside the block: such a sequence cannot be the block's last one
        const uint32_t nseg = iend >= ip + 50u ? umin(kWgTileSegs, (iend - ip - 50u) / kWgSeg + 1u) : 0u;
        if (nseg == 0u) break;  // the block's last bytes: one sequence at a time below
        cover(nseg * kWgSeg + kWgInPad);
        ++n_tile;
        uint32_t base = ip_r + lane * kWgSeg;
        if (base >= kWgInw) base -= kWgInw;
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_memcpy(&w[k], &L.inw[base + 4u * k], 4);
        // ---- (1) exits: ex(i) = where a chain through position i lands in the next segment; 31 = through a token the
        // window form does not cover.  win0..2 = ex(i + 1 .. i + 18), 5 bits each; tab0..3 = ex(0 .. 19).
        uint32_t win0 = 0, win1 = 0, win2 = 0, tab0 = 0, tab1 = 0;
        if (PROF) t_mark = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            const uint32_t tok = (w[i >> 2] >> ((i & 3) * 8)) & 255u;
            const uint32_t ll = tok >> 4;
            const uint32_t d = 3u + ll + ((tok & 15u) == 15u ? 1u : 0u);  // length of the sequence, if its lengths end here (3..18)
            const uint32_t j = d - 1u;
            const uint32_t word = j < 6u ? win0 : (j < 12u ? win1 : win2);
            const uint32_t slot = j < 6u ? j : (j < 12u ? j - 6u : j - 12u);
            uint32_t ex = (word >> (slot * 5u)) & 31u;
            if (d >= 32u - static_cast<uint32_t>(i)) ex = static_cast<uint32_t>(i) + d - 32u;
            if (ll == 15u) ex = 31u;
            win2 = ((win2 << 5) | (win1 >> 25)) & 0x3FFFFFFFu;
            win1 = ((win1 << 5) | (win0 >> 25)) & 0x3FFFFFFFu;
            win0 = ((win0 << 5) | ex) & 0x3FFFFFFFu;
            if (i < 12) {
                const uint32_t put = ex << ((i % 6) * 5);
                if (i / 6 == 0) tab0 |= put;
                if (i / 6 == 1) tab1 |= put;
            }
        }
        // ex(0 .. 11) as one 60-bit word (a chain enters a segment at 0..17, at 12 or more only behind nine or more literals:
        // the tile then ends with the segment before, which costs nothing but a short tile)
        const uint32_t tlo = tab0 | (tab1 << 30), thi = tab1 >> 2;
        // (the two unrolled passes share nothing but w[]: without this the compiler keeps 32 decoded tokens alive, 149 VGPRs)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(w[k]));
        tock(t_dp);
        // ---- (2) the chain through the tile: entry of every segment.  r04 walked it: 64 dependent scalar table lookups, 186 cycles
        // each = 12 k of a tile's 27 k cycles, on the one wave every byte of the block waits for.  r05: a prefix scan over the
        // segments' exit tables instead -- lane s ends up with F(s) = T(s) o ... o T(0), six doubling rounds in which every lane
        // composes its table with the one 1, 2, 4 ... 32 lanes to its left (twelve 5-bit entries, looked up in a 60-bit word; an
        // exit of 12 or more -- the tile ends behind that segment -- stays what it is through every later table).  F(s)(0) is the
        // exit of segment s for the chain that enters the tile at 0, the entry of segment s + 1; the first lane whose exit is 12 or
        // more, or the tile's last segment, ends the tile.
        uint32_t flo = tlo, fhi = thi;
        {
            constexpr uint64_t kIdentity = 0ull | (1ull << 5) | (2ull << 10) | (3ull << 15) | (4ull << 20) | (5ull << 25) | (6ull << 30) | (7ull << 35) |
                                           (8ull << 40) | (9ull << 45) | (10ull << 50) | (11ull << 55);
#pragma unroll
            for (uint32_t d = 1; d < kWgTileSegs; d <<= 1) {
                const int from = static_cast<int>((lane - d) << 2);
                uint32_t glo = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(from, static_cast<int>(flo)));
                uint32_t ghi = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(from, static_cast<int>(fhi)));
                glo = lane < d ? static_cast<uint32_t>(kIdentity) : glo;
                ghi = lane < d ? static_cast<uint32_t>(kIdentity >> 32) : ghi;
                const uint64_t g = static_cast<uint64_t>(glo) | (static_cast<uint64_t>(ghi) << 32);
                const uint64_t f = static_cast<uint64_t>(flo) | (static_cast<uint64_t>(fhi) << 32);
                uint64_t n = 0;
#pragma unroll
                for (uint32_t en = 0; en < 12u; ++en) {
                    const uint32_t v = static_cast<uint32_t>(g >> (5u * en)) & 31u;      // where the tables to the left leave a chain that entered them at `en`
                    const uint32_t t = static_cast<uint32_t>(f >> ((5u * v) & 63u)) & 31u;   // ... and this lane's tables take it from there (v >= 12: not used)
                    n |= static_cast<uint64_t>(v >= 12u ? v : t) << (5u * en);
                }
                flo = static_cast<uint32_t>(n);
                fhi = static_cast<uint32_t>(n >> 32);
            }
        }
        const uint32_t e2 = flo & 31u;   // exit of this lane's segment for the chain that entered the tile at 0 (>= 12: the tile has ended at or before it)
        const uint32_t ent_shifted = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(e2), 0x138, 0xF, 0xF, false));  // wave_shr:1 (lane 0: 0)
        const uint64_t endm = __builtin_amdgcn_ballot_w64((e2 >= 12u) | (lane + 1u >= nseg));
        const uint32_t last_seg = static_cast<uint32_t>(__builtin_ctzll(endm));   // (nseg >= 1: lane nseg - 1 is in the mask)
        uint32_t nvalid = last_seg + 1u;
        const uint32_t e_next = __builtin_amdgcn_readlane(e2, last_seg);
        const uint32_t ent = ent_shifted;   // (lanes up to last_seg: below 12; the others are not used)
        tock(t_chain);
        // ---- (3) members: the positions reachable from the entry; their output bytes; straight-line
        uint32_t r_lo = lane < nvalid ? 1u << ent : 0u, r_hi = 0;  // reach: bit i = position i is a token (bits >= 32: in the next segment)
        uint32_t stop_pos = 32u, olen = 0, c15m = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint32_t tok = (w[i >> 2] >> ((i & 3) * 8)) & 255u;
            const uint32_t ll = tok >> 4, mlc = tok & 15u;
            const uint32_t d = 3u + ll + (mlc == 15u ? 1u : 0u);
            const bool on = (r_lo >> i) & 1u;
            const bool go = on & (ll != 15u);
            stop_pos = (on & (ll == 15u)) ? umin(stop_pos, static_cast<uint32_t>(i)) : stop_pos;
            c15m |= (go & (mlc == 15u)) ? 1u << i : 0u;
            olen += go ? ll + mlc + 4u : 0u;
            const uint32_t step = go ? 1u << d : 0u;  // bit d <= 18
            r_lo |= step << i;
            if (i > 13) r_hi |= step >> (32 - i);
        }
        (void)r_hi;
        // a match length byte was taken to be the only one: it is if it is not 255 (and it adds to the output)
        {
            uint32_t chk = c15m;
            while (__builtin_amdgcn_ballot_w64(chk != 0u)) {
                const uint32_t i = chk ? static_cast<uint32_t>(__builtin_ctz(chk)) : 0u;
                const uint32_t tk = L.inw[base + i];
                const uint32_t extb = L.inw[base + i + 3u + (tk >> 4)];  // (<= 31 + 17 behind base: inside the mirror)
                if (chk) {
                    if (extb == 255u) stop_pos = umin(stop_pos, i);
                    olen += extb;  // (19 + ext; the 19 = 15 + 4 is counted above)
                }
                chk &= chk - 1u;
            }
        }
        uint32_t members = r_lo;
        tock(t_reach);
        const uint64_t stopm = __builtin_amdgcn_ballot_w64((stop_pos < 32u) & (lane < nvalid));
        uint32_t adv;
        bool slow = false;
        if (stopm) {
            // the tile ends inside segment ls: its sequences before the stop are still taken here; what the passes above
            // counted for this lane behind the stop, and for all later lanes, is void
            const uint32_t ls = static_cast<uint32_t>(__builtin_ctzll(stopm));
            nvalid = ls + 1u;
            adv = ls * kWgSeg + __builtin_amdgcn_readlane(stop_pos, ls);
            slow = true;
            if (lane == ls) {
                // recount this lane's output up to the stop
                members &= (1u << stop_pos) - 1u;
                olen = 0;
                uint32_t m = members;
                while (m) {
                    const uint32_t i = static_cast<uint32_t>(__builtin_ctz(m));
                    const uint32_t tk = L.inw[base + i];
                    const uint32_t ll = tk >> 4, mlc = tk & 15u;
                    olen += ll + (mlc == 15u ? 19u + static_cast<uint32_t>(L.inw[base + i + 3u + ll]) : mlc + 4u);
                    m &= m - 1u;
                }
            }
        } else if (e_next >= 18u) {
            err = 7;  // (cannot happen: the chain saw a token for the scalar code that the members pass did not)
            break;
        } else {
            adv = nvalid * kWgSeg + e_next;
        }
        if (lane >= nvalid) {
            members = 0u;
            olen = 0u;
        }
        // ---- output positions: a prefix sum over the segments; window t = segments 2t, 2t + 1 starts where 2t does
        const uint32_t oincl = wave_scan_add(olen);
        const uint32_t tile_out = __builtin_amdgcn_readlane(oincl, 63);
        if (tile_out > oend - op) {
            err = 5;
            break;
        }
        const uint32_t wop = op + oincl - olen;
        // ---- records, eight at a time
        const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(members), 0xF5, 0xF, 0xF, false));  // quad_perm:[1,1,3,3]
        const uint32_t nw = (nvalid + 1u) / 2u;
        bool ok = true;
        for (uint32_t g0 = 0; g0 < nw; g0 += 8u) {
            const uint32_t cnt = umin(8u, nw - g0);
            if (!have(cnt)) {
                ok = false;
                break;
            }
            const uint32_t t = (lane >> 1) - g0;
            if (!(lane & 1u) && (lane >> 1) >= g0 && t < cnt)
                L.q[(nrec + t) & (kWgQ - 1u)] = make_uint4((static_cast<uint32_t>(REC_WINDOW) << 28) | wop, base, members, hi);
            nrec += cnt;
            wg_st(&L.q_head, nrec);
            wg_st(&L.p_walk, g0 + cnt < nw ? __builtin_amdgcn_readlane(wop, 2u * (g0 + cnt)) : op + tile_out);
        }
        if (!ok) {
            err = 8;
            break;
        }
        if (PROF) n_win += nw;
        op += tile_out;
        tock(t_push);
        advance(adv);
        if (slow) slow_sequence();
    }
    while (!err && ip < iend) slow_sequence();
    if (!err && (ip != iend || op != oend)) err = 6;
    if (err) wg_st(&L.err, err);
    if (!err)
        for (uint32_t i = 0; i < kWgEmit; ++i) (void)push(REC_END, op, 0u, 0u, 0u);  // one for each emitter
    if (PROF && lane == 0u) {
        atomicAdd(&tally[2], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        atomicAdd(&tally[3], t_wait);
        atomicAdd(&tally[5], n_win);
        atomicAdd(&tally[6], n_seq);
        atomicAdd(&tally[7], n_tile);
        atomicAdd(&tally[17], t_dp);
        atomicAdd(&tally[18], t_chain);
        atomicAdd(&tally[19], t_reach);
        atomicAdd(&tally[20], t_push);
    }
}

// ---- emitters: records -> markers, literal bytes.  Emitter `which` takes records which, which + kWgEmit, ...; a record
// carries its output position, so the emitters do not depend on each other.  The window path is straight-line: lanes
// with nothing to store write to a scratch word of their own instead of branching around the store.
template <bool PROF>
__device__ void lz4wg_emit(WgLds& L, const uint32_t oend, const uint32_t lane, const uint32_t which,
                           unsigned long long* __restrict__ tally)
{
    uint32_t err = 0, nseq = 0, s_seen = 0, d_seen = 0, head_seen = 0;
    unsigned long long t_wait = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    // room for output positions [at, at + n): marker slots the scanners have cleared, ring bytes the copier no longer reads
    auto room = [&](uint32_t at, uint32_t n) -> bool {
        if (__builtin_expect(at + n <= s_seen + kWgMR && at + n <= d_seen + kWgAhead, 1)) return true;
        return wg_wait_timed<PROF>(L, t_wait, [&] {
            const uint4 t = wg_ld4(L.s_clr);
            s_seen = umin3(t.x, t.y, t.z);
            d_seen = t.w;
            return at + n <= s_seen + kWgMR && at + n <= d_seen + kWgAhead;
        });
    };
    uint32_t* const idle = &L.scratch[lane];
    for (uint32_t r = which;; r += kWgEmit) {
        if (__builtin_expect(head_seen <= r, 0)) {
            if (!wg_wait_timed<PROF>(L, t_wait, [&] {
                    head_seen = wg_ld(&L.q_head);
                    return head_seen > r;
                }))
                break;
        }
        const uint4 rec = L.q[r & (kWgQ - 1u)];
        const uint32_t rx = __builtin_amdgcn_readfirstlane(rec.x), ry = __builtin_amdgcn_readfirstlane(rec.y);
        const uint32_t rz = __builtin_amdgcn_readfirstlane(rec.z), rw = __builtin_amdgcn_readfirstlane(rec.w);
        asm volatile("" ::: "memory");
        const uint32_t kind = rx >> 28, op = rx & 0xFFFFFFFu;
        if (__builtin_expect(kind == REC_WINDOW, 1)) {
            uint32_t wi = ry + lane;  // (ry: index into inw of the window's first byte)
            if (wi >= kWgInw) wi -= kWgInw;
            uint64_t q;
            __builtin_memcpy(&q, &L.inw[wi], 8);  // token + the first 7 bytes behind it (all the literals of most sequences)
            const uint32_t tok = static_cast<uint32_t>(q) & 255u;
            const uint32_t ll = tok >> 4;
            uint32_t ob;
            __builtin_memcpy(&ob, &L.inw[wi + 1u + ll], 4);  // offset, first match length byte
            const uint32_t offv = ob & 0xFFFFu;
            const uint32_t ml = (tok & 15u) == 15u ? 19u + ((ob >> 16) & 255u) : (tok & 15u) + 4u;  // (the walker made sure that byte is not 255)
            const uint32_t lits = static_cast<uint32_t>(q >> 8);
            const bool member = ((static_cast<uint64_t>(rz) | (static_cast<uint64_t>(rw) << 32)) >> lane) & 1ull;
            const uint32_t len = member ? ll + ml : 0u;
            const uint32_t incl = wave_scan_add(len);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            nseq += static_cast<uint32_t>(__builtin_popcount(rz) + __builtin_popcount(rw));
            const uint32_t rel = incl - len;           // this sequence's first output byte, relative to op
            const uint32_t mpos = op + rel + ll;       // ... and its match's
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(member & ((offv == 0u) | (offv > mpos))) != 0ull || total > oend - op, 0)) {
                err = 5;
                break;
            }
            const uint32_t op_idx = op % kWgNR;
            uint32_t ri = op_idx + rel;
            if (ri >= kWgNR) ri -= kWgNR;
            const bool haslit = member & (ll > 0u);
            // whether the short form below covers this window: all of it fits before the next publication, no literal run
            // longer than 4 bytes, none across the end of the ring
            const bool plain = total <= kWgSpan && !__builtin_amdgcn_ballot_w64(haslit & ((ll > 4u) | (ri + 4u > kWgNR)));
            if (__builtin_expect(plain, 1)) {
                if (!room(op, total)) break;
                *(member ? &L.mark[mpos & (kWgMR - 1u)] : idle) = offv;
                *(haslit ? &L.mark[(op + rel) & (kWgMR - 1u)] : idle) = kMarkLiteral;
                // four bytes for a literal run of 1..4: what it writes too much lies in the sequence's own match (>= 4
                // bytes), which the copier writes later
                __builtin_memcpy(haslit ? static_cast<void*>(&L.ring[ri]) : static_cast<void*>(idle), &lits, 4);
            } else {
                // in batches of <= kWgSpan output bytes, literals byte by byte
                uint32_t done = 0;
                bool failed = false;
                while (done < total) {
                    const uint32_t upto = done + kWgSpan;
                    const bool now = member & (rel >= done) & (incl <= upto);
                    const uint64_t nowm = __builtin_amdgcn_ballot_w64(now);
                    const uint32_t last = 63u - static_cast<uint32_t>(__builtin_clzll(nowm));   // (never empty: one sequence is <= 287 bytes)
                    const uint32_t end = __builtin_amdgcn_readlane(incl, last);
                    if (!room(op + done, end - done)) {
                        failed = true;
                        break;
                    }
                    if (now) L.mark[mpos & (kWgMR - 1u)] = offv;
                    const bool nowlit = now & (ll > 0u);
                    if (nowlit) L.mark[(op + rel) & (kWgMR - 1u)] = kMarkLiteral;
                    uint32_t rj = ri;
                    uint64_t lb = q >> 8;
                    for (uint32_t j = 0; j < 14u; ++j) {
                        const bool a = nowlit & (j < ll);
                        if (!__builtin_amdgcn_ballot_w64(a)) break;
                        if (a) L.ring[rj] = j < 7u ? static_cast<uint8_t>(lb) : L.inw[wi + 1u + j];
                        lb >>= 8;
                        rj = rj + 1u == kWgNR ? 0u : rj + 1u;
                    }
                    done = end;
                    // (what is written so far, for the scanners: the record's position moves on)
                    if (lane == 0u) L.q[r & (kWgQ - 1u)].x = (static_cast<uint32_t>(REC_WINDOW) << 28) | (op + done);
                    asm volatile("" ::: "memory");
                }
                if (failed) break;
            }
        } else if (kind == REC_SEQ) {
            ++nseq;
            // literals [ry & 0xFFFFFF, + ll) (ll <= 64), then a match of ml bytes at distance rz (none: ml == 0)
            const uint32_t ll = ry >> 24, ml = rw, off = rz;
            if (ll + ml > oend - op || (ml && (off == 0u || off > op + ll))) {
                err = 5;
                break;
            }
            if (!room(op, ll + 1u)) break;
            const uint32_t op_idx = op % kWgNR;
            if (lane < ll) {
                uint32_t ri = op_idx + lane;
                if (ri >= kWgNR) ri -= kWgNR;
                uint32_t li = (ry & 0xFFFFFFu) + lane;
                if (li >= kWgInw) li -= kWgInw;
                L.ring[ri] = L.inw[li];
            }
            if (lane == 0u) {
                if (ll) L.mark[op & (kWgMR - 1u)] = kMarkLiteral;
                if (ml) L.mark[(op + ll) & (kWgMR - 1u)] = off;
            }
        } else {
            wg_st(&L.q_tail[which], r + kWgEmit);
            break;
        }
        wg_st(&L.q_tail[which], r + kWgEmit);
    }
    if (err) wg_st(&L.err, err);
    if (lane == 0u) {
        atomicAdd(&tally[0], static_cast<unsigned long long>(nseq));
        if (PROF) {
            atomicAdd(&tally[15], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
            atomicAdd(&tally[16], t_wait);
        }
    }
}


// (eight waves: two per SIMD, so that the two workgroups the LDS allows on a CU always fit side by side; with ten, 3 + 3
// waves of the two workgroups landed on one SIMD, more than 92 VGPRs allow, and a CU held ONE workgroup)
template <bool PROF>
__global__ __launch_bounds__(kWgThreads, 4) void lz4_decode_wg(const uint8_t* __restrict__ comp, const GpuBlock* __restrict__ blocks,
                                                            uint8_t* __restrict__ out, uint32_t* __restrict__ status,
                                                            unsigned long long* __restrict__ tally)
{
    __shared__ WgLds L;
    const GpuBlock b = blocks[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // marker slots must start out zero (the scanners leave them zero, but LDS is not cleared between workgroups); the
    // positions too
    for (uint32_t i = threadIdx.x; i < kWgMR; i += kWgThreads) L.mark[i] = 0u;
    if (threadIdx.x == 0u) {
        L.q_head = 0u;
        L.p_walk = 0u;
        for (uint32_t i = 0; i < kWgEmit; ++i) L.q_tail[i] = i;
        for (uint32_t i = 0; i < kWgScan; ++i) {
            L.s_clr[i] = i * kWgChunk;
            L.s_done[i] = 0u;
        }
        L.d_op = 0u;
        L.c_ready = 0u;
        L.err = 0u;
    }
    __syncthreads();
    if (role == 0u)
        lz4wg_walk<PROF>(L, comp + b.src_off, b.src_len, b.dst_len, lane, tally);
    else if (role <= kWgEmit)
        lz4wg_emit<PROF>(L, b.dst_len, lane, role - 1u, tally);
    else if (role <= kWgEmit + kWgScan) {
        // the position below which every marker and literal is in place: where the first record that is not finished starts
        // (the emitters finish their records in order each, so it is the smallest "next record" of any of them; its position
        // stands in its queue slot, which nobody reuses before it is finished) -- or, with every pushed record finished, where
        // the walker is.  p_walk is read first: it can only be too small for the records found finished afterwards.
        auto frontier = []() -> uint32_t {
            const uint32_t walk = wg_ld(&L.p_walk);
            for (;;) {
                const uint4 a = wg_ld4(L.q_tail);
                const uint32_t t = umin3(a.x, a.y, a.z);
                if (t >= a.w) return walk;
                const uint32_t at = wg_ld(&L.q[t & (kWgQ - 1u)].x) & 0xFFFFFFFu;
                const uint4 b = wg_ld4(L.q_tail);
                if (umin3(b.x, b.y, b.z) == t) return at;
            }
        };
        wgpipe_scan<PROF>(L, b.dst_len, lane, role - 1u - kWgEmit, frontier, tally);
    } else
        wgpipe_copy<PROF>(L, out + b.dst_off, b.dst_len, lane, tally);
    __syncthreads();  // every wave comes here, failed block or not
    if (threadIdx.x == 0u) status[blockIdx.x] = L.err;
}

// ---------------------------------------------------------------------------------------------- lz4_decode_wave (r03)
// ONE WAVE PER BLOCK:
//   * compressed bytes staged through a 1 KiB LDS window (coalesced 16-byte loads),
//   * the last 8 KiB of output kept in an LDS ring, so a match copy is ds_read -> ds_write for every offset up to
//     8,128; farther matches read the already flushed output back from global memory,
//   * up to 16 bare sequences (3 input bytes each) parsed AT ONCE by 16 lanes, output positions by a DPP prefix sum,
//   * copied in PASSES of up to four sequences that do not read each other's output: one LDS read and one write for
//     all of them, each on a row of 16 lanes, their words gathered through a small LDS table read one pass ahead,
//   * the ring flushed to global memory 2 KiB at a time with coalesced 16-byte stores,
//   * every index masked or checked: a malformed block sets its status word and stops, it cannot fault.
// 9.3 KiB of LDS per wave = 17 waves per CU.  216 cycles and 18.3 instructions per sequence (profiles/r03/lz4_gpu_pmc.txt).
// RING: bytes of recent output kept in LDS (matches up to RING - 64 back are LDS -> LDS); INWIN: staged input window.
// 8 KiB + 1 KiB (the default) = 17 waves per CU (4352 blocks in flight: a 4 GiB file's 4195 blocks all at once);
// 16 KiB + 4 KiB (env FLAGSTATS_HIP_GPU_LZ4_RING=16, tuning) = 7 per CU, fewer matches behind the ring, measured slower.
template <uint32_t RING, uint32_t INWIN, bool PROF = false>
__global__ __launch_bounds__(64) void lz4_decode_wave(const uint8_t* __restrict__ comp, const GpuBlock* __restrict__ blocks,
                                                     uint8_t* __restrict__ out, uint32_t* __restrict__ status,
                                                     unsigned long long* __restrict__ tally)
{
    constexpr uint32_t kRingMask = RING - 1, kFlush = RING / 4;
    static_assert(RING <= 65536, "pass rows keep a ring offset in 16 bits");
    constexpr uint32_t kScratch = RING + INWIN + 16;  // 64 bytes nobody reads: where idle lanes of a copy pass point
    constexpr uint32_t kRows = kScratch + 64;          // 20 x 2 words + 20 words: the batch's per-sequence rows and far sources
    constexpr uint32_t kTabN = 20;                     // (16 sequences + the 3 entries a pass may read past them; 17 waves per CU
    __shared__ __attribute__((aligned(16))) uint8_t lds[kRows + kTabN * 12];  //  need <= 9637 bytes per wave: this is 9536)
    uint2* const tab_row = reinterpret_cast<uint2*>(lds + kRows);
    uint32_t* const tab_far = reinterpret_cast<uint32_t*>(lds + kRows + kTabN * 8);
    uint8_t* const ring = lds;
    uint8_t* const inw = lds + RING;
    const GpuBlock b = blocks[blockIdx.x];
    const uint8_t* src = comp + b.src_off;
    uint8_t* dst = out + b.dst_off;
    const uint32_t iend = b.src_len, oend = b.dst_len;
    const uint32_t lane = threadIdx.x;
    uint32_t ip = 0, op = 0, in_base = 0, in_valid = 0, flushed = 0;
    uint32_t err = 0, nseq = 0, nfar = 0;
    // PROF: wave cycles per phase (s_memtime), summed over all waves into tally[2..]
    unsigned long long n_pass = 0, n_single = 0, t_lit = 0, n_lit = 0, t_copy = 0, t_far = 0, t_slow = 0, t_flush = 0, t_cover = 0, t_parse = 0, n_batch = 0, n_slow = 0, t_mark = 0;
    auto tick = [&]() { if (PROF) t_mark = __builtin_readcyclecounter(); };
    auto tock = [&](unsigned long long& acc) { if (PROF) { const unsigned long long now = __builtin_readcyclecounter(); acc += now - t_mark; t_mark = now; } };
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    const uint32_t magic = lane ? 65535u / lane + 1u : 0u;  // ceil(2^16 / lane): (j * magic) >> 16 == j / lane for j < 64

    // make inw[] cover [ip, ip + need) (need <= 80) unless the block ends first
    auto cover = [&](uint32_t need) {
        if (ip + need <= in_base + in_valid || in_base + in_valid >= iend) return;
        in_base = ip & ~15u;
        uint32_t n = iend - in_base;
        if (n > INWIN) n = INWIN;
        for (uint32_t k = 0; k < INWIN; k += 1024) {
            const uint32_t o = k + lane * 16;
            if (o < n) *reinterpret_cast<uint4*>(&inw[o]) = *reinterpret_cast<const uint4*>(src + in_base + o);  // image is padded by 64 B
        }
        in_valid = n;  // (one wave: LDS operations execute in program order, no barrier needed)
    };
    auto in_byte = [&](uint32_t pos) -> uint32_t {
        uint32_t i = pos - in_base;
        if (i > INWIN + 15) i = INWIN + 15;  // cannot happen after cover(); keeps a logic error inside the array
        return __builtin_amdgcn_readfirstlane(inw[i]);
    };
    // token and the two bytes behind it with ONE wait (the usual sequence of these streams has no literals, so they
    // are its offset)
    auto in_3bytes = [&](uint32_t pos) -> uint32_t {
        uint32_t i = pos - in_base;
        if (i > INWIN + 13) i = INWIN + 13;
        const uint32_t v = inw[i] | (static_cast<uint32_t>(inw[i + 1]) << 8) | (static_cast<uint32_t>(inw[i + 2]) << 16);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // write the finished part of the ring to global memory, a quarter of the ring at a time
    auto flush_to = [&](uint32_t upto) {
        while (upto - flushed >= kFlush) {
            for (uint32_t k = 0; k < kFlush; k += 1024) {
                const uint32_t o = flushed + k + lane * 16;
                *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(&ring[o & kRingMask]);
            }
            flushed += kFlush;
            // a far match may read these bytes back through another lane: they must have left this wave first
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    while (ip < iend && !err) {
        // ---- fast path: batches of up to 16 "bare" sequences -- no literals, match of 4..18 bytes -- which is 95 % of an
        // LZ4-fast FLAG stream.  A bare sequence is exactly 3 input bytes, so lanes 0..15 parse 16 of them AT ONCE (one
        // unaligned LDS read each), a 16-lane prefix sum of the match lengths gives every sequence its output position,
        // and one ballot says how many leading sequences of the batch are bare and valid.  The copies follow in passes
        // (below); a match behind the ring reads the flushed output (RING - 64 > kFlush + 16 * 18: that source always
        // lies below `flushed`).
        for (;;) {
            tick();
            cover(64);
            tock(t_cover);
            const uint32_t in_limit = in_base + in_valid;
            const uint32_t pos = ip + 3u * lane;
            bool ok = lane < 16u && pos + 3u <= in_limit;
            uint32_t w = 0xFFu;
            if (ok) __builtin_memcpy(&w, &inw[pos - in_base], 4);               // token, offset lo, offset hi, (next token)
            const uint32_t tok = w & 255u, offk = (w >> 8) & 0xFFFFu, mlk = tok + 4u;
            uint32_t incl = ok ? mlk : 0u;
            incl += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(incl), 0x111, 0xF, 0xF, false));  // row_shr:1
            incl += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(incl), 0x112, 0xF, 0xF, false));  // row_shr:2
            incl += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(incl), 0x114, 0xF, 0xF, false));  // row_shr:4
            incl += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(incl), 0x118, 0xF, 0xF, false));  // row_shr:8
            const uint32_t opk = op + incl - mlk;                                // where sequence k writes (if all before it are bare)
            ok = ok && tok < 15u && offk != 0u && offk <= opk && opk + mlk <= oend;
            const uint64_t bad = __builtin_amdgcn_ballot_w64(!ok);               // lanes >= 16 are never ok: bad != 0
            const uint32_t nb = static_cast<uint32_t>(__builtin_ctzll(bad));     // leading bare sequences of this batch, 0..16
            ++n_batch;
            tock(t_parse);
            // ---- the copies, in PASSES of up to four sequences (one wave decodes one block, so what bounds it is the chain of
            // dependent LDS round trips, ~150 cycles each: one per sequence when they are copied one by one).  97 % of the
            // sequences of a flag stream do not read what the few sequences before them wrote, so a pass takes up to four
            // consecutive sequences whose sources all end at or before the pass's first output byte -- or lie behind the
            // ring, in flushed output -- gives each a row of 16 lanes, and copies all of them with ONE read and ONE write
            // (matches of 17 or 18 bytes, matches that overlap their own output, off < ml, and matches within 16 bytes
            // of the ring's end go alone).  What a row needs
            // -- ring offsets of source and destination, length -- is packed into one word per sequence above
            // and read out of lanes k0..k0+3 as scalars.
            const bool fark = offk > RING - 64u;
            const uint32_t endk = opk + mlk;  // where sequence k's output ends
            // row words: ring offset of the source | (length - 1) << 16; ring offset of the destination.  A row adds its
            // column without masking, so a sequence within 16 bytes of the end of the ring goes alone (0.4 %)
            const uint32_t srck = (opk - offk) & kRingMask, dstk = opk & kRingMask;
            const bool wrapk = (srck > RING - 16u) | (dstk > RING - 16u);
            const uint32_t gsrck = opk - offk;  // source offset in the block's output (far matches read it from global memory)
            const uint64_t farm = __builtin_amdgcn_ballot_w64(fark) & ((1ull << nb) - 1ull);
            const uint32_t row = lane >> 4, col = lane & 15u;
            const uint32_t src_end = fark ? 0u : endk - offk;  // (far sources lie in flushed output: never after P)
            const uint64_t never = __builtin_amdgcn_ballot_w64((mlk > 16u) | wrapk) | (~0ull << nb);
            // Rows get their sequence's word through LDS: the batch's 16 words are stored once, and each pass's `row + k0`
            // gather is ONE read issued a pass ahead (before the previous pass's ring read, so it returns first) instead of
            // four v_readlane and a three-way select per pass -- the decoder is issue-bound when every wave slot is taken.
            if (lane < kTabN) {
                tab_row[lane] = lane < 16u ? make_uint2(srck | ((mlk - 1u) << 16), dstk) : make_uint2(0u, 0u);
                tab_far[lane] = lane < 16u ? gsrck : 0u;
            }
            uint2 v_next = tab_row[row];
            uint32_t k0 = 0;
            while (k0 < nb) {
                const uint32_t P = __builtin_amdgcn_readlane(opk, k0);  // first output byte of the pass
                // a pass ends at the first sequence that cannot join: a near source that ends after P, or (fixed per batch)
                // 17..18 bytes / past the batch; four rows at most
                const uint64_t reads_pass = __builtin_amdgcn_uicmp(src_end, P, 34 /* unsigned > : the lane mask straight from v_cmp */);
                const uint32_t k1 = k0 + static_cast<uint32_t>(__builtin_ctz(static_cast<uint32_t>((reads_pass | never) >> k0) | 16u));
                if (PROF) { if (k1 == k0) ++n_single; else ++n_pass; }
                if (k1 == k0) {
                    // alone: 17..18 bytes, or a source that overlaps its own output (period off < ml)
                    const uint32_t off = __builtin_amdgcn_readlane(offk, k0), ml = __builtin_amdgcn_readlane(mlk, k0);
                    if (off <= RING - 64u) {
                        uint32_t m = __builtin_amdgcn_readlane(magic, off & 63u);
                        if (off >= 64u) m = 0;
                        const uint32_t j = lane - __umul24(__umul24(lane, m) >> 16, off);  // lane mod off
                        if (lane < ml) ring[(P + lane) & kRingMask] = ring[(P - off + j) & kRingMask];
                    } else {
                        ++nfar;
                        if (lane < ml)
                            ring[(P + lane) & kRingMask] = __hip_atomic_load(&dst[P - off + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ++k0;
                    v_next = tab_row[k0 + row];
                    continue;
                }
                const uint2 v = v_next;
                v_next = tab_row[k1 + row];  // (k1 + row <= 19 < kTabN)
                const bool act = (row < k1 - k0) & (col <= (v.x >> 16));
                // (lanes with nothing to copy read and write a scratch byte of their own: straight-line LDS traffic, so the
                // only wait the compiler needs is the one between this read and this write)
                const uint32_t ra = act ? (v.x & 0xFFFFu) + col : kScratch + lane;
                const uint32_t wa = act ? v.y + col : kScratch + lane;
                uint32_t d = lds[ra];
                const uint32_t farbits = static_cast<uint32_t>(farm >> k0) & ((1u << (k1 - k0)) - 1u);
                if (farbits) {
                    const uint32_t gv = tab_far[k0 + row];
                    if (act && ((farbits >> row) & 1u)) d = __hip_atomic_load(&dst[gv + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nfar += static_cast<uint32_t>(__builtin_popcount(farbits));
                }
                lds[wa] = static_cast<uint8_t>(d);
                k0 = k1;
            }
            if (nb) op = __builtin_amdgcn_readlane(endk, nb - 1u);
            tock(t_copy);
            ip += 3u * nb;
            nseq += nb;
            flush_to(op);
            tock(t_flush);
            if (nb == 16u) continue;
            // The sequence that ended the batch.  The usual one has 1..14 literals and a short match: its token is
            // already here (lane nb's word), ONE LDS read brings the literals and the offset behind them into lanes, the
            // literals go from those lanes to the ring, then the match as above.  Everything else -- long literal runs,
            // long matches, the end of the block or of the staged window -- goes to the general code below.
            const uint32_t tq = __builtin_amdgcn_readlane(w, nb) & 255u;        // (0xFF when lane nb had nothing to read)
            const uint32_t ll = tq >> 4, mq = (tq & 15u) + 4u;
            if (ll == 0u || ll == 15u || mq == 19u || ip + 3u + ll > in_limit || ll + mq > oend - op) break;
            uint32_t lb = 0;
            if (lane < ll + 2u) lb = inw[ip + 1u + lane - in_base];
            const uint32_t offq = __builtin_amdgcn_readlane(lb, ll) | (__builtin_amdgcn_readlane(lb, ll + 1u) << 8);
            if (lane < ll) ring[(op + lane) & kRingMask] = static_cast<uint8_t>(lb);
            op += ll;
            if (offq == 0u || offq > op) { err = 5; break; }
            if (offq <= RING - 64u) {
                uint32_t m = __builtin_amdgcn_readlane(magic, offq & 63u);
                if (offq >= 64u) m = 0;
                const uint32_t jq = lane - __umul24(__umul24(lane, m) >> 16, offq);
                if (lane < mq) ring[(op + lane) & kRingMask] = ring[(op - offq + jq) & kRingMask];
            } else {
                ++nfar;
                if (lane < mq)
                    ring[(op + lane) & kRingMask] = __hip_atomic_load(&dst[op - offq + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            op += mq;
            ip += 3u + ll;
            ++nseq;
            ++n_lit;
            flush_to(op);
            tock(t_lit);
        }
        if (ip >= iend) break;
        cover(24);
        tick();
        ++n_slow;
        const uint32_t t3 = in_3bytes(ip);
        const uint32_t token = t3 & 255u;
        ++ip;
        ++nseq;
        // ---- literals
        uint32_t ll = token >> 4;
        if (ll == 15) {
            uint32_t e;
            do {
                cover(1);
                if (ip >= iend) { err = 1; break; }
                e = in_byte(ip);
                ++ip;
                ll += e;
            } while (e == 255);
            if (err) break;
        }
        if (ll > iend - ip || ll > oend - op) { err = 2; break; }
        const bool bare = ll == 0;
        while (ll) {
            const uint32_t n = ll < 64 ? ll : 64;
            cover(n);
            if (lane < n) ring[(op + lane) & kRingMask] = inw[ip + lane - in_base];
            ip += n;
            op += n;
            ll -= n;
            flush_to(op);
        }
        if (ip >= iend) break;  // the last sequence has no match
        // ---- match
        uint32_t off;
        if (bare) {
            off = t3 >> 8;  // already here
            if (ip + 2 > iend) { err = 3; break; }
        } else {
            cover(2);
            if (ip + 2 > iend) { err = 3; break; }
            off = in_byte(ip) | (in_byte(ip + 1) << 8);
        }
        ip += 2;
        uint32_t ml = token & 15u;
        if (ml == 15) {
            uint32_t e;
            do {
                cover(1);
                if (ip >= iend) { err = 4; break; }
                e = in_byte(ip);
                ++ip;
                ml += e;
            } while (e == 255);
            if (err) break;
        }
        ml += 4;
        if (off == 0 || off > op || ml > oend - op) { err = 5; break; }
        const bool near = off <= RING - 64;
        nfar += near ? 0u : 1u;
        while (ml) {
            const uint32_t n = ml < 64 ? ml : 64;
            // out[op + j] = out[op + j - off]; for off < n the source repeats with period off
            const uint32_t j = (off >= n) ? lane : lane % off;
            uint32_t v = 0;
            if (near) {
                if (lane < n) v = ring[(op - off + j) & kRingMask];
            } else {
                // farther back than the ring: already flushed (op - off + n <= flushed); device-scope load, past the L1
                if (lane < n) v = __hip_atomic_load(&dst[op - off + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane < n) ring[(op + lane) & kRingMask] = static_cast<uint8_t>(v);
            op += n;
            ml -= n;
            flush_to(op);
        }
        tock(t_slow);
    }
    if (!err && op != oend) err = 6;
    // tail of the ring.  An odd trailing byte of a block is dropped like the reference's N = size >> 1
    // (benchmark/flagstats.cpp:323): it stays zero in the padded slot, so the counting kernel sees no stray flag.
    if (!err) {
        for (uint32_t o = flushed + lane; o < (op & ~1u); o += 64) dst[o] = ring[o & kRingMask];
    }
    if (lane == 0) {
        status[blockIdx.x] = err;
        atomicAdd(&tally[0], static_cast<unsigned long long>(nseq));
        atomicAdd(&tally[1], static_cast<unsigned long long>(nfar));
        if (PROF) {
            atomicAdd(&tally[2], t_copy);
            atomicAdd(&tally[3], t_far);
            atomicAdd(&tally[4], t_slow);
            atomicAdd(&tally[5], t_flush);
            atomicAdd(&tally[6], t_cover);
            atomicAdd(&tally[7], t_parse);
            atomicAdd(&tally[8], n_batch);
            atomicAdd(&tally[9], n_slow);
            atomicAdd(&tally[10], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
            atomicAdd(&tally[11], t_lit);
            atomicAdd(&tally[12], n_lit);
            atomicAdd(&tally[13], n_pass);
            atomicAdd(&tally[14], n_single);
        }
    }
}

}  // namespace fsk

extern "C" hipError_t fsk_lz4_decode(int kernel, const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out,
                                     uint32_t* status, unsigned long long* tally, int prof, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (!comp || !blocks || !out || !status || !tally) return hipErrorInvalidValue;
    const dim3 grid(nblocks);
    switch (kernel) {
    case fsk::LZ4K_WORKGROUP:
        if (prof)
            hipLaunchKernelGGL((fsk::lz4_decode_wg<true>), grid, dim3(fsk::kWgThreads), 0, stream, comp, blocks, out, status, tally);
        else
            hipLaunchKernelGGL((fsk::lz4_decode_wg<false>), grid, dim3(fsk::kWgThreads), 0, stream, comp, blocks, out, status, tally);
        break;
    case fsk::LZ4K_WAVE:
        if (prof)
            hipLaunchKernelGGL((fsk::lz4_decode_wave<8192, 1024, true>), grid, dim3(64), 0, stream, comp, blocks, out, status, tally);
        else
            hipLaunchKernelGGL((fsk::lz4_decode_wave<8192, 1024>), grid, dim3(64), 0, stream, comp, blocks, out, status, tally);
        break;
    case fsk::LZ4K_WAVE_RING16:
        hipLaunchKernelGGL((fsk::lz4_decode_wave<16384, 4096>), grid, dim3(64), 0, stream, comp, blocks, out, status, tally);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" int fsk_lz4_blocks_per_cu(int kernel)
{
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
    if (kernel == fsk::LZ4K_WORKGROUP) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fsk::lz4_decode_wg<false>, fsk::kWgThreads, 0);
    if (kernel == fsk::LZ4K_WAVE) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fsk::lz4_decode_wave<8192, 1024>, 64, 0);
    if (kernel == fsk::LZ4K_WAVE_RING16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fsk::lz4_decode_wave<16384, 4096>, 64, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
