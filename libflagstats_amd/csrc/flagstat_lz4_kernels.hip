// flagstat_lz4_kernels.hip -- LZ4 block decode ON the GPU (row f1: the reference decodes every block with liblz4's
// LZ4_decompress_safe on the host, benchmark/flagstats.cpp:311-316).  Device code only; the host orchestration is
// flagstat_lz4_gpu.hip.
//
// Two kernels:
//
// lz4_decode_wg (r04, default) -- ONE WORKGROUP OF THREE WAVES PER BLOCK, the whole 64 KiB LZ4 window in LDS.
//   An LZ4 block is a serial chain twice over: the position of token k + 1 depends on token k, and a match may read what
//   the match before it wrote.  r03's kernel walked both chains with one wave, one to four sequences per LDS round trip
//   (216 cycles per sequence, 16-30 ms per block).  Here the two chains are taken apart and each is walked in units the
//   hardware is wide enough for:
//     wave 0, PARSE: 64 lanes decode the token at 64 CONSECUTIVE input bytes at once (every byte position speculatively:
//       literal length, match length, where the next token would be); which of them are real tokens is a walk over that
//       "next" table in scalar registers -- runs of bare 3-byte sequences by one ctz over a ballot, anything else one
//       v_readlane per sequence.  The real sequences get their output positions from a DPP prefix sum and leave, per
//       sequence, ONE 16-bit marker (the offset) at the match's first output byte; literal bytes go straight into the
//       output ring with a bit set in a literal bitmap.  No copies, no dependence on decoded data.
//     wave 1, SCAN: per 256-byte chunk of OUTPUT (4 bytes a lane) a DPP "last marker" scan turns the markers into one
//       source pointer per byte; pointers that land inside the chunk itself are chased to their roots by pointer
//       doubling on packed byte indices (ds_bpermute, data-independent: a byte -> byte pointer composes without the
//       bytes), so that every byte ends up with the ring index of a byte that is FINAL before the chunk starts.
//     wave 2, COPY: per chunk one 16-byte read of four final sources, four byte gathers from the ring, one aligned
//       4-byte write: 256 output bytes per LDS round trip, whatever the sequences were; every 4 KiB the ring goes to
//       global memory with 16-byte stores.
//   The ring holds 64 KiB + 4 KiB: LZ4 offsets reach at most 65,535 bytes back, so NO match ever reads global memory
//   (r03: 9 % / 25 % of the sequences of an LZ4-fast / HC-9 flag stream went behind its 8 KiB ring, a global round trip
//   each) and the 4 KiB are what the parser may write ahead of the copier.  The waves hand over through four words in
//   LDS (positions reached, first error) polled with s_sleep; every LDS operation of a wave executes in order, so "data,
//   then position" needs no fence.  80 KB of LDS per block: two blocks per CU, 512 in flight -- a block takes ~1-2 ms
//   instead of 16-30, so 512 at a time decode faster than 4,352 did.  Every index is masked, clamped or checked; a
//   malformed block sets its status word and the three waves leave through the same barrier; every wait is bounded.
//
// lz4_decode_wave (r03) -- one wave per block, 8 KiB ring; kept as the yardstick (knob "lz4_gpu_kernel" = 1).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "flagstat_lz4_kernels.h"

namespace fsk {

// ------------------------------------------------------------------------------------------------ lz4_decode_wg
// waves of a workgroup: 0 walk (token chain), 1-2 emit (even / odd records), 3-5 scan (chunks k mod 3), 6 copy
constexpr uint32_t kWgEmit = 2, kWgScan = 3;
constexpr uint32_t kWgThreads = 64u * (1u + kWgEmit + kWgScan + 1u);
constexpr uint32_t kWgNR = 67584;                  // output ring: 64 KiB window + 2 KiB of write-ahead (66 x 1024, 264 x 256)
constexpr uint32_t kWgChunk = 256;                 // output bytes per scan / copy step (4 per lane)
constexpr uint32_t kWgAhead = kWgNR - 65536 - kWgChunk;  // the emitters' position may lead the copier's by this much
constexpr uint32_t kWgMR = 1024;                   // marker slots: the emitters may lead the scanners by this many output bytes
constexpr uint32_t kWgK = 512;                     // final-source slots: the scanners may lead the copier by this many
constexpr uint32_t kWgInw = 4096, kWgInPad = 96;   // input window ring + mirror of its first bytes behind its end
constexpr uint32_t kWgQ = 16;                      // records the walker may lead the emitters by (each <= 82 input bytes)
constexpr uint32_t kWgSpan = 384;                  // most output bytes the emitters write before publishing (more: in batches)
constexpr uint32_t kWgFlush = 1024;
constexpr uint32_t kWgSpinLimit = 1u << 19;        // polls before a wait gives up (a logic error must not hang the GPU)
constexpr uint32_t kMarkLiteral = 0x10000u;        // marker: a literal run starts here (matches: their offset, 1..65535)
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint64_t kStride3 = 0x9249249249249249ull;  // bits 0, 3, 6, ..., 63
enum { REC_WINDOW = 0, REC_SEQ = 1, REC_END = 2 };
static_assert(kWgNR % kWgFlush == 0 && kWgNR % kWgChunk == 0 && kWgMR % kWgChunk == 0 && kWgK % kWgChunk == 0, "grids");
static_assert(kWgSpan + kWgChunk <= kWgMR && kWgSpan + kWgChunk + kWgK <= kWgAhead + kWgChunk, "no cyclic wait");
static_assert(kWgQ * 96u + 1024u + 96u < kWgInw - 1024u, "the walker cannot overwrite input an emitter still reads");

struct __attribute__((aligned(16))) WgLds {
    uint8_t ring[kWgNR];
    uint32_t mark[kWgMR];          // per output byte: 0, the offset of the match that starts there, or kMarkLiteral
    uint32_t fsrc[kWgK];           // per output byte: ring index of the byte it is a copy of (final before its chunk starts)
    uint8_t inw[kWgInw + kWgInPad + 16];
    uint4 q[kWgQ];                 // walker -> emitters: x = kind, then WINDOW: y = input position, z / w = member lanes;
                                   // SEQ: y = position of the literals | their count (<= 64) << 24, z = offset, w = match length (0: no match)
    uint32_t q_head, q_tail[kWgEmit];   // records pushed; per emitter the next record it will take
    uint32_t h_seq, h_op;               // records whose output has been accounted for, and the output position behind them
    uint32_t p_safe[kWgEmit];           // per emitter: every marker it owes below this position is written
    uint32_t s_clr[kWgScan], s_done[kWgScan];  // per scanner: start of the next chunk it will clear / end of the last chunk it has finished
    uint32_t d_op, c_ready, s_carry[4], err;
};

__device__ __forceinline__ uint32_t wg_ld(const uint32_t* p)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// "data, then position": LDS operations of one wave execute in order, so the compiler barrier is the only fence needed
__device__ __forceinline__ void wg_st(uint32_t* p, uint32_t v)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// poll until cond() holds; false if the block failed (here or in another wave) or the wait ran out.  The naps grow:
// ten waiting waves per CU polling every ~150 cycles took the LDS pipeline away from the waves that had work.
template <class F>
__device__ __forceinline__ bool wg_wait(WgLds& L, F cond)
{
    for (uint32_t spins = 0;; ++spins) {
        if (cond()) break;
        if ((spins & 3u) == 3u && wg_ld(&L.err)) return false;
        if (spins > kWgSpinLimit) {
            wg_st(&L.err, 9u);
            return false;
        }
        if (spins < 2u)
            __builtin_amdgcn_s_sleep(2);
        else if (spins < 6u)
            __builtin_amdgcn_s_sleep(6);
        else
            __builtin_amdgcn_s_sleep(16);
    }
    asm volatile("" ::: "memory");
    return true;
}
template <bool PROF, class F>
__device__ __forceinline__ bool wg_wait_timed(WgLds& L, unsigned long long& t_wait, F cond)
{
    if (cond()) {
        asm volatile("" ::: "memory");
        return true;
    }
    const unsigned long long t0 = PROF ? __builtin_readcyclecounter() : 0ull;
    const bool ok = wg_wait(L, cond);
    if (PROF) t_wait += __builtin_readcyclecounter() - t0;
    return ok;
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x)
{
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xF, 0xF, false));  // row_shr:1
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xF, 0xF, false));  // row_shr:2
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xF, 0xF, false));  // row_shr:4
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xF, 0xF, false));  // row_shr:8
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xA, 0xF, false));  // row_bcast:15
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xC, 0xF, false));  // row_bcast:31
    return x;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// inclusive prefix maximum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x)
{
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xA, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xC, 0xF, false)));
    return x;
}

// ---- wave 0: the token chain.  Stages the input, finds which byte positions are tokens, hands the emitters one record
// per 64-byte window (its member lanes) or per sequence the window form does not cover.
template <bool PROF>
__device__ void lz4wg_walk(WgLds& L, const uint8_t* __restrict__ src, const uint32_t iend, const uint32_t oend,
                           const uint32_t lane, unsigned long long* __restrict__ tally)
{
    uint32_t ip = 0, in_hi = 0, err = 0, nrec = 0, tail_seen = 0;
    unsigned long long t_wait = 0, n_win = 0, n_seq = 0, n_walk = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    uint4 pend = make_uint4(0, 0, 0, 0);
    // input: a ring of 4 KiB filled 1 KiB at a time, the NEXT KiB always in flight in registers.  A KiB is committed
    // once ip is within 1 KiB of the staged end; what it overwrites lies > 2 KiB behind ip, and the emitters are at most
    // kWgQ records (of <= 82 input bytes) behind.  The ring's first 96 bytes are mirrored behind its end: no read wraps.
    // (unconditional, the offset clamped instead: a load under a condition made the compiler wait for it on the spot.
    // At most 16 bytes from iend on are read: the image is padded by 64)
    auto issue = [&]() { pend = *reinterpret_cast<const uint4*>(src + umin(in_hi + lane * 16u, iend)); };
    auto cover = [&]() {
        while (in_hi < iend && ip + 1024u >= in_hi) {
            const uint32_t r = in_hi & (kWgInw - 1u);
            *reinterpret_cast<uint4*>(&L.inw[r + lane * 16u]) = pend;
            if (r == 0u && lane < kWgInPad / 16u) *reinterpret_cast<uint4*>(&L.inw[kWgInw + lane * 16u]) = pend;
            in_hi += 1024u;
            issue();
        }
    };
    auto inb = [&](uint32_t pos) -> uint32_t { return __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(L.inw[pos & (kWgInw - 1u)])); };
    auto push = [&](uint32_t kind, uint32_t y, uint32_t z, uint32_t w) -> bool {
        if (nrec >= tail_seen + kWgQ) {
            const bool ok = wg_wait_timed<PROF>(L, t_wait, [&] {
                tail_seen = umin(wg_ld(&L.q_tail[0]), wg_ld(&L.q_tail[1]));
                return nrec < tail_seen + kWgQ;
            });
            if (!ok) return false;
        }
        if (lane == 0u) L.q[nrec & (kWgQ - 1u)] = make_uint4(kind, y, z, w);
        ++nrec;
        wg_st(&L.q_head, nrec);
        return true;
    };
    // every queued record has been taken (before the walker runs far ahead of input that a record still refers to)
    auto drain = [&]() -> bool {
        return wg_wait_timed<PROF>(L, t_wait, [&] { return umin(wg_ld(&L.q_tail[0]), wg_ld(&L.q_tail[1])) >= nrec; });
    };
    // one sequence of any shape at ip, scalar: its literals in pieces of <= 64 bytes (by reference), the last piece with
    // the match (offset and length by value).  No record makes the walker advance more than 96 input bytes.
    auto slow_sequence = [&]() {
        ++n_seq;
        cover();
        const uint32_t token = inb(ip);
        ++ip;
        uint32_t ll = token >> 4;
        if (ll == 15u) {
            uint32_t e, n = 0;
            do {
                cover();
                if (ip >= iend) { err = 1; return; }
                if (++n == 8u && !drain()) { err = 8; return; }
                e = inb(ip);
                ++ip;
                ll += e;
                if (ll > oend) { err = 2; return; }
            } while (e == 255u);
        }
        if (ll > iend - ip) { err = 2; return; }
        while (ll > 64u) {
            cover();
            if (!push(REC_SEQ, ip | (64u << 24), 0u, 0u)) { err = 8; return; }
            ip += 64u;
            ll -= 64u;
        }
        const uint32_t lit_at = ip;
        ip += ll;
        cover();
        if (ip >= iend) {  // the last sequence has no match
            if (!push(REC_SEQ, lit_at | (ll << 24), 0u, 0u)) err = 8;
            return;
        }
        if (ip + 2u > iend) { err = 3; return; }
        const uint32_t off = inb(ip) | (inb(ip + 1u) << 8);
        ip += 2u;
        if (off == 0u) { err = 5; return; }
        uint32_t ml = token & 15u;
        if (ml == 15u) {
            // (the literals of this sequence go first: the length bytes may be many)
            if (ll && !push(REC_SEQ, lit_at | (ll << 24), 0u, 0u)) { err = 8; return; }
            ll = 0;
            uint32_t e, n = 0;
            do {
                cover();
                if (ip >= iend) { err = 4; return; }
                if (++n == 8u && !drain()) { err = 8; return; }
                e = inb(ip);
                ++ip;
                ml += e;
                if (ml > oend) { err = 5; return; }
            } while (e == 255u);
        }
        if (!push(REC_SEQ, lit_at | (ll << 24), off, ml + 4u)) err = 8;
    };

    if (iend >= (1u << 24)) err = 10;  // (records carry input positions in 24 bits; the format's blocks are 1,024,000 bytes)
    issue();
    while (!err) {
        cover();
        if (ip + 96u > iend) break;  // the block's last bytes: one sequence at a time below
        ++n_win;
        // ---- every byte position of [ip, ip + 64) as if it were a token: where would the next token be
        const uint32_t wi = (ip + lane) & (kWgInw - 1u);
        const uint32_t tok = L.inw[wi];
        const uint32_t ll = tok >> 4, mlc = tok & 15u;
        const bool simple = (ll < 15u) & (mlc < 15u);      // no length continues in further bytes
        const uint32_t nxt = lane + 3u + ll;
        // not bare (bare = token, offset and nothing else); position 63 counts as not bare so that a run's ctz always ends
        const int64_t nb = static_cast<int64_t>(~__builtin_amdgcn_ballot_w64(simple & (ll == 0u)) | (1ull << 63));
        // per position, data-parallel: g = where the run of bare sequences that starts here ends (the first position on the
        // 3-byte stride that is not bare; may lie behind the window), h = where the chain is after the sequence at g
        const uint64_t x = static_cast<uint64_t>(nb >> lane) & kStride3;
        const uint32_t g = lane + static_cast<uint32_t>(__builtin_ctzll(x));  // (x != 0: bit 63 of nb)
        const uint32_t at_g = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>((g & 63u) << 2), static_cast<int>(simple ? nxt : 0x80u + lane)));
        const uint32_t h = g < 64u ? at_g : g;             // >= 0x80: the sequence at h - 0x80 is not simple
        // ---- the chain itself: from lane 0, one step per run of bare sequences + the sequence behind it
        uint64_t members = 0;
        uint32_t p = 0, stop_at = kNone;
        for (;;) {
            const uint32_t gp = __builtin_amdgcn_readlane(g, p), hp = __builtin_amdgcn_readlane(h, p);
            const uint32_t run = gp - p;                    // 3 x bare sequences from p; p + run <= 65
            members |= (kStride3 & ((1ull << run) - 1ull)) << p;
            if (PROF) ++n_walk;
            if (gp >= 64u) {
                p = gp;
                break;
            }
            if (hp >= 0x80u) {
                stop_at = gp;
                p = gp;
                break;
            }
            members |= 1ull << gp;
            p = hp;
            if (p >= 64u) break;
        }
        if (members && !push(REC_WINDOW, ip, static_cast<uint32_t>(members), static_cast<uint32_t>(members >> 32))) {
            err = 8;
            break;
        }
        ip += p;
        if (stop_at != kNone) slow_sequence();
    }
    while (!err && ip < iend) slow_sequence();
    if (!err && ip != iend) err = 6;
    if (err) wg_st(&L.err, err);
    if (!err) {
        (void)push(REC_END, 0u, 0u, 0u);  // one for each emitter
        (void)push(REC_END, 0u, 0u, 0u);
    }
    if (PROF && lane == 0u) {
        atomicAdd(&tally[2], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        atomicAdd(&tally[3], t_wait);
        atomicAdd(&tally[5], n_win);
        atomicAdd(&tally[6], n_seq);
        atomicAdd(&tally[7], n_walk);
    }
}

// ---- waves 1 and 2: records -> output positions, markers, literal bytes.  Emitter `which` takes records which, which + 2, ...
template <bool PROF>
__device__ void lz4wg_emit(WgLds& L, const uint32_t oend, const uint32_t lane, const uint32_t which,
                           unsigned long long* __restrict__ tally)
{
    uint32_t err = 0, nseq = 0, s_seen = 0, d_seen = 0;
    unsigned long long t_wait = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    // room for output positions [at, at + n): marker slots the scanners have cleared, ring bytes the copier no longer reads
    auto room = [&](uint32_t at, uint32_t n) -> bool {
        if (at + n <= s_seen + kWgMR && at + n <= d_seen + kWgAhead) return true;
        return wg_wait_timed<PROF>(L, t_wait, [&] {
            s_seen = umin(umin(wg_ld(&L.s_clr[0]), wg_ld(&L.s_clr[1])), wg_ld(&L.s_clr[2]));
            d_seen = wg_ld(&L.d_op);
            return at + n <= s_seen + kWgMR && at + n <= d_seen + kWgAhead;
        });
    };
    for (uint32_t r = which;; r += kWgEmit) {
        if (!wg_wait_timed<PROF>(L, t_wait, [&] { return wg_ld(&L.q_head) > r; })) break;
        const uint4 rec = L.q[r & (kWgQ - 1u)];
        const uint32_t kind = __builtin_amdgcn_readfirstlane(rec.x), ry = __builtin_amdgcn_readfirstlane(rec.y);
        const uint32_t rz = __builtin_amdgcn_readfirstlane(rec.z), rw = __builtin_amdgcn_readfirstlane(rec.w);
        asm volatile("" ::: "memory");
        // what this record adds to the output; then its place in the output, from the record before it
        uint32_t total = 0, ll = 0, ml = 0, len = 0, incl = 0, offv = 0;
        uint64_t lits = 0;
        bool member = false;
        uint32_t wi = 0;
        if (kind == REC_WINDOW) {
            wi = (ry + lane) & (kWgInw - 1u);
            uint64_t q;
            __builtin_memcpy(&q, &L.inw[wi], 8);  // token + the first 7 bytes behind it (all the literals of most sequences)
            const uint32_t tok = static_cast<uint32_t>(q) & 255u;
            ll = tok >> 4;
            ml = (tok & 15u) + 4u;
            uint16_t o16;
            __builtin_memcpy(&o16, &L.inw[wi + 1u + ll], 2);
            offv = o16;
            lits = q >> 8;
            member = ((static_cast<uint64_t>(rz) | (static_cast<uint64_t>(rw) << 32)) >> lane) & 1ull;
            len = member ? ll + ml : 0u;
            incl = wave_scan_add(len);
            total = __builtin_amdgcn_readlane(incl, 63);
        } else if (kind == REC_SEQ) {
            ll = ry >> 24;
            ml = rw;
            total = ll + ml;
        }
        if (!wg_wait_timed<PROF>(L, t_wait, [&] { return wg_ld(&L.h_seq) >= r; })) break;
        const uint32_t op = wg_ld(&L.h_op);
        wg_st(&L.p_safe[which], op);  // everything this emitter owes below its current record is written
        if (kind == REC_END) {
            if (op != oend) err = 6;
            wg_st(&L.h_seq, r + 1u);
            wg_st(&L.p_safe[which], kNone);
            wg_st(&L.q_tail[which], r + kWgEmit);
            break;
        }
        if (total > oend - op) {
            err = 5;
            break;
        }
        if (lane == 0u) L.h_op = op + total;
        wg_st(&L.h_seq, r + 1u);
        uint32_t op_idx = op % kWgNR;
        if (kind == REC_WINDOW) {
            nseq += static_cast<uint32_t>(__builtin_popcount(rz) + __builtin_popcount(rw));
            const uint32_t rel = incl - len;           // this sequence's first output byte, relative to op
            const uint32_t mpos = op + rel + ll;       // ... and its match's
            if (__builtin_amdgcn_ballot_w64(member & ((offv == 0u) | (offv > mpos)))) {
                err = 5;
                break;
            }
            // in batches of <= kWgSpan output bytes (one, but for windows full of long matches)
            uint32_t done = 0;
            bool failed = false;
            while (done < total) {
                const uint32_t upto = done + kWgSpan;
                const bool now = member & (rel >= done) & (incl <= upto);
                const uint64_t nowm = __builtin_amdgcn_ballot_w64(now);
                const uint32_t last = 63u - static_cast<uint32_t>(__builtin_clzll(nowm));   // (never empty: one sequence is <= 32 bytes)
                const uint32_t end = __builtin_amdgcn_readlane(incl, last);
                if (!room(op + done, end - done)) {
                    failed = true;
                    break;
                }
                if (now) L.mark[mpos & (kWgMR - 1u)] = offv;
                const bool haslit = now & (ll > 0u);
                if (__builtin_amdgcn_ballot_w64(haslit)) {
                    if (haslit) L.mark[(op + rel) & (kWgMR - 1u)] = kMarkLiteral;
                    uint32_t ri = op_idx + rel;
                    if (ri >= kWgNR) ri -= kWgNR;
                    uint64_t lb = lits;
                    for (uint32_t j = 0; j < 7u; ++j) {
                        const bool a = haslit & (j < ll);
                        if (!__builtin_amdgcn_ballot_w64(a)) break;
                        if (a) L.ring[ri] = static_cast<uint8_t>(lb);
                        lb >>= 8;
                        ri = ri + 1u == kWgNR ? 0u : ri + 1u;
                    }
                    if (__builtin_amdgcn_ballot_w64(haslit & (ll > 7u))) {
                        for (uint32_t j = 7; j < 14u; ++j) {
                            const bool a = haslit & (j < ll);
                            if (!__builtin_amdgcn_ballot_w64(a)) break;
                            if (a) L.ring[ri] = L.inw[wi + 1u + j];
                            ri = ri + 1u == kWgNR ? 0u : ri + 1u;
                        }
                    }
                }
                done = end;
                wg_st(&L.p_safe[which], op + done);
            }
            if (failed) break;
        } else {
            ++nseq;
            // literals [ry & 0xFFFFFF, + ll) (ll <= 64), then a match of ml bytes at distance rz (none: ml == 0)
            const uint32_t off = rz;
            if (ml && (off == 0u || off > op + ll)) {
                err = 5;
                break;
            }
            if (!room(op, ll + 1u)) break;
            if (lane < ll) {
                uint32_t ri = op_idx + lane;
                if (ri >= kWgNR) ri -= kWgNR;
                L.ring[ri] = L.inw[((ry & 0xFFFFFFu) + lane) & (kWgInw - 1u)];
            }
            if (lane == 0u) {
                if (ll) L.mark[op & (kWgMR - 1u)] = kMarkLiteral;
                if (ml) L.mark[(op + ll) & (kWgMR - 1u)] = off;
            }
            wg_st(&L.p_safe[which], op + total);
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

// ---- waves 3 to 5: markers -> one final source per output byte; scanner `which` takes chunks which, which + 3, ...
template <bool PROF>
__device__ void lz4wg_scan(WgLds& L, const uint32_t oend, const uint32_t lane, const uint32_t which,
                           unsigned long long* __restrict__ tally)
{
    unsigned long long t_wait = 0, n_rounds = 0, n_inchunk = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    for (uint32_t kc = which; kc * kWgChunk < oend; kc += kWgScan) {
        const uint32_t c = kc * kWgChunk;
        const uint32_t need = c + kWgChunk < oend ? c + kWgChunk : oend;
        const uint32_t cidx = c % kWgNR;  // ring index of the chunk's first byte
        if (!wg_wait_timed<PROF>(L, t_wait, [&] { return umin(wg_ld(&L.p_safe[0]), wg_ld(&L.p_safe[1])) >= need; })) break;
        const uint32_t mslot = c & (kWgMR - 1u);
        const uint4 mk = *reinterpret_cast<const uint4*>(&L.mark[mslot + 4u * lane]);
        asm volatile("" ::: "memory");
        *reinterpret_cast<uint4*>(&L.mark[mslot + 4u * lane]) = make_uint4(0u, 0u, 0u, 0u);
        wg_st(&L.s_clr[which], c + kWgScan * kWgChunk);
        const uint32_t r0 = 4u * lane;
        // "the last marker at or before this byte": keys grow with the position, so it is a maximum
        uint32_t k[4];
        k[0] = mk.x ? ((r0 + 1u) << 17) | mk.x : 0u;
        k[1] = mk.y ? ((r0 + 2u) << 17) | mk.y : k[0];
        k[2] = mk.z ? ((r0 + 3u) << 17) | mk.z : k[1];
        k[3] = mk.w ? ((r0 + 4u) << 17) | mk.w : k[2];
        const uint32_t upto = wave_scan_max(k[3]);
        const uint32_t before = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(upto), 0x138, 0xF, 0xF, false));  // wave_shr:1
        const uint32_t last = __builtin_amdgcn_readlane(upto, 63);
        // the marker that runs into this chunk comes from the scanner of chunk kc - 1, and ours goes to the next one
        uint32_t carry = 0;
        if (kc) {
            if (!wg_wait_timed<PROF>(L, t_wait, [&] { return wg_ld(&L.c_ready) >= kc; })) break;
            carry = wg_ld(&L.s_carry[kc & 3u]);
        }
        if (lane == 0u) L.s_carry[(kc + 1u) & 3u] = last ? (last & 0x1FFFFu) : carry;
        wg_st(&L.c_ready, kc + 1u);
        uint32_t ptr[4], ext[4];
        bool any_in = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t kk = k[j] ? k[j] : before;
            const uint32_t m = kk ? (kk & 0x1FFFFu) : carry;
            const uint32_t off = m & 0xFFFFu;
            const uint32_t rel = r0 + static_cast<uint32_t>(j);
            const bool lit = ((m >> 16) != 0u) | (m == 0u);
            const bool in = !lit & (off <= rel);
            any_in |= in;
            ptr[j] = in ? rel - off : rel;
            // where a root's byte comes from: itself (a literal an emitter wrote) or the ring at distance off
            uint32_t s = cidx + rel;
            if (!lit) s = s >= off ? s - off : s + kWgNR - off;
            ext[j] = s;
        }
        if (__builtin_amdgcn_ballot_w64(any_in)) {
            // pointers inside the chunk: chase to the roots, doubling (<= 255 hops -> <= 8 rounds), four byte indices a dword
            ++n_inchunk;
            uint32_t p4 = ptr[0] | (ptr[1] << 8) | (ptr[2] << 16) | (ptr[3] << 24);
            for (int round = 0; round < 8; ++round) {
                uint32_t n[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t t = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(ptr[j] & ~3u), static_cast<int>(p4)));
                    n[j] = (t >> ((ptr[j] & 3u) * 8u)) & 255u;
                }
                const uint32_t n4 = n[0] | (n[1] << 8) | (n[2] << 16) | (n[3] << 24);
                const bool changed = n4 != p4;
                p4 = n4;
#pragma unroll
                for (int j = 0; j < 4; ++j) ptr[j] = n[j];
                if (PROF) ++n_rounds;
                if (!__builtin_amdgcn_ballot_w64(changed)) break;
            }
        }
        if (!wg_wait_timed<PROF>(L, t_wait, [&] { return c + kWgChunk <= wg_ld(&L.d_op) + kWgK; })) break;
        const uint32_t base = c & (kWgK - 1u);
        *reinterpret_cast<uint4*>(&L.fsrc[base + r0]) = make_uint4(ext[0], ext[1], ext[2], ext[3]);
        if (__builtin_amdgcn_ballot_w64(any_in)) {
            const uint4 f = make_uint4(L.fsrc[base + ptr[0]], L.fsrc[base + ptr[1]], L.fsrc[base + ptr[2]], L.fsrc[base + ptr[3]]);
            asm volatile("" ::: "memory");  // (every lane's reads are one instruction each, all before this write)
            *reinterpret_cast<uint4*>(&L.fsrc[base + r0]) = f;
        }
        wg_st(&L.s_done[which], c + kWgChunk);
    }
    if (PROF && lane == 0u) {
        atomicAdd(&tally[8], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        atomicAdd(&tally[9], t_wait);
        atomicAdd(&tally[10], n_rounds);
        atomicAdd(&tally[11], n_inchunk);
    }
}

// ---- wave 6: gather, write, flush
template <bool PROF>
__device__ void lz4wg_copy(WgLds& L, uint8_t* __restrict__ dst, const uint32_t oend, const uint32_t lane,
                           unsigned long long* __restrict__ tally)
{
    uint32_t cidx = 0, flushed = 0, fidx = 0;
    const uint32_t oend_even = oend & ~1u;  // an odd trailing byte of a block is dropped like the reference's N = size >> 1
    unsigned long long t_wait = 0, n_chunks = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    bool ok = true;
    for (uint32_t c = 0; c < oend; c += kWgChunk) {
        ok = wg_wait_timed<PROF>(L, t_wait, [&] { return wg_ld(&L.s_done[(c / kWgChunk) % kWgScan]) >= c + kWgChunk; });
        if (!ok) break;
        ++n_chunks;
        const uint4 f = *reinterpret_cast<const uint4*>(&L.fsrc[(c & (kWgK - 1u)) + 4u * lane]);
        const uint32_t b0 = L.ring[f.x < kWgNR ? f.x : kWgNR - 1u], b1 = L.ring[f.y < kWgNR ? f.y : kWgNR - 1u];
        const uint32_t b2 = L.ring[f.z < kWgNR ? f.z : kWgNR - 1u], b3 = L.ring[f.w < kWgNR ? f.w : kWgNR - 1u];
        *reinterpret_cast<uint32_t*>(&L.ring[cidx + 4u * lane]) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        wg_st(&L.d_op, c + kWgChunk);
        cidx += kWgChunk;
        if (cidx == kWgNR) cidx = 0;
        if (c + kWgChunk - flushed == kWgFlush && c + kWgChunk <= oend_even) {
            *reinterpret_cast<uint4*>(dst + flushed + lane * 16u) = *reinterpret_cast<const uint4*>(&L.ring[fidx + lane * 16u]);
            flushed += kWgFlush;
            fidx += kWgFlush;
            if (fidx == kWgNR) fidx = 0;
        }
    }
    if (ok && !wg_ld(&L.err)) {
        // what is left in the ring: < 2 KiB, contiguous from fidx (a flush unit never wraps, the rest may)
        uint32_t o = flushed;
        for (; o + 1024u <= oend_even; o += 1024u) {
            uint32_t ri = fidx + (o - flushed) + lane * 16u;
            if (ri >= kWgNR) ri -= kWgNR;  // (16-byte groups stay whole: kWgNR and the group starts are multiples of 16)
            *reinterpret_cast<uint4*>(dst + o + lane * 16u) = *reinterpret_cast<const uint4*>(&L.ring[ri]);
        }
        for (uint32_t b = o + lane; b < oend_even; b += 64u) {
            uint32_t ri = fidx + (b - flushed);
            if (ri >= kWgNR) ri -= kWgNR;
            dst[b] = L.ring[ri];
        }
    }
    if (PROF && lane == 0u) {
        atomicAdd(&tally[12], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        atomicAdd(&tally[13], t_wait);
        atomicAdd(&tally[14], n_chunks);
    }
}

template <bool PROF>
__global__ __launch_bounds__(kWgThreads) void lz4_decode_wg(const uint8_t* __restrict__ comp, const GpuBlock* __restrict__ blocks,
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
        L.h_seq = 0u;
        L.h_op = 0u;
        for (uint32_t i = 0; i < kWgEmit; ++i) {
            L.q_tail[i] = i;
            L.p_safe[i] = 0u;
        }
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
    else if (role <= kWgEmit + kWgScan)
        lz4wg_scan<PROF>(L, b.dst_len, lane, role - 1u - kWgEmit, tally);
    else
        lz4wg_copy<PROF>(L, out + b.dst_off, b.dst_len, lane, tally);
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
