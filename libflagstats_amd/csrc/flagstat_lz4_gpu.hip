// flagstat_lz4_gpu.hip -- the LZ4 block decode of row f1 ON the GPU (behind the block-file entries
// for large files, knob "lz4_decoder").
//
// The reference's block reader decodes every block with liblz4's LZ4_decompress_safe on the host
// (benchmark/flagstats.cpp:311-316); the host pipeline of this repo (flagstat_blocks.hip) does the same on N host threads
// and is PCIe-bound on the DECODED bytes (27 Gflags/s).  Here the file crosses PCIe as it is (2.1-3.4x fewer bytes) and is
// decoded on the device.  An LZ4 block is one serial chain of ~156,000 sequences per 1,024,000-byte block, so it is ONE
// WAVE PER BLOCK -- a wave decodes ~35 MB/s, but 4,352 of them are resident at once:
//   * compressed bytes staged through a 1 KiB LDS window (coalesced 16-byte loads),
//   * the last 8 KiB of output kept in an LDS ring, so a match copy is ds_read -> ds_write for every offset up to
//     8,128; farther matches read the already flushed output back from global memory,
//   * up to 16 bare sequences (3 input bytes each) parsed AT ONCE by 16 lanes, output positions by a DPP prefix sum,
//   * copied in PASSES of up to four sequences that do not read each other's output: one LDS read and one write for
//     all of them, each on a row of 16 lanes, their words gathered through a small LDS table read one pass ahead,
//   * the ring flushed to global memory 2 KiB at a time with coalesced 16-byte stores,
//   * every index masked or checked: a malformed block sets its status word and stops, it cannot fault.
// 9.3 KiB of LDS per wave = 17 waves per CU.  The decoder is bound by the chain of dependent LDS round trips of a wave
// when the chip is half empty and by instruction issue when every wave slot is taken: both reward fewer instructions
// per sequence, which is what every step from the first version (161 ms for a 4 GiB file) to this one (30 ms) did.
// Host side: lz4_gpu_run (pieces of the file on the copy stream, one decode launch per piece on its own stream, one
// K1/K2 pass; file mode with a reader pool; segments for files larger than the device).  Measurements:
// profiles/r03/gpu_lz4_4GiB.log, lz4_decoder_sweep.log; DESIGN.md section 4.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <thread>
#include <unistd.h>

#include "../../include/libflagstats_hip.h"
#include "flagstat_engine.h"

namespace fsk {

struct GpuBlock {
    uint64_t src_off;  // payload offset in the image
    uint64_t dst_off;  // offset in the decoded buffer (multiple of 16)
    uint32_t src_len;
    uint32_t dst_len;
};

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

namespace fsint {

// LZ4 block file, decoded ON the GPU.  The compressed bytes cross PCIe (2-3x fewer than the decoded flags the host pipeline
// sends), in `pieces` of whole blocks on the engine's copy stream; every piece's blocks are decoded by their own launch of
// lz4_decode_wave (one wave per block) on a decode stream of its own as soon as the piece has landed, so all but the last
// piece's decode hides behind the copies; one K1/K2 pass then counts the whole decoded buffer.  Image mode copies
// straight out of the caller's memory; file mode reads the file with `threads` parallel preads into the engine's three
// pinned chunk buffers and copies from there.  e.mu must be held.
// one segment: file bytes [file_lo, file_lo + bytes) hold `blocks` (offsets relative to the segment), dpos decoded bytes
static int lz4_gpu_segment(Engine& e, const Lz4GpuSource& in, uint64_t file_lo, const std::vector<fsk::GpuBlock>& blocks,
                           uint64_t bytes, uint64_t dpos, uint64_t n_flags, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats)
{
    const uint8_t* img = in.img ? in.img + file_lo : nullptr;
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    fsk::GpuBlock* d_blocks = nullptr;
    uint32_t* d_status = nullptr;
    unsigned long long* d_tally = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t landed[8] = {}, joined[8] = {}, pin_free[3] = {};
    hipStream_t dec_stream[8] = {};
    int rc = 0;
    auto cleanup = [&] {
        (void)hipStreamSynchronize(e.stream[0]);
        for (hipStream_t x : dec_stream)
            if (x) {
                (void)hipStreamSynchronize(x);
                (void)hipStreamDestroy(x);
            }
        if (d_blocks) (void)hipFree(d_blocks);
        if (d_status) (void)hipFree(d_status);
        if (d_tally) (void)hipFree(d_tally);
        for (hipEvent_t x : ev)
            if (x) (void)hipEventDestroy(x);
        for (hipEvent_t x : landed)
            if (x) (void)hipEventDestroy(x);
        for (hipEvent_t x : joined)
            if (x) (void)hipEventDestroy(x);
        for (hipEvent_t x : pin_free)
            if (x) (void)hipEventDestroy(x);
    };
#define LZG_TRY(expr)                            \
    do {                                         \
        hipError_t e_ = (expr);                  \
        if (e_ != hipSuccess) {                  \
            rc = fail_hip(#expr, e_);            \
            cleanup();                           \
            return rc;                           \
        }                                        \
    } while (0)
    hipStream_t s = e.stream[0];
    for (hipEvent_t& x : ev) LZG_TRY(hipEventCreate(&x));
    // the two large buffers belong to the engine and are reused by the next segment / file (knob "lz4_gpu_keep_bytes")
    const uint64_t want[2] = {bytes + 64, dpos + 16};
    for (int i = 0; i < 2; ++i)
        if (e.lz4_cap[i] < want[i]) {
            if (e.lz4_buf[i]) LZG_TRY(hipFree(e.lz4_buf[i]));
            e.lz4_buf[i] = nullptr;
            e.lz4_cap[i] = 0;
            const uint64_t cap = (want[i] + (64ull << 20) - 1) & ~((64ull << 20) - 1);
            LZG_TRY(hipMalloc(&e.lz4_buf[i], cap));
            e.lz4_cap[i] = cap;
        }
    d_comp = e.lz4_buf[0];
    d_out = e.lz4_buf[1];
    LZG_TRY(hipMalloc(&d_blocks, blocks.size() * sizeof(fsk::GpuBlock)));
    LZG_TRY(hipMalloc(&d_status, blocks.size() * sizeof(uint32_t)));
    LZG_TRY(hipMalloc(&d_tally, 16 * 8));
    // only the slack between blocks (16-byte slots) and behind dropped odd bytes needs zero flags: blocks of the reference's
    // writer are whole multiples of 16 bytes, so this is normally nothing at all
    bool ragged = false;
    for (const fsk::GpuBlock& b : blocks) ragged = ragged || (b.dst_len & 15u);
    if (ragged) LZG_TRY(hipMemsetAsync(d_out, 0, dpos + 16, s));
    LZG_TRY(hipMemsetAsync(d_status, 0xFF, blocks.size() * sizeof(uint32_t), s));
    LZG_TRY(hipMemsetAsync(d_tally, 0, 16 * 8, s));
    LZG_TRY(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(fsk::GpuBlock), hipMemcpyHostToDevice, s));
    // env FLAGSTATS_HIP_GPU_LZ4_RING = 8 (default) | 16: KiB of recent output per wave in LDS (see lz4_decode_wave)
    const char* rk = std::getenv("FLAGSTATS_HIP_GPU_LZ4_RING");
    const bool big_ring = rk && std::atoi(rk) == 16;
    const char* pk = std::getenv("FLAGSTATS_HIP_GPU_LZ4_PROFILE");  // tuning: per-phase wave cycles on stderr
    const bool prof = pk && std::atoi(pk) != 0;
    // pieces: a decode launch lasts as long as its slowest block whatever its size (one wave decodes ~35 MB/s), and
    // launches on one stream run one after the other -- so few, large pieces, each on a stream of its own.  Default:
    // one piece per 1024 blocks, at most 4.  env FLAGSTATS_HIP_GPU_LZ4_CHUNKS / _STREAMS override (tuning).
    const char* ck = std::getenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS");
    const char* sk = std::getenv("FLAGSTATS_HIP_GPU_LZ4_STREAMS");
    uint32_t npieces = ck ? static_cast<uint32_t>(std::atoi(ck)) : static_cast<uint32_t>((blocks.size() + 1023) / 1024);
    if (!ck && npieces > 4) npieces = 4;
    if (npieces < 1) npieces = 1;
    if (npieces > 8) npieces = 8;
    if (npieces > blocks.size()) npieces = static_cast<uint32_t>(blocks.size());
    uint32_t nstreams = sk ? static_cast<uint32_t>(std::atoi(sk)) : npieces;
    if (nstreams < 1) nstreams = 1;
    if (nstreams > npieces) nstreams = npieces;
    for (uint32_t i = 0; i < nstreams; ++i) LZG_TRY(hipStreamCreateWithFlags(&dec_stream[i], hipStreamNonBlocking));
    for (uint32_t i = 0; i < npieces; ++i) LZG_TRY(hipEventCreateWithFlags(&landed[i], hipEventDisableTiming));
    for (uint32_t i = 0; i < nstreams; ++i) LZG_TRY(hipEventCreateWithFlags(&joined[i], hipEventDisableTiming));
    // file mode: the engine's pinned chunk buffers, filled by parallel preads
    uint8_t* pinned[3] = {nullptr, nullptr, nullptr};
    uint64_t span_cap = 0;
    int readers = 0;
    if (!img) {
        span_cap = (chunk_bytes() + 15) & ~15ull;
        if (span_cap < (4ull << 20)) span_cap = 4ull << 20;
        void* bufs[3];
        rc = pinned_reserve(e, span_cap, bufs);
        if (rc) {
            cleanup();
            return rc;
        }
        for (int i = 0; i < 3; ++i) {
            pinned[i] = static_cast<uint8_t*>(bufs[i]);
            LZG_TRY(hipEventCreateWithFlags(&pin_free[i], hipEventDisableTiming | hipEventBlockingSync));
        }
        readers = in.threads > 0 ? in.threads : static_cast<int>(std::thread::hardware_concurrency());
        if (readers > 16) readers = 16;
        if (readers < 1) readers = 1;
    }
    LZG_TRY(hipEventRecord(ev[0], s));
    for (uint32_t i = 0; i < nstreams; ++i) LZG_TRY(hipStreamWaitEvent(dec_stream[i], ev[0], 0));  // index on the device, status preset
    // pieces: blocks [first, last) = segment bytes [lo, hi), split by compressed bytes
    struct Piece {
        uint64_t first, last, lo, hi;
    };
    std::vector<Piece> pieces;
    for (uint64_t first = 0; pieces.size() < npieces && first < blocks.size();) {
        const uint64_t c = pieces.size();
        const uint64_t target = bytes / npieces * (c + 1);
        uint64_t last = first + 1;
        while (last < blocks.size() && (c + 1 == npieces || blocks[last].src_off + blocks[last].src_len <= target)) ++last;
        pieces.push_back(Piece{first, last, blocks[first].src_off - 8, blocks[last - 1].src_off + blocks[last - 1].src_len});
        first = last;
    }
    const uint32_t pieces_done = static_cast<uint32_t>(pieces.size());
    // piece c has been queued on the copy stream: decode its blocks behind it
    auto launch_piece = [&](uint32_t c) -> int {
        hipError_t e_ = hipEventRecord(landed[c], s);
        hipStream_t ds = dec_stream[c % nstreams];
        if (e_ == hipSuccess) e_ = hipStreamWaitEvent(ds, landed[c], 0);
        if (e_ != hipSuccess) return fail_hip("hipEventRecord / hipStreamWaitEvent(piece landed)", e_);
        const Piece& pc = pieces[c];
        const dim3 grid(static_cast<uint32_t>(pc.last - pc.first));
        if (prof)
            hipLaunchKernelGGL((fsk::lz4_decode_wave<8192, 1024, true>), grid, dim3(64), 0, ds, d_comp, d_blocks + pc.first, d_out,
                               d_status + pc.first, d_tally);
        else if (!big_ring)
            hipLaunchKernelGGL((fsk::lz4_decode_wave<8192, 1024>), grid, dim3(64), 0, ds, d_comp, d_blocks + pc.first, d_out,
                               d_status + pc.first, d_tally);
        else
            hipLaunchKernelGGL((fsk::lz4_decode_wave<16384, 4096>), grid, dim3(64), 0, ds, d_comp, d_blocks + pc.first, d_out,
                               d_status + pc.first, d_tally);
        e_ = hipGetLastError();
        return e_ == hipSuccess ? 0 : fail_hip("lz4_decode_wave launch", e_);
    };
    if (img) {
        for (uint32_t c = 0; c < pieces.size() && !rc; ++c) {
            hipError_t e_ = hipMemcpyAsync(d_comp + pieces[c].lo, img + pieces[c].lo, pieces[c].hi - pieces[c].lo, hipMemcpyHostToDevice, s);
            rc = e_ == hipSuccess ? launch_piece(c) : fail_hip("hipMemcpyAsync(block image piece)", e_);
        }
    } else {
        // File mode.  The pieces are cut into spans of one pinned buffer; a pool of `readers` threads preads span i + 1
        // (every thread its share) while this thread queues the copy of span i; a buffer is refilled once the copy that
        // last used it has left the host.  (Threads started per span cost a third of the read time: r03, 83 -> 7x ms.)
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
        std::mutex m;
        std::condition_variable cv_work, cv_done;
        size_t released = 0;                       // spans [0, released) may be read
        std::vector<int> done(spans.size(), 0);    // reader threads finished per span
        bool stop = false, failed = false;
        auto reader = [&](int t) {
            for (size_t i = 0; i < spans.size(); ++i) {
                {
                    std::unique_lock<std::mutex> ul(m);
                    cv_work.wait(ul, [&] { return released > i || stop; });
                    if (stop) return;
                }
                const Span& sp = spans[i];
                const uint64_t share = ((sp.len + static_cast<uint64_t>(readers) - 1) / static_cast<uint64_t>(readers) + 4095) & ~4095ull;
                uint64_t o = share * static_cast<uint64_t>(t);
                const uint64_t end = o + share < sp.len ? o + share : sp.len;
                bool ok = true;
                uint8_t* base = pinned[i % 3];
                while (o < end) {
                    const ssize_t r = pread(in.fd, base + o, end - o, static_cast<off_t>(file_lo + sp.at + o));
                    if (r <= 0) {
                        ok = false;
                        break;
                    }
                    o += static_cast<uint64_t>(r);
                }
                std::lock_guard<std::mutex> g(m);
                if (!ok) failed = true;
                if (++done[i] == readers) cv_done.notify_one();
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < readers && !spans.empty(); ++t) pool.emplace_back(reader, t);
        auto queue_span = [&](size_t i) -> int {
            {
                std::unique_lock<std::mutex> ul(m);
                cv_done.wait(ul, [&] { return done[i] == readers; });
                if (failed) return fail_text("block file: short read");
            }
            hipError_t e_ = hipMemcpyAsync(d_comp + spans[i].at, pinned[i % 3], spans[i].len, hipMemcpyHostToDevice, s);
            if (e_ == hipSuccess) e_ = hipEventRecord(pin_free[i % 3], s);
            if (e_ != hipSuccess) return fail_hip("hipMemcpyAsync(block file span)", e_);
            return spans[i].ends_piece >= 0 ? launch_piece(static_cast<uint32_t>(spans[i].ends_piece)) : 0;
        };
        for (size_t i = 0; i < spans.size() && !rc; ++i) {
            if (i >= 3) {
                hipError_t e_ = hipEventSynchronize(pin_free[i % 3]);  // the copy of span i - 3 has left this buffer
                if (e_ != hipSuccess) rc = fail_hip("hipEventSynchronize(pinned span)", e_);
            }
            if (!rc) {
                {
                    std::lock_guard<std::mutex> g(m);
                    released = i + 1;
                }
                cv_work.notify_all();
                if (i >= 1) rc = queue_span(i - 1);
            }
        }
        if (!rc && !spans.empty()) rc = queue_span(spans.size() - 1);
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cv_work.notify_all();
        for (std::thread& t : pool) t.join();
    }
    if (rc) {
        cleanup();
        return rc;
    }
    LZG_TRY(hipEventRecord(ev[1], s));  // every piece has landed
    for (uint32_t i = 0; i < nstreams; ++i) {
        LZG_TRY(hipEventRecord(joined[i], dec_stream[i]));
        LZG_TRY(hipStreamWaitEvent(s, joined[i], 0));
    }
    LZG_TRY(hipEventRecord(ev[2], s));  // ... and is decoded
    LZG_TRY(hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), s));
    rc = count_device_async(e, reinterpret_cast<const uint16_t*>(d_out), dpos / 2, e.d_out[0], s, e.ws[0],
                            OP_FLAGSTAT | (in.superset ? OP_SUPERSET : 0));
    if (rc) {
        cleanup();
        return rc;
    }
    LZG_TRY(hipEventRecord(ev[3], s));
    LZG_TRY(hipMemcpyAsync(e.h_out, e.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    unsigned long long tally[16] = {0};
    LZG_TRY(hipMemcpyAsync(tally, d_tally, sizeof tally, hipMemcpyDeviceToHost, s));
    std::vector<uint32_t> st(blocks.size());
    LZG_TRY(hipMemcpyAsync(st.data(), d_status, st.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LZG_TRY(hipStreamSynchronize(s));
    uint64_t bad = 0;
    for (uint32_t x : st) bad += x != 0;
    float h2d = 0, dec = 0, cnt = 0, pipe = 0;
    LZG_TRY(hipEventElapsedTime(&h2d, ev[0], ev[1]));
    LZG_TRY(hipEventElapsedTime(&dec, ev[1], ev[2]));
    LZG_TRY(hipEventElapsedTime(&cnt, ev[2], ev[3]));
    LZG_TRY(hipEventElapsedTime(&pipe, ev[0], ev[3]));
    if (prof) {
        int per_cu = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fsk::lz4_decode_wave<8192, 1024, true>, 64, 0);
        std::fprintf(stderr, "lz4 gpu profile: %d waves per CU fit\n", per_cu);
        const double tot = static_cast<double>(tally[10]) + 1e-9;
        std::fprintf(stderr, "lz4 gpu profile (ring 8 KiB): wave cycles %.3g | copy %.1f %% far %.1f %% lit %.1f %% slow %.1f %% flush %.1f %% cover %.1f %% parse %.1f %% | "
                             "%llu batches (%.1f seq each), %llu passes + %llu alone, %llu short-literal + %llu slow sequences, %llu far\n",
                     tot, 100 * tally[2] / tot, 100 * tally[3] / tot, 100 * tally[11] / tot, 100 * tally[4] / tot, 100 * tally[5] / tot, 100 * tally[6] / tot,
                     100 * tally[7] / tot, tally[8], tally[8] ? static_cast<double>(tally[0] - tally[9] - tally[12]) / tally[8] : 0.0, tally[13], tally[14], tally[12], tally[9], tally[1]);
    }
    stats->bad_blocks += bad;
    stats->h2d_ms += h2d;
    stats->decode_ms += dec;
    stats->count_ms += cnt;
    stats->sequences += tally[0];
    stats->far_matches += tally[1];
    stats->ring_kib = (!big_ring || prof) ? 8 : 16;
    stats->chunks += pieces_done;
    stats->pipeline_ms += pipe;
    stats->readers = static_cast<uint64_t>(readers);
    if (!bad) {
        if (in.superset) e.h_out[9] -= dpos / 2 - n_flags;  // zero flags in the slack between blocks are not reads (see run_pipeline)
        for (int k = 0; k < 32; ++k) out[k] += e.h_out[k];
    }
    cleanup();
    if (bad) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "block file: %llu block(s) failed to decode to their declared size (GPU LZ4 decoder)",
                      static_cast<unsigned long long>(bad));
        return fail_text(buf);
    }
    return 0;
#undef LZG_TRY
}

int lz4_gpu_run(Engine& e, const Lz4GpuSource& in, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats)
{
    const auto t_start = std::chrono::steady_clock::now();
    const uint8_t* img = in.img;
    const uint64_t bytes = in.bytes;
    // index: int32 uncompressed size, int32 compressed size, payload (benchmark/flagstats.cpp:119-138)
    std::vector<fsk::GpuBlock> blocks;
    uint64_t pos = 0, dpos = 0, n_flags = 0, usum = 0;
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
        blocks.push_back(fsk::GpuBlock{pos + 8, dpos, static_cast<uint32_t>(cs), static_cast<uint32_t>(us)});
        n_flags += static_cast<uint64_t>(us) >> 1;  // as benchmark/flagstats.cpp:323
        usum += static_cast<uint64_t>(us);
        dpos += (static_cast<uint64_t>(us) + 15) & ~15ull;
        pos += 8 + static_cast<uint64_t>(cs);
    }
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
    // Segments: compressed and decoded bytes of a segment are resident on the device together, so a file larger than the
    // card can hold goes through in several of them, one after the other (each with its own pieces, decode launches and
    // counting pass).  Default: a third of the free device memory, at most 16 GiB of decoded bytes; env
    // FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES overrides (tests).
    uint64_t seg_cap = 16ull << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 3 < seg_cap) seg_cap = free_b / 3;
        const char* sb = std::getenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES");
        if (sb && *sb) seg_cap = std::strtoull(sb, nullptr, 0);
    }
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
        const int rc = lz4_gpu_segment(e, in, file_lo, seg, file_hi - file_lo, dsz, seg_flags, out, stats);
        if (rc) return rc;
        ++stats->segments;
        b0 = b1;
    }
    if (e.lz4_cap[0] + e.lz4_cap[1] > knobs().lz4_gpu_keep_bytes.load()) {
        for (int i = 0; i < 2; ++i) {
            if (e.lz4_buf[i]) (void)hipFree(e.lz4_buf[i]);
            e.lz4_buf[i] = nullptr;
            e.lz4_cap[i] = 0;
        }
    }
    stats->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return 0;
}

}  // namespace fsint

extern "C" int FLAGSTATS_hip_blockimage_lz4_gpu(const void* image, uint64_t bytes, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats)
{
    if (!out) return fsint::fail_text("NULL out");
    if (!image && bytes) return fsint::fail_text("NULL image");
    fsint::Engine* ep = fsint::default_engine();
    if (!ep) return -1;
    fsint::Engine& e = *ep;
    std::lock_guard<std::mutex> lk(e.mu);
    if (fsint::engine_alive(e)) return -1;
    fsint::DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    static const uint8_t empty = 0;
    fsint::Lz4GpuSource src;
    src.img = image ? static_cast<const uint8_t*>(image) : &empty;
    src.bytes = bytes;
    return fsint::lz4_gpu_run(e, src, out, stats);
}
