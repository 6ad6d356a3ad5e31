// flagstat_zstd_kernels.hip -- Zstandard frame decode ON the GPU (row f1: the reference decodes every .zst block payload
// with libzstd's ZSTD_decompress on the host, benchmark/flagstats.cpp:636-682).  Device code only; the host orchestration
// is flagstat_gpu_decode.hip (shared with the LZ4 decoder).  Written from RFC 8878; tests/zstd_model.py restates the format
// and tests/zstd_gpu_model.py the record / checkpoint layout, both pinned against libzstd on the CPU.
//
// A frame of the reference's writer is 1,024,000 decoded bytes in eight blocks of 128 KiB; a block is a Huffman-coded
// literals section and an FSE-coded sequences section (literal length, match length, offset), one serial bit stream
// each, read BACKWARDS.  The one truly serial piece of work is the walk of a block's three FSE states (15 k steps a
// block); everything around it is taken off that walk.  Four kernels per piece of a file, through scratch in global memory:
//
// zstd_prepare -- ONE WAVE PER FRAME, a lane per block (eight side by side, more in further passes): block and section
//   headers; tables a block repeats from an earlier one are rebuilt by the lane that needs them from the earlier block's
//   description (no sharing, no ordering between lanes); Huffman streams decoded by 32 lanes (four streams a block), four
//   symbols per window, into the frame's literal buffer; the three FSE tables of every block built and written out twice:
//   2-byte CHAIN entries (next-state number | extra bits << 10 -- the bits a state reads are log - highbit(number), its
//   base (number << bits) - size) and 1-byte symbols.
// zstd_chain -- ONE WAVE PER TWO FRAMES, a lane per block, the walk and nothing else: a wave costs the same with one lane
//   or sixty-four and a block's chain tables take 2.5 KB of LDS, so sixteen blocks walk side by side.  A step is ONE LDS
//   round trip -- three table entries and a 16-byte window of the bit stream (staged through a 512-byte ring per lane;
//   one refill pass serves all sixteen blocks, a lane per 16-byte piece, both loads in flight) -- and ~85 instructions; what it saw (three states, 64 bits of the
//   window) goes to the stash in global memory.
// zstd_records -- ONE WORKGROUP PER FRAME, a wave per block, a lane per sequence, 64 at a time: symbols gathered from the
//   states (a batch ahead), values of the codes, positions by prefix sums, repeat offsets by RELAXATION over the lanes (a
//   lane's history is its left neighbour's after the neighbour's sequence; three plain offsets in a row fix it whatever
//   came before: ~6 rounds settle all 64).  What leaves is flat: 8-byte RECORDS (offset | literal length | match length,
//   runs above 16,383 split) and one CHECKPOINT per 64 records (output position, literal position, records valid).  A
//   block starts with an unknown offset history: the few repeat codes it cannot resolve stay in the records as they are,
//   and one wave replays those prefixes in frame order at the end.
// zstd_execute -- ONE WORKGROUP OF EIGHT WAVES PER FRAME (two per SIMD, 47 KB of LDS: a CU holds three frames; with ten waves it held one,
//   flagstat_zstd_kernels.h), the pipeline of the LZ4 kernel (flagstat_wgpipe.h) behind a new
//   front end: four EMIT waves take batches of 64 records (a DPP prefix sum places them; markers and literal bytes go into
//   the 32 KiB + 4 KiB output ring; a match from farther back than the ring keeps -- 11-38 % of the sequences of a flag stream --
//   is read from the output already flushed to global memory and becomes literal bytes), three SCAN waves turn markers into
//   one final source per byte, one COPY wave gathers, writes and flushes.
//
// Every index is masked, clamped or checked; a damaged frame sets its status word and cannot fault; every wait is bounded.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "flagstat_wgpipe.h"
#include "flagstat_zstd_kernels.h"

namespace fsk {

// ------------------------------------------------------------------------------------------------ scratch layout
struct ZLayout {
    uint64_t hdr_at, blk_at, tab_at, ck_at, rec_at, lit_at, stash_at, total;
    uint32_t rec_stride, ck_stride, lit_stride, blk_cap, nframes;
};
constexpr uint32_t kTabBytes = 3584;   // per block: chain entries LL 512 x 2, OF 256 x 2, ML 512 x 2; symbols LL 512, ML 512
constexpr uint32_t kTabOF = 1024, kTabML = 1536, kTabSymLL = 2560, kTabSymML = 3072;
struct ZFrameHdr {
    uint32_t nslots, nrec, out_len, nblk;
};
struct ZBlk {  // one Zstandard block, between the kernels
    uint32_t type, size, at;           // block type, size field, payload position in the frame
    uint32_t nseq, nlit, lit_at;       // sequences; literals and where they start in the frame's literal buffer
    uint32_t rec_at, rec_cap, stash_at;  // its records (a multiple of 64), its sequences in the stash
    uint32_t bits_at, bend, logs;      // sequence bit stream [bits_at, bend); accuracy logs LL | OF << 8 | ML << 16
    uint32_t chain_err;                // zstd_chain: 0 or a status code
    uint32_t nrec, out_len, out_at, n_sym, hist_known, hist[3];   // zstd_records
};
static inline ZLayout zstd_layout(uint32_t max_dst_len, uint32_t nframes, uint32_t min_blocks = 0)
{
    ZLayout y;
    y.nframes = nframes;
    // blocks per frame the scratch holds tables for: four times what 128 KiB blocks need (compressors that split blocks), and
    // some (more: kZstdTooManyBlocks; the host then decodes those frames once more with `min_blocks` = what a frame of 1 KiB
    // blocks -- the smallest window the format allows -- needs: 3.5 KB of tables a block are not reserved for every frame)
    y.blk_cap = 4u * ((max_dst_len + 131071u) / 131072u) + 8u;
    if (y.blk_cap > kZstdMaxBlocks) y.blk_cap = kZstdMaxBlocks;
    if (min_blocks > y.blk_cap) y.blk_cap = min_blocks < kZstdMaxBlocksRetry ? min_blocks : kZstdMaxBlocksRetry;
    // records: a sequence makes at least three bytes; per block up to 95 more (split runs, the literals behind the last
    // sequence, rounding to whole batches of 64)
    y.rec_stride = ((max_dst_len / 3u + 64u) & ~63u) + 96u * y.blk_cap;
    y.ck_stride = y.rec_stride / 64u;
    y.lit_stride = (max_dst_len + 64u + 15u) & ~15u;
    auto up = [](uint64_t v) { return (v + 255u) & ~255ull; };
    const uint64_t nf = nframes;
    y.hdr_at = 0;
    y.blk_at = up(nf * sizeof(ZFrameHdr));
    y.tab_at = y.blk_at + up(nf * y.blk_cap * sizeof(ZBlk));
    y.ck_at = y.tab_at + up(nf * y.blk_cap * kTabBytes);
    y.rec_at = y.ck_at + up(nf * y.ck_stride * 16u);
    y.lit_at = y.rec_at + up(nf * y.rec_stride * 8u);
    y.stash_at = y.lit_at + up(nf * y.lit_stride);
    y.total = y.stash_at + up(nf * y.rec_stride * 12u) + 256u;
    return y;
}

constexpr uint32_t kRecLL = 16383u, kRecML = 16383u;   // longest runs of one record
constexpr uint32_t kRecRep = 1u << 31;                  // word 0: a repeat code not resolved yet (low byte: the code; 0x10: "what the record before resolved to")
constexpr uint32_t kRecFlag = 1u << 30;                 // word 0: with kRecRep, the sequence had no literals; without, the record does not enter the offset history
constexpr uint32_t kBlockMax = 1u << 17;

// ------------------------------------------------------------------------------------------------ zstd_prepare
constexpr uint32_t kZeLanes = 8;          // blocks of a frame prepared side by side

__device__ const uint32_t kLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ const uint8_t kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint32_t kMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                                         35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ const uint8_t kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                        1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const int8_t kLLDefault[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ const int8_t kMLDefault[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__device__ const int8_t kOFDefault[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

struct ZeTables {  // one lane's table: Huffman while the literals are decoded, one FSE table at a time afterwards
    union {
        uint16_t huf[2048];  // symbol | bits << 8, indexed by the next max-bits bits of the stream
        uint32_t fse[512];   // symbol | state bits << 6 | extra bits << 10 | base << 16
    };
    uint32_t pad_[4];        // (the eight lanes' tables 4 KB + 16 B apart: the same entry of each in its own LDS bank, not all in one; worth 2 %)
};
struct ZeShare {  // what a lane tells the others about its block
    uint32_t def_huf;        // position of the Huffman tree description it brings, or ~0
    uint32_t def_tab[3];     // per table: ~0 leaves it as it is, ~1 repeats, else mode << 28 | position
    uint32_t type, size, at; // block type, size field, payload position
    uint32_t lit_type, nlit, lit_at, lit_data, lit_end, streams, max_bits, stream_at;
    uint32_t rec_at, rec_cap;
};
struct __attribute__((aligned(16))) ZeLds {
    ZeTables tab[kZeLanes];
    uint8_t weights[kZeLanes][256 + 16];   // (rows padded like the tables: lane-strided accesses spread over the banks)
    int16_t counts[kZeLanes][64 + 8];
    uint32_t rank[kZeLanes][16 + 1];
    ZeShare sh[kZeLanes];
};
constexpr uint32_t kNone = ~0u, kRepeat = ~1u;

__device__ __forceinline__ uint32_t ld_le16(const uint8_t* p) { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8); }
__device__ __forceinline__ uint32_t ld_le24(const uint8_t* p) { return ld_le16(p) | (static_cast<uint32_t>(p[2]) << 16); }
__device__ __forceinline__ uint32_t ld_le32(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld_le64(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t highbit(uint32_t v) { return 31u - static_cast<uint32_t>(__builtin_clz(v)); }
// the top n bits (n <= 63) of a top-aligned window, which moves on
__device__ __forceinline__ uint32_t take(uint64_t& w, uint32_t n)
{
    const uint64_t v = (w >> 1) >> (63u - n);
    w <<= n;
    return static_cast<uint32_t>(v);
}

// Forward little-endian bit reader over the frame (FSE table descriptions): position in bits, every read checked
// against `limit` (bytes) by the caller afterwards.
struct FwdBits {
    const uint8_t* base;
    uint32_t bit;
    __device__ __forceinline__ uint32_t peek(uint32_t n) const
    {
        const uint64_t v = ld_le64(base + (bit >> 3));
        return static_cast<uint32_t>(v >> (bit & 7u)) & ((1u << n) - 1u);
    }
};

// FSE table description at byte `at` of the frame -> accuracy log, counts[0..max_symbol] (LDS, -1: "less than one"),
// position behind it.  Returns false on a damaged description.  (RFC 8878 4.1.1)
__device__ bool fse_read_counts(const uint8_t* frame, uint32_t at, uint32_t limit, uint32_t max_symbol, uint32_t max_log, int16_t* counts,
                                uint32_t& log_out, uint32_t& after)
{
    if (at >= limit) return false;
    FwdBits br{frame, at * 8u};
    const uint32_t log = br.peek(4) + 5u;
    br.bit += 4u;
    if (log > max_log) return false;
    int32_t remaining = (1 << log) + 1;
    int32_t threshold = 1 << log;
    uint32_t nbits = log + 1u;
    uint32_t sym = 0;
    _Pragma("unroll 1") for (uint32_t s = 0; s <= max_symbol; ++s) counts[s] = 0;
    while (remaining > 1 && sym <= max_symbol) {
        if ((br.bit >> 3) >= limit) return false;
        const int32_t mx = (2 * threshold - 1) - remaining;
        const int32_t low = static_cast<int32_t>(br.peek(nbits - 1u));
        int32_t v;
        if (low < mx) {
            v = low;
            br.bit += nbits - 1u;
        } else {
            v = static_cast<int32_t>(br.peek(nbits));
            if (v >= threshold) v -= mx;
            br.bit += nbits;
        }
        v -= 1;
        remaining -= v < 0 ? -v : v;
        if (remaining < 1) return false;
        counts[sym++] = static_cast<int16_t>(v);
        if (v == 0) {
            _Pragma("unroll 1") for (;;) {
                if ((br.bit >> 3) >= limit) return false;
                const uint32_t rep = br.peek(2);
                br.bit += 2u;
                sym += rep;
                if (rep != 3u) break;
            }
        }
        while (remaining < threshold) {
            nbits -= 1u;
            threshold >>= 1;
        }
    }
    if (remaining != 1 || sym > max_symbol + 1u) return false;
    after = (br.bit + 7u) >> 3;
    if (after > limit) return false;
    log_out = log;
    return true;
}

// counts -> decode table of packed entries (RFC 8878 4.1.1; xbits(symbol) = extra bits the code reads).  One lane, serial.
template <class XB>
__device__ bool fse_build(uint32_t* table, int16_t* counts, uint32_t nsym, uint32_t log, XB xbits)
{
    const uint32_t size = 1u << log, mask = size - 1u;
    uint32_t high = size - 1u;
    _Pragma("unroll 1") for (uint32_t s = 0; s < nsym; ++s)
        if (counts[s] == -1) table[high--] = s;
    const uint32_t step = (size >> 1) + (size >> 3) + 3u;
    uint32_t pos = 0;
    _Pragma("unroll 1") for (uint32_t s = 0; s < nsym; ++s) {
        const int32_t c = counts[s];
        _Pragma("unroll 1") for (int32_t i = 0; i < c; ++i) {
            table[pos] = s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    _Pragma("unroll 1") for (uint32_t s = 0; s < nsym; ++s)
        if (counts[s] == -1) counts[s] = 1;  // (from here on: the next state number of the symbol)
    if (pos != 0u) return false;
    _Pragma("unroll 1") for (uint32_t u = 0; u < size; ++u) {
        const uint32_t s = table[u];
        const uint32_t x = static_cast<uint32_t>(counts[s]);
        counts[s] = static_cast<int16_t>(x + 1u);
        const uint32_t nb = log - highbit(x);
        table[u] = s | (nb << 6) | (xbits(s) << 10) | (((x << nb) - size) << 16);
    }
    return true;
}

// One of the three sequence tables of a block, first half: its symbol counts from the table's source (mode << 28 | position)
// into `counts` (LDS, 64 entries).  One lane, serial (a description is a chain of variable-length fields).  Returns
// log | nsym << 8, for an RLE table 0x10000 | symbol << 24, or ~0 for a damaged description.
__device__ uint32_t seq_counts(const uint8_t* frame, uint32_t limit, uint32_t src, int16_t* counts, const int8_t* dflt, uint32_t dflt_n, uint32_t dflt_log,
                               uint32_t max_symbol, uint32_t max_log)
{
    const uint32_t mode = src >> 28, at = src & 0xFFFFFFFu;
    if (mode == 0u) {
        _Pragma("unroll 1") for (uint32_t s = 0; s < dflt_n; ++s) counts[s] = dflt[s];
        return dflt_log | (dflt_n << 8);
    }
    if (mode == 1u) {
        if (at >= limit) return ~0u;
        const uint32_t sym = frame[at];
        if (sym > max_symbol) return ~0u;
        return 0x10000u | (sym << 24);
    }
    uint32_t log, after;
    if (!fse_read_counts(frame, at, limit, max_symbol, max_log, counts, log, after)) return ~0u;
    return log | ((max_symbol + 1u) << 8);
}

// ... second half: counts -> the table as the other kernels read it, by the WHOLE WAVE (RFC 8878 4.1.1, restated so that no cell
// waits for another).  r04 built the tables a lane per block (8 of 64 lanes busy, three serial passes of dependent LDS round trips
// per table: 0.96 M of the kernel's 1.66 M cycles).  Here, for a table of S = 2^log cells:
//   * the "less than one" symbols take the top cells, in symbol order downwards; `high` is the last cell below them;
//   * the spread visits cell (k * step) mod S at step k and skips the top cells, so cell u is visited at k(u) = u * step^-1 mod S
//     (step is odd: the inverse exists) and receives slot i = k - (number of skipped visits before k); slot i belongs to the symbol
//     whose run of `count` slots covers it: run starts as markers, a max-scan over the slots fills the runs;
//   * a symbol's cells get the state numbers count, count + 1, ... in order of u: cells are taken 64 at a time in order of u, a cell's
//     number is its symbol's running count (kept in the symbol's own lane) plus its rank among the group's cells of that symbol.
// Written straight out: out16[u] = number | extra bits << 10 (the bits a state reads are log - highbit(number), its base
// (number << bits) - S), outsym[u] = symbol.  `scratch` = 640 bytes of LDS.  `xb` = extra bits of the symbol whose number is the
// lane's.  Returns false for counts that do not fill the table.
__device__ bool fse_build_wave(const int16_t* counts, uint32_t nsym, uint32_t log, uint32_t lane, uint32_t xb, uint16_t* __restrict__ out16,
                               uint8_t* __restrict__ outsym, uint8_t* scratch)
{
    const uint32_t S = 1u << log, mask = S - 1u;
    uint8_t* const slotsym = scratch;          // [S <= 512]
    uint8_t* const lowlist = scratch + 512;    // [<= 64]
    const int32_t c = lane < nsym ? counts[lane] : 0;
    const bool is_low = c == -1;
    const uint32_t n_s = c > 0 ? static_cast<uint32_t>(c) : 0u;
    const unsigned long long lowmask = __builtin_amdgcn_ballot_w64(is_low);
    const uint32_t nlow = static_cast<uint32_t>(__builtin_popcountll(lowmask));
    if (nlow >= S) return false;
    const uint32_t high = S - 1u - nlow;
    const uint32_t incl = wave_scan_add(n_s), cum = incl - n_s;
    if (__builtin_amdgcn_readlane(incl, 63) != high + 1u) return false;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t i = lane; i < S; i += 64u) slotsym[i] = 0u;
    __syncthreads();   // (one wave: orders the LDS stores above before the ones below)
    if (n_s) slotsym[cum] = static_cast<uint8_t>(lane + 1u);
    if (is_low) lowlist[__builtin_popcountll(lowmask & below)] = static_cast<uint8_t>(lane);
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t base = 0; base < S; base += 64u) {
        uint32_t v = base + lane < S ? slotsym[base + lane] : 0u;
        v = wave_scan_max(v);
        v = v > carry ? v : carry;
        carry = __builtin_amdgcn_readlane(v, 63);
        if (base + lane < S) slotsym[base + lane] = static_cast<uint8_t>(v);
    }
    __syncthreads();
    const uint32_t step = (S >> 1) + (S >> 3) + 3u;
    uint32_t inv = step;   // Newton: the number of correct low bits doubles every round (3 -> 6 -> 12)
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    inv *= 2u - step * inv;
    const uint32_t skipk = ((S - 1u - lane) * inv) & mask;   // lane r < nlow: the visit that would have hit the r-th top cell
    uint32_t run = n_s;                                        // lane s: the next state number of symbol s
    for (uint32_t base = 0; base < S; base += 64u) {
        const uint32_t u = base + lane;
        const bool valid = u < S, normal = valid && u <= high;
        uint32_t sym = 0, x = 1u;
        if (valid && !normal) sym = lowlist[S - 1u - u];
        const uint32_t k = (u * inv) & mask;
        uint32_t skipped = 0;
        _Pragma("unroll 1") for (uint32_t r = 0; r < nlow; ++r) skipped += __builtin_amdgcn_readlane(skipk, r) < k ? 1u : 0u;
        if (normal) sym = static_cast<uint32_t>(slotsym[(k - skipped) & mask]) - 1u;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(normal);
        _Pragma("unroll 1") while (todo) {
            const uint32_t s0 = __builtin_amdgcn_readlane(sym, static_cast<uint32_t>(__builtin_ctzll(todo)));
            const unsigned long long m = __builtin_amdgcn_ballot_w64(normal && sym == s0);
            const uint32_t first = __builtin_amdgcn_readlane(run, s0 & 63u);
            if (normal && sym == s0) x = first + static_cast<uint32_t>(__builtin_popcountll(m & below));
            if (lane == s0) run += static_cast<uint32_t>(__builtin_popcountll(m));
            todo &= ~m;
        }
        const uint32_t xbs = __shfl(xb, static_cast<int>(sym & 63u));
        if (valid) {
            out16[u] = static_cast<uint16_t>(x | (xbs << 10));
            if (outsym) outsym[u] = static_cast<uint8_t>(sym);
        }
    }
    __syncthreads();   // (the scratch may be reused at once)
    return true;
}

// Huffman tree description at `at` (RFC 8878 4.2.1) -> table in LDS; returns max bits (0 on failure), position behind it.
__device__ uint32_t huf_build(const uint8_t* frame, uint32_t at, uint32_t limit, ZeTables& T, uint8_t* weights, int16_t* counts, uint32_t* rank,
                              uint32_t& after)
{
    if (at >= limit) return 0u;
    const uint32_t hb = frame[at];
    uint32_t n = 0;
    if (hb >= 128u) {
        n = hb - 127u;
        const uint32_t nbytes = (n + 1u) >> 1;
        if (at + 1u + nbytes > limit) return 0u;
        _Pragma("unroll 1") for (uint32_t i = 0; i < n; ++i) {
            const uint32_t b = frame[at + 1u + (i >> 1)];
            weights[i] = static_cast<uint8_t>((i & 1u) ? (b & 15u) : (b >> 4));
        }
        after = at + 1u + nbytes;
    } else {
        // FSE-compressed weights: two interleaved states over a backward stream
        if (hb < 2u || at + 1u + hb > limit) return 0u;
        const uint32_t end = at + 1u + hb;
        uint32_t log, tafter;
        if (!fse_read_counts(frame, at + 1u, end, 12u, 6u, counts, log, tafter)) return 0u;
        uint32_t* table = T.fse;  // (the Huffman table is built after the weights are out)
        if (!fse_build(table, counts, 13u, log, [](uint32_t) { return 0u; })) return 0u;
        if (tafter >= end || frame[end - 1u] == 0u) return 0u;
        int32_t left = static_cast<int32_t>((end - tafter) * 8u) - static_cast<int32_t>(8u - highbit(frame[end - 1u]));  // bits below the end mark
        // (short stream: read bit by bit group through a 64-bit window reloaded per read)
        auto read = [&](uint32_t nb) -> uint32_t {
            left -= static_cast<int32_t>(nb);
            if (nb == 0u) return 0u;
            if (left >= 0) {
                const uint32_t bit = tafter * 8u + static_cast<uint32_t>(left);
                const uint64_t v = ld_le64(frame + (bit >> 3));
                return static_cast<uint32_t>(v >> (bit & 7u)) & ((1u << nb) - 1u);
            }
            const int32_t have = left + static_cast<int32_t>(nb);
            if (have <= 0) return 0u;
            const uint64_t v = ld_le64(frame + tafter);
            return (static_cast<uint32_t>(v) & ((1u << have) - 1u)) << (nb - static_cast<uint32_t>(have));
        };
        uint32_t s1 = read(log), s2 = read(log);
        _Pragma("unroll 1") for (;;) {
            if (n >= 254u) return 0u;
            uint32_t e = table[s1];
            weights[n++] = static_cast<uint8_t>(e & 63u);
            s1 = (e >> 16) + read((e >> 6) & 15u);
            if (left < 0) {
                weights[n++] = static_cast<uint8_t>(table[s2] & 63u);
                break;
            }
            e = table[s2];
            weights[n++] = static_cast<uint8_t>(e & 63u);
            s2 = (e >> 16) + read((e >> 6) & 15u);
            if (left < 0) {
                weights[n++] = static_cast<uint8_t>(table[s1] & 63u);
                break;
            }
        }
        after = end;
    }
    if (n > 255u) return 0u;
    // the last weight completes the sum of 2^(w - 1) to a power of two
    uint32_t total = 0;
    _Pragma("unroll 1") for (uint32_t w = 0; w < 16u; ++w) rank[w] = 0u;
    _Pragma("unroll 1") for (uint32_t i = 0; i < n; ++i) {
        const uint32_t w = weights[i];
        if (w > 11u) return 0u;
        if (w) total += 1u << (w - 1u);
        rank[w] += 1u;
    }
    if (total == 0u) return 0u;
    const uint32_t max_bits = highbit(total) + 1u;
    if (max_bits > 11u) return 0u;
    const uint32_t rest = (1u << max_bits) - total;
    if (rest & (rest - 1u)) return 0u;
    const uint32_t lastw = highbit(rest) + 1u;
    weights[n++] = static_cast<uint8_t>(lastw);
    rank[lastw] += 1u;
    // codes of weight w take 2^(w - 1) slots each, lowest weights first, symbols in order within a weight
    uint32_t next = 0;
    _Pragma("unroll 1") for (uint32_t w = 1; w <= max_bits; ++w) {
        const uint32_t cur = next;
        next += rank[w] << (w - 1u);
        rank[w] = cur;
    }
    _Pragma("unroll 1") for (uint32_t s = 0; s < n; ++s) {
        const uint32_t w = weights[s];
        if (!w) continue;
        const uint32_t len = 1u << (w - 1u);
        const uint16_t e = static_cast<uint16_t>(s | ((max_bits + 1u - w) << 8));
        const uint32_t first = rank[w];
        _Pragma("unroll 1") for (uint32_t u = 0; u < len; ++u) T.huf[(first + u) & 2047u] = e;
        rank[w] = first + len;
    }
    return max_bits;
}

template <bool PROF>
__global__ __launch_bounds__(64) void zstd_prepare(const uint8_t* __restrict__ comp, const GpuBlock* __restrict__ blocks, uint8_t* __restrict__ scratch,
                                                   const ZLayout lay, uint32_t* __restrict__ status, unsigned long long* __restrict__ tally)
{
    __shared__ ZeLds L;
    const uint32_t fi = blockIdx.x;
    const GpuBlock gb = blocks[fi];
    const uint32_t lane = threadIdx.x;
    const uint8_t* const frame = comp + gb.src_off;
    const uint32_t n = gb.src_len, dst_len = gb.dst_len;
    ZFrameHdr* const hdr = reinterpret_cast<ZFrameHdr*>(scratch + lay.hdr_at) + fi;
    ZBlk* const blk = reinterpret_cast<ZBlk*>(scratch + lay.blk_at) + static_cast<uint64_t>(fi) * lay.blk_cap;
    uint8_t* const tabs = scratch + lay.tab_at + static_cast<uint64_t>(fi) * lay.blk_cap * kTabBytes;
    uint8_t* const lits = scratch + lay.lit_at + static_cast<uint64_t>(fi) * lay.lit_stride;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_lit = 0, t_tab = 0;
    uint32_t err = 0;
    auto fail = [&](uint32_t code) {
        if (!err) err = code;
    };
    // ---- frame header (uniform)
    uint32_t p = 0;
    if (dst_len > kZstdMaxFrameBytes || n >= (1u << 27))
        fail(kZstdTooLarge);
    else if (n < 6u)
        fail(kZstdBadHeader);
    else if (ld_le32(frame) != 0xFD2FB528u)
        fail(kZstdUnsupported);
    else {
        const uint32_t fhd = frame[4];
        const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1u;
        p = 5u + (single ? 0u : 1u);
        if (fhd & 8u)
            fail(kZstdBadHeader);
        else if (!single && (frame[5] >> 3) > 21u)   // a window above 2^31 bytes: libzstd refuses the frame
            fail(kZstdBadHeader);
        else if (fhd & 4u)
            fail(kZstdChecksum);
        else if (fhd & 3u)
            fail(kZstdDictionary);
        else {
            const uint32_t fcs_bytes = fcs_flag == 0u ? single : (1u << fcs_flag);
            if (p + fcs_bytes > n)
                fail(kZstdBadHeader);
            else if (fcs_bytes) {
                uint64_t content = 0;
                for (uint32_t i = 0; i < fcs_bytes; ++i) content |= static_cast<uint64_t>(frame[p + i]) << (8u * i);
                if (fcs_bytes == 2u) content += 256u;
                if (content != dst_len) fail(kZstdBadHeader);
                p += fcs_bytes;
            }
        }
    }
    uint32_t carry_huf = kNone, carry_tab[3] = {kNone, kNone, kNone};
    uint32_t rec_top = 0, lit_top = 0, seq_top = 0, nblk = 0;
    bool last = false;
    while (!last && !err) {
        // ---- block headers of this pass (uniform walk: a header gives the position of the next one)
        uint32_t nb = 0;
        uint32_t my_type = 0, my_size = 0, my_at = 0;
        while (nb < kZeLanes && !last) {
            if (p + 3u > n) {
                fail(kZstdBadBlock);
                break;
            }
            const uint32_t bh = ld_le24(frame + p);
            last = bh & 1u;
            const uint32_t type = (bh >> 1) & 3u, size = bh >> 3;
            const uint32_t at = p + 3u;
            const uint32_t span = type == 1u ? 1u : size;
            if (type == 3u || size > kBlockMax || at + span > n) {
                fail(kZstdBadBlock);
                break;
            }
            if (lane == nb) {
                my_type = type;
                my_size = size;
                my_at = at;
            }
            p = at + span;
            ++nb;
            if (++nblk > lay.blk_cap) {
                fail(kZstdTooManyBlocks);
                break;
            }
        }
        if (err) break;
        const bool mine = lane < nb;
        // ---- section headers, every lane its block
        uint32_t lt = 0, nlit = 0, lit_data = 0, lit_end = 0, streams = 0, nseq = 0, bits_at = 0, bend = 0;
        uint32_t def_huf = kNone, def_tab[3] = {kNone, kNone, kNone};
        uint32_t lerr = 0;
        if (mine && my_type == 2u) {
            bend = my_at + my_size;
            if (my_size < 2u)
                lerr = kZstdBadBlock;
            else {
                const uint32_t b0 = frame[my_at];
                lt = b0 & 3u;
                const uint32_t fmt = (b0 >> 2) & 3u;
                if (lt < 2u) {
                    uint32_t hl;
                    if (fmt == 1u) {
                        nlit = ld_le16(frame + my_at) >> 4;
                        hl = 2u;
                    } else if (fmt == 3u) {
                        nlit = ld_le24(frame + my_at) >> 4;
                        hl = 3u;
                    } else {
                        nlit = b0 >> 3;
                        hl = 1u;
                    }
                    lit_data = my_at + hl;
                    lit_end = lit_data + (lt == 0u ? nlit : 1u);
                } else {
                    uint32_t csize, hl;
                    const uint64_t h = ld_le64(frame + my_at);
                    if (fmt < 2u) {
                        nlit = static_cast<uint32_t>(h >> 4) & 1023u;
                        csize = static_cast<uint32_t>(h >> 14) & 1023u;
                        hl = 3u;
                    } else if (fmt == 2u) {
                        nlit = static_cast<uint32_t>(h >> 4) & 16383u;
                        csize = static_cast<uint32_t>(h >> 18) & 16383u;
                        hl = 4u;
                    } else {
                        nlit = static_cast<uint32_t>(h >> 4) & 262143u;
                        csize = static_cast<uint32_t>(h >> 22) & 262143u;
                        hl = 5u;
                    }
                    streams = fmt == 0u ? 1u : 4u;
                    lit_data = my_at + hl;
                    lit_end = lit_data + csize;
                    if (lt == 2u) def_huf = lit_data;
                }
                if (lit_end > bend || nlit > kBlockMax)
                    lerr = kZstdBadLiterals;
                else if (lit_end >= bend)
                    lerr = kZstdBadSequences;
                else {
                    uint32_t q = lit_end;
                    const uint32_t s0 = frame[q];
                    if (s0 == 0u) {
                        if (q + 1u != bend) lerr = kZstdBadSequences;
                    } else {
                        if (s0 < 128u) {
                            nseq = s0;
                            q += 1u;
                        } else if (s0 < 255u) {
                            if (q + 2u > bend) lerr = kZstdBadSequences;
                            nseq = ((s0 - 128u) << 8) + frame[q + 1u];
                            q += 2u;
                        } else {
                            if (q + 3u > bend) lerr = kZstdBadSequences;
                            nseq = ld_le16(frame + q + 1u) + 0x7F00u;
                            q += 3u;
                        }
                        if (!lerr && (q >= bend || nseq > kBlockMax / 3u)) lerr = kZstdBadSequences;
                        if (!lerr) {
                            const uint32_t modes = frame[q];
                            q += 1u;
                            if (modes & 3u) lerr = kZstdBadSequences;
                            const uint32_t max_sym[3] = {35u, 31u, 52u}, max_log[3] = {9u, 8u, 9u};
                            for (uint32_t t = 0; t < 3u && !lerr; ++t) {
                                const uint32_t mode = (modes >> (6u - 2u * t)) & 3u;
                                if (mode == 3u) {
                                    def_tab[t] = kRepeat;
                                    continue;
                                }
                                def_tab[t] = (mode << 28) | q;
                                if (mode == 1u) {
                                    if (q >= bend) lerr = kZstdBadSequences;
                                    q += 1u;
                                } else if (mode == 2u) {
                                    uint32_t lg, after;
                                    if (!fse_read_counts(frame, q, bend, max_sym[t], max_log[t], L.counts[lane], lg, after))
                                        lerr = kZstdBadSequences;
                                    else
                                        q = after;
                                }
                            }
                            bits_at = q;
                        }
                    }
                }
            }
        } else if (mine) {
            nlit = my_type == 0u ? my_size : (my_size ? 1u : 0u);
        }
        if (__builtin_amdgcn_ballot_w64(lerr != 0u)) {
            const uint32_t first = static_cast<uint32_t>(__builtin_ctzll(__builtin_amdgcn_ballot_w64(lerr != 0u)));
            fail(__builtin_amdgcn_readlane(lerr, first));
            break;
        }
        // ---- scratch placement: records (a multiple of 64 per block), literals
        const uint32_t rec_cap = mine ? (my_type == 2u ? (nseq + 40u + 63u) & ~63u : 64u) : 0u;
        const uint32_t rec_incl = wave_scan_add(rec_cap), lit_incl = wave_scan_add(mine ? nlit : 0u);
        const uint32_t rec_at = rec_top + rec_incl - rec_cap, lit_at = lit_top + lit_incl - (mine ? nlit : 0u);
        rec_top += __builtin_amdgcn_readlane(rec_incl, 63);
        lit_top += __builtin_amdgcn_readlane(lit_incl, 63);
        if (rec_top > lay.rec_stride || lit_top > lay.lit_stride - 16u) {
            fail(kZstdBadSize);
            break;
        }
        // ---- what the others need to know
        if (mine) {
            ZeShare& s = L.sh[lane];
            s.def_huf = def_huf;
            s.def_tab[0] = def_tab[0];
            s.def_tab[1] = def_tab[1];
            s.def_tab[2] = def_tab[2];
            s.type = my_type;
            s.size = my_size;
            s.at = my_at;
            s.lit_type = lt;
            s.nlit = nlit;
            s.lit_at = lit_at;
            s.lit_data = lit_data;
            s.lit_end = lit_end;
            s.streams = streams;
            s.rec_at = rec_at;
            s.rec_cap = rec_cap;
        }
        __syncthreads();  // (one wave: orders the LDS writes above before the reads below)
        // sources of the tables this block repeats: the nearest earlier definition
        uint32_t huf_src = def_huf, tab_src[3] = {def_tab[0], def_tab[1], def_tab[2]};
        if (mine && my_type == 2u) {
            if (lt == 3u) {
                huf_src = carry_huf;
                for (uint32_t j = 0; j < lane; ++j)
                    if (L.sh[j].def_huf != kNone) huf_src = L.sh[j].def_huf;
                if (huf_src == kNone) lerr = kZstdNoTable;
            }
            for (uint32_t t = 0; t < 3u; ++t)
                if (def_tab[t] == kRepeat) {
                    uint32_t src = carry_tab[t];
                    for (uint32_t j = 0; j < lane; ++j)
                        if (L.sh[j].def_tab[t] < kRepeat) src = L.sh[j].def_tab[t];
                    if (src == kNone) lerr = kZstdNoTable;
                    tab_src[t] = src;
                }
        }
        for (uint32_t j = 0; j < nb; ++j) {
            if (L.sh[j].def_huf != kNone) carry_huf = L.sh[j].def_huf;
            for (uint32_t t = 0; t < 3u; ++t)
                if (L.sh[j].def_tab[t] < kRepeat) carry_tab[t] = L.sh[j].def_tab[t];
        }
        // ---- literals: Huffman tables (a lane per block), then the streams (four lanes per block)
        const unsigned long long t0 = PROF ? __builtin_readcyclecounter() : 0ull;
        if (mine && my_type == 2u && lt >= 2u && !lerr) {
            uint32_t after = 0;
            const uint32_t mb = huf_build(frame, huf_src, lt == 2u ? lit_end : n, L.tab[lane], L.weights[lane], L.counts[lane], L.rank[lane], after);
            if (!mb) lerr = kZstdBadHuffman;
            L.sh[lane].max_bits = mb;
            L.sh[lane].stream_at = lt == 2u ? after : lit_data;
        }
        __syncthreads();
        {
            const uint32_t bk = lane >> 2, st = lane & 3u;
            bool act = false;
            uint32_t s_begin = 0, s_end = 0, count = 0, mb = 0, o_at = 0;
            uint32_t herr = 0;
            if (bk < nb && L.sh[bk].type == 2u && L.sh[bk].lit_type >= 2u && L.sh[bk].max_bits) {
                const ZeShare& s = L.sh[bk];
                mb = s.max_bits;
                if (s.streams == 1u) {
                    act = st == 0u;
                    s_begin = s.stream_at;
                    s_end = s.lit_end;
                    count = s.nlit;
                    o_at = s.lit_at;
                } else if (s.stream_at + 6u > s.lit_end) {
                    herr = kZstdBadLiterals;
                } else {
                    const uint32_t j1 = ld_le16(frame + s.stream_at), j2 = ld_le16(frame + s.stream_at + 2u), j3 = ld_le16(frame + s.stream_at + 4u);
                    const uint32_t b0 = s.stream_at + 6u, b1 = b0 + j1, b2 = b1 + j2, b3 = b2 + j3;
                    const uint32_t per = (s.nlit + 3u) >> 2;
                    if (b3 >= s.lit_end || per * 3u > s.nlit)
                        herr = kZstdBadLiterals;
                    else {
                        act = true;
                        s_begin = st == 0u ? b0 : (st == 1u ? b1 : (st == 2u ? b2 : b3));
                        s_end = st == 0u ? b1 : (st == 1u ? b2 : (st == 2u ? b3 : s.lit_end));
                        count = st < 3u ? per : s.nlit - 3u * per;
                        o_at = s.lit_at + st * per;
                    }
                }
                if (act && (s_end <= s_begin || frame[s_end - 1u] == 0u)) {
                    herr = kZstdBadLiterals;
                    act = false;
                }
            }
            // backward stream: `pos` bits are unread; the window is the 8 bytes that end with the byte pos falls into
            int32_t pos = act ? static_cast<int32_t>(8u * (s_end - 1u) + highbit(frame[s_end - 1u])) : 0;
            const int32_t start_bit = static_cast<int32_t>(8u * s_begin);
            const uint16_t* const table = L.tab[bk & (kZeLanes - 1u)].huf;
            uint8_t* o = lits + o_at;
            uint32_t left = act ? count : 0u;
            while (__builtin_amdgcn_ballot_w64(left > 0u)) {
                if (left > 0u) {
                    const int32_t byte = pos >> 3;
                    const int32_t lo = byte - 7;                   // (>= 0: a stream starts behind at least 9 bytes of headers)
                    uint64_t v = ld_le64(frame + (lo < 0 ? 0 : lo));
                    const int32_t cut = start_bit - 8 * lo;        // bits of the window below the stream read as zero
                    if (cut > 0) v = cut >= 64 ? 0ull : (v >> cut) << cut;
                    uint64_t w = v << (8u - (static_cast<uint32_t>(pos) & 7u));
                    const uint32_t take_n = left < 4u ? left : 4u;
                    uint32_t word = 0;
                    for (uint32_t k = 0; k < take_n; ++k) {
                        const uint32_t e = table[static_cast<uint32_t>(w >> (64u - mb)) & 2047u];
                        word |= (e & 255u) << (8u * k);
                        w <<= (e >> 8);
                        pos -= static_cast<int32_t>(e >> 8);
                    }
                    if (take_n == 4u) {
                        __builtin_memcpy(o, &word, 4);
                    } else {
                        for (uint32_t k = 0; k < take_n; ++k) o[k] = static_cast<uint8_t>(word >> (8u * k));
                    }
                    o += take_n;
                    left -= take_n;
                    if (pos < start_bit - 64) left = 0u;   // (far past the start: the stream is damaged, checked below)
                }
            }
            if (act && pos != start_bit) herr = kZstdBadLiterals;
            if (herr && !lerr) lerr = herr;
        }
        // raw / RLE literals, raw / RLE blocks: the bytes as they are (the whole wave, block after block)
        for (uint32_t j = 0; j < nb; ++j) {
            const ZeShare& s = L.sh[j];
            uint32_t from = 0, cnt = 0;
            bool fill = false;
            if (s.type == 0u) {
                from = s.at;
                cnt = s.size;
            } else if (s.type == 1u) {
                from = s.at;
                cnt = s.nlit;
            } else if (s.lit_type == 0u) {
                from = s.lit_data;
                cnt = s.nlit;
            } else if (s.lit_type == 1u) {
                from = s.lit_data;
                cnt = s.nlit;
                fill = true;
            }
            for (uint32_t i = lane; i < cnt; i += 64u) lits[s.lit_at + i] = frame[fill ? from : from + i];
        }
        if (PROF) t_lit += __builtin_readcyclecounter() - t0;
        __syncthreads();
        // ---- sequences: tables, then the serial chain of every block in its lane
        const unsigned long long t1 = PROF ? __builtin_readcyclecounter() : 0ull;
        uint32_t logl = 0, logo = 0, logm = 0;
        bool seq_act = mine && my_type == 2u && nseq > 0u && !lerr;
        // (1) a lane per block: the symbol counts of its three tables into the lane's own LDS area (the Huffman table that lay there
        // has done its work); (2) the whole wave builds table after table, block after block (fse_build_wave), and sends them off
        // at once: to the chain kernel next-state number | extra bits << 10, to the records kernel the symbols.
        uint32_t info[3] = {~0u, ~0u, ~0u};
        if (seq_act) {
            int16_t* const cnt = reinterpret_cast<int16_t*>(L.tab[lane].huf);
            info[0] = seq_counts(frame, n, tab_src[0], cnt, kLLDefault, 36u, 6u, 35u, 9u);
            info[1] = seq_counts(frame, n, tab_src[1], cnt + 64, kOFDefault, 29u, 5u, 31u, 8u);
            info[2] = seq_counts(frame, n, tab_src[2], cnt + 128, kMLDefault, 53u, 6u, 52u, 9u);
            if (info[0] == ~0u || info[1] == ~0u || info[2] == ~0u) {
                lerr = kZstdBadSequences;
                seq_act = false;
            } else if (bend <= bits_at || frame[bend - 1u] == 0u) {
                lerr = kZstdBadBitstream;
                seq_act = false;
            }
        }
        __syncthreads();
        {
            // extra bits of the symbol whose number is the lane's, per table
            const uint32_t xb_t[3] = {lane < 36u ? static_cast<uint32_t>(kLLBits[lane]) : 0u, lane, lane < 53u ? static_cast<uint32_t>(kMLBits[lane]) : 0u};
            const uint32_t at16[3] = {0u, kTabOF / 2u, kTabML / 2u}, sym_at[3] = {kTabSymLL, 0u, kTabSymML};
            const unsigned long long act = __builtin_amdgcn_ballot_w64(seq_act);
            bool built = true;
            for (uint32_t j = 0; j < nb; ++j) {
                if (!((act >> j) & 1ull)) continue;
                uint8_t* const tg = tabs + static_cast<uint64_t>(nblk - nb + j) * kTabBytes;
                uint16_t* const t16 = reinterpret_cast<uint16_t*>(tg);
                uint8_t* const area = reinterpret_cast<uint8_t*>(L.tab[j].huf);
                for (uint32_t t = 0; t < 3u; ++t) {
                    const uint32_t inf = __builtin_amdgcn_readlane(info[t], j);
                    if (inf & 0x10000u) {   // one symbol: a table of one cell
                        const uint32_t sym = inf >> 24;
                        const uint32_t xbs = t == 0u ? (sym < 36u ? kLLBits[sym] : 0u) : (t == 1u ? sym : (sym < 53u ? kMLBits[sym] : 0u));
                        if (lane == 0u) {
                            t16[at16[t]] = static_cast<uint16_t>(1u | (xbs << 10));
                            if (sym_at[t]) tg[sym_at[t]] = static_cast<uint8_t>(sym);
                        }
                        if (lane == j) info[t] = 0u;   // (its accuracy log)
                        continue;
                    }
                    const bool ok = fse_build_wave(reinterpret_cast<const int16_t*>(area) + 64u * t, (inf >> 8) & 255u, inf & 255u, lane, xb_t[t], t16 + at16[t],
                                                   sym_at[t] ? tg + sym_at[t] : nullptr, area + 512u);
                    built = built && ok;
                    if (lane == j) info[t] &= 255u;
                }
            }
            if (!built && !lerr && seq_act) lerr = kZstdBadSequences;   // (`built` is uniform: every active lane of a pass reports)
            if (!built) seq_act = false;
            if (seq_act) {
                logl = info[0];
                logo = info[1];
                logm = info[2];
            }
        }
        if (PROF) t_tab += __builtin_readcyclecounter() - t1;
        if (__builtin_amdgcn_ballot_w64(lerr != 0u)) {
            const uint32_t first = static_cast<uint32_t>(__builtin_ctzll(__builtin_amdgcn_ballot_w64(lerr != 0u)));
            fail(__builtin_amdgcn_readlane(lerr, first));
            break;
        }
        // ---- the block's description
        const uint32_t seq_incl = wave_scan_add(mine ? nseq : 0u);
        if (mine) {
            ZBlk d;
            d.type = my_type;
            d.size = my_size;
            d.at = my_at;
            d.nseq = nseq;
            d.nlit = nlit;
            d.lit_at = lit_at;
            d.rec_at = rec_at;
            d.rec_cap = rec_cap;
            d.stash_at = seq_top + seq_incl - nseq;
            d.bits_at = bits_at;
            d.bend = bend;
            d.logs = logl | (logo << 8) | (logm << 16);
            d.chain_err = 0u;
            d.nrec = 0u;
            d.out_len = 0u;
            d.out_at = 0u;
            d.n_sym = 0u;
            d.hist_known = 0u;
            d.hist[0] = d.hist[1] = d.hist[2] = 0u;
            blk[nblk - nb + lane] = d;
        }
        seq_top += __builtin_amdgcn_readlane(seq_incl, 63);
        if (seq_top > lay.rec_stride) {
            fail(kZstdBadSize);
            break;
        }
        __syncthreads();
    }
    if (!err && p != n) fail(kZstdTrailingData);
    if (lane == 0u) {
        hdr->nslots = err ? 0u : rec_top >> 6;
        hdr->nrec = 0u;
        hdr->out_len = 0u;
        hdr->nblk = err ? 0u : nblk;
        status[fi] = err;
        if (PROF) {
            atomicAdd(&tally[17], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
            atomicAdd(&tally[18], t_lit);
            atomicAdd(&tally[19], t_tab);
        }
    }
}

// ------------------------------------------------------------------------------------------------ zstd_chain
// The serial part of a block's sequences section and nothing else: three FSE states -> three table entries and the bit
// window (ONE LDS round trip) -> the bits of the next states.  A wave costs the same with one lane or sixty-four, and a
// block's chain tables take 2.5 KB of LDS, so a wave walks the blocks of TWO frames at once, a lane per block; what a step
// saw (the three states, 64 bits of the window) goes to the stash in global memory for zstd_records.
constexpr uint32_t kZcFrames = 2, kZcLanes = 8u * kZcFrames;
// Waves per workgroup: THREE, each with its own two frames and its own 49 KB of LDS, so that a workgroup fills a CU's LDS and
// a launch of W waves takes W / 3 CUs whole instead of leaving one or two waves (49-98 KB) on every CU of the chip, beside
// which not one 80 KB workgroup of the other stream's execution kernel fits.  The waves never meet: no barrier between them.
#ifndef FLAGSTAT_ZSTD_CHAIN_WAVES
#define FLAGSTAT_ZSTD_CHAIN_WAVES 3
#endif
constexpr uint32_t kZcWaves = FLAGSTAT_ZSTD_CHAIN_WAVES;
constexpr uint32_t kZcRing = 512, kZcChunk = 128;   // bytes of a block's bit stream staged in LDS, bytes a refill

struct __attribute__((aligned(16))) ZcLds {
    uint16_t tab[kZcLanes][1280];          // LL 512, OF 256, ML 512 entries: next-state number | extra bits << 10
    uint8_t ring[kZcLanes][kZcRing + 16];  // frame bytes [rlo, rlo + kZcRing) at index (byte & (kZcRing - 1)), the first 16 once more behind the end
};

template <bool PROF>
__global__ __launch_bounds__(64 * kZcWaves) void zstd_chain(const uint8_t* __restrict__ comp, const GpuBlock* __restrict__ blocks, uint8_t* __restrict__ scratch,
                                                 const ZLayout lay, const uint32_t* __restrict__ status, unsigned long long* __restrict__ tally)
{
    __shared__ ZcLds Lw[kZcWaves];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ZcLds& L = Lw[wave];
    const uint32_t pair = blockIdx.x * kZcWaves + wave;   // this wave's two frames
    const uint32_t lane = threadIdx.x & 63u;
    // (orders this wave's LDS writes before its reads; the LDS operations of one wave execute in order)
    auto wave_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    const uint32_t slot = lane >> 3, bk = lane & 7u;
    const uint32_t fi = pair * kZcFrames + slot;
    const bool fact = lane < kZcLanes && fi < lay.nframes && status[fi] == 0u;
    const GpuBlock gb = fact ? blocks[fi] : GpuBlock{0, 0, 0, 0};
    const uint8_t* const frame = comp + gb.src_off;
    const uint32_t n = gb.src_len;
    ZBlk* const blk = reinterpret_cast<ZBlk*>(scratch + lay.blk_at) + static_cast<uint64_t>(fact ? fi : 0u) * lay.blk_cap;
    uint32_t* const stash = reinterpret_cast<uint32_t*>(scratch + lay.stash_at) + static_cast<uint64_t>(fact ? fi : 0u) * lay.rec_stride * 3u;
    const uint32_t nblk = fact ? reinterpret_cast<const ZFrameHdr*>(scratch + lay.hdr_at)[fi].nblk : 0u;
    // the two frames' payloads, wave-uniform (lanes 0 and 8 hold them; a frame that is not walked has length 0 and no block asks)
    const uint64_t off_a = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(gb.src_off)), 0))) |
                           (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(gb.src_off >> 32)), 0))) << 32);
    const uint64_t off_b = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(gb.src_off)), 8))) |
                           (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(gb.src_off >> 32)), 8))) << 32);
    const int32_t n_a = __builtin_amdgcn_readlane(static_cast<int>(n), 0), n_b = __builtin_amdgcn_readlane(static_cast<int>(n), 8);
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    unsigned long long n_steps = 0, t_refill = 0, n_refill = 0;
    // passes of eight blocks per frame, as many as the frame with the most blocks needs
    uint32_t passes = (nblk + 7u) >> 3;
    passes = __builtin_amdgcn_readlane(wave_scan_max(passes), 63);
    for (uint32_t pass = 0; pass < passes; ++pass) {
        const uint32_t b = pass * 8u + bk;
        const bool have = fact && b < nblk;
        const ZBlk d = have ? blk[b] : ZBlk{};
        bool act = have && d.type == 2u && d.nseq > 0u;
        const uint32_t nseq = act ? d.nseq : 0u;
        const uint32_t logl = d.logs & 255u, logo = (d.logs >> 8) & 255u, logm = (d.logs >> 16) & 255u;
        // ---- the chain tables of the active blocks: global -> LDS, the whole wave, block after block
        {
            uint64_t m = __builtin_amdgcn_ballot_w64(act);
            while (m) {
                const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(m));
                m &= m - 1ull;
                const uint32_t jf = pair * kZcFrames + (j >> 3), jb = pass * 8u + (j & 7u);
                const uint4* const src = reinterpret_cast<const uint4*>(scratch + lay.tab_at + (static_cast<uint64_t>(jf) * lay.blk_cap + jb) * kTabBytes);
                uint4* const dst = reinterpret_cast<uint4*>(L.tab[j]);
                for (uint32_t q = lane; q < 2560u / 16u; q += 64u) dst[q] = src[q];
            }
        }
        // ---- the bit stream: `pos` bits are unread; staged through the ring, refilled downwards by the whole wave
        uint32_t lerr = 0;
        if (act && (d.bend <= d.bits_at || d.bend > n || frame[d.bend - 1u] == 0u)) {
            lerr = kZstdBadBitstream;
            act = false;
        }
        int32_t pos = act ? static_cast<int32_t>(8u * (d.bend - 1u) + highbit(frame[d.bend - 1u])) : 0;
        const int32_t start_bit = static_cast<int32_t>(8u * d.bits_at);
        int32_t rlo = act ? ((pos >> 3) & ~static_cast<int32_t>(kZcChunk - 1u)) + static_cast<int32_t>(kZcChunk) : 0;
        // Refill: block j's chunk is 128 bytes = eight 16-byte pieces; lane l serves piece (l & 7) of block (l >> 3) (frame A's
        // blocks) AND of block 8 + (l >> 3) (frame B's), so ONE pass serves all sixteen blocks with both global loads in flight
        // together (a loop over the asking blocks, one dependent load each, paid a full memory latency per block -- and twice
        // that while a copy from the host was running).
        auto refill = [&](bool want) {
            // lanes that want a chunk and have room for it: the bytes it overwrites lie above the window
            const bool can = want && ((pos >> 3) - (rlo - static_cast<int32_t>(kZcChunk)) <= static_cast<int32_t>(kZcRing) - 1);
            const uint32_t m = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(can)) & ((1u << kZcLanes) - 1u);
            if (m) {
                const uint32_t ba = lane >> 3, bb = 8u + (lane >> 3), piece = 16u * (lane & 7u);
                const int32_t nlo_a = __builtin_amdgcn_ds_bpermute(static_cast<int>(4u * ba), rlo) - static_cast<int32_t>(kZcChunk);
                const int32_t nlo_b = __builtin_amdgcn_ds_bpermute(static_cast<int>(4u * bb), rlo) - static_cast<int32_t>(kZcChunk);
                const bool sa = (m >> ba) & 1u, sb = (m >> bb) & 1u;
                typedef uint32_t v4u1 __attribute__((ext_vector_type(4), aligned(1)));
                typedef uint32_t v4u16 __attribute__((ext_vector_type(4)));
                // (never below the buffer, never beyond the 64 readable bytes behind the payload: what lies there is never used)
                auto source = [&](const uint64_t joff, const int32_t jn, const int32_t nlo) -> const uint8_t* {
                    int32_t rel = nlo + static_cast<int32_t>(piece);
                    if (rel > jn + 48) rel = jn + 48;
                    const int64_t src = static_cast<int64_t>(joff) + rel;
                    return comp + (src < 0 ? 0 : src);
                };
                v4u16 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
                if (sa) va = *reinterpret_cast<const v4u1*>(source(off_a, n_a, nlo_a));
                if (sb) vb = *reinterpret_cast<const v4u1*>(source(off_b, n_b, nlo_b));
                if (sa) {
                    const uint32_t ri = (static_cast<uint32_t>(nlo_a) & (kZcRing - 1u)) + piece;
                    *reinterpret_cast<v4u16*>(&L.ring[ba][ri]) = va;
                    if (ri == 0u) *reinterpret_cast<v4u16*>(&L.ring[ba][kZcRing]) = va;
                }
                if (sb) {
                    const uint32_t ri = (static_cast<uint32_t>(nlo_b) & (kZcRing - 1u)) + piece;
                    *reinterpret_cast<v4u16*>(&L.ring[bb][ri]) = vb;
                    if (ri == 0u) *reinterpret_cast<v4u16*>(&L.ring[bb][kZcRing]) = vb;
                }
            }
            if (can) rlo -= static_cast<int32_t>(kZcChunk);
        };
        for (int k = 0; k < 4; ++k) refill(act);
        wave_sync();
        const uint32_t home = lane & (kZcLanes - 1u);
        const uint16_t* const T = L.tab[home];
        const uint32_t sizel = 1u << logl, sizeo = 1u << logo, sizem = 1u << logm;
        // the window: 16 bytes that end with the byte the position falls into, as four dwords; (pos & 7) + 120 of its bits are unread
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        auto window = [&](v4u& dw) {
            const uint32_t ri = static_cast<uint32_t>((pos >> 3) - 15) & (kZcRing - 1u);
            typedef uint32_t v4u1 __attribute__((ext_vector_type(4), aligned(1)));
            dw = *reinterpret_cast<const v4u1*>(&L.ring[home][ri]);
        };
        uint32_t sl = 0, so = 0, sm = 0;
        if (act) {
            v4u dw;
            window(dw);
            const uint32_t sh = static_cast<uint32_t>(pos) & 7u;
            uint64_t w = (static_cast<uint64_t>(__builtin_amdgcn_alignbit(dw[3], dw[2], sh + 24u)) << 32) | __builtin_amdgcn_alignbit(dw[2], dw[1], sh + 24u);
            sl = take(w, logl) & 511u;
            so = take(w, logo) & 255u;
            sm = take(w, logm) & 511u;
            pos -= static_cast<int32_t>(logl + logo + logm);
            if (pos < start_bit) {
                lerr = kZstdBadBitstream;
                act = false;
            }
        }
        // One step: `update` = the states move on (every sequence but a block's last).  The entries and the window of a step
        // are loaded at the END of the step before it, ahead of that step's book-keeping (stash store, counters), which so
        // runs while the LDS round trip is under way.
        uint32_t* my_st = stash + static_cast<uint64_t>(d.stash_at) * 3u;   // this lane's next stash entry: states | window (two dwords)
        uint32_t i = 0;
        struct Loaded {   // what a step needs from LDS: its three table entries and the window
            uint32_t el, eo, em;
            v4u dw;
        };
        Loaded la, lb;   // (two sets, used in turn: the loads of a step land in the set the step after it reads, no copies)
        la.el = T[sl];
        la.eo = T[512u + so];
        la.em = T[768u + sm];
        window(la.dw);
        lb = la;   // (only so that nothing is read uninitialised)
        auto chain_step = [&](const bool step, const bool update, const Loaded& in, Loaded& out) {
            const uint32_t el = in.el, eo = in.eo, em = in.em;
            const v4u dw = in.dw;
            const uint32_t oc = eo >> 10, mlb = em >> 10, llb = el >> 10;
            const uint32_t xl = el & 1023u, xo = eo & 1023u, xm = em & 1023u;
            // bits of the next state: log - highbit(next-state number)
            const uint32_t nbl = update ? logl - highbit(xl) : 0u, nbm = update ? logm - highbit(xm) : 0u, nbo = update ? logo - highbit(xo) : 0u;
            const uint32_t ext = oc + mlb + llb;
            const uint32_t sh = static_cast<uint32_t>(pos) & 7u;
            // the state bits follow the extra bits: the 32 bits of the window below bit t - ext (<= 26 are used), t = sh + 120
            const uint32_t f = sh + 88u - ext;   // t - ext - 32: 25..95
            const uint32_t fk = f >> 5;
            const uint32_t x_lo = fk == 0u ? dw[0] : (fk == 1u ? dw[1] : dw[2]), x_hi = fk == 0u ? dw[1] : (fk == 1u ? dw[2] : dw[3]);
            const uint32_t x = __builtin_amdgcn_alignbit(x_hi, x_lo, f & 31u);
            const uint32_t o1 = 32u - nbl, o2 = o1 - nbm, o3 = o2 - nbo;
            const uint32_t bl = __builtin_amdgcn_ubfe(x, o1, nbl), bm = __builtin_amdgcn_ubfe(x, o2, nbm), bo = __builtin_amdgcn_ubfe(x, o3, nbo);
            if (step) {
                const uint32_t packed = sl | (so << 9) | (sm << 17);
                sl = (xl << nbl) - sizel + bl;
                sm = (xm << nbm) - sizem + bm;
                so = (xo << nbo) - sizeo + bo;
                pos -= static_cast<int32_t>(ext + 32u - o3);
                if (update) {
                    // the next step's reads (a block whose stream ran out reads below its start: inside the ring, never used)
                    out.el = T[sl];
                    out.eo = T[512u + so];
                    out.em = T[768u + sm];
                    window(out.dw);
                }
                // ... and under them this step's book-keeping
                typedef uint32_t v3u __attribute__((ext_vector_type(3)));
                v3u e;
                e.x = packed;
                e.y = __builtin_amdgcn_alignbit(dw[2], dw[1], sh + 24u);   // bits [t - 64, t) with t = sh + 120: the extra bits of the
                e.z = __builtin_amdgcn_alignbit(dw[3], dw[2], sh + 24u);   // three codes lead them (for zstd_records)
                *reinterpret_cast<v3u*>(my_st) = e;
                my_st += 3;
                ++i;
                if (PROF) ++n_steps;
                if (pos < start_bit) {
                    lerr = kZstdBadBitstream;
                    act = false;
                }
            }
        };
        const uint32_t nupd = nseq ? nseq - 1u : 0u;   // steps that move the states on
        while (__builtin_amdgcn_ballot_w64(act && i < nupd)) {
            // (a lane whose window would run out of staged bytes within five steps asks for a chunk; the wave copies -- and the
            // lanes the refill served load their window again: it was read before the chunk arrived)
            if (__builtin_amdgcn_ballot_w64(act && i < nupd && (pos >> 3) - rlo < 96)) {
                const unsigned long long t0 = PROF ? __builtin_readcyclecounter() : 0ull;
                refill(act && i < nupd && (pos >> 3) - rlo < static_cast<int32_t>(kZcRing / 2u));
                window(la.dw);
                if (PROF) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    t_refill += __builtin_readcyclecounter() - t0;
                    ++n_refill;
                }
            }
            // (every lane that still steps has taken a multiple of four steps here: its next step reads `la`)
            chain_step(act && i < nupd, true, la, lb);
            chain_step(act && i < nupd, true, lb, la);
            chain_step(act && i < nupd, true, la, lb);
            chain_step(act && i < nupd, true, lb, la);
        }
        // the last sequence of every block: its codes and extra bits, no bits for further states.  What its step reads was
        // loaded by the step before it into the set that step wrote: `la` after an even number of steps, `lb` after an odd one.
        Loaded fin;
        fin.el = (nupd & 1u) ? lb.el : la.el;
        fin.eo = (nupd & 1u) ? lb.eo : la.eo;
        fin.em = (nupd & 1u) ? lb.em : la.em;
        if (__builtin_amdgcn_ballot_w64(act && (pos >> 3) - rlo < 96)) refill(act && (pos >> 3) - rlo < static_cast<int32_t>(kZcRing / 2u));
        window(fin.dw);
        chain_step(act && i < nseq, false, fin, lb);
        if (have && d.type == 2u && d.nseq > 0u && !lerr && pos != start_bit) lerr = kZstdBadBitstream;
        if (have && lerr) blk[b].chain_err = lerr;
        wave_sync();
    }
    if (PROF && lane == 0u) {
        atomicAdd(&tally[20], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        atomicAdd(&tally[21], 1ull);
        atomicAdd(&tally[26], t_refill);
        atomicAdd(&tally[27], n_refill);
    }
    if (PROF) atomicAdd(&tally[22], n_steps);
}

// ------------------------------------------------------------------------------------------------ zstd_records
// What the chains left behind becomes records: a workgroup per frame, a wave per block, a lane per sequence, 64 at a time:
// symbols from the states, values from the codes and the window, positions by prefix sums, repeat offsets by relaxation
// over the lanes (a lane's history is its left neighbour's after the neighbour's sequence; three plain offsets in a row
// fix it whatever came before, so a few rounds settle all 64), records and checkpoints.  Then, with every block of the
// frame done: the repeat codes a block could not resolve because they reach into the block before it are replayed in frame
// order by one wave, and the checkpoints get their positions in the frame.
constexpr uint32_t kZvWaves = 8;
constexpr uint32_t kUnknown = 1u << 31;   // an offset-history slot whose value the block does not know (yet)

struct ZvLds {
    uint32_t ll_base[36], ml_base[53];  // value | extra bits << 24
    uint32_t err;
};

template <bool PROF>
__global__ __launch_bounds__(64 * kZvWaves) void zstd_records(const GpuBlock* __restrict__ blocks, uint8_t* __restrict__ scratch, const ZLayout lay,
                                                              uint32_t* __restrict__ status, unsigned long long* __restrict__ tally)
{
    __shared__ ZvLds L;
    const uint32_t fi = blockIdx.x;
    if (status[fi] != 0u) return;  // (uniform for the workgroup)
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t dst_len = blocks[fi].dst_len;
    ZFrameHdr* const hdr = reinterpret_cast<ZFrameHdr*>(scratch + lay.hdr_at) + fi;
    ZBlk* const blk = reinterpret_cast<ZBlk*>(scratch + lay.blk_at) + static_cast<uint64_t>(fi) * lay.blk_cap;
    const uint8_t* const tabs = scratch + lay.tab_at + static_cast<uint64_t>(fi) * lay.blk_cap * kTabBytes;
    uint4* const ck = reinterpret_cast<uint4*>(scratch + lay.ck_at) + static_cast<uint64_t>(fi) * lay.ck_stride;
    uint64_t* const recs = reinterpret_cast<uint64_t*>(scratch + lay.rec_at) + static_cast<uint64_t>(fi) * lay.rec_stride;
    const uint32_t* const stash = reinterpret_cast<const uint32_t*>(scratch + lay.stash_at) + static_cast<uint64_t>(fi) * lay.rec_stride * 3u;
    const uint32_t nblk = hdr->nblk <= lay.blk_cap ? hdr->nblk : 0u;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    unsigned long long n_rounds = 0, n_batches = 0;
    if (threadIdx.x < 36u) L.ll_base[threadIdx.x] = kLLBase[threadIdx.x] | (static_cast<uint32_t>(kLLBits[threadIdx.x]) << 24);
    if (threadIdx.x < 53u) L.ml_base[threadIdx.x] = kMLBase[threadIdx.x] | (static_cast<uint32_t>(kMLBits[threadIdx.x]) << 24);
    if (threadIdx.x == 0u) L.err = 0u;
    __syncthreads();
    for (uint32_t b = wave; b < nblk; b += kZvWaves) {
        const ZBlk d = blk[b];
        const uint8_t* const tg = tabs + static_cast<uint64_t>(b) * kTabBytes;
        // (all of this wave-uniform) the block's records, output and literal bytes so far; its view of the offset history
        uint32_t nrec = 0, out = 0, lit_pos = 0, n_sym = 0;
        uint32_t h0 = kUnknown, h1 = kUnknown, h2 = kUnknown;   // (a slot this block does not know yet: bit 31)
        uint32_t berr = d.chain_err;
        const uint32_t rec_at = d.rec_at, rec_cap = d.rec_cap;
        auto put = [&](uint32_t w0, uint32_t ll, uint32_t ml) {   // one record, by lane 0
            if (lane == 0u && nrec < rec_cap) {
                if ((nrec & 63u) == 0u) ck[(rec_at + nrec) >> 6] = make_uint4(out, lit_pos, 0u, 0u);
                recs[rec_at + nrec] = static_cast<uint64_t>(w0) | (static_cast<uint64_t>(ll | (ml << 14)) << 32);
            }
            ++nrec;
            out += ll + ml;
            lit_pos += ll;
        };
        auto literal_run = [&](uint32_t count) {
            while (count) {
                const uint32_t piece = count < kRecLL ? count : kRecLL;
                put(kRecFlag, piece, 0u);
                count -= piece;
            }
        };
        if (d.type == 0u) {
            literal_run(d.size);
        } else if (d.type == 1u) {
            if (d.size) {
                uint32_t left = d.size - 1u;
                uint32_t piece = left < kRecML ? left : kRecML;
                put((piece ? 1u : 0u) | kRecFlag, 1u, piece);
                left -= piece;
                while (left) {
                    piece = left < kRecML ? left : kRecML;
                    put(1u | kRecFlag, 0u, piece);
                    left -= piece;
                }
            }
        } else if (!berr) {
            const uint32_t nseq = d.nseq;
            const uint32_t* const my_st = stash + static_cast<uint64_t>(d.stash_at) * 3u;   // entries of three dwords: states | window
            auto load_stash = [&](uint32_t at, uint32_t& st_out, uint64_t& w_out) {
                const bool in = at < nseq;
                const uint32_t* const e = my_st + static_cast<uint64_t>(in ? at : 0u) * 3u;
                st_out = in ? e[0] : 0u;
                w_out = in ? (static_cast<uint64_t>(e[2]) << 32) | e[1] : 0ull;
            };
            // Two batches ahead the stash is loaded, one batch ahead the symbols of its states are gathered from the tables: a
            // batch never waits for global memory.
            auto gather = [&](uint32_t st, uint32_t& lsym_out, uint32_t& msym_out, uint32_t& oc_out) {
                const uint32_t sl = st & 511u, so = (st >> 9) & 255u, sm = st >> 17;
                lsym_out = tg[kTabSymLL + sl];
                msym_out = tg[kTabSymML + sm];
                oc_out = static_cast<uint32_t>(reinterpret_cast<const uint16_t*>(tg + kTabOF)[so]) >> 10;
            };
            uint32_t st_a, st_b;
            uint64_t w_a, w_b;
            uint32_t lsym_a, msym_a, oc_a;
            load_stash(lane, st_a, w_a);
            load_stash(64u + lane, st_b, w_b);
            gather(st_a, lsym_a, msym_a, oc_a);
            for (uint32_t done = 0; done < nseq && !berr; done += 64u) {
                const uint32_t nq = nseq - done < 64u ? nseq - done : 64u;
                const bool valid = lane < nq;
                uint64_t w = w_a;
                const uint32_t lsym = lsym_a, msym = msym_a, oc = oc_a;
                // (for the next batch and the one after it)
                gather(st_b, lsym_a, msym_a, oc_a);
                st_a = st_b;
                w_a = w_b;
                load_stash(done + 128u + lane, st_b, w_b);
                if (PROF) ++n_batches;
                const uint32_t lle = L.ll_base[lsym < 36u ? lsym : 35u], mle = L.ml_base[msym < 53u ? msym : 52u];
                const uint32_t obits = take(w, oc), mbits = take(w, mle >> 24), lbits = take(w, lle >> 24);
                const uint32_t ofv = (1u << oc) + obits;
                const uint32_t mlv = valid ? (mle & 0xFFFFFFu) + mbits : 0u, llv = valid ? (lle & 0xFFFFFFu) + lbits : 0u;
                const uint32_t lincl = wave_scan_add(llv), oincl = wave_scan_add(llv + mlv);
                const uint32_t my_lit = lit_pos + lincl - llv, my_out = out + oincl - llv - mlv;
                const uint32_t lit_all = __builtin_amdgcn_readlane(lincl, 63), out_all = __builtin_amdgcn_readlane(oincl, 63);
                if (__builtin_amdgcn_ballot_w64(valid && oc > 26u))
                    berr = kZstdBadOffset;
                else if (lit_pos + lit_all > d.nlit)
                    berr = kZstdBadLiterals;
                else if (out + out_all > kBlockMax)
                    berr = kZstdBadSize;
                if (berr) break;
                // ---- repeat offsets (RFC 8878 3.1.1.5) by relaxation: (p0, p1, p2) is the history AFTER this lane's sequence,
                // (q0, q1, q2) the one before it = the left neighbour's (lane 0: the block's so far); a slot nobody knows yet has bit
                // 31 set.  A plain offset v makes (v, q0, q1); code 1 / 2 / 3 (+ 1 without literals) picks q0 / q1 / q2 / q0 - 1 and
                // moves it to the front.  Which slot goes where is fixed per lane, so a round is three shifts and five selects.
                const bool isrep = valid && ofv <= 3u;
                const uint32_t idx = ofv - 1u + (llv == 0u ? 1u : 0u);   // (for repeat codes: 0..3)
                const uint32_t kind = !valid ? 5u : (isrep ? idx : 4u);  // 0..3 repeat, 4 plain offset, 5 nothing
                const bool push = kind == 4u, take1 = kind == 1u, take2 = kind == 2u, dec = kind == 3u;
                const bool sel1 = kind >= 1u && kind <= 4u, sel2 = kind >= 2u && kind <= 4u;   // second slot <- q0, third slot <- q1
                const uint32_t v = ofv - 3u;
                uint32_t p0 = push ? v : kUnknown, p1 = kUnknown, p2 = kUnknown;
                uint32_t q0 = 0u, q1 = 0u, q2 = 0u;
                for (uint32_t round = 0; round < 66u; ++round) {
                    if (PROF) ++n_rounds;
                    q0 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(h0), static_cast<int>(p0), 0x138, 0xF, 0xF, false));  // wave_shr:1
                    q1 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(h1), static_cast<int>(p1), 0x138, 0xF, 0xF, false));
                    q2 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(h2), static_cast<int>(p2), 0x138, 0xF, 0xF, false));
                    uint32_t t = take1 ? q1 : (take2 ? q2 : q0);
                    t = (t - (dec ? 1u : 0u)) | (t & kUnknown);
                    const uint32_t n0 = push ? v : t, n1 = sel1 ? q0 : q1, n2 = sel2 ? q1 : q2;
                    const bool changed = n0 != p0 || n1 != p1 || n2 != p2;
                    p0 = n0;
                    p1 = n1;
                    p2 = n2;
                    if (!__builtin_amdgcn_ballot_w64(changed)) break;
                }
                // a repeat code met while the history before it still has an unknown slot stays in the record as it is
                const bool resolved = ((q0 | q1 | q2) & kUnknown) == 0u;
                uint32_t w0 = p0 & 0x3FFFFFFFu;
                if (isrep && !resolved) w0 = kRecRep | (llv == 0u ? kRecFlag : 0u) | ofv;
                if (__builtin_amdgcn_ballot_w64(isrep && resolved && p0 == 0u)) {
                    berr = kZstdBadOffset;
                    break;
                }
                const uint64_t flagged = __builtin_amdgcn_ballot_w64(isrep && !resolved);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(valid && (llv > kRecLL || mlv > kRecML)) == 0ull, 1)) {
                    // one record per sequence
                    const uint32_t ridx = nrec + lane;
                    if (valid && ridx < rec_cap) {
                        if ((ridx & 63u) == 0u) ck[(rec_at + ridx) >> 6] = make_uint4(my_out, my_lit, 0u, 0u);
                        recs[rec_at + ridx] = static_cast<uint64_t>(w0) | (static_cast<uint64_t>(llv | (mlv << 14)) << 32);
                    }
                    if (flagged) n_sym = nrec + 64u - static_cast<uint32_t>(__builtin_clzll(flagged));
                    nrec += nq;
                    out += out_all;
                    lit_pos += lit_all;
                } else {
                    // a run above 16,383 somewhere in the batch: record after record, the long ones in pieces
                    for (uint32_t q = 0; q < nq; ++q) {
                        uint32_t q_ll = __builtin_amdgcn_readlane(llv, q), q_ml = __builtin_amdgcn_readlane(mlv, q);
                        const uint32_t q_w0 = __builtin_amdgcn_readlane(w0, q);
                        while (q_ll > kRecLL) {
                            put(kRecFlag, kRecLL, 0u);
                            q_ll -= kRecLL;
                        }
                        uint32_t piece = q_ml < kRecML ? q_ml : kRecML;
                        put(q_w0, q_ll, piece);
                        q_ml -= piece;
                        while (q_ml) {
                            piece = q_ml < kRecML ? q_ml : kRecML;
                            put((q_w0 & kRecRep) ? (kRecRep | kRecFlag | 0x10u) : (q_w0 | kRecFlag), 0u, piece);
                            q_ml -= piece;
                        }
                        if (q_w0 & kRecRep) n_sym = nrec;
                    }
                }
                // the history after the batch: the last valid lane's
                h0 = __builtin_amdgcn_readlane(p0, nq - 1u);
                h1 = __builtin_amdgcn_readlane(p1, nq - 1u);
                h2 = __builtin_amdgcn_readlane(p2, nq - 1u);
            }
            if (!berr) {
                literal_run(d.nlit - lit_pos);
                if ((h0 | h1 | h2) & kUnknown) n_sym = nrec;
                if (out > kBlockMax || nrec > rec_cap) berr = kZstdBadSize;
            }
        }
        if (berr) {
            if (lane == 0u) atomicMax(&L.err, berr);
        } else if (lane == 0u) {
            ZBlk& o = blk[b];
            o.nrec = nrec;
            o.out_len = out;
            o.n_sym = n_sym;
            o.hist_known = ((h0 | h1 | h2) & kUnknown) ? 0u : 1u;
            o.hist[0] = h0;
            o.hist[1] = h1;
            o.hist[2] = h2;
        }
    }
    __threadfence();
    __syncthreads();
    uint32_t err = L.err;
    // ---- in frame order, one wave: the prefixes written with an unknown history replayed; where each block's output starts
    if (wave == 0u && !err) {
        uint32_t hist0 = 1u, hist1 = 4u, hist2 = 8u, out_top = 0, rerr = 0, nrec_all = 0;
        for (uint32_t b = 0; b < nblk; ++b) {
            ZBlk& d = blk[b];
            const uint32_t n_sym = __hip_atomic_load(&d.n_sym, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t* const rj = recs + d.rec_at;
            for (uint32_t k0 = 0; k0 < n_sym; k0 += 64u) {
                // 64 records at a time into the lanes, walked by readlane (uniform); the stores by the lane that holds the record
                const uint32_t k = k0 + lane;
                const uint64_t w = k < n_sym ? __hip_atomic_load(&rj[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                const uint32_t w0v = static_cast<uint32_t>(w);
                uint32_t fixed = w0v;
                const uint32_t cnt = n_sym - k0 < 64u ? n_sym - k0 : 64u;
                uint64_t todo = __builtin_amdgcn_ballot_w64(k < n_sym && ((w0v & kRecRep) || !(w0v & kRecFlag)));
                (void)cnt;
                while (todo) {
                    const uint32_t q = static_cast<uint32_t>(__builtin_ctzll(todo));
                    todo &= todo - 1ull;
                    const uint32_t w0 = __builtin_amdgcn_readlane(w0v, q);
                    if (w0 & kRecRep) {
                        const uint32_t code = w0 & 0xFFu;
                        uint32_t off;
                        if (code == 0x10u) {
                            off = hist0;
                        } else {
                            const uint32_t idx = code - 1u + ((w0 & kRecFlag) ? 1u : 0u);
                            if (idx == 0u) {
                                off = hist0;
                            } else if (idx == 1u) {
                                off = hist1;
                                hist1 = hist0;
                                hist0 = off;
                            } else if (idx == 2u) {
                                off = hist2;
                                hist2 = hist1;
                                hist1 = hist0;
                                hist0 = off;
                            } else {
                                off = hist0 - 1u;
                                hist2 = hist1;
                                hist1 = hist0;
                                hist0 = off;
                            }
                            if (off == 0u) rerr = kZstdBadOffset;
                        }
                        fixed = lane == q ? (off & 0x3FFFFFFFu) : fixed;
                    } else {
                        hist2 = hist1;
                        hist1 = hist0;
                        hist0 = w0;
                    }
                }
                if (k < n_sym && (w0v & kRecRep)) rj[k] = (w & 0xFFFFFFFF00000000ull) | fixed;
            }
            if (__hip_atomic_load(&d.hist_known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                hist0 = __hip_atomic_load(&d.hist[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hist1 = __hip_atomic_load(&d.hist[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hist2 = __hip_atomic_load(&d.hist[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0u) d.out_at = out_top;
            out_top += __hip_atomic_load(&d.out_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nrec_all += __hip_atomic_load(&d.nrec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!rerr && out_top != dst_len) rerr = kZstdBadSize;
        if (lane == 0u) {
            if (rerr) atomicMax(&L.err, rerr);
            hdr->nrec = nrec_all;
            hdr->out_len = out_top;
            if (!rerr) atomicAdd(&tally[0], static_cast<unsigned long long>(nrec_all));
        }
    }
    __threadfence();
    __syncthreads();
    err = L.err;
    // ---- checkpoints: positions in the frame, records valid; the slots a block did not use point at its end
    if (!err) {
        for (uint32_t b = wave; b < nblk; b += kZvWaves) {
            const ZBlk& d = blk[b];
            const uint32_t out_at = __hip_atomic_load(&d.out_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t nrec = d.nrec, out_len = d.out_len;
            const uint32_t first_slot = d.rec_at >> 6, nslot = d.rec_cap >> 6;
            for (uint32_t q = lane; q < nslot; q += 64u) {
                const uint32_t first = q * 64u;
                uint4 c;
                if (first < nrec) {
                    c = ck[first_slot + q];
                    c.x += out_at;
                    c.y += d.lit_at;
                    c.z = nrec - first < 64u ? nrec - first : 64u;
                } else {
                    c = make_uint4(out_at + out_len, d.lit_at + d.nlit, 0u, 0u);
                }
                ck[first_slot + q] = c;
            }
        }
    }
    if (threadIdx.x == 0u) {
        if (err) status[fi] = err;
        if (PROF) {
            atomicAdd(&tally[23], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        }
    }
    if (PROF && lane == 0u) {
        atomicAdd(&tally[24], n_rounds);
        atomicAdd(&tally[25], n_batches);
    }
}

// ------------------------------------------------------------------------------------------------ zstd_execute
constexpr uint32_t kZxEmit = kZstdEmitters, kZxScan = kZstdScanners;   // (six emitters: a batch with far matches waits a microsecond or two for the global loads)
constexpr uint32_t kZxThreads = 64u * (kZxEmit + kZxScan + 1u);
constexpr uint32_t kZxWindow = kZstdWindow;   // (flagstat_zstd_kernels.h)
constexpr uint32_t kZxNear = kZxWindow - 1u;   // matches up to this far back read the ring; farther ones the flushed output

struct __attribute__((aligned(16))) ZxLds {
    static constexpr uint32_t kNR = kZxWindow + 4096u, kMR = 2048u, kK = 512u, kChunk = 256u, kFlush = 1024u, kScan = kZxScan;
    static constexpr bool kPublishFlush = true;
    static constexpr uint32_t kAhead = kNR - kZxWindow - kChunk;   // the emitters' position may lead the copier's by this much
    static constexpr uint32_t kSpan = 1536u;                    // most output bytes an emitter writes before publishing
    uint8_t ring[kNR];
    uint32_t mark[kMR];
    uint32_t fsrc[kK];
    uint32_t e_pos[8];                   // per emitter: where its first unfinished batch starts (unused slots: the end; two 16-byte reads)
    uint32_t f_op, pad_[3];              // copier: output flushed AND landed
    uint32_t s_clr[7], d_op;             // per scanner: start of the next chunk it will clear (unused slots: ~0); copier: end of the last chunk copied (two 16-byte reads)
    uint32_t s_done[kZxScan];
    uint32_t c_ready, s_carry[8], err;
};
static_assert(ZxLds::kSpan + ZxLds::kChunk <= ZxLds::kMR && ZxLds::kSpan + ZxLds::kChunk + ZxLds::kK <= ZxLds::kAhead + ZxLds::kChunk, "no cyclic wait");
static_assert(sizeof(ZxLds) <= (kZxWindow == 32768u ? 54000u : 81920u) && kZxEmit <= 8 && kZxScan <= 7, "three (two) workgroups per CU");
#ifdef FLAGSTAT_ZSTD_SHIPPED_SPLIT
static_assert(kZxThreads == 512, "two waves per SIMD: the second workgroup of a CU finds room on every SIMD");
#endif

// ---- emitters: records -> markers, literal bytes.  Emitter `which` takes the batches (64 records, one checkpoint)
// which, which + kZxEmit, ...; a checkpoint carries the batch's output and literal positions, so the emitters do not depend
// on each other.  Records and the first 256 literal bytes of a batch are loaded one batch ahead, checkpoints two.
// scope of the far-match loads.  The copier that wrote the bytes is a wave of the same workgroup and publishes "flushed and landed"
// behind an s_waitcnt vmcnt(0), so workgroup scope ("sc0") would do; measured the same to the cycle (3.64 against 3.65 M cycles a
// frame: the wait is for the L2 either way), so the wider scope, which has the fuzzers' mileage, stays
#ifndef FLAGSTAT_ZSTD_FAR_SCOPE
#define FLAGSTAT_ZSTD_FAR_SCOPE "sc0 sc1"
#endif
template <bool PROF>
__device__ void zx_emit(ZxLds& L, const uint4* __restrict__ ck, const uint64_t* __restrict__ recs, const uint8_t* __restrict__ lits,
                        const uint32_t lit_cap, const uint8_t* __restrict__ dst, const uint32_t nslots, const uint32_t oend, const uint32_t lane,
                        const uint32_t which, unsigned long long* __restrict__ tally)
{
    uint32_t err = 0, s_seen = 0, d_seen = 0, f_seen = 0, nfar = 0;
    unsigned long long t_wait = 0, n_groups = 0, n_batches = 0, n_long = 0, n_far_rounds = 0, n_short = 0, n_far_groups = 0, t_pre = 0, t_mark = 0, t_long = 0, t_far = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    auto room = [&](uint32_t at, uint32_t nbytes) -> bool {
        if (__builtin_expect(at + nbytes <= s_seen + ZxLds::kMR && at + nbytes <= d_seen + ZxLds::kAhead, 1)) return true;
        return wg_wait_timed<PROF>(L, t_wait, [&] {
            const uint4 t = wg_ld4(L.s_clr), u = wg_ld4(L.s_clr + 4);
            s_seen = umin(umin(umin3(t.x, t.y, t.z), t.w), umin3(u.x, u.y, u.z));
            d_seen = u.w;
            return at + nbytes <= s_seen + ZxLds::kMR && at + nbytes <= d_seen + ZxLds::kAhead;
        });
    };
    auto ring_at = [](uint32_t pos) -> uint32_t { return pos % ZxLds::kNR; };
    const uint4 ck_end = make_uint4(oend, 0u, 0u, 0u);
    auto load_ck = [&](uint32_t g) -> uint4 {
        const uint4 c = g < nslots ? ck[g] : ck_end;
        return make_uint4(__builtin_amdgcn_readfirstlane(c.x), __builtin_amdgcn_readfirstlane(c.y), __builtin_amdgcn_readfirstlane(c.z), 0u);
    };
    auto load_lit = [&](const uint4& c) -> uint32_t {
        // the 256 literal bytes from the 4-byte boundary at or below the batch's literal position
        const uint32_t at = (c.y & ~3u) + 4u * lane;
        return c.z && at + 4u <= lit_cap ? *reinterpret_cast<const uint32_t*>(lits + at) : 0u;
    };
    uint4 c0 = load_ck(which), c1 = load_ck(which + kZxEmit);
    uint64_t rec0 = which < nslots && lane < c0.z ? recs[which * 64u + lane] : 0ull;
    uint32_t lit0 = load_lit(c0);
    wg_st(&L.e_pos[which], c0.x);
    for (uint32_t g = which; g < nslots; g += kZxEmit) {
        // loads for the next two batches first
        const uint4 c2 = load_ck(g + 2u * kZxEmit);
        // (the next batch's records and literal bytes are asked for behind this batch's own arithmetic, just before its stores and
        // far loads: issued at the top, the compiler made the batch wait for them there -- a memory latency a batch)
        uint64_t rec1 = 0ull;
        uint32_t lit1 = 0u;
        auto prefetch = [&] {
            rec1 = g + kZxEmit < nslots && lane < c1.z ? recs[(g + kZxEmit) * 64u + lane] : 0ull;
            lit1 = load_lit(c1);
        };
        const uint32_t nvalid = c0.z;
        if (nvalid) {
            ++n_batches;
            const unsigned long long tb0 = PROF ? __builtin_readcyclecounter() : 0ull;
            const bool valid = lane < nvalid;
            const uint32_t w0 = static_cast<uint32_t>(rec0), w1 = static_cast<uint32_t>(rec0 >> 32);
            const uint32_t ll = valid ? w1 & 16383u : 0u, ml = valid ? (w1 >> 14) & 16383u : 0u, off = w0 & 0x3FFFFFFFu;
            const uint32_t len = ll + ml;
            const uint32_t incl = wave_scan_add(len), lincl = wave_scan_add(ll);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63), ltotal = __builtin_amdgcn_readlane(lincl, 63);
            const uint32_t rel = incl - len;
            const uint32_t op = c0.x + rel, mpos = op + ll, lp = c0.y + lincl - ll;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(valid & (ml > 0u) & ((off == 0u) | (off > mpos))) != 0ull, 0)) {
                err = kZstdBadOffset;
                break;
            }
            if (__builtin_expect(c0.x > oend || total > oend - c0.x || c0.y + ltotal > lit_cap || __builtin_amdgcn_ballot_w64(valid & ((w0 & kRecRep) != 0u)) != 0ull, 0)) {
                err = kZstdBadSize;
                break;
            }
            const bool far = valid & (ml > 0u) & (off > kZxNear);
            // literal bytes of short runs come out of the 256 prefetched ones: byte a of them is this run's first
            const uint32_t a = (c0.y & 3u) + (lincl - ll);
            const uint32_t d_lo = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>((a >> 2) << 2), static_cast<int>(lit0)));
            const uint32_t d_hi = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(((a >> 2) + 1u) << 2), static_cast<int>(lit0)));
            const uint32_t lit4 = __builtin_amdgcn_alignbyte(d_hi, d_lo, a & 3u);
            const uint32_t ri = ring_at(op);
            // short: 1..4 literal bytes in reach of the prefetch, a match behind them that takes what a 4-byte store writes too
            // much, not across the end of the ring
            const bool shortlit = valid & (ll >= 1u) & (ll <= 4u) & (ll + ml >= 4u) & (a + 4u <= 252u) & (ri + 4u <= ZxLds::kNR);
            const bool longlit = valid & (ll >= 1u) & !shortlit;
            prefetch();
            uint32_t lo = 0;  // records [0, lo) of the batch are done
            bool failed = false;
            if (PROF) t_pre += __builtin_readcyclecounter() - tb0;
            while (lo < nvalid) {
                ++n_groups;
                // the records [lo, hi) whose output fits one publication
                const uint32_t base_rel = __builtin_amdgcn_readlane(rel, lo);
                const bool ingroup = valid & (lane >= lo) & (incl - base_rel <= ZxLds::kSpan);
                const uint32_t cnt = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(ingroup)));
                if (cnt == 0u) {
                    // ONE record longer than a publication (a long literal run, a long far match): in pieces, the whole wave
                    const uint32_t r_ll = __builtin_amdgcn_readlane(ll, lo), r_ml = __builtin_amdgcn_readlane(ml, lo);
                    const uint32_t r_op = __builtin_amdgcn_readlane(op, lo), r_lp = __builtin_amdgcn_readlane(lp, lo);
                    const uint32_t r_off = __builtin_amdgcn_readlane(off, lo);
                    const bool r_far = r_ml > 0u && r_off > kZxNear;
                    const uint32_t nlit_bytes = r_ll + (r_far ? r_ml : 0u);   // bytes this record puts into the ring itself
                    if (r_far) ++nfar;
                    for (uint32_t done = 0; done < nlit_bytes || done == 0u;) {
                        const uint32_t piece = nlit_bytes - done < 1024u ? nlit_bytes - done : 1024u;
                        // A piece with match bytes waits for ITS source only (one past its last source byte flushed and landed).
                        // Waiting for the whole match's source before the first piece can never end: behind 16 KiB of literals
                        // a 16 KiB match from 32 KiB back ends a few bytes before this record starts, and the copier cannot pass
                        // the record's start until the pieces before have been published (found by the fuzz when the ring went
                        // from 64 to 32 KiB; with 64 KiB the source always lay 32 KiB behind the record).  A piece's own source
                        // lies > 31 KiB behind it, the copier at most 4 KiB + two flush units.
                        if (r_far && done + piece > r_ll) {
                            const uint32_t need = r_op + done + piece - r_off;
                            if (need > f_seen && !wg_wait_timed<PROF>(L, t_wait, [&] {
                                    f_seen = wg_ld(&L.f_op);
                                    return f_seen >= need;
                                })) {
                                failed = true;
                                break;
                            }
                        }
                        if (!room(r_op + done, piece + 1u)) {
                            failed = true;
                            break;
                        }
                        // (markers where room has just been granted: "literal" with the first piece, the match with the last)
                        if (lane == 0u) {
                            if (done == 0u && nlit_bytes) L.mark[r_op & (ZxLds::kMR - 1u)] = kMarkLiteral;
                            if (done + piece == nlit_bytes && r_ml && !r_far) L.mark[(r_op + r_ll) & (ZxLds::kMR - 1u)] = r_off;
                        }
                        for (uint32_t b = done + lane; b < done + piece; b += 64u) {
                            uint8_t v;
                            if (b < r_ll)
                                v = lits[r_lp + b];
                            else
                                v = __hip_atomic_load(&dst[r_op + b - r_off], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            L.ring[ring_at(r_op + b)] = v;
                        }
                        done += piece;
                        if (done < nlit_bytes) wg_st(&L.e_pos[which], r_op + done);
                        if (piece == 0u) break;
                    }
                    if (failed) break;
                    lo += 1u;
                    wg_st(&L.e_pos[which], lo < nvalid ? __builtin_amdgcn_readlane(op, lo) : c1.x);
                    continue;
                }
                const uint32_t hi = lo + cnt;
                const uint32_t span_end = __builtin_amdgcn_readlane(incl, hi - 1u);
                if (!room(c0.x + base_rel, span_end - base_rel)) {
                    failed = true;
                    break;
                }
                const unsigned long long tm0 = PROF ? __builtin_readcyclecounter() : 0ull;
                const bool now = valid & (lane >= lo) & (lane < hi);
                // markers: a near match at its first byte, "literal" where this record's own bytes start
                if (now & (ml > 0u) & !far) L.mark[mpos & (ZxLds::kMR - 1u)] = off;
                if (now & ((ll > 0u) | far)) L.mark[(ll > 0u ? op : mpos) & (ZxLds::kMR - 1u)] = kMarkLiteral;
                if (now & shortlit) __builtin_memcpy(&L.ring[ri], &lit4, 4);
                // longer literal runs: the whole wave, run after run
                uint64_t lm = __builtin_amdgcn_ballot_w64(now & longlit);
                const unsigned long long tm1 = PROF ? __builtin_readcyclecounter() : 0ull;
                if (PROF) t_mark += tm1 - tm0;
                if (PROF) n_short += static_cast<unsigned long long>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(now & shortlit)));
                while (lm) {
                    const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(lm));
                    lm &= lm - 1ull;
                    const uint32_t r_ll = __builtin_amdgcn_readlane(ll, j), r_op = __builtin_amdgcn_readlane(op, j), r_lp = __builtin_amdgcn_readlane(lp, j);
                    for (uint32_t b = lane; b < r_ll; b += 64u) L.ring[ring_at(r_op + b)] = lits[r_lp + b];
                    if (PROF) ++n_long;
                }
                const unsigned long long tm2 = PROF ? __builtin_readcyclecounter() : 0ull;
                if (PROF) t_long += tm2 - tm1;
                // far matches: every lane its own, read from the flushed output (device-scope loads, past the L1)
                const uint64_t fm = __builtin_amdgcn_ballot_w64(now & far);
                if (fm) {
                    if (PROF) ++n_far_groups;
                    nfar += static_cast<uint32_t>(__builtin_popcountll(fm));
                    const uint32_t src = mpos - off;
                    uint32_t src_end = (now & far) ? src + ml : 0u;
                    src_end = wave_scan_max(src_end);
                    src_end = __builtin_amdgcn_readlane(src_end, 63);
                    // (the copier's "flushed and landed" mark, as last seen: a far source lies 32 KiB back, the mark 2 KiB)
                    if (__builtin_expect(src_end > f_seen, 0)) {
                        if (!wg_wait_timed<PROF>(L, t_wait, [&] {
                                f_seen = wg_ld(&L.f_op);
                                return f_seen >= src_end;
                            })) {
                            failed = true;
                            break;
                        }
                    }
                    const bool fl = now & far;
                    const uint8_t* sp = dst + (fl ? src : 0u);
                    const uint32_t rm = ring_at(mpos);
                    // Sixteen bytes a round, every far lane its own match, straight-line for the first round (most matches are
                    // shorter): four dwords at the match's own (byte) address, past the L1 (what they hold behind the match is
                    // inside the output buffer and not used), stored whole where the match has them and the ring does not wrap,
                    // as bytes otherwise.
                    for (uint32_t b = 0;; b += 16u) {
                        if (PROF) ++n_far_rounds;
                        uint32_t u0, u1, u2, u3;
                        asm volatile("global_load_dword %0, %4, off " FLAGSTAT_ZSTD_FAR_SCOPE "\n\t"
                                     "global_load_dword %1, %4, off offset:4 " FLAGSTAT_ZSTD_FAR_SCOPE "\n\t"
                                     "global_load_dword %2, %4, off offset:8 " FLAGSTAT_ZSTD_FAR_SCOPE "\n\t"
                                     "global_load_dword %3, %4, off offset:12 " FLAGSTAT_ZSTD_FAR_SCOPE "\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
                                     : "v"(sp)
                                     : "memory");
                        const bool live = fl & (b < ml);
                        const bool whole = rm + b + 16u <= ZxLds::kNR;   // (the sixteen bytes of this round lie in the ring without a wrap)
                        const uint32_t left = live ? ml - b : 0u;        // bytes of the match still to be stored
                        const uint32_t uu[4] = {u0, u1, u2, u3};
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(live & !whole) != 0ull, 0)) {
                            // across the ring's end (one round in 2,300): byte by byte
                            if (live & !whole) {
#pragma unroll
                                for (uint32_t x = 0; x < 16u; ++x)
                                    if (x < left) L.ring[ring_at(mpos + b + x)] = static_cast<uint8_t>(uu[x >> 2] >> (8u * (x & 3u)));
                            }
                        }
                        // whole dwords where the match has them, then what is left of it below a dword: two bytes and / or one.
                        // (Byte stores only where they must be: the byte behind the match is the next record's, maybe another emitter's.)
                        const bool lw = live & whole;
                        uint8_t* const at = &L.ring[rm + b];
#pragma unroll
                        for (uint32_t k = 0; k < 4u; ++k)
                            if (lw & (left >= 4u * k + 4u)) __builtin_memcpy(at + 4u * k, &uu[k], 4);
                        const uint32_t tk = left >> 2;   // the dword the tail lies in (< 4 when there is one)
                        const uint32_t ut = tk == 0u ? u0 : (tk == 1u ? u1 : (tk == 2u ? u2 : u3));
                        const bool tail = lw & (left < 16u);
                        if (tail & ((left & 2u) != 0u)) {
                            const uint16_t h = static_cast<uint16_t>(ut);
                            __builtin_memcpy(at + (left & ~3u), &h, 2);
                        }
                        if (tail & ((left & 1u) != 0u)) at[(left & ~3u) + (left & 2u)] = static_cast<uint8_t>(ut >> (8u * (left & 2u)));
                        if (!__builtin_amdgcn_ballot_w64(fl & (b + 16u < ml))) break;
                        sp += (fl & (b + 16u < ml)) ? 16 : 0;   // (a lane that is done stays where it is: inside the buffer)
                    }
                }
                if (PROF) t_far += __builtin_readcyclecounter() - tm2;
                lo = hi;
                wg_st(&L.e_pos[which], lo < nvalid ? c0.x + span_end : c1.x);
            }
            if (failed) break;
        } else {
            prefetch();
            wg_st(&L.e_pos[which], c1.x);
        }
        c0 = c1;
        c1 = c2;
        rec0 = rec1;
        lit0 = lit1;
    }
    if (err) wg_st(&L.err, err);
    if (!err && !wg_ld(&L.err)) wg_st(&L.e_pos[which], oend);
    if (lane == 0u) {
        atomicAdd(&tally[1], static_cast<unsigned long long>(nfar));
        if (PROF) {
            atomicAdd(&tally[15], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
            atomicAdd(&tally[16], t_wait);
            atomicAdd(&tally[5], n_batches);
            atomicAdd(&tally[6], n_groups);
            atomicAdd(&tally[2], n_long);
            atomicAdd(&tally[3], t_long);
            atomicAdd(&tally[4], n_far_rounds);
            atomicAdd(&tally[7], n_short);
            atomicAdd(&tally[28], n_far_groups);
            atomicAdd(&tally[29], t_far);
            atomicAdd(&tally[30], t_pre);
            atomicAdd(&tally[31], t_mark);
        }
    }
}

template <bool PROF>
__global__ __launch_bounds__(kZxThreads, FLAGSTAT_ZSTD_EXEC_WAVES_PER_SIMD) void zstd_execute(const GpuBlock* __restrict__ blocks, const uint8_t* __restrict__ scratch, const ZLayout lay,
                                                              uint8_t* __restrict__ out, uint32_t* __restrict__ status, unsigned long long* __restrict__ tally)
{
    __shared__ ZxLds L;
    const uint32_t fi = blockIdx.x;
    if (status[fi] != 0u) return;  // (the entropy stage failed this frame: its status stands; uniform for the workgroup)
    const GpuBlock b = blocks[fi];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const ZFrameHdr hdr = reinterpret_cast<const ZFrameHdr*>(scratch + lay.hdr_at)[fi];
    const uint4* const ck = reinterpret_cast<const uint4*>(scratch + lay.ck_at) + static_cast<uint64_t>(fi) * lay.ck_stride;
    const uint64_t* const recs = reinterpret_cast<const uint64_t*>(scratch + lay.rec_at) + static_cast<uint64_t>(fi) * lay.rec_stride;
    const uint8_t* const lits = scratch + lay.lit_at + static_cast<uint64_t>(fi) * lay.lit_stride;
    const uint32_t nslots = hdr.nslots <= lay.ck_stride ? hdr.nslots : 0u;
    for (uint32_t i = threadIdx.x; i < ZxLds::kMR; i += kZxThreads) L.mark[i] = 0u;
    if (threadIdx.x == 0u) {
        for (uint32_t i = 0; i < 8u; ++i) L.e_pos[i] = i < kZxEmit ? 0u : ~0u;
        L.f_op = 0u;
        for (uint32_t i = 0; i < 7u; ++i) L.s_clr[i] = i < kZxScan ? i * ZxLds::kChunk : ~0u;
        for (uint32_t i = 0; i < kZxScan; ++i) L.s_done[i] = 0u;
        L.d_op = 0u;
        L.c_ready = 0u;
        L.err = 0u;
    }
    __syncthreads();
    if (role < kZxEmit)
        zx_emit<PROF>(L, ck, recs, lits, lay.lit_stride, out + b.dst_off, nslots, b.dst_len, lane, role, tally);
    else if (role < kZxEmit + kZxScan) {
        // the position below which every marker and literal byte is in place: the first batch some emitter has not finished
        auto frontier = []() -> uint32_t {
            const uint4 a = wg_ld4(L.e_pos), c = wg_ld4(L.e_pos + 4);
            return umin(umin(umin3(a.x, a.y, a.z), a.w), umin(umin3(c.x, c.y, c.z), c.w));
        };
        wgpipe_scan<PROF>(L, b.dst_len, lane, role - kZxEmit, frontier, tally);
    } else
        wgpipe_copy<PROF>(L, out + b.dst_off, b.dst_len, lane, tally);
    __syncthreads();
    if (threadIdx.x == 0u) status[fi] = L.err == 9u ? static_cast<uint32_t>(kZstdStuck) : L.err;
}

}  // namespace fsk

extern "C" uint64_t fsk_zstd_scratch_bytes(uint32_t max_dst_len, uint32_t nframes) { return fsk::zstd_layout(max_dst_len, nframes).total; }
extern "C" uint64_t fsk_zstd_scratch_bytes_ex(uint32_t max_dst_len, uint32_t nframes, uint32_t min_blocks)
{
    return fsk::zstd_layout(max_dst_len, nframes, min_blocks).total;
}

extern "C" hipError_t fsk_zstd_decode(const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out, uint32_t* status,
                                      unsigned long long* tally, void* scratch, uint64_t scratch_bytes, uint32_t max_dst_len, int prof, hipStream_t stream)
{
    return fsk_zstd_decode_ex(comp, blocks, nblocks, out, status, tally, scratch, scratch_bytes, max_dst_len, 0u, prof, stream);
}

extern "C" hipError_t fsk_zstd_decode_ex(const uint8_t* comp, const fsk::GpuBlock* blocks, uint32_t nblocks, uint8_t* out, uint32_t* status,
                                         unsigned long long* tally, void* scratch, uint64_t scratch_bytes, uint32_t max_dst_len, uint32_t min_blocks, int prof,
                                         hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (!comp || !blocks || !out || !status || !tally || !scratch) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(comp) & 7u) || (reinterpret_cast<uintptr_t>(scratch) & 255u)) return hipErrorInvalidValue;
    const fsk::ZLayout lay = fsk::zstd_layout(max_dst_len, nblocks, min_blocks);
    if (lay.total > scratch_bytes) return hipErrorInvalidValue;
    uint8_t* const sc = static_cast<uint8_t*>(scratch);
    const dim3 grid(nblocks);
    const uint32_t npairs = (nblocks + fsk::kZcFrames - 1u) / fsk::kZcFrames;
    const dim3 pairs((npairs + fsk::kZcWaves - 1u) / fsk::kZcWaves);
    if (prof) {
        hipLaunchKernelGGL((fsk::zstd_prepare<true>), grid, dim3(64), 0, stream, comp, blocks, sc, lay, status, tally);
        hipLaunchKernelGGL((fsk::zstd_chain<true>), pairs, dim3(64 * fsk::kZcWaves), 0, stream, comp, blocks, sc, lay, status, tally);
        hipLaunchKernelGGL((fsk::zstd_records<true>), grid, dim3(64 * fsk::kZvWaves), 0, stream, blocks, sc, lay, status, tally);
        hipLaunchKernelGGL((fsk::zstd_execute<true>), grid, dim3(fsk::kZxThreads), 0, stream, blocks, sc, lay, out, status, tally);
    } else {
        hipLaunchKernelGGL((fsk::zstd_prepare<false>), grid, dim3(64), 0, stream, comp, blocks, sc, lay, status, tally);
        hipLaunchKernelGGL((fsk::zstd_chain<false>), pairs, dim3(64 * fsk::kZcWaves), 0, stream, comp, blocks, sc, lay, status, tally);
        hipLaunchKernelGGL((fsk::zstd_records<false>), grid, dim3(64 * fsk::kZvWaves), 0, stream, blocks, sc, lay, status, tally);
        hipLaunchKernelGGL((fsk::zstd_execute<false>), grid, dim3(fsk::kZxThreads), 0, stream, blocks, sc, lay, out, status, tally);
    }
    return hipGetLastError();
}

extern "C" void fsk_zstd_role_waves(int* emitters, int* scanners)
{
    *emitters = fsk::kZstdEmitters;
    *scanners = fsk::kZstdScanners;
}

extern "C" int fsk_zstd_frames_per_cu(void)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fsk::zstd_execute<false>, fsk::kZxThreads, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
