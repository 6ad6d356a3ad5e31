// lz4_block_decode.h -- the host LZ4 *block* decoder of the block-file pipeline (SURVEY.md section 8
// row f1; the reference calls liblz4's LZ4_decompress_safe, benchmark/flagstats.cpp:316).  Written from
// the published block format (token = 4 bits literal length | 4 bits match length - 4, 255-continued
// length bytes, 2-byte little-endian offset, last sequence literals-only); checked in tests/ against the
// image's liblz4.so.1 acting as oracle.  Header-only so that tests/perf/lz4_decode_bench.cpp and
// tests/lz4_fuzz_asan.cpp exercise exactly the code the library runs.
//
// Shaped for FLAG streams, not for text: a 1,024,000-byte block of NA12878-like flags is ~156,000
// sequences of 6.6 output bytes; with LZ4-fast 95 % of them are "no literals, match of 4..18 bytes" =
// exactly 3 input bytes (token, 2-byte offset).  A byte-at-a-time parser is bound by the token ->
// next-token address chain (one L1 load latency per sequence; liblz4 sits there too).  Loop A below
// loads 8 input bytes at once and retires up to TWO such sequences per load: under correctly predicted
// branches the next load address (ip + 6) does not depend on the loaded bytes, so loads run ahead of
// the parsing.  Mixed streams (LZ4-HC: a literal in 1/4 .. 1/2 of the sequences) would mispredict their
// way through loop A and are handed to loop B, which treats every short sequence alike.
#ifndef FLAGSTAT_LZ4_BLOCK_DECODE_H_
#define FLAGSTAT_LZ4_BLOCK_DECODE_H_

#include <stddef.h>
#include <stdint.h>

#include <cstring>

namespace fslz4 {

namespace lz4d {

inline uint64_t ld64(const uint8_t* p)
{
    uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
}
inline size_t ld16(const uint8_t* p)
{
    uint16_t v;
    std::memcpy(&v, p, 2);  // little-endian host (x86-64)
    return v;
}
inline void cp8(uint8_t* d, const uint8_t* s)
{
    uint64_t v;
    std::memcpy(&v, s, 8);
    std::memcpy(d, &v, 8);
}
inline void cp2(uint8_t* d, const uint8_t* s)
{
    uint16_t v;
    std::memcpy(&v, s, 2);
    std::memcpy(d, &v, 2);
}
inline void cp16(uint8_t* d, const uint8_t* s)
{
    struct V16 { uint64_t a, b; } v;   // one 16-byte (SSE) load and store
    std::memcpy(&v, s, 16);
    std::memcpy(d, &v, 16);
}
// match of ml <= 18 bytes at distance off >= 1; may write up to 24 bytes.  off >= 16 (97 % of a FLAG
// stream's matches): one 16-byte load/store (+ 8 more for ml 17, 18) -- few, wide stores matter: a
// later match often reads bytes still in the store buffer, and a load that straddles two pending
// stores cannot be forwarded.  8 <= off < 16: 8-byte pieces in order reproduce LZ4's overlapping-copy
// semantics.  off 2 / 4 (a run of one flag / one flag pair): the period is broadcast.  Other small
// offsets: byte by byte.
// WIDE = false (loop B, mixed streams whose offsets straddle 16 unpredictably): no off >= 16 case.
template <bool WIDE>
inline void short_match(uint8_t* op, size_t off, size_t ml)
{
    const uint8_t* m = op - off;
    if (WIDE && off >= 16) {
        cp16(op, m);
        if (ml > 16) cp2(op + 16, m + 16);
    } else if (off >= 8) {
        cp8(op, m);
        cp8(op + 8, m + 8);
#ifdef FSLZ4_TAIL8
        cp8(op + 16, m + 16);
#else
        cp2(op + 16, m + 16);   // bytes 16, 17 only: every surplus byte stored is one more pending store a later match may straddle
#endif
    } else if (off == 2 || off == 4) {
        uint64_t pat;
        if (off == 2) {
            uint16_t h;
            std::memcpy(&h, m, 2);
            pat = 0x0001000100010001ull * h;
        } else {
            uint32_t q;
            std::memcpy(&q, m, 4);
            pat = 0x0000000100000001ull * q;
        }
        std::memcpy(op, &pat, 8);
        std::memcpy(op + 8, &pat, 8);
        std::memcpy(op + 16, &pat, 8);
    } else {
        for (size_t i = 0; i < ml; ++i) op[i] = m[i];
    }
}

#ifdef FSLZ4_B_WIDE   // measurement switch: loop B with the 16-byte match copy too
constexpr bool kLoopBWide = true;
#else
constexpr bool kLoopBWide = false;
#endif

struct State {
    const uint8_t* ip;
    uint8_t* op;
    bool pairs;           // which fast loop suits the stream; re-judged as it goes
    unsigned slow;        // loop A: visits of its literal path since `mark`
    const uint8_t* mark;
    unsigned seqs, bare;  // loop B: sequences seen / of which bare matches
};

// Runs the fast loops from s.ip / s.op until one sequence needs the general path (long literal run or
// match, offset 0, offset beyond the output start, or the end regions) or -- CHECKED only -- until
// `ostop` has been produced.  CHECKED: every offset is compared with the output produced so far;
// !CHECKED is used once 64 KiB exist, where every 16-bit offset is valid.
// Bounds: an iteration reads < 20 input bytes (callers keep >= 32 readable: ip < ifast, which also rules
// out the literals-only last sequence, whose > 14 literals would have to run to the end) and writes < 60
// output bytes (callers keep >= 96 writable: op < ofast).
template <bool CHECKED>
inline void fast_loops(State& s, const uint8_t* const ifast, uint8_t* const ofast, const uint8_t* const dst,
                       const uint8_t* const ostop, const bool may_pair)
{
    const uint8_t* ip = s.ip;
    uint8_t* op = s.op;
    for (;;) {
        if (s.pairs) {
            bool swap = false;
            unsigned slow = s.slow;
            const uint8_t* mark = s.mark;
            while (ip < ifast && op < ofast && (!CHECKED || op < ostop)) {
                const uint64_t w = ld64(ip);
                const unsigned tok = static_cast<unsigned>(w & 0xFFu);
                if (tok < 0x0Fu) {  // no literals, match nibble 0..14: the sequence is 3 bytes of w
                    const size_t off = static_cast<size_t>((w >> 8) & 0xFFFFu);
                    if (off == 0 || (CHECKED && off > static_cast<size_t>(op - dst))) break;
                    short_match<true>(op, off, tok + 4);
                    op += tok + 4;
                    ip += 3;
                    const unsigned tok2 = static_cast<unsigned>((w >> 24) & 0xFFu);  // the next token is in w too
                    if (tok2 < 0x0Fu) {
                        const size_t off2 = static_cast<size_t>((w >> 32) & 0xFFFFu);
                        if (off2 == 0 || (CHECKED && off2 > static_cast<size_t>(op - dst))) break;
                        short_match<true>(op, off2, tok2 + 4);
                        op += tok2 + 4;
                        ip += 3;
                    }
                    continue;
                }
                // a short sequence with literals.  64 visits here within ~8 sequences each: mixed stream
                if (++slow == 64) {
                    if (ip - mark < 64 * 24) {
                        swap = true;
                        break;
                    }
                    slow = 0;
                    mark = ip;
                }
                const size_t lit = tok >> 4;
                const unsigned mln = tok & 15u;
                if (lit == 15 || mln == 15) break;  // length continuation bytes
                const size_t off = ld16(ip + 1 + lit);
                if (off == 0 || (CHECKED && off > static_cast<size_t>(op - dst) + lit)) break;
                cp16(op, ip + 1);  // <= 14 literals: one fixed 16-byte copy, the surplus is overwritten next
                op += lit;
                ip += 3 + lit;
                short_match<true>(op, off, mln + 4);
                op += mln + 4;
            }
            s.slow = slow;
            s.mark = mark;
            if (!swap) goto out;
            s.pairs = false;
            s.seqs = s.bare = 0;
        } else {
            bool swap = false;
            unsigned seqs = s.seqs, bare = s.bare;  // locals: byte stores alias everything, State would be reloaded per sequence
            while (ip < ifast && op < ofast && (!CHECKED || op < ostop)) {
                const unsigned tok = *ip;
                const size_t lit = tok >> 4;
                const unsigned mln = tok & 15u;
                if (lit == 15 || mln == 15) break;
                const size_t off = ld16(ip + 1 + lit);
                if (off == 0 || (CHECKED && off > static_cast<size_t>(op - dst) + lit)) break;
                cp16(op, ip + 1);
                op += lit;
                ip += 3 + lit;
                short_match<kLoopBWide>(op, off, mln + 4);
                op += mln + 4;
                if (CHECKED) {
                    // which loop suits the stream is judged while the first 64 KiB of the block are
                    // produced (the checked phase); the long unchecked phase then carries no accounting
                    bare += (tok < 0x0Fu);
                    if (++seqs >= 512) {
                        if (may_pair && bare > seqs - seqs / 10) {  // > 90 % bare matches: loop A territory
                            swap = true;
                            break;
                        }
                        seqs = bare = 0;
                    }
                }
            }
            s.seqs = seqs;
            s.bare = bare;
            if (!swap) goto out;
            s.pairs = true;
            s.slow = 0;
            s.mark = ip;
        }
    }
out:
    s.ip = ip;
    s.op = op;
}

}  // namespace lz4d

// Returns the decoded byte count, or -1 on malformed input / output overflow.  Never reads outside
// [src, src+n) nor writes outside [dst, dst+cap).
inline int64_t lz4_block_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    using namespace lz4d;
    const uint8_t* const iend = src + n;
    uint8_t* const oend = dst + cap;
    if (n == 0) return -1;
    const uint8_t* const ifast = n >= 32 ? iend - 32 : src;
    uint8_t* const ofast = cap >= 96 ? oend - 96 : dst;
    const uint8_t* const ostart = dst + 65535;  // from here on every 16-bit offset lies inside the output
#ifdef FSLZ4_NO_PAIR_LOOP   // measurement switch (tests/perf/lz4_decode_bench.cpp): loop B only
    const bool may_pair = false;
#else
    const bool may_pair = true;
#endif
    State s{src, dst, may_pair, 0, src, 0, 0};
    for (;;) {
        if (s.op < ostart) fast_loops<true>(s, ifast, ofast, dst, ostart, may_pair);
        if (s.op >= ostart) fast_loops<false>(s, ifast, ofast, dst, ostart, may_pair);
        // ---- general path: ONE sequence with every bound checked, then back to the fast loops
        const uint8_t* ip = s.ip;
        uint8_t* op = s.op;
        if (ip >= iend) return -1;
        const unsigned token = *ip++;
        size_t lit = token >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                lit += b;
            } while (b == 255);
        }
        if (lit > static_cast<size_t>(iend - ip) || lit > static_cast<size_t>(oend - op)) return -1;
        if (lit <= 32 && static_cast<size_t>(iend - ip) >= 32 && static_cast<size_t>(oend - op) >= 32) {
            std::memcpy(op, ip, 32);  // short literal run: one fixed-size copy, tail bytes are overwritten later
        } else {
            std::memcpy(op, ip, lit);
        }
        ip += lit;
        op += lit;
        if (ip == iend) {  // last sequence carries literals only
            s.op = op;
            break;
        }
        if (iend - ip < 2) return -1;
        const size_t off = static_cast<size_t>(ip[0]) | (static_cast<size_t>(ip[1]) << 8);
        ip += 2;
        if (off == 0 || off > static_cast<size_t>(op - dst)) return -1;
        size_t ml = token & 15u;
        if (ml == 15) {
            unsigned b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (ml > static_cast<size_t>(oend - op)) return -1;
        const uint8_t* m = op - off;
        if (off >= 16 && static_cast<size_t>(oend - op) >= ml + 16) {
            // non-overlapping at 16-byte granularity: wild copy in 16-byte steps
            uint8_t* d = op;
            const uint8_t* const dend = op + ml;
            do {
                cp16(d, m);
                d += 16;
                m += 16;
            } while (d < dend);
        } else if (off >= ml) {
            std::memcpy(op, m, ml);
        } else if (off == 1) {
            std::memset(op, m[0], ml);  // run of one byte
        } else if (off == 2 || off == 4) {
            // run of one flag / one flag pair: build the 8-byte period once, store it in 8-byte steps
            // while there is room, finish byte-wise
            uint64_t pat;
            if (off == 2) {
                uint16_t h;
                std::memcpy(&h, m, 2);
                pat = 0x0001000100010001ull * h;
            } else {
                uint32_t q;
                std::memcpy(&q, m, 4);
                pat = 0x0000000100000001ull * q;
            }
            size_t i = 0;
            for (; i + 8 <= ml; i += 8) std::memcpy(op + i, &pat, 8);
            for (; i < ml; ++i) op[i] = op[i - off];
        } else {
            for (size_t i = 0; i < ml; ++i) op[i] = m[i];  // overlapping run (RLE-like)
        }
        op += ml;
        s.ip = ip;
        s.op = op;
    }
    return s.op - dst;
}

}  // namespace fslz4

#endif
