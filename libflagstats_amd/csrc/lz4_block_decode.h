// lz4_block_decode.h -- the host LZ4 *block* decoder of the block-file pipeline (SURVEY.md section 8
// row f1; the reference calls liblz4's LZ4_decompress_safe, benchmark/flagstats.cpp:316).  Written from
// the published block format (token = 4 bits literal length | 4 bits match length - 4, 255-continued
// length bytes, 2-byte little-endian offset, last sequence literals-only); checked in tests/ against the
// image's liblz4.so.1 acting as oracle.  Header-only so that tests/perf/lz4_decode_bench.cpp can time
// exactly the code the library runs.
#ifndef FLAGSTAT_LZ4_BLOCK_DECODE_H_
#define FLAGSTAT_LZ4_BLOCK_DECODE_H_

#include <stddef.h>
#include <stdint.h>

#include <cstring>

namespace fslz4 {

// Returns decoded byte count, or -1 on malformed input / output overflow.  Never reads outside
// [src, src+n) nor writes outside [dst, dst+cap).
//
// Shaped for FLAG streams, not for text: a 1,024,000-byte block of NA12878-like flags is ~156,000
// sequences of 6.6 output bytes, 95 % of them "no literals, match of 4..18 bytes" = exactly 3 input
// bytes (token, 2-byte offset).  A byte-at-a-time parser is bound by the token -> next-token address
// chain (one L1 load latency per sequence, where liblz4 also sits).  The fast loop below loads 8 input
// bytes at once and retires up to TWO such sequences per load, so that chain is paid once per pair.
namespace lz4d {

inline uint64_t ld64(const uint8_t* p)
{
    uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
}
inline void cp8(uint8_t* d, const uint8_t* s)
{
    uint64_t v;
    std::memcpy(&v, s, 8);
    std::memcpy(d, &v, 8);
}
inline void cp16(uint8_t* d, const uint8_t* s)
{
    struct V16 { uint64_t a, b; } v;   // one 16-byte (SSE) load and store
    std::memcpy(&v, s, 16);
    std::memcpy(d, &v, 16);
}
// match of ml <= 18 bytes at distance off >= 8.  off >= 16 (97 % of a FLAG stream's matches): one
// 16-byte load/store (+ 8 more for ml 17, 18).  8 <= off < 16: 8-byte pieces in order reproduce LZ4's
// overlapping-copy semantics.  Few, wide stores matter: a later match often reads bytes that are still
// in the store buffer, and a load that straddles two pending stores cannot be forwarded.
inline void short_match(uint8_t* op, size_t off, size_t ml)
{
    const uint8_t* m = op - off;
    if (off >= 16) {
        cp16(op, m);
        if (ml > 16) cp8(op + 16, m + 16);
    } else {
        cp8(op, m);
        cp8(op + 8, m + 8);
        cp8(op + 16, m + 16);
    }
}

}  // namespace lz4d

inline int64_t lz4_block_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    using namespace lz4d;
    const uint8_t* ip = src;
    const uint8_t* const iend = src + n;
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    if (n == 0) return -1;
    // Fast region: while ip < ifast there are >= 32 readable input bytes (an iteration reads 9) --
    // which also rules out the literals-only last sequence, whose literals would have to run to iend
    // (> 14 of them) -- and while op < ofast there are >= 96 writable output bytes (an iteration
    // writes < 72).  It starts once 64 KiB have been produced: from there every 16-bit offset points
    // inside the output, so the loop carries no offset-vs-start check.
    const uint8_t* const ifast = n >= 32 ? iend - 32 : src;
    uint8_t* const ofast = cap >= 96 ? oend - 96 : dst;
    uint8_t* const ostart = dst + 65535;
#ifdef FSLZ4_NO_PAIR_LOOP   // measurement switch (tests/perf/lz4_decode_bench.cpp): loop B only
    bool pairs = false;
    constexpr bool may_pair = false;
#else
    bool pairs = true;  // which fast loop suits the stream; re-judged every few hundred sequences
    constexpr bool may_pair = true;
#endif
    for (;;) {
        if (op >= ostart && pairs) {
            // Loop A -- streams of bare matches (LZ4-fast on flags: 95 % of the sequences).  One 8-byte
            // load holds two whole 3-byte sequences (token, offset; no literals, match nibble < 15).
            // Under correctly predicted branches the next load address (ip + 6) does not depend on the
            // loaded bytes, so the loads run ahead of the parsing instead of waiting for each token.
            unsigned slow = 0;
            const uint8_t* mark = ip;
            while (ip < ifast && op < ofast) {
                const uint64_t w = ld64(ip);
                const unsigned tok = static_cast<unsigned>(w & 0xFFu);
                if (tok < 0x0Fu) {
                    const size_t off = static_cast<size_t>((w >> 8) & 0xFFFFu);
                    if (off < 8) break;
                    short_match(op, off, tok + 4);
                    op += tok + 4;
                    const unsigned tok2 = static_cast<unsigned>((w >> 24) & 0xFFu);
                    if (tok2 < 0x0Fu) {
                        const size_t off2 = static_cast<size_t>((w >> 32) & 0xFFFFu);
                        if (off2 < 8) {
                            ip += 3;
                            break;
                        }
                        short_match(op, off2, tok2 + 4);
                        op += tok2 + 4;
                        ip += 6;
                        continue;
                    }
                    ip += 3;
                    continue;
                }
                // short sequence with literals (<= 14) and a short match.  A stream that keeps coming here
                // (LZ4-HC: 1/4 of the sequences carry a literal) mispredicts its way through this loop:
                // 64 visits within fewer than ~8 sequences each hand it to loop B.
                if (++slow == 64) {
                    if (ip - mark < 64 * 24) {
                        pairs = false;
                        break;
                    }
                    slow = 0;
                    mark = ip;
                }
                const size_t lit = tok >> 4;
                const unsigned mln = tok & 15u;
                if (lit == 15 || mln == 15) break;  // length continuation bytes: general path
                const size_t off = static_cast<size_t>(ip[1 + lit]) | (static_cast<size_t>(ip[2 + lit]) << 8);
                if (off < 8) break;
                cp16(op, ip + 1);  // one fixed 16-byte copy, the surplus is overwritten by the match
                op += lit;
                ip += 3 + lit;
                short_match(op, off, mln + 4);
                op += mln + 4;
            }
            if (!pairs) continue;
        } else if (op >= ostart) {
            // Loop B -- mixed streams: every short sequence (<= 14 literals, match <= 18) takes the same
            // branch-free-in-the-literal-count route; the token -> next-token address chain is paid per
            // sequence (where liblz4 sits too), but nothing here depends on what the tokens look like.
            unsigned seqs = 0, bare = 0;
            while (ip < ifast && op < ofast) {
                const unsigned tok = *ip;
                const size_t lit = tok >> 4;
                const unsigned mln = tok & 15u;
                if (lit == 15 || mln == 15) break;
                const size_t off = static_cast<size_t>(ip[1 + lit]) | (static_cast<size_t>(ip[2 + lit]) << 8);
                if (off < 8) break;
                cp16(op, ip + 1);
                op += lit;
                ip += 3 + lit;
                short_match(op, off, mln + 4);
                op += mln + 4;
                bare += (tok < 0x0Fu);
                if (++seqs == 512) {
                    if (may_pair && bare > 460) {  // > 90 % bare matches: loop A territory
                        pairs = true;
                        break;
                    }
                    seqs = bare = 0;
                }
            }
            if (pairs) continue;
        }
        // ---- general path: ONE sequence with every bound checked, then back to the fast loop
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
        if (ip == iend) break;  // last sequence carries literals only
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
            // run of one flag / one flag pair (the common small offsets of a FLAG stream): build the
            // 8-byte period once, store it in 8-byte steps while there is room, finish byte-wise
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
    }
    return op - dst;
}

}  // namespace fslz4

#endif
