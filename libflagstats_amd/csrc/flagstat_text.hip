// flagstat_text.hip -- SURVEY.md section 8 row f3: the reference's `utility` input maker
// (benchmark/utility.cpp:9-16: `samtools view FILE | cut -f 2 | utility > FLAGS.bin`), the data-format
// step in FRONT of the hot path: decimal FLAG text, one value per line, -> the uint16 array K1 reads.
// Host code (text parsing is branchy byte work next to the file; the array then goes through
// FLAGSTATS_u16_x64 / a streaming session like any other host array).
//
// Line and number rules are std::getline + atoi's, as in the reference: lines end at '\n'; a final
// unterminated line counts, a trailing '\n' opens no extra line; an empty line is 0; leading white
// space and one sign are skipped, digits are read up to the first other character ("99\r" is 99),
// anything non-numeric is 0; the int is truncated to 16 bits.
#include <stdint.h>

#include "../../include/libflagstats_hip.h"
#include "flagstat_engine.h"

extern "C" int64_t FLAGSTATS_text_to_u16(const char* text, uint64_t len, uint16_t* out, uint64_t cap)
{
    if (len && !text) return fsint::fail_text("NULL text");
    if (cap && !out) return fsint::fail_text("NULL out");
    uint64_t n = 0;
    const char* p = text;
    const char* const end = text + len;
    while (p < end) {
        // one line: [p, eol)
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f' || *p == '\r')) ++p;
        bool neg = false;
        if (p < end && (*p == '+' || *p == '-')) neg = (*p++ == '-');
        int64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            v = v * 10 + (*p++ - '0');
            if (v > 0x7FFFFFFFll + (neg ? 1 : 0)) v = 0x7FFFFFFFll + (neg ? 1 : 0);  // strtol-style clamp of atoi's int
        }
        while (p < end && *p != '\n') ++p;   // rest of the line is ignored
        if (p < end) ++p;                    // the '\n'
        if (n >= cap) return fsint::fail_text("FLAGSTATS_text_to_u16: more lines than the output holds");
        out[n++] = static_cast<uint16_t>(static_cast<uint32_t>(static_cast<int32_t>(neg ? -v : v)));
    }
    return static_cast<int64_t>(n);
}

extern "C" uint64_t FLAGSTATS_text_count_lines(const char* text, uint64_t len)
{
    if (!text || !len) return 0;
    uint64_t n = 0;
    for (uint64_t i = 0; i < len; ++i) n += (text[i] == '\n');
    return n + (text[len - 1] != '\n');
}
