/*
 * selfcheck.c -- TEST INFRASTRUCTURE ONLY.
 * Pins the restatement (liboracle.so) to the reference itself
 * (_ref/libflagstats_ref.so = /root/reference's FLAGSTAT_scalar and
 * FLAGSTAT_avx512_improved3) over the length x range matrix of SURVEY.md
 * section 4, plus the exhaustive 0..65535 sweep and every single-flag value.
 * Exit code 0 = all 32 slots equal everywhere.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flagstat_oracle.h"

int ref_FLAGSTAT_scalar(const uint16_t*, uint32_t, uint32_t*);
int ref_FLAGSTAT_avx512_improved3(const uint16_t*, uint32_t, uint32_t*);

static int cmp(const char* what, uint64_t n, const uint32_t* ref, const uint64_t* got)
{
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        if ((uint64_t)ref[i] != got[i]) {
            fprintf(stderr, "MISMATCH %s n=%llu slot %d: ref=%u got=%llu\n", what,
                    (unsigned long long)n, i, ref[i], (unsigned long long)got[i]);
            bad = 1;
        }
    return bad;
}

int main(void)
{
    static const uint64_t lens[] = {0, 1, 2, 7, 8, 9, 127, 128, 255, 256, 257, 511, 512, 513,
                                    767, 768, 1023, 1024, 1025, 4095, 4096, 65535, 65536, 65537,
                                    131071, 131073, 512000, 1048575, 1048576, 1048577, 3000001};
    static const uint32_t masks[] = {0x0FFF, 0xFFFF};
    int bad = 0, checks = 0;
    const uint64_t maxn = 3000001;
    uint16_t* buf = (uint16_t*)malloc(sizeof(uint16_t) * (maxn + 1));

    /* every single value on its own */
    for (uint32_t v = 0; v < 65536; ++v) {
        uint16_t x = (uint16_t)v;
        uint32_t r[32] = {0};
        uint64_t g[32] = {0};
        ref_FLAGSTAT_scalar(&x, 1, r);
        oracle_flagstat_update(x, g);
        bad |= cmp("single", v, r, g);
        ++checks;
    }

    for (size_t m = 0; m < sizeof masks / sizeof masks[0]; ++m)
        for (size_t l = 0; l < sizeof lens / sizeof lens[0]; ++l)
            for (int misalign = 0; misalign < 2; ++misalign) {
                const uint64_t n = lens[l];
                uint16_t* a = buf + misalign;
                oracle_generate_u16(ORACLE_GEN_UNIFORM, 1234 + l, masks[m], 17 * l, n, a);
                uint32_t r[32] = {0}, r3[32] = {0};
                ref_FLAGSTAT_scalar(a, (uint32_t)n, r);
                const int have3 = ref_FLAGSTAT_avx512_improved3(a, (uint32_t)n, r3) == 0;
                uint64_t g1[32] = {0}, g2[32] = {0}, g3[32] = {0};
                oracle_flagstat_u16(a, n, g1);
                oracle_flagstat_hist_u16(a, n, g2);
                oracle_flagstat_mt_u16(a, n, 3, g3);
                bad |= cmp("loop-vs-scalar", n, r, g1);
                bad |= cmp("hist-vs-scalar", n, r, g2);
                bad |= cmp("mt-vs-scalar", n, r, g3);
                if (have3) bad |= cmp("loop-vs-avx512_improved3", n, r3, g1);
                checks += 3 + have3;
            }

    /* NA12878-like and ramp makers through the generated-stream path */
    for (int kind = 0; kind < 3; ++kind) {
        const uint64_t n = 2000003;
        oracle_generate_u16(kind, 99, kind == 0 ? 0xFFFF : 1, 5, n, buf);
        uint32_t r[32] = {0};
        ref_FLAGSTAT_scalar(buf, (uint32_t)n, r);
        uint64_t g[32] = {0};
        oracle_flagstat_generated(kind, 99, kind == 0 ? 0xFFFF : 1, 5, n, 4, g);
        bad |= cmp("generated-vs-scalar", n, r, g);
        ++checks;
    }

    free(buf);
    printf("oracle selfcheck: %d comparisons, %s\n", checks, bad ? "FAILED" : "all equal");
    return bad;
}
