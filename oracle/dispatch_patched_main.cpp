/*
 * dispatch_patched_main.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Exercises INTEGRATION.md section B: PATCHED_HEADER is a scratch copy of the reference's libflagstats.h
 * with integration/apply_dispatch_patch.py applied (made in a temp dir by oracle/Makefile, never stored).
 * Two builds of this file:
 *   _ref/dispatch_patched       linked with libflagstats_hip.so          (runs on the GPU box too)
 *   _ref/dispatch_patched_stub  -DSTUB_ENGINE: the two engine symbols are stubs defined right here, so the
 *                               dispatch RULE can be tested on a machine without a GPU
 *
 *   dispatch_patched <n> [seed]   prints: which kernel FLAGSTATS_get_function(n) chose, then the 32 counters
 *                                 from calling it and from FLAGSTATS_u16 on the same n pseudo-random 12-bit flags
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include PATCHED_HEADER

#ifdef STUB_ENGINE
extern "C" int FLAGSTATS_hip_available(void) { return getenv("STUB_NO_GPU") ? 0 : 1; }
extern "C" int FLAGSTAT_hip(const uint16_t* a, uint32_t n, uint32_t* flags)
{
    FLAGSTAT_scalar(a, n, flags); /* stand-in arithmetic; slot 31 marks that the engine branch ran */
    flags[31] += 0xABCD;
    return 0;
}
#endif

static const char* name_of(FLAGSTATS_func f)
{
    if (f == &FLAGSTAT_hip) return "FLAGSTAT_hip";
    if (f == &FLAGSTAT_scalar) return "FLAGSTAT_scalar";
#if defined(STORM_HAVE_SSE42)
    if (f == &FLAGSTAT_sse4) return "FLAGSTAT_sse4";
#endif
#if defined(STORM_HAVE_AVX2)
    if (f == &FLAGSTAT_avx2) return "FLAGSTAT_avx2";
#endif
#if defined(STORM_HAVE_AVX512)
    if (f == &FLAGSTAT_avx512) return "FLAGSTAT_avx512";
#endif
    return "unknown";
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 1000;
    uint64_t x = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    std::vector<uint16_t> a(n ? n : 1);
    for (uint32_t i = 0; i < n; ++i) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        a[i] = (uint16_t)(((x * 0x2545F4914F6CDD1Dull) >> 48) & 0x0FFF); /* bits 12-15 clear: all kernels agree on the live slots */
    }
    FLAGSTATS_func f = FLAGSTATS_get_function(n);
    printf("chosen %s\n", name_of(f));
    uint32_t c1[32] = {0}, c2[32] = {0}, c3[32] = {0};
    (*f)(a.data(), n, c1);
    FLAGSTATS_u16(a.data(), n, c2);
    FLAGSTAT_scalar(a.data(), n, c3);
    printf("func");  for (int i = 0; i < 32; ++i) printf(" %u", c1[i]); printf("\n");
    printf("u16");   for (int i = 0; i < 32; ++i) printf(" %u", c2[i]); printf("\n");
    printf("scalar"); for (int i = 0; i < 32; ++i) printf(" %u", c3[i]); printf("\n");
    return 0;
}
