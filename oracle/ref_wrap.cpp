/*
 * ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" wrappers around the reference's own header-static kernels so
 * the unmodified reference can be called from tests / bench.py's cpu_baseline
 * through ctypes.  No reference source is copied: the reference header is
 * #included from where it lies (REF_HEADER = /root/reference/libflagstats.h,
 * with -I/root/reference/python so its `#include "libalgebra.h"` resolves to
 * the vendored copy; the libalgebra/ submodule dir is empty).
 * Built by oracle/Makefile into oracle/_ref/libflagstats_ref.so (git-ignored).
 *
 * Built WITHOUT -march=native: the reference selects ISA per function with
 * target attributes (python/libalgebra.h:286-290) and dispatches at run time
 * by cpuid (libflagstats.h:2976-3022), so the .so also runs on the GPU box's
 * host CPU, whatever it is.
 */
#include <cstdint>
#include <cstring>

#include REF_HEADER

extern "C" {

int ref_FLAGSTAT_scalar(const uint16_t* a, uint32_t n, uint32_t* flags)
{
    return FLAGSTAT_scalar(a, n, flags); /* libflagstats.h:170-176 */
}

/* what the reference's public entry point does on this host
 * (libflagstats.h:3024-3070) */
uint64_t ref_FLAGSTATS_u16(const uint16_t* a, uint32_t n, uint32_t* flags)
{
    return FLAGSTATS_u16(a, n, flags);
}

/* name of the kernel FLAGSTATS_get_function(n) picks on this host
 * (libflagstats.h:2976-3022) */
const char* ref_dispatch_name(uint32_t n)
{
    FLAGSTATS_func f = FLAGSTATS_get_function(n);
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

int ref_cpuid(void) { return STORM_get_cpuid(); }
int ref_has_avx512bw(void) { return (STORM_get_cpuid() & STORM_CPUID_runtime_bit_AVX512BW) != 0; }
int ref_has_avx2(void) { return (STORM_get_cpuid() & STORM_CPUID_runtime_bit_AVX2) != 0; }
int ref_has_sse42(void) { return (STORM_get_cpuid() & STORM_CPUID_runtime_bit_SSE42) != 0; }

/* Individual SIMD variants; each returns -1 when the host lacks the ISA. */
#if defined(STORM_HAVE_SSE42)
int ref_FLAGSTAT_sse4(const uint16_t* a, uint32_t n, uint32_t* f)
{
    return ref_has_sse42() ? FLAGSTAT_sse4(a, n, f) : -1;
}
#endif
#if defined(STORM_HAVE_AVX2)
int ref_FLAGSTAT_avx2(const uint16_t* a, uint32_t n, uint32_t* f)
{
    return ref_has_avx2() ? FLAGSTAT_avx2(a, n, f) : -1;
}
#endif
#if defined(STORM_HAVE_AVX512)
int ref_FLAGSTAT_avx512(const uint16_t* a, uint32_t n, uint32_t* f)
{
    return ref_has_avx512bw() ? FLAGSTAT_avx512(a, n, f) : -1; /* :1644-1846 */
}
int ref_FLAGSTAT_avx512_improved3(const uint16_t* a, uint32_t n, uint32_t* f)
{
    return ref_has_avx512bw() ? FLAGSTAT_avx512_improved3(a, n, f) : -1; /* :2445-2644 */
}
#endif

/* libalgebra's positional popcount entry (python/libalgebra.h:3496-3551; zeroes out[16] first)
 * and its scalar statement (:566-574; accumulates) */
int ref_STORM_pospopcnt_u16(const uint16_t* a, size_t n, uint32_t* out) { return STORM_pospopcnt_u16(a, n, out); }
int ref_STORM_pospopcnt_u16_scalar_naive(const uint16_t* a, size_t n, uint32_t* out)
{
    return STORM_pospopcnt_u16_scalar_naive(a, n, out);
}

/* 64-bit convenience for the CPU baseline: chunk into <= 2^30-flag calls of
 * the dispatcher's choice, private uint32[32] per chunk, summed to uint64
 * (BASELINE.md section 4 step 2/3). */
void ref_dispatch_x64(const uint16_t* a, uint64_t n, uint64_t out[32])
{
    const uint64_t CH = 1ull << 30;
    for (uint64_t done = 0; done < n;) {
        uint64_t c = n - done;
        if (c > CH) c = CH;
        uint32_t part[32];
        std::memset(part, 0, sizeof part);
        FLAGSTATS_func f = FLAGSTATS_get_function((uint32_t)c);
        (*f)(a + done, (uint32_t)c, part);
        for (int i = 0; i < 32; ++i) out[i] += part[i];
        done += c;
    }
}

/* `reps` passes inside one call, so a Python caller times the kernel, not ctypes */
void ref_dispatch_repeat(const uint16_t* a, uint64_t n, uint32_t reps, uint64_t out[32])
{
    for (uint32_t r = 0; r < reps; ++r) ref_dispatch_x64(a, n, out);
}

void ref_scalar_x64(const uint16_t* a, uint64_t n, uint64_t out[32])
{
    const uint64_t CH = 1ull << 30;
    for (uint64_t done = 0; done < n;) {
        uint64_t c = n - done;
        if (c > CH) c = CH;
        uint32_t part[32];
        std::memset(part, 0, sizeof part);
        FLAGSTAT_scalar(a + done, (uint32_t)c, part);
        for (int i = 0; i < 32; ++i) out[i] += part[i];
        done += c;
    }
}

} /* extern "C" */
