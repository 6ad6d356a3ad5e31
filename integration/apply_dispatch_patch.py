#!/usr/bin/env python3
"""INTEGRATION.md section B as an executable patch: add the MI355X engine as the FIRST branch of the
reference's own dispatcher, keeping its CPU kernels for short inputs.

    python integration/apply_dispatch_patch.py /path/to/libflagstats.h  patched/libflagstats.h

Three insertions, nothing else changes (anchors are the reference's own lines, libflagstats.h @ v0.1.x):
  1. after `typedef int (*FLAGSTATS_func)...` (:2970): declarations of the two symbols taken from
     libflagstats_hip.so and the selection helper FLAGSTATS_hip_wanted(n_len);
  2. in FLAGSTATS_get_function, ahead of the AVX-512 branch (:2999): return &FLAGSTAT_hip;
  3. in FLAGSTATS_u16, ahead of the AVX-512 branch (:3047): return FLAGSTAT_hip(array, n_len, flags).
Everything is inside `#if defined(FLAGSTATS_HAVE_HIP)`: without that macro the header is the reference's.
Build the consumer with -DFLAGSTATS_HAVE_HIP and link -lflagstats_hip.

Selection rule (the length-aware rule of libflagstats.h:2999-3021 extended by one branch):
  GPU  iff  n_len >= FLAGSTATS_HIP_MIN_LEN (compile-time default 2^17, env FLAGSTATS_HIP_MIN_LEN overrides)
            and env FLAGSTATS_BACKEND is not "cpu"  and  FLAGSTATS_hip_available();
  otherwise the reference's own rule picks among its CPU kernels, unchanged.
"""
import sys

DECLS = r'''
/* ---- MI355X engine (libflagstats_hip.so): first branch of the dispatcher; INTEGRATION.md section B ---- */
#if defined(FLAGSTATS_HAVE_HIP)
#include <stdlib.h>
#include <string.h>
int FLAGSTAT_hip(const uint16_t* array, uint32_t len, uint32_t* flags); /* same shape as every FLAGSTAT_<impl> */
int FLAGSTATS_hip_available(void);
#ifndef FLAGSTATS_HIP_MIN_LEN
#define FLAGSTATS_HIP_MIN_LEN (1u << 17) /* break-even of a host-pointer call vs FLAGSTAT_avx512 on 2x EPYC 9575F:
                                          * ~1.1e5 flags (13 us call floor; 131,072 flags: 23 vs 24 us; 512,000: 39 vs 93 us;
                                          * profiles/r03/small_calls.log) */
#endif
static int FLAGSTATS_hip_wanted(uint32_t n_len)
{
    /* -1 unknown, 0 no, 1 yes; benign race: every thread computes the same answers */
    static int backend_ok = -1;
    static long long min_len = -1;
    if (min_len < 0) {
        const char* s = getenv("FLAGSTATS_HIP_MIN_LEN");
        min_len = (s && *s) ? atoll(s) : (long long)FLAGSTATS_HIP_MIN_LEN;
    }
    if ((long long)n_len < min_len) return 0;
    if (backend_ok < 0) {
        const char* b = getenv("FLAGSTATS_BACKEND");
        backend_ok = (b && strcmp(b, "cpu") == 0) ? 0 : (FLAGSTATS_hip_available() ? 1 : 0);
    }
    return backend_ok;
}
#endif
'''

BRANCH_FUNC = r'''
#if defined(FLAGSTATS_HAVE_HIP)
    if (FLAGSTATS_hip_wanted(n_len)) {
        return &FLAGSTAT_hip;
    }
#endif
'''

BRANCH_CALL = r'''
#if defined(FLAGSTATS_HAVE_HIP)
    if (FLAGSTATS_hip_wanted(n_len)) {
        return FLAGSTAT_hip(array, n_len, flags);
    }
#endif
'''


def insert_before_guard(text, needle, payload):
    """Insert payload before the `#if defined(STORM_HAVE_AVX512)` line that guards `needle`."""
    at = text.index(needle)
    guard = text.rindex("#if defined(STORM_HAVE_AVX512)", 0, at)
    return text[:guard] + payload.lstrip("\n") + "\n" + text[guard:]


def patch(text):
    typedef = "typedef int (*FLAGSTATS_func)(const uint16_t*, uint32_t, uint32_t*);"
    assert text.count(typedef) == 1, "anchor 1 (FLAGSTATS_func typedef) not found exactly once"
    assert "FLAGSTATS_HAVE_HIP" not in text, "already patched"
    at = text.index(typedef) + len(typedef)
    text = text[:at] + "\n" + DECLS + text[at:]
    text = insert_before_guard(text, "return &FLAGSTAT_avx512;", BRANCH_FUNC)
    text = insert_before_guard(text, "return FLAGSTAT_avx512(array, n_len, flags);", BRANCH_CALL)
    return text


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    with open(sys.argv[1]) as f:
        out = patch(f.read())
    with open(sys.argv[2], "w") as f:
        f.write(out)
