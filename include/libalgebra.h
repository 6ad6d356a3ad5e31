/*
 * libalgebra.h -- header shim (companion of the libflagstats.h shim in this directory).
 *
 * The reference's header and its bench programs `#include "libalgebra.h"`
 * (/root/reference/libflagstats.h:61, benchmark/flagstats.cpp:37) for three things on the flagstat
 * path: the aligned allocator the block readers use for their decode buffer
 * (STORM_aligned_malloc / STORM_aligned_free, python/libalgebra.h:112-139, called at
 * benchmark/flagstats.cpp:239,284,300,356,381,411,436,466,491,527), the alignment query
 * (STORM_get_alignment, python/libalgebra.h:3045-3088) and the plain positional popcount
 * (STORM_pospopcnt_u16, :3496-3551 -- exported by libflagstats_hip.so, declared in
 * libflagstats_hip.h).  With this directory first on the include path those consumers compile
 * unchanged; nothing else of libalgebra (cpuid, set algebra, popcount kernels) is on the path.
 */
#ifndef LIBALGEBRA_H_SHIM_HIP_
#define LIBALGEBRA_H_SHIM_HIP_

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifndef STORM_aligned_malloc
/* same contract as the reference's: `alignment` a power of two multiple of sizeof(void*); NULL on failure */
static inline void* STORM_aligned_malloc(size_t alignment, size_t size)
{
    void* p = NULL;
    return posix_memalign(&p, alignment, size) == 0 ? p : NULL;
}
#endif

#ifndef STORM_aligned_free
static inline void STORM_aligned_free(void* memblock) { free(memblock); }
#endif

/* The reference answers with the vector width of the best x86 ISA (64 / 32 / 16 / 8).  The GPU engine
 * reads host arrays at any 2-byte alignment; 64 (one cache line, the reference's AVX-512 answer) keeps
 * a decode buffer from straddling lines for the host-side memcpy / DMA. */
static inline uint32_t STORM_get_alignment(void) { return 64; }

#include "libflagstats_hip.h" /* STORM_pospopcnt_u16 */

#endif
