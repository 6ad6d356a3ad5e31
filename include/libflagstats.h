/*
 * libflagstats.h -- header shim.
 *
 * The reference ships its whole library as one header of `static` functions
 * (/root/reference/libflagstats.h); consumers `#include "libflagstats.h"` and
 * call FLAGSTATS_u16 / FLAGSTATS_get_function (python/libflagstats.pyx:5-6,
 * benchmark/flagstats.cpp:38-39).  Put THIS directory first on the include path
 * and link -lflagstats_hip: the same names resolve to the MI355X engine in
 * libflagstats_hip.so instead, with no source change in the consumer.
 *
 * Shimmed: the SAM FLAG constants (values are the SAM specification's; names as
 * at libflagstats.h:69-112), the three dispatch symbols (libflagstats.h:2970,
 * :2976-2977, :3024-3025) and -- through the libalgebra.h shim next to this
 * file, which the reference's header also includes (:61) -- the STORM_* helpers
 * its block readers call (aligned malloc / free, alignment query).  With those,
 * /root/reference/benchmark/flagstats.cpp builds unmodified against this
 * directory (oracle/Makefile target refbench -> oracle/_ref/bench_hip; run on the GPU by
 * tests/test_blockfiles_golden.py::test_reference_main_program_on_the_gpu_engine).
 */
#ifndef LIBFLAGSTATS_H_SHIM_HIP_
#define LIBFLAGSTATS_H_SHIM_HIP_

#include <stdint.h>

#include "libalgebra.h"

/* SAM FLAG bits and their counter-slot offsets */
#define FLAGSTAT_FPAIRED 1
#define FLAGSTAT_FPAIRED_OFF 0
#define FLAGSTAT_FPROPER_PAIR 2
#define FLAGSTAT_FPROPER_PAIR_OFF 1
#define FLAGSTAT_FUNMAP 4
#define FLAGSTAT_FUNMAP_OFF 2
#define FLAGSTAT_FMUNMAP 8
#define FLAGSTAT_FMUNMAP_OFF 3
#define FLAGSTAT_FREVERSE 16
#define FLAGSTAT_FREVERSE_OFF 4
#define FLAGSTAT_FMREVERSE 32
#define FLAGSTAT_FMREVERSE_OFF 5
#define FLAGSTAT_FREAD1 64
#define FLAGSTAT_FREAD1_OFF 6
#define FLAGSTAT_FREAD2 128
#define FLAGSTAT_FREAD2_OFF 7
#define FLAGSTAT_FSECONDARY 256
#define FLAGSTAT_FSECONDARY_OFF 8
#define FLAGSTAT_FQCFAIL 512
#define FLAGSTAT_FQCFAIL_OFF 9
#define FLAGSTAT_FDUP 1024
#define FLAGSTAT_FDUP_OFF 10
#define FLAGSTAT_FSUPPLEMENTARY 2048
#define FLAGSTAT_FSUPPLEMENTARY_OFF 11
/* synthetic slots: n_pair_good, n_sgltn, n_pair_map */
#define FLAGSTAT_BIT12 (1 << 12)
#define FLAGSTAT_BIT12_OFF 12
#define FLAGSTAT_BIT13 (1 << 13)
#define FLAGSTAT_BIT13_OFF 13
#define FLAGSTAT_BIT14 (1 << 14)
#define FLAGSTAT_BIT14_OFF 14

#include "libflagstats_hip.h"

#endif
