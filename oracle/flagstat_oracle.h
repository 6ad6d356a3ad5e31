/*
 * flagstat_oracle.h -- CPU restatement of libflagstats' FLAGSTAT_scalar path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under libflagstats_amd/ (the product) may
 * include, link or dlopen this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py,
 * oracle/selfcheck.c) against
 *   - the reference's own FLAGSTAT_scalar / FLAGSTAT_avx512_improved3 compiled
 *     from /root/reference (oracle/_ref/libflagstats_ref.so, built by
 *     oracle/Makefile) over the length x range matrix of SURVEY.md section 4, and
 *   - the committed golden vectors under tests/golden/ that were produced by
 *     that reference build (tests/golden/make_golden.py).
 *
 * Reference lines followed:
 *   libflagstats.h:69-112   FLAGSTAT_* bit constants / counter slot offsets
 *   libflagstats.h:118-142  FLAGSTAT_scalar_update  (per-flag rule, 19 live slots)
 *   libflagstats.h:170-176  FLAGSTAT_scalar         (loop, accumulate with +=)
 */
#ifndef FLAGSTAT_ORACLE_H_
#define FLAGSTAT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One flag -> += into 32 slots ([0..15] pass-QC, [16..31] fail-QC).
 * Follows libflagstats.h:118-142 statement by statement. */
void oracle_flagstat_update(uint16_t val, uint64_t out[32]);

/* Loop form, libflagstats.h:170-176, with 64-bit length and 64-bit counters
 * (the reference's uint32 len/counters cannot hold 2^32 flags). */
void oracle_flagstat_u16(const uint16_t* array, uint64_t n, uint64_t out[32]);

/* Same result through a 65536-bin histogram: counters = sum_v hist[v] *
 * update(v).  Exactly equal because the rule is per-flag and integer; ~20x
 * faster than the branchy loop, used for the multi-GiB parity checks. */
void oracle_flagstat_hist_u16(const uint16_t* array, uint64_t n, uint64_t out[32]);

/* Histogram form over `threads` contiguous shards (pthreads). */
void oracle_flagstat_mt_u16(const uint16_t* array, uint64_t n, int threads, uint64_t out[32]);

/* uint32 ABI twin of the reference signature (accumulates, returns 0). */
int oracle_FLAGSTAT_scalar(const uint16_t* array, uint32_t len, uint32_t* flags);

/* Plain 16-bit positional popcount, restating STORM_pospopcnt_u16_scalar_naive
 * (python/libalgebra.h:566-574): out[j] += number of words with bit j set (row f4). */
void oracle_pospopcnt_u16(const uint16_t* array, uint64_t n, uint64_t out[16]);

/* The samtools counting loop of the reference's bench (`flagstat_loop`, benchmark/flagstats.cpp:51-70):
 * out[2 * field + w] += ..., fields in the order of bam_flagstat_t (:42-48), w = 1 for fail-QC reads.
 * Oracle of the report layer (row f2): n_pair_all is counted here, not derived. */
void oracle_samtools_update(uint16_t c, uint64_t out[26]);
void oracle_samtools_u16(const uint16_t* array, uint64_t n, uint64_t out[26]);

/* ---- host twins of the product's on-device input makers ------------------
 * (libflagstats_amd/csrc/flagstat_generate.hip).  Counter-based, so any
 * sub-range can be regenerated independently.  kind:
 *   0 = uniform   : 16-bit slices of mix64(seed, i/4), & mask
 *                   (mask 0x0FFF == benchmark/generate.cpp:8-14's U[0,4095])
 *   1 = NA12878-like categorical draw from README.md:178-192 marginals
 *       (SURVEY.md section 8(d) config 3); mask bit0 = "+eps" variant that ORs
 *       ~1 % FDUP and ~0.1 % FQCFAIL
 *   2 = ramp      : x[i] = (uint16_t)(i + seed)   (exhaustive KAT, repeated)
 */
#define ORACLE_GEN_UNIFORM 0
#define ORACLE_GEN_NA12878 1
#define ORACLE_GEN_RAMP    2
void oracle_generate_u16(int kind, uint64_t seed, uint32_t mask,
                         uint64_t first_index, uint64_t n, uint16_t* out);

/* generate [first_index, first_index+n) chunk-wise on `threads` threads and
 * count it, never holding more than a few MiB. */
void oracle_flagstat_generated(int kind, uint64_t seed, uint32_t mask,
                               uint64_t first_index, uint64_t n, int threads,
                               uint64_t out[32]);

#ifdef __cplusplus
}
#endif
#endif
