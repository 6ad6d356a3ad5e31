/*
 * flagstat_oracle.c -- see flagstat_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C restatement of the reference's scalar flagstat rule
 * (libflagstats.h:118-142, :170-176) plus host twins of the product's
 * synthetic input makers.  Parity pinned against oracle/_ref (the reference
 * itself, compiled here) and tests/golden/.
 */
#include "flagstat_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* SAM FLAG bits, libflagstats.h:69-112 (values are the SAM spec's). */
enum {
    F_PAIRED = 0x001, F_PROPER = 0x002, F_UNMAP = 0x004, F_MUNMAP = 0x008,
    F_READ1 = 0x040, F_READ2 = 0x080, F_SECONDARY = 0x100, F_QCFAIL = 0x200,
    F_DUP = 0x400, F_SUPPLEMENTARY = 0x800
};
/* counter slot = bit offset of the flag; 12/13/14 are the synthetic
 * n_pair_good / n_sgltn / n_pair_map slots (libflagstats.h:103-112). */
enum {
    S_UNMAP = 2, S_READ1 = 6, S_READ2 = 7, S_SECONDARY = 8, S_QCFAIL = 9,
    S_DUP = 10, S_SUPPLEMENTARY = 11, S_PAIR_GOOD = 12, S_SGLTN = 13, S_PAIR_MAP = 14
};

void oracle_flagstat_update(uint16_t val, uint64_t out[32])
{
    /* :122-123  fail-QC reads go to the upper 16 slots */
    uint64_t* f = out + ((val & F_QCFAIL) ? 16 : 0);
    /* :127      only the fail class counts its own size (slot 25) */
    if (val & F_QCFAIL) f[S_QCFAIL] += 1;

    /* :129-138  secondary wins over supplementary wins over "primary paired" */
    if (val & F_SECONDARY) {
        f[S_SECONDARY] += 1;
    } else if (val & F_SUPPLEMENTARY) {
        f[S_SUPPLEMENTARY] += 1;
    } else if (val & F_PAIRED) {
        const int mapped = !(val & F_UNMAP);
        if ((val & F_PROPER) && mapped) f[S_PAIR_GOOD] += 1;          /* :133 */
        if (val & F_READ1) f[S_READ1] += 1;                           /* :134 */
        if (val & F_READ2) f[S_READ2] += 1;                           /* :135 */
        if ((val & F_MUNMAP) && mapped) f[S_SGLTN] += 1;              /* :136 */
        if (mapped && !(val & F_MUNMAP)) f[S_PAIR_MAP] += 1;          /* :137 */
    }
    /* :140-141  unconditional */
    if (val & F_UNMAP) f[S_UNMAP] += 1;
    if (val & F_DUP) f[S_DUP] += 1;
}

void oracle_flagstat_u16(const uint16_t* array, uint64_t n, uint64_t out[32])
{
    for (uint64_t i = 0; i < n; ++i) oracle_flagstat_update(array[i], out);
}

int oracle_FLAGSTAT_scalar(const uint16_t* array, uint32_t len, uint32_t* flags)
{
    uint64_t wide[32];
    memset(wide, 0, sizeof wide);
    oracle_flagstat_u16(array, len, wide);
    for (int i = 0; i < 32; ++i) flags[i] += (uint32_t)wide[i];
    return 0;
}

/* ---- histogram evaluation ------------------------------------------------ */

static void hist_accumulate(const uint16_t* a, uint64_t n, uint64_t* hist /*65536*/)
{
    /* four sub-histograms to break the store-to-load dependency on equal keys */
    enum { K = 4 };
    uint32_t* h = (uint32_t*)calloc((size_t)K * 65536, sizeof(uint32_t));
    uint64_t done = 0;
    while (done < n) {
        /* uint32 bins: flush before any bin can wrap */
        uint64_t chunk = n - done;
        if (chunk > 0xFFFFFFF0ull) chunk = 0xFFFFFFF0ull;
        const uint16_t* p = a + done;
        uint64_t i = 0;
        for (; i + K <= chunk; i += K) {
            h[0 * 65536 + p[i + 0]]++;
            h[1 * 65536 + p[i + 1]]++;
            h[2 * 65536 + p[i + 2]]++;
            h[3 * 65536 + p[i + 3]]++;
        }
        for (; i < chunk; ++i) h[p[i]]++;
        for (int k = 0; k < K; ++k)
            for (int v = 0; v < 65536; ++v) {
                hist[v] += h[k * 65536 + v];
                h[k * 65536 + v] = 0;
            }
        done += chunk;
    }
    free(h);
}

static void hist_to_counters(const uint64_t* hist, uint64_t out[32])
{
    for (int v = 0; v < 65536; ++v) {
        if (!hist[v]) continue;
        uint64_t one[32];
        memset(one, 0, sizeof one);
        oracle_flagstat_update((uint16_t)v, one);
        for (int s = 0; s < 32; ++s) out[s] += one[s] * hist[v];
    }
}

/* ---- the samtools counting loop the reference's bench carries beside its own kernels ----
 * Restates the `flagstat_loop` macro, benchmark/flagstats.cpp:51-70, field by field; field order of
 * `bam_flagstat_t` (:42-48): n_reads, n_mapped, n_pair_all, n_pair_map, n_pair_good, n_sgltn, n_read1,
 * n_read2, n_dup, n_diffchr, n_diffhigh, n_secondary, n_supp; out[2 * field + w], w = 1 for fail-QC. */
enum { SAM_READS, SAM_MAPPED, SAM_PAIR_ALL, SAM_PAIR_MAP, SAM_PAIR_GOOD, SAM_SGLTN, SAM_READ1, SAM_READ2, SAM_DUP,
       SAM_DIFFCHR, SAM_DIFFHIGH, SAM_SECONDARY, SAM_SUPP };

void oracle_samtools_update(uint16_t c, uint64_t out[26])
{
    const int w = (c & F_QCFAIL) ? 1 : 0;                                        /* :52 */
    out[2 * SAM_READS + w] += 1;                                                 /* :53 */
    if (c & F_SECONDARY) {                                                       /* :54 */
        out[2 * SAM_SECONDARY + w] += 1;
    } else if (c & F_SUPPLEMENTARY) {                                            /* :56 */
        out[2 * SAM_SUPP + w] += 1;
    } else if (c & F_PAIRED) {                                                   /* :58 */
        out[2 * SAM_PAIR_ALL + w] += 1;
        if ((c & F_PROPER) && !(c & F_UNMAP)) out[2 * SAM_PAIR_GOOD + w] += 1;   /* :60 */
        if (c & F_READ1) out[2 * SAM_READ1 + w] += 1;                            /* :61 */
        if (c & F_READ2) out[2 * SAM_READ2 + w] += 1;                            /* :62 */
        if ((c & F_MUNMAP) && !(c & F_UNMAP)) out[2 * SAM_SGLTN + w] += 1;       /* :63 */
        if (!(c & F_UNMAP) && !(c & F_MUNMAP)) out[2 * SAM_PAIR_MAP + w] += 1;   /* :64-66 */
    }
    if (!(c & F_UNMAP)) out[2 * SAM_MAPPED + w] += 1;                            /* :68 */
    if (c & F_DUP) out[2 * SAM_DUP + w] += 1;                                    /* :69 */
}

void oracle_samtools_u16(const uint16_t* array, uint64_t n, uint64_t out[26])
{
    uint64_t* hist = (uint64_t*)calloc(65536, sizeof(uint64_t));
    hist_accumulate(array, n, hist);
    for (int v = 0; v < 65536; ++v) {
        if (!hist[v]) continue;
        uint64_t one[26];
        memset(one, 0, sizeof one);
        oracle_samtools_update((uint16_t)v, one);
        for (int s = 0; s < 26; ++s) out[s] += one[s] * hist[v];
    }
    free(hist);
}

void oracle_flagstat_hist_u16(const uint16_t* array, uint64_t n, uint64_t out[32])
{
    uint64_t* hist = (uint64_t*)calloc(65536, sizeof(uint64_t));
    hist_accumulate(array, n, hist);
    hist_to_counters(hist, out);
    free(hist);
}

struct shard_job {
    const uint16_t* a;
    uint64_t n;
    uint64_t out[32];
};

static void* shard_main(void* arg)
{
    struct shard_job* j = (struct shard_job*)arg;
    memset(j->out, 0, sizeof j->out);
    oracle_flagstat_hist_u16(j->a, j->n, j->out);
    return NULL;
}

void oracle_flagstat_mt_u16(const uint16_t* array, uint64_t n, int threads, uint64_t out[32])
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    struct shard_job* jobs = (struct shard_job*)calloc((size_t)threads, sizeof *jobs);
    pthread_t* tid = (pthread_t*)calloc((size_t)threads, sizeof *tid);
    const uint64_t per = n / (uint64_t)threads;
    for (int t = 0; t < threads; ++t) {
        jobs[t].a = array + per * (uint64_t)t;
        jobs[t].n = (t == threads - 1) ? n - per * (uint64_t)t : per;
        pthread_create(&tid[t], NULL, shard_main, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) {
        pthread_join(tid[t], NULL);
        for (int s = 0; s < 32; ++s) out[s] += jobs[t].out[s];
    }
    free(jobs);
    free(tid);
}

/* python/libalgebra.h:566-574 (scalar naive pospopcnt), through a 65536-bin histogram */
void oracle_pospopcnt_u16(const uint16_t* array, uint64_t n, uint64_t out[16])
{
    uint64_t* hist = (uint64_t*)calloc(65536, sizeof(uint64_t));
    hist_accumulate(array, n, hist);
    for (int v = 0; v < 65536; ++v)
        if (hist[v])
            for (int j = 0; j < 16; ++j)
                if (v & (1 << j)) out[j] += hist[v];
    free(hist);
}

/* ---- synthetic input makers (host twins) --------------------------------- */

static inline uint64_t mix64(uint64_t seed, uint64_t ctr)
{
    /* splitmix64 finaliser keyed by (seed, counter) */
    uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* NA12878-like categorical table: README.md:178-192 marginals over
 * N = 824,541,892 reads, SURVEY.md section 8(d) config 3. */
#define NA_TOTAL 824541892ull
static const uint64_t na_cum[6] = {
    781085884ull,                                   /* proper pairs            */
    781085884ull + 16865006ull,                     /* both mapped, not proper */
    781085884ull + 16865006ull + 2038885ull,        /* singleton (mate unmapped) */
    781085884ull + 16865006ull + 2038885ull + 2038885ull, /* unmapped, mate mapped */
    781085884ull + 16865006ull + 2038885ull + 2038885ull + 17119604ull, /* both unmapped */
    NA_TOTAL                                        /* supplementary           */
};
static const uint16_t na_vals[6][8] = {
    {99, 147, 83, 163, 99, 147, 83, 163},
    {65, 129, 97, 145, 81, 161, 113, 177},
    {73, 137, 89, 153, 73, 137, 89, 153},
    {69, 133, 101, 165, 69, 133, 101, 165},
    {77, 141, 77, 141, 77, 141, 77, 141},
    {2113, 2177, 2129, 2193, 2113, 2177, 2129, 2193},
};

static inline uint16_t na_flag(uint64_t seed, uint64_t i, uint32_t eps)
{
    const uint64_t h = mix64(seed, i);
    const uint64_t t = ((h >> 32) * NA_TOTAL) >> 32; /* uniform in [0, NA_TOTAL) */
    int c = 0;
    while (t >= na_cum[c]) ++c;
    uint16_t v = na_vals[c][h & 7];
    if (eps & 1) {
        if (((h >> 3) & 0x3FF) < 10) v |= F_DUP;     /* ~0.98 % */
        if (((h >> 13) & 0x3FF) < 1) v |= F_QCFAIL;  /* ~0.098 % */
    }
    return v;
}

void oracle_generate_u16(int kind, uint64_t seed, uint32_t mask,
                         uint64_t first_index, uint64_t n, uint16_t* out)
{
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t i = first_index + k;
        uint16_t v;
        switch (kind) {
        case ORACLE_GEN_UNIFORM:
            v = (uint16_t)((mix64(seed, i >> 2) >> (16 * (i & 3))) & mask);
            break;
        case ORACLE_GEN_NA12878:
            v = na_flag(seed, i, mask);
            break;
        default:
            v = (uint16_t)(i + seed);
            break;
        }
        out[k] = v;
    }
}

struct gen_job {
    int kind;
    uint64_t seed;
    uint32_t mask;
    uint64_t first, n;
    uint64_t out[32];
};

static void* gen_main(void* arg)
{
    struct gen_job* j = (struct gen_job*)arg;
    enum { CHUNK = 1 << 20 };
    uint16_t* buf = (uint16_t*)malloc(sizeof(uint16_t) * CHUNK);
    uint64_t* hist = (uint64_t*)calloc(65536, sizeof(uint64_t));
    memset(j->out, 0, sizeof j->out);
    for (uint64_t done = 0; done < j->n;) {
        uint64_t c = j->n - done;
        if (c > CHUNK) c = CHUNK;
        oracle_generate_u16(j->kind, j->seed, j->mask, j->first + done, c, buf);
        hist_accumulate(buf, c, hist);
        done += c;
    }
    hist_to_counters(hist, j->out);
    free(hist);
    free(buf);
    return NULL;
}

void oracle_flagstat_generated(int kind, uint64_t seed, uint32_t mask,
                               uint64_t first_index, uint64_t n, int threads,
                               uint64_t out[32])
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    struct gen_job* jobs = (struct gen_job*)calloc((size_t)threads, sizeof *jobs);
    pthread_t* tid = (pthread_t*)calloc((size_t)threads, sizeof *tid);
    const uint64_t per = n / (uint64_t)threads;
    for (int t = 0; t < threads; ++t) {
        jobs[t].kind = kind;
        jobs[t].seed = seed;
        jobs[t].mask = mask;
        jobs[t].first = first_index + per * (uint64_t)t;
        jobs[t].n = (t == threads - 1) ? n - per * (uint64_t)t : per;
        pthread_create(&tid[t], NULL, gen_main, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) {
        pthread_join(tid[t], NULL);
        for (int s = 0; s < 32; ++s) out[s] += jobs[t].out[s];
    }
    free(jobs);
    free(tid);
}
