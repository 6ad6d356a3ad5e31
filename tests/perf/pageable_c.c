/* A large host array of a C caller -> counters: hipMemcpyAsync out of the caller's pageable memory, which the HIP runtime pins as it
 * goes (knob staged_min_flags = 0: what FLAGSTATS_u16_x64 did for every size before r05) against the staged way
 * (FLAGSTATS_hip_host_staged_u16: worker threads copy into page-locked chunks) and against the product entry with its size rule, for
 * memory in 4 KiB pages (plain malloc) and in transparent huge pages (what numpy asks for), a NEW buffer for every call.
 *   pageable_c N_FLAGS [reps]     gcc -O2 pageable_c.c -I../../include -L../../libflagstats_amd -l:libflagstats_hip.so -Wl,-rpath,... */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

#include "libflagstats_hip.h"

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static uint16_t* fresh(const uint16_t* src, uint64_t n, int huge)
{
    const size_t len = (n * 2 + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    uint8_t* p = mmap(NULL, len + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) exit(3);
    p = (uint8_t*)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    madvise(p, len, huge ? MADV_HUGEPAGE : MADV_NOHUGEPAGE);
    memcpy(p, src, n * 2);   /* (never unmapped: every call sees an address the runtime has not seen) */
    return (uint16_t*)p;
}

int main(int argc, char** argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 0) : 100000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    uint16_t* base = malloc(n * 2);
    uint64_t x = 88172645463325252ull;
    for (uint64_t i = 0; i < n; ++i) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        base[i] = (uint16_t)(x >> 20);
    }
    if (FLAGSTATS_hip_init(0)) return 1;
    uint64_t want[32] = {0};
    if (FLAGSTATS_u16_x64(base, n, want)) return 1;
    printf("%11llu flags (%5.0f MiB), a new buffer per call, best of %d (median): ", (unsigned long long)n, n * 2 / 1048576.0, reps);
    for (int huge = 0; huge < 2; ++huge)
        for (int how = 0; how < 3; ++how) {   /* 0 the runtime's copy, 1 staged with 8 threads, 2 FLAGSTATS_u16_x64 with the shipped rule */
            double t[64];
            for (int r = 0; r < reps && r < 64; ++r) {
                uint16_t* a = fresh(base, n, huge);
                uint64_t out[32] = {0};
                FLAGSTATS_hip_set("staged_min_flags", how == 0 ? 0 : 1ull << 27);
                const double t0 = now_ms();
                const int rc = how == 1 ? FLAGSTATS_hip_host_staged_u16(a, n, 8, out, NULL) : FLAGSTATS_u16_x64(a, n, out);
                t[r] = now_ms() - t0;
                if (rc || memcmp(out, want, sizeof want)) return fprintf(stderr, "wrong counters (%d)\n", rc), 2;
            }
            for (int i = 0; i < reps; ++i)
                for (int j = i + 1; j < reps; ++j)
                    if (t[j] < t[i]) {
                        const double s = t[i];
                        t[i] = t[j], t[j] = s;
                    }
            printf("%s %s %.2f ms (%.2f) = %.1f GB/s | ", huge ? "huge pages" : "4 KiB pages", how == 0 ? "runtime copy" : (how == 1 ? "staged/8" : "FLAGSTATS_u16_x64"), t[0], t[reps / 2],
                   n * 2 / t[0] / 1e6);
        }
    printf("\n");
    return 0;
}
