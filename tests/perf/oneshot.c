/* One-shot consumer for tests/perf/cold_start.py: a FRESH process that does what a user of the reference's one-shot
 * programs does (`bench decompress -i file -d`, benchmark/flagstats.cpp:288-342, is such a program: start, read, count,
 * print, exit) and stamps every phase against CLOCK_MONOTONIC, which is system-wide -- the parent passes the time at
 * which it spawned this process, so "since spawn" includes exec, the dynamic loader and the HIP runtime's constructors.
 *   oneshot SPAWN_NS u16 N          FLAGSTATS_u16 on N generated flags (N < 2^32)
 *   oneshot SPAWN_NS blockfile PATH FLAGSTATS_hip_blockfile (codec by extension)
 *   oneshot SPAWN_NS raw PATH       FLAGSTATS_hip_file_raw
 * optional 4th argument: how many times to repeat the call in the same process (the later calls are the warm ones).
 * env ONESHOT_EXIT=shutdown: FLAGSTATS_hip_shutdown() before main returns (timed, on stderr); =fast: _exit(0) after the output is
 * flushed -- no atexit handlers, neither the HIP runtime's nor anybody's; unset: return from main.  The parent sees when the
 * process is gone, so "return from main -> process gone" is what the runtime's and the driver's teardown cost.
 * env ONESHOT_FRESH_BUFFER=1 (u16 mode): every call gets its own newly allocated copy of the flags -- the HIP runtime pins a pageable
 * buffer the first time it copies out of it and remembers that, so repeats on ONE buffer are not what a caller with a new buffer per call sees.
 * env ONESHOT_LAZY_INIT=1: no FLAGSTATS_hip_init first -- the first call creates the engine itself, as a caller that only
 * knows the reference's API would have it (the library then opens the file and asks for readahead BEFORE the 90 ms of
 * runtime initialisation: what a cold page cache gains from).
 * Links libflagstats_hip.so directly: gcc -O2 oneshot.c -I../../include -L../../libflagstats_amd -lflagstats_hip -Wl,-rpath,... */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "libflagstats_hip.h"

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char** argv)
{
    const double t_main = now_ms();
    if (argc < 4) return fprintf(stderr, "usage: oneshot SPAWN_NS u16|blockfile|raw ARG [calls]\n"), 2;
    const double t_spawn = strtoull(argv[1], NULL, 10) * 1e-6;
    const char* mode = argv[2];
    const int calls = argc > 4 ? atoi(argv[4]) : 1;
    uint16_t* flags = NULL;
    uint64_t n = 0;
    if (!strcmp(mode, "u16")) {
        n = strtoull(argv[3], NULL, 0);
        flags = (uint16_t*)malloc(n * 2 + 2);
        uint64_t x = 88172645463325252ull;
        for (uint64_t i = 0; i < n; ++i) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            flags[i] = (uint16_t)(x >> 20);
        }
    }
    const double t_input = now_ms();
    FLAGSTATS_hip_set("on_error", 0);
    const char* lazy = getenv("ONESHOT_LAZY_INIT");
    int rc = (lazy && atoi(lazy)) ? 0 : FLAGSTATS_hip_init(0); /* hipInit, device selection, the default engine (streams, staging, counters) */
    const double t_init = now_ms();
    if (rc) return fprintf(stderr, "init failed: %s\n", FLAGSTATS_hip_last_error()), 1;
    printf("{\"mode\": \"%s\", \"since_spawn_at_main_ms\": %.3f, \"input_ms\": %.3f, \"init_ms\": %.3f, \"calls_ms\": [", mode, t_main - t_spawn, t_input - t_main,
           t_init - t_input);
    uint64_t out[32];
    FLAGSTATS_blockfile_stats st;
    memset(&st, 0, sizeof st);
    double t_first_done = 0;
    for (int c = 0; c < calls; ++c) {
        double t0 = now_ms();
        memset(out, 0, sizeof out);
        if (flags) {
            uint32_t o32[32] = {0};
            const char* fresh = getenv("ONESHOT_FRESH_BUFFER");
            if (fresh && atoi(fresh) && c > 0) {
                uint16_t* copy = (uint16_t*)malloc(n * 2 + 2);   /* (never freed: the next call must not get the same pages back) */
                memcpy(copy, flags, n * 2);
                flags = copy;
            }
            t0 = now_ms();   /* (the copy above is the caller's business, not the call's) */
            rc = (int)FLAGSTATS_u16(flags, (uint32_t)n, o32);
            for (int k = 0; k < 32; ++k) out[k] = o32[k];
        } else if (!strcmp(mode, "raw")) {
            rc = FLAGSTATS_hip_file_raw(argv[3], out, &st);
        } else {
            rc = FLAGSTATS_hip_blockfile(argv[3], 0, out, &st);
        }
        const double t1 = now_ms();
        if (rc) return fprintf(stderr, "call failed: %s\n", FLAGSTATS_hip_last_error()), 1;
        if (c == 0) t_first_done = t1;
        printf("%s%.3f", c ? ", " : "", t1 - t0);
    }
    unsigned long long sum = 0;
    for (int k = 0; k < 32; ++k) sum += out[k] * (unsigned long long)(k + 1);
    printf("], \"counters_ready_since_spawn_ms\": %.3f, \"n_flags\": %llu, \"gpu_decode\": %d, \"checksum\": %llu}\n", t_first_done - t_spawn,
           (unsigned long long)(flags ? n : st.n_flags), (int)st.gpu_decode, sum);
    const char* how = getenv("ONESHOT_EXIT");
    if (how && !strcmp(how, "shutdown")) {
        const double t0 = now_ms();
        FLAGSTATS_hip_shutdown();
        fprintf(stderr, "oneshot: FLAGSTATS_hip_shutdown %.3f ms\n", now_ms() - t0);
    }
    printf("{\"main_returns_since_spawn_ms\": %.3f}\n", now_ms() - t_spawn);
    fflush(stdout);
    fflush(stderr);
    if (how && !strcmp(how, "fast")) _exit(0);
    return 0;
}
