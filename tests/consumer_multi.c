/* A plain C consumer of the multi-GPU entry points (no HIP, no torch): what a C host program like
 * the reference's bench (benchmark/flagstats.cpp) would call to use more than one GPU.
 *   consumer_multi <n> <ndev> [dev0 dev1 ...]
 * Prints three lines of 32 counters: the single-engine result, the multi-device result, and the sum
 * of two explicit contexts driven concurrently from two threads. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libflagstats_hip.h"

struct job {
    FLAGSTATS_hip_ctx* ctx;
    const uint16_t* a;
    uint64_t n;
    uint64_t out[32];
    int rc;
};

static void* run(void* p)
{
    struct job* j = (struct job*)p;
    j->rc = 0;
    for (int rep = 0; rep < 3 && !j->rc; ++rep) {   /* several calls per context, concurrently with the other */
        memset(j->out, 0, sizeof j->out);
        j->rc = FLAGSTATS_hip_ctx_u16_x64(j->ctx, j->a, j->n, j->out);
    }
    return 0;
}

static void print32(const char* tag, const uint64_t* c)
{
    printf("%s", tag);
    for (int i = 0; i < 32; ++i) printf(" %llu", (unsigned long long)c[i]);
    printf("\n");
}

int main(int argc, char** argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000;
    int ndev = argc > 2 ? atoi(argv[2]) : 2;
    int devices[64];
    for (int i = 0; i < ndev && i < 64; ++i) devices[i] = argc > 3 + i ? atoi(argv[3 + i]) : 0;
    uint16_t* a = (uint16_t*)malloc(sizeof(uint16_t) * (n ? n : 1));
    for (uint64_t i = 0; i < n; ++i) a[i] = (uint16_t)(((uint32_t)i * 2654435761u) >> 13);

    uint64_t one[32] = {0}, multi[32] = {0}, two[32] = {0};
    if (FLAGSTATS_u16_x64(a, n, one)) return 3;
    if (FLAGSTATS_hip_multi_u16_x64(a, n, devices, ndev, multi)) return 4;

    struct job j[2];
    uint64_t b0, e0, b1, e1;
    FLAGSTATS_hip_shard_range(n, 0, 2, &b0, &e0);
    FLAGSTATS_hip_shard_range(n, 1, 2, &b1, &e1);
    j[0].ctx = FLAGSTATS_hip_ctx_create(devices[0]);
    j[1].ctx = FLAGSTATS_hip_ctx_create(devices[ndev > 1 ? 1 : 0]);
    if (!j[0].ctx || !j[1].ctx) return 5;
    j[0].a = a + b0; j[0].n = e0 - b0;
    j[1].a = a + b1; j[1].n = e1 - b1;
    pthread_t t[2];
    pthread_create(&t[0], 0, run, &j[0]);
    pthread_create(&t[1], 0, run, &j[1]);
    pthread_join(t[0], 0);
    pthread_join(t[1], 0);
    if (j[0].rc || j[1].rc) return 6;
    for (int i = 0; i < 32; ++i) two[i] = j[0].out[i] + j[1].out[i];
    FLAGSTATS_hip_ctx_destroy(j[0].ctx);
    FLAGSTATS_hip_ctx_destroy(j[1].ctx);

    print32("one", one);
    print32("multi", multi);
    print32("two", two);
    FLAGSTATS_hip_shutdown();
    free(a);
    return 0;
}
