/* A consumer written against the reference's public API only (what
 * benchmark/flagstats.cpp:304,328-329 does per block), compiled against the header
 * shim include/libflagstats.h and linked with libflagstats_hip.so. */
#include <stdio.h>
#include <stdlib.h>

#include "libflagstats.h"

int main(int argc, char** argv)
{
    uint32_t n = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 1000;
    uint16_t* flags_in = (uint16_t*)malloc(sizeof(uint16_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; ++i) flags_in[i] = (uint16_t)(i * 2654435761u >> 16);
    uint32_t counters[32] = {0};
    FLAGSTATS_func func = FLAGSTATS_get_function(n);
    int rc = (*func)(flags_in, n, counters);
    uint64_t rc2 = FLAGSTATS_u16(flags_in, n, counters);
    printf("rc=%d rc2=%llu unmapped=%u qcfail=%u dup=%u\n", rc, (unsigned long long)rc2,
           counters[FLAGSTAT_FUNMAP_OFF], counters[16 + FLAGSTAT_FQCFAIL_OFF], counters[FLAGSTAT_FDUP_OFF]);
    free(flags_in);
    return (rc == 0 && rc2 == 0) ? 0 : 3;
}
