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
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

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

/* All-core CPU baseline with reproducible placement: `threads` workers, worker k pinned to the k-th CPU
 * this process may run on, each allocating, first-touching and filling its OWN shard of
 * `per_thread_flags` uniform-random uint16 (so the pages are local to the socket that reads them),
 * then `reps` passes of the dispatcher's kernel between two barriers.  Returns the seconds between the
 * barriers; out[32] += the counters of one pass over all shards. */
double ref_dispatch_mt_bench(uint64_t per_thread_flags, int threads, uint32_t reps, uint64_t seed, uint64_t out[32])
{
    if (threads < 1 || reps < 1) return -1.0;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    std::vector<int> cpus;
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
    std::atomic<int> arrived{0};
    std::atomic<int> phase{0};
    std::vector<std::vector<uint64_t>> part(threads, std::vector<uint64_t>(32, 0));
    std::vector<double> t_begin(threads, 0.0), t_end(threads, 0.0);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto barrier = [&](int target_phase) {
        if (arrived.fetch_add(1) + 1 == threads * target_phase) phase.store(target_phase);
        while (phase.load() < target_phase) std::this_thread::yield();
    };
    auto work = [&](int k) {
        if (!cpus.empty()) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[k % cpus.size()], &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
        }
        uint16_t* a = static_cast<uint16_t*>(std::malloc(per_thread_flags * 2 + 64));
        uint64_t x = seed * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(k) * 0xD1B54A32D192ED03ull + 1;
        for (uint64_t i = 0; i < per_thread_flags; ++i) {  /* xorshift64*: uniform over all 16 bits */
            x ^= x >> 12;
            x ^= x << 25;
            x ^= x >> 27;
            a[i] = static_cast<uint16_t>((x * 0x2545F4914F6CDD1Dull) >> 48);
        }
        uint64_t warm[32] = {0};
        ref_dispatch_x64(a, per_thread_flags, warm);
        barrier(1);
        t_begin[k] = now();
        uint64_t acc[32] = {0};
        for (uint32_t r = 0; r < reps; ++r) ref_dispatch_x64(a, per_thread_flags, acc);
        t_end[k] = now();
        barrier(2);
        for (int i = 0; i < 32; ++i) part[k][i] = acc[i] / reps;
        std::free(a);
    };
    std::vector<std::thread> pool;   /* every worker is a fresh thread: the caller's own affinity stays as it was */
    for (int k = 0; k < threads; ++k) pool.emplace_back(work, k);
    for (auto& t : pool) t.join();
    double b = t_begin[0], e = t_end[0];
    for (int k = 1; k < threads; ++k) {
        if (t_begin[k] < b) b = t_begin[k];
        if (t_end[k] > e) e = t_end[k];
    }
    for (int k = 0; k < threads; ++k)
        for (int i = 0; i < 32; ++i) out[i] += part[k][i];
    return e - b;
}

/* All cores over DRAM-resident data (contiguous shards of the workload): `threads` workers,
 * worker k pinned to the k-th CPU this process may run on, reads the k-th contiguous shard of the CALLER's array (which the
 * caller generated with the same pinning, so a shard's pages lie where its reader runs), ONE pass of the dispatcher's kernel
 * per round, a barrier before each round; seconds[r] = first start .. last end of round r.  A shard of 256 MiB and more is
 * far beyond the caches: every round reads DRAM.  out[32] += the counters of one pass.  Returns 0. */
int ref_dispatch_mt_shards(const uint16_t* a, uint64_t n, int threads, int rounds, double* seconds, uint64_t out[32])
{
    if (threads < 1 || rounds < 1 || !seconds) return -1;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    std::vector<int> cpus;
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
    std::atomic<int> arrived{0};
    std::atomic<int> phase{0};
    std::vector<std::vector<uint64_t>> part(threads, std::vector<uint64_t>(32, 0));
    std::vector<std::vector<double>> t_begin(rounds, std::vector<double>(threads, 0.0)), t_end(rounds, std::vector<double>(threads, 0.0));
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto barrier = [&](int target_phase) {
        if (arrived.fetch_add(1) + 1 == threads * target_phase) phase.store(target_phase);
        while (phase.load() < target_phase) std::this_thread::yield();
    };
    auto work = [&](int k) {
        if (!cpus.empty()) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[k % cpus.size()], &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
        }
        const uint64_t per = n / static_cast<uint64_t>(threads);
        const uint64_t b = per * static_cast<uint64_t>(k), e = (k == threads - 1) ? n : b + per;
        for (int r = 0; r < rounds; ++r) {
            uint64_t acc[32] = {0};
            barrier(r + 1);
            t_begin[r][k] = now();
            ref_dispatch_x64(a + b, e - b, acc);
            t_end[r][k] = now();
            if (r == 0)
                for (int i = 0; i < 32; ++i) part[k][i] = acc[i];
        }
    };
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; ++k) pool.emplace_back(work, k);
    for (auto& t : pool) t.join();
    for (int r = 0; r < rounds; ++r) {
        double b = t_begin[r][0], e = t_end[r][0];
        for (int k = 1; k < threads; ++k) {
            if (t_begin[r][k] < b) b = t_begin[r][k];
            if (t_end[r][k] > e) e = t_end[r][k];
        }
        seconds[r] = e - b;
    }
    for (int k = 0; k < threads; ++k)
        for (int i = 0; i < 32; ++i) out[i] += part[k][i];
    return 0;
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
