// TEST-ONLY driver for the host-stub build (make -C libflagstats_amd/csrc hoststub SAN=address,undefined): fork() at the
// C boundary.  The reference's entry point is a pure function and works in a forked child (libflagstats.h:3024-3070); the
// replacement is an engine, and a child forked after its first use must be REFUSED by every entry point -- loudly, at once,
// without touching a mutex the parent's (now non-existent) threads may hold -- while the parent goes on counting correctly.
//   1. a child forked BEFORE the library's first call owns it: it counts, correctly;
//   2. the parent counts, opens a session and a context, and keeps a thread inside the engine (its lock is held nearly all
//      the time), then forks: the child calls every entry family; each must return an error naming the fork within the
//      alarm, the release-type entries must do nothing, FLAGSTATS_hip_forked() says 1;
//   3. a second child with the default "on_error" policy: the reference-shaped entry aborts (SIGABRT), as documented;
//   4. the parent is unaffected: its counters, its session and its context still work.
#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/libflagstats_hip_probe.h"
extern "C" {
#include "../../oracle/flagstat_oracle.h"
}

static int g_fail = 0;
#define CHECK(cond, ...)                                              \
    do {                                                              \
        if (!(cond)) {                                                \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);                        \
            std::fprintf(stderr, "\n");                               \
            ++g_fail;                                                 \
        }                                                             \
    } while (0)

static bool same(const uint64_t* a, const uint64_t* b) { return std::memcmp(a, b, 32 * sizeof(uint64_t)) == 0; }
static bool names_fork() { return std::strstr(FLAGSTATS_hip_last_error(), "fork()ed") != nullptr; }

// every entry family, in a refused child; returns the number of entries that did not refuse properly
static int child_calls_everything(const std::vector<uint16_t>& flags, FLAGSTATS_hip_stream* inherited_session, FLAGSTATS_hip_ctx* inherited_ctx,
                                  void* inherited_device_ptr)
{
    int bad = 0;
    uint64_t out[32] = {0}, zero[32] = {0};
    uint32_t out32[32] = {0};
    FLAGSTATS_blockfile_stats st;
#define REFUSED(expr)                                                                             \
    do {                                                                                          \
        const long long rc_ = static_cast<long long>(expr);                                       \
        if (rc_ == 0 || !names_fork()) {                                                          \
            std::fprintf(stderr, "child: %s was not refused (rc %lld, text '%s')\n", #expr, rc_, FLAGSTATS_hip_last_error()); \
            ++bad;                                                                                \
        }                                                                                         \
    } while (0)
#define REFUSED_NULL(expr)                                                                        \
    do {                                                                                          \
        const void* p_ = (expr);                                                                  \
        if (p_ != nullptr || !names_fork()) {                                                     \
            std::fprintf(stderr, "child: %s was not refused (text '%s')\n", #expr, FLAGSTATS_hip_last_error()); \
            ++bad;                                                                                \
        }                                                                                         \
    } while (0)
    if (FLAGSTATS_hip_forked() != 1) {
        std::fprintf(stderr, "child: FLAGSTATS_hip_forked() is not 1\n");
        ++bad;
    }
    // reference-shaped entries ("on_error" was set to `return` by the parent before the fork)
    REFUSED(FLAGSTATS_u16(flags.data(), 1000, out32));
    REFUSED(FLAGSTAT_hip(flags.data(), 1000, out32));
    REFUSED(FLAGSTATS_get_function(1000)(flags.data(), 1000, out32));
    REFUSED(STORM_pospopcnt_u16(flags.data(), 1000, out32));
    // 64-bit host entries, small (polling path) and multi-chunk
    REFUSED(FLAGSTATS_u16_x64(flags.data(), 1000, out));
    REFUSED(FLAGSTATS_u16_x64(flags.data(), flags.size(), out));
    REFUSED(FLAGSTATS_u16_x64_superset(flags.data(), flags.size(), out));
    REFUSED(FLAGSTATS_hip_pospopcnt_u16_x64(flags.data(), 1000, out));
    REFUSED(FLAGSTATS_hip_host_staged_u16(flags.data(), flags.size(), 2, out, &st));
    // device-pointer entries, allocators, copies
    REFUSED(FLAGSTATS_hip_device_u16(static_cast<const uint16_t*>(inherited_device_ptr), 100, static_cast<uint64_t*>(inherited_device_ptr), nullptr));
    REFUSED(FLAGSTATS_hip_device_u16_store(static_cast<const uint16_t*>(inherited_device_ptr), 100, static_cast<uint64_t*>(inherited_device_ptr), nullptr));
    REFUSED(FLAGSTATS_hip_device_u16_sync(static_cast<const uint16_t*>(inherited_device_ptr), 100, out));
    REFUSED(FLAGSTATS_hip_device_u16_superset_sync(static_cast<const uint16_t*>(inherited_device_ptr), 100, out));
    REFUSED(FLAGSTATS_hip_device_pospopcnt_u16(static_cast<const uint16_t*>(inherited_device_ptr), 100, static_cast<uint64_t*>(inherited_device_ptr), nullptr));
    REFUSED_NULL(FLAGSTATS_hip_device_alloc(4096));
    REFUSED_NULL(FLAGSTATS_hip_device_alloc_on(0, 4096));
    REFUSED_NULL(FLAGSTATS_hip_host_alloc(4096));
    REFUSED(FLAGSTATS_hip_memcpy_h2d(inherited_device_ptr, flags.data(), 64));
    REFUSED(FLAGSTATS_hip_memcpy_d2h(out, inherited_device_ptr, 64));
    REFUSED(FLAGSTATS_hip_synchronize());
    REFUSED(FLAGSTATS_hip_generate_u16(static_cast<uint16_t*>(inherited_device_ptr), 100, 0, 1, 0xFFFF, 0, nullptr));
    // context / knobs
    REFUSED(FLAGSTATS_hip_init(0));
    REFUSED(FLAGSTATS_hip_set("poll", 0));
    REFUSED(FLAGSTATS_hip_compute_units());
    if (FLAGSTATS_hip_available() != 0 || FLAGSTATS_hip_device_count() != 0 || FLAGSTATS_hip_device_id() != -1 || FLAGSTATS_hip_get("grid") != 0) {
        std::fprintf(stderr, "child: available / device_count / device_id / get must say 'nothing here'\n");
        ++bad;
    }
    REFUSED_NULL(FLAGSTATS_hip_ctx_create(0));
    REFUSED(FLAGSTATS_hip_ctx_u16_x64(inherited_ctx, flags.data(), 1000, out));
    REFUSED(FLAGSTATS_hip_ctx_device_u16_sync(inherited_ctx, static_cast<const uint16_t*>(inherited_device_ptr), 100, out));
    if (FLAGSTATS_hip_ctx_device(inherited_ctx) != -1) ++bad;
    // sessions: a new one, and the one inherited from the parent
    REFUSED_NULL(FLAGSTATS_hip_stream_open());
    REFUSED_NULL(FLAGSTATS_hip_stream_acquire(inherited_session, 100));
    REFUSED(FLAGSTATS_hip_stream_commit(inherited_session, 0));
    REFUSED(FLAGSTATS_hip_stream_push(inherited_session, flags.data(), 1000));
    REFUSED(FLAGSTATS_hip_stream_finish(inherited_session, out));
    // block files and raw files
    REFUSED(FLAGSTATS_hip_blockimage_lz4(flags.data(), 64, 1, out, &st));
    REFUSED(FLAGSTATS_hip_blockimage_zstd(flags.data(), 64, 1, out, &st));
    REFUSED(FLAGSTATS_hip_blockfile("/nonexistent.lz4", 1, out, &st));
    REFUSED(FLAGSTATS_hip_blockfile_lz4("/nonexistent.lz4", 1, out, &st));
    REFUSED(FLAGSTATS_hip_blockfile_superset("/nonexistent.zst", 1, out, &st));
    REFUSED(FLAGSTATS_hip_file_raw("/nonexistent.bin", out, &st));
    REFUSED(FLAGSTATS_hip_blockimage_lz4_gpu(flags.data(), 64, out, nullptr));
    // multi-device and collectives
    REFUSED(FLAGSTATS_hip_multi_u16_x64(flags.data(), flags.size(), nullptr, 2, out));
    {
        const uint16_t* arrs[1] = {static_cast<const uint16_t*>(inherited_device_ptr)};
        const uint64_t ns[1] = {100};
        REFUSED(FLAGSTATS_hip_multi_device_u16(arrs, ns, 1, out));
    }
    {
        char id[128];
        REFUSED(FLAGSTATS_hip_comm_unique_id(id));
        REFUSED_NULL(FLAGSTATS_hip_comm_init_rank(id, 1, 0, 0));
        REFUSED(FLAGSTATS_hip_allreduce_counters(static_cast<uint64_t*>(inherited_device_ptr), id, nullptr));
        REFUSED(FLAGSTATS_hip_stream_wait_stream(nullptr, nullptr, 0));
    }
    // measurement entries
    {
        float ms = 0;
        double mhz = 0;
        REFUSED(FLAGSTATS_hip_time_device_u16(static_cast<const uint16_t*>(inherited_device_ptr), 100, 0, 1, &ms, out));
        REFUSED(FLAGSTATS_hip_read_probe(inherited_device_ptr, 4096, 1, 0, 1, &ms));
        REFUSED(FLAGSTATS_hip_sclk_under_load(static_cast<const uint16_t*>(inherited_device_ptr), 100, 1, &mhz));
    }
    // nothing was counted
    if (!same(out, zero)) {
        std::fprintf(stderr, "child: a refused call changed the caller's counters\n");
        ++bad;
    }
    for (uint32_t v : out32)
        if (v) {
            std::fprintf(stderr, "child: a refused reference-shaped call changed the caller's counters\n");
            ++bad;
            break;
        }
    // release-type entries: silent no-ops (a child's interpreter may run them on inherited objects at exit)
    FLAGSTATS_hip_stream_close(inherited_session);
    FLAGSTATS_hip_ctx_destroy(inherited_ctx);
    FLAGSTATS_hip_device_free(inherited_device_ptr);
    FLAGSTATS_hip_host_free(nullptr);
    FLAGSTATS_hip_shutdown();
    // the stateless host helpers keep working (no GPU state behind them)
    {
        uint64_t b = 0, e = 0;
        FLAGSTATS_hip_shard_range(100, 1, 4, &b, &e);
        uint16_t v[4];
        if (b != 25 || e != 50 || FLAGSTATS_text_to_u16("99\n1024\n", 8, v, 4) != 2 || v[0] != 99 || v[1] != 1024) {
            std::fprintf(stderr, "child: the stateless helpers must keep working\n");
            ++bad;
        }
    }
    return bad;
}

static int wait_child(pid_t pid, int* signalled)
{
    int status = 0;
    *signalled = 0;
    if (waitpid(pid, &status, 0) != pid) return -1;
    if (WIFSIGNALED(status)) {
        *signalled = WTERMSIG(status);
        return -1;
    }
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

int main()
{
    const size_t n = 3000017;
    std::vector<uint16_t> flags(n);
    oracle_generate_u16(ORACLE_GEN_NA12878, 11, 1, 0, n, flags.data());
    uint64_t want[32] = {0};
    oracle_flagstat_u16(flags.data(), n, want);
    int sig = 0;

    // 1. forked before the first call: the child owns the library
    std::fflush(nullptr);
    pid_t pid = fork();
    if (pid == 0) {
        alarm(120);
        uint64_t out[32] = {0};
        const int rc = FLAGSTATS_u16_x64(flags.data(), n, out);
        _exit((rc == 0 && same(out, want) && FLAGSTATS_hip_forked() == 0) ? 0 : 1);
    }
    CHECK(wait_child(pid, &sig) == 0, "a child forked before the first call must count correctly (signal %d)", sig);

    // 2. the parent uses the library, keeps a thread inside the engine, forks
    uint64_t got[32] = {0};
    CHECK(FLAGSTATS_hip_forked() == 0, "parent: forked() before the first call");
    CHECK(FLAGSTATS_u16_x64(flags.data(), n, got) == 0 && same(got, want), "parent, first count: %s", FLAGSTATS_hip_last_error());
    FLAGSTATS_hip_stream* session = FLAGSTATS_hip_stream_open();
    FLAGSTATS_hip_ctx* ctx = FLAGSTATS_hip_ctx_create(0);
    void* dptr = FLAGSTATS_hip_device_alloc(4096);
    CHECK(session && ctx && dptr, "parent: session / context / device memory: %s", FLAGSTATS_hip_last_error());
    CHECK(FLAGSTATS_hip_stream_push(session, flags.data(), 500000) == 0, "parent: session push");
    CHECK(FLAGSTATS_hip_set("on_error", 0) == 0, "on_error 0");
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> worker_calls{0}, worker_bad{0};
    std::thread worker([&] {
        while (!stop.load()) {
            uint64_t o[32] = {0};
            if (FLAGSTATS_u16_x64(flags.data(), n, o) != 0 || !same(o, want)) ++worker_bad;
            ++worker_calls;
        }
    });
    while (worker_calls.load() < 2) usleep(1000);
    for (int round = 0; round < 8; ++round) {   // (several forks: some of them land while the worker holds the engine's lock)
        std::fflush(nullptr);
        pid = fork();
        if (pid == 0) {
            alarm(60);   // a child that blocks on an inherited mutex dies here, by SIGALRM
            _exit(child_calls_everything(flags, session, ctx, dptr) ? 1 : 0);
        }
        CHECK(wait_child(pid, &sig) == 0, "round %d: the forked child must be refused by every entry, quickly (signal %d)", round, sig);
        usleep(3000);
    }

    // 3. default policy: the reference-shaped entry aborts in the child
    CHECK(FLAGSTATS_hip_set("on_error", 1) == 0, "on_error 1");
    std::fflush(nullptr);
    pid = fork();
    if (pid == 0) {
        alarm(60);
        uint32_t o[32] = {0};
        (void)FLAGSTATS_u16(flags.data(), 1000, o);
        _exit(0);   // not reached
    }
    CHECK(wait_child(pid, &sig) == -1 && sig == SIGABRT, "the reference-shaped entry must abort in a forked child (signal %d)", sig);
    CHECK(FLAGSTATS_hip_set("on_error", 0) == 0, "on_error 0");

    // 4. the parent is unaffected
    stop = true;
    worker.join();
    CHECK(worker_bad.load() == 0 && worker_calls.load() >= 2, "parent's worker thread: %llu bad of %llu calls",
          static_cast<unsigned long long>(worker_bad.load()), static_cast<unsigned long long>(worker_calls.load()));
    CHECK(FLAGSTATS_hip_forked() == 0, "parent: forked() after the children");
    std::memset(got, 0, sizeof got);
    CHECK(FLAGSTATS_u16_x64(flags.data(), n, got) == 0 && same(got, want), "parent, count after the forks: %s", FLAGSTATS_hip_last_error());
    {
        uint64_t s_out[32] = {0}, s_want[32] = {0};
        oracle_flagstat_u16(flags.data(), 500000, s_want);
        CHECK(FLAGSTATS_hip_stream_finish(session, s_out) == 0 && same(s_out, s_want), "parent: the session opened before the forks: %s", FLAGSTATS_hip_last_error());
        std::memset(got, 0, sizeof got);
        CHECK(FLAGSTATS_hip_ctx_u16_x64(ctx, flags.data(), n, got) == 0 && same(got, want), "parent: the context made before the forks");
    }
    FLAGSTATS_hip_stream_close(session);
    FLAGSTATS_hip_ctx_destroy(ctx);
    FLAGSTATS_hip_device_free(dptr);
    FLAGSTATS_hip_shutdown();
    std::printf(g_fail ? "fork_driver: %d FAILED checks\n" : "fork_driver: all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
