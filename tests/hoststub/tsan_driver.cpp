// TEST-ONLY driver for the host-stub build (make -C libflagstats_amd/csrc hoststub): the product's host code linked against
// tests/hoststub/hip_stub.cpp, run under ThreadSanitizer.  Every result is checked against the oracle.
//   tsan_driver <dir with the reference-written golden block files>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../../include/libflagstats_hip_probe.h"
extern "C" {
#include "../../oracle/flagstat_oracle.h"
}

static int g_fail = 0;
#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);    \
            std::fprintf(stderr, "\n");           \
            ++g_fail;                             \
        }                                         \
    } while (0)

static bool same(const uint64_t* a, const uint64_t* b) { return std::memcmp(a, b, 32 * sizeof(uint64_t)) == 0; }

// a valid LZ4 block that stores `n` bytes as one run of literals (token, length bytes, literals; no match)
static void lz4_store(const unsigned char* src, size_t n, std::vector<unsigned char>& out)
{
    out.push_back(static_cast<unsigned char>((n < 15 ? n : 15) << 4));
    if (n >= 15) {
        size_t rest = n - 15;
        while (rest >= 255) {
            out.push_back(255);
            rest -= 255;
        }
        out.push_back(static_cast<unsigned char>(rest));
    }
    out.insert(out.end(), src, src + n);
}

// block file image: int32 uncompressed_size, int32 compressed_size, <LZ4 block>, ...  (benchmark/flagstats.cpp:119-138)
static std::vector<unsigned char> make_image(const std::vector<uint16_t>& flags, const std::vector<size_t>& block_flags)
{
    std::vector<unsigned char> img;
    size_t pos = 0;
    for (size_t b = 0; pos < flags.size(); ++b) {
        size_t m = block_flags[b % block_flags.size()];
        if (m > flags.size() - pos) m = flags.size() - pos;
        std::vector<unsigned char> blk;
        lz4_store(reinterpret_cast<const unsigned char*>(flags.data() + pos), m * 2, blk);
        const int32_t usz = static_cast<int32_t>(m * 2), csz = static_cast<int32_t>(blk.size());
        img.insert(img.end(), reinterpret_cast<const unsigned char*>(&usz), reinterpret_cast<const unsigned char*>(&usz) + 4);
        img.insert(img.end(), reinterpret_cast<const unsigned char*>(&csz), reinterpret_cast<const unsigned char*>(&csz) + 4);
        img.insert(img.end(), blk.begin(), blk.end());
        pos += m;
    }
    return img;
}

static void block_pipeline(const std::vector<uint16_t>& flags, const uint64_t* want)
{
    const std::vector<unsigned char> img = make_image(flags, {51200, 512000, 7, 300001, 123456});
    // The decoders are threads of the ENGINE's worker pool (made on first need, parked between calls): the counts go up and
    // down, so a job runs on fewer threads than the pool holds, then on more than it has (the pool grows), call after call --
    // every job must run on exactly the number asked for (a parked thread that joined in, or one that missed its job, shows as
    // a wrong count or a hang).
    for (int threads : {1, 5, 20, 2, 20, 1, 9}) {
        uint64_t out[32] = {0};
        FLAGSTATS_blockfile_stats st;
        const int rc = FLAGSTATS_hip_blockimage_lz4(img.data(), img.size(), threads, out, &st);
        CHECK(rc == 0 && same(out, want), "block image, %d decoder threads: rc %d (%s)", threads, rc, FLAGSTATS_hip_last_error());
        CHECK(st.n_flags == flags.size(), "block image: %llu flags seen", static_cast<unsigned long long>(st.n_flags));
        const int expect = static_cast<uint64_t>(threads) < st.n_blocks ? threads : static_cast<int>(st.n_blocks);   // (never more than blocks)
        CHECK(st.threads == expect, "block image: %d decoder threads asked for, %d expected, %d ran", threads, expect, st.threads);
    }
}

// the GPU LZ4 decoder's host side (flagstat_gpu_decode.hip): pieces on four decode streams with a count behind each, the
// reader pool of file mode with its pinned spans recycled by events, segments, the kept buffers and their release --
// image and file mode, 1 / 5 / 16 readers, a truncated file, a damaged block.  The stand-in kernel is the product's
// host decoder.
static void gpu_decoder_host_side(const std::vector<uint16_t>& flags, const uint64_t* want, const std::string& dir)
{
    const std::vector<unsigned char> img = make_image(flags, {51200, 512000, 7, 300001, 123456});
    CHECK(FLAGSTATS_hip_set("lz4_decoder", 1) == 0, "lz4_decoder 1");
    FLAGSTATS_hip_set("chunk_flags", 8);  // (spans of the floor size, 4 MiB: several per file)
    for (const char* pieces : {"", "1", "7"}) {
        if (*pieces)
            setenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS", pieces, 1);
        else
            unsetenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS");
        uint64_t out[32] = {0};
        FLAGSTATS_blockfile_stats st;
        int rc = FLAGSTATS_hip_blockimage_lz4(img.data(), img.size(), 0, out, &st);
        CHECK(rc == 0 && same(out, want) && st.gpu_decode == 1, "GPU decoder, image, pieces '%s': rc %d (%s)", pieces, rc, FLAGSTATS_hip_last_error());
        const std::string path = "/tmp/flagstats_tsan_" + std::to_string(getpid()) + ".lz4";
        FILE* f = std::fopen(path.c_str(), "wb");
        CHECK(f && std::fwrite(img.data(), 1, img.size(), f) == img.size(), "write %s", path.c_str());
        if (f) std::fclose(f);
        for (int readers : {1, 5, 16}) {
            uint64_t o2[32] = {0};
            rc = FLAGSTATS_hip_blockfile_lz4(path.c_str(), readers, o2, &st);
            CHECK(rc == 0 && same(o2, want) && st.gpu_decode == 1, "GPU decoder, file, %d readers: rc %d (%s)", readers, rc, FLAGSTATS_hip_last_error());
        }
        // a file cut inside a payload, and one with a damaged block: loud, and nothing stays on the device
        f = std::fopen(path.c_str(), "wb");
        if (f) {
            std::fwrite(img.data(), 1, img.size() - 1000, f);
            std::fclose(f);
        }
        uint64_t o3[32] = {0};
        CHECK(FLAGSTATS_hip_blockfile_lz4(path.c_str(), 5, o3, nullptr) != 0, "a truncated file must fail");
        std::vector<unsigned char> bad = img;
        bad[9] = 0x00;  // the first block's literal run now ends after 15 bytes: what follows is no valid sequence
        uint64_t o4[32] = {0};
        CHECK(FLAGSTATS_hip_blockimage_lz4(bad.data(), bad.size(), 0, o4, nullptr) != 0, "a damaged block must fail");
        CHECK(FLAGSTATS_hip_get("lz4_gpu_kept_bytes") == 0, "a failed call keeps nothing on the device");
        std::remove(path.c_str());
    }
    unsetenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS");
    // segments (files larger than the device may hold at once) and the idle rule of the kept buffers
    setenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES", "2000000", 1);
    {
        uint64_t out[32] = {0};
        FLAGSTATS_gpu_lz4_stats gs;
        const int rc = FLAGSTATS_hip_blockimage_lz4_gpu(img.data(), img.size(), out, &gs);
        CHECK(rc == 0 && same(out, want) && gs.segments >= 3, "segments: rc %d, %llu segments", rc, static_cast<unsigned long long>(gs.segments));
    }
    unsetenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES");
    CHECK(FLAGSTATS_hip_get("lz4_gpu_kept_bytes") > 0, "the buffers stay for the next call");
    for (int i = 0; i < 8; ++i) {
        uint32_t got[32] = {0};
        FLAGSTATS_u16(flags.data(), 1000, got);
    }
    CHECK(FLAGSTATS_hip_get("lz4_gpu_kept_bytes") == 0, "... and go after eight other calls");
    if (!dir.empty()) {
        for (const char* name : {"exact2_fast_a1.lz4", "hc_HC_c9.lz4", "ragged_fast_a2.lz4", "tiny_fast_a2.lz4"}) {
            uint64_t a[32] = {0}, b[32] = {0};
            CHECK(FLAGSTATS_hip_blockfile((dir + "/" + name).c_str(), 5, a, nullptr) == 0, "%s on the GPU decoder's host side", name);
            FLAGSTATS_hip_set("lz4_decoder", 0);
            CHECK(FLAGSTATS_hip_blockfile((dir + "/" + name).c_str(), 5, b, nullptr) == 0 && same(a, b), "%s: host pipeline gives the same", name);
            FLAGSTATS_hip_set("lz4_decoder", 1);
        }
    }
    // the same orchestration with Zstandard frames (scratch between the kernels per decode stream, two pieces at least)
    if (!dir.empty() && FLAGSTATS_hip_zstd_available()) {
        FLAGSTATS_hip_set("zstd_decoder", 1);
        for (const char* name : {"zexact1_c1.zst", "zragged_c3.zst", "ztiny_c19.zst"}) {
            uint64_t a[32] = {0}, b[32] = {0}, c[32] = {0};
            FLAGSTATS_blockfile_stats st;
            for (int readers : {1, 5}) {
                for (uint64_t& v : a) v = 0;
                CHECK(FLAGSTATS_hip_blockfile_zstd((dir + "/" + name).c_str(), readers, a, &st) == 0 && st.gpu_decode == 1, "%s on the GPU decoder's host side, %d readers: %s",
                      name, readers, FLAGSTATS_hip_last_error());
            }
            FLAGSTATS_hip_set("zstd_decoder", 0);
            const int rc_h = FLAGSTATS_hip_blockfile_zstd((dir + "/" + name).c_str(), 5, b, &st);
            CHECK(rc_h == 0 && st.gpu_decode == 0 && same(a, b), "%s: host pipeline gives the same (rc %d, gpu_decode %d, a[0] %llu b[0] %llu, a[16] %llu b[16] %llu: %s)", name, rc_h,
                  st.gpu_decode, (unsigned long long)a[0], (unsigned long long)b[0], (unsigned long long)a[16], (unsigned long long)b[16], FLAGSTATS_hip_last_error());
            FLAGSTATS_hip_set("zstd_decoder", 1);
            std::vector<unsigned char> zimg;
            if (FILE* f = std::fopen((dir + "/" + name).c_str(), "rb")) {
                unsigned char buf[65536];
                size_t got;
                while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) zimg.insert(zimg.end(), buf, buf + got);
                std::fclose(f);
            }
            CHECK(FLAGSTATS_hip_blockimage_zstd(zimg.data(), zimg.size(), 0, c, &st) == 0 && st.gpu_decode == 1 && same(a, c), "%s as an image", name);
            if (zimg.size() > 40) {
                zimg[8] ^= 0x5A;  // the first frame's magic number
                uint64_t d[32] = {0};
                CHECK(FLAGSTATS_hip_blockimage_zstd(zimg.data(), zimg.size(), 0, d, nullptr) != 0, "%s damaged must fail", name);
                CHECK(FLAGSTATS_hip_get("lz4_gpu_kept_bytes") == 0, "a failed call keeps nothing on the device");
            }
        }
        FLAGSTATS_hip_set("zstd_decoder", 2);
    }
    FLAGSTATS_hip_set("chunk_flags", 32ull << 20);
    FLAGSTATS_hip_set("lz4_decoder", 2);
}

static void golden_files(const std::string& dir)
{
    // counters of the reference-written files are in the manifest; here the files only have to give the same answer at
    // every thread count (the CPU / GPU suites check them against the manifest)
    for (const char* name : {"exact2_fast_a1.lz4", "hc_HC_c9.lz4", "ragged_fast_a2.lz4", "tiny_fast_a2.lz4"}) {
        uint64_t first[32];
        bool have = false;
        for (int threads : {1, 5, 20}) {
            uint64_t out[32] = {0};
            const int rc = FLAGSTATS_hip_blockfile((dir + "/" + name).c_str(), threads, out, nullptr);
            CHECK(rc == 0, "%s with %d threads: rc %d (%s)", name, threads, rc, FLAGSTATS_hip_last_error());
            if (!have) {
                std::memcpy(first, out, sizeof first);
                have = true;
            }
            CHECK(same(out, first), "%s: counters differ between thread counts", name);
        }
    }
}

static void sessions(const std::vector<uint16_t>& flags, const uint64_t* want)
{
    // two concurrent sessions, each with its own streams and pinned buffers, pushing blocks of different sizes
    auto one = [&](size_t block, int passes) {
        FLAGSTATS_hip_stream* s = FLAGSTATS_hip_stream_open();
        CHECK(s != nullptr, "stream_open: %s", FLAGSTATS_hip_last_error());
        if (!s) return;
        for (int p = 0; p < passes; ++p) {
            uint64_t out[32] = {0};
            for (size_t pos = 0; pos < flags.size(); pos += block) {
                const size_t m = block < flags.size() - pos ? block : flags.size() - pos;
                if ((pos / block) & 1) {
                    CHECK(FLAGSTATS_hip_stream_push(s, flags.data() + pos, m) == 0, "push");
                } else {
                    uint16_t* dst = FLAGSTATS_hip_stream_acquire(s, m);
                    CHECK(dst != nullptr, "acquire");
                    if (!dst) break;
                    std::memcpy(dst, flags.data() + pos, m * 2);
                    CHECK(FLAGSTATS_hip_stream_commit(s, m) == 0, "commit");
                }
            }
            CHECK(FLAGSTATS_hip_stream_finish(s, out) == 0 && same(out, want), "session (block %zu) pass %d", block, p);
        }
        FLAGSTATS_hip_stream_close(s);
    };
    std::thread a(one, static_cast<size_t>(512000), 2), b(one, static_cast<size_t>(70001), 2);
    a.join();
    b.join();
}

static void callers(const std::vector<uint16_t>& flags)
{
    // the reference API from several threads at once: small calls (polled result pairs), mid-size (one chunk), and
    // one multi-chunk call with a small chunk size
    std::vector<std::thread> pool;
    for (int t = 0; t < 4; ++t)
        pool.emplace_back([&, t] {
            for (int it = 0; it < 60; ++it) {
                const size_t n = (t * 7919 + it * 104729) % 200000 + 1, off = (it * 31337) % (flags.size() - n);
                uint32_t got[32] = {0};
                uint64_t want[32] = {0};
                CHECK(FLAGSTATS_u16(flags.data() + off, static_cast<uint32_t>(n), got) == 0, "FLAGSTATS_u16");
                oracle_flagstat_u16(flags.data() + off, n, want);
                for (int k = 0; k < 32; ++k) CHECK(got[k] == want[k], "thread %d call %d slot %d", t, it, k);
            }
        });
    for (auto& th : pool) th.join();
    // large pageable arrays take the staged way (worker threads copy into the page-locked chunks; here from 1,000 flags on):
    // two such callers and two small-call callers at once -- the staged call holds the default engine, the small calls find it
    // taken and go to the side engines
    FLAGSTATS_hip_set("staged_min_flags", 1000);
    FLAGSTATS_hip_set("chunk_flags", 300000);
    {
        std::vector<std::thread> mixed;
        uint64_t whole[32] = {0};
        oracle_flagstat_u16(flags.data(), flags.size(), whole);
        for (int t = 0; t < 2; ++t)
            mixed.emplace_back([&, t] {
                for (int it = 0; it < 3; ++it) {
                    uint64_t o[32] = {0};
                    CHECK(FLAGSTATS_u16_x64(flags.data() + t, flags.size() - static_cast<size_t>(t), o) == 0, "staged host call");
                    if (t == 0) CHECK(same(o, whole), "staged host call counters (thread %d, call %d)", t, it);
                }
            });
        for (int t = 0; t < 2; ++t)
            mixed.emplace_back([&, t] {
                for (int it = 0; it < 40; ++it) {
                    const size_t m = 500 + static_cast<size_t>(t) * 37 + static_cast<size_t>(it), off = static_cast<size_t>(it) * 4001;
                    uint32_t g32[32] = {0};
                    uint64_t w[32] = {0};
                    CHECK(FLAGSTATS_u16(flags.data() + off, static_cast<uint32_t>(m), g32) == 0, "small call beside staged ones");
                    oracle_flagstat_u16(flags.data() + off, m, w);
                    for (int k = 0; k < 32; ++k) CHECK(g32[k] == w[k], "small call beside staged ones: slot %d", k);
                }
            });
        for (auto& th : mixed) th.join();
        CHECK(FLAGSTATS_hip_get("staged_calls") >= 6, "the staged rule was not taken");
    }
    FLAGSTATS_hip_set("staged_min_flags", 1ull << 27);
    FLAGSTATS_hip_set("chunk_flags", 100000);
    uint64_t got[32] = {0}, want[32] = {0};
    CHECK(FLAGSTATS_u16_x64(flags.data(), flags.size(), got) == 0, "multi-chunk host call");
    oracle_flagstat_u16(flags.data(), flags.size(), want);
    CHECK(same(got, want), "multi-chunk host call counters");
    FLAGSTATS_hip_set("chunk_flags", 32ull << 20);
}

static void multi(const std::vector<uint16_t>& flags, const uint64_t* want)
{
    const int devs[3] = {0, 1, 0};
    uint64_t out[32] = {0};
    CHECK(FLAGSTATS_hip_multi_u16_x64(flags.data(), flags.size(), devs, 3, out) == 0 && same(out, want), "multi_u16_x64 over {0,1,0}");
    FLAGSTATS_hip_ctx* c0 = FLAGSTATS_hip_ctx_create(0);
    FLAGSTATS_hip_ctx* c1 = FLAGSTATS_hip_ctx_create(1);
    CHECK(c0 && c1, "ctx_create");
    std::thread a([&] {
        uint64_t o[32] = {0};
        CHECK(FLAGSTATS_hip_ctx_u16_x64(c0, flags.data(), flags.size(), o) == 0 && same(o, want), "ctx 0");
    });
    std::thread b([&] {
        uint64_t o[32] = {0};
        CHECK(FLAGSTATS_hip_ctx_u16_x64(c1, flags.data(), flags.size(), o) == 0 && same(o, want), "ctx 1");
    });
    a.join();
    b.join();
    FLAGSTATS_hip_ctx_destroy(c0);
    FLAGSTATS_hip_ctx_destroy(c1);
}

int main(int argc, char** argv)
{
    const size_t n = 3000017;
    std::vector<uint16_t> flags(n);
    oracle_generate_u16(ORACLE_GEN_NA12878, 7, 1, 0, n, flags.data());
    uint64_t want[32] = {0};
    oracle_flagstat_u16(flags.data(), n, want);
    CHECK(FLAGSTATS_hip_init(0) == 0, "init: %s", FLAGSTATS_hip_last_error());
    for (int round = 0; round < 2; ++round) {
        block_pipeline(flags, want);
        if (argc > 1) golden_files(argv[1]);
        gpu_decoder_host_side(flags, want, argc > 1 ? argv[1] : "");
        sessions(flags, want);
        callers(flags);
        multi(flags, want);
        FLAGSTATS_hip_shutdown();  // second round: everything is created again
    }
    std::printf(g_fail ? "tsan_driver: %d FAILED checks\n" : "tsan_driver: all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
