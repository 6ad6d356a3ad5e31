// lz4_decode_bench.cpp -- times the product's host LZ4 block decoder (csrc/lz4_block_decode.h) against
// liblz4's LZ4_decompress_safe (dlopen'ed; what the reference calls, benchmark/flagstats.cpp:316) on a
// file in the reference's block format.  Single thread, best of R passes over all blocks.
//   clang++ -O3 -std=c++17 -o /tmp/lz4b tests/perf/lz4_decode_bench.cpp -ldl && /tmp/lz4b file.lz4 [R]
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../libflagstats_amd/csrc/lz4_block_decode.h"

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const int R = argc > 2 ? std::atoi(argv[2]) : 10;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> img(static_cast<size_t>(sz));
    if (std::fread(img.data(), 1, img.size(), f) != img.size()) return 2;
    std::fclose(f);
    struct Blk { size_t pos; int32_t us, cs; };
    std::vector<Blk> blocks;
    size_t pos = 0, total = 0, maxus = 0;
    while (pos + 8 <= img.size()) {
        int32_t us, cs;
        std::memcpy(&us, &img[pos], 4);
        std::memcpy(&cs, &img[pos + 4], 4);
        blocks.push_back({pos + 8, us, cs});
        pos += 8 + static_cast<size_t>(cs);
        total += static_cast<size_t>(us);
        if (static_cast<size_t>(us) > maxus) maxus = static_cast<size_t>(us);
    }
    typedef int (*lz4fn)(const char*, char*, int, int);
    lz4fn ref = nullptr;
    for (const char* name : {"liblz4.so.1", "/opt/conda/lib/liblz4.so", "liblz4.so"}) {
        if (void* h = dlopen(name, RTLD_NOW)) {
            ref = reinterpret_cast<lz4fn>(dlsym(h, "LZ4_decompress_safe"));
            if (ref) break;
        }
    }
    std::vector<uint8_t> a(maxus + 64), b(maxus + 64);
    double best_own = 1e9, best_ref = 1e9;
    for (int r = 0; r < R; ++r) {
        double t0 = now();
        for (const Blk& k : blocks) {
            const int64_t got = fslz4::lz4_block_decode(&img[k.pos], static_cast<size_t>(k.cs), a.data(), static_cast<size_t>(k.us));
            if (got != k.us) { std::printf("own decoder failed: %lld vs %d\n", static_cast<long long>(got), k.us); return 1; }
        }
        double t1 = now();
        if (t1 - t0 < best_own) best_own = t1 - t0;
        if (ref) {
            t0 = now();
            for (const Blk& k : blocks) {
                const int got = ref(reinterpret_cast<const char*>(&img[k.pos]), reinterpret_cast<char*>(b.data()), k.cs, k.us);
                if (got != k.us) { std::printf("liblz4 failed\n"); return 1; }
            }
            t1 = now();
            if (t1 - t0 < best_ref) best_ref = t1 - t0;
        }
    }
    // last block decoded by both must agree byte for byte
    if (ref && !blocks.empty() && std::memcmp(a.data(), b.data(), static_cast<size_t>(blocks.back().us)) != 0) {
        std::printf("MISMATCH vs liblz4\n");
        return 1;
    }
    std::printf("%s: %zu blocks, %zu -> %ld bytes (ratio %.2f)  own %.3f GB/s  liblz4 %.3f GB/s  (output bytes, 1 thread, best of %d)\n",
                argv[1], blocks.size(), total, sz, static_cast<double>(total) / sz, total / best_own / 1e9,
                ref ? total / best_ref / 1e9 : 0.0, R);
    return 0;
}
