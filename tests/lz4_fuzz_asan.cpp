// lz4_fuzz_asan.cpp -- memory-safety fuzz of the product's LZ4 block decoder (csrc/lz4_block_decode.h),
// built by tests/test_lz4_decoder.py with -fsanitize=address,undefined (CPU only).  Blocks big enough to
// run the decoder's fast loops (they start after 64 KiB of output) are compressed with the image's
// liblz4, damaged (byte flips, truncation, wrong capacities) and decoded into exact-size heap buffers:
// any read outside [src, src+n) or write outside [dst, dst+cap) aborts under ASan.  Undamaged blocks and
// damaged ones that both decoders accept must give identical bytes.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../libflagstats_amd/csrc/lz4_block_decode.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return rng_state * 0x2545F4914F6CDD1Dull;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 300;
    typedef int (*cfn)(const char*, char*, int, int);
    typedef int (*dfn)(const char*, char*, int, int);
    typedef int (*hfn)(const char*, char*, int, int, int);
    cfn comp = nullptr;
    dfn ref = nullptr;
    hfn comp_hc = nullptr;
    for (const char* name : {"liblz4.so.1", "/opt/conda/lib/liblz4.so", "liblz4.so"}) {
        if (void* h = dlopen(name, RTLD_NOW)) {
            comp = reinterpret_cast<cfn>(dlsym(h, "LZ4_compress_default"));
            ref = reinterpret_cast<dfn>(dlsym(h, "LZ4_decompress_safe"));
            comp_hc = reinterpret_cast<hfn>(dlsym(h, "LZ4_compress_HC"));
            if (comp && ref) break;
        }
    }
    if (!comp || !ref) {
        std::printf("SKIP no liblz4\n");
        return 0;
    }
    long accepted = 0, rejected = 0, compared = 0;
    for (int kind = 0; kind < 4; ++kind) {
        // flag-like streams: kind 0 few values (bare matches), 1 more values (literals + matches),
        // 2 long runs (overlapping matches, small offsets), 3 near-random (mostly literals)
        const size_t n = 400000 + (rnd() % 200000);
        std::vector<uint8_t> raw(n);
        static const uint16_t common[4] = {99, 147, 83, 163};
        for (size_t i = 0; i + 1 < n; i += 2) {
            uint16_t v;
            const uint64_t r = rnd();
            if (kind == 0) v = (r % 100 < 95) ? common[(r >> 8) & 3] : static_cast<uint16_t>((r >> 16) & 0xFFF);
            else if (kind == 1) v = (r % 100 < 70) ? common[(r >> 8) & 3] : static_cast<uint16_t>((r >> 16) & 0x3FF);
            else if (kind == 2) v = ((i / 2) % 5000 < 4000) ? 99 : common[(r >> 8) & 3];
            else v = static_cast<uint16_t>(r >> 20);
            raw[i] = static_cast<uint8_t>(v);
            raw[i + 1] = static_cast<uint8_t>(v >> 8);
        }
        for (int hc = 0; hc < 2; ++hc) {
            std::vector<uint8_t> c(n + n / 200 + 64);
            const int cs = (hc && comp_hc)
                               ? comp_hc(reinterpret_cast<const char*>(raw.data()), reinterpret_cast<char*>(c.data()), static_cast<int>(n), static_cast<int>(c.size()), 9)
                               : comp(reinterpret_cast<const char*>(raw.data()), reinterpret_cast<char*>(c.data()), static_cast<int>(n), static_cast<int>(c.size()));
            if (cs <= 0) return 2;
            // exact-size copies so ASan sees every overrun
            {
                std::vector<uint8_t> src(c.begin(), c.begin() + cs), dst(n);
                const int64_t got = fslz4::lz4_block_decode(src.data(), src.size(), dst.data(), dst.size());
                if (got != static_cast<int64_t>(n) || std::memcmp(dst.data(), raw.data(), n) != 0) {
                    std::printf("FAIL clean block kind %d hc %d: %lld\n", kind, hc, static_cast<long long>(got));
                    return 1;
                }
                std::vector<uint8_t> small(n - 1);
                if (fslz4::lz4_block_decode(src.data(), src.size(), small.data(), small.size()) >= 0) {
                    std::printf("FAIL capacity n-1 accepted\n");
                    return 1;
                }
            }
            for (int r = 0; r < rounds; ++r) {
                size_t len = static_cast<size_t>(cs);
                const int what = static_cast<int>(rnd() % 4);
                if (what == 0) len = 1 + rnd() % len;                       // truncation
                std::vector<uint8_t> src(c.begin(), c.begin() + static_cast<long>(len));
                if (what != 0 || (rnd() & 1)) {
                    const int flips = 1 + static_cast<int>(rnd() % 4);
                    for (int k = 0; k < flips; ++k) {
                        // damage beyond the first 40 KB of input so it lands where the fast loops run
                        const size_t lo = len > 60000 ? 40000 : 0;
                        src[lo + rnd() % (len - lo)] = static_cast<uint8_t>(rnd());
                    }
                }
                size_t cap = n;
                if (what == 3) cap = n - (rnd() % 200);                     // too-small output
                std::vector<uint8_t> dst(cap), dref(cap);
                const int64_t got = fslz4::lz4_block_decode(src.data(), src.size(), dst.data(), dst.size());
                const int want = ref(reinterpret_cast<const char*>(src.data()), reinterpret_cast<char*>(dref.data()), static_cast<int>(src.size()), static_cast<int>(cap));
                if (got > static_cast<int64_t>(cap)) {
                    std::printf("FAIL returned more than the capacity\n");
                    return 1;
                }
                if (got >= 0) ++accepted; else ++rejected;
                if (got >= 0 && want >= 0) {
                    ++compared;
                    if (got != want || std::memcmp(dst.data(), dref.data(), static_cast<size_t>(got)) != 0) {
                        std::printf("FAIL decoders disagree on an input both accept (kind %d hc %d round %d): %lld vs %d\n", kind, hc, r,
                                    static_cast<long long>(got), want);
                        return 1;
                    }
                }
            }
        }
    }
    std::printf("OK accepted %ld rejected %ld compared %ld\n", accepted, rejected, compared);
    return 0;
}
