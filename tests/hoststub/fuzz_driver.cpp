// TEST-ONLY fuzz driver for the host-stub build under AddressSanitizer + UndefinedBehaviorSanitizer
// (make -C libflagstats_amd/csrc hoststub SAN=address,undefined): everything on the host side that takes attacker-shaped
// sizes from a file -- the block-file index parsers (host pipeline and GPU-decoder orchestration), the piece / span / segment
// arithmetic, the reader pool of file mode, the raw-file reader, the FLAG-text parser, the engine's staging paths -- is fed
// block files whose headers are negative, huge, overlapping or truncated and whose payloads are damaged, as images and as
// files, through every entry, with the decoder on the host threads and "on the GPU" (in this build the stand-in kernels are
// the product's host LZ4 decoder and the image's libzstd), segments forced small, pieces forced odd.
//   fuzz_driver <inputs> [dir with the reference-written golden block files] [seed]
// Beyond "no sanitizer report" every input is a differential test of the orchestration: all paths must agree on accept /
// reject and, when they accept, on the 32 counters; an undamaged input must give the oracle's counters; a rejected input
// must leave the caller's counters untouched (the += contract: never partial sums); the reference fails loudly on such
// input too (run_screaming, benchmark/flagstats.cpp:105-108,256-259).
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/libflagstats_hip_probe.h"
extern "C" {
#include "../../oracle/flagstat_oracle.h"
}

typedef std::vector<unsigned char> Bytes;

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t below(uint64_t n) { return n ? rnd() % n : 0; }

static long g_fail = 0;
static long g_refused = 0, g_refused_host = 0;   // path 6: calls made, calls that the host threads took
#define CHECK(cond, ...)                                              \
    do {                                                              \
        if (!(cond)) {                                                \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);                        \
            std::fprintf(stderr, "\n");                               \
            if (++g_fail > 20) std::exit(1);                          \
        }                                                             \
    } while (0)

struct Block {
    Bytes payload;
    Bytes decoded;  // what an undamaged payload decodes to
};

// ---- valid LZ4 blocks made here: random sequences (literal runs, matches with small / overlapping / far offsets, length bytes)
static void put_len(Bytes& c, size_t v)
{
    while (v >= 255) {
        c.push_back(255);
        v -= 255;
    }
    c.push_back(static_cast<unsigned char>(v));
}

static Block make_lz4_block(size_t target)
{
    Block b;
    Bytes& out = b.decoded;
    Bytes& c = b.payload;
    while (out.size() < target) {
        size_t ll = below(4) == 0 ? below(40) : 0, ml = 4 + (below(8) == 0 ? below(600) : below(16));
        if (out.empty() && ll == 0) ll = 4;
        const size_t have = out.size() + ll;
        size_t off = below(3) == 0 ? 1 + below(8) : 1 + below(have < 65535 ? have : 65535);
        if (off > have) off = have;
        c.push_back(static_cast<unsigned char>(((ll < 15 ? ll : 15) << 4) | (ml - 4 < 15 ? ml - 4 : 15)));
        if (ll >= 15) put_len(c, ll - 15);
        for (size_t i = 0; i < ll; ++i) {
            const unsigned char v = static_cast<unsigned char>(below(7) ? below(4) * 16 + 3 : rnd());  // flag-like low bytes, some noise
            c.push_back(v);
            out.push_back(v);
        }
        c.push_back(static_cast<unsigned char>(off & 255));
        c.push_back(static_cast<unsigned char>(off >> 8));
        if (ml - 4 >= 15) put_len(c, ml - 4 - 15);
        for (size_t i = 0; i < ml; ++i) out.push_back(out[out.size() - off]);
    }
    const size_t ll = 12 + below(30);  // the closing literals-only sequence (the format's end-of-block rules)
    c.push_back(static_cast<unsigned char>((ll < 15 ? ll : 15) << 4));
    if (ll >= 15) put_len(c, ll - 15);
    for (size_t i = 0; i < ll; ++i) {
        const unsigned char v = static_cast<unsigned char>(rnd());
        c.push_back(v);
        out.push_back(v);
    }
    return b;
}

// ---- Zstandard frames from the image's libzstd (what the reference's writer calls, benchmark/flagstats.cpp:86)
typedef size_t (*zcompress_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zbound_fn)(size_t);
typedef unsigned (*ziserr_fn)(size_t);
static zcompress_fn z_compress = nullptr;
static zbound_fn z_bound = nullptr;
static ziserr_fn z_iserr = nullptr;

static Block make_zstd_block(size_t target)
{
    Block b;
    b.decoded.resize(target);
    const int kind = static_cast<int>(below(4));
    static const unsigned short common[4] = {99, 147, 83, 163};
    for (size_t i = 0; i < target; ++i) {
        const unsigned short v = kind == 0 ? common[below(4)] : (kind == 1 ? (below(10) ? common[below(4)] : static_cast<unsigned short>(rnd() & 0xFFF)) : (kind == 2 ? 99 : static_cast<unsigned short>(rnd())));
        b.decoded[i] = static_cast<unsigned char>((i & 1) ? v >> 8 : v);
    }
    b.payload.resize(z_bound(target) + 16);
    // (the high levels allocate and release tables of many MiB per call, which is all a profile of this driver then shows)
    static const int levels[8] = {1, 1, 3, 3, -5, 2, 9, 19};
    const size_t n = z_compress(b.payload.data(), b.payload.size(), b.decoded.data(), target, levels[below(below(6) ? 6 : 8)]);
    if (z_iserr(n)) std::exit(4);
    b.payload.resize(n);
    return b;
}

// ... and frames of MANY small blocks: a streaming compressor flushing every `every` bytes (>= 1024: a block per KiB is what the
// GPU decoder's second pass has room for; the first pass answers kZstdTooManyBlocks)
struct ZBuf {
    void* p;
    size_t size, pos;
};
typedef void* (*zcreate_fn)(void);
typedef size_t (*zfree_fn)(void*);
typedef size_t (*zstream2_fn)(void*, ZBuf*, ZBuf*, int);
static zcreate_fn z_create = nullptr;
static zfree_fn z_free = nullptr;
static zstream2_fn z_stream2 = nullptr;

static Block make_zstd_block_flushed(size_t target, size_t every)
{
    Block b = make_zstd_block(target);
    if (!z_create || !z_free || !z_stream2 || target == 0) return b;
    void* cctx = z_create();
    b.payload.assign(target + target / every * 32 + 1024, 0);
    ZBuf ob{b.payload.data(), b.payload.size(), 0};
    for (size_t at = 0; at < target;) {
        const size_t n = every < target - at ? every : target - at;
        ZBuf ib{b.decoded.data() + at, n, 0};
        const bool last = at + n == target;
        for (;;) {
            const size_t left = z_stream2(cctx, &ob, &ib, last ? 2 : 1);   // ZSTD_e_end / ZSTD_e_flush
            if (z_iserr(left)) std::exit(4);
            if (left == 0 && ib.pos == ib.size) break;
        }
        at += n;
    }
    z_free(cctx);
    b.payload.resize(ob.pos);
    return b;
}

static void put32(Bytes& img, int32_t v)
{
    const unsigned char* p = reinterpret_cast<const unsigned char*>(&v);
    img.insert(img.end(), p, p + 4);
}

// block file image (benchmark/flagstats.cpp:119-138) + the oracle's counters over size >> 1 flags of every block (:323)
static Bytes image_of(const std::vector<Block>& blocks, uint64_t want[32])
{
    Bytes img;
    for (int k = 0; k < 32; ++k) want[k] = 0;
    for (const Block& b : blocks) {
        put32(img, static_cast<int32_t>(b.decoded.size()));
        put32(img, static_cast<int32_t>(b.payload.size()));
        img.insert(img.end(), b.payload.begin(), b.payload.end());
        std::vector<uint16_t> fl(b.decoded.size() / 2);
        if (!fl.empty()) std::memcpy(fl.data(), b.decoded.data(), fl.size() * 2);
        uint64_t one[32] = {0};
        if (!fl.empty()) oracle_flagstat_u16(fl.data(), fl.size(), one);
        for (int k = 0; k < 32; ++k) want[k] += one[k];
    }
    return img;
}

// ---- damage.  Returns what was done (for the failure message); 0 = nothing (the input stays valid)
static int damage(Bytes& img, const std::vector<size_t>& header_at)
{
    const int how = static_cast<int>(below(12));
    if (how == 0 || img.empty() || header_at.empty()) return 0;
    const size_t h = header_at[below(header_at.size())];
    int32_t us, cs;
    std::memcpy(&us, &img[h], 4);
    std::memcpy(&cs, &img[h + 4], 4);
    // (sizes of hundreds of MiB are honoured by the readers -- buffers of that size are allocated and zero-filled before the
    // payload turns out not to decode to them -- which costs seconds under the sanitizers: they come up a handful of times in 10^5 inputs)
    static const int32_t nasty[] = {-1, -2, INT32_MIN, -65536, 0, 1, 7, 8, 9, 15, 16, 17, 4095, 4096, 65535, 65536, 1 << 20, (1 << 24) - 1, 1 << 24};
    static const int32_t huge[] = {INT32_MAX, INT32_MAX - 1, 1 << 28, (1 << 28) - 1, 1 << 30, 600 << 20};
    auto pick = [&](int32_t base) -> int32_t {
        if (below(4000) == 0) return below(2) ? huge[below(sizeof huge / sizeof huge[0])] : static_cast<int32_t>(rnd() & 0x7FFFFFFF);
        switch (below(4)) {
        case 0: return nasty[below(sizeof nasty / sizeof nasty[0])];
        case 1: return base + static_cast<int32_t>(below(5)) - 2;
        case 2: return static_cast<int32_t>(below(static_cast<uint64_t>(base) * 2 + 2));
        default: return below(2) ? static_cast<int32_t>(below(1 << 21)) : -static_cast<int32_t>(below(1u << 31));
        }
    };
    switch (how) {
    case 1: us = pick(us); std::memcpy(&img[h], &us, 4); break;            // declared decoded size
    case 2: cs = pick(cs); std::memcpy(&img[h + 4], &cs, 4); break;        // payload size: blocks overlap / run past the end
    case 3: us = pick(us); cs = pick(cs); std::memcpy(&img[h], &us, 4); std::memcpy(&img[h + 4], &cs, 4); break;
    case 4: img.resize(below(img.size())); break;                           // truncated anywhere
    case 5: img.resize(h + below(9)); break;                                // truncated inside a header
    case 6: for (int i = 0, n = 1 + static_cast<int>(below(4)); i < n; ++i) img[below(img.size())] ^= static_cast<unsigned char>(1u << below(8)); break;
    case 7: for (int i = 0, n = 1 + static_cast<int>(below(16)); i < n; ++i) img[below(img.size())] = static_cast<unsigned char>(rnd()); break;
    case 8: img.insert(img.end(), static_cast<size_t>(below(20)), static_cast<unsigned char>(rnd())); break;  // bytes behind the last block
    case 9: if (cs > 0) { img[h + 8 + below(static_cast<uint64_t>(cs))] ^= 0xFF; } break;
    case 10: { const size_t a = below(img.size()), n = below(img.size() - a); img.erase(img.begin() + static_cast<long>(a), img.begin() + static_cast<long>(a + n)); break; }
    default: { const size_t a = below(img.size()); img.insert(img.begin() + static_cast<long>(a), static_cast<size_t>(1 + below(9)), static_cast<unsigned char>(rnd())); break; }
    }
    return how;
}

static const uint64_t kPreset = 1000003;  // what the caller's counters hold before a call: += on success, untouched on failure

struct Verdict {
    int rc;
    uint64_t out[32];
};

static bool untouched(const uint64_t* out)
{
    for (int k = 0; k < 32; ++k)
        if (out[k] != kPreset + static_cast<uint64_t>(k)) return false;
    return true;
}

static std::string g_path;

static void write_file(const Bytes& img, const char* suffix)
{
    g_path = "/tmp/flagstats_fuzz_" + std::to_string(getpid()) + suffix;
    const int fd = open(g_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0 || (img.size() && write(fd, img.data(), img.size()) != static_cast<ssize_t>(img.size()))) std::exit(5);
    close(fd);
}

// one block-file input through up to six paths; returns how many calls were made
static int run_blockfile_input(bool zstd, const Bytes& img, bool valid, const uint64_t* want, long input_no, int how)
{
    const char* knob = zstd ? "zstd_decoder" : "lz4_decoder";
    std::vector<Verdict> got;
    std::vector<std::string> names;
    bool file_written = false;
    uint64_t max_us = 0;
    for (size_t pos = 0; pos + 8 <= img.size();) {   // (what a parser that trusts nothing would see; only to know whether paths may differ)
        int32_t us, cs;
        std::memcpy(&us, &img[pos], 4);
        std::memcpy(&cs, &img[pos + 4], 4);
        if (us < 0 || cs < 0) break;
        if (static_cast<uint64_t>(us) > max_us) max_us = static_cast<uint64_t>(us);
        pos += 8 + static_cast<uint64_t>(cs);
    }
    const bool huge = max_us > (8ull << 20);   // (beyond the chunk buffer the host pipeline refuses what the GPU path may take)
    const char* pm = std::getenv("FUZZ_PATHS");
    for (int path = 0; path < 7; ++path) {
        if (pm && !std::strchr(pm, '0' + path)) continue;
        // 0 host threads / image, 1 GPU path / image, 2 GPU path / file, 3 host threads / file, 4 GPU path / image / small segments,
        // 5 GPU path / file / odd pieces + small segments, 6 decoder chosen by size / image or file / the device "cannot hold" the decoded
        // buffer (found out on the allocation's own thread, with copies already queued): the host threads take the file
        if (path >= 2 && below(3) && !(path == 2 && input_no % 4 == 0)) continue;   // the file and segment paths on a third of the inputs
        if (huge && path != 1 && path != 2) continue;
        if (path == 6 && input_no % 8 != 3) continue;
        FLAGSTATS_hip_set(knob, (path == 0 || path == 3) ? 0 : (path == 6 ? 2 : 1));
        const bool p6_file = path == 6 && below(2);
        if (path == 6) {
            FLAGSTATS_hip_set(zstd ? "zstd_gpu_min_bytes" : "lz4_gpu_min_bytes", 1);
            FLAGSTATS_hip_set("lz4_gpu_keep_bytes", 0);   // (nothing kept from the input before: the buffer has to be asked for)
            setenv("FLAGSTATS_HIP_GPU_OUT_CAP", "1", 1);
        }
        if (path == 4 || path == 5) {
            setenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES", below(2) ? "1" : std::to_string(1 + below(200000)).c_str(), 1);
            if (path == 5) setenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS", std::to_string(1 + below(9)).c_str(), 1);
        }
        Verdict v;
        for (int k = 0; k < 32; ++k) v.out[k] = kPreset + static_cast<uint64_t>(k);
        const int threads = static_cast<int>(below(6));
        FLAGSTATS_blockfile_stats st;
        if (path == 2 || path == 3 || path == 5 || p6_file) {
            if (!file_written) write_file(img, zstd ? ".zst" : ".lz4");
            file_written = true;
            v.rc = below(2) ? FLAGSTATS_hip_blockfile(g_path.c_str(), threads, v.out, &st)
                            : (zstd ? FLAGSTATS_hip_blockfile_zstd(g_path.c_str(), threads, v.out, &st) : FLAGSTATS_hip_blockfile_lz4(g_path.c_str(), threads, v.out, &st));
        } else {
            // (an exact-size heap copy: a read past the image's end is an ASan report)
            Bytes copy(img);
            v.rc = zstd ? FLAGSTATS_hip_blockimage_zstd(copy.data(), copy.size(), threads, v.out, &st) : FLAGSTATS_hip_blockimage_lz4(copy.data(), copy.size(), threads, v.out, &st);
        }
        if (path == 6) {
            unsetenv("FLAGSTATS_HIP_GPU_OUT_CAP");
            FLAGSTATS_hip_set(zstd ? "zstd_gpu_min_bytes" : "lz4_gpu_min_bytes", 64ull << 20);
            FLAGSTATS_hip_set("lz4_gpu_keep_bytes", ~0ull);
            // (a buffer kept from the input before may be large enough: then nothing is asked for and the GPU decoder runs)
            ++g_refused;
            if (v.rc == 0 && st.gpu_decode == 0) ++g_refused_host;
        }
        if (path == 4 || path == 5) {
            unsetenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES");
            unsetenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS");
        }
        static const char* const pname[7] = {"host/image", "gpu/image", "gpu/file", "host/file", "gpu/image/segments", "gpu/file/pieces+segments", "by size/decoded buffer refused"};
        names.push_back(pname[path]);
        got.push_back(v);
        if (v.rc != 0) CHECK(untouched(v.out), "input %ld (%s, damage %d), %s: a failed call changed the caller's counters", input_no, zstd ? "zstd" : "lz4", how, pname[path]);
        if (v.rc != 0) CHECK(FLAGSTATS_hip_last_error()[0] != 0, "input %ld, %s: failure without a message", input_no, pname[path]);
        if (valid) {
            bool ok = v.rc == 0;
            for (int k = 0; k < 32 && ok; ++k) ok = v.out[k] == kPreset + static_cast<uint64_t>(k) + want[k];
            CHECK(ok, "input %ld (%s), %s: a valid file gives rc %d / other counters than the oracle (%s)", input_no, zstd ? "zstd" : "lz4", pname[path], v.rc, FLAGSTATS_hip_last_error());
        }
    }
    for (size_t i = 1; i < got.size(); ++i) {
        const bool same_rc = (got[i].rc == 0) == (got[0].rc == 0);
        CHECK(same_rc, "input %ld (%s, damage %d): %s gives rc %d, %s gives rc %d", input_no, zstd ? "zstd" : "lz4", how, names[0].c_str(), got[0].rc, names[i].c_str(), got[i].rc);
        if (same_rc && got[i].rc == 0)
            CHECK(std::memcmp(got[i].out, got[0].out, sizeof got[0].out) == 0, "input %ld (%s, damage %d): %s and %s accept with different counters", input_no, zstd ? "zstd" : "lz4", how,
                  names[0].c_str(), names[i].c_str());
    }
    FLAGSTATS_hip_set(knob, 2);
    return static_cast<int>(got.size());
}

static std::vector<Block> read_golden(const std::string& path)
{
    std::vector<Block> blocks;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return blocks;
    Bytes img;
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) img.insert(img.end(), buf, buf + n);
    std::fclose(f);
    for (size_t pos = 0; pos + 8 <= img.size();) {
        int32_t us, cs;
        std::memcpy(&us, &img[pos], 4);
        std::memcpy(&cs, &img[pos + 4], 4);
        Block b;
        b.payload.assign(img.begin() + static_cast<long>(pos + 8), img.begin() + static_cast<long>(pos + 8 + static_cast<size_t>(cs)));
        b.decoded.resize(static_cast<size_t>(us));  // (contents unknown here: golden blocks are only used damaged or for path agreement)
        blocks.push_back(b);
        pos += 8 + static_cast<size_t>(cs);
    }
    return blocks;
}

static void fuzz_text(long input_no)
{
    // FLAG text -> uint16 (benchmark/utility.cpp:9-16): random bytes, digit-rich lines, no trailing newline, NULs, huge numbers
    const size_t len = below(3) ? below(200) : below(5000);
    std::string s(len, '\0');
    static const char alphabet[] = "0123456789\n\n -+\r\t99 147\0x";
    const int kind = static_cast<int>(below(3));
    for (size_t i = 0; i < len; ++i) s[i] = kind == 0 ? static_cast<char>(rnd()) : alphabet[below(kind == 1 ? 12 : sizeof alphabet)];
    char* text = static_cast<char*>(std::malloc(len ? len : 1));  // exact size, not NUL-terminated
    if (len) std::memcpy(text, s.data(), len);
    const uint64_t lines = FLAGSTATS_text_count_lines(text, len);
    std::vector<uint16_t> out(lines);
    const int64_t n = FLAGSTATS_text_to_u16(text, len, out.data(), lines);
    CHECK(n == static_cast<int64_t>(lines), "text input %ld: %lld values for %llu lines", input_no, static_cast<long long>(n), static_cast<unsigned long long>(lines));
    if (lines > 0) {
        std::vector<uint16_t> fewer(lines - 1);
        CHECK(FLAGSTATS_text_to_u16(text, len, fewer.data(), lines - 1) < 0, "text input %ld: a short output buffer must be refused", input_no);
    }
    // every value against atoi's rule on the line's own bytes (std::getline + atoi, as the reference reads them)
    size_t at = 0, idx = 0;
    while (at < len && idx < lines) {
        size_t eol = at;
        while (eol < len && s[eol] != '\n') ++eol;
        std::string line = s.substr(at, eol - at);
        const size_t nul = line.find('\0');
        if (nul != std::string::npos) line.resize(nul);
        long long v = std::strtoll(line.c_str(), nullptr, 10);
        if (v > 2147483647ll) v = 2147483647ll;
        if (v < -2147483648ll) v = -2147483648ll;
        // (strtoll skips '\n'-free white space like atoi; the product stops at the first non-digit as both do)
        CHECK(out[idx] == static_cast<uint16_t>(static_cast<uint32_t>(static_cast<int32_t>(v))), "text input %ld line %zu: got %u for '%s'", input_no, idx, out[idx], line.c_str());
        ++idx;
        at = eol + 1;
    }
    std::free(text);
}

static void fuzz_raw_file(long input_no)
{
    // `bench decompress -D` (benchmark/flagstats.cpp:415-468): a raw uint16 file of any length, odd ones included
    const size_t n = below(4) ? below(5000) : below(300000);
    Bytes raw(n);
    for (size_t i = 0; i < n; ++i) raw[i] = static_cast<unsigned char>(rnd());
    write_file(raw, ".bin");
    uint64_t out[32], want[32] = {0};
    for (int k = 0; k < 32; ++k) out[k] = kPreset + static_cast<uint64_t>(k);
    std::vector<uint16_t> fl(n / 2);
    if (!fl.empty()) std::memcpy(fl.data(), raw.data(), fl.size() * 2);
    if (!fl.empty()) oracle_flagstat_u16(fl.data(), fl.size(), want);
    if (below(2)) FLAGSTATS_hip_set("chunk_flags", 8 + below(100000));
    FLAGSTATS_blockfile_stats st;
    const int rc = FLAGSTATS_hip_file_raw(g_path.c_str(), out, &st);
    FLAGSTATS_hip_set("chunk_flags", 4ull << 20);
    bool ok = rc == 0 && st.n_flags == fl.size();
    for (int k = 0; k < 32 && ok; ++k) ok = out[k] == kPreset + static_cast<uint64_t>(k) + want[k];
    CHECK(ok, "raw file input %ld (%zu bytes): rc %d (%s)", input_no, n, rc, FLAGSTATS_hip_last_error());
    // the same flags as a host ARRAY in pageable memory, through the size rule that sends large ones into the page-locked chunks
    // (knob staged_min_flags, set to 1 here) and through the runtime's own copy (0): an exact-size heap copy, odd starts included
    if (!fl.empty()) {
        const size_t skip = below(2) ? 0 : below(fl.size() < 3 ? fl.size() : 3);
        std::vector<uint16_t> exact(fl.begin() + static_cast<std::ptrdiff_t>(skip), fl.end());
        uint64_t w2[32] = {0};
        if (!exact.empty()) oracle_flagstat_u16(exact.data(), exact.size(), w2);
        for (int staged = 0; staged < 2; ++staged) {
            FLAGSTATS_hip_set("staged_min_flags", staged ? 1 : 0);
            if (below(2)) FLAGSTATS_hip_set("chunk_flags", 8 + below(100000));
            uint64_t o2[32];
            for (int k = 0; k < 32; ++k) o2[k] = kPreset + static_cast<uint64_t>(k);
            const int rc2 = exact.empty() ? 0 : FLAGSTATS_u16_x64(exact.data(), exact.size(), o2);
            FLAGSTATS_hip_set("chunk_flags", 4ull << 20);
            bool ok2 = rc2 == 0;
            for (int k = 0; k < 32 && ok2; ++k) ok2 = o2[k] == kPreset + static_cast<uint64_t>(k) + w2[k];
            CHECK(ok2, "host array input %ld (%zu flags, %s): rc %d (%s)", input_no, exact.size(), staged ? "staged" : "runtime copy", rc2, FLAGSTATS_hip_last_error());
        }
        FLAGSTATS_hip_set("staged_min_flags", 1ull << 27);
    }
}

int main(int argc, char** argv)
{
    const long inputs = argc > 1 ? std::atol(argv[1]) : 1000;
    const std::string dir = argc > 2 ? argv[2] : "";
    if (argc > 3) g_rng = std::strtoull(argv[3], nullptr, 0);
    if (void* h = dlopen("libzstd.so.1", RTLD_NOW)) {
        z_compress = reinterpret_cast<zcompress_fn>(dlsym(h, "ZSTD_compress"));
        z_bound = reinterpret_cast<zbound_fn>(dlsym(h, "ZSTD_compressBound"));
        z_iserr = reinterpret_cast<ziserr_fn>(dlsym(h, "ZSTD_isError"));
        z_create = reinterpret_cast<zcreate_fn>(dlsym(h, "ZSTD_createCCtx"));
        z_free = reinterpret_cast<zfree_fn>(dlsym(h, "ZSTD_freeCCtx"));
        z_stream2 = reinterpret_cast<zstream2_fn>(dlsym(h, "ZSTD_compressStream2"));
    }
    const bool have_zstd = z_compress && z_bound && z_iserr && FLAGSTATS_hip_zstd_available();
    setenv("FLAGSTATS_HIP_GPU_BUFFER_GRAIN", "16", 1);   // exact-size "device" buffers: an access behind them is a report
    FLAGSTATS_hip_set("on_error", 0);
    CHECK(FLAGSTATS_hip_init(0) == 0, "init: %s", FLAGSTATS_hip_last_error());
    FLAGSTATS_hip_set("chunk_flags", 4ull << 20);       // 8 MiB chunk buffers: what a "huge" header is measured against
    FLAGSTATS_hip_set("lz4_gpu_min_bytes", 1);
    FLAGSTATS_hip_set("zstd_gpu_min_bytes", 1);
    std::vector<std::vector<Block>> golden_lz4, golden_zstd;
    if (!dir.empty()) {
        for (const char* name : {"exact2_fast_a1.lz4", "hc_HC_c9.lz4", "ragged_fast_a2.lz4", "tiny_fast_a2.lz4"}) golden_lz4.push_back(read_golden(dir + "/" + name));
        for (const char* name : {"zexact1_c1.zst", "zragged_c3.zst", "ztiny_c19.zst"}) golden_zstd.push_back(read_golden(dir + "/" + name));
    }
    static const size_t sizes[] = {0, 1, 2, 15, 16, 17, 100, 1000, 4097, 30000};
    std::vector<Block> zpool;   // Zstandard frames are made once (compression is what costs here): inputs differ by which they take and by the damage
    if (have_zstd)
        for (int i = 0; i < 400; ++i) {
            if (i % 25 == 7)
                zpool.push_back(make_zstd_block_flushed(20000 + below(60000), 1024 + below(1500)));   // 10-80 blocks a frame: the second pass
            else
                zpool.push_back(make_zstd_block(i % 40 == 0 ? 70000 + below(80000) : sizes[below(sizeof sizes / sizeof sizes[0])] + below(40)));
        }
    long calls = 0, n_valid = 0, n_lz4 = 0, n_zstd = 0, n_text = 0, n_raw = 0, n_golden = 0;
    for (long i = 0; i < inputs; ++i) {
        uint64_t what = below(100);
        if (const char* only = std::getenv("FUZZ_ONLY")) what = !std::strcmp(only, "text") ? 0 : (!std::strcmp(only, "raw") ? 13 : (!std::strcmp(only, "lz4") ? 20 : 80));
        if (what < 12) {
            fuzz_text(i);
            ++n_text;
            ++calls;
            continue;
        }
        if (what < 16) {
            fuzz_raw_file(i);
            ++n_raw;
            ++calls;
            continue;
        }
        const bool zstd = have_zstd && what >= 70;
        std::vector<Block> blocks;
        bool from_golden = false;
        const std::vector<std::vector<Block>>& gold = zstd ? golden_zstd : golden_lz4;
        if (!gold.empty() && below(400) == 0) {
            blocks = gold[below(gold.size())];   // a reference-written file (up to 1,024,000-byte blocks): rarely, they are large
            from_golden = true;
            ++n_golden;
        } else {
            const size_t nb = below(20) == 0 ? 0 : 1 + below(below(4) ? 4 : 24);
            for (size_t b = 0; b < nb; ++b) {
                const size_t target = below(30) == 0 ? 70000 + below(80000) : sizes[below(sizeof sizes / sizeof sizes[0])] + below(40);
                blocks.push_back(zstd ? zpool[below(zpool.size())] : make_lz4_block(target));
            }
        }
        uint64_t want[32];
        Bytes img = image_of(blocks, want);
        std::vector<size_t> header_at;
        for (size_t pos = 0, b = 0; b < blocks.size(); ++b) {
            header_at.push_back(pos);
            pos += 8 + blocks[b].payload.size();
        }
        const int how = damage(img, header_at);
        const bool valid = how == 0 && !from_golden;
        n_valid += valid;
        (zstd ? n_zstd : n_lz4) += 1;
        const auto t0 = std::chrono::steady_clock::now();
        calls += run_blockfile_input(zstd, img, valid, want, i, how);
        const double took = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (took > 0.5) std::fprintf(stderr, "fuzz_driver: input %ld (%s, %zu blocks, %zu bytes, damage %d%s) took %.1f s\n", i, zstd ? "zstd" : "lz4", blocks.size(), img.size(), how, from_golden ? ", reference-written" : "", took);
        if (i % 5000 == 4999) std::fprintf(stderr, "fuzz_driver: %ld inputs, %ld calls\n", i + 1, calls);
    }
    if (!g_path.empty()) std::remove(g_path.c_str());
    FLAGSTATS_hip_shutdown();
    if (g_refused >= 100) CHECK(g_refused_host > 0, "the refused decoded buffer never sent a file to the host threads (%ld calls)", g_refused);
    std::fprintf(stderr, "fuzz_driver: decoded buffer refused in %ld calls, %ld of them valid files that the host threads then took\n", g_refused, g_refused_host);
    std::printf("fuzz_driver: %ld inputs (%ld LZ4 block files, %ld Zstandard, %ld of them undamaged, %ld reference-written; %ld FLAG texts, %ld raw files), %ld calls: %s\n", inputs, n_lz4,
                n_zstd, n_valid, n_golden, n_text, n_raw, calls, g_fail ? "FAILED" : "all checks passed");
    return g_fail ? 1 : 0;
}
