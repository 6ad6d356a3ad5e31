// How fast can a file in the page cache reach device memory?  (row f1, file mode of the GPU decoders: the compressed bytes
// must cross PCIe; in r04 every byte was first copied by a CPU -- 16 parallel preads into three pinned spans, 27-33 GB/s --
// and file mode was 1.5x slower than image mode on the same bytes.)
//   hipcc --offload-arch=gfx950 -O2 -pthread file_h2d.hip -o file_h2d ; ./file_h2d FILE [threads=16] [span MiB=64] [reps=3]
// Ways tried, each end to end (open -> every byte on the device, hipStreamSynchronize), file already in the page cache:
//   pread     N threads pread shares of a span into one of three pinned spans, the copy of span i overlaps the reads of i + 1
//   mmap      mmap the file, hipMemcpyAsync straight out of the mapping (the runtime pins what it copies)
//   populate  the same, after N threads have populated the mapping's page tables (MADV_POPULATE_READ) span by span
//   register  mmap, N threads hipHostRegister shares of a span (page-locked + mapped into the device), one hipMemcpyAsync per
//             span out of the registered range, everything unregistered at the end (timed separately: it can run after the
//             counters have been returned)
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif

#define CK(x)                                                                                \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                                    \
        }                                                                                    \
    } while (0)

using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

template <class F>
static void parallel(int n, F f)
{
    std::vector<std::thread> t;
    for (int i = 1; i < n; ++i) t.emplace_back(f, i);
    f(0);
    for (auto& x : t) x.join();
}

__global__ void checksum(const uint32_t* p, uint64_t n, unsigned long long* out)
{
    unsigned long long s = 0;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) s += p[i];
    atomicAdd(out, s);
}

// the job's CPU quota at work: periods in which the cgroup was throttled, and for how long (cgroup v2 cpu.stat)
static void throttle(const char* when)
{
    FILE* f = std::fopen("/sys/fs/cgroup/cpu.stat", "r");
    if (!f) return;
    char line[256];
    unsigned long long nr = 0, us = 0, periods = 0;
    while (std::fgets(line, sizeof line, f)) {
        (void)std::sscanf(line, "nr_throttled %llu", &nr);
        (void)std::sscanf(line, "throttled_usec %llu", &us);
        (void)std::sscanf(line, "nr_periods %llu", &periods);
    }
    std::fclose(f);
    std::printf("cgroup %s: %llu periods, %llu throttled, %.1f ms throttled in all\n", when, periods, nr, us / 1e3);
}

int main(int argc, char** argv)
{
    if (argc < 2) return std::fprintf(stderr, "usage: %s FILE [threads] [span MiB] [reps]\n", argv[0]), 2;
    const int nthr = argc > 2 ? std::atoi(argv[2]) : 16;
    const uint64_t span = (argc > 3 ? std::strtoull(argv[3], nullptr, 0) : 64) << 20;
    const int reps = argc > 4 ? std::atoi(argv[4]) : 3;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb)) return std::perror("open"), 1;
    const uint64_t bytes = static_cast<uint64_t>(sb.st_size) & ~4095ull;
    CK(hipSetDevice(0));
    uint8_t* d = nullptr;
    CK(hipMalloc(&d, bytes + 4096));
    unsigned long long* d_sum = nullptr;
    CK(hipMalloc(&d_sum, 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto device_sum = [&] {
        unsigned long long h = 0;
        CK(hipMemsetAsync(d_sum, 0, 8, s));
        hipLaunchKernelGGL(checksum, dim3(1024), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(d), bytes / 4, d_sum);
        CK(hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        return h;
    };
    std::printf("file %s: %.1f MiB, %d threads, spans of %llu MiB\n", argv[1], bytes / 1048576.0, nthr, (unsigned long long)(span >> 20));
    unsigned long long want = 0;
    throttle("at start");

    // ---- pread into three pinned spans (what r04 ships)
    {
        uint8_t* pin[3];
        auto t0 = clk::now();
        for (auto& p : pin) CK(hipHostMalloc(&p, span, hipHostMallocDefault));
        std::printf("pread:    hipHostMalloc of 3 x %llu MiB pinned: %.1f ms (once per process)\n", (unsigned long long)(span >> 20), ms(t0, clk::now()));
        hipEvent_t freed[3];
        for (auto& e : freed) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
        for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(d, 0, bytes, s));
            CK(hipStreamSynchronize(s));
            t0 = clk::now();
            uint64_t k = 0;
            for (uint64_t at = 0; at < bytes; at += span, ++k) {
                const uint64_t len = std::min(span, bytes - at);
                if (k >= 3) CK(hipEventSynchronize(freed[k % 3]));
                uint8_t* b = pin[k % 3];
                parallel(nthr, [&](int t) {
                    const uint64_t share = ((len + nthr - 1) / nthr + 4095) & ~4095ull;
                    uint64_t o = share * t;
                    const uint64_t end = std::min(len, o + share);
                    while (o < end) {
                        ssize_t got = pread(fd, b + o, end - o, at + o);
                        if (got <= 0) std::exit(3);
                        o += got;
                    }
                });
                CK(hipMemcpyAsync(d + at, b, len, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(freed[k % 3], s));
            }
            CK(hipStreamSynchronize(s));
            const double t = ms(t0, clk::now());
            want = device_sum();
            std::printf("pread:    %.2f ms = %.1f GB/s\n", t, bytes / t / 1e6);
        }
        for (auto& p : pin) CK(hipHostFree(p));
    }
    // ---- mmap, the runtime pins what it copies; with and without populating the page tables first
    for (int populate = 0; populate < 2; ++populate) {
        for (int r = 0; r < 1; ++r) {
            CK(hipMemsetAsync(d, 0, bytes, s));
            CK(hipStreamSynchronize(s));
            auto t0 = clk::now();
            uint8_t* m = static_cast<uint8_t*>(mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0));
            if (m == MAP_FAILED) return std::perror("mmap"), 1;
            double t_pop = 0;
            for (uint64_t at = 0; at < bytes; at += span) {
                const uint64_t len = std::min(span, bytes - at);
                if (populate) {
                    auto p0 = clk::now();
                    parallel(nthr, [&](int t) {
                        const uint64_t share = ((len + nthr - 1) / nthr + 4095) & ~4095ull;
                        const uint64_t o = share * t;
                        if (o < len && madvise(m + at + o, std::min(share, len - o), MADV_POPULATE_READ) != 0) {
                            volatile uint8_t sink = 0;
                            for (uint64_t q = o; q < std::min(len, o + share); q += 4096) sink += m[at + q];
                        }
                    });
                    t_pop += ms(p0, clk::now());
                }
                CK(hipMemcpyAsync(d + at, m + at, len, hipMemcpyHostToDevice, s));
            }
            auto t1 = clk::now();
            CK(hipStreamSynchronize(s));
            auto t2 = clk::now();
            munmap(m, bytes);
            const double t = ms(t0, clk::now());
            const bool ok = device_sum() == want;
            std::printf("%s %.2f ms = %.1f GB/s (queued after %.2f ms%s, synchronised after %.2f, munmap %.2f)%s\n", populate ? "populate:" : "mmap:    ", t, bytes / t / 1e6,
                        ms(t0, t1), populate ? (", of which populating " + std::to_string(t_pop) + " ms").c_str() : "", ms(t0, t2), ms(t2, clk::now()) , ok ? "" : "  WRONG BYTES");
        }
    }
    // ---- mmap + parallel hipHostRegister of every span, DMA out of the registered range
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(d, 0, bytes, s));
        CK(hipStreamSynchronize(s));
        auto t0 = clk::now();
        uint8_t* m = static_cast<uint8_t*>(mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0));
        if (m == MAP_FAILED) return std::perror("mmap"), 1;
        double t_reg = 0;
        std::atomic<int> bad{0};
        for (uint64_t at = 0; at < bytes; at += span) {
            const uint64_t len = std::min(span, bytes - at);
            auto p0 = clk::now();
            parallel(nthr, [&](int t) {
                const uint64_t share = ((len + nthr - 1) / nthr + 4095) & ~4095ull;
                const uint64_t o = share * t;
                if (o < len && hipHostRegister(m + at + o, std::min(share, len - o), hipHostRegisterDefault | hipHostRegisterReadOnly) != hipSuccess) {
                    (void)hipGetLastError();
                    if (hipHostRegister(m + at + o, std::min(share, len - o), hipHostRegisterDefault) != hipSuccess) ++bad;
                }
            });
            t_reg += ms(p0, clk::now());
            if (bad.load()) break;
            {   // one copy per registered share: a copy may not span two registrations
                const uint64_t share = ((len + nthr - 1) / nthr + 4095) & ~4095ull;
                for (uint64_t o = 0; o < len; o += share) CK(hipMemcpyAsync(d + at + o, m + at + o, std::min(share, len - o), hipMemcpyHostToDevice, s));
            }
        }
        if (bad.load()) {
            std::printf("register: hipHostRegister refuses the file mapping (%s)\n", hipGetErrorString(hipGetLastError()));
            munmap(m, bytes);
            break;
        }
        auto t1 = clk::now();
        CK(hipStreamSynchronize(s));
        auto t2 = clk::now();
        for (uint64_t at = 0; at < bytes; at += span) {
            const uint64_t len = std::min(span, bytes - at);
            parallel(nthr, [&](int t) {
                const uint64_t share = ((len + nthr - 1) / nthr + 4095) & ~4095ull;
                const uint64_t o = share * t;
                if (o < len) (void)hipHostUnregister(m + at + o);
            });
        }
        auto t3 = clk::now();
        munmap(m, bytes);
        const bool ok = device_sum() == want;
        std::printf("register: %.2f ms = %.1f GB/s to the last byte on the device (registering %.2f ms of it, queued after %.2f); unregister %.2f ms, munmap %.2f%s\n", ms(t0, t2),
                    bytes / ms(t0, t2) / 1e6, t_reg, ms(t0, t1), ms(t2, t3), ms(t3, clk::now()), ok ? "" : "  WRONG BYTES");
    }
    // ---- pread STRAIGHT into device memory through the PCIe BAR (fine-grained device memory the host can address): the kernel's
    //      copy_to_user is the only copy, no pinned spans, no DMA
    {
        int large_bar = 0;
        (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
        void* fg = nullptr;
        auto t0 = clk::now();
        if (large_bar && hipExtMallocWithFlags(&fg, bytes + 4096, hipDeviceMallocFinegrained) == hipSuccess) {
            std::printf("bar:      hipExtMallocWithFlags(fine-grained, %.0f MiB): %.2f ms\n", bytes / 1048576.0, ms(t0, clk::now()));
            uint8_t* b = static_cast<uint8_t*>(fg);
            for (int r = 0; r < reps; ++r) {
                CK(hipMemsetAsync(fg, 0, bytes, s));
                CK(hipStreamSynchronize(s));
                t0 = clk::now();
                std::atomic<int> short_reads{0};
                parallel(nthr, [&](int t) {
                    // every thread walks the file in 1 MiB slices, round robin: early parts of the file are complete early
                    const uint64_t slice = 1ull << 20;
                    for (uint64_t at = slice * t; at < bytes; at += slice * nthr) {
                        uint64_t o = at;
                        const uint64_t end = std::min(bytes, at + slice);
                        while (o < end) {
                            ssize_t got = pread(fd, b + o, end - o, o);
                            if (got <= 0) { ++short_reads; return; }
                            o += got;
                        }
                    }
                    __builtin_ia32_sfence();
                });
                const double t = ms(t0, clk::now());
                CK(hipMemcpyAsync(d, fg, bytes, hipMemcpyDeviceToDevice, s));  // (only so that the checksum kernel reads the same buffer)
                const bool ok = device_sum() == want && !short_reads.load();
                std::printf("bar:      %.2f ms = %.1f GB/s (pread into the BAR mapping, %d threads)%s\n", t, bytes / t / 1e6, nthr, ok ? "" : "  WRONG BYTES / pread refused");
            }
            CK(hipFree(fg));
        } else {
            (void)hipGetLastError();
            std::printf("bar:      no large BAR / no fine-grained allocation\n");
        }
    }
    // ---- what the first call of a process pays: pinned spans one after the other and three at a time, streams
    {
        uint8_t* pin[3] = {nullptr, nullptr, nullptr};
        auto t0 = clk::now();
        parallel(3, [&](int t) { (void)hipHostMalloc(&pin[t], span, hipHostMallocDefault); });
        std::printf("alloc:    3 x %llu MiB pinned on three threads at once: %.1f ms\n", (unsigned long long)(span >> 20), ms(t0, clk::now()));
        for (auto& p : pin) if (p) CK(hipHostFree(p));
        t0 = clk::now();
        uint8_t* one = nullptr;
        CK(hipHostMalloc(&one, 3 * span, hipHostMallocDefault));
        std::printf("alloc:    one pinned allocation of %llu MiB: %.1f ms\n", (unsigned long long)(3 * span >> 20), ms(t0, clk::now()));
        CK(hipHostFree(one));
        // transparent huge pages + hipHostRegister instead of hipHostMalloc: 32x fewer pages to lock and map
        for (int huge = 0; huge < 2; ++huge) {
            const size_t bytes_r = 3 * span;
            t0 = clk::now();
            void* raw = mmap(nullptr, bytes_r + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (raw == MAP_FAILED) break;
            uint8_t* base = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(raw) + (2u << 20) - 1) & ~static_cast<uintptr_t>((2u << 20) - 1));
            if (huge) (void)madvise(base, bytes_r, 14 /* MADV_HUGEPAGE */);
            parallel(nthr, [&](int t) {
                const size_t share = ((bytes_r + nthr - 1) / nthr + 4095) & ~size_t(4095);
                const size_t o = share * t;
                if (o < bytes_r) std::memset(base + o, 0, std::min(share, bytes_r - o));
            });
            const double t_touch = ms(t0, clk::now());
            auto t1 = clk::now();
            const hipError_t er = hipHostRegister(base, bytes_r, hipHostRegisterDefault);
            const double t_reg = ms(t1, clk::now());
            if (er != hipSuccess) {
                (void)hipGetLastError();
                std::printf("alloc:    %s + hipHostRegister: refused\n", huge ? "huge pages" : "4 KiB pages");
            } else {
                auto t2 = clk::now();
                CK(hipMemcpyAsync(d, base, bytes_r, hipMemcpyHostToDevice, s));
                CK(hipStreamSynchronize(s));
                const double t_copy = ms(t2, clk::now());
                auto t3 = clk::now();
                CK(hipHostUnregister(base));
                std::printf("alloc:    %llu MiB anonymous memory (%s), touched by %d threads in %.2f ms, hipHostRegister %.2f ms, first copy out of it %.1f GB/s, unregister %.2f ms\n",
                            (unsigned long long)(bytes_r >> 20), huge ? "MADV_HUGEPAGE" : "4 KiB pages", nthr, t_touch, t_reg, bytes_r / t_copy / 1e6, ms(t3, clk::now()));
            }
            munmap(raw, bytes_r + (2u << 20));
        }
        t0 = clk::now();
        hipStream_t x[2];
        for (auto& q : x) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        const double t_create = ms(t0, clk::now());
        t0 = clk::now();
        for (auto& q : x) {
            CK(hipMemsetAsync(d_sum, 0, 8, q));
            CK(hipStreamSynchronize(q));
        }
        std::printf("alloc:    two streams created in %.2f ms, first use of both %.2f ms\n", t_create, ms(t0, clk::now()));
        t0 = clk::now();
        hipEvent_t ev[75];
        for (auto& q : ev) CK(hipEventCreateWithFlags(&q, hipEventDisableTiming));
        std::printf("alloc:    75 events created in %.2f ms\n", ms(t0, clk::now()));
        for (auto& q : ev) CK(hipEventDestroy(q));
        for (auto& q : x) CK(hipStreamDestroy(q));
    }
    // ---- for scale: the same bytes out of anonymous memory (image mode) and out of pinned memory
    {
        std::vector<uint8_t> img(bytes);
        if (pread(fd, img.data(), bytes, 0) != static_cast<ssize_t>(bytes)) { /* large preads may be short */
            uint64_t o = 0;
            while (o < bytes) {
                ssize_t got = pread(fd, img.data() + o, bytes - o, o);
                if (got <= 0) break;
                o += got;
            }
        }
        for (int r = 0; r < reps; ++r) {
            auto t0 = clk::now();
            for (uint64_t at = 0; at < bytes; at += span) CK(hipMemcpyAsync(d + at, img.data() + at, std::min(span, bytes - at), hipMemcpyHostToDevice, s));
            auto t1 = clk::now();
            CK(hipStreamSynchronize(s));
            const double t = ms(t0, clk::now());
            std::printf("image:    %.2f ms = %.1f GB/s (anonymous memory, queued after %.2f ms)\n", t, bytes / t / 1e6, ms(t0, t1));
        }
        uint8_t* pin = nullptr;
        if (hipHostMalloc(&pin, bytes, hipHostMallocDefault) == hipSuccess) {
            std::memcpy(pin, img.data(), bytes);
            for (int r = 0; r < reps; ++r) {
                auto t0 = clk::now();
                CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, s));
                CK(hipStreamSynchronize(s));
                const double t = ms(t0, clk::now());
                std::printf("pinned:   %.2f ms = %.1f GB/s (one copy out of page-locked memory: the link)\n", t, bytes / t / 1e6);
            }
            CK(hipHostFree(pin));
        }
    }
    throttle("at end");
    close(fd);
    return 0;
}
