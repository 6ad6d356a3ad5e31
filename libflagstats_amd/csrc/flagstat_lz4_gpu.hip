// flagstat_lz4_gpu.hip -- EXPERIMENT: the LZ4 block decode of row f1 moved to the GPU.
//
// The reference's block reader decodes every block with liblz4's LZ4_decompress_safe on the host
// (benchmark/flagstats.cpp:311-316); this repo's product path (flagstat_blocks.hip) does the same on N host threads and
// is PCIe-bound on the DECODED bytes.  Sending the compressed bytes instead (4.2x fewer for NA12878-like flags) only pays
// if the GPU can decode fast enough, and an LZ4 block is one serial chain of ~156,000 sequences per 1,024,000-byte block.
// DESIGN.md (r02) rejected the idea on an estimate; this file is the measurement: ONE WAVE PER BLOCK,
//   * compressed bytes staged through a 4 KiB LDS window (coalesced 16-byte loads),
//   * the last 16 KiB of output kept in an LDS ring, so a match copy is ds_read -> ds_write for every offset up to
//     16,320; farther matches read the already flushed output back from global memory,
//   * the ring flushed to global memory 4 KiB at a time with coalesced 16-byte stores,
//   * every index masked or checked: a malformed block sets its status word and stops, it cannot fault.
// 20.5 KiB of LDS per wave = 7 waves per CU = 1792 blocks in flight on the chip.
// Entry: FLAGSTATS_hip_blockimage_lz4_gpu (H2D of the image, decode kernel, K1 over the decoded buffer, timings).
// Result (profiles/r03/gpu_lz4_*.log) decides whether it replaces the host pipeline; it is not wired into
// FLAGSTATS_hip_blockfile*.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/libflagstats_hip.h"
#include "flagstat_engine.h"

namespace fsk {

struct GpuBlock {
    uint64_t src_off;  // payload offset in the image
    uint64_t dst_off;  // offset in the decoded buffer (multiple of 16)
    uint32_t src_len;
    uint32_t dst_len;
};

// RING: bytes of recent output kept in LDS (matches up to RING - 64 back are LDS -> LDS); INWIN: staged input window.
// 16 KiB + 4 KiB = 7 waves per CU (1792 blocks in flight); 8 KiB + 1 KiB = 17 per CU (4352: a 4 GiB file's 4195 blocks
// all at once), at the price of more matches that reach behind the ring.
template <uint32_t RING, uint32_t INWIN>
__global__ __launch_bounds__(64) void lz4_decode_wave(const uint8_t* __restrict__ comp, const GpuBlock* __restrict__ blocks,
                                                     uint8_t* __restrict__ out, uint32_t* __restrict__ status,
                                                     unsigned long long* __restrict__ tally)
{
    constexpr uint32_t kRingMask = RING - 1, kFlush = RING / 4;
    __shared__ __attribute__((aligned(16))) uint8_t ring[RING];
    __shared__ __attribute__((aligned(16))) uint8_t inw[INWIN + 16];
    const GpuBlock b = blocks[blockIdx.x];
    const uint8_t* src = comp + b.src_off;
    uint8_t* dst = out + b.dst_off;
    const uint32_t iend = b.src_len, oend = b.dst_len;
    const uint32_t lane = threadIdx.x;
    uint32_t ip = 0, op = 0, in_base = 0, in_valid = 0, flushed = 0;
    uint32_t err = 0, nseq = 0, nfar = 0;

    // make inw[] cover [ip, ip + need) (need <= 80) unless the block ends first
    auto cover = [&](uint32_t need) {
        if (ip + need <= in_base + in_valid || in_base + in_valid >= iend) return;
        in_base = ip & ~15u;
        uint32_t n = iend - in_base;
        if (n > INWIN) n = INWIN;
        for (uint32_t k = 0; k < INWIN; k += 1024) {
            const uint32_t o = k + lane * 16;
            if (o < n) *reinterpret_cast<uint4*>(&inw[o]) = *reinterpret_cast<const uint4*>(src + in_base + o);  // image is padded by 64 B
        }
        in_valid = n;  // (one wave: LDS operations execute in program order, no barrier needed)
    };
    auto in_byte = [&](uint32_t pos) -> uint32_t {
        uint32_t i = pos - in_base;
        if (i > INWIN + 15) i = INWIN + 15;  // cannot happen after cover(); keeps a logic error inside the array
        return __builtin_amdgcn_readfirstlane(inw[i]);
    };
    // token and the two bytes behind it with ONE wait (the usual sequence of these streams has no literals, so they
    // are its offset)
    auto in_3bytes = [&](uint32_t pos) -> uint32_t {
        uint32_t i = pos - in_base;
        if (i > INWIN + 13) i = INWIN + 13;
        const uint32_t v = inw[i] | (static_cast<uint32_t>(inw[i + 1]) << 8) | (static_cast<uint32_t>(inw[i + 2]) << 16);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // write the finished part of the ring to global memory, a quarter of the ring at a time
    auto flush_to = [&](uint32_t upto) {
        while (upto - flushed >= kFlush) {
            for (uint32_t k = 0; k < kFlush; k += 1024) {
                const uint32_t o = flushed + k + lane * 16;
                *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(&ring[o & kRingMask]);
            }
            flushed += kFlush;
            // a far match may read these bytes back through another lane: they must have left this wave first
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    while (ip < iend) {
        cover(24);
        const uint32_t t3 = in_3bytes(ip);
        const uint32_t token = t3 & 255u;
        ++ip;
        ++nseq;
        // ---- literals
        uint32_t ll = token >> 4;
        if (ll == 15) {
            uint32_t e;
            do {
                cover(1);
                if (ip >= iend) { err = 1; break; }
                e = in_byte(ip);
                ++ip;
                ll += e;
            } while (e == 255);
            if (err) break;
        }
        if (ll > iend - ip || ll > oend - op) { err = 2; break; }
        const bool bare = ll == 0;
        while (ll) {
            const uint32_t n = ll < 64 ? ll : 64;
            cover(n);
            if (lane < n) ring[(op + lane) & kRingMask] = inw[ip + lane - in_base];
            ip += n;
            op += n;
            ll -= n;
            flush_to(op);
        }
        if (ip >= iend) break;  // the last sequence has no match
        // ---- match
        uint32_t off;
        if (bare) {
            off = t3 >> 8;  // already here
            if (ip + 2 > iend) { err = 3; break; }
        } else {
            cover(2);
            if (ip + 2 > iend) { err = 3; break; }
            off = in_byte(ip) | (in_byte(ip + 1) << 8);
        }
        ip += 2;
        uint32_t ml = token & 15u;
        if (ml == 15) {
            uint32_t e;
            do {
                cover(1);
                if (ip >= iend) { err = 4; break; }
                e = in_byte(ip);
                ++ip;
                ml += e;
            } while (e == 255);
            if (err) break;
        }
        ml += 4;
        if (off == 0 || off > op || ml > oend - op) { err = 5; break; }
        const bool near = off <= RING - 64;
        nfar += near ? 0u : 1u;
        while (ml) {
            const uint32_t n = ml < 64 ? ml : 64;
            // out[op + j] = out[op + j - off]; for off < n the source repeats with period off
            const uint32_t j = (off >= n) ? lane : lane % off;
            uint32_t v = 0;
            if (near) {
                if (lane < n) v = ring[(op - off + j) & kRingMask];
            } else {
                // farther back than the ring: already flushed (op - off + n <= flushed); device-scope load, past the L1
                if (lane < n) v = __hip_atomic_load(&dst[op - off + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane < n) ring[(op + lane) & kRingMask] = static_cast<uint8_t>(v);
            op += n;
            ml -= n;
            flush_to(op);
        }
    }
    if (!err && op != oend) err = 6;
    // tail of the ring.  An odd trailing byte of a block is dropped like the reference's N = size >> 1
    // (benchmark/flagstats.cpp:323): it stays zero in the padded slot, so the counting kernel sees no stray flag.
    if (!err) {
        for (uint32_t o = flushed + lane; o < (op & ~1u); o += 64) dst[o] = ring[o & kRingMask];
    }
    if (lane == 0) {
        status[blockIdx.x] = err;
        atomicAdd(&tally[0], static_cast<unsigned long long>(nseq));
        atomicAdd(&tally[1], static_cast<unsigned long long>(nfar));
    }
}

}  // namespace fsk

extern "C" int FLAGSTATS_hip_blockimage_lz4_gpu(const void* image, uint64_t bytes, uint64_t* out, FLAGSTATS_gpu_lz4_stats* stats)
{
    using fsint::fail_hip;
    using fsint::fail_text;
    if (!image || !out) return fail_text("NULL image or out");
    fsint::Engine* ep = fsint::default_engine();
    if (!ep) return -1;
    fsint::Engine& e = *ep;
    std::lock_guard<std::mutex> lk(e.mu);
    if (fsint::engine_alive(e)) return -1;
    fsint::DeviceGuard guard(e.device);
    if (!guard.ok()) return -1;
    // index: int32 uncompressed size, int32 compressed size, payload (benchmark/flagstats.cpp:119-138)
    const uint8_t* img = static_cast<const uint8_t*>(image);
    std::vector<fsk::GpuBlock> blocks;
    uint64_t pos = 0, dpos = 0, n_flags = 0;
    while (pos < bytes) {
        if (bytes - pos < 8) return fail_text("block image: truncated block header");
        int32_t us, cs;
        std::memcpy(&us, img + pos, 4);
        std::memcpy(&cs, img + pos + 4, 4);
        if (us < 0 || cs < 0 || static_cast<uint64_t>(cs) > bytes - pos - 8) return fail_text("block image: bad block header");
        blocks.push_back(fsk::GpuBlock{pos + 8, dpos, static_cast<uint32_t>(cs), static_cast<uint32_t>(us)});
        n_flags += static_cast<uint64_t>(us) >> 1;
        dpos += (static_cast<uint64_t>(us) + 15) & ~15ull;
        pos += 8 + static_cast<uint64_t>(cs);
    }
    if (blocks.empty()) return 0;
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    fsk::GpuBlock* d_blocks = nullptr;
    uint32_t* d_status = nullptr;
    unsigned long long* d_tally = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = 0;
    auto cleanup = [&] {
        (void)hipStreamSynchronize(e.stream[0]);
        if (d_comp) (void)hipFree(d_comp);
        if (d_out) (void)hipFree(d_out);
        if (d_blocks) (void)hipFree(d_blocks);
        if (d_status) (void)hipFree(d_status);
        if (d_tally) (void)hipFree(d_tally);
        for (hipEvent_t x : ev)
            if (x) (void)hipEventDestroy(x);
    };
#define LZG_TRY(expr)                            \
    do {                                         \
        hipError_t e_ = (expr);                  \
        if (e_ != hipSuccess) {                  \
            rc = fail_hip(#expr, e_);            \
            cleanup();                           \
            return rc;                           \
        }                                        \
    } while (0)
    hipStream_t s = e.stream[0];
    for (hipEvent_t& x : ev) LZG_TRY(hipEventCreate(&x));
    LZG_TRY(hipMalloc(&d_comp, bytes + 64));
    LZG_TRY(hipMalloc(&d_out, dpos + 16));
    LZG_TRY(hipMalloc(&d_blocks, blocks.size() * sizeof(fsk::GpuBlock)));
    LZG_TRY(hipMalloc(&d_status, blocks.size() * sizeof(uint32_t)));
    LZG_TRY(hipMemsetAsync(d_out, 0, dpos + 16, s));            // padding between blocks counts nothing
    LZG_TRY(hipMemsetAsync(d_status, 0xFF, blocks.size() * sizeof(uint32_t), s));
    LZG_TRY(hipMalloc(&d_tally, 16));
    LZG_TRY(hipMemsetAsync(d_tally, 0, 16, s));
    LZG_TRY(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(fsk::GpuBlock), hipMemcpyHostToDevice, s));
    LZG_TRY(hipEventRecord(ev[0], s));
    LZG_TRY(hipMemcpyAsync(d_comp, image, bytes, hipMemcpyHostToDevice, s));
    LZG_TRY(hipEventRecord(ev[1], s));
    // env FLAGSTATS_HIP_GPU_LZ4_RING = 16 (default) | 8: KiB of recent output per wave in LDS (see lz4_decode_wave)
    const char* rk = std::getenv("FLAGSTATS_HIP_GPU_LZ4_RING");
    const bool small_ring = rk && std::atoi(rk) == 8;
    if (small_ring)
        hipLaunchKernelGGL((fsk::lz4_decode_wave<8192, 1024>), dim3(static_cast<uint32_t>(blocks.size())), dim3(64), 0, s, d_comp,
                           d_blocks, d_out, d_status, d_tally);
    else
        hipLaunchKernelGGL((fsk::lz4_decode_wave<16384, 4096>), dim3(static_cast<uint32_t>(blocks.size())), dim3(64), 0, s, d_comp,
                           d_blocks, d_out, d_status, d_tally);
    LZG_TRY(hipGetLastError());
    LZG_TRY(hipEventRecord(ev[2], s));
    LZG_TRY(hipMemsetAsync(e.d_out[0], 0, 32 * sizeof(uint64_t), s));
    rc = fsint::count_device_async(e, reinterpret_cast<const uint16_t*>(d_out), dpos / 2, e.d_out[0], s, e.ws[0]);
    if (rc) {
        cleanup();
        return rc;
    }
    LZG_TRY(hipEventRecord(ev[3], s));
    LZG_TRY(hipMemcpyAsync(e.h_out, e.d_out[0], 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    unsigned long long tally[2] = {0, 0};
    LZG_TRY(hipMemcpyAsync(tally, d_tally, 16, hipMemcpyDeviceToHost, s));
    std::vector<uint32_t> st(blocks.size());
    LZG_TRY(hipMemcpyAsync(st.data(), d_status, st.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LZG_TRY(hipStreamSynchronize(s));
    uint64_t bad = 0;
    for (uint32_t x : st) bad += x != 0;
    float h2d = 0, dec = 0, cnt = 0;
    LZG_TRY(hipEventElapsedTime(&h2d, ev[0], ev[1]));
    LZG_TRY(hipEventElapsedTime(&dec, ev[1], ev[2]));
    LZG_TRY(hipEventElapsedTime(&cnt, ev[2], ev[3]));
    if (stats) {
        stats->n_blocks = blocks.size();
        stats->n_flags = n_flags;
        stats->bad_blocks = bad;
        stats->compressed_bytes = bytes;
        stats->decoded_bytes = dpos;
        stats->h2d_ms = h2d;
        stats->decode_ms = dec;
        stats->count_ms = cnt;
        stats->sequences = tally[0];
        stats->far_matches = tally[1];
        stats->ring_kib = small_ring ? 8 : 16;
    }
    if (!bad)
        for (int k = 0; k < 32; ++k) out[k] += e.h_out[k];
    cleanup();
    if (bad) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "GPU LZ4 decode: %llu malformed block(s)", static_cast<unsigned long long>(bad));
        return fail_text(buf);
    }
    return 0;
#undef LZG_TRY
}
