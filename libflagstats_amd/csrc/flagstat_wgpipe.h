// flagstat_wgpipe.h -- what the workgroup decoders (flagstat_lz4_kernels.hip, flagstat_zstd_kernels.hip) share: the
// hand-over primitives between the waves of a workgroup (positions in LDS, polled with back-off, bounded), wave-wide
// scans, and the two back-end stages of the pipeline -- SCAN (markers -> one final source per output byte) and COPY
// (gather, write, flush).  Device code; included by the kernel translation units only.
//
// The LDS type a kernel passes in holds
//   uint8_t ring[kNR]; uint32_t mark[kMR]; uint32_t fsrc[kK];
//   uint32_t s_clr[kScan .. 7], d_op;   (16 or 32 bytes, aligned: read with one or two wg_ld4 by the kernel's own front end)
//   uint32_t s_done[kScan]; uint32_t c_ready, s_carry[8], err;   (and f_op when kPublishFlush)
// and the constants kNR, kMR, kK, kChunk, kFlush, kScan, kPublishFlush.
#ifndef FLAGSTAT_WGPIPE_H_
#define FLAGSTAT_WGPIPE_H_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace fsk {

constexpr uint32_t kWgSpinLimit = 1u << 19;        // polls before a wait gives up (a logic error must not hang the GPU)
constexpr uint32_t kMarkLiteral = 0x10000u;        // marker: a literal run starts here (matches: their distance, 1..65535)

__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wg_ld(const uint32_t* p)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// "data, then position": LDS operations of one wave execute in order, so the compiler barrier is the only fence needed
__device__ __forceinline__ void wg_st(uint32_t* p, uint32_t v)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// poll until cond() holds; false if the block failed (here or in another wave) or the wait ran out.  The naps grow:
// ten waiting waves per CU polling every ~150 cycles took the LDS pipeline away from the waves that had work.
template <class LDS, class F>
__device__ __forceinline__ bool wg_wait(LDS& L, F cond)
{
    for (uint32_t spins = 0;; ++spins) {
        if (cond()) break;
        if ((spins & 3u) == 3u && wg_ld(&L.err)) return false;
        if (spins > kWgSpinLimit) {
            wg_st(&L.err, 9u);
            return false;
        }
        if (spins < 2u)
            __builtin_amdgcn_s_sleep(2);
        else if (spins < 6u)
            __builtin_amdgcn_s_sleep(6);
        else
            __builtin_amdgcn_s_sleep(16);
    }
    asm volatile("" ::: "memory");
    return true;
}
template <bool PROF, class LDS, class F>
__device__ __forceinline__ bool wg_wait_timed(LDS& L, unsigned long long& t_wait, F cond)
{
    if (cond()) {
        asm volatile("" ::: "memory");
        return true;
    }
    const unsigned long long t0 = PROF ? __builtin_readcyclecounter() : 0ull;
    const bool ok = wg_wait(L, cond);
    if (PROF) t_wait += __builtin_readcyclecounter() - t0;
    return ok;
}

// four consecutive words with ONE read (a poll that read its words one by one, each with its own wait, cost ~200 cycles
// a word)
__device__ __forceinline__ uint4 wg_ld4(const uint32_t* p)
{
    asm volatile("" ::: "memory");
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = *reinterpret_cast<const volatile v4u*>(p);
    asm volatile("" ::: "memory");
    return make_uint4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                      __builtin_amdgcn_readfirstlane(v.w));
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return umin(umin(a, b), c); }

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x)
{
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xF, 0xF, false));  // row_shr:1
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xF, 0xF, false));  // row_shr:2
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xF, 0xF, false));  // row_shr:4
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xF, 0xF, false));  // row_shr:8
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xA, 0xF, false));  // row_bcast:15
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xC, 0xF, false));  // row_bcast:31
    return x;
}
// inclusive prefix maximum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x)
{
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xF, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xA, 0xF, false)));
    x = umax(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xC, 0xF, false)));
    return x;
}

// ---- scanners: markers -> one final source per output byte; scanner `which` takes chunks which, which + kScan, ...
// (a marker: 0, kMarkLiteral, or the distance 1..65535 of the match that starts at that byte)
template <bool PROF, class LDS, class FRONT>
__device__ void wgpipe_scan(LDS& L, const uint32_t oend, const uint32_t lane, const uint32_t which, FRONT frontier,
                            unsigned long long* __restrict__ tally)
{
    unsigned long long t_wait = 0, n_rounds = 0, n_inchunk = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    // frontier(): the position below which every marker and literal byte is in place
    uint32_t front = 0;
    for (uint32_t kc = which; kc * LDS::kChunk < oend; kc += LDS::kScan) {
        const uint32_t c = kc * LDS::kChunk;
        const uint32_t need = c + LDS::kChunk < oend ? c + LDS::kChunk : oend;
        const uint32_t cidx = c % LDS::kNR;  // ring index of the chunk's first byte
        if (__builtin_expect(front < need, 0)) {
            if (!wg_wait_timed<PROF>(L, t_wait, [&] {
                    front = frontier();
                    return front >= need;
                }))
                break;
        }
        const uint32_t mslot = c & (LDS::kMR - 1u);
        const uint4 mk = *reinterpret_cast<const uint4*>(&L.mark[mslot + 4u * lane]);
        asm volatile("" ::: "memory");
        *reinterpret_cast<uint4*>(&L.mark[mslot + 4u * lane]) = make_uint4(0u, 0u, 0u, 0u);
        wg_st(&L.s_clr[which], c + LDS::kScan * LDS::kChunk);
        const uint32_t r0 = 4u * lane;
        // "the last marker at or before this byte": keys grow with the position, so it is a maximum
        uint32_t k[4];
        k[0] = mk.x ? ((r0 + 1u) << 17) | mk.x : 0u;
        k[1] = mk.y ? ((r0 + 2u) << 17) | mk.y : k[0];
        k[2] = mk.z ? ((r0 + 3u) << 17) | mk.z : k[1];
        k[3] = mk.w ? ((r0 + 4u) << 17) | mk.w : k[2];
        const uint32_t upto = wave_scan_max(k[3]);
        const uint32_t before = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(upto), 0x138, 0xF, 0xF, false));  // wave_shr:1
        const uint32_t last = __builtin_amdgcn_readlane(upto, 63);
        // the marker that runs into this chunk comes from the scanner of chunk kc - 1, and ours goes to the next one
        uint32_t carry = 0;
        if (kc) {
            if (!wg_wait_timed<PROF>(L, t_wait, [&] { return wg_ld(&L.c_ready) >= kc; })) break;
            carry = wg_ld(&L.s_carry[kc & 7u]);
        }
        if (lane == 0u) L.s_carry[(kc + 1u) & 7u] = last ? (last & 0x1FFFFu) : carry;
        wg_st(&L.c_ready, kc + 1u);
        // Per byte: the last marker at or before it (a key in this lane, else the scan's value from the lanes before, else the
        // carry -- keys carry their position above bit 17, so "the last" is the maximum and the carry is below every key); its low
        // 16 bits are the distance, 0 for a literal run (kMarkLiteral) and for "nothing yet": such a byte is its own root.
        static_assert((kMarkLiteral & 0xFFFFu) == 0u, "a literal marker reads as distance 0");
        const uint32_t before2 = umax(before, carry);
        uint32_t ptr[4], ext[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t off = umax(k[j], before2) & 0xFFFFu;
            const uint32_t rel = r0 + static_cast<uint32_t>(j);
            const uint32_t t = rel - off;                        // (negative: the source lies before this chunk)
            ptr[j] = static_cast<int32_t>(t) >= 0 ? t : rel;     // inside the chunk: a pointer to chase; else a root
            // where a root's byte comes from: the ring at distance off (itself for a literal, which an emitter wrote)
            const uint32_t d = t + cidx;
            ext[j] = umin(d, d + LDS::kNR);                      // (d < 0 wraps to the ring's end)
        }
        uint32_t p4 = ptr[0] | (ptr[1] << 8) | (ptr[2] << 16) | (ptr[3] << 24);
        const uint32_t self4 = 0x03020100u + 0x04040404u * lane;
        const bool any_in = __builtin_amdgcn_ballot_w64(p4 != self4) != 0ull;
        if (any_in) {
            // pointers inside the chunk: chase to the roots, doubling (<= 255 hops -> <= 8 rounds), four byte indices a dword.
            // ds_bpermute takes the lane from address bits 7:2, so a byte index IS the address of the dword that holds it (and what
            // lies above the byte in the shifted dword is ignored); the four fetched dwords give up their bytes to two v_perm_b32.
            ++n_inchunk;
            for (int round = 0; round < 8; ++round) {
                const uint32_t t0 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(p4), static_cast<int>(p4)));
                const uint32_t t1 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(p4 >> 8), static_cast<int>(p4)));
                const uint32_t t2 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(p4 >> 16), static_cast<int>(p4)));
                const uint32_t t3 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(p4 >> 24), static_cast<int>(p4)));
                // selectors: byte 0 <- t0[p & 3], byte 1 <- t1[p' & 3] (the first operand's bytes are 4..7), 0x0C = a zero byte
                const uint32_t s01 = (p4 & 0x00000303u) | 0x0C0C0400u, s23 = (p4 & 0x03030000u) | 0x04000C0Cu;
                const uint32_t n4 = __builtin_amdgcn_perm(t1, t0, s01) | __builtin_amdgcn_perm(t3, t2, s23);
                const bool changed = n4 != p4;
                p4 = n4;
                if (PROF) ++n_rounds;
                if (!__builtin_amdgcn_ballot_w64(changed)) break;
            }
            ptr[0] = p4 & 255u;
            ptr[1] = (p4 >> 8) & 255u;
            ptr[2] = (p4 >> 16) & 255u;
            ptr[3] = p4 >> 24;
        }
        if (!wg_wait_timed<PROF>(L, t_wait, [&] { return c + LDS::kChunk <= wg_ld(&L.d_op) + LDS::kK; })) break;
        const uint32_t base = c & (LDS::kK - 1u);
        *reinterpret_cast<uint4*>(&L.fsrc[base + r0]) = make_uint4(ext[0], ext[1], ext[2], ext[3]);
        if (any_in) {
            const uint4 f = make_uint4(L.fsrc[base + ptr[0]], L.fsrc[base + ptr[1]], L.fsrc[base + ptr[2]], L.fsrc[base + ptr[3]]);
            asm volatile("" ::: "memory");  // (every lane's reads are one instruction each, all before this write)
            *reinterpret_cast<uint4*>(&L.fsrc[base + r0]) = f;
        }
        wg_st(&L.s_done[which], c + LDS::kChunk);
    }
    if (PROF && lane == 0u) {
        atomicAdd(&tally[8], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        atomicAdd(&tally[9], t_wait);
        atomicAdd(&tally[10], n_rounds);
        atomicAdd(&tally[11], n_inchunk);
    }
}

// ---- the last wave: gather, write, flush
template <bool PROF, class LDS>
__device__ void wgpipe_copy(LDS& L, uint8_t* __restrict__ dst, const uint32_t oend, const uint32_t lane,
                            unsigned long long* __restrict__ tally)
{
    static_assert(LDS::kFlush == 4u * LDS::kChunk && LDS::kNR % LDS::kFlush == 0u, "a flush unit is four chunks and never wraps");
    uint32_t cidx = 0, flushed = 0, fidx = 0;
    const uint32_t oend_even = oend & ~1u;  // an odd trailing byte of a block is dropped like the reference's N = size >> 1
    unsigned long long t_wait = 0, n_chunks = 0;
    const unsigned long long t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    bool ok = true;
    uint32_t sidx = 0;   // the scanner of the coming chunk: (c / kChunk) % kScan
    // One chunk: "is it scanned?", its final sources, the bytes behind them, the write, the new position -- three dependent
    // LDS round trips and as little else as possible: this wave is the one every byte of the block passes through in order,
    // and a taken branch costs it ~50 cycles, a scalar instruction 5-12.
    auto one_chunk = [&](const uint32_t c, const uint32_t at) {
        if (ok && __builtin_expect(wg_ld(&L.s_done[sidx]) < c + LDS::kChunk, 0))
            ok = wg_wait_timed<PROF>(L, t_wait, [&] { return wg_ld(&L.s_done[sidx]) >= c + LDS::kChunk; });
        if (ok) {
            const uint4 f = *reinterpret_cast<const uint4*>(&L.fsrc[(c & (LDS::kK - 1u)) + 4u * lane]);
            const uint32_t b0 = L.ring[f.x < LDS::kNR ? f.x : LDS::kNR - 1u], b1 = L.ring[f.y < LDS::kNR ? f.y : LDS::kNR - 1u];
            const uint32_t b2 = L.ring[f.z < LDS::kNR ? f.z : LDS::kNR - 1u], b3 = L.ring[f.w < LDS::kNR ? f.w : LDS::kNR - 1u];
            *reinterpret_cast<uint32_t*>(&L.ring[at + 4u * lane]) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            wg_st(&L.d_op, c + LDS::kChunk);
            sidx = sidx + 1u == LDS::kScan ? 0u : sidx + 1u;
        }
    };
    uint32_t c = 0;
    // whole flush units: four chunks, then the KiB they made goes to global memory
    for (; ok && c + LDS::kFlush <= oend_even; c += LDS::kFlush) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) one_chunk(c + k * LDS::kChunk, cidx + k * LDS::kChunk);
        if (!ok) break;
        n_chunks += 4u;
        if constexpr (LDS::kPublishFlush) {
            // what was flushed before has landed: readers of far matches (device-scope loads) may see it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_st(&L.f_op, flushed);
        }
        *reinterpret_cast<uint4*>(dst + flushed + lane * 16u) = *reinterpret_cast<const uint4*>(&L.ring[fidx + lane * 16u]);
        flushed += LDS::kFlush;
        fidx = fidx + LDS::kFlush == LDS::kNR ? 0u : fidx + LDS::kFlush;
        cidx = fidx;
    }
    // the chunks behind the last whole unit (at most four, the last one maybe short)
    for (; ok && c < oend; c += LDS::kChunk) {
        one_chunk(c, cidx);
        ++n_chunks;
        cidx = cidx + LDS::kChunk == LDS::kNR ? 0u : cidx + LDS::kChunk;
    }
    if (ok && !wg_ld(&L.err)) {
        // what is left in the ring: < 2 KiB, contiguous from fidx (a flush unit never wraps, the rest may)
        uint32_t o = flushed;
        for (; o + 1024u <= oend_even; o += 1024u) {
            uint32_t ri = fidx + (o - flushed) + lane * 16u;
            if (ri >= LDS::kNR) ri -= LDS::kNR;  // (16-byte groups stay whole: LDS::kNR and the group starts are multiples of 16)
            *reinterpret_cast<uint4*>(dst + o + lane * 16u) = *reinterpret_cast<const uint4*>(&L.ring[ri]);
        }
        for (uint32_t b = o + lane; b < oend_even; b += 64u) {
            uint32_t ri = fidx + (b - flushed);
            if (ri >= LDS::kNR) ri -= LDS::kNR;
            dst[b] = L.ring[ri];
        }
    }
    if (PROF && lane == 0u) {
        atomicAdd(&tally[12], static_cast<unsigned long long>(__builtin_readcyclecounter()) - t_begin);
        atomicAdd(&tally[13], t_wait);
        atomicAdd(&tally[14], n_chunks);
    }
}

}  // namespace fsk

#endif  // FLAGSTAT_WGPIPE_H_
