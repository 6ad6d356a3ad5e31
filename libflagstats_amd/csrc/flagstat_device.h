// flagstat_device.h -- device-side helpers shared by the gfx950 kernels.
#ifndef FLAGSTAT_DEVICE_H_
#define FLAGSTAT_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsk {

__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi, lo, sel);  // v_perm_b32: bytes 0-3 = lo, 4-7 = hi
}

// (a << 1) | b and (a & mask) | c as the single VALU instructions they are on gfx950
__device__ __forceinline__ uint32_t lshl1_or(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c)
{
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(c));
    return r;
}

// carry-save adder on 32 one-bit columns: 2 VALU ops on gfx950 (v_bitop3_b32)
__device__ __forceinline__ void csa(uint32_t& carry, uint32_t& sum, uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t s = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // a ^ b ^ c
    const uint32_t k = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);  // majority
    sum = s;
    carry = k;
}

// Horner over planes from the heaviest down: acc = 2*acc + popcount(plane & mask)
__device__ __forceinline__ uint32_t hstep(uint32_t acc, uint32_t plane, uint32_t mask, bool dbl)
{
    return __builtin_popcount(plane & mask) + (dbl ? (acc << 1) : acc);
}

// Sum of x over the 64 lanes of a wave, valid in lane 63 only.  Six DPP adds on the VALU (quad
// swaps, half-row and row mirrors, then the two row broadcasts), no LDS round trips: the
// ds_bpermute butterfly __shfl_xor compiles to waits out the LDS latency at every one of its
// 6 x 21 steps, which was 2.5 us of every launch (profiles/r02/launch_anatomy.log).
// Every lane of the wave must be active (the kernels call it outside any divergent region).
__device__ __forceinline__ uint32_t wave_sum_lane63(uint32_t x)
{
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x141, 0xF, 0xF, false));  // row_half_mirror
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x140, 0xF, 0xF, false));  // row_mirror: every lane = its row's sum
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
    return x;
}

// The array is addressed on the 16-byte grid of its aligned-down base `a0`:
// vector j holds flag positions [8j, 8j+8); positions in [lo, hi) are the
// caller's flags, everything else reads as zero (a zero flag counts nothing).
__device__ __forceinline__ uint4 load_guarded(const uint4* __restrict__ a0, uint64_t j, uint64_t lo, uint64_t hi)
{
    const uint64_t f0 = j * 8;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (f0 >= lo && f0 + 8 <= hi) return a0[j];
    if (f0 + 8 <= lo || f0 >= hi) return r;
    const uint16_t* p = reinterpret_cast<const uint16_t*>(a0 + j);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t f = f0 + e;
        if (f >= lo && f < hi) w[e >> 1] |= static_cast<uint32_t>(p[e]) << (16 * (e & 1));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool NT>
__device__ __forceinline__ uint4 load_vec(const uint4* __restrict__ p)
{
    if constexpr (NT) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}

// Copy a vector out with real v_movs at this point of the instruction stream (the registers it
// lived in are about to be re-targeted by an in-flight load; see K1's ROLL path).
__device__ __forceinline__ uint4 copy_out(const uint4& o)
{
    uint4 x;
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(x.x), "=&v"(x.y), "=&v"(x.z), "=&v"(x.w)
                 : "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
    return x;
}

}  // namespace fsk

#endif
