// flagstat_probe.hip -- measurement only: a parameterised read-only kernel used to map out which
// access pattern the chip reads fastest (tools/probe_sweep.py).  Not on any product path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flagstat_device.h"

namespace fsk {

// MODE 0: grid-stride over steps of blockDim*UNROLL vectors, threads interleaved (u*blockDim + tid)
// MODE 1: every workgroup streams its own contiguous region of nvec/gridDim vectors
template <int UNROLL, bool NT, int MODE>
__global__ void read_probe2(const uint4* __restrict__ a0, uint64_t nvec, uint32_t* __restrict__ sink)
{
    const uint64_t step = static_cast<uint64_t>(blockDim.x) * UNROLL;
    uint64_t begin, end, stride;
    if (MODE == 0) {
        begin = blockIdx.x * step;
        end = nvec - (nvec % step);
        stride = gridDim.x * step;
    } else {
        const uint64_t per = (nvec / gridDim.x) - ((nvec / gridDim.x) % step);
        begin = blockIdx.x * per;
        end = begin + per;
        stride = step;
    }
    uint32_t acc = 0;
    for (uint64_t j = begin; j < end; j += stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = load_vec<NT>(a0 + j + static_cast<uint64_t>(u) * blockDim.x + threadIdx.x);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

}  // namespace fsk

template <int UNROLL>
static void launch_u(int mode, int nt, uint32_t grid, uint32_t threads, const uint4* p, uint64_t nvec, uint32_t* sink,
                     hipStream_t s)
{
    if (mode == 0) {
        if (nt) hipLaunchKernelGGL((fsk::read_probe2<UNROLL, true, 0>), dim3(grid), dim3(threads), 0, s, p, nvec, sink);
        else hipLaunchKernelGGL((fsk::read_probe2<UNROLL, false, 0>), dim3(grid), dim3(threads), 0, s, p, nvec, sink);
    } else {
        if (nt) hipLaunchKernelGGL((fsk::read_probe2<UNROLL, true, 1>), dim3(grid), dim3(threads), 0, s, p, nvec, sink);
        else hipLaunchKernelGGL((fsk::read_probe2<UNROLL, false, 1>), dim3(grid), dim3(threads), 0, s, p, nvec, sink);
    }
}

extern "C" hipError_t fsk_read_probe2(const void* d_buf, uint64_t bytes, int mode, int unroll, uint32_t threads,
                                      uint32_t grid, int nt, uint32_t* d_sink, hipStream_t stream)
{
    if ((reinterpret_cast<uintptr_t>(d_buf) & 15u) || grid == 0 || threads == 0 || threads > 1024 || (threads & 63u))
        return hipErrorInvalidValue;
    const uint4* p = reinterpret_cast<const uint4*>(d_buf);
    const uint64_t nvec = bytes / 16;
    switch (unroll) {
    case 2: launch_u<2>(mode, nt, grid, threads, p, nvec, d_sink, stream); break;
    case 4: launch_u<4>(mode, nt, grid, threads, p, nvec, d_sink, stream); break;
    case 8: launch_u<8>(mode, nt, grid, threads, p, nvec, d_sink, stream); break;
    case 16: launch_u<16>(mode, nt, grid, threads, p, nvec, d_sink, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}


// ------------------------------------------------------------------ cache-policy probe (measurement only)
// The fastest pattern of the sweep (384 threads x 4 vectors, grid-stride) with the load's cache-policy bits spelled out:
// POLICY 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt, 6 sc0, 7 sc0 nt (tools/policy_probe.py).
namespace fsk {
template <int POLICY>
__device__ __forceinline__ uint4 load_policy(const uint4* p)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    if constexpr (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int POLICY>
__global__ __launch_bounds__(384) void read_probe_policy(const uint4* __restrict__ a0, uint64_t nvec, uint32_t* __restrict__ sink)
{
    const uint64_t step = 384ull * 4;
    uint32_t acc = 0;
    for (uint64_t j = blockIdx.x * step; j + step <= nvec; j += gridDim.x * step) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load_policy<POLICY>(a0 + j + static_cast<uint64_t>(u) * 384 + threadIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loads are asm: the compiler does not wait for them
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}
}  // namespace fsk

extern "C" hipError_t fsk_read_probe_policy(const void* d_buf, uint64_t bytes, int policy, uint32_t grid, uint32_t* d_sink, hipStream_t s)
{
    if ((reinterpret_cast<uintptr_t>(d_buf) & 15u) || grid == 0) return hipErrorInvalidValue;
    const uint4* p = reinterpret_cast<const uint4*>(d_buf);
    const uint64_t nvec = bytes / 16;
    switch (policy) {
    case 0: hipLaunchKernelGGL(fsk::read_probe_policy<0>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    case 1: hipLaunchKernelGGL(fsk::read_probe_policy<1>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    case 2: hipLaunchKernelGGL(fsk::read_probe_policy<2>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    case 3: hipLaunchKernelGGL(fsk::read_probe_policy<3>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    case 4: hipLaunchKernelGGL(fsk::read_probe_policy<4>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    case 5: hipLaunchKernelGGL(fsk::read_probe_policy<5>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    case 6: hipLaunchKernelGGL(fsk::read_probe_policy<6>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    case 7: hipLaunchKernelGGL(fsk::read_probe_policy<7>, dim3(grid), dim3(384), 0, s, p, nvec, d_sink); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ shader clock under load
// One wave per workgroup, `grid` workgroups (one per XCD is enough): samples the shader-clock counter (s_memtime,
// clock64) and the constant 100 MHz reference counter (s_memrealtime, wall_clock64) around `ticks` reference ticks of
// spinning.  Launched on a side stream while K1 launches run on another, cycles / ticks x 100 MHz is the shader
// clock the chip sustained UNDER that load -- the number sysfs does not give reliably (bench.py: roofline.sclk_mhz).
// The exit condition is the reference clock, which always advances.
namespace fsk {
__global__ void clock_probe(uint64_t* __restrict__ out, uint64_t ticks)
{
    const uint64_t w0 = wall_clock64();
    const uint64_t c0 = clock64();
    uint64_t w1 = w0;
    while (w1 - w0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        w1 = wall_clock64();
    }
    const uint64_t c1 = clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = w1 - w0;
    }
}
}  // namespace fsk

extern "C" hipError_t fsk_clock_probe(uint64_t* d_out, uint32_t grid, uint64_t ticks, hipStream_t stream)
{
    if (!d_out || grid == 0 || grid > 64 || ticks == 0 || ticks > 100000000ull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fsk::clock_probe, dim3(grid), dim3(64), 0, stream, d_out, ticks);
    return hipGetLastError();
}
