// Dependent-VALU latency on one wave: hipcc --offload-arch=gfx950 valu_latency.hip -o valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void chain(uint32_t* out, unsigned long long* cycles, int steps, uint32_t k)
{
    uint32_t a = threadIdx.x * 2654435761u + k, b = k * 7u + 1u;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) a = a + b;                                             // v_add
            if (MODE == 1) a = __builtin_amdgcn_alignbit(a, b, a & 31u);          // v_and + v_alignbit
            if (MODE == 2) a = 31u - static_cast<uint32_t>(__builtin_clz(a | 1u)) + b;   // v_or, v_ffbh, v_sub/xor, v_add
            if (MODE == 3) a = (a & 64u) ? a + b : a ^ b;                         // v_and, v_cmp, v_add, v_xor, v_cndmask
            if (MODE == 4) a = __builtin_amdgcn_ubfe(b, a & 31u, 5u) + a;         // v_and, v_bfe, v_add
            if (MODE == 5) a = (a << (b & 7u)) - b + (a >> 20);                   // shifts, sub, add
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int ops_per_iter)
{
    uint32_t* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256);
    hipMalloc(&cyc, 8);
    const int steps = 4000;
    hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, steps, 3u);
    hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, steps, 5u);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    std::printf("%-56s %.1f cycles per dependent group of %d op(s) = %.1f per op\n", what, double(h) / steps / 16, ops_per_iter, double(h) / steps / 16 / ops_per_iter);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<0>("v_add", 1);
    run<1>("v_and + v_alignbit", 2);
    run<2>("v_or + v_ffbh + v_sub + v_add", 4);
    run<3>("v_and + v_cmp + {v_add, v_xor} + v_cndmask", 4);
    run<4>("v_and + v_bfe + v_add", 3);
    run<5>("v_and + v_lshl + v_sub + v_lshr + v_add", 4);
    return 0;
}
