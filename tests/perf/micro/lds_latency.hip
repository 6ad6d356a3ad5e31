// Dependent-LDS-read latency on one wave (what a step of zstd_chain waits for): hipcc --offload-arch=gfx950 lds_latency.hip -o lds_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ uint32_t rd16(const uint8_t* p, uint32_t at)
{
    return *reinterpret_cast<const uint16_t*>(p + at);
}

template <int MODE>
__global__ void chase(uint32_t* out, unsigned long long* cycles, int steps)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384 + 64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 16384u / 4u; i += 64u) reinterpret_cast<uint32_t*>(lds)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    uint32_t a = lane * 52u, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {            // one aligned dword
            a = (*reinterpret_cast<const uint32_t*>(lds + (a & 0x3FFCu))) & 0x3FFFu;
        } else if (MODE == 1) {     // one ushort
            a = rd16(lds, a & 0x3FFEu) & 0x3FFFu;
        } else if (MODE == 2) {     // three ushorts + 16 unaligned bytes, all issued together (the chain's step)
            const uint32_t x = rd16(lds, a & 0x3FFEu), y = rd16(lds, (a * 3u) & 0x3FFEu), z = rd16(lds, (a * 5u) & 0x3FFEu);
            typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(1)));
            const v4u d = *reinterpret_cast<const v4u*>(lds + (((a * 7u) | 1u) & 0x3FFFu));
            a = (x + y + z + ((d.x >> 3) ^ (d.y >> 9) ^ (d.z >> 15) ^ (d.w >> 17))) & 0x3FFFu;
        } else if (MODE == 3) {     // three ushorts + 16 ALIGNED bytes
            const uint32_t x = rd16(lds, a & 0x3FFEu), y = rd16(lds, (a * 3u) & 0x3FFEu), z = rd16(lds, (a * 5u) & 0x3FFEu);
            const uint4 d = *reinterpret_cast<const uint4*>(lds + ((a * 7u) & 0x3FF0u));
            a = (x + y + z + ((d.x >> 3) ^ (d.y >> 9) ^ (d.z >> 15) ^ (d.w >> 17))) & 0x3FFFu;
        } else if (MODE == 4) {     // three ushorts only
            const uint32_t x = rd16(lds, a & 0x3FFEu), y = rd16(lds, (a * 3u) & 0x3FFEu), z = rd16(lds, (a * 5u) & 0x3FFEu);
            a = (x + y + z) & 0x3FFFu;
        } else if (MODE == 5) {     // 16 bytes at an odd address only
            typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(1)));
            const v4u d = *reinterpret_cast<const v4u*>(lds + (((a * 7u) | 1u) & 0x3FFFu));
            a = ((d.x >> 3) ^ (d.y >> 9) ^ (d.z >> 15) ^ (d.w >> 17)) & 0x3FFFu;
        } else {                    // three ushorts + five dwords at 4-byte alignment
            const uint32_t x = rd16(lds, a & 0x3FFEu), y = rd16(lds, (a * 3u) & 0x3FFEu), z = rd16(lds, (a * 5u) & 0x3FFEu);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(lds + ((a * 7u) & 0x3FFCu));
            a = (x + y + z + ((q[0] >> 3) ^ (q[1] >> 9) ^ (q[2] >> 15) ^ (q[3] >> 17) ^ (q[4] >> 5))) & 0x3FFFu;
        }
        acc += a;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int blocks, int lanes)
{
    uint32_t* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * 4);
    hipMalloc(&cyc, blocks * 8);
    const int steps = 20000;
    hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(lanes), 0, 0, out, cyc, steps);
    hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(lanes), 0, 0, out, cyc, steps);
    hipDeviceSynchronize();
    unsigned long long h[4096];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += h[i];
    std::printf("%-64s %4d workgroups of %2d lanes: %.0f cycles per dependent step\n", what, blocks, lanes, sum / blocks / steps);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int blocks : {1, 768, 2304}) {
        run<0>("one aligned dword", blocks, 64);
        run<1>("one ushort", blocks, 64);
        run<4>("three ushorts", blocks, 64);
        run<5>("16 bytes at an odd address", blocks, 64);
        run<2>("three ushorts + 16 bytes at an odd address (the chain's step)", blocks, 64);
        run<3>("three ushorts + 16 aligned bytes", blocks, 64);
        run<6>("three ushorts + five dwords at 4-byte alignment", blocks, 64);
    }
    return 0;
}
