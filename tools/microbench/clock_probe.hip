// Shader clock actually held by a VALU loop at different occupancies (gfx950): s_memtime (shader cycles) against
// s_memrealtime (100 MHz) around the same loop.  Build: hipcc -w -O3 --offload-arch=gfx950 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(0.999f), "v"(0.5f));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *d;
    unsigned long long *st;
    hipMalloc(&d, sizeof(float) * 256 * cus * 8);
    hipMalloc(&st, sizeof(unsigned long long) * 2 * 4 * cus * 8);
    const int iters = 20000;
    static const int order[] = {8, 4, 2, 1, 1, 2, 4, 8};
    for (int oi = 0; oi < 8; ++oi) {
        const int wps = order[oi], blocks = cus * wps;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, st, 200, 1.0f);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, st, iters, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * 4 * blocks);
        hipMemcpy(h.data(), st, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::vector<double> ghz, cpi;
        for (int w = 0; w < 4 * blocks; ++w) {
            ghz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0));           // cycles per ns
            cpi.push_back((double)h[2 * w] / ((double)iters * 16));
        }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cpi.begin(), cpi.end());
        printf("waves/SIMD %d: shader clock %.2f GHz (median over waves), %.2f shader cycles per instruction per WAVE, %.2f per SIMD\n", wps,
               ghz[ghz.size() / 2], cpi[cpi.size() / 2], cpi[cpi.size() / 2] / wps);
    }
    return 0;
}
