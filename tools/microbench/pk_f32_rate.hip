// Issue rate of packed-f32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against their
// scalar forms on gfx950, in a VALU-only kernel (no MFMA nearby).  Build:
//   hipcc -w -O3 --offload-arch=gfx950 -o pk_f32_rate pk_f32_rate.hip
// Prints cycles per wave-instruction per SIMD for 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    // 8 independent chains: 8 packed registers (MODE 1,2,3) or 16 scalar registers (MODE 0)
    if (MODE == 0) {
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(0.999f), "v"(0.5f));
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f2 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f2{seed + (float)(threadIdx.x + i), seed - (float)i};
        const f2 m = {0.999f, 0.998f}, c = {0.5f, 0.25f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 1) a[i] = __builtin_elementwise_fma(a[i], m, c);
                if (MODE == 2) a[i] = a[i] * m;
                if (MODE == 3) a[i] = a[i] + c;
            }
        }
        f2 s = {0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    }
}

template <int MODE>
static void run(const char *name, int insts_per_iter, float *d, int clock_khz, int cus)
{
    const int iters = 20000;
    static const int order[] = {8, 4, 2, 1, 1, 2, 4, 8};   // both directions: a clock ramp would show as asymmetry
    for (int oi = 0; oi < 8; ++oi) {   // waves per SIMD: blocks of 256 threads = 1 wave per SIMD per block
        const int wps = order[oi];
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double cycles = (double)ms * 1e-3 * (double)clock_khz * 1e3;
        const double wave_insts_per_simd = (double)iters * insts_per_iter * wps;
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD\n", name, wps, ms,
               cycles / wave_insts_per_simd);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * p.multiProcessorCount * 8);
    run<0>("v_fma_f32", 16, d, p.clockRate, p.multiProcessorCount);
    run<1>("v_pk_fma_f32", 8, d, p.clockRate, p.multiProcessorCount);
    run<2>("v_pk_mul_f32", 8, d, p.clockRate, p.multiProcessorCount);
    run<3>("v_pk_add_f32", 8, d, p.clockRate, p.multiProcessorCount);
    hipFree(d);
    return 0;
}
