// Issue cost of single instruction kinds on gfx950 at a given number of waves per SIMD: shader cycles (s_memtime) per
// wave-instruction per SIMD, 16 independent chains per wave.  Plain VOP2 fmac, the same with a DPP row broadcast on
// src0 (blend_range's operand form), VOP2 with a literal (v_fmaak), VOP3 fma, ds_read_b32 / b128 with a uniform
// address per 16-lane row.  Build: hipcc -w -O3 --offload-arch=gfx950 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters, float seed)
{
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    float b = seed * 0.999f, c = 0.5f;
    const unsigned addr = ((threadIdx.x >> 4) & 15) * 48 + (KIND == 4 ? (threadIdx.x & 15) * 4 : 0);
    float4 q[4];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3d635766" : "+v"(a[i]) : "v"(b));
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if constexpr (KIND == 4) {   // 16 ds_read_b32, one dword per lane of a row's record
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"(addr), "i"(i * 768));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 5) {   // 16 ds_read_b128, the row's lanes on one address (broadcast)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[j]) : "v"(addr), "i"((4 * i + j) * 768));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) a[4 * i + j] = q[j].x;
            }
        } else if constexpr (KIND == 6) {   // quad_perm DPP
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if constexpr (KIND == 7) {   // s_nop
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("s_nop 0");
        } else if constexpr (KIND == 8) {   // SALU
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
        }
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

template <int KIND>
void run(const char *name, int cus, float *d, unsigned long long *st)
{
    const int iters = 5000;
    static const int order[] = {8, 5, 4, 2, 1};
    for (int oi = 0; oi < 5; ++oi) {
        const int wps = order[oi], blocks = cus * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, st, 200, 1.0f);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, st, iters, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * 4 * blocks);
        hipMemcpy(h.data(), st, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::vector<double> ghz, cpi;
        for (int w = 0; w < 4 * blocks; ++w) {
            ghz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0));
            cpi.push_back((double)h[2 * w] / ((double)iters * 16));
        }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cpi.begin(), cpi.end());
        printf("%-28s waves/SIMD %d: clock %.2f GHz, %6.2f cycles per instruction per WAVE, %5.2f per SIMD\n", name, wps, ghz[ghz.size() / 2],
               cpi[cpi.size() / 2], cpi[cpi.size() / 2] / wps);
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
    run<0>("v_fmac_f32 (VOP2)", cus, d, st);
    run<1>("v_fmac_f32_dpp row_newbcast", cus, d, st);
    run<6>("v_fmac_f32_dpp quad_perm", cus, d, st);
    run<2>("v_fmaak_f32 (literal)", cus, d, st);
    run<3>("v_fma_f32 (VOP3)", cus, d, st);
    run<4>("ds_read_b32 row record", cus, d, st);
    run<5>("ds_read_b128 row broadcast", cus, d, st);
    run<7>("s_nop 0", cus, d, st);
    run<8>("s_add_u32", cus, d, st);
    return 0;
}
