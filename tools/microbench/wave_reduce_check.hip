// The DPP wave reductions / scan of csrc/sas_device.h against a host loop, on random and on adversarial lanes.
//   hipcc -w -O2 --offload-arch=gfx950 -I sim_a_splat_amd/csrc -I include -o /tmp/wave_reduce_check tools/microbench/wave_reduce_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "sas_device.h"
__global__ void k(const int *in, int *out)
{
    const int v = in[blockIdx.x * 64 + threadIdx.x];
    int *o = out + 6 * (blockIdx.x * 64 + threadIdx.x);
    o[0] = wave_min_i32(v); o[1] = wave_max_i32(v); o[2] = wave_sum_i32(v);
    o[3] = (int)wave_min_u32((unsigned)v); o[4] = (int)wave_max_u32((unsigned)v); o[5] = (int)wave_inclusive_sum_u32((unsigned)v);
}
int main()
{
    const int W = 4096;
    int *h = (int *)malloc(sizeof(int) * 64 * W), *r = (int *)malloc(sizeof(int) * 6 * 64 * W);
    srand(7);
    for (int w = 0; w < W; ++w)
        for (int l = 0; l < 64; ++l) {
            int v = (rand() << 16) ^ rand();
            if (w % 5 == 1) v = (l == w % 64) ? -2147483647 - 1 : 2147483647;      // one extreme lane
            if (w % 5 == 2) v = l;                                                // ramps
            if (w % 5 == 3) v = (l == (w / 5) % 64) ? 1 : 0;                       // one hot
            h[64 * w + l] = v;
        }
    int *d, *o;
    hipMalloc(&d, sizeof(int) * 64 * W); hipMalloc(&o, sizeof(int) * 6 * 64 * W);
    hipMemcpy(d, h, sizeof(int) * 64 * W, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(int) * 6 * 64 * W, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int w = 0; w < W; ++w) {
        int mn = 2147483647, mx = -2147483647 - 1; unsigned umn = ~0u, umx = 0u, sum = 0u, run = 0u;
        for (int l = 0; l < 64; ++l) { int v = h[64 * w + l]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; umn = (unsigned)v < umn ? (unsigned)v : umn; umx = (unsigned)v > umx ? (unsigned)v : umx; sum += (unsigned)v; }
        for (int l = 0; l < 64; ++l) {
            run += (unsigned)h[64 * w + l];
            const int *q = r + 6 * (64 * w + l);
            bad += q[0] != mn || q[1] != mx || (unsigned)q[2] != sum || (unsigned)q[3] != umn || (unsigned)q[4] != umx || (unsigned)q[5] != run;
        }
    }
    printf("%d waves x 64 lanes x 6 results: %ld wrong\n", W, bad);
    return bad != 0;
}
