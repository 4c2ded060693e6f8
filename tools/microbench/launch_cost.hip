// Host cost of a kernel launch: hipLaunchKernelGGL against hipModuleLaunchKernel on a handle resolved once
// (hipGetFuncBySymbol), with a small and a 2 KB argument block.   hipcc -O2 --offload-arch=gfx950 launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Small { int a[8]; };
struct Big { int a[512]; };
__global__ void k_small(Small s, int *out) { if (s.a[0] == 12345) out[0] = 1; }
__global__ void k_big(Big s, int *out) { if (s.a[0] == 12345) out[0] = 1; }
template <typename F> static double per_call_us(F f, int n)
{
    f();
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main()
{
    int *out;
    (void)hipMalloc(&out, 4);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    Small s{}; Big b{};
    const int n = 20000;
    printf("hipLaunchKernelGGL, 32-byte args:  %.2f us per launch\n", per_call_us([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, s, out); }, n));
    printf("hipLaunchKernelGGL, 2 KB args:     %.2f us per launch\n", per_call_us([&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, b, out); }, n));
    hipFunction_t fs = nullptr, fb = nullptr;
    hipError_t e1 = hipGetFuncBySymbol(&fs, reinterpret_cast<const void *>(k_small)), e2 = hipGetFuncBySymbol(&fb, reinterpret_cast<const void *>(k_big));
    if (e1 != hipSuccess || e2 != hipSuccess) { printf("hipGetFuncBySymbol failed (%d, %d)\n", (int)e1, (int)e2); return 0; }
    void *as[] = {&s, &out}, *ab[] = {&b, &out};
    printf("hipModuleLaunchKernel, 32-byte:    %.2f us per launch\n", per_call_us([&] { (void)hipModuleLaunchKernel(fs, 1, 1, 1, 64, 1, 1, 0, st, as, nullptr); }, n));
    printf("hipModuleLaunchKernel, 2 KB:       %.2f us per launch\n", per_call_us([&] { (void)hipModuleLaunchKernel(fb, 1, 1, 1, 64, 1, 1, 0, st, ab, nullptr); }, n));
    // the packed-buffer form (HIP_LAUNCH_PARAM_BUFFER_POINTER): no per-argument marshalling
    struct { Big b; int *out; } pk{b, out};
    size_t sz = sizeof(pk);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    printf("hipModuleLaunchKernel, 2 KB packed: %.2f us per launch\n", per_call_us([&] { (void)hipModuleLaunchKernel(fb, 1, 1, 1, 64, 1, 1, 0, st, nullptr, extra); }, n));
    return 0;
}
