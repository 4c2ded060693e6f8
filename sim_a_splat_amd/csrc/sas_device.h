// sas_device.h -- device-side helpers shared by sas_kernels.hip and sas_tile.hip.
// Everything here follows the arithmetic contract of DESIGN.md: IEEE binary32 operations, fused
// only where fma_() is written (translation units are built with -ffp-contract=off).
#pragma once
#include "sas_internal.h"

#pragma clang fp contract(off)

#define DEV __device__ __forceinline__

constexpr float kNear = 0.01f, kFar = 1e10f, kEps2d = 0.3f;
constexpr float kAlphaThr = 1.0f / 255.0f, kMaxAlpha = 0.999f, kTStop = 1e-4f;

DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// set bits of m below this lane's position (v_mbcnt_lo/hi: two instructions; hipcc does not form them from
// __popcll(m & lanes_below))
DEV unsigned mbcnt64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// ---- bounds-checked build (-DSAS_DEBUG_BOUNDS; GPU AddressSanitizer is not available on this pool) ----------
// SAS_IN(i, n, code) is `true` in the product build.  In the debug build it tests 0 <= i < n; a violation
// is counted, its code and values are kept, and the guarded access is SKIPPED (no fault, the run goes on):
// `if (SAS_IN(pos, cap, 12)) keys[pos] = key;`.  Read back through sas_debug_bounds() (sas_api.cpp).
#ifdef SAS_DEBUG_BOUNDS
static __device__ unsigned long long g_sas_bounds[4];   // [0] violations [1] first code [2] first index [3] first limit (per translation unit)
DEV bool sas_in_bounds(long long i, long long n, int code)
{
    if (i >= 0 && i < n) return true;
    if (atomicAdd(&g_sas_bounds[0], 1ull) == 0ull) {
        g_sas_bounds[1] = (unsigned long long)code;
        g_sas_bounds[2] = (unsigned long long)i;
        g_sas_bounds[3] = (unsigned long long)n;
    }
    return false;
}
#define SAS_IN(i, n, code) sas_in_bounds((long long)(i), (long long)(n), (code))
#define SAS_BOUNDS_ACCESSOR(name)                                                                              \
    extern "C" int name(unsigned long long *out, int reset)                                                    \
    {                                                                                                          \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sas_bounds), sizeof(g_sas_bounds)) != hipSuccess) return -1; \
        unsigned long long z[4] = {0, 0, 0, 0};                                                                \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_sas_bounds), z, sizeof(z)) != hipSuccess) return -1;       \
        return 0;                                                                                              \
    }
#else
#define SAS_IN(i, n, code) true
#define SAS_BOUNDS_ACCESSOR(name)
#endif

// 16-byte non-temporal (streaming) load
DEV float4 nt_load(const float4 *p)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// ---- wave64 reductions and scan on the DPP network --------------------------------------------------------------------------------
// hipcc lowers __shfl_xor / __shfl_up to ds_bpermute_b32: a trip through the LDS crossbar per step, six DEPENDENT trips per reduction
// (~0.3 us on the chain of a workgroup that does nothing else meanwhile).  The same steps as DPP operand modifiers cost a few
// issue slots each: butterflies inside the quads and rows (quad_perm, row_half_mirror, row_mirror: every lane of a row then holds
// its row's value), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's value, read by
// v_readlane.  Every lane of the wave must be active (the call sites are wave-uniform).  Result: uniform.
template <int CTRL, int ROW_MASK>
DEV int dpp_take(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
template <typename Op>
DEV int wave_reduce_i32(int v, int identity, Op op)
{
    v = op(v, dpp_take<0xB1, 0xf>(identity, v));    // quad_perm [1,0,3,2]
    v = op(v, dpp_take<0x4E, 0xf>(identity, v));    // quad_perm [2,3,0,1]
    v = op(v, dpp_take<0x141, 0xf>(identity, v));   // row_half_mirror
    v = op(v, dpp_take<0x140, 0xf>(identity, v));   // row_mirror
    v = op(v, dpp_take<0x142, 0xa>(identity, v));   // row_bcast:15 -> rows 1, 3
    v = op(v, dpp_take<0x143, 0xc>(identity, v));   // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
DEV int wave_min_i32(int v) { return wave_reduce_i32(v, 0x7fffffff, [](int a, int b) { return a < b ? a : b; }); }
DEV int wave_max_i32(int v) { return wave_reduce_i32(v, (int)0x80000000, [](int a, int b) { return a > b ? a : b; }); }
DEV int wave_sum_i32(int v) { return wave_reduce_i32(v, 0, [](int a, int b) { return a + b; }); }
DEV unsigned wave_min_u32(unsigned v)
{
    return (unsigned)wave_reduce_i32((int)v, -1, [](int a, int b) { return (int)((unsigned)a < (unsigned)b ? (unsigned)a : (unsigned)b); });
}
DEV unsigned wave_max_u32(unsigned v)
{
    return (unsigned)wave_reduce_i32((int)v, 0, [](int a, int b) { return (int)((unsigned)a > (unsigned)b ? (unsigned)a : (unsigned)b); });
}
// inclusive prefix sum over the wave's 64 lanes (row_shr 1, 2, 3 on the input, then 4 and 8 with bank masks, then the two row
// broadcasts: the scan of the GCN3 cross-lane note)
DEV unsigned wave_inclusive_sum_u32(unsigned x)
{
    const int v0 = (int)x;
    int v = v0 + __builtin_amdgcn_update_dpp(0, v0, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v0, 0x112, 0xf, 0xf, false);              // row_shr:2 (of the input)
    v += __builtin_amdgcn_update_dpp(0, v0, 0x113, 0xf, 0xf, false);              // row_shr:3 (of the input)
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false);               // row_shr:4, banks 1-3
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false);               // row_shr:8, banks 2-3
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);               // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);               // row_bcast:31 -> rows 2, 3
    return (unsigned)v;
}

// lane `src` (uniform) of a register: v_readlane_b32 (hipcc's __shfl(v, src) is a ds_bpermute_b32 round trip even when src is uniform)
DEV int lane_get(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
DEV unsigned lane_get(unsigned v, int src) { return (unsigned)__builtin_amdgcn_readlane((int)v, src); }
DEV float lane_get(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }

// ---- contract logarithm (mirrors sas_oracle_logf; the contract exponential lives with its only
//      user, the compositing loop: c_expf_neg in sas_tile.hip) ------------------------------------
DEV float c_logf(float x)
{
    unsigned u = __float_as_uint(x);
    int e = (int)(u >> 23) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = 0.1111111111f;
    p = fma_(p, s2, 0.1428571429f);
    p = fma_(p, s2, 0.2f);
    p = fma_(p, s2, 0.3333333333f);
    p = fma_(p, s2, 1.0f);
    float lnm = (2.0f * s) * p;
    return fma_((float)e, 0.6931471805599453f, lnm);
}

DEV float affine3(float r0, float r1, float r2, float t, float v0, float v1, float v2)
{
    return fma_(r0, v0, fma_(r1, v1, fma_(r2, v2, t)));
}
DEV float dot3(float a0, float a1, float a2, float b0, float b1, float b2)
{
    return fma_(a2, b2, fma_(a1, b1, a0 * b0));
}

// out = R s R^T, s = xx xy xz yy yz zz
DEV void rot_sym3(const float *R, const float *s, float *out)
{
    const float S[3][3] = {{s[0], s[1], s[2]}, {s[1], s[3], s[4]}, {s[2], s[4], s[5]}};
    float T[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            T[i][j] = dot3(R[3 * i + 0], R[3 * i + 1], R[3 * i + 2], S[0][j], S[1][j], S[2][j]);
    out[0] = dot3(T[0][0], T[0][1], T[0][2], R[0], R[1], R[2]);
    out[1] = dot3(T[0][0], T[0][1], T[0][2], R[3], R[4], R[5]);
    out[2] = dot3(T[0][0], T[0][1], T[0][2], R[6], R[7], R[8]);
    out[3] = dot3(T[1][0], T[1][1], T[1][2], R[3], R[4], R[5]);
    out[4] = dot3(T[1][0], T[1][1], T[1][2], R[6], R[7], R[8]);
    out[5] = dot3(T[2][0], T[2][1], T[2][2], R[6], R[7], R[8]);
}

