// sas_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the render-image hot path.
//
// Stage map (the algorithm is gsplat 1.5.2's, the decomposition is not):
//   k_project   T1+T2  one lane per Gaussian: 16-byte plane loads, EWA projection, SH colour,
//                      48-byte projected record, tile rectangle, per-tile counts
//   k_scan      T5     exclusive scan of the per-tile counts (tile_offset, scatter cursors)
//   k_scatter   T3     (depth bits | index) keys into per-tile segments
//   k_sort      T4     per-tile stable LSD radix sort on depth bits in LDS, wave-ballot ranks
//                      (replaces the global 64-bit radix sort)
//   k_blend     T6+T0  one workgroup per 16x16 tile (longest list first), one wave per 8x8
//                      quadrant, LDS-staged splat queue compacted per wave by ballot from the
//                      quadrant hit masks, front-to-back compositing,
//                      background / clamp / uint8 / expected-depth epilogue
//
// ARITHMETIC CONTRACT: every value that reaches an output is produced by the same
// sequence of IEEE binary32 operations as oracle/sas_oracle.c -- explicit __builtin_fmaf where
// the contract fuses, no other contraction (-ffp-contract=off), correctly rounded / and sqrt
// (hipcc default), polynomial exp/log.  No MFMA: nothing here is a dense contraction.
#include "sas_internal.h"

#pragma clang fp contract(off)

#define DEV __device__ __forceinline__

namespace {

constexpr float kNear = 0.01f, kFar = 1e10f, kEps2d = 0.3f;
constexpr float kAlphaThr = 1.0f / 255.0f, kMaxAlpha = 0.999f, kTStop = 1e-4f;

DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- contract transcendental functions (mirror sas_oracle_expf / sas_oracle_logf) -------------
DEV float c_expf(float x)
{
    float t = x * 1.4426950408889634f;
    t = fmaxf(t, -125.0f);
    t = fminf(t, 126.0f);
    float n = __builtin_rintf(t);
    float f = t - n;
    float p = 0.0013400432653725147f;
    p = fma_(p, f, 0.009676037356257439f);
    p = fma_(p, f, 0.05550327152013779f);
    p = fma_(p, f, 0.2402210682630539f);
    p = fma_(p, f, 0.6931471824645996f);
    p = fma_(p, f, 1.0000001192092896f);
    return __builtin_ldexpf(p, (int)n);
}

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

// ---- upload: AoS inputs -> 16-byte planes ------------------------------------------------------
__global__ __launch_bounds__(256) void k_relayout(int64_t n, int64_t n_pad, const int *perm, const float *means,
                                                  const float *quats, const float *scales, const float *cov6,
                                                  const float *opac, const float *colors, int coeff_floats, int planes,
                                                  const uint8_t *gid, float4 *g0, float4 *g1, float4 *g2, float4 *col)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;   // slot
    if (j >= n) return;
    const int64_t i = perm[j];                                    // caller's index
    float gbits = __uint_as_float(gid ? (unsigned)gid[i] : 0u);
    g0[j] = make_float4(means[3 * i], means[3 * i + 1], means[3 * i + 2], opac[i]);
    if (quats) {
        g1[j] = make_float4(quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3]);
        g2[j] = make_float4(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2], gbits);
    } else {
        const float *c = cov6 + 6 * i;
        g1[j] = make_float4(c[0], c[1], c[2], c[3]);
        g2[j] = make_float4(c[4], c[5], 0.0f, gbits);
    }
    const float *src = colors + (int64_t)coeff_floats * i;
    for (int p = 0; p < planes; ++p) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (4 * p + k < coeff_floats) ? src[4 * p + k] : 0.0f;
        col[(int64_t)p * n_pad + j] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- T2: SH colour from register-resident coefficients -----------------------------------------
template <int DEG>
DEV void sh_to_color(const float *sh, float dx, float dy, float dz, float *rgb)
{
    float inorm = 1.0f / sqrtf(fma_(dz, dz, fma_(dy, dy, dx * dx)));
    float x = dx * inorm, y = dy * inorm, z = dz * inorm;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z;
    float fC1 = fma_(x, x, -(y * y));
    float fS1 = 2.0f * x * y;
    float pSH6 = fma_(0.9461746957575601f, z2, -0.3153915652525201f);
    float pSH7 = fTmp0B * x;
    float pSH5 = fTmp0B * y;
    float pSH8 = 0.5462742152960395f * fC1;
    float pSH4 = 0.5462742152960395f * fS1;
    float fTmp0C = fma_(-2.285228997322329f, z2, 0.4570457994644658f);
    float fTmp1B = 1.445305721320277f * z;
    float fC2 = fma_(x, fC1, -(y * fS1));
    float fS2 = fma_(x, fS1, y * fC1);
    float pSH12 = z * fma_(1.865881662950577f, z2, -1.119528997770346f);
    float pSH13 = fTmp0C * x;
    float pSH11 = fTmp0C * y;
    float pSH14 = fTmp1B * fC1;
    float pSH10 = fTmp1B * fS1;
    float pSH15 = -0.5900435899266435f * fC2;
    float pSH9 = -0.5900435899266435f * fS2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = 0.2820947917738781f * sh[0 * 3 + c];
        if constexpr (DEG >= 1) {
            float t = fma_(-x, sh[3 * 3 + c], fma_(z, sh[2 * 3 + c], (-y) * sh[1 * 3 + c]));
            r = fma_(0.48860251190292f, t, r);
        }
        if constexpr (DEG >= 2) {
            r = fma_(pSH4, sh[4 * 3 + c], r);
            r = fma_(pSH5, sh[5 * 3 + c], r);
            r = fma_(pSH6, sh[6 * 3 + c], r);
            r = fma_(pSH7, sh[7 * 3 + c], r);
            r = fma_(pSH8, sh[8 * 3 + c], r);
        }
        if constexpr (DEG >= 3) {
            r = fma_(pSH9, sh[9 * 3 + c], r);
            r = fma_(pSH10, sh[10 * 3 + c], r);
            r = fma_(pSH11, sh[11 * 3 + c], r);
            r = fma_(pSH12, sh[12 * 3 + c], r);
            r = fma_(pSH13, sh[13 * 3 + c], r);
            r = fma_(pSH14, sh[14 * 3 + c], r);
            r = fma_(pSH15, sh[15 * 3 + c], r);
        }
        rgb[c] = fmaxf(r + 0.5f, 0.0f);
    }
}

// Visit every tile of a rectangle.  Small rectangles are walked by their own lane; a lane with a
// large rectangle hands it to the whole wave (64 lanes stride over its tiles) so that one
// screen-filling Gaussian does not serialise a wave.  v0/v1 are the owning lane's payload; they
// are broadcast while the wave is still convergent (a cross-lane read of an inactive lane
// returns 0), then handed to emit(tile, v0, v1).  Must be reached by all 64 lanes.
constexpr int kSmallRect = 8;

template <typename F>
DEV void for_each_tile(bool active, int x0, int x1, int y0, int y1, int tw, unsigned v0, unsigned v1, F emit)
{
    const int lane = threadIdx.x & 63;
    const int w = x1 - x0;
    const int area = active ? w * (y1 - y0) : 0;
    if (area > 0 && area <= kSmallRect) {
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) emit(ty * tw + tx, v0, v1);
    }
    unsigned long long big = __ballot(area > kSmallRect);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bx0 = __shfl(x0, src), by0 = __shfl(y0, src);
        const int bw = __shfl(w, src), ba = __shfl(area, src);
        const unsigned b0 = __shfl(v0, src), b1 = __shfl(v1, src);
        for (int i = lane; i < ba; i += 64) emit((by0 + i / bw) * tw + bx0 + i % bw, b0, b1);
    }
}

// Workgroup screen window.  The scene is stored along a Hilbert curve, so the 256 Gaussians of a
// workgroup land in a compact block of tiles: their per-tile counts are accumulated in an LDS
// histogram over that window and leave the CU as ONE global atomic per touched tile instead of
// one per intersection (device-scope atomics execute at the memory side: ~10 G/s scattered).
// Gaussians with a large rectangle stay out of the window and use the wave-cooperative walk.
constexpr int kHistBins = 2048;  // 8 KiB of LDS
constexpr int kWinRect = 64;     // largest rectangle (tiles) that takes part in the window

struct Window {
    int X0, Y0, ww, area;   // origin, width, bin count (0: no participant)
    bool fits;
};

DEV Window wg_window(bool part, int x0, int x1, int y0, int y1, int *s_win)
{
    int mnx = part ? x0 : 0x7fffffff, mny = part ? y0 : 0x7fffffff;
    int mxx = part ? x1 : 0, mxy = part ? y1 : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, d));
        mny = min(mny, __shfl_xor(mny, d));
        mxx = max(mxx, __shfl_xor(mxx, d));
        mxy = max(mxy, __shfl_xor(mxy, d));
    }
    if (threadIdx.x == 0) { s_win[0] = 0x7fffffff; s_win[1] = 0x7fffffff; s_win[2] = 0; s_win[3] = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_win[0], mnx); atomicMin(&s_win[1], mny);
        atomicMax(&s_win[2], mxx); atomicMax(&s_win[3], mxy);
    }
    __syncthreads();
    Window w;
    w.X0 = s_win[0]; w.Y0 = s_win[1];
    const int X1 = s_win[2], Y1 = s_win[3];
    w.ww = X1 - w.X0;
    w.area = (X1 > w.X0 && Y1 > w.Y0) ? w.ww * (Y1 - w.Y0) : 0;
    w.fits = w.area > 0 && w.area <= kHistBins;
    return w;
}

// ---- k_project: T1 + T2 + tile counts ----------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(256) void k_project(SasScene s, SasCam c, SasFrame f)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in_range = i < s.n;
    bool vis = false;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    if (in_range) {
        const float4 a0 = s.g0[i];
        const float4 a1 = s.g1[i];
        const float4 a2 = s.g2[i];
        float m[3] = {a0.x, a0.y, a0.z};
        const float op = a0.w;
        const float *G = nullptr;
        if (s.group_Rt) {
            G = s.group_Rt + 12 * (__float_as_uint(a2.w) & 255u);
            float mg0 = affine3(G[0], G[1], G[2], G[3], m[0], m[1], m[2]);
            float mg1 = affine3(G[4], G[5], G[6], G[7], m[0], m[1], m[2]);
            float mg2 = affine3(G[8], G[9], G[10], G[11], m[0], m[1], m[2]);
            m[0] = mg0; m[1] = mg1; m[2] = mg2;
        }
        const float x = affine3(c.R[0], c.R[1], c.R[2], c.t[0], m[0], m[1], m[2]);
        const float y = affine3(c.R[3], c.R[4], c.R[5], c.t[1], m[0], m[1], m[2]);
        const float z = affine3(c.R[6], c.R[7], c.R[8], c.t[2], m[0], m[1], m[2]);
        bool ok = !(z < kNear || z > kFar);
        ok = ok && !(op < kAlphaThr);   // opacity cull moved up: it has no side effect before the det test
        if (ok) {
            float cov[6];
            if (!s.cov_mode) {
                float qw = a1.x, qx = a1.y, qy = a1.z, qz = a1.w;
                float n2 = fma_(qz, qz, fma_(qy, qy, fma_(qx, qx, qw * qw)));
                float inv = 1.0f / sqrtf(n2);
                qw *= inv; qx *= inv; qy *= inv; qz *= inv;
                float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
                float xy = qx * qy, xz = qx * qz, yz = qy * qz;
                float wx = qw * qx, wy = qw * qy, wz = qw * qz;
                float R[9];
                R[0] = fma_(-2.0f, y2 + z2, 1.0f); R[1] = 2.0f * (xy - wz);           R[2] = 2.0f * (xz + wy);
                R[3] = 2.0f * (xy + wz);           R[4] = fma_(-2.0f, x2 + z2, 1.0f); R[5] = 2.0f * (yz - wx);
                R[6] = 2.0f * (xz - wy);           R[7] = 2.0f * (yz + wx);           R[8] = fma_(-2.0f, x2 + y2, 1.0f);
                if (G) {
                    float R2[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            R2[3 * r + k] = dot3(G[4 * r + 0], G[4 * r + 1], G[4 * r + 2], R[0 + k], R[3 + k], R[6 + k]);
#pragma unroll
                    for (int k = 0; k < 9; ++k) R[k] = R2[k];
                }
                const float sc[3] = {a2.x, a2.y, a2.z};
                float M[9];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) M[3 * r + k] = R[3 * r + k] * sc[k];
                cov[0] = dot3(M[0], M[1], M[2], M[0], M[1], M[2]);
                cov[1] = dot3(M[0], M[1], M[2], M[3], M[4], M[5]);
                cov[2] = dot3(M[0], M[1], M[2], M[6], M[7], M[8]);
                cov[3] = dot3(M[3], M[4], M[5], M[3], M[4], M[5]);
                cov[4] = dot3(M[3], M[4], M[5], M[6], M[7], M[8]);
                cov[5] = dot3(M[6], M[7], M[8], M[6], M[7], M[8]);
            } else {
                cov[0] = a1.x; cov[1] = a1.y; cov[2] = a1.z; cov[3] = a1.w; cov[4] = a2.x; cov[5] = a2.y;
                if (G) {
                    const float Rg[9] = {G[0], G[1], G[2], G[4], G[5], G[6], G[8], G[9], G[10]};
                    float c2[6];
                    rot_sym3(Rg, cov, c2);
#pragma unroll
                    for (int k = 0; k < 6; ++k) cov[k] = c2[k];
                }
            }
            float cc3[6];
            rot_sym3(c.R, cov, cc3);

            const float rz = 1.0f / z;
            const float rz2 = rz * rz;
            const float tx = z * fminf(c.lim_x_pos, fmaxf(-c.lim_x_neg, x * rz));
            const float ty = z * fminf(c.lim_y_pos, fmaxf(-c.lim_y_neg, y * rz));
            const float ja = c.fx * rz, jb = -((c.fx * tx) * rz2);
            const float jc = c.fy * rz, jd = -((c.fy * ty) * rz2);
            const float t00 = fma_(jb, cc3[2], ja * cc3[0]);
            const float t01 = fma_(jb, cc3[4], ja * cc3[1]);
            const float t02 = fma_(jb, cc3[5], ja * cc3[2]);
            const float t11 = fma_(jd, cc3[4], jc * cc3[3]);
            const float t12 = fma_(jd, cc3[5], jc * cc3[4]);
            float c00 = fma_(t02, jb, t00 * ja);
            const float c01 = fma_(t02, jd, t01 * jc);
            float c11 = fma_(t12, jd, t11 * jc);
            const float mx = fma_(c.fx, x * rz, c.cx);
            const float my = fma_(c.fy, y * rz, c.cy);
            c00 += kEps2d;
            c11 += kEps2d;
            const float det = fma_(c00, c11, -(c01 * c01));
            if (det > 0.0f) {
                const float inv_det = 1.0f / det;
                const float ca = c11 * inv_det, cb = -c01 * inv_det, ccn = c00 * inv_det;
                const float lnq = c_logf(op / kAlphaThr);
                const float extent = fminf(3.33f, sqrtf(2.0f * lnq));
                const float b = 0.5f * (c00 + c11);
                const float tmp = sqrtf(fmaxf(0.01f, fma_(b, b, -det)));
                const float v1 = b + tmp;
                const float r1 = extent * sqrtf(v1);
                const float rx = ceilf(fminf(extent * sqrtf(c00), r1));
                const float ry = ceilf(fminf(extent * sqrtf(c11), r1));
                bool keep = !(rx <= 0.0f && ry <= 0.0f);
                keep = keep && !(mx + rx <= 0.0f || mx - rx >= c.Wf || my + ry <= 0.0f || my - ry >= c.Hf);
                keep = keep && rx > 0.0f && ry > 0.0f;
                if (keep) {
                    vis = true;
                    // T3 tile rectangle
                    const float ts = (float)SAS_TILE;
                    const float trx = rx / ts, try_ = ry / ts;
                    const float ttx = mx / ts, tty = my / ts;
                    const float twf = (float)c.tw, thf = (float)c.th;
                    x0 = (int)fminf(fmaxf(floorf(ttx - trx), 0.0f), twf);
                    x1 = (int)fminf(fmaxf(ceilf(ttx + trx), 0.0f), twf);
                    y0 = (int)fminf(fmaxf(floorf(tty - try_), 0.0f), thf);
                    y1 = (int)fminf(fmaxf(ceilf(tty + try_), 0.0f), thf);
                    // colour
                    float rgb[3];
                    if constexpr (DEG >= 0) {
                        constexpr int KF = 3 * (DEG + 1) * (DEG + 1);
                        constexpr int PL = (KF + 3) / 4;
                        float sh[PL * 4];
#pragma unroll
                        for (int p = 0; p < PL; ++p) {
                            const float4 v = s.col[(int64_t)p * s.n_pad + i];
                            sh[4 * p] = v.x; sh[4 * p + 1] = v.y; sh[4 * p + 2] = v.z; sh[4 * p + 3] = v.w;
                        }
                        sh_to_color<DEG>(sh, m[0] - c.campos[0], m[1] - c.campos[1], m[2] - c.campos[2], rgb);
                    } else {
                        const float4 v = s.col[i];
                        rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z;
                    }
                    // conservative skip threshold for the blend stage: alpha >= 1/255 implies
                    // sigma <= ln(255 op) + rounding; 1e-3 is > 100x the worst rounding.
                    const float thr = lnq + 1e-3f;
                    f.rec[3 * i + 0] = make_float4(mx, my, ca, cb);
                    f.rec[3 * i + 1] = make_float4(ccn, op, thr, z);
                    f.rec[3 * i + 2] = make_float4(rgb[0], rgb[1], rgb[2], 0.0f);
                    f.info[i] = make_uint4((unsigned)x0 | ((unsigned)x1 << 16), (unsigned)y0 | ((unsigned)y1 << 16),
                                           __float_as_uint(z), (unsigned)(int)rx | ((unsigned)(int)ry << 16));
                }
            }
        }
        if (!vis) f.info[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // per-tile counts: LDS histogram over the workgroup's window, one global atomic per touched tile
    __shared__ int s_win[4];
    __shared__ int s_hist[kHistBins];
    const int rect_area = vis ? (x1 - x0) * (y1 - y0) : 0;
    const bool in_win = rect_area > 0 && rect_area <= kWinRect;
    const Window w = wg_window(in_win, x0, x1, y0, y1, s_win);
    if (w.fits) {
        for (int b = threadIdx.x; b < w.area; b += 256) s_hist[b] = 0;
        __syncthreads();
        if (in_win)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) atomicAdd(&s_hist[(ty - w.Y0) * w.ww + (tx - w.X0)], 1);
        __syncthreads();
        for (int b = threadIdx.x; b < w.area; b += 256) {
            const int cnt = s_hist[b];
            if (cnt) atomicAdd(&f.tile_count[(w.Y0 + b / w.ww) * c.tw + w.X0 + b % w.ww], cnt);
        }
    } else if (threadIdx.x == 0 && w.area > 0) {
        atomicAdd(&f.stats[5], 1u);
    }
    for_each_tile(vis && !(w.fits && in_win), x0, x1, y0, y1, c.tw, 0u, 0u,
                  [&](int tile, unsigned, unsigned) { atomicAdd(&f.tile_count[tile], 1); });
    // visible count: one plain store per workgroup (a same-address atomic per wave would
    // serialise at ~90 atomics/us); k_scan adds the per-workgroup counts up
    __shared__ int s_nvis;
    if (threadIdx.x == 0) s_nvis = 0;
    __syncthreads();
    const unsigned long long vb = __ballot(vis);
    if ((threadIdx.x & 63) == 0 && vb) atomicAdd(&s_nvis, (int)__popcll(vb));
    __syncthreads();
    if (threadIdx.x == 0) f.wg_vis[blockIdx.x] = s_nvis;
}

// ---- k_scan: exclusive scan over tiles (one workgroup) ------------------------------------------
// Each thread owns 8 consecutive tiles per round and loads them before anything else, so a round
// costs one memory latency (the counts were written by memory-side atomics and miss every cache).
// Also: blend launch order (tiles bucketed by floor(log2(length)), longest first, so long lists
// start early and short ones fill the tail), visible count, max list length, overflow flag.
constexpr int kScanPer = 8;

__global__ __launch_bounds__(1024) void k_scan(SasFrame f, int tiles)
{
    __shared__ int wsum[16];
    __shared__ int s_bucket[33];
    __shared__ int s_bbase[33];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 33) s_bucket[tid] = 0;
    int nvis = 0;
    for (int i = tid; i < f.n_wg; i += 1024) nvis += f.wg_vis[i];
    __syncthreads();
    int carry = 0, maxlen = 0;
    const int rounds = (tiles + 1024 * kScanPer - 1) / (1024 * kScanPer);
    // pass 1: offsets + bucket histogram
    for (int r = 0; r < rounds; ++r) {
        const int base = (r * 1024 + tid) * kScanPer;
        int v[kScanPer];
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) v[k] = (base + k < tiles) ? f.tile_count[base + k] : 0;
        int sum = 0;
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) {
            sum += v[k];
            maxlen = max(maxlen, v[k]);
            if (base + k < tiles) atomicAdd(&s_bucket[v[k] ? 32 - __clz(v[k]) : 0], 1);
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = wsum[k];
            if (k < wv) woff += x;
            total += x;
        }
        int run = carry + woff + incl - sum;
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) {
            if (base + k < tiles) { f.tile_offset[base + k] = run; f.tile_cursor[base + k] = run; }
            run += v[k];
        }
        carry += total;
        __syncthreads();
    }
    // bucket bases, longest lists first
    if (tid == 0) {
        int run = 0;
        for (int bkt = 32; bkt >= 0; --bkt) { s_bbase[bkt] = run; run += s_bucket[bkt]; }
        // bucket b holds lengths [2^(b-1), 2^b): sort classes are bucket ranges, hence contiguous
        f.sort_class[0] = 0;             // large: length >= 4096 (buckets >= 13)
        f.sort_class[1] = s_bbase[12];   // mid:   1024..4095     (buckets 11, 12)
        f.sort_class[2] = s_bbase[10];   // small: < 1024         (buckets <= 10)
        f.sort_class[3] = tiles;
    }
    __syncthreads();
    // pass 2: placement (counts are L2-resident now)
    for (int r = 0; r < rounds; ++r) {
        const int base = (r * 1024 + tid) * kScanPer;
#pragma unroll
        for (int k = 0; k < kScanPer; ++k)
            if (base + k < tiles) {
                const int v = f.tile_count[base + k];
                f.tile_order[atomicAdd(&s_bbase[v ? 32 - __clz(v) : 0], 1)] = base + k;
            }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        maxlen = max(maxlen, __shfl_xor(maxlen, d));
        nvis += __shfl_xor(nvis, d);
    }
    if (lane == 0) {
        atomicMax(&f.stats[4], (unsigned)maxlen);
        if (nvis) atomicAdd(&f.stats[0], (unsigned)nvis);
    }
    if (tid == 0) {
        f.tile_offset[tiles] = carry;
        f.stats[1] = (unsigned)carry;
        if ((long long)carry > f.cap) f.stats[2] = 1u;
    }
}

// ---- k_scatter: T3 emit ---------------------------------------------------------------------------
// Same window as k_project: count in LDS, reserve one contiguous run per touched tile with a
// single returning global atomic, then rank inside the run with LDS atomics.  The key carries the
// CALLER's Gaussian index (perm[slot]) so that depth ties order exactly as in the reference.
__global__ __launch_bounds__(256) void k_scatter(SasScene s, SasCam c, SasFrame f)
{
    __shared__ int s_win[4];
    __shared__ int s_hist[kHistBins];
    __shared__ int s_base[kHistBins];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint4 inf = make_uint4(0u, 0u, 0u, 0u);
    unsigned orig = 0;
    if (i < s.n) { inf = f.info[i]; orig = (unsigned)s.perm[i]; }
    const int x0 = inf.x & 0xffff, x1 = inf.x >> 16, y0 = inf.y & 0xffff, y1 = inf.y >> 16;
    const bool vis = x1 > x0 && y1 > y0;
    const unsigned long long key = ((unsigned long long)inf.z << 32) | (unsigned long long)orig;
    const int rect_area = vis ? (x1 - x0) * (y1 - y0) : 0;
    const bool in_win = rect_area > 0 && rect_area <= kWinRect;
    const Window w = wg_window(in_win, x0, x1, y0, y1, s_win);
    if (w.fits) {
        for (int b = threadIdx.x; b < w.area; b += 256) s_hist[b] = 0;
        __syncthreads();
        if (in_win)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) atomicAdd(&s_hist[(ty - w.Y0) * w.ww + (tx - w.X0)], 1);
        __syncthreads();
        for (int b = threadIdx.x; b < w.area; b += 256) {
            const int cnt = s_hist[b];
            s_base[b] = cnt ? atomicAdd(&f.tile_cursor[(w.Y0 + b / w.ww) * c.tw + w.X0 + b % w.ww], cnt) : 0;
            s_hist[b] = 0;
        }
        __syncthreads();
        if (in_win)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    const int b = (ty - w.Y0) * w.ww + (tx - w.X0);
                    const long long pos = (long long)s_base[b] + atomicAdd(&s_hist[b], 1);
                    if (pos < f.cap) f.keys[pos] = key;
                }
    }
    const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
    for_each_tile(vis && !(w.fits && in_win), x0, x1, y0, y1, c.tw, klo, khi, [&](int tile, unsigned lo, unsigned hi) {
        const int pos = atomicAdd(&f.tile_cursor[tile], 1);
        if ((long long)pos < f.cap) f.keys[pos] = ((unsigned long long)hi << 32) | lo;
    });
}

// ---- k_sort: per-tile ascending sort of 64-bit keys ---------------------------------------------
// Size classes are launched over all tiles; a workgroup whose tile is not in its class exits at
// once.  The LDS classes run a stable LSD radix sort (8-bit digits) over the depth word only,
// skipping bytes that do not vary inside the tile, then order the rare runs of identical depth by
// the caller's Gaussian index: the result equals the reference's stable sort of
// (tile | depth bits) keys emitted in index order.  Stable ranks come from wave ballots, not LDS
// atomics.  Sorted caller indices are translated to storage slots (inv_perm) on the way out.
// key low word = caller index  ->  list entry = storage slot
DEV int entry_of(const int *inv_perm, unsigned lo) { return inv_perm[lo]; }

DEV void sort_global_bitonic(unsigned long long *g, int *out, int n, const int *inv_perm, int tid, int nthreads);

template <int CAP, int THREADS, bool LAST_CLASS>
__global__ __launch_bounds__(THREADS) void k_sort_radix(SasFrame f, const int *inv_perm, int cls)
{
    constexpr int W = THREADS / 64;     // waves
    constexpr int NB = CAP / THREADS;   // 64-key batches per wave
    static_assert(THREADS >= 256 && CAP % THREADS == 0, "radix sort geometry");
    __shared__ unsigned long long buf[CAP];   // one buffer: between barriers the keys live in registers
    __shared__ unsigned cnt[W][256];
    __shared__ unsigned s_dbase[256];
    __shared__ unsigned s_wsum[4];
    __shared__ unsigned s_or, s_and;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int oi = f.sort_class[cls] + (int)blockIdx.x;
    if (oi >= f.sort_class[cls + 1]) return;   // not a tile of this class
    const int t = f.tile_order[oi];
    const long long beg = f.tile_offset[t];
    long long end = f.tile_offset[t + 1];
    if (end > f.cap) end = f.cap;
    const int n = (int)(end - beg);
    if (n <= 0 || (!LAST_CLASS && n > CAP)) return;
    unsigned long long *g = f.keys + beg;
    int *out = f.sorted_ids + beg;
    if (n == 1) {
        if (tid == 0) out[0] = entry_of(inv_perm, (unsigned)g[0]);
        return;
    }
    if (LAST_CLASS && n > CAP) {   // longer than any LDS class: in place on the global segment
        sort_global_bitonic(g, out, n, inv_perm, tid, THREADS);
        return;
    }
    if (tid == 0) { s_or = 0u; s_and = ~0u; }   // s_or: max depth word, s_and: min depth word
    __syncthreads();
    // contiguous chunk per wave, balanced over the waves: nbu batches of 64 keys each
    const int nbu = (((n + W - 1) / W) + 63) >> 6;   // <= NB because n <= CAP
    const int base = wv * (nbu * 64) + lane;          // element index of batch b: base + 64 b
    unsigned long long k[NB];
    unsigned mn = ~0u, mx = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = base + 64 * b;
        const bool in = b < nbu && i < n;
        k[b] = in ? g[i] : ~0ull;
        if (in) { mn = min(mn, (unsigned)(k[b] >> 32)); mx = max(mx, (unsigned)(k[b] >> 32)); }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, d)); mx = max(mx, (unsigned)__shfl_xor((int)mx, d)); }
    if (lane == 0) { atomicMax(&s_or, mx); atomicMin(&s_and, mn); }
    __syncthreads();
    // sort depth - min(depth): same order, and only the bytes below the range's top bit need a pass
    const unsigned dmin = s_and;
    const unsigned vary = s_or - dmin;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b < nbu && base + 64 * b < n) k[b] -= (unsigned long long)dmin << 32;
    unsigned long long *const src = buf, *const dst = buf;
    bool first = true;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int byte = 0; byte < 4; ++byte) {
        if ((vary >> (8 * byte)) == 0u) break;   // uniform
        if (!first) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int i = base + 64 * b;
                k[b] = (b < nbu && i < n) ? src[i] : ~0ull;
            }
        }
        for (int d = lane; d < 256; d += 64) cnt[wv][d] = 0u;
        unsigned rank[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b >= nbu) break;   // uniform
            const int i = base + 64 * b;
            const bool act = i < n;
            const unsigned d = ((unsigned)(k[b] >> 32) >> (8 * byte)) & 255u;
            unsigned long long m = __ballot(act);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool on = (d >> bit) & 1u;
                const unsigned long long bm = __ballot(on);
                m &= on ? bm : ~bm;
            }
            const unsigned below = (unsigned)__popcll(m & lt_mask);
            const unsigned total = (unsigned)__popcll(m);
            const unsigned prev = act ? cnt[wv][d] : 0u;
            if (act && below == 0u) cnt[wv][d] = prev + total;
            rank[b] = prev + below;
        }
        __syncthreads();
        unsigned tot = 0u, incl = 0u;
        if (tid < 256) {
            unsigned run = 0u;
#pragma unroll
            for (int w = 0; w < W; ++w) { const unsigned cc = cnt[w][tid]; cnt[w][tid] = run; run += cc; }
            tot = run;
            incl = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (lane == 63) s_wsum[wv] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            unsigned off = incl - tot;
            for (int w = 0; w < wv; ++w) off += s_wsum[w];
            s_dbase[tid] = off;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = base + 64 * b;
            if (b < nbu && i < n) {
                const unsigned d = ((unsigned)(k[b] >> 32) >> (8 * byte)) & 255u;
                dst[s_dbase[d] + cnt[wv][d] + rank[b]] = k[b];
            }
        }
        __syncthreads();
        first = false;
    }
    if (first) {   // every depth identical: keys are still only in registers
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = base + 64 * b;
            if (b < nbu && i < n) src[i] = k[b];
        }
        __syncthreads();
    }
    // runs of identical depth bits: order by caller index (the run's first element does it;
    // the depth words other threads look at do not change under the permutation)
    for (int i = tid; i < n; i += THREADS) {
        const unsigned hd = (unsigned)(src[i] >> 32);
        const bool lead = (i == 0 || (unsigned)(src[i - 1] >> 32) != hd) && (i + 1 < n) && (unsigned)(src[i + 1] >> 32) == hd;
        if (lead) {
            int j = i + 1;
            while (j < n && (unsigned)(src[j] >> 32) == hd) ++j;
            for (int a = i + 1; a < j; ++a) {
                const unsigned long long v = src[a];
                int q = a - 1;
                while (q >= i && src[q] > v) { src[q + 1] = src[q]; --q; }
                src[q + 1] = v;
            }
        }
    }
    __syncthreads();
    // output: gathers of all of a thread's entries in flight together
    unsigned lo[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = b * THREADS + tid;
        lo[b] = (i < n) ? (unsigned)entry_of(inv_perm, (unsigned)src[i]) : 0u;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = b * THREADS + tid;
        if (i < n) out[i] = (int)lo[b];
    }
}

// Lists shorter than 1024: one wave per tile, no workgroup barrier anywhere (a wave executes its
// LDS operations in order).  The sort runs on (depth word, position in the unsorted segment):
// 6 bytes of LDS per entry; caller indices are fetched from the L2-hot segment only for depth ties
// and for the final translation to storage slots.
__global__ __launch_bounds__(64) void k_sort_wave(SasFrame f, const int *inv_perm, int cls)
{
    constexpr int CAP = 1024, NB = 16;
    __shared__ unsigned sd[CAP];
    __shared__ unsigned short si[CAP];
    __shared__ __attribute__((aligned(16))) unsigned cnt[256];
    const int lane = threadIdx.x;
    const int oi = f.sort_class[cls] + (int)blockIdx.x;
    if (oi >= f.sort_class[cls + 1]) return;
    const int t = f.tile_order[oi];
    const long long beg = f.tile_offset[t];
    long long end = f.tile_offset[t + 1];
    if (end > f.cap) end = f.cap;
    const int n = (int)(end - beg);
    if (n <= 0 || n > CAP) return;
    const unsigned long long *g = f.keys + beg;
    int *out = f.sorted_ids + beg;
    if (n == 1) {
        if (lane == 0) out[0] = entry_of(inv_perm, (unsigned)g[0]);
        return;
    }
    const int nb = (n + 63) >> 6;
    unsigned kd[NB], ki[NB];
    unsigned mn = ~0u, mx = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = 64 * b + lane;
        const bool in = b < nb && i < n;
        kd[b] = in ? (unsigned)(g[i] >> 32) : ~0u;
        ki[b] = (unsigned)i;
        if (in) { mn = min(mn, kd[b]); mx = max(mx, kd[b]); }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, d)); mx = max(mx, (unsigned)__shfl_xor((int)mx, d)); }
    // sort depth - min(depth): same order, and only the bytes below the range's top bit need a pass
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b < nb && 64 * b + lane < n) kd[b] -= mn;
    const unsigned span = mx - mn;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int byte = 0; byte < 4; ++byte) {
        if ((span >> (8 * byte)) == 0u) break;   // uniform
        reinterpret_cast<uint4 *>(cnt)[lane] = make_uint4(0u, 0u, 0u, 0u);
        unsigned rank[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b >= nb) break;   // uniform
            const bool act = 64 * b + lane < n;
            const unsigned d = (kd[b] >> (8 * byte)) & 255u;
            unsigned long long m = __ballot(act);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool on = (d >> bit) & 1u;
                const unsigned long long bm = __ballot(on);
                m &= on ? bm : ~bm;
            }
            const unsigned below = (unsigned)__popcll(m & lt_mask);
            const unsigned total = (unsigned)__popcll(m);
            const unsigned prev = act ? cnt[d] : 0u;
            if (act && below == 0u) cnt[d] = prev + total;
            rank[b] = prev + below;
        }
        // exclusive scan of the 256 digit counters: 4 per lane + wave scan
        const uint4 c4 = reinterpret_cast<uint4 *>(cnt)[lane];
        const unsigned s3 = c4.x + c4.y + c4.z + c4.w;
        unsigned incl = s3;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const unsigned ex = incl - s3;
        reinterpret_cast<uint4 *>(cnt)[lane] = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < nb && 64 * b + lane < n) {
                const unsigned pos = cnt[(kd[b] >> (8 * byte)) & 255u] + rank[b];
                sd[pos] = kd[b];
                si[pos] = (unsigned short)ki[b];
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = 64 * b + lane;
            if (b < nb && i < n) { kd[b] = sd[i]; ki[b] = si[i]; }
        }
    }
    // current order back to LDS (also covers "no byte varied")
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = 64 * b + lane;
        if (b < nb && i < n) { sd[i] = kd[b]; si[i] = (unsigned short)ki[b]; }
    }
    // runs of identical depth bits: order by caller index
    for (int i = lane; i < n; i += 64) {
        const unsigned hd = sd[i];
        const bool lead = (i == 0 || sd[i - 1] != hd) && (i + 1 < n) && sd[i + 1] == hd;
        if (lead) {
            int j = i + 1;
            while (j < n && sd[j] == hd) ++j;
            for (int a = i + 1; a < j; ++a) {
                const unsigned short va = si[a];
                const unsigned ka = (unsigned)g[va];
                int q = a - 1;
                while (q >= i && (unsigned)g[si[q]] > ka) { si[q + 1] = si[q]; --q; }
                si[q + 1] = va;
            }
        }
    }
    // output: all gathers of a phase in flight together (a plain loop would serialise two dependent
    // L2 round trips per 64 entries, because the stores may alias the tables)
    unsigned lo[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = 64 * b + lane;
        lo[b] = (b < nb && i < n) ? (unsigned)g[si[i]] : 0u;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = 64 * b + lane;
        lo[b] = (b < nb && i < n) ? (unsigned)entry_of(inv_perm, lo[b]) : 0u;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = 64 * b + lane;
        if (b < nb && i < n) out[i] = (int)lo[b];
    }
}

// Lists longer than the largest LDS class: bitonic network in its all-ascending form (the first
// step of each merge mirrors), virtual +inf padding, in place on the global segment.
DEV void sort_global_bitonic(unsigned long long *g, int *out, int n, const int *inv_perm, int tid, int nthreads)
{
    int P = 2;
    while (P < n) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {
        const int hk = k >> 1;
        for (int p = tid; p < (P >> 1); p += nthreads) {
            const int blk = (p / hk) * k, o = p % hk;
            const int l = blk + o, r = blk + k - 1 - o;
            if (r < n) {
                const unsigned long long a = g[l], b = g[r];
                if (a > b) { g[l] = b; g[r] = a; }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int p = tid; p < (P >> 1); p += nthreads) {
                const int l = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int r = l | j;
                if (r < n) {
                    const unsigned long long a = g[l], b = g[r];
                    if (a > b) { g[l] = b; g[r] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += nthreads) out[i] = entry_of(inv_perm, (unsigned)g[i]);
}

// ---- k_blend: T6 + T0 epilogue -------------------------------------------------------------------
// One workgroup (4 waves) per 16x16 tile, launched longest list first.  Wave w owns the 8x8
// quadrant w, one pixel per lane.  Per batch of 256 list entries every thread stages one 48-byte
// record in LDS; each wave then ballots the entries whose quadrant mask names it into its own
// compacted queue and walks only those, front to back.  A wave whose 64 pixels are all
// terminated stops blending; the workgroup leaves when all four have.
// Minimum of sigma(dx,dy) = 0.5 (A dx^2 + C dy^2) + B dx dy over a rectangle of pixel centres.
// A convex quadratic whose centre lies outside the rectangle attains its minimum on an edge.
DEV float min_sigma_rect(float mx, float my, float A, float B, float C, float nBoverC, float nBoverA,
                         float xa, float xb, float ya, float yb)
{
    const float dxl = mx - xb, dxh = mx - xa, dyl = my - yb, dyh = my - ya;
    if (dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f) return 0.0f;
    float best = 3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dx = e ? dxh : dxl;
        const float dy = fminf(fmaxf(nBoverC * dx, dyl), dyh);
        best = fminf(best, 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy);
        const float ey = e ? dyh : dyl;
        const float ex = fminf(fmaxf(nBoverA * ey, dxl), dxh);
        best = fminf(best, 0.5f * (A * ex * ex + C * ey * ey) + B * ex * ey);
    }
    return best;
}

// 4-bit mask of the 8x8 quadrants of tile (tx,ty) that the Gaussian can reach.  Bit q = qx + 2 qy.
// Computed by the thread that stages the record (one entry per thread, no divergence).
// A quadrant is dropped only when sigma exceeds the blend stage's skip threshold by a margin
// (0.05) four orders of magnitude above any rounding difference between this estimate and the
// contract's per-pixel sigma, so dropping it never changes a pixel.
DEV unsigned quadrant_mask(int tx, int ty, float mx, float my, float A, float B, float C, float nBoverC,
                           float nBoverA, float thr)
{
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float xa = (float)(tx * SAS_TILE + (q & 1) * 8) + 0.5f;
        const float ya = (float)(ty * SAS_TILE + (q >> 1) * 8) + 0.5f;
        const float ms = min_sigma_rect(mx, my, A, B, C, nBoverC, nBoverA, xa, xa + 7.0f, ya, ya + 7.0f);
        if (!(ms > thr + 0.05f)) m |= 1u << q;
    }
    return m;
}

struct PixState {
    float T, r, g, b, d;
    bool done;
};

// exp for candidate lanes only: 0 <= sigma <= thr <= ln(255)+1e-3, so the contract's clamps on
// the exponent are no-ops and are left out (identical bits, two VALU ops fewer).
DEV float c_expf_neg_small(float x)
{
    const float t = x * 1.4426950408889634f;
    const float n = __builtin_rintf(t);
    const float fr = t - n;
    float p = 0.0013400432653725147f;
    p = fma_(p, fr, 0.009676037356257439f);
    p = fma_(p, fr, 0.05550327152013779f);
    p = fma_(p, fr, 0.2402210682630539f);
    p = fma_(p, fr, 0.6931471824645996f);
    p = fma_(p, fr, 1.0000001192092896f);
    return __builtin_ldexpf(p, (int)n);
}

// `cand` = pixel alive, sigma >= 0 and sigma <= thr.  sigma > thr implies alpha < 1/255 with a
// margin far above rounding, so excluding those lanes takes the same decision as the contract.
template <bool FAST_EXP>
DEV void blend_one(PixState &p, bool cand, float sigma, float op, float cr, float cg, float cb, float dep)
{
    if (cand) {
        float E;
        if (FAST_EXP) E = __expf(-sigma);
        else E = c_expf_neg_small(-sigma);
        const float alpha = fminf(kMaxAlpha, op * E);
        if (!(alpha < kAlphaThr)) {
            const float nT = p.T * (1.0f - alpha);
            if (nT <= kTStop) {
                p.done = true;
            } else {
                const float vis = alpha * p.T;
                p.r = fma_(cr, vis, p.r);
                p.g = fma_(cg, vis, p.g);
                p.b = fma_(cb, vis, p.b);
                p.d = fma_(dep, vis, p.d);
                p.T = nT;
            }
        }
    }
}

template <bool FAST_EXP, bool WANT_MAX>
__global__ __launch_bounds__(256) void k_blend(SasCam c, SasFrame f, SasOutputs o, long long n_gauss)
{
    __shared__ float4 q0[256], q1[256], q2[256];
    __shared__ unsigned s_mask[256];
    __shared__ unsigned short s_queue[4][256];
    __shared__ unsigned s_wmax[4];
    const int tile = f.tile_order[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = tile % c.tw, ty = tile / c.tw;
    const int ix = tx * SAS_TILE + (wv & 1) * 8 + (lane & 7);
    const int iy = ty * SAS_TILE + (wv >> 1) * 8 + (lane >> 3);
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const bool inside = ix < c.W && iy < c.H;
    PixState p = {1.0f, 0.f, 0.f, 0.f, 0.f, !inside};
    bool wdone = __all(p.done);   // wave-uniform

    const long long beg = f.tile_offset[tile];
    long long end = f.tile_offset[tile + 1];
    if (end > f.cap) end = f.cap;

    float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
    unsigned ment = 0u;
    auto fetch = [&](long long at) {
        const long long idx = at + tid;
        ment = 0u;
        if (idx < end) {
            long long id = (unsigned)f.sorted_ids[idx];
            if (id >= n_gauss) id = n_gauss - 1;   // never dereference a bad index
            ra = f.rec[3 * id + 0];
            rb = f.rec[3 * id + 1];
            rc = f.rec[3 * id + 2];
            // approximate reciprocals are fine: the mask is conservative by a 0.05 margin in sigma
            ment = quadrant_mask(tx, ty, ra.x, ra.y, ra.z, ra.w, rb.x, -ra.w * __builtin_amdgcn_rcpf(rb.x),
                                 -ra.w * __builtin_amdgcn_rcpf(ra.z), rb.z);
        }
    };
    if (beg < end) fetch(beg);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (long long at = beg; at < end; at += 256) {
        // the previous batch is fully consumed; leave once every wave has terminated
        if (__syncthreads_and(wdone)) break;
        q0[tid] = ra; q1[tid] = rb; q2[tid] = rc; s_mask[tid] = ment;
        __syncthreads();
        if (at + 256 < end) fetch(at + 256);   // next batch in flight while this one is blended
        if (!wdone) {
            const int cnt = (int)((end - at) < 256 ? (end - at) : 256);
            int qn = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = j * 64 + lane;
                const bool has = e < cnt && ((s_mask[e] >> wv) & 1u);
                const unsigned long long m = __ballot(has);
                if (has) s_queue[wv][qn + (int)__popcll(m & lt_mask)] = (unsigned short)e;
                qn += (int)__popcll(m);
            }
            for (int k = 0; k < qn; ++k) {
                const int e = s_queue[wv][k];
                const float4 A = q0[e], B = q1[e], C = q2[e];
                const float dx = A.x - px, dy = A.y - py;
                const float sg = fma_(0.5f, fma_(B.x * dy, dy, (A.z * dx) * dx), (A.w * dx) * dy);
                const bool cand = !p.done && sg >= 0.0f && sg <= B.z;
                if (__any(cand)) {
                    blend_one<FAST_EXP>(p, cand, sg, B.y, C.x, C.y, C.z, B.w);
                    if (__all(p.done)) break;
                }
            }
            wdone = __all(p.done);
        }
    }

    float ED = 0.0f;
    if (inside) {
        const float a = 1.0f - p.T;
        ED = p.d / fmaxf(a, 1e-10f);
        const long long pix = (long long)iy * c.W + ix;
        const float w = 1.0f - a;
        float v0 = p.r + w * o.bg[0], v1 = p.g + w * o.bg[1], v2 = p.b + w * o.bg[2];
        v0 = fminf(fmaxf(v0, 0.0f), 1.0f);
        v1 = fminf(fmaxf(v1, 0.0f), 1.0f);
        v2 = fminf(fmaxf(v2, 0.0f), 1.0f);
        if (o.rgb) { o.rgb[3 * pix] = v0; o.rgb[3 * pix + 1] = v1; o.rgb[3 * pix + 2] = v2; }
        if (o.alpha) o.alpha[pix] = a;
        if (o.depth) o.depth[pix] = ED;
        if (o.rgb8) {
            o.rgb8[3 * pix] = (uint8_t)(int)floorf(fma_(v0, 255.0f, 0.5f));
            o.rgb8[3 * pix + 1] = (uint8_t)(int)floorf(fma_(v1, 255.0f, 0.5f));
            o.rgb8[3 * pix + 2] = (uint8_t)(int)floorf(fma_(v2, 255.0f, 0.5f));
        }
    }
    if (WANT_MAX) {   // uniform
        float maxed = ED;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) maxed = fmaxf(maxed, __shfl_xor(maxed, d));
        if (lane == 0) s_wmax[wv] = __float_as_uint(maxed);
        __syncthreads();
        if (tid == 0) f.tile_max[tile] = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));   // reduced by k_depth_fill
    }
}

// depth = where(alpha > 0, ED, max ED)  (T0).  alpha == 0 <=> nothing blended <=> ED == 0.
// Every workgroup first reduces the per-tile maxima (a few KB, L2-resident): no extra launch,
// no same-address atomics.  ED >= 0, so the float order is the order of the bit patterns.
__global__ __launch_bounds__(256) void k_depth_fill(const unsigned *tile_max, int tiles, float *depth, long long npix)
{
    __shared__ unsigned s_max[4];
    unsigned m = 0;
    for (int t = threadIdx.x; t < tiles; t += 256) m = max(m, tile_max[t]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    const float mx = __uint_as_float(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256)
        if (depth[p] == 0.0f) depth[p] = mx;
}

}  // namespace

// ---- launchers -------------------------------------------------------------------------------------
void sas_launch_relayout(hipStream_t st, int64_t n, int64_t n_pad, const int *perm, const float *means, const float *quats,
                         const float *scales, const float *cov6, const float *opac, const float *colors,
                         int coeff_floats, int planes, const uint8_t *gid, float4 *g0, float4 *g1, float4 *g2,
                         float4 *col)
{
    if (n <= 0) return;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_relayout, dim3(grid), dim3(256), 0, st, n, n_pad, perm, means, quats, scales, cov6, opac, colors,
                       coeff_floats, planes, gid, g0, g1, g2, col);
}

void sas_launch_project(hipStream_t st, const SasScene &s, const SasCam &c, const SasFrame &f)
{
    if (s.n <= 0) return;
    const unsigned grid = (unsigned)((s.n + 255) / 256);
    switch (s.sh_degree) {
        case 0: hipLaunchKernelGGL(k_project<0>, dim3(grid), dim3(256), 0, st, s, c, f); break;
        case 1: hipLaunchKernelGGL(k_project<1>, dim3(grid), dim3(256), 0, st, s, c, f); break;
        case 2: hipLaunchKernelGGL(k_project<2>, dim3(grid), dim3(256), 0, st, s, c, f); break;
        case 3: hipLaunchKernelGGL(k_project<3>, dim3(grid), dim3(256), 0, st, s, c, f); break;
        default: hipLaunchKernelGGL(k_project<-1>, dim3(grid), dim3(256), 0, st, s, c, f); break;
    }
}

void sas_launch_scan(hipStream_t st, const SasCam &c, const SasFrame &f)
{
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, f, c.tw * c.th);
}

void sas_launch_scatter(hipStream_t st, const SasScene &s, const SasCam &c, const SasFrame &f)
{
    if (s.n <= 0) return;
    const unsigned grid = (unsigned)((s.n + 255) / 256);
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, st, s, c, f);
}

constexpr int kSortMid = 4096, kSortLarge = 16384;

// class 0: >= 4096 (LDS up to 16384, longer lists in place), class 1: 1024..4095, class 2: < 1024.
// The classes are independent and each alone leaves most of the chip idle (few long lists), so
// they run concurrently: classes 0 and 1 on two side streams forked from / joined to `st`.
void sas_launch_sort(hipStream_t st, const SasScene &s, const SasCam &c, const SasFrame &f, const SasSortStreams &ss)
{
    const unsigned tiles = (unsigned)(c.tw * c.th);
    hipStream_t s0 = st, s1 = st;
    if (ss.side[0]) {
        (void)hipEventRecord(ss.fork, st);
        (void)hipStreamWaitEvent(ss.side[0], ss.fork, 0);
        (void)hipStreamWaitEvent(ss.side[1], ss.fork, 0);
        s0 = ss.side[0];
        s1 = ss.side[1];
    }
    hipLaunchKernelGGL((k_sort_radix<kSortLarge, 1024, true>), dim3(tiles), dim3(1024), 0, s0, f, s.inv_perm, 0);
    hipLaunchKernelGGL((k_sort_radix<kSortMid, 256, false>), dim3(tiles), dim3(256), 0, s1, f, s.inv_perm, 1);
    hipLaunchKernelGGL(k_sort_wave, dim3(tiles), dim3(64), 0, st, f, s.inv_perm, 2);
    if (ss.side[0]) {
        (void)hipEventRecord(ss.join[0], s0);
        (void)hipEventRecord(ss.join[1], s1);
        (void)hipStreamWaitEvent(st, ss.join[0], 0);
        (void)hipStreamWaitEvent(st, ss.join[1], 0);
    }
}

void sas_launch_blend(hipStream_t st, const SasScene &s, const SasCam &c, const SasFrame &f, const SasOutputs &o,
                      bool fast_exp, bool want_max)
{
    const unsigned grid = (unsigned)(c.tw * c.th);
    const long long n = s.n > 0 ? s.n : 1;
    if (fast_exp) {
        if (want_max) hipLaunchKernelGGL((k_blend<true, true>), dim3(grid), dim3(256), 0, st, c, f, o, n);
        else hipLaunchKernelGGL((k_blend<true, false>), dim3(grid), dim3(256), 0, st, c, f, o, n);
    } else {
        if (want_max) hipLaunchKernelGGL((k_blend<false, true>), dim3(grid), dim3(256), 0, st, c, f, o, n);
        else hipLaunchKernelGGL((k_blend<false, false>), dim3(grid), dim3(256), 0, st, c, f, o, n);
    }
}

void sas_launch_depth_fill(hipStream_t st, const SasCam &c, const SasFrame &f, float *depth)
{
    const long long npix = (long long)c.W * c.H;
    unsigned grid = (unsigned)((npix + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_depth_fill, dim3(grid), dim3(256), 0, st, (const unsigned *)f.tile_max, c.tw * c.th, depth, npix);
}
