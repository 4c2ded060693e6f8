// sas_kernels.hip -- projection and binning kernels of the render-image hot path, hand-written for
// gfx950 (CDNA4, wave64).  The per-tile stages (depth ordering, compositing) are in sas_tile.hip.
//
// Stage map (the algorithm is gsplat 1.5.2's, the decomposition is not):
//   k_relayout  upload  caller's AoS arrays -> Hilbert-ordered 16-byte planes (once per scene)
//   k_project   T1+T2   one lane per Gaussian: non-temporal 16-byte plane loads, EWA projection,
//                       SH colour, 48-byte projected record, tile rectangle; per-tile counts through
//                       an LDS histogram over the workgroup's tile window (one global atomic per
//                       touched tile)
//               T3      SINGLE-PASS BINNING (round 4, the product path): every tile owns a fixed-stride segment of
//                       the key buffer, so the returning count atomic IS the place of the workgroup's run: the
//                       same window pass emits the (depth bits | storage slot) keys (LDS ranks) -- no scatter launch
//               T5      its LAST workgroup to finish (ticket) scans the counts: tiles ordered by list-length
//                       class, statistics straight to pinned host memory (two-pass binning: also tile_offset and the
//                       scatter cursors) -- no scan kernel, no memset, no read-back copy around a frame
//   k_scatter   T3      two-pass binning only (SAS_FULL_SORT frames, SAS_DIRECT=0, segments over budget): keys into
//                       compact per-tile segments, into the runs the projection reserved per (workgroup, tile); its
//                       front workgroups put the tiles in launch order
//
// ARITHMETIC CONTRACT: every value that reaches an output is produced by the same
// sequence of IEEE binary32 operations as oracle/sas_oracle.c -- explicit __builtin_fmaf where
// the contract fuses, no other contraction (-ffp-contract=off), correctly rounded / and sqrt
// (hipcc default), polynomial exp/log.  No MFMA: nothing here is a dense contraction.
#include "sas_device.h"

#include <cstdlib>
#include <cstring>

#pragma clang fp contract(off)

namespace {

// ---- upload: AoS inputs -> 16-byte planes ------------------------------------------------------
__global__ __launch_bounds__(256) void k_relayout(int64_t n, int64_t n_pad, const int *perm, const float *means,
                                                  const float *quats, const float *scales, const float *cov6,
                                                  const float *opac, const float *colors, int coeff_floats, int planes,
                                                  const uint8_t *gid, float4 *g0, float4 *g1, float4 *g2, float4 *col, uint8_t *gid8)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;   // slot
    if (j >= n) return;
    const int64_t i = perm[j];                                    // caller's index
    float gbits = __uint_as_float(gid ? (unsigned)gid[i] : 0u);
    gid8[j] = gid ? gid[i] : (uint8_t)0;
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
    // Degree by degree, the three channels inside: a channel's sum is the same fused chain in coefficient order as in
    // oracle/sas_oracle.c (sh_to_color), but only one degree's basis values are alive at a time (registers: the pair
    // projection's colour role holds the 48 coefficients across two evaluations).
    float inorm = 1.0f / sqrtf(fma_(dz, dz, fma_(dy, dy, dx * dx)));
    float x = dx * inorm, y = dy * inorm, z = dz * inorm;
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = 0.2820947917738781f * sh[0 * 3 + c];
    if constexpr (DEG >= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = fma_(-x, sh[3 * 3 + c], fma_(z, sh[2 * 3 + c], (-y) * sh[1 * 3 + c]));
            r[c] = fma_(0.48860251190292f, t, r[c]);
        }
    }
    float z2 = z * z;
    float fC1 = fma_(x, x, -(y * y));
    float fS1 = 2.0f * x * y;
    if constexpr (DEG >= 2) {
        float fTmp0B = -1.092548430592079f * z;
        float pSH6 = fma_(0.9461746957575601f, z2, -0.3153915652525201f);
        float pSH7 = fTmp0B * x;
        float pSH5 = fTmp0B * y;
        float pSH8 = 0.5462742152960395f * fC1;
        float pSH4 = 0.5462742152960395f * fS1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r[c] = fma_(pSH4, sh[4 * 3 + c], r[c]);
            r[c] = fma_(pSH5, sh[5 * 3 + c], r[c]);
            r[c] = fma_(pSH6, sh[6 * 3 + c], r[c]);
            r[c] = fma_(pSH7, sh[7 * 3 + c], r[c]);
            r[c] = fma_(pSH8, sh[8 * 3 + c], r[c]);
        }
    }
    if constexpr (DEG >= 3) {
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
            r[c] = fma_(pSH9, sh[9 * 3 + c], r[c]);
            r[c] = fma_(pSH10, sh[10 * 3 + c], r[c]);
            r[c] = fma_(pSH11, sh[11 * 3 + c], r[c]);
            r[c] = fma_(pSH12, sh[12 * 3 + c], r[c]);
            r[c] = fma_(pSH13, sh[13 * 3 + c], r[c]);
            r[c] = fma_(pSH14, sh[14 * 3 + c], r[c]);
            r[c] = fma_(pSH15, sh[15 * 3 + c], r[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = fmaxf(r[c] + 0.5f, 0.0f);
}

// Degree 3 in two halves, for the projection's colour role: the first six coefficient planes (coefficients 0 .. 7 of the three
// channels) are consumed before the last six are loaded, so a lane holds 24 coefficients at a time instead of 48 -- the pair
// kernel (two evaluations per lane) then fits the 64 registers of eight waves per SIMD.  A channel's sum is the same chain in
// coefficient order as sh_to_color's: identical bits.
struct ShHalf {
    float x, y, z;   // (z2, fC1, fS1 are three multiplications away: recomputed by the second half rather than carried)
    float r[3];
};
DEV void sh3_first(const float *a /* floats 0 .. 23 */, float dx, float dy, float dz, ShHalf &S)
{
    float inorm = 1.0f / sqrtf(fma_(dz, dz, fma_(dy, dy, dx * dx)));
    S.x = dx * inorm; S.y = dy * inorm; S.z = dz * inorm;
    const float x = S.x, y = S.y, z = S.z;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = 0.2820947917738781f * a[0 * 3 + c];
        float t = fma_(-x, a[3 * 3 + c], fma_(z, a[2 * 3 + c], (-y) * a[1 * 3 + c]));
        S.r[c] = fma_(0.48860251190292f, t, r);
    }
    const float z2 = z * z;
    const float fS1 = 2.0f * x * y;
    const float fTmp0B = -1.092548430592079f * z;
    const float pSH6 = fma_(0.9461746957575601f, z2, -0.3153915652525201f);
    const float pSH7 = fTmp0B * x;
    const float pSH5 = fTmp0B * y;
    const float pSH4 = 0.5462742152960395f * fS1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = S.r[c];
        r = fma_(pSH4, a[4 * 3 + c], r);
        r = fma_(pSH5, a[5 * 3 + c], r);
        r = fma_(pSH6, a[6 * 3 + c], r);
        r = fma_(pSH7, a[7 * 3 + c], r);
        S.r[c] = r;
    }
}
DEV void sh3_second(const float *b /* floats 24 .. 47: coefficient k of channel c at b[3 (k - 8) + c] */, const ShHalf &S, float *rgb)
{
    const float x = S.x, y = S.y, z = S.z;
    const float z2 = z * z, fC1 = fma_(x, x, -(y * y)), fS1 = 2.0f * x * y;   // (the same operations on the same values as in sh_to_color)
    const float pSH8 = 0.5462742152960395f * fC1;
    const float fTmp0C = fma_(-2.285228997322329f, z2, 0.4570457994644658f);
    const float fTmp1B = 1.445305721320277f * z;
    const float fC2 = fma_(x, fC1, -(y * fS1));
    const float fS2 = fma_(x, fS1, y * fC1);
    const float pSH12 = z * fma_(1.865881662950577f, z2, -1.119528997770346f);
    const float pSH13 = fTmp0C * x;
    const float pSH11 = fTmp0C * y;
    const float pSH14 = fTmp1B * fC1;
    const float pSH10 = fTmp1B * fS1;
    const float pSH15 = -0.5900435899266435f * fC2;
    const float pSH9 = -0.5900435899266435f * fS2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = S.r[c];
        r = fma_(pSH8, b[0 * 3 + c], r);
        r = fma_(pSH9, b[1 * 3 + c], r);
        r = fma_(pSH10, b[2 * 3 + c], r);
        r = fma_(pSH11, b[3 * 3 + c], r);
        r = fma_(pSH12, b[4 * 3 + c], r);
        r = fma_(pSH13, b[5 * 3 + c], r);
        r = fma_(pSH14, b[6 * 3 + c], r);
        r = fma_(pSH15, b[7 * 3 + c], r);
        rgb[c] = fmaxf(r + 0.5f, 0.0f);
    }
}

// Visit every tile of a rectangle.  Small rectangles are walked by their own lane; a lane with a
// large rectangle hands it to the whole wave (64 lanes stride over its tiles) so that one
// screen-filling Gaussian does not serialise a wave.  v0/v1 are the owning lane's payload; they
// are broadcast while the wave is still convergent (a cross-lane read of an inactive lane
// returns 0), then handed to emit(tile, v0, v1).  Must be reached by all 64 lanes.
constexpr int kSmallRect = 8;
// timing experiments only (-DSAS_TUNE_PABL=mask: 1 no count atomics, 2 no key stores, 4 no LDS atomics, 8 no record stores, 16 one tile per Gaussian,
// 32 colour blocks leave at once, 64 geometry blocks leave at once, 128 no binning (geometry = T1 + records + ticket + tail)): wrong frames
#ifndef SAS_TUNE_PABL
#define SAS_TUNE_PABL 0
#endif

#ifdef SAS_TUNE_PTIME
// A/B builds only (tools/proj_time.py): per geometry workgroup, EXCLUSIVE laps of thread 0's wall time (100 MHz ticks):
// [0] loads + T1 + record stores issued, [1] window, [2] histogram zero + cull/count pass, [3] count atomics (returning),
// [4] emit pass, [5] rectangles outside the window, [6] visible counts, [7] wait for the stores' acknowledgements + barrier,
// [8] ticket, [9] tail; [14] begin, [15] end.  Colour workgroups: g_dbg_pcol[2 wg] begin, [2 wg + 1] end.
constexpr int kDbgPMax = 8192;
__device__ unsigned long long g_dbg_plap[16 * kDbgPMax];
__device__ unsigned long long g_dbg_pcol[2 * kDbgPMax];
extern "C" int sas_debug_proj_laps(unsigned long long *geo, unsigned long long *col, int n)
{
    const size_t m = (size_t)(n < kDbgPMax ? n : kDbgPMax);
    if (hipMemcpyFromSymbol(geo, HIP_SYMBOL(g_dbg_plap), sizeof(unsigned long long) * 16 * m) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(col, HIP_SYMBOL(g_dbg_pcol), sizeof(unsigned long long) * 2 * m) == hipSuccess ? 0 : -1;
}
#define PL_LAP(i) do { if (threadIdx.x == 0 && pl_wg_ < (unsigned)kDbgPMax) { const unsigned long long now_ = wall_clock64(); g_dbg_plap[16 * pl_wg_ + (i)] = now_ - pl_t_; pl_t_ = now_; } } while (0)
#else
#define PL_LAP(i) do { } while (0)
#endif

template <typename F>
DEV void for_each_tile(bool active, int x0, int x1, int y0, int y1, int tw, unsigned v0, unsigned v1, F emit)
{
    const int lane = threadIdx.x & 63;
    const int w = x1 - x0;
    const int area = active ? w * (y1 - y0) : 0;
    if (area > 0 && area <= kSmallRect) {
#pragma unroll 1
        for (int ty = y0; ty < y1; ++ty)
#pragma unroll 1
            for (int tx = x0; tx < x1; ++tx) emit(ty * tw + tx, v0, v1);
    }
    unsigned long long big = __ballot(area > kSmallRect);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bx0 = lane_get(x0, src), by0 = lane_get(y0, src);   // (src is uniform: v_readlane)
        const int bw = lane_get(w, src), ba = lane_get(area, src);
        const unsigned b0 = lane_get(v0, src), b1 = lane_get(v1, src);
#pragma unroll 1
        for (int i = lane; i < ba; i += 64) emit((by0 + i / bw) * tw + bx0 + i % bw, b0, b1);
    }
}

// Workgroup screen window.  The scene is stored along a Hilbert curve, so the 256 Gaussians of a
// workgroup land in a compact block of tiles: their per-tile counts are accumulated in an LDS
// histogram over that window and leave the CU as ONE global atomic per touched tile instead of
// one per intersection (device-scope atomics execute at the memory side: ~10 G/s scattered).
// Gaussians with a large rectangle stay out of the window and use the wave-cooperative walk.
#ifndef SAS_TUNE_HIST
#define SAS_TUNE_HIST SAS_WIN_BINS
#endif
static_assert(SAS_TUNE_HIST <= SAS_WIN_BINS, "wg_base holds SAS_WIN_BINS entries per workgroup");
#ifndef SAS_TUNE_WINRECT
#define SAS_TUNE_WINRECT 64
#endif
constexpr int kHistBins = SAS_TUNE_HIST;      // 8 KiB of LDS
static_assert(SAS_TUNE_WINRECT <= 64, "count_tiles keeps one mask bit per tile of a window rectangle");
constexpr int kWinRect = SAS_TUNE_WINRECT;    // largest rectangle (tiles) that takes part in the window

struct Window {
    int X0, Y0, ww, area;   // origin, width, bin count (0: no participant)
    bool fits;
};

DEV Window wg_window(bool part, int x0, int x1, int y0, int y1, int *s_win)
{
    int mnx = part ? x0 : 0x7fffffff, mny = part ? y0 : 0x7fffffff;
    int mxx = part ? x1 : 0, mxy = part ? y1 : 0;
    mnx = wave_min_i32(mnx); mny = wave_min_i32(mny);   // (DPP steps: sas_device.h)
    mxx = wave_max_i32(mxx); mxy = wave_max_i32(mxy);
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

// The projection's form of wg_window: ONE barrier instead of two and no LDS atomics -- every wave leaves its four extrema in
// its own words of s_win16, all threads reduce the sixteen; the window's bins (all kHistBins of them: eight words per thread)
// are zeroed in front of the same barrier, so the count pass starts right behind it.  (A geometry workgroup is resident for
// ~17 us, of which its barriers and their stragglers are a good part: tools/proj_time.py.)
DEV Window wg_window_zeroed(bool part, int x0, int x1, int y0, int y1, int *s_win16, int *s_hist)
{
    int mnx = part ? x0 : 0x7fffffff, mny = part ? y0 : 0x7fffffff;
    int mxx = part ? x1 : 0, mxy = part ? y1 : 0;
    mnx = wave_min_i32(mnx); mny = wave_min_i32(mny);   // (DPP steps: sas_device.h)
    mxx = wave_max_i32(mxx); mxy = wave_max_i32(mxy);
    if ((threadIdx.x & 63) == 0) *reinterpret_cast<int4 *>(s_win16 + 4 * (threadIdx.x >> 6)) = make_int4(mnx, mny, mxx, mxy);
    static_assert(kHistBins == 2048, "eight bins per thread");
    reinterpret_cast<int4 *>(s_hist)[threadIdx.x] = make_int4(0, 0, 0, 0);
    reinterpret_cast<int4 *>(s_hist)[256 + threadIdx.x] = make_int4(0, 0, 0, 0);
    __syncthreads();
    const int4 a = reinterpret_cast<const int4 *>(s_win16)[0], b = reinterpret_cast<const int4 *>(s_win16)[1],
               c = reinterpret_cast<const int4 *>(s_win16)[2], d = reinterpret_cast<const int4 *>(s_win16)[3];
    Window w;
    w.X0 = min(min(a.x, b.x), min(c.x, d.x));
    w.Y0 = min(min(a.y, b.y), min(c.y, d.y));
    const int X1 = max(max(a.z, b.z), max(c.z, d.z)), Y1 = max(max(a.w, b.w), max(c.w, d.w));
    w.ww = X1 - w.X0;
    w.area = (X1 > w.X0 && Y1 > w.Y0) ? w.ww * (Y1 - w.Y0) : 0;
    w.fits = w.area > 0 && w.area <= kHistBins;
    return w;
}

// ---- k_project: T1 + T2 + T3 + T5 ----------------------------------------------------------------
// camera-dependent results of one Gaussian for one view
struct ViewGeom {
    bool vis;
    int x0, x1, y0, y1;
    float mx, my, ca, cb, ccn, thr, z, rx, ry;
};

// T1 after the camera transform: cull, EWA projection, conic, radii, tile rectangle
DEV void project_view(const SasCam &c, const float *cov, float op, float x, float y, float z, ViewGeom &g)
{
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
    if (!(det > 0.0f)) return;
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
    if (!keep) return;
    g.vis = true;
    // T3 tile rectangle
    const float ts = (float)c.tile_px;   // 16, or 8 (quad layout): powers of two, so the 8-pixel rectangle halves to the 16-pixel one exactly
    const float trx = rx / ts, try_ = ry / ts;
    const float ttx = mx / ts, tty = my / ts;
    const float twf = (float)c.tw, thf = (float)c.th;
    g.x0 = (int)fminf(fmaxf(floorf(ttx - trx), 0.0f), twf);
    g.x1 = (int)fminf(fmaxf(ceilf(ttx + trx), 0.0f), twf);
    g.y0 = (int)fminf(fmaxf(floorf(tty - try_), 0.0f), thf);
    g.y1 = (int)fminf(fmaxf(ceilf(tty + try_), 0.0f), thf);
    // Quad layout (8-pixel bins), opacity NOT <= 1 (no sigmoid produces one: garbage in, but the layouts must agree on it): such a
    // Gaussian passes the 1/255 test OUTSIDE its 3.33-sigma box, and the contract evaluates it on every pixel of the 16-pixel
    // tiles of its rectangle -- its 8-pixel rectangle is widened to whole 16-pixel tiles.  (Opacities <= 1 reach nothing outside
    // the box: alpha <= exp(-3.33^2 / 2) < 1/255.)
    if (c.tile_px == 8 && !(op <= 1.0f)) {
        g.x0 &= ~1; g.y0 &= ~1;
        g.x1 = min((g.x1 + 1) & ~1, c.tw); g.y1 = min((g.y1 + 1) & ~1, c.th);
    }
    g.mx = mx; g.my = my; g.ca = ca; g.cb = cb; g.ccn = ccn; g.z = z; g.rx = rx; g.ry = ry;
    // conservative skip threshold for the blend stage: alpha >= 1/255 implies
    // sigma <= ln(255 op) + rounding; 1e-3 is > 100x the worst rounding.
    // The loop clamps the exponent's argument at -86: once ln(255 o) reaches that, op * exp(-86) >= 1/255 and EVERY pixel passes,
    // whatever its sigma (opacities beyond 8.7e34, +Inf, NaN -- c_logf reads >= 88 off their exponent bits): nothing may be
    // culled for such a Gaussian, in the binning or in the block masks.
    const float t = lnq + 1e-3f;
    g.thr = t < 85.9f ? t : __builtin_inff();
}

// ---- exact tile culling (single-pass binning) --------------------------------------------------------------------
// T3 bins a Gaussian into every tile of its bounding rectangle; T6 then composites it onto a pixel only where
// alpha = o exp(-sigma) >= 1/255, i.e. sigma <= ln(255 o).  A tile none of whose pixel centres satisfies that receives
// nothing from the Gaussian, whatever its place in the list: leaving it out of the list changes no pixel.  A fifth of
// a 1080p frame's intersections are of that kind (the rectangle's corners).  The test is the minimum of sigma over the
// tile's box of pixel centres: sigma is a convex quadratic with its minimum (0) at the mean, so over a box that does
// not contain the mean the minimum lies on the edges facing the mean -- the edge x = (box x nearest the mean), where
// the best y is -B dx / C clamped to the box, and likewise for y.  Margins: 0.05 (as block_mask16) plus 2e-5 of the
// two square terms, which bound the terms' magnitudes (340 ulp: the contract's own polynomial about the tile centre
// rounds terms of that size, and what it decides is the contract) -- a tile is dropped only if the minimum exceeds
// the threshold by more than both.  tests/tools/tile_cull_model.py restates the test in NumPy for the CPU suite.
struct CullGeom {
    float mx, my, ha, b, hc, nba, nbc, lim;
};
DEV CullGeom cull_geom(const ViewGeom &g)
{
    return CullGeom{g.mx, g.my, 0.5f * g.ca, g.cb, 0.5f * g.ccn, -g.cb / g.ca, -g.cb / g.ccn, g.thr + 0.05f};
}
// The test for one row of tiles: what depends on the row alone is taken once (CullRow), a tile then costs some 20
// instructions.  |B dx dy| <= A dx^2 / 2 + C dy^2 / 2 (the conic is positive definite), so twice the sum of the two
// square terms bounds the magnitude of all three: the relative margin is taken from that sum.
struct CullRow {
    float ly, hy, dyc, c2, t, bdyc;
};
DEV CullRow cull_row(const CullGeom &q, int ty, float px)
{
    CullRow r;
    r.ly = ((float)ty * px + 0.5f) - q.my;   // pixel centres of the row, relative to the mean
    r.hy = r.ly + (px - 1.0f);
    r.dyc = __builtin_amdgcn_fmed3f(0.0f, r.ly, r.hy);
    r.c2 = q.hc * r.dyc * r.dyc;
    r.t = q.nba * r.dyc;
    r.bdyc = q.b * r.dyc;
    return r;
}
DEV bool tile_reached(const CullGeom &q, const CullRow &r, int tx, float px)
{
    const float lx = ((float)tx * px + 0.5f) - q.mx, hx = lx + (px - 1.0f);
    const float dxc = __builtin_amdgcn_fmed3f(0.0f, lx, hx);
    const float dys = __builtin_amdgcn_fmed3f(q.nbc * dxc, r.ly, r.hy), dxs = __builtin_amdgcn_fmed3f(r.t, lx, hx);
    // edge x = dxc, y free:  s1 = the square terms, q1 = sigma
    const float t1 = q.ha * dxc, t3 = q.hc * dys;
    const float s1 = fma_(t1, dxc, t3 * dys);
    const float q1 = fma_(q.b * dxc, dys, s1);
    const float m1 = fma_(-2e-5f, s1, q1);
    // edge y = dyc, x free
    const float s2 = fma_(q.ha * dxs, dxs, r.c2);
    const float q2 = fma_(r.bdyc, dxs, s2);
    const float m2 = fma_(-2e-5f, s2, q2);
    return !(m1 > q.lim && m2 > q.lim);   // (a NaN anywhere keeps the tile)
}
DEV bool tile_reached(const CullGeom &q, int tx, int ty, float px) { return tile_reached(q, cull_row(q, ty, px), tx, px); }
// for_each_tile over the tiles the Gaussian can reach (cull: uniform; off = every tile of the rectangle)
template <typename F>
DEV void for_each_reached_tile(bool active, bool cull, const CullGeom &q, float px, int x0, int x1, int y0, int y1, int tw,
                               unsigned v0, unsigned v1, F emit)
{
    const int lane = threadIdx.x & 63;
    const int w = x1 - x0;
    const int area = active ? w * (y1 - y0) : 0;
    if (area > 0 && area <= kSmallRect) {
#pragma unroll 1
        for (int ty = y0; ty < y1; ++ty) {
            const CullRow r = cull_row(q, ty, px);
#pragma unroll 1
            for (int tx = x0; tx < x1; ++tx)
                if (!cull || tile_reached(q, r, tx, px)) emit(ty * tw + tx, v0, v1);
        }
    }
    unsigned long long big = __ballot(area > kSmallRect);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bx0 = lane_get(x0, src), by0 = lane_get(y0, src);   // (src is uniform: v_readlane)
        const int bw = lane_get(w, src), ba = lane_get(area, src);
        const unsigned b0 = lane_get(v0, src), b1 = lane_get(v1, src);
        const CullGeom o{lane_get(q.mx, src), lane_get(q.my, src), lane_get(q.ha, src), lane_get(q.b, src),
                         lane_get(q.hc, src), lane_get(q.nba, src), lane_get(q.nbc, src), lane_get(q.lim, src)};
#pragma unroll 1
        for (int i = lane; i < ba; i += 64) {
            const int ty = by0 + i / bw, tx = bx0 + i % bw;
            if (!cull || tile_reached(o, tx, ty, px)) emit(ty * tw + tx, b0, b1);
        }
    }
}

// a store the projection's tail may read from another XCD (see the hand-off rule in count_tiles)
DEV void agent_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// per-tile counts of one view: LDS histogram over the workgroup's window, one global atomic per
// touched tile; then the workgroup's visible count.  Reached by all 256 threads.
DEV void count_tiles(const SasFrame &f, int tw, int tile_px, const ViewGeom &g, unsigned slot, unsigned wg /* this workgroup's chunk of 256 Gaussians */, int *s_win, int *s_hist, int *s_base, int *s_nvis, unsigned long long &pl_t_)
{
    const unsigned pl_wg_ = wg;
    (void)pl_wg_;
    const bool vis = g.vis;
    const int seg = f.seg;   // (uniform) > 0: single-pass binning -- this workgroup EMITS its keys as well
    const bool cull = f.cull != 0;   // (uniform; single-pass frames only) tiles the Gaussian cannot reach are left out: tile_reached
    const CullGeom cg = cull_geom(g);
    const float px = (float)tile_px;
    const unsigned long long key = ((unsigned long long)__float_as_uint(g.z) << 32) | (unsigned long long)slot;
    const int x0 = g.x0, y0 = g.y0;
    const int x1 = (SAS_TUNE_PABL & 16) ? min(g.x1, g.x0 + 1) : g.x1, y1 = (SAS_TUNE_PABL & 16) ? min(g.y1, g.y0 + 1) : g.y1;   // (16: one tile per Gaussian)
    const int rect_area = vis ? (x1 - x0) * (y1 - y0) : 0;
    const bool in_win = rect_area > 0 && rect_area <= kWinRect;
    const Window w = wg_window_zeroed(in_win, x0, x1, y0, y1, s_win, s_hist);   // (the bins are zero behind its barrier)
    PL_LAP(1);
    unsigned long long reached = 0ull;   // the tiles of the rectangle that were counted (kept for the emit pass)
    if (w.fits) {
        if (in_win) {
            int k = 0;   // tile k of the rectangle, row by row (at most kWinRect = 64 of them: one mask bit each)
            for (int ty = y0; ty < y1; ++ty) {
                const CullRow r = cull_row(cg, ty, px);
                for (int tx = x0; tx < x1; ++tx, ++k) {
                    const int b = (ty - w.Y0) * w.ww + (tx - w.X0);
                    if (cull && !tile_reached(cg, r, tx, px)) continue;
                    reached |= 1ull << k;
                    if (!(SAS_TUNE_PABL & 4) && SAS_IN(b, kHistBins, 101)) atomicAdd(&s_hist[b], 1);
                }
            }
        }
        __syncthreads();
        PL_LAP(2);
        // one RETURNING atomic per touched tile: the workgroup's run inside the tile's segment is reserved here, and
        // k_scatter (same workgroup, same window) reads where it starts instead of reserving it itself
        int *wb = f.wg_base + (size_t)wg * SAS_WIN_BINS;
        {   // bin b = row * ww + col of the window, walked in steps of 256 bins without a division per step
            const unsigned ww = (unsigned)w.ww, tid = threadIdx.x;
            unsigned row = tid / ww, col = tid - row * ww;
            const unsigned drow = 256u / ww, dcol = 256u - drow * ww;
#pragma clang loop unroll(disable)
            for (unsigned b = tid; b < (unsigned)w.area; b += 256u) {
                const int cnt = s_hist[b];
                const int tile = (w.Y0 + (int)row) * tw + w.X0 + (int)col;
                const int base = (!(SAS_TUNE_PABL & 1) && cnt && SAS_IN(tile, f.n_tiles, 102)) ? atomicAdd(&f.tile_count[tile], cnt) : 0;
                if (seg > 0) {   // where the workgroup's run starts inside the tile's segment stays in LDS; the bin becomes its rank counter
                    s_base[b] = base;
                    s_hist[b] = 0;
                } else {
                    wb[b] = base;
                }
                row += drow;
                col += dcol;
                if (col >= ww) { col -= ww; ++row; }
            }
        }
        if (seg > 0) {
            // single-pass binning: position = tile segment + the run's start + the key's rank inside the run (LDS atomic)
            __syncthreads();
            PL_LAP(3);
            if (in_win)
#pragma unroll 1
                for (int ty = y0; ty < y1; ++ty)
#pragma unroll 1
                    for (int tx = x0; tx < x1; ++tx, reached >>= 1) {
                        const int b = (ty - w.Y0) * w.ww + (tx - w.X0);
                        if (!SAS_IN(b, kHistBins, 117)) continue;
                        if (!(reached & 1ull)) continue;   // (what the count pass found)
                        const int pos = s_base[b] + ((SAS_TUNE_PABL & 4) ? (int)(threadIdx.x & 63) : atomicAdd(&s_hist[b], 1));
                        // pos >= seg: the tile has outgrown its segment (the tail reports it; the frame is rendered again)
                        if (!(SAS_TUNE_PABL & 2) && pos < seg && SAS_IN((long long)(ty * tw + tx) * seg + pos, f.cap, 118)) f.keys[(long long)(ty * tw + tx) * seg + pos] = key;
                    }
        }
    } else if (threadIdx.x == 0 && w.area > 0) {
        atomicAdd(&f.stats[5], 1u);
    }
    PL_LAP(4);
    if (seg > 0) {   // rectangles outside the window scheme: one returning atomic per intersection, the key goes where it points
        for_each_reached_tile(vis && !(w.fits && in_win), cull, cg, px, x0, x1, y0, y1, tw, (unsigned)key, (unsigned)(key >> 32), [&](int tile, unsigned lo, unsigned hi) {
            if (!SAS_IN(tile, f.n_tiles, 103)) return;
            const int pos = atomicAdd(&f.tile_count[tile], 1);
            if (pos < seg && SAS_IN((long long)tile * seg + pos, f.cap, 119)) f.keys[(long long)tile * seg + pos] = ((unsigned long long)hi << 32) | lo;
        });
    } else {
        for_each_tile(vis && !(w.fits && in_win), x0, x1, y0, y1, tw, 0u, 0u,
                      [&](int tile, unsigned, unsigned) { if (SAS_IN(tile, f.n_tiles, 103)) atomicAdd(&f.tile_big[tile], 1); });
    }
    PL_LAP(5);
    // visible count (and the contract's 16-pixel intersections where the lists are not T3's).
    const unsigned long long vb = __ballot(vis);
    int a16 = 0;
    if (f.wg_isect16) {   // (uniform) 8-pixel binning, culled lists: the frame still reports the intersections with the contract's 16-pixel tiles
        a16 = !vis ? 0 : tile_px == 8 ? (((x1 + 1) >> 1) - (x0 >> 1)) * (((y1 + 1) >> 1) - (y0 >> 1)) : rect_area;
        a16 = wave_sum_i32(a16);
    }
    // HAND-OFF RULE (projection workgroups -> the tail, possibly on another XCD, whose L2 is not coherent with this one):
    // everything the tail reads from other workgroups must be written by an AGENT-scope atomic (performed where all XCDs
    // see it): the per-tile counts (atomicAdd on tile_count / tile_big), the window-miss counter, and the sums below.
    // A plain store here -- e.g. a vectorised store of counts -- would sit in this XCD's L2 and reach the
    // tail stale, silently; the bounds build checks the rule's effect (the tail: the sum it collects against a counter that
    // every wave also adds its count to atomically).
    if (seg > 0) {
        // single-pass frames: every WAVE adds its sums to words 1 and 2 of the workgroup's ticket line (64 lines, ~240 adds
        // each per frame: far from the rate at which one address serialises), where the tail's first wave finds all of them --
        // no LDS round, no barrier (the two barriers of a per-workgroup sum cost a geometry workgroup 1.2 of its 17 us)
        if ((threadIdx.x & 63) == 0) {
            unsigned *tk = f.tickets + 32u * (wg & 63u);
            if (vb) atomicAdd(&tk[1], (unsigned)__popcll(vb));
            if (a16) atomicAdd(&tk[2], (unsigned)a16);
#ifdef SAS_DEBUG_BOUNDS
            if (vb) atomicAdd(&f.stats[6], (unsigned)__popcll(vb));
#endif
        }
    } else {
        // two-pass frames: one plain (agent-scope) store per workgroup; their tail adds the per-workgroup counts up
        if (threadIdx.x == 0) { s_nvis[0] = 0; s_nvis[1] = 0; }
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && vb) atomicAdd(&s_nvis[0], (int)__popcll(vb));
        if ((threadIdx.x & 63) == 0 && a16) atomicAdd(&s_nvis[1], a16);
        __syncthreads();
        if (threadIdx.x == 0) {
            agent_store(&f.wg_vis[wg], s_nvis[0]);
            if (f.wg_isect16) agent_store(&f.wg_isect16[wg], s_nvis[1]);
#ifdef SAS_DEBUG_BOUNDS
            atomicAdd(&f.stats[6], (unsigned)s_nvis[0]);   // (bounds build only) the same count through a device atomic: the tail compares
#endif
        }
    }
    PL_LAP(6);
}

// ---- the projection's tail (T5): offsets, cursors, list-length classes, statistics -----------------------------------
// Run by the 256 threads of the projection's last workgroup, which every later kernel of the frame waits for: written
// for few instructions (a lone wave issues one per ~7 cycles) and few dependent memory round trips.  Thread t owns the
// `per` consecutive tiles from t * per (a multiple of four: 16-byte loads and stores; the count arrays are zero-padded
// past their ends).  Two phases, each reading the thread's counts (16 tiles per 16-byte-load burst: L2 hits the second
// time): held in registers across the barrier they raised the PROJECTION's register count and cost it occupancy.
// The tile kernel starts its long lists first: tiles are ordered by list-length class (floor(log2) + 1, 15 and up
// together), longest first.  Here only the class SIZES are counted (LDS bins replicated 32 times: the lanes of a wave
// then hit different words -- same-address LDS atomics serialise, the scan kernel this replaces spent most of its
// 14 us on them) and turned into the classes' first positions (class_cursor); the tiles are put in place by the front
// workgroups of k_scatter, off this critical path.
DEV int len_class(int v) { return v ? min(15, 32 - __clz(v)) : 0; }

// ---- the tail of a single-pass frame (the product path), round 5 ----------------------------------------------------
// Per-workgroup laps (tools/proj_time.py) showed the tail of rounds 3-4 taking 17-21 us on the launch's critical path, not
// the ~10 its ablation had suggested: three passes over the counts (each a dependent burst of L2 round trips per thread),
// sixteen sequential loads per thread for the visible counts, and the chip idle meanwhile.  This form reads every count
// ONCE, in one volley of coalesced 16-byte loads (thread t takes tiles 4 (t + 256 j): all of a thread's loads are in
// flight together), keeps a list-length class byte per GROUP of four tiles in LDS (the geometry role's 16 KiB of window
// bins are free by now: 40 448 groups; larger frames re-read their counts for the second pass), and places the groups from there.
// lds: 2 * kHistBins ints.
constexpr int kTailScratch = 16 + 512 + 512 + 512 + 16;   // cross-wave words, class bins, (class, copy) offsets, ranks, class totals
DEV void scan_tail_single(const SasFrame *fp, int *lds, unsigned pl_wg_, unsigned long long &pl_t_)
{
    (void)pl_wg_;
    const SasFrame &f = *fp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles = f.n_tiles;
    int *s_w = lds;              // [16]
    int *s_bins = lds + 16;      // [16 classes (descending)][32 copies]
    int *s_off = s_bins + 512, *s_rank = s_off + 512, *s_ctot = s_rank + 512;
    unsigned char *s_clsb = reinterpret_cast<unsigned char *>(lds + kTailScratch);   // class of group w = tiles 4 w .. 4 w + 3
#if SAS_TILE_GROUP == 1
    constexpr int kClsGroups = 2 * kHistBins - kTailScratch;   // (a word of four class bytes per group)
#else
    constexpr int kClsGroups = 4 * (2 * kHistBins - kTailScratch);
#endif
    const int words = (tiles + 3) >> 2;       // groups of four tiles = 16-byte words of counts
    const bool in_lds = words <= kClsGroups;   // (uniform)
    const int4 *cnt4 = reinterpret_cast<const int4 *>(f.tile_count);   // (zero-padded past its end: sas_count_stride)
    // ---- the frame's sums of visible Gaussians and of the contract's 16-pixel intersections: every geometry workgroup added
    //      its own to words 1 and 2 of its ticket line (64 lines: no address takes more than ~60 adds); wave 0 collects
    //      them -- agent-scope loads, performed where the adds were -- and leaves the lines zeroed for the next frame
    int nvis = 0, n16 = 0;
    if (wv == 0) {
        unsigned *tk = f.tickets + 32u * (unsigned)lane;
        nvis = (int)__hip_atomic_load(&tk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n16 = (int)__hip_atomic_load(&tk[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk[1] = 0u;
        tk[2] = 0u;
    }
    s_bins[tid] = 0;
    s_bins[256 + tid] = 0;
    s_rank[tid] = 0;
    s_rank[256 + tid] = 0;
    __syncthreads();
    PL_LAP(10);
    // ---- pass A: totals, longest list, class sizes, class bytes.  The unit of the ORDER is a group of four consecutive tiles
    //      (one 16-byte word of counts; neighbours in a tile row, whose lists are alike), classed by its longest list: a
    //      quarter of the LDS atomics, class bytes and order stores of a per-tile order, on a path where a lone workgroup
    //      issues one instruction per ~7 cycles and wave.  Eight 16-byte loads per thread in one volley (8 192 tiles).
    int total = 0, maxlen = 0;
#pragma unroll 1
    for (int w0 = tid; w0 < words; w0 += 8 * 256) {
        int4 c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = (w0 + 256 * j < words) ? cnt4[w0 + 256 * j] : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = w0 + 256 * j;
            if (w >= words) break;
            // (counts past the last tile are zero: the arrays are zero-padded)
            total += (c[j].x + c[j].y) + (c[j].z + c[j].w);
            const int gmax = max(max(c[j].x, c[j].y), max(c[j].z, c[j].w));
            maxlen = max(maxlen, gmax);
#if SAS_TILE_GROUP == 1   // A/B build: the order's unit is a tile (four class bytes per word, four times the atomics and order stores)
            const int cc[4] = {c[j].x, c[j].y, c[j].z, c[j].w};
            unsigned packed = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (4 * w + q >= tiles) break;
                const int cl = 15 - len_class(cc[q]);
                packed |= (unsigned)cl << (8 * q);
                atomicAdd(&s_bins[cl * 32 + (lane & 31)], 1);
            }
            if (in_lds) reinterpret_cast<unsigned *>(s_clsb)[w] = packed;
#else
            const int cl = 15 - len_class(gmax);
            atomicAdd(&s_bins[cl * 32 + (lane & 31)], 1);
            if (in_lds) s_clsb[w] = (unsigned char)cl;
#endif
        }
    }
    total = wave_sum_i32(total); maxlen = wave_max_i32(maxlen);   // (DPP steps: sas_device.h)
    nvis = wave_sum_i32(nvis); n16 = wave_sum_i32(n16);
    if (lane == 0) { s_w[wv] = total; s_w[4 + wv] = maxlen; }
    if (tid == 0) { s_w[8] = nvis; s_w[12] = n16; }
    __syncthreads();
    PL_LAP(11);
    // ---- per (class, copy): inclusive prefix over the class's 32 copies (a half wave each); the class totals
    int my_incl[2], my_v[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int idx = 256 * half + tid;
        const int v = s_bins[idx];
        // (the scan over the wave's 64 lanes, less the lower half's total in the upper half: two scans of 32 lanes)
        const int incl64 = (int)wave_inclusive_sum_u32((unsigned)v);
        const int incl = lane < 32 ? incl64 : incl64 - lane_get(incl64, 31);
        my_incl[half] = incl;
        my_v[half] = v;
        if ((lane & 31) == 31) s_ctot[idx >> 5] = incl;
    }
    __syncthreads();
    int cstart[2] = {0, 0};   // where the class of this thread's (class, copy) bin starts in the order: the classes in front of it
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int t = s_ctot[j];
        if (j < (tid >> 5)) cstart[0] += t;
        if (j < 8 + (tid >> 5)) cstart[1] += t;
    }
    s_off[tid] = cstart[0] + my_incl[0] - my_v[0];
    s_off[256 + tid] = cstart[1] + my_incl[1] - my_v[1];
    __syncthreads();
    PL_LAP(12);
    // ---- pass B: the group ORDER (longest classes first; the order inside a class is free): position = the (class, copy)
    //      offset + the group's rank among the groups this copy counted.  tile_order[p] = the p-th GROUP (k_tile_lazy: workgroup
    //      b renders tile 4 tile_order[b / 4] + b % 4)
#pragma unroll 4
    for (int w = tid; w < words; w += 256) {
#if SAS_TILE_GROUP == 1
        unsigned packed;
        if (in_lds) {
            packed = reinterpret_cast<const unsigned *>(s_clsb)[w];
        } else {
            const int4 c = cnt4[w];
            packed = (unsigned)(15 - len_class(c.x)) | ((unsigned)(15 - len_class(c.y)) << 8) | ((unsigned)(15 - len_class(c.z)) << 16) |
                     ((unsigned)(15 - len_class(c.w)) << 24);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (4 * w + q >= tiles) break;
            const int b = (int)((packed >> (8 * q)) & 255u) * 32 + (lane & 31);
            const int pos = s_off[b] + atomicAdd(&s_rank[b], 1);
            if (SAS_IN(pos, tiles, 105)) f.tile_order[pos] = 4 * w + q;
        }
#else
        int cl;
        if (in_lds) {
            cl = (int)s_clsb[w];
        } else {
            const int4 c = cnt4[w];
            cl = 15 - len_class(max(max(c.x, c.y), max(c.z, c.w)));
        }
        const int b = cl * 32 + (lane & 31);
        const int pos = s_off[b] + atomicAdd(&s_rank[b], 1);
        if (SAS_IN(pos, words, 105)) f.tile_order[pos] = w;
#endif
    }
    if (tid == 0) {
        const int carry = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const int maxlen_all = max(max(s_w[4], s_w[5]), max(s_w[6], s_w[7]));
        const int nvis_all = s_w[8];
        const int n16_all = s_w[12];
        int start = 0;
        for (int k = 0; k < 16; ++k) {   // descending classes: entry k = class 15 - k
            f.class_cursor[k] = start;
            if (k == 3) f.sort_class[1] = start;
            if (k == 5) f.sort_class[2] = start;
            start += s_ctot[k];
        }
        f.sort_class[0] = 0;
        f.sort_class[3] = tiles;
        f.sort_class[4] = 0;
        f.sort_class[5] = tiles;
        f.tile_offset[tiles] = carry;
        unsigned *h = f.stats_host;      // the statistics, straight to pinned host memory
        h[0] = (unsigned)nvis_all;
        h[1] = (unsigned)carry;
        h[2] = maxlen_all > f.seg ? 1u : 0u;   // a tile outgrew its segment
        h[3] = f.wg_isect16 ? (unsigned)n16_all : (unsigned)carry;   // intersections with the contract's 16-pixel tiles
        h[4] = (unsigned)maxlen_all;
        h[5] = f.stats[5];
        h[6] = 0u;
        h[7] = 0u;
        f.stats[5] = 0u;
#ifdef SAS_DEBUG_BOUNDS
        {   // self-check of the hand-off: what the tail summed from the workgroups' adds on the ticket lines == what they added to one counter
            const unsigned twin = __hip_atomic_load(&f.stats[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)SAS_IN((long long)nvis_all == (long long)twin ? 0 : -1, 1, 130);
            f.stats[6] = 0u;
        }
#endif
    }
    PL_LAP(13);
}

// The tail of a TWO-PASS frame (SAS_FULL_SORT, SAS_DIRECT=0, frames whose segments exceed the budget): offsets, cursors of the
// `big` entries, class sizes; the tiles are put in order by k_scatter's front workgroups.  Its registers are the PROJECTION's
// registers (the kernel's allocation is the maximum over its roles): it reads its two count arrays in bursts of 8 tiles --
// 16 registers of counts -- so that it stays below the geometry role's own 59-64.
DEV void scan_tail_two_pass(const SasFrame *fp, int *lds /* >= 1568 ints */)
{
    const SasFrame &f = *fp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles = f.n_tiles;
    int *s_w = lds;           // [16] cross-wave scratch
    int *s_bins = lds + 16;   // [16 classes (descending)][32 copies]
    const int per = (((tiles + 255) >> 8) + 3) & ~3;
    const int t0 = tid * per;
    const int4 *win4 = reinterpret_cast<const int4 *>(f.tile_count + t0);
    const int4 *big4 = reinterpret_cast<const int4 *>(f.tile_big + t0);
    constexpr int NB = 2;   // 16-byte loads per array and burst
    s_bins[tid] = 0;
    s_bins[256 + tid] = 0;
    // ---- phase 1: totals
    int nvis = 0, n16 = 0;
    for (int i = tid; i < f.n_wg; i += 256) nvis += f.wg_vis[i];
    if (f.wg_isect16)
        for (int i = tid; i < f.n_wg; i += 256) n16 += f.wg_isect16[i];
    int total = 0, maxlen = 0;
#pragma unroll 1
    for (int k0 = 0; k0 < per; k0 += 4 * NB) {
        int4 c[NB], g[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool in = k0 + 4 * j < per;
            c[j] = in ? win4[(k0 >> 2) + j] : make_int4(0, 0, 0, 0);
            g[j] = in ? big4[(k0 >> 2) + j] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int sx = c[j].x + g[j].x, sy = c[j].y + g[j].y, sz = c[j].z + g[j].z, sw = c[j].w + g[j].w;
            total += (sx + sy) + (sz + sw);
            maxlen = max(max(maxlen, max(sx, sy)), max(sz, sw));
        }
    }
    int incl = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        maxlen = max(maxlen, __shfl_xor(maxlen, d));
        nvis += __shfl_xor(nvis, d);
        n16 += __shfl_xor(n16, d);
    }
    if (lane == 63) s_w[wv] = incl;
    if (lane == 0) { s_w[4 + wv] = maxlen; s_w[8 + wv] = nvis; s_w[12 + wv] = n16; }
    __syncthreads();
    int run = incl - total;
    for (int w = 0; w < wv; ++w) run += s_w[w];
    const int carry = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    const int maxlen_all = max(max(s_w[4], s_w[5]), max(s_w[6], s_w[7]));
    const int nvis_all = s_w[8] + s_w[9] + s_w[10] + s_w[11];
    const int n16_all = s_w[12] + s_w[13] + s_w[14] + s_w[15];
    // ---- phase 2: class sizes (two-pass binning: also the offsets and the cursors of the `big` entries)
#pragma unroll 1
    for (int k0 = 0; k0 < per; k0 += 4 * NB) {
        int4 c[NB], g[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool in = k0 + 4 * j < per;
            c[j] = in ? win4[(k0 >> 2) + j] : make_int4(0, 0, 0, 0);
            g[j] = in ? big4[(k0 >> 2) + j] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int k = k0 + 4 * j, t = t0 + k;
            if (k >= per || t >= tiles) continue;
            const int cc[4] = {c[j].x + g[j].x, c[j].y + g[j].y, c[j].z + g[j].z, c[j].w + g[j].w};
            {
                const int cw[4] = {c[j].x, c[j].y, c[j].z, c[j].w};
                const int4 o = make_int4(run, run + cc[0], run + cc[0] + cc[1], run + cc[0] + cc[1] + cc[2]);
                run = o.w + cc[3];
                const int4 cur = make_int4(o.x + cw[0], o.y + cw[1], o.z + cw[2], o.w + cw[3]);
                if (t + 3 < tiles) {
                    *reinterpret_cast<int4 *>(f.tile_offset + t) = o;
                    *reinterpret_cast<int4 *>(f.tile_cursor + t) = cur;
                } else {
                    const int oo[4] = {o.x, o.y, o.z, o.w}, uu[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (t + q < tiles) { f.tile_offset[t + q] = oo[q]; f.tile_cursor[t + q] = uu[q]; }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (t + q < tiles) atomicAdd(&s_bins[(15 - len_class(cc[q])) * 32 + (lane & 31)], 1);
        }
    }
    __syncthreads();   // (also: every thread has read the cross-wave scratch of phase 1)
    if (tid < 16) {    // size of class (15 - tid): its 32 copies
        int sum = 0;
        for (int k = 0; k < 32; ++k) sum += s_bins[32 * tid + k];
        s_w[tid] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        int start = 0;
        for (int k = 0; k < 16; ++k) {   // descending classes: entry k = class 15 - k
            f.class_cursor[k] = start;
            // the full path's sort classes are class ranges, hence contiguous in tile_order
            if (k == 3) f.sort_class[1] = start;   // mid:   1024..4095     (classes 11, 12)
            if (k == 5) f.sort_class[2] = start;   // small: < 1024         (classes <= 10)
            start += s_w[k];
        }
        f.sort_class[0] = 0;             // large: length >= 4096 (classes >= 13)
        f.sort_class[3] = tiles;
        f.sort_class[4] = 0;             // every tile, for the full-path blend
        f.sort_class[5] = tiles;
        f.tile_offset[tiles] = carry;
        unsigned *h = f.stats_host;      // the statistics, straight to pinned host memory
        h[0] = (unsigned)nvis_all;
        h[1] = (unsigned)carry;
        h[2] = (long long)carry > f.cap ? 1u : 0u;   // the compact lists outgrew the key buffer
        h[3] = f.wg_isect16 ? (unsigned)n16_all : (unsigned)carry;   // intersections with the contract's 16-pixel tiles
        h[4] = (unsigned)maxlen_all;
        h[5] = f.stats[5];
        h[6] = 0u;
        h[7] = 0u;
        f.stats[5] = 0u;
#ifdef SAS_DEBUG_BOUNDS
        {   // self-check of the hand-off: what the tail summed from the per-workgroup stores == what the workgroups added atomically
            const unsigned twin = __hip_atomic_load(&f.stats[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)SAS_IN((long long)nvis_all == (long long)twin ? 0 : -1, 1, 130);
            f.stats[6] = 0u;
        }
#endif
    }
}

// Scene loads.  A view pair streams the colour planes once, non-temporally, so that they do not evict the records
// and keys the tile kernels re-read (+1 % frames/s).  Single views are submitted up to four deep and
// their projections overlap: ordinary loads let them share the scene through L2 / the memory-side cache
// (+3 % frames/s at 1 M Gaussians, +11 % at 5 M over non-temporal loads).
template <int NV>
DEV float4 scene_load(const float4 *p)
{
    if constexpr (NV >= 2) return nt_load(p);
    else return *p;
}

// Group pose of this lane's Gaussian.  Poses reach a lane in one of two ways: from the slot's device block (12 per-lane
// loads), or -- small pose blocks, `inline_row(g, k)` -- from the launch's ARGUMENT segment: the scene is stored by group,
// so a wave almost always holds one group; its row is fetched with scalar loads at a uniform index (no upload kernel in
// front of the projection, no address of the argument struct taken: that would make the compiler copy it to scratch).
// Must be called by the in-range lanes of a wave together.  Returns whether the scene has poses at all.
template <typename RowFn>
DEV bool group_pose(unsigned gid, const float *poses, bool poses_inline, RowFn inline_row, float *G)
{
    if (poses_inline) {
        unsigned long long todo = __ballot(true);
        while (todo) {
            const unsigned g = (unsigned)__builtin_amdgcn_readlane((int)gid, __ffsll((long long)todo) - 1);
            const bool mine = gid == g;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const float r = inline_row(g, k);   // uniform: scalar load
                if (mine) G[k] = r;
            }
            todo &= ~__ballot(mine);
        }
        return true;
    }
    if (poses) {
        const float *Gp = poses + 12 * gid;
#pragma unroll
        for (int k = 0; k < 12; ++k) G[k] = Gp[k];
        return true;
    }
    return false;
}

// ---- the projection, round 5: TWO ROLES IN ONE LAUNCH ----------------------------------------------------------------
// Rounds 1-4 ran T1, T2 and the binning in one body: every workgroup streamed its 256 Gaussians' 236 bytes, held the 48
// SH coefficients beside the projected geometry (81-102 registers: four or five waves per SIMD), and then sat through the
// binning's barriers, atomics and scattered key stores with those registers -- the streaming phase reached 4.4 TB/s and
// the launch ended on a 10 us single-workgroup tail (the scan of the tile counts) with the chip idle.  Now the launch's
// workgroups have one of two roles, chosen by blockIdx (uniform):
//   GEOMETRY (one workgroup per 256 Gaussians and VIEW): 48 bytes in, T1, the 32-byte record + info out, the window
//            histogram, count atomics, exact tile culling and key emit of T3; the last of a view's geometry workgroups
//            runs that view's tail (T5).  Latency-bound (barriers, returning atomics), HBM-light.
//   COLOUR   (one workgroup per 256 Gaussians, all views of the launch): mean + the 192 bytes of SH planes in, T2 for each
//            view, 16 bytes per view out.  No LDS, no barrier, no atomic: a pure stream.
// Both kinds are resident together (the block index interleaves them), so the HBM-bound stream runs UNDER the
// latency-bound binning instead of in front of it, and the geometry blocks are weighted towards the front of the grid
// (mix_k eighths of the leading blocks) so that the tail runs while the last colour blocks still stream.  One launch,
// one stream: no cross-queue hop on the blocking path.
struct ProjArgs {
    SasCam cam[2];
    SasFrame f[2];
    int mix_k;                                   // of 8 consecutive leading blocks, this many are geometry blocks
    int pose_inline;                             // 1: the group poses are pose_rows (small blocks: no upload kernel)
    float pose_rows[12 * SAS_PROJ_INLINE_ROWS];
};

// GEOMETRY role: T1 of Gaussians [256 wg, 256 wg + 256) for ONE view, then that view's binning; reached by all 256 threads.
template <typename RowFn>
DEV void geom_role(const SasScene &s, const SasCam &c, const SasFrame &f, unsigned wg, const float *poses, bool poses_inline,
                   RowFn inline_row, int *s_win, int *s_hist, int *s_base, int *s_nvis, int *s_last)
{
    const int64_t i = (int64_t)wg * 256 + threadIdx.x;
    unsigned long long pl_t_ = 0ull;
    const unsigned pl_wg_ = wg;
    (void)pl_wg_;
#ifdef SAS_TUNE_PTIME
    pl_t_ = wall_clock64();
    if (threadIdx.x == 0 && wg < (unsigned)kDbgPMax) g_dbg_plap[16 * wg + 14] = pl_t_;
#endif
    ViewGeom g;
    g.vis = false; g.x0 = g.x1 = g.y0 = g.y1 = 0; g.z = 0.0f;
    if (i < s.n) {
        const float4 a0 = s.g0[i];
        const float4 a1 = s.g1[i];
        const float4 a2 = s.g2[i];
        float m[3] = {a0.x, a0.y, a0.z};
        const float op = a0.w;
        float G[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
        const bool hasG = group_pose(__float_as_uint(a2.w) & 255u, poses, poses_inline, inline_row, G);
        if (hasG) {
            float mg0 = affine3(G[0], G[1], G[2], G[3], m[0], m[1], m[2]);
            float mg1 = affine3(G[4], G[5], G[6], G[7], m[0], m[1], m[2]);
            float mg2 = affine3(G[8], G[9], G[10], G[11], m[0], m[1], m[2]);
            m[0] = mg0; m[1] = mg1; m[2] = mg2;
        }
        const float cx = affine3(c.R[0], c.R[1], c.R[2], c.t[0], m[0], m[1], m[2]);   // argument segment: scalar loads
        const float cy = affine3(c.R[3], c.R[4], c.R[5], c.t[1], m[0], m[1], m[2]);
        const float cz = affine3(c.R[6], c.R[7], c.R[8], c.t[2], m[0], m[1], m[2]);
        bool ok = !(cz < kNear || cz > kFar);
        ok = ok && !(op < kAlphaThr);   // opacity cull moved up: it has no side effect before the det test
        if (ok) {
            // 3-D covariance in the world (group) frame
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
                if (hasG) {
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
                if (hasG) {
                    const float Rg[9] = {G[0], G[1], G[2], G[4], G[5], G[6], G[8], G[9], G[10]};
                    float c2[6];
                    rot_sym3(Rg, cov, c2);
#pragma unroll
                    for (int k = 0; k < 6; ++k) cov[k] = c2[k];
                }
            }
            project_view(c, cov, op, cx, cy, cz, g);
        }
        if (g.vis) {
            if (!(SAS_TUNE_PABL & 8)) {
                f.rec[SAS_RS * i + 0] = make_float4(g.mx, g.my, g.ca, g.cb);
                f.rec[SAS_RS * i + 1] = make_float4(g.ccn, op, g.thr, g.z);
                // radii (parity hook only): full 32 bits each (a camera inside the cloud produces radii beyond 65535
                // pixels); saturated to INT_MAX
                const int irx = (int)fminf(g.rx, 2147483520.0f), iry = (int)fminf(g.ry, 2147483520.0f);
                if (f.keep_info) f.info[i] = make_uint4((unsigned)g.x0 | ((unsigned)g.x1 << 16), (unsigned)g.y0 | ((unsigned)g.y1 << 16), (unsigned)irx, (unsigned)iry);
            }
        } else if (f.keep_info) {
            f.info[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    PL_LAP(0);
    if (!(SAS_TUNE_PABL & 128)) count_tiles(f, c.tw, c.tile_px, g, (unsigned)i, wg, s_win, s_hist, s_base, s_nvis, pl_t_);
    // ---- the last geometry workgroup of the view to get here scans the counts of its frame.
    // Everything the tail reads from other workgroups was written by AGENT-scope atomics (the per-tile counts, the
    // window-miss counter, wg_vis), which are performed at the point all XCDs share; what remains is ordering:
    // every wave waits until its own outstanding stores and atomics have been acknowledged (s_waitcnt vmcnt(0) -- the
    // wait an agent-scope release consists of, without its cache write-back; a workgroup-scope fence compiles to
    // nothing here), the barrier collects the waves, then thread 0 takes the ticket.  A __threadfence() in this place
    // -- a write-back of the XCD's whole L2 by every workgroup while the projection streams 60 MB of records through
    // it -- cost 740 us per launch.
    // The ticket has two levels: same-address atomics serialise at ~90 per us at the memory side, and 3 907
    // workgroups taking one counter cost the projection 23 us; 64 sub-counters (one cache line each) take ~61
    // tickets each, the last taker of each takes one of 64 master tickets.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PL_LAP(7);
    if (threadIdx.x == 0) {
        unsigned *tk = f.tickets;
        const unsigned n_wg = (unsigned)f.n_wg;
        const unsigned sub = wg & 63u, n_sub = (n_wg - sub + 63u) >> 6, subs = min(n_wg, 64u);
        int last = 0;
        if (atomicAdd(&tk[32u * sub], 1u) == n_sub - 1u) {
            tk[32u * sub] = 0u;                                   // nobody else touches it any more in this frame
            last = atomicAdd(&tk[32u * 64u], 1u) == subs - 1u;
            if (last) tk[32u * 64u] = 0u;
        }
        *s_last = last;
    }
    __syncthreads();
    PL_LAP(8);
#ifdef SAS_TUNE_PTIME
    if (threadIdx.x == 0 && wg < (unsigned)kDbgPMax) g_dbg_plap[16 * wg + 15] = wall_clock64();
#endif
    if (!*s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the tail's loads below are not served from a stale cache line
#ifndef SAS_TUNE_NOTAIL
    if (f.seg > 0) scan_tail_single(&f, s_hist, pl_wg_, pl_t_);   // (uniform)
    else scan_tail_two_pass(&f, s_hist);
#endif
    PL_LAP(9);
#ifdef SAS_TUNE_PTIME
    if (threadIdx.x == 0 && wg < (unsigned)kDbgPMax) g_dbg_plap[16 * wg + 15] = wall_clock64();
#endif
}

// COLOUR role: T2 of Gaussians [256 wg, 256 wg + 256) for every view of the launch.  A pure stream: no LDS, no barrier.
// The colour of a Gaussian that the geometry role culls (off screen, degenerate) is computed all the same -- nobody
// reads it; only what is decided by the mean and the opacity alone is decided here too, with the geometry role's own
// expressions (near / far plane, transparent), so that scenes mostly behind the camera do not stream their planes.
// Contract T2 (round 5): a colour is FINITE when it leaves the projection: clamped to +-FLT_MAX (the identity on every finite value; a
// NaN becomes -FLT_MAX).  The compositing loop adds every staged entry to every pixel of its block with weight +0 where the entry is
// skipped -- exact for finite colours, but 0 * Inf = NaN would spread one bad SH coefficient over whole blocks.
DEV float finite_colour(float v) { return fminf(fmaxf(v, -3.402823466e38f), 3.402823466e38f); }
template <int DEG, int NV, typename RowFn>
DEV void color_role(const SasScene &s, const ProjArgs &vs, unsigned wg, const float *poses, bool poses_inline, RowFn inline_row)
{
    const int64_t i = (int64_t)wg * 256 + threadIdx.x;
#ifdef SAS_TUNE_PTIME
    if (threadIdx.x == 0 && wg < (unsigned)kDbgPMax) g_dbg_pcol[2 * wg] = wall_clock64();
#endif
    if (i >= s.n) return;
    const float4 a0 = s.g0[i];
    float m[3] = {a0.x, a0.y, a0.z};
    const float op = a0.w;
    if (s.n_groups > 0) {   // (uniform)
        float G[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
        if (group_pose((unsigned)s.gid8[i], poses, poses_inline, inline_row, G)) {
            float mg0 = affine3(G[0], G[1], G[2], G[3], m[0], m[1], m[2]);
            float mg1 = affine3(G[4], G[5], G[6], G[7], m[0], m[1], m[2]);
            float mg2 = affine3(G[8], G[9], G[10], G[11], m[0], m[1], m[2]);
            m[0] = mg0; m[1] = mg1; m[2] = mg2;
        }
    }
    bool ok[NV], any_ok = false;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const SasCam &c = vs.cam[v];
        const float cz = affine3(c.R[6], c.R[7], c.R[8], c.t[2], m[0], m[1], m[2]);
        ok[v] = !(cz < kNear || cz > kFar) && !(op < kAlphaThr);
        any_ok = any_ok || ok[v];
    }
    if (!any_ok) return;
    if constexpr (DEG == 3) {
        // two halves of six planes each (sh3_first / sh3_second): 24 coefficients in registers at a time
        float a[24];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const float4 q = scene_load<NV>(s.col + (int64_t)p * s.n_pad + i);
            a[4 * p] = q.x; a[4 * p + 1] = q.y; a[4 * p + 2] = q.z; a[4 * p + 3] = q.w;
        }
        ShHalf S[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const SasCam &c = vs.cam[v];
            sh3_first(a, m[0] - c.campos[0], m[1] - c.campos[1], m[2] - c.campos[2], S[v]);
            // (pin the half-way state HERE: hipcc otherwise sinks the first half's arithmetic below the second half's loads -- its
            // results are first used there -- and all 48 coefficients are in registers at once after all)
            asm volatile("" : "+v"(S[v].x), "+v"(S[v].y), "+v"(S[v].z), "+v"(S[v].r[0]), "+v"(S[v].r[1]), "+v"(S[v].r[2]));
        }
        __builtin_amdgcn_sched_barrier(0);   // (the second half's loads stay behind the first half's arithmetic: that is the point)
        float b[24];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const float4 q = scene_load<NV>(s.col + (int64_t)(p + 6) * s.n_pad + i);
            b[4 * p] = q.x; b[4 * p + 1] = q.y; b[4 * p + 2] = q.z; b[4 * p + 3] = q.w;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float rgb[3];
            sh3_second(b, S[v], rgb);
            if (ok[v] && !(SAS_TUNE_PABL & 8)) vs.f[v].col[SAS_CS * i] = make_float4(finite_colour(rgb[0]), finite_colour(rgb[1]), finite_colour(rgb[2]), 0.0f);
        }
    } else {
        constexpr int KF = DEG >= 0 ? 3 * (DEG + 1) * (DEG + 1) : 4;
        constexpr int PL = (KF + 3) / 4;
        float sh[PL * 4];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            const float4 q = scene_load<NV>(s.col + (int64_t)p * s.n_pad + i);
            sh[4 * p] = q.x; sh[4 * p + 1] = q.y; sh[4 * p + 2] = q.z; sh[4 * p + 3] = q.w;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v > 0) __builtin_amdgcn_sched_barrier(0);   // (one view after the other: interleaving the evaluations costs registers)
            if (!ok[v]) continue;
            const SasCam &c = vs.cam[v];
            float rgb[3];
            if constexpr (DEG >= 0) {
                sh_to_color<DEG>(sh, m[0] - c.campos[0], m[1] - c.campos[1], m[2] - c.campos[2], rgb);
            } else {
                rgb[0] = sh[0]; rgb[1] = sh[1]; rgb[2] = sh[2];
            }
            if (!(SAS_TUNE_PABL & 8)) vs.f[v].col[SAS_CS * i] = make_float4(finite_colour(rgb[0]), finite_colour(rgb[1]), finite_colour(rgb[2]), 0.0f);
        }
    }
#ifdef SAS_TUNE_PTIME
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (instrumented build: the stamp is taken when the stores have been acknowledged)
    if (threadIdx.x == 0 && wg < (unsigned)kDbgPMax) g_dbg_pcol[2 * wg + 1] = wall_clock64();
#endif
}

// Block -> role.  n_geo geometry blocks and n_col colour blocks share the grid; of every 8 leading blocks mix_k are
// geometry blocks until those are used up: geo_before(b) = min(n_geo, ceil(b * mix_k / 8)) geometry blocks lie in front
// of block b.  (mix_k / 8 >= n_geo / (n_geo + n_col): the launchers see to it -- every geometry block gets a place.)
struct Role {
    bool geom;
    unsigned idx;   // index among the blocks of the role
};
// Geometry blocks do not take their chunks of 256 Gaussians in storage order: a chunk's binning work follows its
// Gaussians' footprints, the scene is stored along a space-filling curve, so the chunks near the camera -- several times
// the tiles per Gaussian -- are neighbours in storage order, and dispatched in that order they would start together and
// late and end the launch on a long run of stragglers (tools/proj_time.py: 90 % of the geometry done at 56 us, the last
// workgroup at 70).  Dispatch index -> chunk is a transpose over kSpreadRows rows: consecutive blocks are n_wg / 16
// chunks apart, any run of heavy chunks is spread over the whole launch.  (Blocks whose chunk lies past the end leave.)
constexpr unsigned kSpreadRows = 16;
__host__ DEV unsigned spread_cols(unsigned n_wg) { return (n_wg + kSpreadRows - 1u) / kSpreadRows; }
__host__ DEV unsigned spread_chunk(unsigned gi, unsigned n_wg) { return (gi % kSpreadRows) * spread_cols(n_wg) + gi / kSpreadRows; }
__host__ DEV Role block_role(unsigned b, unsigned n_geo, unsigned mix_k)
{
    const unsigned c0 = (b * mix_k + 7u) >> 3, c1 = ((b + 1u) * mix_k + 7u) >> 3;
    const unsigned g0 = c0 < n_geo ? c0 : n_geo, g1 = c1 < n_geo ? c1 : n_geo;
    Role r;
    r.geom = g1 > g0;
    r.idx = r.geom ? g0 : b - g0;
    return r;
}

#ifdef SAS_TUNE_POCC   // A/B builds: waves per SIMD forced for the projection
#define SAS_PROJECT_ATTRS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SAS_TUNE_POCC, SAS_TUNE_POCC)))
#else
#define SAS_PROJECT_ATTRS __launch_bounds__(256)
#endif
// grid.x = (NV + 1) * n_wg: NV * n_wg geometry blocks (view = index % NV: the views' blocks of one chunk of Gaussians are
// neighbours in dispatch order and share its 48 bytes through the cache) and n_wg colour blocks
template <int DEG, int NV>
__global__ SAS_PROJECT_ATTRS void k_project(SasScene s, ProjArgs vs)
{
    __shared__ __attribute__((aligned(16))) int s_win[16];
    __shared__ __attribute__((aligned(16))) int s_bins2[2 * kHistBins];   // one block: the tail uses all 16 KiB of it
    int *const s_hist = s_bins2;
    int *const s_base = s_bins2 + kHistBins;   // single-pass binning: start of the workgroup's run inside each window tile's segment
    __shared__ int s_nvis[2];
    __shared__ int s_last;
    const unsigned n_wg = (unsigned)vs.f[0].n_wg;
    const Role r = block_role(blockIdx.x, (unsigned)NV * kSpreadRows * spread_cols(n_wg), (unsigned)vs.mix_k);
    auto row = [&](unsigned g, int k) { return vs.pose_rows[12u * g + (unsigned)k]; };
    if (!r.geom) {
        if (!(SAS_TUNE_PABL & 32) && r.idx < n_wg) color_role<DEG, NV>(s, vs, r.idx, vs.f[0].group_Rt, vs.pose_inline != 0, row);
        return;
    }
    if (SAS_TUNE_PABL & 64) return;
    const unsigned wg = spread_chunk(NV == 1 ? r.idx : r.idx >> 1, n_wg);
    if (wg >= n_wg) return;
    // (two inlined copies for a pair, chosen by a uniform branch: the view's camera and frame are then read from the
    // argument segment at constant offsets; indexing them by a run-time view would copy them to registers or scratch)
    if (NV == 1 || (r.idx & 1u) == 0u)
        geom_role(s, vs.cam[0], vs.f[0], wg, vs.f[0].group_Rt, vs.pose_inline != 0, row, s_win, s_hist, s_base, s_nvis, &s_last);
    else
        geom_role(s, vs.cam[1], vs.f[1], wg, vs.f[0].group_Rt, vs.pose_inline != 0, row, s_win, s_hist, s_base, s_nvis, &s_last);
}

// all views of a group in one launch: blockIdx.y = view, one pass over the (small) scene per view, both roles per view
template <int DEG>
__global__ __launch_bounds__(256) void k_project_multi(SasScene s, SasMulti mf)
{
    __shared__ __attribute__((aligned(16))) int s_win[16];
    __shared__ __attribute__((aligned(16))) int s_bins2[2 * kHistBins];
    int *const s_hist = s_bins2;
    int *const s_base = s_bins2 + kHistBins;
    __shared__ int s_nvis[2];
    __shared__ int s_last;
    ProjArgs vs;
    vs.cam[0] = vs.cam[1] = mf.P[blockIdx.y].cam;
    vs.f[0] = vs.f[1] = mf.f[blockIdx.y];
    const unsigned off = (unsigned)mf.pose_off[blockIdx.y];
    auto row = [&](unsigned g, int k) { return mf.pose_rows[off + 12u * g + (unsigned)k]; };
    const unsigned n_wg = (unsigned)vs.f[0].n_wg;
    const Role r = block_role(blockIdx.x, kSpreadRows * spread_cols(n_wg), (unsigned)mf.mix_k);
    if (!r.geom) {
        if (r.idx < n_wg) color_role<DEG, 1>(s, vs, r.idx, mf.f[blockIdx.y].group_Rt, mf.pose_inline != 0, row);
        return;
    }
    const unsigned wg = spread_chunk(r.idx, n_wg);
    if (wg >= n_wg) return;
    geom_role(s, vs.cam[0], vs.f[0], wg, mf.f[blockIdx.y].group_Rt, mf.pose_inline != 0, row, s_win, s_hist, s_base, s_nvis, &s_last);
}

// ---- small kernels around a frame ---------------------------------------------------------------------
// Group poses of the views of a launch (SasPoseUpload): from the ARGUMENT segment, which the command processor
// hands to the kernel without a PCIe round trip, or -- more rows than fit there -- read from pinned host memory by
// the kernel itself.  grid.x = view.
struct SasPoseArgs {
    SasPoseUpload u;
    int inline_off[SAS_MAX_GROUP];
    int use_inline;
    float rows[12 * SAS_POSE_INLINE_ROWS];
};
__global__ __launch_bounds__(256) void k_pose_upload(SasPoseArgs a)
{
    const int v = blockIdx.x;
    const float *src = a.use_inline ? a.rows + a.inline_off[v] : a.u.src_host[v];
    float *dst = a.u.dst[v];
    for (int k = threadIdx.x; k < a.u.floats[v]; k += 256) dst[k] = src[k];
}
// grid = (blocks, views): every block copies its share of its view's uint8 frame to pinned host memory, 16 bytes per
// lane where source, destination and size allow.
__global__ __launch_bounds__(256) void k_host_copy(SasHostCopy h)
{
    const uint8_t *src = h.src[blockIdx.y];
    uint8_t *dst = h.dst[blockIdx.y];
    if (!src || !dst) return;
    const size_t n = h.bytes, stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((((size_t)src | (size_t)dst | n) & 15) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (size_t i = i0; i < n / 16; i += stride) d4[i] = s4[i];
    } else {
        for (size_t i = i0; i < n; i += stride) dst[i] = src[i];
    }
}

// ---- k_scatter: T3 emit ---------------------------------------------------------------------------
// Same window as k_project: count in LDS, reserve one contiguous run per touched tile with a
// single returning global atomic, then rank inside the run with LDS atomics.  Key = depth bits << 32
// | storage slot; the rare runs of identical depth are ordered by the caller's index (perm[slot])
// when a tile is sorted, exactly as the reference's stable sort orders them.
// ---- k_scatter: T3 emit + tile order --------------------------------------------------------------------------
// Binning workgroups (one per projection workgroup, same 256 Gaussians, same window): the projection has already
// reserved this workgroup's run in every tile its window touches (wg_base), so a key's position is
// tile_offset + wg_base + its rank among the workgroup's keys for that tile (LDS atomics): no counting pass and no
// global atomic.  Gaussians outside the window scheme (large rectangles) take one returning atomic per intersection
// on tile_cursor, which starts behind the tile's window runs.  Key = depth bits << 32 | storage slot; the rare runs
// of identical depth are ordered by the caller's index (perm[slot]) when a tile is sorted, exactly as the
// reference's stable sort orders them.
// The FRONT workgroups of the launch (256 tiles each) put the tiles into tile_order by list-length class, longest
// first: class sizes in LDS, one returning atomic per (workgroup, class) on the class cursors the projection's tail
// has set to the classes' first positions.
DEV int order_workgroups(int tiles) { return (tiles + 255) >> 8; }

DEV void order_body(const SasFrame &f, int wg)
{
    __shared__ int s_cls[16 * 32];   // [class (descending)][copy]
    __shared__ int s_cbase[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = wg * 256 + tid;
    s_cls[tid] = 0;
    s_cls[256 + tid] = 0;
    int cls = 0;
    const bool in = t < f.n_tiles;
    if (in) cls = 15 - len_class(f.tile_count[t] + f.tile_big[t]);
    __syncthreads();
    int rank = 0;
    if (in) rank = atomicAdd(&s_cls[32 * cls + (lane & 31)], 1);
    __syncthreads();
    if (tid < 16) {   // the class's 32 copies become starts inside the workgroup's run of that class
        int sum = 0;
#pragma unroll 4
        for (int k = 0; k < 32; ++k) { const int v = s_cls[32 * tid + k]; s_cls[32 * tid + k] = sum; sum += v; }
        s_cbase[tid] = sum ? atomicAdd(&f.class_cursor[tid], sum) : 0;
    }
    __syncthreads();
    if (in) {
        const int pos = s_cbase[cls] + s_cls[32 * cls + (lane & 31)] + rank;
        if (SAS_IN(pos, f.n_tiles, 104)) f.tile_order[pos] = t;
    }
}

DEV void scatter_body(const SasScene &s, int tw, const SasFrame &f)
{
    const int n_order = order_workgroups(f.n_tiles);
    if ((int)blockIdx.x < n_order) {   // uniform per workgroup
        order_body(f, (int)blockIdx.x);
        return;
    }
    const unsigned wg = blockIdx.x - (unsigned)n_order;
    __shared__ int s_win[4];
    __shared__ int s_hist[kHistBins];
    __shared__ int s_base[kHistBins];
    const int64_t i = (int64_t)wg * 256 + threadIdx.x;
    uint4 inf = make_uint4(0u, 0u, 0u, 0u);
    if (i < s.n) inf = f.info[i];
    const int x0 = inf.x & 0xffff, x1 = inf.x >> 16, y0 = inf.y & 0xffff, y1 = inf.y >> 16;
    const bool vis = x1 > x0 && y1 > y0;
    const unsigned zbits = vis ? __float_as_uint(f.rec[SAS_RS * i + 1].w) : 0u;   // (info holds the rectangle and the radii; the depth is the record's)
    const unsigned long long key = ((unsigned long long)zbits << 32) | (unsigned long long)(unsigned)i;
    const int rect_area = vis ? (x1 - x0) * (y1 - y0) : 0;
    const bool in_win = rect_area > 0 && rect_area <= kWinRect;
    const Window w = wg_window(in_win, x0, x1, y0, y1, s_win);
    if (w.fits) {
        const int *wb = f.wg_base + (size_t)wg * SAS_WIN_BINS;
        {   // bin b = row * ww + col of the window, walked in steps of 256 bins without a division per step
            const unsigned ww = (unsigned)w.ww, tid = threadIdx.x;
            unsigned row = tid / ww, col = tid - row * ww;
            const unsigned drow = 256u / ww, dcol = 256u - drow * ww;
#pragma clang loop unroll(disable)
            for (unsigned b = tid; b < (unsigned)w.area; b += 256u) {
                const int tile = (w.Y0 + (int)row) * tw + w.X0 + (int)col;
                s_base[b] = SAS_IN(tile, f.n_tiles, 112) ? f.tile_offset[tile] + wb[b] : 0;
                s_hist[b] = 0;
                row += drow;
                col += dcol;
                if (col >= ww) { col -= ww; ++row; }
            }
        }
        __syncthreads();
        if (in_win)
#pragma unroll 1
            for (int ty = y0; ty < y1; ++ty)
#pragma unroll 1
                for (int tx = x0; tx < x1; ++tx) {
                    const int b = (ty - w.Y0) * w.ww + (tx - w.X0);
                    if (!SAS_IN(b, kHistBins, 113)) continue;
                    const long long pos = (long long)s_base[b] + atomicAdd(&s_hist[b], 1);
                    // pos >= cap is the documented overflow path (the frame is rendered again); a position
                    // beyond the tile's own segment would be a bug
                    if (pos < f.cap && SAS_IN(pos, (long long)f.tile_offset[ty * tw + tx + 1], 114)) f.keys[pos] = key;
                }
    }
    const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
    for_each_tile(vis && !(w.fits && in_win), x0, x1, y0, y1, tw, klo, khi, [&](int tile, unsigned lo, unsigned hi) {
        if (!SAS_IN(tile, f.n_tiles, 115)) return;
        const int pos = atomicAdd(&f.tile_cursor[tile], 1);
        if ((long long)pos < f.cap && SAS_IN(pos, f.tile_offset[tile + 1], 116)) f.keys[pos] = ((unsigned long long)hi << 32) | lo;
    });
}

// 32 VGPRs: in steady state the register file is what the frames in flight compete for (five tile-kernel waves of 96
// VGPRs leave 32 per SIMD lane): a scatter wave that fits into the remainder runs BESIDE a full complement of tile waves
// instead of displacing one (wave slots 5 + 1 of 8, LDS 5 x 27 + 18.5 of 160 KB).  Left to itself the compiler takes 46:
// the kernel's LDS caps it at eight waves per SIMD, below which registers look free.
#ifndef SAS_TUNE_SCATTER_VGPR
#define SAS_TUNE_SCATTER_VGPR 32
#endif
#define SAS_SCATTER_ATTRS __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(SAS_TUNE_SCATTER_VGPR)))
__global__ SAS_SCATTER_ATTRS void k_scatter(SasScene s, int tw, SasFrame f) { scatter_body(s, tw, f); }
__global__ SAS_SCATTER_ATTRS void k_scatter_multi(SasScene s, int tw, SasMulti mf) { scatter_body(s, tw, mf.f[blockIdx.y]); }

}  // namespace

SAS_BOUNDS_ACCESSOR(sas_debug_bounds_kernels)

// ---- launchers -------------------------------------------------------------------------------------
void sas_launch_relayout(hipStream_t st, int64_t n, int64_t n_pad, const int *perm, const float *means, const float *quats,
                         const float *scales, const float *cov6, const float *opac, const float *colors,
                         int coeff_floats, int planes, const uint8_t *gid, float4 *g0, float4 *g1, float4 *g2,
                         float4 *col, uint8_t *gid8)
{
    if (n <= 0) return;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_relayout, dim3(grid), dim3(256), 0, st, n, n_pad, perm, means, quats, scales, cov6, opac, colors,
                       coeff_floats, planes, gid, g0, g1, g2, col, gid8);
}

static unsigned sas_spread_blocks(unsigned n_wg) { return 16u * ((n_wg + 15u) / 16u); }   // kSpreadRows * spread_cols
// geometry blocks per 8 leading blocks (block_role): enough for every geometry block to get a place, weighted so that the
// geometry -- whose last workgroup runs the serial tail -- is dispatched ahead of the last colour blocks.  SAS_PROJ_MIX=k overrides.
static int project_mix(int nv, unsigned n_wg)
{
    static const int env = [] { const char *e = getenv("SAS_PROJ_MIX"); return e ? atoi(e) : 0; }();
    const unsigned n_geo = (unsigned)nv * sas_spread_blocks(n_wg), total = n_geo + n_wg;
    const int kmin = (int)((8u * n_geo + total - 1u) / total);
    const int k = env > 0 ? env : (nv == 1 ? 5 : 7);
    return k < kmin ? kmin : (k > 8 ? 8 : k);
}

// Test hook (CPU, no GPU needed): what block `b` of a projection launch over n_wg chunks of 256 Gaussians and nv views does --
// through the very functions the kernel and its launcher use.  out = {grid size, mix_k, is geometry, chunk, view}; chunk >= n_wg:
// the block leaves at once (padding of the transposed dispatch).  tests/test_host_logic.py holds the mapping to "every
// (chunk, view) has exactly one geometry block, every chunk exactly one colour block".
extern "C" int sas_debug_projection_block(unsigned b, unsigned n_wg, int nv, unsigned *out)
{
    if (!out || n_wg == 0 || nv < 1 || nv > 2) return -1;
    const unsigned n_geo = (unsigned)nv * sas_spread_blocks(n_wg), grid = n_geo + n_wg;
    const unsigned mix = (unsigned)project_mix(nv, n_wg);
    out[0] = grid;
    out[1] = mix;
    if (b >= grid) return -2;
    const Role r = block_role(b, (unsigned)nv * kSpreadRows * spread_cols(n_wg), mix);
    out[2] = r.geom ? 1u : 0u;
    out[3] = r.geom ? spread_chunk(nv == 1 ? r.idx : r.idx >> 1, n_wg) : r.idx;
    out[4] = (r.geom && nv == 2) ? (r.idx & 1u) : 0u;
    return 0;
}

template <int NV>
static void launch_project(hipStream_t st, const SasScene &s, const ProjArgs &vs)
{
    const unsigned n_wg = (unsigned)vs.f[0].n_wg;   // >= 1: an empty scene still takes the tail
    const unsigned grid = (unsigned)NV * sas_spread_blocks(n_wg) + n_wg;   // geometry blocks (per view, padded to the spread's rows) + colour blocks
    switch (s.sh_degree) {
        case 0: hipLaunchKernelGGL((k_project<0, NV>), dim3(grid), dim3(256), 0, st, s, vs); break;
        case 1: hipLaunchKernelGGL((k_project<1, NV>), dim3(grid), dim3(256), 0, st, s, vs); break;
        case 2: hipLaunchKernelGGL((k_project<2, NV>), dim3(grid), dim3(256), 0, st, s, vs); break;
        case 3: hipLaunchKernelGGL((k_project<3, NV>), dim3(grid), dim3(256), 0, st, s, vs); break;
        default: hipLaunchKernelGGL((k_project<-1, NV>), dim3(grid), dim3(256), 0, st, s, vs); break;
    }
}

static void inline_poses(ProjArgs &vs, const SasScene &s)
{
    vs.pose_inline = sas_poses_inline(s.n_groups, 1, false) && vs.f[0].group_host ? 1 : 0;
    if (vs.pose_inline) memcpy(vs.pose_rows, vs.f[0].group_host, sizeof(float) * 12 * (size_t)s.n_groups);
}

void sas_launch_project(hipStream_t st, const SasScene &s, const SasParams &P, const SasFrame &f)
{
    ProjArgs vs;
    vs.cam[0] = vs.cam[1] = P.cam;
    vs.f[0] = vs.f[1] = f;
    vs.mix_k = project_mix(1, (unsigned)f.n_wg);
    inline_poses(vs, s);
    launch_project<1>(st, s, vs);
}

void sas_launch_project2(hipStream_t st, const SasScene &s, const SasParams &P0, const SasFrame &f0, const SasParams &P1,
                         const SasFrame &f1)
{
    ProjArgs vs;
    vs.cam[0] = P0.cam; vs.cam[1] = P1.cam;
    vs.f[0] = f0; vs.f[1] = f1;
    vs.mix_k = project_mix(2, (unsigned)f0.n_wg);
    inline_poses(vs, s);
    launch_project<2>(st, s, vs);
}

void sas_launch_project_multi(hipStream_t st, const SasScene &s, const SasMulti &mf_in)
{
    static_assert(sizeof(SasMulti) <= 4096, "kernel argument segment");
    SasMulti mf = mf_in;
    mf.pose_inline = sas_poses_inline(s.n_groups, mf.nv, true) ? 1 : 0;
    for (int k = 0; k < mf.nv && mf.pose_inline; ++k) {
        if (!mf.f[k].group_host) { mf.pose_inline = 0; break; }
        mf.pose_off[k] = 12 * s.n_groups * k;
        memcpy(mf.pose_rows + mf.pose_off[k], mf.f[k].group_host, sizeof(float) * 12 * (size_t)s.n_groups);
    }
    mf.mix_k = project_mix(1, (unsigned)mf.f[0].n_wg);
    const dim3 grid(sas_spread_blocks((unsigned)mf.f[0].n_wg) + (unsigned)mf.f[0].n_wg, (unsigned)mf.nv);   // geometry + colour blocks per view
    switch (s.sh_degree) {
        case 0: hipLaunchKernelGGL((k_project_multi<0>), grid, dim3(256), 0, st, s, mf); break;
        case 1: hipLaunchKernelGGL((k_project_multi<1>), grid, dim3(256), 0, st, s, mf); break;
        case 2: hipLaunchKernelGGL((k_project_multi<2>), grid, dim3(256), 0, st, s, mf); break;
        case 3: hipLaunchKernelGGL((k_project_multi<3>), grid, dim3(256), 0, st, s, mf); break;
        default: hipLaunchKernelGGL((k_project_multi<-1>), grid, dim3(256), 0, st, s, mf); break;
    }
}

void sas_launch_scatter_multi(hipStream_t st, const SasScene &s, int tw, const SasMulti &mf)
{
    const unsigned grid = (unsigned)((mf.f[0].n_tiles + 255) / 256) + (unsigned)((s.n + 255) / 256);   // tile-order workgroups first
    hipLaunchKernelGGL(k_scatter_multi, dim3(grid, (unsigned)mf.nv), dim3(256), 0, st, s, tw, mf);
}

void sas_launch_pose_upload(hipStream_t st, const SasPoseUpload &u)
{
    static_assert(sizeof(SasPoseArgs) <= 4096, "kernel argument segment");
    SasPoseArgs a{};
    a.u = u;
    int total = 0;
    for (int v = 0; v < u.nv; ++v) total += u.floats[v];
    a.use_inline = total <= 12 * SAS_POSE_INLINE_ROWS;
    if (a.use_inline) {
        int off = 0;
        for (int v = 0; v < u.nv; ++v) {
            a.inline_off[v] = off;
            for (int k = 0; k < u.floats[v]; ++k) a.rows[off + k] = u.src_host[v][k];
            off += u.floats[v];
        }
    }
    hipLaunchKernelGGL(k_pose_upload, dim3((unsigned)u.nv), dim3(256), 0, st, a);
}

void sas_launch_host_copy(hipStream_t st, const SasHostCopy &h)
{
    if (h.bytes == 0 || h.nv <= 0) return;
    const size_t units = (h.bytes + 16 * 256 - 1) / (16 * 256);
    const unsigned blocks = (unsigned)(units < 128 ? (units ? units : 1) : 128);
    hipLaunchKernelGGL(k_host_copy, dim3(blocks, (unsigned)h.nv), dim3(256), 0, st, h);
}

void sas_launch_scatter(hipStream_t st, const SasScene &s, int tw, const SasFrame &f)
{
    const unsigned grid = (unsigned)((f.n_tiles + 255) / 256) + (unsigned)((s.n + 255) / 256);   // tile-order workgroups first
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, st, s, tw, f);
}

