/*
 * sas_oracle.c -- CPU ORACLE for the sim_a_splat render-image hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it.  The product path
 * (sim_a_splat_amd/) never links or imports anything under oracle/.
 *
 * What it restates (citations relative to /root/reference):
 *   T0  nerfstudio SplatfactoModel.get_outputs        called at sim_a_splat/ns_utils/nerfstudio_utils.py:166-172
 *   T1  gsplat 1.5.2 fully_fused_projection (fwd)     (pixi.lock:1805-1807; source not in the tree)
 *   T2  gsplat 1.5.2 spherical_harmonics (fwd)        + the "+0.5, clamp_min 0" of gsplat.rasterization
 *   T3  gsplat 1.5.2 isect_tiles                      16-pixel tiles
 *   T4  radix sort of (tile | depth bits) keys        ties keep ascending Gaussian index (stable sort)
 *   T5  isect_offset_encode
 *   T6  rasterize_to_pixels (fwd) + "RGB+ED" normalisation
 *   T7  uint8 frame of the viser door                 sim_a_splat/env/splat/splat_env_wrapper.py:148-157
 *   group poses (Door B): sim_a_splat/splat/splat_handler.py:283-288 sets one SE3 per splat group
 *
 * PARITY STATUS: gsplat / nerfstudio / viser are third-party packages that are NOT present
 * under /root/reference and cannot be installed here; the reference has no test, golden image
 * or fixture for this path.  For rows T0-T7 this oracle is therefore
 * **parity unpinned**: it restates the published algorithm of the pinned versions.  The in-tree
 * rows (a3 SH2RGB, a5 compute_cov, a8/a9 pose algebra) are pinned by golden vectors generated
 * from the importable reference module (oracle/make_golden.py, tests/golden/).
 *
 * ARITHMETIC CONTRACT.  Every float operation below is an IEEE-754 binary32 operation with
 * round-to-nearest-even; a*b+c is fused ONLY where fmaf() is written (build with
 * -ffp-contract=off).  exp and log are the explicit polynomials oc_expf/oc_logf, not libm,
 * so that an independent implementation that performs the same sequence of operations
 * (the HIP kernels) reproduces every threshold decision (alpha < 1/255, T <= 1e-4, radius
 * ceil) bit for bit.  DESIGN.md "Arithmetic contract" is the normative text.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define OC_TILE 16
#define OC_TILE_CENTRE 8.0f   /* the tile-local frame of T6 (u, v, x, y) has its origin at the tile's centre */
#define OC_NEAR 0.01f
#define OC_FAR 1e10f
#define OC_EPS2D 0.3f
#define OC_ALPHA_THRESHOLD (1.0f / 255.0f)
#define OC_MAX_ALPHA 0.999f
#define OC_T_STOP 1e-4f

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int sas_oracle_num_threads(void);
static inline int oc_thread_id(void)
{
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}

/* ---- contract transcendental functions ------------------------------------------------- */

/* 2^f on [-0.5,0.5], degree-5 near-minimax, max rel. error 2.3e-7 after f32 rounding. */
#define OC_E0 1.0000001192092896f
#define OC_E1 0.6931471824645996f
#define OC_E2 0.2402210682630539f
#define OC_E3 0.05550327152013779f
#define OC_E4 0.009676037356257439f
#define OC_E5 0.0013400432653725147f
#define OC_LOG2E 1.4426950408889634f

float sas_oracle_expf(float x)
{
    /* argument clamped so that 2^n stays a normal number; n = round-half-even(x log2 e) is taken
     * from the low mantissa bits of x log2 e + 1.5 * 2^23 (one fused multiply-add, the hardware
     * v_fma_f32) and added into the exponent field of the polynomial's value */
    const float magic = 12582912.0f;
    x = fmaxf(x, -86.0f);
    x = fminf(x, 86.0f);
    float tm = fmaf(x, OC_LOG2E, magic);
    float n = tm - magic;
    float f = fmaf(x, OC_LOG2E, -n);
    float p = OC_E5;
    p = fmaf(p, f, OC_E4);
    p = fmaf(p, f, OC_E3);
    p = fmaf(p, f, OC_E2);
    p = fmaf(p, f, OC_E1);
    p = fmaf(p, f, OC_E0);
    return u2f(f2u(p) + (f2u(tm) << 23));
}

/* natural log for positive normal x: x = 2^e * m, m in [sqrt(1/2), sqrt(2)),
 * ln m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^9. */
float sas_oracle_logf(float x)
{
    uint32_t u = f2u(x);
    int e = (int)(u >> 23) - 127;
    float m = u2f((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = 0.1111111111f;          /* 1/9 */
    p = fmaf(p, s2, 0.1428571429f);   /* 1/7 */
    p = fmaf(p, s2, 0.2f);            /* 1/5 */
    p = fmaf(p, s2, 0.3333333333f);   /* 1/3 */
    p = fmaf(p, s2, 1.0f);
    float lnm = (2.0f * s) * p;
    return fmaf((float)e, 0.6931471805599453f, lnm);
}

/* ---- small fixed-order linear algebra --------------------------------------------------- */

/* r0*v0 + r1*v1 + r2*v2 + t, innermost term first */
static inline float affine3(const float *r, float t, const float *v)
{
    return fmaf(r[0], v[0], fmaf(r[1], v[1], fmaf(r[2], v[2], t)));
}
/* a0*b0 + a1*b1 + a2*b2 */
static inline float dot3(float a0, float a1, float a2, float b0, float b1, float b2)
{
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

/* symmetric 3x3 as s[6] = xx xy xz yy yz zz;  out = R s R^T  (R row-major 3x3) */
static void rot_sym3(const float R[9], const float s[6], float out[6])
{
    const float S[3][3] = {{s[0], s[1], s[2]}, {s[1], s[3], s[4]}, {s[2], s[4], s[5]}};
    float T[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[i][j] = dot3(R[3 * i + 0], R[3 * i + 1], R[3 * i + 2], S[0][j], S[1][j], S[2][j]);
    out[0] = dot3(T[0][0], T[0][1], T[0][2], R[0], R[1], R[2]);
    out[1] = dot3(T[0][0], T[0][1], T[0][2], R[3], R[4], R[5]);
    out[2] = dot3(T[0][0], T[0][1], T[0][2], R[6], R[7], R[8]);
    out[3] = dot3(T[1][0], T[1][1], T[1][2], R[3], R[4], R[5]);
    out[4] = dot3(T[1][0], T[1][1], T[1][2], R[6], R[7], R[8]);
    out[5] = dot3(T[2][0], T[2][1], T[2][2], R[6], R[7], R[8]);
}

/* T1: quaternion (wxyz, any norm) -> rotation, gsplat quat_to_rotmat */
static void quat_to_rotmat(const float q[4], float R[9])
{
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float n2 = fmaf(z, z, fmaf(y, y, fmaf(x, x, w * w)));
    float inv = 1.0f / sqrtf(n2);
    w *= inv; x *= inv; y *= inv; z *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = fmaf(-2.0f, y2 + z2, 1.0f); R[1] = 2.0f * (xy - wz);           R[2] = 2.0f * (xz + wy);
    R[3] = 2.0f * (xy + wz);           R[4] = fmaf(-2.0f, x2 + z2, 1.0f); R[5] = 2.0f * (yz - wx);
    R[6] = 2.0f * (xz - wy);           R[7] = 2.0f * (yz + wx);           R[8] = fmaf(-2.0f, x2 + y2, 1.0f);
}

/* ---- T2: real SH, degree <= 3, coefficient layout [16][3] (k-major) ---------------------- */
static void sh_to_color(int degree, const float *sh, float dx, float dy, float dz, float rgb[3])
{
    float inorm = 1.0f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    float x = dx * inorm, y = dy * inorm, z = dz * inorm;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z;
    float fC1 = fmaf(x, x, -(y * y));
    float fS1 = 2.0f * x * y;
    float pSH6 = fmaf(0.9461746957575601f, z2, -0.3153915652525201f);
    float pSH7 = fTmp0B * x;
    float pSH5 = fTmp0B * y;
    float pSH8 = 0.5462742152960395f * fC1;
    float pSH4 = 0.5462742152960395f * fS1;
    float fTmp0C = fmaf(-2.285228997322329f, z2, 0.4570457994644658f);
    float fTmp1B = 1.445305721320277f * z;
    float fC2 = fmaf(x, fC1, -(y * fS1));
    float fS2 = fmaf(x, fS1, y * fC1);
    float pSH12 = z * fmaf(1.865881662950577f, z2, -1.119528997770346f);
    float pSH13 = fTmp0C * x;
    float pSH11 = fTmp0C * y;
    float pSH14 = fTmp1B * fC1;
    float pSH10 = fTmp1B * fS1;
    float pSH15 = -0.5900435899266435f * fC2;
    float pSH9 = -0.5900435899266435f * fS2;
    for (int c = 0; c < 3; ++c) {
        float r = 0.2820947917738781f * sh[0 * 3 + c];
        if (degree >= 1) {
            float t = fmaf(-x, sh[3 * 3 + c], fmaf(z, sh[2 * 3 + c], (-y) * sh[1 * 3 + c]));
            r = fmaf(0.48860251190292f, t, r);
        }
        if (degree >= 2) {
            r = fmaf(pSH4, sh[4 * 3 + c], r);
            r = fmaf(pSH5, sh[5 * 3 + c], r);
            r = fmaf(pSH6, sh[6 * 3 + c], r);
            r = fmaf(pSH7, sh[7 * 3 + c], r);
            r = fmaf(pSH8, sh[8 * 3 + c], r);
        }
        if (degree >= 3) {
            r = fmaf(pSH9, sh[9 * 3 + c], r);
            r = fmaf(pSH10, sh[10 * 3 + c], r);
            r = fmaf(pSH11, sh[11 * 3 + c], r);
            r = fmaf(pSH12, sh[12 * 3 + c], r);
            r = fmaf(pSH13, sh[13 * 3 + c], r);
            r = fmaf(pSH14, sh[14 * 3 + c], r);
            r = fmaf(pSH15, sh[15 * 3 + c], r);
        }
        rgb[c] = fmaxf(r + 0.5f, 0.0f); /* gsplat.rasterization: clamp_min(colors + 0.5, 0) */
    }
}

/* ---- scene / camera description ---------------------------------------------------------- */
typedef struct {
    int64_t n;
    const float *means;     /* [n,3] */
    const float *quats;     /* [n,4] wxyz, unnormalised, or NULL */
    const float *scales;    /* [n,3] activated (exp applied), or NULL */
    const float *cov6;      /* [n,6] xx xy xz yy yz zz, used when quats == NULL */
    const float *opacities; /* [n] activated (sigmoid applied) */
    const float *colors;    /* sh_degree>=0: [n,(d+1)^2,3] SH; sh_degree<0: [n,3] final rgb */
    int32_t sh_degree;
    const uint8_t *group_id; /* [n] or NULL */
    int32_t n_groups;
    const float *group_Rt;   /* [n_groups,12] row-major 3x4 (R|t), or NULL */
} sas_oracle_scene;

typedef struct {
    float R[9], t[3];   /* world -> camera */
    float campos[3];
    float fx, fy, cx, cy;
    int W, H, tw, th;
} cam_t;

static void cam_from(const float viewmat[16], const float K[9], int W, int H, cam_t *c)
{
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c->R[3 * i + j] = viewmat[4 * i + j];
        c->t[i] = viewmat[4 * i + 3];
    }
    /* campos = -R^T t  (gsplat takes torch.inverse(viewmat)[:3,3]; a rigid viewmat makes these equal) */
    for (int i = 0; i < 3; ++i)
        c->campos[i] = -dot3(c->R[0 + i], c->R[3 + i], c->R[6 + i], c->t[0], c->t[1], c->t[2]);
    c->fx = K[0]; c->fy = K[4]; c->cx = K[2]; c->cy = K[5];
    c->W = W; c->H = H;
    c->tw = (W + OC_TILE - 1) / OC_TILE;
    c->th = (H + OC_TILE - 1) / OC_TILE;
}

/* per-Gaussian projection result */
typedef struct {
    int32_t rx, ry;          /* 0,0 = culled; saturated at 2147483520 (the largest float below 2^31) */
    float rxf, ryf;          /* the radii as computed (integer-valued floats): what T3 takes its rectangle from */
    float mx, my, depth;
    float ca, cb, cc;        /* conic */
    float rgb[3];
    float opac;
    float thr;               /* ln(255 opacity) + 1e-3: no pixel with sigma above it can reach alpha >= 1/255 */
} proj_t;

/* T1 + T2 for one Gaussian */
static void project_one(const sas_oracle_scene *s, const cam_t *c, int64_t i, proj_t *o)
{
    memset(o, 0, sizeof(*o));
    float m[3] = {s->means[3 * i], s->means[3 * i + 1], s->means[3 * i + 2]};
    const float *G = NULL;
    if (s->group_id && s->group_Rt) {
        G = s->group_Rt + 12 * (int64_t)s->group_id[i];
        float mg[3];
        for (int r = 0; r < 3; ++r) mg[r] = affine3(G + 4 * r, G[4 * r + 3], m);
        m[0] = mg[0]; m[1] = mg[1]; m[2] = mg[2];
    }
    float pc[3];
    for (int r = 0; r < 3; ++r) pc[r] = affine3(c->R + 3 * r, c->t[r], m);
    if (pc[2] < OC_NEAR || pc[2] > OC_FAR) return;

    /* world covariance */
    float cov[6];
    if (s->quats) {
        float R[9];
        quat_to_rotmat(s->quats + 4 * i, R);
        if (G) {
            float R2[9];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k)
                    R2[3 * r + k] = dot3(G[4 * r + 0], G[4 * r + 1], G[4 * r + 2], R[0 + k], R[3 + k], R[6 + k]);
            memcpy(R, R2, sizeof(R));
        }
        const float *sc = s->scales + 3 * i;
        float M[9];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) M[3 * r + k] = R[3 * r + k] * sc[k];
        cov[0] = dot3(M[0], M[1], M[2], M[0], M[1], M[2]);
        cov[1] = dot3(M[0], M[1], M[2], M[3], M[4], M[5]);
        cov[2] = dot3(M[0], M[1], M[2], M[6], M[7], M[8]);
        cov[3] = dot3(M[3], M[4], M[5], M[3], M[4], M[5]);
        cov[4] = dot3(M[3], M[4], M[5], M[6], M[7], M[8]);
        cov[5] = dot3(M[6], M[7], M[8], M[6], M[7], M[8]);
    } else {
        memcpy(cov, s->cov6 + 6 * i, sizeof(cov));
        if (G) {
            float Rg[9] = {G[0], G[1], G[2], G[4], G[5], G[6], G[8], G[9], G[10]};
            float c2[6];
            rot_sym3(Rg, cov, c2);
            memcpy(cov, c2, sizeof(cov));
        }
    }
    float cc3[6];
    rot_sym3(c->R, cov, cc3); /* camera-frame covariance */

    /* pinhole EWA projection (gsplat persp_proj) */
    float x = pc[0], y = pc[1], z = pc[2];
    float Wf = (float)c->W, Hf = (float)c->H;
    float tan_fovx = (0.5f * Wf) / c->fx;
    float tan_fovy = (0.5f * Hf) / c->fy;
    float lim_x_pos = fmaf(0.3f, tan_fovx, (Wf - c->cx) / c->fx);
    float lim_x_neg = fmaf(0.3f, tan_fovx, c->cx / c->fx);
    float lim_y_pos = fmaf(0.3f, tan_fovy, (Hf - c->cy) / c->fy);
    float lim_y_neg = fmaf(0.3f, tan_fovy, c->cy / c->fy);
    float rz = 1.0f / z;
    float rz2 = rz * rz;
    float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
    float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
    float ja = c->fx * rz, jb = -((c->fx * tx) * rz2);
    float jc = c->fy * rz, jd = -((c->fy * ty) * rz2);
    /* rows of J * Sigma_c */
    float t00 = fmaf(jb, cc3[2], ja * cc3[0]);
    float t01 = fmaf(jb, cc3[4], ja * cc3[1]);
    float t02 = fmaf(jb, cc3[5], ja * cc3[2]);
    float t11 = fmaf(jd, cc3[4], jc * cc3[3]);
    float t12 = fmaf(jd, cc3[5], jc * cc3[4]);
    float c00 = fmaf(t02, jb, t00 * ja);
    float c01 = fmaf(t02, jd, t01 * jc);
    float c11 = fmaf(t12, jd, t11 * jc);
    float mx = fmaf(c->fx, x * rz, c->cx);
    float my = fmaf(c->fy, y * rz, c->cy);

    c00 += OC_EPS2D;
    c11 += OC_EPS2D;
    float det = fmaf(c00, c11, -(c01 * c01));
    if (!(det > 0.0f)) return;
    float inv_det = 1.0f / det;
    float ca = c11 * inv_det, cb = -c01 * inv_det, ccn = c00 * inv_det;

    float op = s->opacities[i];
    if (op < OC_ALPHA_THRESHOLD) return;
    float lnq = sas_oracle_logf(op / OC_ALPHA_THRESHOLD);
    float extent = fminf(3.33f, sqrtf(2.0f * lnq));
    float b = 0.5f * (c00 + c11);
    float tmp = sqrtf(fmaxf(0.01f, fmaf(b, b, -det)));
    float v1 = b + tmp;
    float r1 = extent * sqrtf(v1);
    float rx = ceilf(fminf(extent * sqrtf(c00), r1));
    float ry = ceilf(fminf(extent * sqrtf(c11), r1));
    if (rx <= 0.0f && ry <= 0.0f) return;
    if (mx + rx <= 0.0f || mx - rx >= Wf || my + ry <= 0.0f || my - ry >= Hf) return;

    /* (a float beyond the int32 range is undefined behaviour in a C cast: absurd scales or a camera inside a Gaussian produce
     * such radii; the HIP side saturates its parity hook the same way and, like here, takes the rectangle from the float) */
    o->rx = (int32_t)fminf(rx, 2147483520.0f); o->ry = (int32_t)fminf(ry, 2147483520.0f);
    o->rxf = rx; o->ryf = ry;
    o->mx = mx; o->my = my; o->depth = z;
    o->ca = ca; o->cb = cb; o->cc = ccn;
    o->opac = op;
    o->thr = lnq + 1e-3f;
    if (o->rx <= 0 || o->ry <= 0) return; /* colour mask of gsplat.rasterization: (radii > 0).all(-1) */
    if (s->sh_degree >= 0) {
        int K = (s->sh_degree + 1) * (s->sh_degree + 1);
        sh_to_color(s->sh_degree, s->colors + (int64_t)3 * K * i,
                    m[0] - c->campos[0], m[1] - c->campos[1], m[2] - c->campos[2], o->rgb);
    } else {
        o->rgb[0] = s->colors[3 * i]; o->rgb[1] = s->colors[3 * i + 1]; o->rgb[2] = s->colors[3 * i + 2];
    }
    /* Contract T2 (round 5): a colour is finite when it leaves the projection (clamped to +-FLT_MAX: the identity on every
     * finite value, a NaN becomes -FLT_MAX) -- gsplat would carry an Inf to the pixels the Gaussian touches, where the
     * wrapper's clamp to [0, 1] ends it; the HIP loop adds skipped entries with weight +0, which only finite colours survive. */
    for (int ch = 0; ch < 3; ++ch) o->rgb[ch] = fminf(fmaxf(o->rgb[ch], -FLT_MAX), FLT_MAX);
}

/* T3: tile rectangle [x0,x1) x [y0,y1) of a projected Gaussian */
static inline void tile_rect(const proj_t *p, const cam_t *c, int *x0, int *x1, int *y0, int *y1)
{
    const float ts = (float)OC_TILE;
    float trx = p->rxf / ts, try_ = p->ryf / ts;
    float tx = p->mx / ts, ty = p->my / ts;
    float fx0 = floorf(tx - trx), fx1 = ceilf(tx + trx);
    float fy0 = floorf(ty - try_), fy1 = ceilf(ty + try_);
    float tw = (float)c->tw, th = (float)c->th;
    *x0 = (int)fminf(fmaxf(fx0, 0.0f), tw);
    *x1 = (int)fminf(fmaxf(fx1, 0.0f), tw);
    *y0 = (int)fminf(fmaxf(fy0, 0.0f), th);
    *y1 = (int)fminf(fmaxf(fy1, 0.0f), th);
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* T6, per (Gaussian, tile): the mean relative to the TILE'S CENTRE (X0, Y0), u = mx - X0, v = my - Y0, and the halved conic
 * hA = A/2, hC = C/2.  Per pixel with centre (x, y) relative to the same origin (-7.5 .. 7.5):
 *   dx = u - x, dy = v - y,   sigma = fma(dx, fma(B, dy, hA dx), (hC dy) dy)
 * -- gsplat's 0.5 (A dx^2 + C dy^2) + B dx dy evaluated on (dx, dy) itself with two fused steps (round 5).  Its terms are small
 * where sigma is small, so it sits at the float32 noise floor of the written form (one pixel beyond 1e-4 at config 3 against
 * the all-textbook float32 frame, twelve beyond 1e-6: tests/tools/deviation_table.py --full).  Rounds 1-4 expanded sigma into
 * a POLYNOMIAL in (x, y) -- k0 + k1 x + k2 y + hA x^2 + hC y^2 + B x y, five fused operations per pair instead of seven --
 * whose terms cancel (four pixels beyond 1e-4, 119 beyond 1e-6 at config 3); it is kept as study variant 32 (16: about the
 * tile's corner, the form of rounds 1-3).  The conic is positive semi-definite, so gsplat's `sigma < 0` guard can only fire
 * on rounding noise and is not part of the contract. */
typedef struct { float k0, k1, k2, hA, hC, B, u, v; } tcoef_t;

static inline void tile_coefs(const proj_t *g, float X0, float Y0, tcoef_t *t)
{
    float u = g->mx - X0, v = g->my - Y0;
    float A = g->ca, B = g->cb, C = g->cc;
    float hA = 0.5f * A, hC = 0.5f * C;
    float bu = B * u;
    t->k1 = -fmaf(A, u, B * v);
    t->k2 = -fmaf(C, v, bu);
    t->k0 = fmaf(hA * u, u, fmaf(hC * v, v, bu * v));
    t->hA = hA; t->hC = hC; t->B = B;
    t->u = u; t->v = v;
}

/* T6 for one pixel; (x, y) = pixel centre relative to the tile's centre */
static inline void blend_pixel(const proj_t *P, const int32_t *ids, const tcoef_t *tc, int64_t n, float x, float y,
                               float out_acc[4], float *out_T)
{
    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, ad = 0.0f;
    for (int64_t k = 0; k < n; ++k) {
        const proj_t *g = &P[ids[k]];
        const tcoef_t *t = &tc[k];
        const float dx = t->u - x, dy = t->v - y;
        float sigma = fmaf(dx, fmaf(t->B, dy, t->hA * dx), (t->hC * dy) * dy);
        float alpha = fminf(OC_MAX_ALPHA, g->opac * sas_oracle_expf(-sigma));
        if (alpha < OC_ALPHA_THRESHOLD) continue;
        float vis = alpha * T;
        float next_T = T - vis;
        if (next_T <= OC_T_STOP) break;
        ar = fmaf(g->rgb[0], vis, ar);
        ag = fmaf(g->rgb[1], vis, ag);
        ab = fmaf(g->rgb[2], vis, ab);
        ad = fmaf(g->depth, vis, ad);
        T = next_T;
    }
    out_acc[0] = ar; out_acc[1] = ag; out_acc[2] = ab; out_acc[3] = ad;
    *out_T = T;
}

/* ---- deviation study (tests/test_oracle.py, tools/deviation_table.py) ---------------------------------
 * The contract departs from gsplat's WRITTEN arithmetic in four places, each chosen for the GPU loop.
 * A variant mask switches any of them back to the textbook float32 form, so that the effect of each on an
 * image can be measured against the float64 twin (oracle/np_twin.py, which keeps all four textbook forms):
 *   1  sigma = 0.5 (A dx^2 + C dy^2) + B dx dy, unfused, on (dx, dy) = (mx - px, my - py) instead of the contract's fused form on (u - x, v - y)
 *   2  gsplat's `sigma < 0 -> skip` guard restored
 *   4  T' = T (1 - alpha) instead of T - alpha T
 *   8  libm expf instead of the degree-5 polynomial
 *  32  (study) the contract of rounds 1-4: sigma as a polynomial in the tile-local pixel centre (16 on top: about the tile's corner)
 * Mask 0 is the contract and the only thing the parity tests and the CPU baseline ever run. */
static int g_variant = 0;
void sas_oracle_set_variant(int mask) { g_variant = mask; }

static inline void blend_pixel_variant(int variant, const proj_t *P, const int32_t *ids, const tcoef_t *tc, int64_t n,
                                       float x, float y, float px, float py, float out_acc[4], float *out_T)
{
    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, ad = 0.0f;
    const float xx = x * x, yy = y * y, xy = x * y;
    for (int64_t k = 0; k < n; ++k) {
        const proj_t *g = &P[ids[k]];
        const tcoef_t *t = &tc[k];
        float sigma;
        if (variant & 1) {
            float dx = g->mx - px, dy = g->my - py;
            sigma = 0.5f * (g->ca * dx * dx + g->cc * dy * dy) + g->cb * dx * dy;
        } else if (variant & 32) {
            sigma = fmaf(t->B, xy, fmaf(t->hC, yy, fmaf(t->hA, xx, fmaf(t->k2, y, fmaf(t->k1, x, t->k0)))));
        } else {
            float dx = t->u - x, dy = t->v - y;
            sigma = fmaf(dx, fmaf(t->B, dy, t->hA * dx), (t->hC * dy) * dy);
        }
        if ((variant & 2) && sigma < 0.0f) continue;
        float e = (variant & 8) ? expf(-sigma) : sas_oracle_expf(-sigma);
        float alpha = fminf(OC_MAX_ALPHA, g->opac * e);
        if (alpha < OC_ALPHA_THRESHOLD) continue;
        float vis = alpha * T;
        float next_T = (variant & 4) ? T * (1.0f - alpha) : T - vis;
        if (next_T <= OC_T_STOP) break;
        ar = fmaf(g->rgb[0], vis, ar);
        ag = fmaf(g->rgb[1], vis, ag);
        ab = fmaf(g->rgb[2], vis, ab);
        ad = fmaf(g->depth, vis, ad);
        T = next_T;
    }
    out_acc[0] = ar; out_acc[1] = ag; out_acc[2] = ab; out_acc[3] = ad;
    *out_T = T;
}

/* ---- deviation study, one pixel: WHERE two variants part (tests/test_oracle.py) --------------------------------
 * Replays the list of pixel (px, py)'s tile under variant masks va and vb side by side and reports the first entry at
 * which their DECISIONS differ (0 composited, 1 skipped: alpha < 1/255, 2 skipped: sigma < 0 guard, 3 stopped: T' <= 1e-4).
 * Two float32 evaluations of one formula can only part company by more than rounding noise through such a flip; the
 * test shows, for every pixel that differs by more than the tolerance, the flipped splat and how close both sides sat
 * to the threshold.  out = {entry k (-1: no decision differs), Gaussian index, decision a, decision b},
 * vals = {alpha a, alpha b, T' a, T' b, sigma a, sigma b}.  Returns 0, -1 on allocation failure. */
static inline int decide_variant(int variant, const proj_t *g, const tcoef_t *t, float x, float y, float px, float py, float T,
                                 float *o_alpha, float *o_nT, float *o_sigma)
{
    float sigma;
    if (variant & 1) {
        float dx = g->mx - px, dy = g->my - py;
        sigma = 0.5f * (g->ca * dx * dx + g->cc * dy * dy) + g->cb * dx * dy;
    } else if (variant & 32) {
        const float xx = x * x, yy = y * y, xy = x * y;
        sigma = fmaf(t->B, xy, fmaf(t->hC, yy, fmaf(t->hA, xx, fmaf(t->k2, y, fmaf(t->k1, x, t->k0)))));
    } else {
        float dx = t->u - x, dy = t->v - y;
        sigma = fmaf(dx, fmaf(t->B, dy, t->hA * dx), (t->hC * dy) * dy);
    }
    *o_sigma = sigma;
    *o_alpha = 0.0f;
    *o_nT = T;
    if ((variant & 2) && sigma < 0.0f) return 2;
    float e = (variant & 8) ? expf(-sigma) : sas_oracle_expf(-sigma);
    float alpha = fminf(OC_MAX_ALPHA, g->opac * e);
    *o_alpha = alpha;
    if (alpha < OC_ALPHA_THRESHOLD) return 1;
    float vis = alpha * T;
    float next_T = (variant & 4) ? T * (1.0f - alpha) : T - vis;
    *o_nT = next_T;
    if (next_T <= OC_T_STOP) return 3;
    return 0;
}

int sas_oracle_trace_pixel(const sas_oracle_scene *s, const float viewmat[16], const float K[9], int W, int H, int px, int py,
                           int va, int vb, int64_t out[4], float vals[6])
{
    cam_t c;
    cam_from(viewmat, K, W, H, &c);
    const int64_t n = s->n;
    const int tx = px / OC_TILE, ty = py / OC_TILE;
    proj_t *P = (proj_t *)malloc(sizeof(proj_t) * (size_t)(n > 0 ? n : 1));
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    if (!P || !keys) { free(P); free(keys); return -1; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) project_one(s, &c, i, &P[i]);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (P[i].rx <= 0 || P[i].ry <= 0) continue;
        int x0, x1, y0, y1;
        tile_rect(&P[i], &c, &x0, &x1, &y0, &y1);
        if (tx >= x0 && tx < x1 && ty >= y0 && ty < y1) keys[m++] = ((uint64_t)f2u(P[i].depth) << 32) | (uint64_t)(uint32_t)i;
    }
    qsort(keys, (size_t)m, sizeof(uint64_t), cmp_u64);
    const float x = (float)(px - tx * OC_TILE) + 0.5f - OC_TILE_CENTRE, y = (float)(py - ty * OC_TILE) + 0.5f - OC_TILE_CENTRE;
    float Ta = 1.0f, Tb = 1.0f;
    out[0] = -1; out[1] = -1; out[2] = 0; out[3] = 0;
    for (int k = 0; k < 6; ++k) vals[k] = 0.0f;
    for (int64_t k = 0; k < m; ++k) {
        const proj_t *g = &P[(uint32_t)(keys[k] & 0xffffffffu)];
        tcoef_t t;
        tile_coefs(g, (float)(tx * OC_TILE) + OC_TILE_CENTRE, (float)(ty * OC_TILE) + OC_TILE_CENTRE, &t);
        float aa, ab, na, nb, sa, sb;
        const int da = decide_variant(va, g, &t, x, y, (float)px + 0.5f, (float)py + 0.5f, Ta, &aa, &na, &sa);
        const int db = decide_variant(vb, g, &t, x, y, (float)px + 0.5f, (float)py + 0.5f, Tb, &ab, &nb, &sb);
        if (da != db) {
            out[0] = k; out[1] = (int64_t)(uint32_t)(keys[k] & 0xffffffffu); out[2] = da; out[3] = db;
            vals[0] = aa; vals[1] = ab; vals[2] = na; vals[3] = nb; vals[4] = sa; vals[5] = sb;
            break;
        }
        if (da == 3) break;
        if (da == 0) { Ta = na; Tb = nb; }
    }
    free(P); free(keys);
    return 0;
}

/*
 * Full frame.  depth_mode: 0 = expected depth ED = acc_d / max(alpha,1e-10) (gsplat "RGB+ED");
 *              1 = nerfstudio fill, where(alpha > 0, ED, max(ED)).
 * Optional outputs may be NULL.  Projection dumps are [n]-sized; tile_offsets is [tiles+1];
 * sorted_ids receives at most sorted_cap entries.  stats = {n_visible, n_intersections (gsplat's rectangles), 0}.
 * Returns 0, or -1 on allocation failure.
 */
int sas_oracle_render(const sas_oracle_scene *s, const float viewmat[16], const float K[9], int W, int H,
                      const float bg[3], int depth_mode,
                      float *rgb, float *alpha, float *depth, uint8_t *rgb8,
                      int32_t *o_radii, float *o_means2d, float *o_depths, float *o_conics, float *o_colors,
                      int32_t *o_tile_offsets, int32_t *o_sorted_ids, int64_t sorted_cap, int64_t *stats)
{
    cam_t c;
    cam_from(viewmat, K, W, H, &c);
    const int64_t n = s->n;
    const int tiles = c.tw * c.th;
    proj_t *P = (proj_t *)malloc(sizeof(proj_t) * (size_t)(n > 0 ? n : 1));
    int64_t *tcount = (int64_t *)calloc((size_t)tiles + 1, sizeof(int64_t));
    if (!P || !tcount) { free(P); free(tcount); return -1; }

#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) project_one(s, &c, i, &P[i]);

    int64_t n_vis = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (P[i].rx <= 0 || P[i].ry <= 0) continue; /* gsplat isect_tiles: radius_x <= 0 || radius_y <= 0 */
        ++n_vis;
        int x0, x1, y0, y1;
        tile_rect(&P[i], &c, &x0, &x1, &y0, &y1);
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) tcount[ty * c.tw + tx + 1]++;
    }
    for (int t = 0; t < tiles; ++t) tcount[t + 1] += tcount[t];
    const int64_t M = tcount[tiles];
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(M > 0 ? M : 1));
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
    int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)(tiles + 1));
    if (!keys || !ids || !cursor) { free(P); free(tcount); free(keys); free(ids); free(cursor); return -1; }
    memcpy(cursor, tcount, sizeof(int64_t) * (size_t)(tiles + 1));
    for (int64_t i = 0; i < n; ++i) {
        if (P[i].rx <= 0 || P[i].ry <= 0) continue;
        int x0, x1, y0, y1;
        tile_rect(&P[i], &c, &x0, &x1, &y0, &y1);
        uint64_t key = ((uint64_t)f2u(P[i].depth) << 32) | (uint64_t)(uint32_t)i;
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) keys[cursor[ty * c.tw + tx]++] = key;
    }
    /* T4: per tile ascending (depth bits, Gaussian index) == the global stable radix sort of gsplat */
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < tiles; ++t) {
        int64_t a = tcount[t], b = tcount[t + 1];
        qsort(keys + a, (size_t)(b - a), sizeof(uint64_t), cmp_u64);
        for (int64_t k = a; k < b; ++k) ids[k] = (int32_t)(uint32_t)(keys[k] & 0xffffffffu);
    }

    /* per-thread scratch for the per-(Gaussian, tile) coefficients of the longest list */
    int64_t max_len = 1;
    for (int t = 0; t < tiles; ++t)
        if (tcount[t + 1] - tcount[t] > max_len) max_len = tcount[t + 1] - tcount[t];
    tcoef_t *tc_all = (tcoef_t *)malloc(sizeof(tcoef_t) * (size_t)max_len * (size_t)sas_oracle_num_threads());
    if (!tc_all) { free(P); free(tcount); free(keys); free(ids); free(cursor); return -1; }

    float maxED = 0.0f;
#pragma omp parallel for schedule(dynamic, 4) reduction(max : maxED)
    for (int t = 0; t < tiles; ++t) {
        int ty = t / c.tw, tx = t % c.tw;
        const int32_t *tl = ids + tcount[t];
        int64_t tn = tcount[t + 1] - tcount[t];
        tcoef_t *tc = tc_all + (size_t)oc_thread_id() * (size_t)max_len;
        /* the tile-local frame has its origin at the tile's CENTRE (variant 16, with 32: the polynomial about the corner, rounds 1-3, for the study) */
        const float cen = (g_variant & 16) ? 0.0f : OC_TILE_CENTRE;
        for (int64_t k = 0; k < tn; ++k) tile_coefs(&P[tl[k]], (float)(tx * OC_TILE) + cen, (float)(ty * OC_TILE) + cen, &tc[k]);
        for (int yy = 0; yy < OC_TILE; ++yy) {
            int i = ty * OC_TILE + yy;
            if (i >= H) break;
            for (int xx = 0; xx < OC_TILE; ++xx) {
                int j = tx * OC_TILE + xx;
                if (j >= W) break;
                float acc[4], T;
                if (g_variant == 0) blend_pixel(P, tl, tc, tn, (float)xx + 0.5f - cen, (float)yy + 0.5f - cen, acc, &T);
                else blend_pixel_variant(g_variant, P, tl, tc, tn, (float)xx + 0.5f - cen, (float)yy + 0.5f - cen,
                                         (float)j + 0.5f, (float)i + 0.5f, acc, &T);
                float a = 1.0f - T;
                int64_t pix = (int64_t)i * W + j;
                float ED = acc[3] / fmaxf(a, 1e-10f);
                if (ED > maxED) maxED = ED;
                if (alpha) alpha[pix] = a;
                if (depth) depth[pix] = ED;
                for (int ch = 0; ch < 3; ++ch) {
                    /* T0: rgb = clamp(render + (1 - alpha) * background, 0, 1), two roundings */
                    float v = acc[ch] + (1.0f - a) * bg[ch];
                    v = fminf(fmaxf(v, 0.0f), 1.0f);
                    if (rgb) rgb[3 * pix + ch] = v;
                    if (rgb8) rgb8[3 * pix + ch] = (uint8_t)(int)floorf(fmaf(v, 255.0f, 0.5f));
                }
            }
        }
    }
    if (depth && depth_mode == 1) {
        const int64_t np = (int64_t)W * H;
        if (alpha) {
            for (int64_t p = 0; p < np; ++p)
                if (!(alpha[p] > 0.0f)) depth[p] = maxED;
        } else {
            /* alpha == 0 exactly <=> no Gaussian blended <=> ED == 0 and acc_d == 0 */
            for (int64_t p = 0; p < np; ++p)
                if (depth[p] == 0.0f) depth[p] = maxED;
        }
    }

    if (o_radii || o_means2d || o_depths || o_conics || o_colors) {
        for (int64_t i = 0; i < n; ++i) {
            if (o_radii) { o_radii[2 * i] = P[i].rx; o_radii[2 * i + 1] = P[i].ry; }
            if (o_means2d) { o_means2d[2 * i] = P[i].mx; o_means2d[2 * i + 1] = P[i].my; }
            if (o_depths) o_depths[i] = P[i].depth;
            if (o_conics) { o_conics[3 * i] = P[i].ca; o_conics[3 * i + 1] = P[i].cb; o_conics[3 * i + 2] = P[i].cc; }
            if (o_colors) { o_colors[3 * i] = P[i].rgb[0]; o_colors[3 * i + 1] = P[i].rgb[1]; o_colors[3 * i + 2] = P[i].rgb[2]; }
        }
    }
    if (o_tile_offsets)
        for (int t = 0; t <= tiles; ++t) o_tile_offsets[t] = (int32_t)tcount[t];
    if (o_sorted_ids) {
        int64_t m = M < sorted_cap ? M : sorted_cap;
        memcpy(o_sorted_ids, ids, sizeof(int32_t) * (size_t)m);
    }
    if (stats) { stats[0] = n_vis; stats[1] = M; stats[2] = 0; }
    free(P); free(tcount); free(keys); free(ids); free(cursor); free(tc_all);
    return 0;
}

/* RGB-D consumer, nerfstudio_utils.py:424-445: x = (u - cx) * d / fx, y = (v - cy) * d / fy, z = d
 * with integer pixel indices u, v (torch.arange), mask = d < max_depth (all ones when max_depth is
 * NULL).  The reference evaluates it in float32 in exactly this order. */
void sas_oracle_unproject(const float *depth, const float K[9], int W, int H, const float *max_depth, float *points,
                          uint8_t *mask)
{
    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const int64_t p = (int64_t)v * W + u;
            const float d = depth[p];
            if (points) {
                points[3 * p] = ((float)u - cx) * d / fx;
                points[3 * p + 1] = ((float)v - cy) * d / fy;
                points[3 * p + 2] = d;
            }
            if (mask) mask[p] = max_depth ? (d < *max_depth ? 1 : 0) : 1;
        }
}

int sas_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void sas_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
