/* Sanitizer leg of the CPU oracle (test infrastructure): renders a small seeded scene, including
 * ragged image sizes, group poses, culled and empty scenes and poisoned scenes (NaN, Inf, absurd magnitudes), under -fsanitize=address,undefined.
 * Exit code 0 and no sanitizer report = pass.  Built by `make -C oracle asan`. */
#include "sas_oracle.c"
#include <stdio.h>

static unsigned long long rs = 88172645463325252ull;
static float frand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 11) * (1.0 / 9007199254740992.0)); }

static int run(int n, int W, int H, int groups, int degree, int poison)
{
    float *means = malloc(sizeof(float) * 3 * (n + 1)), *quats = malloc(sizeof(float) * 4 * (n + 1));
    float *scales = malloc(sizeof(float) * 3 * (n + 1)), *op = malloc(sizeof(float) * (n + 1));
    int kk = degree >= 0 ? (degree + 1) * (degree + 1) : 1;
    float *col = malloc(sizeof(float) * 3 * kk * (n + 1));
    uint8_t *gid = malloc(n + 1);
    float Rt[12 * 4];
    for (int g = 0; g < 4; ++g) { float id[12] = {1, 0, 0, 0.01f * g, 0, 1, 0, 0, 0, 0, 1, -0.02f * g}; memcpy(Rt + 12 * g, id, sizeof(id)); }
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { means[3 * i + k] = 2 * frand() - 1; scales[3 * i + k] = 0.005f + 0.1f * frand(); }
        for (int k = 0; k < 4; ++k) quats[4 * i + k] = 2 * frand() - 1 + (k == 0);
        op[i] = frand();
        for (int k = 0; k < 3 * kk; ++k) col[3 * kk * i + k] = frand() - 0.3f;
        gid[i] = (uint8_t)(i % 4);
    }
    if (poison) {   /* absurd inputs (NaN, Inf, 1e30, scales that give radii beyond int32): no cast, index or shift may leave its range */
        const float bad[10] = {NAN, INFINITY, -INFINITY, 1e30f, -1e30f, 1e-30f, 0.0f, 3e6f, 1e12f, -1.0f};
        for (int i = 0; i < n; i += 7) {
            const float v = bad[(i / 7) % 10];
            switch ((i / 70) % 5) {
            case 0: means[3 * i + (i % 3)] = v; break;
            case 1: scales[3 * i + (i % 3)] = v; break;
            case 2: quats[4 * i + (i % 4)] = v; break;
            case 3: op[i] = v; break;
            default: col[3 * kk * i + (i % (3 * kk))] = v; break;
            }
        }
    }
    sas_oracle_scene s = {n, means, quats, scales, NULL, op, col, degree, groups ? gid : NULL, groups ? 4 : 0, groups ? Rt : NULL};
    float V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 3, 0, 0, 0, 1};
    float K[9] = {0.8f * W, 0, 0.5f * W, 0, 0.8f * W, 0.5f * H, 0, 0, 1};
    float bg[3] = {0.1f, 0.2f, 0.3f};
    float *rgb = malloc(sizeof(float) * 3 * W * H), *a = malloc(sizeof(float) * W * H), *d = malloc(sizeof(float) * W * H);
    uint8_t *r8 = malloc(3 * W * H);
    int tiles = ((W + 15) / 16) * ((H + 15) / 16);
    int32_t *rad = malloc(sizeof(int32_t) * 2 * (n + 1)), *toff = malloc(sizeof(int32_t) * (tiles + 1));
    float *m2 = malloc(sizeof(float) * 2 * (n + 1)), *dep = malloc(sizeof(float) * (n + 1)), *con = malloc(sizeof(float) * 3 * (n + 1)), *cc = malloc(sizeof(float) * 3 * (n + 1));
    int64_t st[3];
    int rc = sas_oracle_render(&s, V, K, W, H, bg, 1, rgb, a, d, r8, rad, m2, dep, con, cc, toff, NULL, 0, st);
    int32_t *ids = malloc(sizeof(int32_t) * (st[1] + 1));
    rc |= sas_oracle_render(&s, V, K, W, H, bg, 0, rgb, NULL, d, NULL, NULL, NULL, NULL, NULL, NULL, NULL, ids, st[1], st);
    printf("n=%d %dx%d groups=%d deg=%d -> visible %lld, intersections %lld\n", n, W, H, groups, degree, (long long)st[0], (long long)st[1]);
    free(means); free(quats); free(scales); free(op); free(col); free(gid); free(rgb); free(a); free(d); free(r8);
    free(rad); free(toff); free(m2); free(dep); free(con); free(cc); free(ids);
    return rc;
}

int main(void)
{
    int rc = 0;
    rc |= run(3000, 97, 61, 0, 3, 0);
    rc |= run(500, 16, 16, 1, 2, 0);
    rc |= run(1, 1, 1, 0, 0, 0);
    rc |= run(0, 33, 20, 0, 3, 0);
    rc |= run(800, 40, 40, 1, -1, 0);
    rc |= run(3000, 97, 61, 1, 3, 1);
    rc |= run(1500, 50, 70, 0, -1, 1);
    return rc;
}
