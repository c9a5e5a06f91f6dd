/*
 * gs_oracle.c -- CPU restatement of the tile-based differentiable Gaussian
 * rasterizer that GaMeS calls through `diff_gaussian_rasterization`.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product path
 * (gaussian-mesh-splatting_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for the rasterizer.  The algorithm lives in
 * a third-party dependency that is ABSENT from /root/reference (empty git
 * submodule submodules/diff-gaussian-rasterization, .gitmodules:4-6, url
 * github.com/graphdeco-inria/diff-gaussian-rasterization, pinned commit lost;
 * API shape = upstream main after the Oct-2024 antialiasing/inverse-depth
 * merge).  The reference ships no tests or golden vectors for it (SURVEY.md
 * section 0.3).  This file restates the PUBLISHED algorithm (Kerbl et al. 2023,
 * "3D Gaussian Splatting", sections 4-6 + appendix A; EWA: Zwicker et al. 2001)
 * with the constants of SURVEY.md appendix A, anchored on the reference's call
 * sites:
 *   renderer/gaussian_renderer/__init__.py:43-57   settings (13 fields)
 *   renderer/gaussian_renderer/__init__.py:94-102  call + 3-tuple return
 *   scene/cameras.py:54-57                         matrix layout (transposed)
 *   scene/gaussian_model.py:27-31, utils/general_utils.py:158-190  cov3D
 *   utils/sh_utils.py:57-112                       SH polynomial + constants
 * It is validated against (a) oracle/dense_torch.py, an independent float64
 * autograd formulation, and (b) the in-tree python cov3D / SH stages.
 *
 * Build:  make -C oracle      (two variants: float32 and float64 `real`)
 *
 * Conventions
 *   - matrices arrive in the reference's transposed layout: element (row r,
 *     col c) of the mathematical matrix is m[4*c + r].
 *   - quaternion is (w, x, y, z) and is NOT re-normalised here (the reference
 *     normalises in python: scene/gaussian_model.py:100-101).
 *   - discrete decisions (cull, radius, tile rect, alpha/T thresholds) use a
 *     fixed documented operation order (explicit FMA chains) so that an
 *     independent float32 implementation can reproduce them bit-for-bit;
 *     where a transcendental makes that impossible (exp), near-threshold
 *     cases are reported through ambiguity masks instead of being hidden.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_DOUBLE
typedef double real;
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#define R_FMA fma
#define R_FABS fabs
#else
typedef float real;
#define R_SQRT sqrtf
#define R_EXP expf
#define R_CEIL ceilf
#define R_FMA fmaf
#define R_FABS fabsf
#endif

#define TILE 16
#define NEAR_Z ((real)0.2)
#define DILATE ((real)0.3)
#define ALPHA_MIN ((real)(1.0 / 255.0))
#define ALPHA_MAX ((real)0.99)
#define T_MIN ((real)0.0001)

/* utils/sh_utils.py:26-43 */
static const real SH_C0 = (real)0.28209479177387814;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C2[5] = {(real)1.0925484305920792, (real)-1.0925484305920792,
                              (real)0.31539156525252005, (real)-1.0925484305920792,
                              (real)0.5462742152960396};
static const real SH_C3[7] = {(real)-0.5900435899266435, (real)2.890611442640554,
                              (real)-0.4570457994644658, (real)0.3731763325901154,
                              (real)-0.4570457994644658, (real)1.445305721320277,
                              (real)-0.5900435899266435};

/* Exponent of the Gaussian at a pixel offset (dx, dy) for the conic (A, B, C), in ONE documented operation order
 * shared by every consumer (forward walk, backward walk, and the HIP kernels' gms_blend.h::pair_power): two explicit
 * FMAs, every other product rounded on its own.  The skip tests (power > 0, alpha < 1/255) are discrete decisions on
 * this value: for elongated splats far from the pixel the three terms cancel (|term| ~ 10^3 for |power| ~ 5), so two
 * evaluation orders can differ by 1e-4 in `power` -- enough to flip the 1/255 test.  A fixed order removes that. */
static inline real pair_power(real A, real B, real C, real dx, real dy)
{
#ifdef ORACLE_UPSTREAM_ORDER
    /* The published formula evaluated as written, left to right, every operation rounded on its own (no FMA):
     *     power = -0.5f * (con_o.x * d.x * d.x + con_o.z * d.y * d.y) - con_o.y * d.x * d.y;
     * A fourth build of the checker (libgs_oracle_f32up): tests/test_oracle_raster.py compares it with the default build to
     * bound what sharing ONE evaluation order between the checker and the kernels could hide. */
    return (real)-0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy;
#else
    const real a = A * dx, c = C * dy, b = B * dx;
    const real s = R_FMA(a, dx, c * dy);
    return R_FMA((real)-0.5, s, -(b * dy));
#endif
}

typedef struct {
    int P;              /* number of Gaussians */
    int D;              /* active SH degree 0..3 */
    int M;              /* SH coefficients stored per Gaussian (16) */
    int W, H;
    const real *means3D;        /* [P,3] */
    const real *shs;            /* [P,M,3] or NULL */
    const real *colors_precomp; /* [P,3] or NULL */
    const real *opacities;      /* [P] */
    const real *scales;         /* [P,3] or NULL */
    const real *rotations;      /* [P,4] or NULL */
    const real *cov3D_precomp;  /* [P,6] or NULL */
    real scale_modifier;
    const real *viewmatrix;     /* 16 */
    const real *projmatrix;     /* 16 */
    const real *campos;         /* 3 */
    const real *bg;             /* 3 */
    real tanfovx, tanfovy;
    int antialiasing;
    int nthreads;               /* <=0: all */
} OrScene;

typedef struct {
    int P, W, H, gx, gy;
    long N;               /* number of (gaussian, tile) instances */
    real *depth;          /* [P] */
    real *xy;             /* [P,2] */
    real *conic_op;       /* [P,4] */
    real *rgb;            /* [P,3] */
    real *cov3D;          /* [P,6] */
    uint8_t *clamped;     /* [P,3] */
    int *radii;           /* [P] */
    int *rect;            /* [P,4] minx,miny,maxx,maxy */
    uint32_t *point_list; /* [N] gaussian id, sorted by (tile, depth, id) */
    long *ranges;         /* [T,2] */
    real *final_T;        /* [HW] */
    int *n_contrib;       /* [HW] */
    uint8_t *gauss_ambig; /* [P]  bit 0: discrete per-Gaussian decision near a threshold; bit 1: a (pixel, Gaussian) skip /
                           *      stop decision within exp() rounding; bit 2: composited in a pixel that holds such a
                           *      decision of ANY Gaussian (see the compositing loop) */
    uint8_t *pix_ambig;   /* [HW] a per-pixel skip/stop test was near its threshold */
    int amb_policy;       /* how this frame decided the pairs whose skip test is within exp() rounding (or_set_amb_policy) */
    double interactions;  /* pixel x Gaussian pairs evaluated */
} OrState;

static void set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* a*b + c*d + e*f + g with the fixed order ((a*b (+) c*d) (+) e*f) + g, each (+) fused */
static inline real dot3p(real a, real b, real c, real d, real e, real f, real g)
{
    real t = a * b;
    t = R_FMA(c, d, t);
    t = R_FMA(e, f, t);
    return t + g;
}

static void cov3d_from_scale_rot(const real *s_in, real mod, const real *q, real *cov)
{
    /* Sigma = R S S^T R^T  (scene/gaussian_model.py:27-31; R from utils/general_utils.py:158-179,
     * minus the normalisation which the reference applies in python before the call) */
    real s[3] = {mod * s_in[0], mod * s_in[1], mod * s_in[2]};
    real r = q[0], x = q[1], y = q[2], z = q[3];
    real R[3][3] = {
        {1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
        {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
        {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
    real L[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * s[j];
    real S[3][3];
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
            S[i][k] = L[i][0] * L[k][0] + L[i][1] * L[k][1] + L[i][2] * L[k][2];
    cov[0] = S[0][0]; cov[1] = S[0][1]; cov[2] = S[0][2];
    cov[3] = S[1][1]; cov[4] = S[1][2]; cov[5] = S[2][2];
}

/* SH -> RGB before the +0.5 and clamp (utils/sh_utils.py:57-112 restated per Gaussian) */
static void sh_to_rgb(int deg, int M, const real *sh /*[M,3]*/, const real *dir, real *out)
{
    (void)M;
    real x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < 3; c++) {
        real res = SH_C0 * sh[0 * 3 + c];
        if (deg > 0) {
            res = res - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
                      SH_C2[2] * (2 * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
                      SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3 * xx - yy) * sh[9 * 3 + c] +
                          SH_C3[1] * xy * z * sh[10 * 3 + c] +
                          SH_C3[2] * y * (4 * zz - xx - yy) * sh[11 * 3 + c] +
                          SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[12 * 3 + c] +
                          SH_C3[4] * x * (4 * zz - xx - yy) * sh[13 * 3 + c] +
                          SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] +
                          SH_C3[6] * x * (xx - 3 * yy) * sh[15 * 3 + c];
                }
            }
        }
        out[c] = res;
    }
}

static inline int near_int(real v, real tol)
{
    real f = v - (real)floor((double)v);
    return f < tol || f > 1 - tol;
}

/* Decision policy for the (pixel, Gaussian) pairs whose skip test lies within exp() rounding of its threshold (the same
 * bands that set pix_ambig bit 0): 0 = decide them as computed; +1 = they all contribute; -1 = none of them does.  An
 * independent float32 implementation lands on either side of each such test, and the outcome changes that pixel's term
 * in the gradient of EVERY Gaussian composited there (transmittance behind the pair, colour behind for those in front):
 * tests accept a row that agrees with the oracle under one of the two forced outcomes (tests/_util.py). */
static int g_amb_policy = 0;
void or_set_amb_policy(int p) { g_amb_policy = p < 0 ? -1 : (p > 0 ? 1 : 0); }
static inline int skip_by_power(int policy, real power)
{
    if (policy && R_FABS(power) < (real)1e-6) return policy < 0;
    return power > 0;
}
static inline int skip_by_alpha(int policy, real araw, real alpha)
{
    if (policy && R_FABS(araw * 255 - 1) < (real)4e-5) return policy < 0;
    return alpha < ALPHA_MIN;
}

typedef struct { uint32_t tile; uint32_t dbits; uint32_t id; } Inst;

static int inst_cmp(const void *a, const void *b)
{
    const Inst *x = (const Inst *)a, *y = (const Inst *)b;
    if (x->tile != y->tile) return x->tile < y->tile ? -1 : 1;
    if (x->dbits != y->dbits) return x->dbits < y->dbits ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

void or_free(OrState *st)
{
    if (!st) return;
    free(st->depth); free(st->xy); free(st->conic_op); free(st->rgb); free(st->cov3D);
    free(st->clamped); free(st->radii); free(st->rect); free(st->point_list); free(st->ranges);
    free(st->final_T); free(st->n_contrib); free(st->gauss_ambig); free(st->pix_ambig);
    free(st);
}

/* ------------------------------------------------------------------ forward */
OrState *or_forward(const OrScene *sc, real *out_color /*[3,H,W]*/, real *out_invdepth /*[H,W]*/,
                    int *out_radii /*[P]*/)
{
    set_threads(sc->nthreads);
    const int P = sc->P, W = sc->W, H = sc->H;
    OrState *st = (OrState *)calloc(1, sizeof(OrState));
    st->P = P; st->W = W; st->H = H;
    st->gx = (W + TILE - 1) / TILE; st->gy = (H + TILE - 1) / TILE;
    const int gx = st->gx, gy = st->gy;
    const long T = (long)gx * gy;
    size_t Pn = P > 0 ? (size_t)P : 1;
    st->depth = (real *)calloc(Pn, sizeof(real));
    st->xy = (real *)calloc(Pn * 2, sizeof(real));
    st->conic_op = (real *)calloc(Pn * 4, sizeof(real));
    st->rgb = (real *)calloc(Pn * 3, sizeof(real));
    st->cov3D = (real *)calloc(Pn * 6, sizeof(real));
    st->clamped = (uint8_t *)calloc(Pn * 3, 1);
    st->radii = (int *)calloc(Pn, sizeof(int));
    st->rect = (int *)calloc(Pn * 4, sizeof(int));
    st->gauss_ambig = (uint8_t *)calloc(Pn, 1);
    st->ranges = (long *)calloc((size_t)T * 2, sizeof(long));
    st->final_T = (real *)calloc((size_t)W * H, sizeof(real));
    st->n_contrib = (int *)calloc((size_t)W * H, sizeof(int));
    st->pix_ambig = (uint8_t *)calloc((size_t)W * H, 1);
    st->amb_policy = g_amb_policy;
    const int policy = st->amb_policy;

    const real *V = sc->viewmatrix, *Mx = sc->projmatrix;
    const real fx = (real)W / (2 * sc->tanfovx), fy = (real)H / (2 * sc->tanfovy);

    /* ---- A.1 per-Gaussian preprocess */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        const real px = sc->means3D[3 * i], py = sc->means3D[3 * i + 1], pz = sc->means3D[3 * i + 2];
        st->radii[i] = 0;
        real tvx = dot3p(V[0], px, V[4], py, V[8], pz, V[12]);
        real tvy = dot3p(V[1], px, V[5], py, V[9], pz, V[13]);
        real tvz = dot3p(V[2], px, V[6], py, V[10], pz, V[14]);
        if (R_FABS(tvz - NEAR_Z) < (real)2e-6) st->gauss_ambig[i] = 1;
        if (tvz <= NEAR_Z) continue;
        real hx = dot3p(Mx[0], px, Mx[4], py, Mx[8], pz, Mx[12]);
        real hy = dot3p(Mx[1], px, Mx[5], py, Mx[9], pz, Mx[13]);
        real hw = dot3p(Mx[3], px, Mx[7], py, Mx[11], pz, Mx[15]);
        real pw = 1 / (hw + (real)0.0000001);
        real ndcx = hx * pw, ndcy = hy * pw;

        real cov[6];
        if (sc->cov3D_precomp) memcpy(cov, sc->cov3D_precomp + 6 * (size_t)i, sizeof(cov));
        else cov3d_from_scale_rot(sc->scales + 3 * (size_t)i, sc->scale_modifier, sc->rotations + 4 * (size_t)i, cov);
        memcpy(st->cov3D + 6 * (size_t)i, cov, sizeof(cov));

        /* EWA projection */
        real limx = (real)1.3 * sc->tanfovx, limy = (real)1.3 * sc->tanfovy;
        real txtz = tvx / tvz, tytz = tvy / tvz;
        real tx = fmin(limx, fmax(-limx, txtz)) * tvz;
        real ty = fmin(limy, fmax(-limy, tytz)) * tvz;
        real tz = tvz;
        real J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
        real J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
        /* rows of Wrot: Wrot[r][c] = V[4*c + r] */
        real T0[3], T1[3];
        for (int c = 0; c < 3; c++) {
            T0[c] = J00 * V[4 * c + 0] + J02 * V[4 * c + 2];
            T1[c] = J11 * V[4 * c + 1] + J12 * V[4 * c + 2];
        }
        real S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
        real ST0[3], ST1[3];
        for (int r = 0; r < 3; r++) {
            ST0[r] = S[r][0] * T0[0] + S[r][1] * T0[1] + S[r][2] * T0[2];
            ST1[r] = S[r][0] * T1[0] + S[r][1] * T1[1] + S[r][2] * T1[2];
        }
        real a = T0[0] * ST0[0] + T0[1] * ST0[1] + T0[2] * ST0[2];
        real b = T0[0] * ST1[0] + T0[1] * ST1[1] + T0[2] * ST1[2];
        real c2 = T1[0] * ST1[0] + T1[1] * ST1[1] + T1[2] * ST1[2];

        real det0 = a * c2 - b * b;
        a += DILATE; c2 += DILATE;
        real det = a * c2 - b * b;
        real hconv = 1;
        if (sc->antialiasing) hconv = R_SQRT(fmax((real)0.000025, det0 / det));
        if (det == 0) continue;
        real dinv = 1 / det;
        real cA = c2 * dinv, cB = -b * dinv, cC = a * dinv;
        real mid = (real)0.5 * (a + c2);
        real disc = R_SQRT(fmax((real)0.1, mid * mid - det));
        real l1 = mid + disc, l2 = mid - disc;
        real rraw = 3 * R_SQRT(fmax(l1, l2));
        real rad = R_CEIL(rraw);
        if (near_int(rraw, (real)2e-5)) st->gauss_ambig[i] = 1;
        real pix = ((ndcx + 1) * W - 1) * (real)0.5;
        real piy = ((ndcy + 1) * H - 1) * (real)0.5;
        real q0 = (pix - rad) / TILE, q1 = (piy - rad) / TILE;
        real q2 = (pix + rad + TILE - 1) / TILE, q3 = (piy + rad + TILE - 1) / TILE;
        if (near_int(q0, (real)1e-5) || near_int(q1, (real)1e-5) || near_int(q2, (real)1e-5) || near_int(q3, (real)1e-5))
            st->gauss_ambig[i] = 1;
        /* truncating float->int conversion, clamped to the grid (clamp first in float so that
         * far-off-screen values cannot overflow the conversion) */
        real big = (real)(1 << 20);
        int minx = (int)fmin(big, fmax(-big, q0)), miny = (int)fmin(big, fmax(-big, q1));
        int maxx = (int)fmin(big, fmax(-big, q2)), maxy = (int)fmin(big, fmax(-big, q3));
        minx = minx < 0 ? 0 : (minx > gx ? gx : minx); maxx = maxx < 0 ? 0 : (maxx > gx ? gx : maxx);
        miny = miny < 0 ? 0 : (miny > gy ? gy : miny); maxy = maxy < 0 ? 0 : (maxy > gy ? gy : maxy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;

        real rgb[3];
        if (sc->colors_precomp) {
            for (int c = 0; c < 3; c++) rgb[c] = sc->colors_precomp[3 * (size_t)i + c];
        } else {
            real dx = px - sc->campos[0], dy = py - sc->campos[1], dz = pz - sc->campos[2];
            real inv = 1 / R_SQRT(dx * dx + dy * dy + dz * dz);
            real dir[3] = {dx * inv, dy * inv, dz * inv};
            sh_to_rgb(sc->D, sc->M, sc->shs + (size_t)i * sc->M * 3, dir, rgb);
            for (int c = 0; c < 3; c++) {
                rgb[c] += (real)0.5;
                st->clamped[3 * (size_t)i + c] = rgb[c] < 0;
                if (rgb[c] < 0) rgb[c] = 0;
            }
        }
        st->depth[i] = tvz;
        st->radii[i] = (int)rad;
        st->xy[2 * (size_t)i] = pix; st->xy[2 * (size_t)i + 1] = piy;
        st->conic_op[4 * (size_t)i] = cA; st->conic_op[4 * (size_t)i + 1] = cB;
        st->conic_op[4 * (size_t)i + 2] = cC; st->conic_op[4 * (size_t)i + 3] = sc->opacities[i] * hconv;
        for (int c = 0; c < 3; c++) st->rgb[3 * (size_t)i + c] = rgb[c];
        st->rect[4 * (size_t)i] = minx; st->rect[4 * (size_t)i + 1] = miny;
        st->rect[4 * (size_t)i + 2] = maxx; st->rect[4 * (size_t)i + 3] = maxy;
    }
    if (out_radii) memcpy(out_radii, st->radii, sizeof(int) * (size_t)P);

    /* ---- A.2 binning: one instance per (Gaussian, tile); order by (tile, depth bits, id) */
    long N = 0;
    for (int i = 0; i < P; i++)
        if (st->radii[i] > 0)
            N += (long)(st->rect[4 * (size_t)i + 2] - st->rect[4 * (size_t)i]) * (st->rect[4 * (size_t)i + 3] - st->rect[4 * (size_t)i + 1]);
    st->N = N;
    Inst *inst = (Inst *)malloc(sizeof(Inst) * (size_t)(N > 0 ? N : 1));
    long k = 0;
    for (int i = 0; i < P; i++) {
        if (st->radii[i] <= 0) continue;
        float df = (float)st->depth[i];
        uint32_t bits; memcpy(&bits, &df, 4);
        for (int ty = st->rect[4 * (size_t)i + 1]; ty < st->rect[4 * (size_t)i + 3]; ty++)
            for (int tx = st->rect[4 * (size_t)i]; tx < st->rect[4 * (size_t)i + 2]; tx++) {
                inst[k].tile = (uint32_t)(ty * gx + tx); inst[k].dbits = bits; inst[k].id = (uint32_t)i; k++;
            }
    }
    qsort(inst, (size_t)N, sizeof(Inst), inst_cmp);
    st->point_list = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(N > 0 ? N : 1));
    for (long s = 0; s < N; s++) {
        st->point_list[s] = inst[s].id;
        if (s == 0 || inst[s].tile != inst[s - 1].tile) st->ranges[2 * (size_t)inst[s].tile] = s;
        if (s == N - 1 || inst[s].tile != inst[s + 1].tile) st->ranges[2 * (size_t)inst[s].tile + 1] = s + 1;
    }
    free(inst);

    /* ---- A.3 per-tile front-to-back compositing */
    double inter = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : inter)
    for (long t = 0; t < T; t++) {
        int tx0 = (int)(t % gx) * TILE, ty0 = (int)(t / gx) * TILE;
        long r0 = st->ranges[2 * t], r1 = st->ranges[2 * t + 1];
        for (int yy = ty0; yy < ty0 + TILE && yy < H; yy++)
            for (int xx = tx0; xx < tx0 + TILE && xx < W; xx++) {
                real Tr = 1, C[3] = {0, 0, 0}, Dp = 0;
                int contributor = 0, last = 0;
                uint8_t amb = 0;
                for (long s = r0; s < r1; s++) {
                    contributor++;
                    uint32_t g = st->point_list[s];
                    real dx = st->xy[2 * (size_t)g] - (real)xx, dy = st->xy[2 * (size_t)g + 1] - (real)yy;
                    const real *co = st->conic_op + 4 * (size_t)g;
                    real power = pair_power(co[0], co[1], co[2], dx, dy);
                    inter += 1;
                    /* bit 0: the decision is within the rounding of exp() for IDENTICAL rasterizer inputs.
                     * bit 1: it is within the rounding of the INPUTS: the three terms of `power` cancel for elongated
                     * splats far from the pixel (|term| ~ 10^3 for |power| ~ 5), so a 1-ulp difference in the conic (two
                     * float32 implementations of the mesh->Gaussian stage in front) moves `power` by ~eps * cond.  Only
                     * comparisons whose inputs went through different float32 code in front use bit 1. */
                    real cond = (real)0.5 * (R_FABS(co[0] * dx * dx) + R_FABS(co[2] * dy * dy)) + R_FABS(co[1] * dx * dy);
                    real in_band = (real)8 * (real)6e-8 * cond;
                    if (R_FABS(power) < (real)1e-6) {
                        /* the pixel centre sits on the splat's mean: `power > 0 -> skip` is decided by the last ulp of the
                         * rasterizer inputs, and what is skipped or not is a splat at its FULL opacity.  gauss_ambig bit 1 for
                         * this Gaussian (marked here, before the skip, so that a skipped pair is marked too). */
                        amb |= 1;
#pragma omp atomic
                        st->gauss_ambig[g] |= 2;
                    }
                    if (R_FABS(power) < (real)1e-6 + in_band) amb |= 2;
                    if (skip_by_power(policy, power)) continue;
                    real ex = R_EXP(power);
                    real araw = co[3] * ex;
                    real alpha = araw < ALPHA_MAX ? araw : ALPHA_MAX;
                    if (R_FABS(araw * 255 - 1) < (real)4e-5) amb |= 1;
                    if (R_FABS(araw * 255 - 1) < (real)4e-5 + in_band) amb |= 2;
                    if (R_FABS(araw * 255 - 1) < (real)4e-5 || R_FABS(power) < (real)1e-6) {
                        /* gauss_ambig bit 1: this Gaussian has a (pixel, Gaussian) skip decision within exp() rounding:
                         * an independent float32 implementation may include or drop that one pixel's term of its gradient */
#pragma omp atomic
                        st->gauss_ambig[g] |= 2;
                    }
                    if (skip_by_alpha(policy, araw, alpha)) continue;
                    real testT = Tr * (1 - alpha);
                    if (R_FABS(testT - T_MIN) < (real)1e-8) {
                        amb |= 3;
#pragma omp atomic
                        st->gauss_ambig[g] |= 2;
                    }
                    if (testT < T_MIN) break;
                    real w = alpha * Tr;
                    for (int c = 0; c < 3; c++) C[c] += st->rgb[3 * (size_t)g + c] * w;
                    Dp += (1 / st->depth[g]) * w;
                    Tr = testT;
                    last = contributor;
                }
                if (amb & 1) {
                    /* gauss_ambig bit 2: composited in a pixel that holds a within-rounding decision (in front of the
                     * pair: the colour behind changes; behind it: the transmittance does) */
                    for (long s = r0; s < r0 + last; s++) {
                        uint32_t g = st->point_list[s];
                        real dx = st->xy[2 * (size_t)g] - (real)xx, dy = st->xy[2 * (size_t)g + 1] - (real)yy;
                        const real *co = st->conic_op + 4 * (size_t)g;
                        real power = pair_power(co[0], co[1], co[2], dx, dy);
                        if (power > (real)1e-6) continue;
                        if (co[3] * R_EXP(power) * 255 < 1 - (real)4e-5) continue;
#pragma omp atomic
                        st->gauss_ambig[g] |= 4;
                    }
                }
                size_t pid = (size_t)yy * W + xx;
                st->final_T[pid] = Tr; st->n_contrib[pid] = last; st->pix_ambig[pid] = amb;
                for (int c = 0; c < 3; c++) out_color[(size_t)c * H * W + pid] = C[c] + Tr * sc->bg[c];
                if (out_invdepth) out_invdepth[pid] = Dp;
            }
    }
    st->interactions = inter;
    return st;
}

/* ----------------------------------------------------------------- backward */
/* Accumulators of the per-(Gaussian, tile) gradient sums.  Default: double, so the float32 build's gradients carry only
 * the per-term rounding (a clean reference value).  -DORACLE_FLOAT_ACCUM: `real`, i.e. float32 sums in a fixed order --
 * what an implementation with float atomics (the reference's CUDA kernels, the HIP kernels) can be expected to reach
 * on rows whose sums cancel; tests use |this build - float64 build| as the float32 noise scale of a row. */
#ifdef ORACLE_FLOAT_ACCUM
typedef real acc_t;
#else
typedef double acc_t;
#endif
/* per-instance gradient slots */
enum { G_MX = 0, G_MY, G_CA, G_CB, G_CC, G_OP, G_R, G_G, G_B, G_ID, G_NUM };

/* -DORACLE_FMA_BACKWARD (libgs_oracle_f32fma): the SAME float32 backward with the compiler free to contract a*b+c into one
 * FMA -- a second rounding realisation of the same formulas, which is what every GPU build of them is (nvcc and hipcc both
 * contract by default).  The forward pass and every decision keep the fixed operation order (this pragma covers
 * or_backward only; the decision helpers use explicit fma chains).  Why it exists: the published cov2D-inverse gradient
 * contains (denom - a*c) with denom = a*c - b*b; for a long splat that difference is rounding noise of size ulp(a*c)
 * against b*b, and the gradient of the rotation moves by up to 2e-3 of the tensor's largest entry between two float32
 * realisations of the very same expression (fuzz seed 5337, one Gaussian: 1.9e-3).  tests/_util.py
 * takes the float32 oracle's own error on a row as the maximum over its realisations (f32, f32acc, f32fma). */
#ifdef ORACLE_FMA_BACKWARD
#pragma GCC push_options
#pragma GCC target("fma")
#pragma GCC optimize("fp-contract=fast")
#endif
void or_backward(const OrScene *sc, const OrState *st, const real *dL_dpix /*[3,H,W]*/,
                 const real *dL_dinvdepth_pix /*[H,W] or NULL*/,
                 real *dL_dmeans3D, real *dL_dmeans2D /*[P,3]*/, real *dL_dsh /*[P,M,3]*/,
                 real *dL_dcolors /*[P,3]*/, real *dL_dopacity /*[P]*/, real *dL_dscales /*[P,3]*/,
                 real *dL_drotations /*[P,4]*/, real *dL_dcov3D /*[P,6]*/,
                 real *dL_dconic_out /*[P,4] optional debug*/,
                 real *dL_dcolor_sh /*[P,3] optional, SH path: the clamp-masked dL/dcolour (zero-initialised by the caller) --
                                      the per-view factor of the factorised SH gradient, dL/dsh[k][c] = Y_k(dir) * this[c] */)
{
    set_threads(sc->nthreads);
    const int P = st->P, W = st->W, H = st->H, gx = st->gx;
    const long T = (long)gx * st->gy, N = st->N;
    acc_t *ginst = (acc_t *)calloc((size_t)(N > 0 ? N : 1) * G_NUM, sizeof(acc_t));

    /* ---- A.4 per-tile back-to-front */
#pragma omp parallel for schedule(dynamic, 1)
    for (long t = 0; t < T; t++) {
        int tx0 = (int)(t % gx) * TILE, ty0 = (int)(t / gx) * TILE;
        long r0 = st->ranges[2 * t], r1 = st->ranges[2 * t + 1];
        (void)r1;
        for (int yy = ty0; yy < ty0 + TILE && yy < H; yy++)
            for (int xx = tx0; xx < tx0 + TILE && xx < W; xx++) {
                size_t pid = (size_t)yy * W + xx;
                const real Tfinal = st->final_T[pid];
                real Tr = Tfinal;
                const int last = st->n_contrib[pid];
                real dpix[3];
                for (int c = 0; c < 3; c++) dpix[c] = dL_dpix[(size_t)c * H * W + pid];
                real dinvd = dL_dinvdepth_pix ? dL_dinvdepth_pix[pid] : 0;
                real accum[3] = {0, 0, 0}, accum_d = 0;
                real last_alpha = 0, last_col[3] = {0, 0, 0}, last_invd = 0;
                real bgdot = sc->bg[0] * dpix[0] + sc->bg[1] * dpix[1] + sc->bg[2] * dpix[2];
                for (int j = last - 1; j >= 0; j--) {
                    long s = r0 + j;
                    uint32_t g = st->point_list[s];
                    real dx = st->xy[2 * (size_t)g] - (real)xx, dy = st->xy[2 * (size_t)g + 1] - (real)yy;
                    const real *co = st->conic_op + 4 * (size_t)g;
                    real power = pair_power(co[0], co[1], co[2], dx, dy);
                    if (skip_by_power(st->amb_policy, power)) continue;
                    real Gv = R_EXP(power);
                    real araw = co[3] * Gv;
                    real alpha = araw < ALPHA_MAX ? araw : ALPHA_MAX;
                    if (skip_by_alpha(st->amb_policy, araw, alpha)) continue;
                    Tr = Tr / (1 - alpha);
                    real w = alpha * Tr;
                    acc_t *gi = ginst + (size_t)s * G_NUM;
                    real dL_dalpha = 0;
                    for (int c = 0; c < 3; c++) {
                        real col = st->rgb[3 * (size_t)g + c];
                        accum[c] = last_alpha * last_col[c] + (1 - last_alpha) * accum[c];
                        last_col[c] = col;
                        dL_dalpha += (col - accum[c]) * dpix[c];
                        gi[G_R + c] += (acc_t)(w * dpix[c]);
                    }
                    real invd = 1 / st->depth[g];
                    accum_d = last_alpha * last_invd + (1 - last_alpha) * accum_d;
                    last_invd = invd;
                    dL_dalpha += (invd - accum_d) * dinvd;
                    gi[G_ID] += (acc_t)(w * dinvd);
                    dL_dalpha *= Tr;
                    last_alpha = alpha;
                    dL_dalpha += (-Tfinal / (1 - alpha)) * bgdot;
                    /* alpha = min(0.99, op*G) is straight-through (A.5 i) */
                    real dL_dG = co[3] * dL_dalpha;
                    real gdx = Gv * dx, gdy = Gv * dy;
                    real dG_ddx = -gdx * co[0] - gdy * co[1];
                    real dG_ddy = -gdy * co[2] - gdx * co[1];
                    gi[G_MX] += (acc_t)(dL_dG * dG_ddx * (real)0.5 * W);
                    gi[G_MY] += (acc_t)(dL_dG * dG_ddy * (real)0.5 * H);
                    gi[G_CA] += (acc_t)((real)-0.5 * gdx * dx * dL_dG);
                    gi[G_CB] += (acc_t)((real)-0.5 * gdx * dy * dL_dG);
                    gi[G_CC] += (acc_t)((real)-0.5 * gdy * dy * dL_dG);
                    gi[G_OP] += (acc_t)(Gv * dL_dalpha);
                }
            }
    }

    /* deterministic per-Gaussian reduction in sorted-instance order */
    size_t Pn = P > 0 ? (size_t)P : 1;
    acc_t *gacc = (acc_t *)calloc(Pn * G_NUM, sizeof(acc_t));
    for (long s = 0; s < N; s++) {
        uint32_t g = st->point_list[s];
        for (int q = 0; q < G_NUM; q++) gacc[(size_t)g * G_NUM + q] += ginst[(size_t)s * G_NUM + q];
    }
    free(ginst);

    const real *V = sc->viewmatrix, *Mx = sc->projmatrix;
    const real fx = (real)W / (2 * sc->tanfovx), fy = (real)H / (2 * sc->tanfovy);
    const int M = sc->M;

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        real dmean[3] = {0, 0, 0};
        real dcov3[6] = {0, 0, 0, 0, 0, 0};
        const acc_t *ga = gacc + (size_t)i * G_NUM;
        if (dL_dmeans2D) { dL_dmeans2D[3 * (size_t)i] = (real)ga[G_MX]; dL_dmeans2D[3 * (size_t)i + 1] = (real)ga[G_MY]; dL_dmeans2D[3 * (size_t)i + 2] = 0; }
        if (dL_dconic_out) { dL_dconic_out[4 * (size_t)i] = (real)ga[G_CA]; dL_dconic_out[4 * (size_t)i + 1] = (real)ga[G_CB]; dL_dconic_out[4 * (size_t)i + 2] = 0; dL_dconic_out[4 * (size_t)i + 3] = (real)ga[G_CC]; }
        real dop = (real)ga[G_OP];
        if (dL_dcolors) for (int c = 0; c < 3; c++) dL_dcolors[3 * (size_t)i + c] = sc->colors_precomp ? (real)ga[G_R + c] : 0;
        if (dL_dsh) memset(dL_dsh + (size_t)i * M * 3, 0, sizeof(real) * M * 3);
        if (dL_dscales) for (int c = 0; c < 3; c++) dL_dscales[3 * (size_t)i + c] = 0;
        if (dL_drotations) for (int c = 0; c < 4; c++) dL_drotations[4 * (size_t)i + c] = 0;

        if (st->radii[i] > 0) {
            const real px = sc->means3D[3 * i], py = sc->means3D[3 * i + 1], pz = sc->means3D[3 * i + 2];
            const real *cov = st->cov3D + 6 * (size_t)i;
            /* --- EWA backward: recompute forward quantities */
            real tvx = dot3p(V[0], px, V[4], py, V[8], pz, V[12]);
            real tvy = dot3p(V[1], px, V[5], py, V[9], pz, V[13]);
            real tvz = dot3p(V[2], px, V[6], py, V[10], pz, V[14]);
            real limx = (real)1.3 * sc->tanfovx, limy = (real)1.3 * sc->tanfovy;
            real txtz = tvx / tvz, tytz = tvy / tvz;
            real tx = fmin(limx, fmax(-limx, txtz)) * tvz;
            real ty = fmin(limy, fmax(-limy, tytz)) * tvz;
            real tz = tvz;
            real xmul = (txtz < -limx || txtz > limx) ? 0 : 1;
            real ymul = (tytz < -limy || tytz > limy) ? 0 : 1;
            real J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
            real J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
            real T0[3], T1[3];
            for (int c = 0; c < 3; c++) {
                T0[c] = J00 * V[4 * c + 0] + J02 * V[4 * c + 2];
                T1[c] = J11 * V[4 * c + 1] + J12 * V[4 * c + 2];
            }
            real S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
            real ST0[3], ST1[3];
            for (int r = 0; r < 3; r++) {
                ST0[r] = S[r][0] * T0[0] + S[r][1] * T0[1] + S[r][2] * T0[2];
                ST1[r] = S[r][0] * T1[0] + S[r][1] * T1[1] + S[r][2] * T1[2];
            }
            real a0 = T0[0] * ST0[0] + T0[1] * ST0[1] + T0[2] * ST0[2];
            real b = T0[0] * ST1[0] + T0[1] * ST1[1] + T0[2] * ST1[2];
            real c0 = T1[0] * ST1[0] + T1[1] * ST1[1] + T1[2] * ST1[2];
            real a = a0 + DILATE, c = c0 + DILATE;
            real det = a * c - b * b;
            /* conic (A',B',C') = (c,-b,a)/det ; stored dL/dB' is HALF the true one (A.4) */
            real gA = (real)ga[G_CA], gB = (real)ga[G_CB], gC = (real)ga[G_CC];
            real d2inv = 1 / (det * det + (real)0.0000001);   /* A.5 vi */
            real dL_da = d2inv * (-c * c * gA + 2 * b * c * gB + (det - a * c) * gC);
            real dL_dc = d2inv * (-a * a * gC + 2 * a * b * gB + (det - a * c) * gA);
            real dL_db = d2inv * 2 * (b * c * gA - (det + 2 * b * b) * gB + a * b * gC);
            if (sc->antialiasing) {
                /* opacity' = opacity * h,  h = sqrt(max(0.000025, det0/det)) */
                real det0 = a0 * c0 - b * b;
                real ratio = det0 / det;
                real h = R_SQRT(fmax((real)0.000025, ratio));
                real dL_dh = dop * sc->opacities[i];
                dop = dop * h;
                real dL_dr = ratio <= (real)0.000025 ? 0 : dL_dh / (2 * h);
                dL_da += dL_dr * (c0 / det - det0 * c / (det * det));
                dL_dc += dL_dr * (a0 / det - det0 * a / (det * det));
                dL_db += dL_dr * (-2 * b / det + det0 * 2 * b / (det * det));
            }
            /* cov2D = T Sigma T^T */
            dcov3[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
            dcov3[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
            dcov3[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
            dcov3[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
            dcov3[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
            dcov3[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
            real dT0[3], dT1[3];
            for (int r = 0; r < 3; r++) {
                dT0[r] = 2 * ST0[r] * dL_da + ST1[r] * dL_db;
                dT1[r] = 2 * ST1[r] * dL_dc + ST0[r] * dL_db;
            }
            real dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
            for (int cc = 0; cc < 3; cc++) {
                dJ00 += dT0[cc] * V[4 * cc + 0]; dJ02 += dT0[cc] * V[4 * cc + 2];
                dJ11 += dT1[cc] * V[4 * cc + 1]; dJ12 += dT1[cc] * V[4 * cc + 2];
            }
            real tzi = 1 / tz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
            real dtx = xmul * -fx * tz2 * dJ02;
            real dty = ymul * -fy * tz2 * dJ12;
            real dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * tx) * tz3 * dJ02 + (2 * fy * ty) * tz3 * dJ12;
            /* inverse-depth output: invd = 1 / t.z */
            dtz -= (real)ga[G_ID] / (tz * tz);
            for (int r = 0; r < 3; r++)
                dmean[r] += V[4 * r + 0] * dtx + V[4 * r + 1] * dty + V[4 * r + 2] * dtz;

            /* --- pixel mean -> world mean through the full projection (A.5 vii, x) */
            real hx = dot3p(Mx[0], px, Mx[4], py, Mx[8], pz, Mx[12]);
            real hy = dot3p(Mx[1], px, Mx[5], py, Mx[9], pz, Mx[13]);
            real hw = dot3p(Mx[3], px, Mx[7], py, Mx[11], pz, Mx[15]);
            real mw = 1 / (hw + (real)0.0000001);
            real gmx = (real)ga[G_MX], gmy = (real)ga[G_MY];
            real mul1 = hx * mw * mw, mul2 = hy * mw * mw;
            for (int r = 0; r < 3; r++)
                dmean[r] += (Mx[4 * r + 0] * mw - Mx[4 * r + 3] * mul1) * gmx + (Mx[4 * r + 1] * mw - Mx[4 * r + 3] * mul2) * gmy;

            /* --- colour: SH backward (A.5 iii) */
            if (!sc->colors_precomp) {
                real dx = px - sc->campos[0], dy = py - sc->campos[1], dz = pz - sc->campos[2];
                real len = R_SQRT(dx * dx + dy * dy + dz * dz), inv = 1 / len;
                real x = dx * inv, y = dy * inv, z = dz * inv;
                const real *sh = sc->shs + (size_t)i * M * 3;
                real dRGB[3];
                for (int cc = 0; cc < 3; cc++) dRGB[cc] = st->clamped[3 * (size_t)i + cc] ? 0 : (real)ga[G_R + cc];
                if (dL_dcolor_sh) for (int cc = 0; cc < 3; cc++) dL_dcolor_sh[3 * (size_t)i + cc] = dRGB[cc];
                real basis[16], bdx[16], bdy[16], bdz[16];
                for (int q = 0; q < 16; q++) basis[q] = bdx[q] = bdy[q] = bdz[q] = 0;
                basis[0] = SH_C0;
                int nb = 1;
                if (sc->D > 0) {
                    nb = 4;
                    basis[1] = -SH_C1 * y; bdy[1] = -SH_C1;
                    basis[2] = SH_C1 * z; bdz[2] = SH_C1;
                    basis[3] = -SH_C1 * x; bdx[3] = -SH_C1;
                    if (sc->D > 1) {
                        nb = 9;
                        real xx = x * x, yy = y * y, zz = z * z;
                        basis[4] = SH_C2[0] * x * y; bdx[4] = SH_C2[0] * y; bdy[4] = SH_C2[0] * x;
                        basis[5] = SH_C2[1] * y * z; bdy[5] = SH_C2[1] * z; bdz[5] = SH_C2[1] * y;
                        basis[6] = SH_C2[2] * (2 * zz - xx - yy); bdx[6] = -2 * SH_C2[2] * x; bdy[6] = -2 * SH_C2[2] * y; bdz[6] = 4 * SH_C2[2] * z;
                        basis[7] = SH_C2[3] * x * z; bdx[7] = SH_C2[3] * z; bdz[7] = SH_C2[3] * x;
                        basis[8] = SH_C2[4] * (xx - yy); bdx[8] = 2 * SH_C2[4] * x; bdy[8] = -2 * SH_C2[4] * y;
                        if (sc->D > 2) {
                            nb = 16;
                            basis[9] = SH_C3[0] * y * (3 * xx - yy); bdx[9] = SH_C3[0] * 6 * x * y; bdy[9] = SH_C3[0] * (3 * xx - 3 * yy);
                            basis[10] = SH_C3[1] * x * y * z; bdx[10] = SH_C3[1] * y * z; bdy[10] = SH_C3[1] * x * z; bdz[10] = SH_C3[1] * x * y;
                            basis[11] = SH_C3[2] * y * (4 * zz - xx - yy); bdx[11] = SH_C3[2] * (-2 * x * y); bdy[11] = SH_C3[2] * (4 * zz - xx - 3 * yy); bdz[11] = SH_C3[2] * 8 * y * z;
                            basis[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy); bdx[12] = SH_C3[3] * (-6 * x * z); bdy[12] = SH_C3[3] * (-6 * y * z); bdz[12] = SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
                            basis[13] = SH_C3[4] * x * (4 * zz - xx - yy); bdx[13] = SH_C3[4] * (4 * zz - 3 * xx - yy); bdy[13] = SH_C3[4] * (-2 * x * y); bdz[13] = SH_C3[4] * 8 * x * z;
                            basis[14] = SH_C3[5] * z * (xx - yy); bdx[14] = SH_C3[5] * 2 * x * z; bdy[14] = SH_C3[5] * (-2 * y * z); bdz[14] = SH_C3[5] * (xx - yy);
                            basis[15] = SH_C3[6] * x * (xx - 3 * yy); bdx[15] = SH_C3[6] * (3 * xx - 3 * yy); bdy[15] = SH_C3[6] * (-6 * x * y);
                        }
                    }
                }
                real ddir[3] = {0, 0, 0};
                for (int q = 0; q < nb; q++)
                    for (int cc = 0; cc < 3; cc++) {
                        if (dL_dsh) dL_dsh[((size_t)i * M + q) * 3 + cc] = basis[q] * dRGB[cc];
                        real w = sh[q * 3 + cc] * dRGB[cc];
                        ddir[0] += bdx[q] * w; ddir[1] += bdy[q] * w; ddir[2] += bdz[q] * w;
                    }
                /* dir = v/|v| */
                real dd = x * ddir[0] + y * ddir[1] + z * ddir[2];
                dmean[0] += (ddir[0] - x * dd) * inv;
                dmean[1] += (ddir[1] - y * dd) * inv;
                dmean[2] += (ddir[2] - z * dd) * inv;
            }

            /* --- cov3D -> scale, rotation (A.5 viii, ix) */
            if (!sc->cov3D_precomp && dL_dscales && dL_drotations) {
                const real *q = sc->rotations + 4 * (size_t)i;
                const real *sin_ = sc->scales + 3 * (size_t)i;
                real mod = sc->scale_modifier;
                real s[3] = {mod * sin_[0], mod * sin_[1], mod * sin_[2]};
                real r = q[0], x = q[1], y = q[2], z = q[3];
                real R[3][3] = {
                    {1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                    {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                    {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
                /* full symmetric gradient matrix */
                real Gs[3][3] = {{dcov3[0], (real)0.5 * dcov3[1], (real)0.5 * dcov3[2]},
                                 {(real)0.5 * dcov3[1], dcov3[3], (real)0.5 * dcov3[4]},
                                 {(real)0.5 * dcov3[2], (real)0.5 * dcov3[4], dcov3[5]}};
                /* Sigma = L L^T, L = R diag(s): dL/dL = 2 Gs L */
                real dLm[3][3];
                for (int a1 = 0; a1 < 3; a1++)
                    for (int j = 0; j < 3; j++) {
                        real acc = 0;
                        for (int k2 = 0; k2 < 3; k2++) acc += Gs[a1][k2] * R[k2][j] * s[j];
                        dLm[a1][j] = 2 * acc;
                    }
                real dR[3][3];
                for (int j = 0; j < 3; j++) {
                    real acc = 0;
                    for (int a1 = 0; a1 < 3; a1++) { acc += R[a1][j] * dLm[a1][j]; dR[a1][j] = dLm[a1][j] * s[j]; }
                    dL_dscales[3 * (size_t)i + j] = acc * mod;
                }
                real dq0 = 2 * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
                real dq1 = 2 * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2 * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2 * x * dR[2][2]);
                real dq2 = 2 * (-2 * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2 * y * dR[2][2]);
                real dq3 = 2 * (-2 * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2 * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
                dL_drotations[4 * (size_t)i] = dq0; dL_drotations[4 * (size_t)i + 1] = dq1;
                dL_drotations[4 * (size_t)i + 2] = dq2; dL_drotations[4 * (size_t)i + 3] = dq3;
            }
        }
        if (dL_dopacity) dL_dopacity[i] = dop;
        if (dL_dmeans3D) for (int c = 0; c < 3; c++) dL_dmeans3D[3 * (size_t)i + c] = dmean[c];
        if (dL_dcov3D) for (int c = 0; c < 6; c++) dL_dcov3D[6 * (size_t)i + c] = sc->cov3D_precomp ? dcov3[c] : 0;
    }
    free(gacc);
}

/* accessors for the python wrapper */
long or_state_N(const OrState *st) { return st->N; }
double or_state_interactions(const OrState *st) { return st->interactions; }
#ifdef ORACLE_FMA_BACKWARD
#pragma GCC pop_options
#endif

void or_state_copy(const OrState *st, real *depth, real *xy, real *conic_op, real *rgb, real *cov3D,
                   uint8_t *clamped, int *rect, real *final_T, int *n_contrib, uint8_t *gauss_ambig,
                   uint8_t *pix_ambig, uint32_t *point_list, long *ranges)
{
    size_t P = (size_t)st->P, HW = (size_t)st->W * st->H, T = (size_t)st->gx * st->gy;
    if (depth) memcpy(depth, st->depth, P * sizeof(real));
    if (xy) memcpy(xy, st->xy, P * 2 * sizeof(real));
    if (conic_op) memcpy(conic_op, st->conic_op, P * 4 * sizeof(real));
    if (rgb) memcpy(rgb, st->rgb, P * 3 * sizeof(real));
    if (cov3D) memcpy(cov3D, st->cov3D, P * 6 * sizeof(real));
    if (clamped) memcpy(clamped, st->clamped, P * 3);
    if (rect) memcpy(rect, st->rect, P * 4 * sizeof(int));
    if (final_T) memcpy(final_T, st->final_T, HW * sizeof(real));
    if (n_contrib) memcpy(n_contrib, st->n_contrib, HW * sizeof(int));
    if (gauss_ambig) memcpy(gauss_ambig, st->gauss_ambig, P);
    if (pix_ambig) memcpy(pix_ambig, st->pix_ambig, HW);
    if (point_list) memcpy(point_list, st->point_list, (size_t)st->N * sizeof(uint32_t));
    if (ranges) memcpy(ranges, st->ranges, T * 2 * sizeof(long));
}
int or_real_size(void) { return (int)sizeof(real); }
int or_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
