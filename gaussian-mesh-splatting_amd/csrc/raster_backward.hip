// raster_backward.hip -- backward pass of the gfx950 Gaussian rasterizer: the per-Gaussian stage and the C ABI entry.
//
//   K7  blend_bwd        (blend.hip) one block per (tile, segment) unit, back-to-front from the segment's end state; per
//                        (wave, splat) ten partial sums -- five moments of q = dL/dG * G, sum q, three colour weights, the
//                        inverse-depth weight -- leave the wave through a transposing DPP / permlane reduction as ONE
//                        atomic instruction into the splat's 64-byte gradient record.
//   K8+K9 preprocess_bwd 1 thread / Gaussian: reads the 64-byte record (and clears it), maps the moments to the gradients
//                        of (pixel mean, conic, opacity) with per-Gaussian constants, then conic -> cov2D -> (cov3D, mean)
//                        through the EWA Jacobian, pixel mean -> world mean through the projection, inverse depth, SH
//                        backward (coefficients staged in, gradients staged out through the same LDS rows, so both the
//                        192-B read and the 192-B write per Gaussian are coalesced float4 streams), cov3D -> scale /
//                        quaternion.
//
// Gradient conventions: SURVEY.md appendix A.4/A.5 (straight-through alpha clamp, constant
// skip tests, clamp masks, 1/(det^2+1e-7), NDC-scaled mean2D gradient).  Two forms are restated from the published
// algorithm and cannot be checked against the CUDA binary (absent): the antialiasing backward (analytic derivative of
// sqrt(det0/det)) and dL/dscale carrying the scale_modifier factor; both are inert in GaMeS training (antialiasing off,
// modifier 1.0).  DESIGN.md section 2 lists them.
#include "gms_blend.h"
#include "gms_mesh.h"
#include "gms_common.h"
#include "gms_project.h"

namespace gms {

// ------------------------------------------------------------------------------------ K8 + K9
constexpr int SH_PITCH_B = 52;

struct PreBwdArgs {
    int P, D, M, W, H;
    const float *means3D, *shs, *shs_rest, *colors, *opac, *scales, *rots, *cov3Dp, *view, *proj, *campos;
    float mod, tanx, tany;
    float scale_grad_mod;   // factor of dL/dscale: `mod` (the derivative of cov3D = R diag(mod s)^2 R^T), or 1 under the upstream-quirk switch
    int aa;
    const int *radii;
    const uint8_t *clamped;
    float *accum;           // [P,16] gradient records filled by blend_bwd
    int rezero;             // clear each record after reading it
    float *dL_dmean2D, *dL_dopacity, *dL_dcolors;
    float *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dsh_rest, *dL_dscales, *dL_drots;
    float *dL_dcolor_sh;    // SH path, factorised mode: clamp-masked dL/dcolour [P,3] INSTEAD of the SH gradient rows
    int campos_row;         // factorised mode: also write the camera centre into row P of dL_dcolor_sh
    GmsMeshArgs mesh;       // MESH instantiation: the frame was rendered straight from this mesh; the thread carries its gradients on through K0
    float *mesh_dvertices, *mesh_dalpha, *mesh_dscale, *mesh_dopacity;
};

// SH backward for one Gaussian.  The coefficient row is read from the lane's LDS row as float4
// (ds_read_b128, conflict-free at the 52-dword pitch), d loss / d sh is written back in place, the
// direction gradient is accumulated.  r[k*3+c] lives in registers (all indices are constants).
// (the arithmetic, on a coefficient row held in registers: r[k*3+c] in, d loss / d sh [k*3+c] out)
template <int DEG>
__device__ __forceinline__ void sh_backward_regs(float *r, float x, float y, float z, const float dRGB[3], float gdir[3])
{
#define GMS_SH_TERM(K, BV, BDX, BDY, BDZ)                                          \
    {                                                                              \
        const float bv = (BV), bx = (BDX), by = (BDY), bz = (BDZ);                 \
        _Pragma("unroll") for (int c = 0; c < 3; c++) {                            \
            const float w = r[(K) * 3 + c] * dRGB[c];                              \
            gdir[0] += bx * w; gdir[1] += by * w; gdir[2] += bz * w;               \
            r[(K) * 3 + c] = bv * dRGB[c];                                         \
        }                                                                          \
    }
    GMS_SH_TERM(0, SH_C0, 0.f, 0.f, 0.f)
    if (DEG > 0) {
        GMS_SH_TERM(1, -SH_C1 * y, 0.f, -SH_C1, 0.f)
        GMS_SH_TERM(2, SH_C1 * z, 0.f, 0.f, SH_C1)
        GMS_SH_TERM(3, -SH_C1 * x, -SH_C1, 0.f, 0.f)
    }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        GMS_SH_TERM(4, SH_C2[0] * x * y, SH_C2[0] * y, SH_C2[0] * x, 0.f)
        GMS_SH_TERM(5, SH_C2[1] * y * z, 0.f, SH_C2[1] * z, SH_C2[1] * y)
        GMS_SH_TERM(6, SH_C2[2] * (2.f * zz - xx - yy), -2.f * SH_C2[2] * x, -2.f * SH_C2[2] * y, 4.f * SH_C2[2] * z)
        GMS_SH_TERM(7, SH_C2[3] * x * z, SH_C2[3] * z, 0.f, SH_C2[3] * x)
        GMS_SH_TERM(8, SH_C2[4] * (xx - yy), 2.f * SH_C2[4] * x, -2.f * SH_C2[4] * y, 0.f)
    }
    if (DEG > 2) {
        const float xx = x * x, yy = y * y, zz = z * z;
        GMS_SH_TERM(9, SH_C3[0] * y * (3.f * xx - yy), SH_C3[0] * 6.f * x * y, SH_C3[0] * (3.f * xx - 3.f * yy), 0.f)
        GMS_SH_TERM(10, SH_C3[1] * x * y * z, SH_C3[1] * y * z, SH_C3[1] * x * z, SH_C3[1] * x * y)
        GMS_SH_TERM(11, SH_C3[2] * y * (4.f * zz - xx - yy), SH_C3[2] * (-2.f * x * y), SH_C3[2] * (4.f * zz - xx - 3.f * yy), SH_C3[2] * 8.f * y * z)
        GMS_SH_TERM(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), SH_C3[3] * (-6.f * x * z), SH_C3[3] * (-6.f * y * z), SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy))
        GMS_SH_TERM(13, SH_C3[4] * x * (4.f * zz - xx - yy), SH_C3[4] * (4.f * zz - 3.f * xx - yy), SH_C3[4] * (-2.f * x * y), SH_C3[4] * 8.f * x * z)
        GMS_SH_TERM(14, SH_C3[5] * z * (xx - yy), SH_C3[5] * 2.f * x * z, SH_C3[5] * (-2.f * y * z), SH_C3[5] * (xx - yy))
        GMS_SH_TERM(15, SH_C3[6] * x * (xx - 3.f * yy), SH_C3[6] * (3.f * xx - 3.f * yy), SH_C3[6] * (-6.f * x * y), 0.f)
    }
#undef GMS_SH_TERM
}

template <int DEG>
__device__ __forceinline__ void sh_backward(float *row_lds, float x, float y, float z, const float dRGB[3], float gdir[3])
{
    constexpr int NQ = ((DEG + 1) * (DEG + 1) * 3 + 3) / 4;
    float r[NQ * 4];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const float4 v = *reinterpret_cast<const float4 *>(row_lds + q * 4);
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
    sh_backward_regs<DEG>(r, x, y, z, dRGB, gdir);
    // coefficients beyond the active degree inside the last chunk get zero gradient
#pragma unroll
    for (int k = (DEG + 1) * (DEG + 1) * 3; k < NQ * 4; k++) r[k] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; q++)
        *reinterpret_cast<float4 *>(row_lds + q * 4) = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
}

template <int NQ>
__device__ __forceinline__ void stage_sh_rows_b(const float *shs, int g0, int rows, int rowq, float *wl, int lane)
{
    const float4 *src = reinterpret_cast<const float4 *>(shs) + (size_t)g0 * rowq;
    for (int idx = lane; idx < rows * NQ; idx += WAVE) {
        const int r = idx / NQ, c = idx - r * NQ;
        *reinterpret_cast<float4 *>(wl + r * 52 + c * 4) = src[(size_t)r * rowq + c];
    }
}

template <bool MESH>
__global__ void __launch_bounds__(BLOCK) preprocess_bwd_kernel(PreBwdArgs a, int pre_bwd_linear)
{
    __shared__ __attribute__((aligned(16))) float sh_lds[4 * WAVE * SH_PITCH_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * BLOCK + tid;
    const bool valid = i < a.P;
    const bool vis = valid && a.radii[i] > 0;
    float *wl = sh_lds + wave * (WAVE * SH_PITCH_B);
    const int rowf = a.M * 3;
    const bool use_sh = a.shs != nullptr;
    const bool vec_ok = use_sh && (rowf % 4 == 0) && (rowf <= 48);
    const int nb = (a.D + 1) * (a.D + 1);
    const int g0 = blockIdx.x * BLOCK + wave * WAVE;
    const int rows = min(WAVE, a.P - g0);

    // stage the SH rows of this wave's 64 Gaussians into LDS (coalesced)
    const bool split = a.shs_rest != nullptr;         // DC [P,3] + REST [P,45] stored separately (M = 16)
    constexpr int RESTF = 45;
    // Round 5, split degree-3 storage (the training layout): the wave's coefficient block goes global -> LDS by LDS-DMA as it lies (64 rows
    // of `_features_rest` = 11 520 contiguous bytes, row pitch 45 dwords, then the 768 bytes of `_features_dc`), the gradients go back
    // into the same linear image and leave as one contiguous float4 stream: no scatter with a division by 45 on either side.
    const bool linear = use_sh && split && a.D == 3 && pre_bwd_linear;
    float *const lin_dc = wl + WAVE * RESTF;
    if (linear) {
        if (__any(vis)) {
            const int rr = max(rows, 0);
            const float *sp = a.shs_rest + (size_t)g0 * RESTF;       // g0 % 64 == 0: 16-byte aligned
            const int nfl = rr * RESTF, nd = rr * 3;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const int e4 = (lane + WAVE * j) * 4;
                if (e4 + 3 < nfl)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + e4),
                                                     (__attribute__((address_space(3))) void *)(wl + WAVE * 4 * j), 16, 0, 0);
            }
            if (lane * 4 + 3 < nd)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.shs + (size_t)g0 * 3 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)lin_dc, 16, 0, 0);
            if (rr < WAVE) {          // last wave of the array: the (at most one) 16-byte chunk of each block that straddles its end
                for (int e = (nfl & ~3) + lane; e < nfl; e += WAVE) wl[e] = sp[e];
                for (int e = (nd & ~3) + lane; e < nd; e += WAVE) lin_dc[e] = a.shs[(size_t)g0 * 3 + e];
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the LDS-DMA copies have landed
        }
        wave_sync();      // the rows are this wave's own
    } else if (use_sh && split) {
        if (__any(vis)) {
            for (int e = lane; e < rows * 3; e += WAVE) wl[(e / 3) * SH_PITCH_B + (e % 3)] = a.shs[(size_t)g0 * 3 + e];
            const int need = nb * 3 - 3;              // floats of each REST row the active degree uses
            if (need == RESTF) {                      // full rows: flat float4 copy of the wave's contiguous block
                const float *sp = a.shs_rest + (size_t)g0 * RESTF;
                const int nfl = rows * RESTF;
                for (int e4 = lane * 4; e4 < nfl; e4 += WAVE * 4) {
                    float v4[4];
                    if (e4 + 3 < nfl) { const float4 v = *reinterpret_cast<const float4 *>(sp + e4); v4[0] = v.x; v4[1] = v.y; v4[2] = v.z; v4[3] = v.w; }
                    else { for (int t = 0; t < 4; t++) v4[t] = e4 + t < nfl ? sp[e4 + t] : 0.f; }
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int e = e4 + t;
                        if (e < nfl) wl[(e / RESTF) * SH_PITCH_B + 3 + (e % RESTF)] = v4[t];
                    }
                }
            } else {
                for (int e = lane; e < rows * need; e += WAVE) {
                    const int r = e / need, c = e - r * need;
                    wl[r * SH_PITCH_B + 3 + c] = a.shs_rest[((size_t)g0 + r) * RESTF + c];
                }
            }
        }
        wave_sync();      // the rows are this wave's own
    } else if (use_sh) {
        if (vec_ok) {
            if (__any(vis)) {
                const int rowq = rowf / 4;
                switch (a.D) {
                case 0: stage_sh_rows_b<1>(a.shs, g0, rows, rowq, wl, lane); break;
                case 1: stage_sh_rows_b<3>(a.shs, g0, rows, rowq, wl, lane); break;
                case 2: stage_sh_rows_b<7>(a.shs, g0, rows, rowq, wl, lane); break;
                default: stage_sh_rows_b<12>(a.shs, g0, rows, rowq, wl, lane); break;
                }
            }
        } else if (vis) {
            for (int k = 0; k < nb * 3; k++) wl[lane * SH_PITCH_B + k] = a.shs[(size_t)i * rowf + k];
        }
        wave_sync();      // the rows are this wave's own
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dcol_sh[3] = {0.f, 0.f, 0.f};
    // the Gaussian's 64-byte gradient record (three coalesced float4 loads)
    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga, gc4 = ga;
    if (valid) {
        float4 *rec = reinterpret_cast<float4 *>(a.accum + (size_t)i * GRAD_STRIDE);
        ga = rec[0]; gb = rec[1]; gc4 = rec[2];
        if (a.rezero && vis) {                      // only visible Gaussians were ever added to
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            rec[0] = z; rec[1] = z; rec[2] = z;
        }
    }
    // blend_bwd accumulates MOMENTS of q = dL/dG * G over the pixels (gms_blend.h GRAD_*): sum q*dx, q*dy, q*dx*dx,
    // q*dx*dy, q*dy*dy and sum q.  The per-splat linear map to the gradients of (pixel mean, conic, opacity') uses
    // only per-Gaussian constants (conic, opacity'), so it is applied once here instead of per pixel pair there.
    const float mom_x = ga.x, mom_y = ga.y, mom_xx = ga.z, mom_xy = ga.w, mom_yy = gb.x, mom_0 = gb.y;
    float acc_mx = 0.f, acc_my = 0.f;
    const float acc_col[3] = {gb.z, gb.w, gc4.x};
    const float acc_id = gc4.y;
    float dop = 0.f;
    float *row = wl + lane * SH_PITCH_B;

    if (vis) {
        const float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
        const float *V = a.view, *Mx = a.proj;
        float vx, vy, vz;
        view_transform(V, px, py, pz, vx, vy, vz);
        Cov3 cv;
        float s_in[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f};
        if (a.cov3Dp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cv.c[k] = a.cov3Dp[6 * (size_t)i + k];
        } else {
            s_in[0] = a.scales[3 * (size_t)i]; s_in[1] = a.scales[3 * (size_t)i + 1]; s_in[2] = a.scales[3 * (size_t)i + 2];
            const float4 qv = *reinterpret_cast<const float4 *>(a.rots + 4 * (size_t)i);
            q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
            cov3d_from_scale_rot(s_in, a.mod, q, cv);
        }
        const float fx = (float)a.W / (2.f * a.tanx), fy = (float)a.H / (2.f * a.tany);
        Ewa e;
        ewa_project(V, vx, vy, vz, cv, fx, fy, 1.3f * a.tanx, 1.3f * a.tany, e);
        const float b = e.b, aD = e.a0 + DILATE, cD = e.c0 + DILATE;
        const float det = aD * cD - b * b;
        {
            // conic exactly as preprocess_fwd formed it; d power / d(dx) = -(A dx + B dy), d power / dA = -dx^2/2,
            // d power / dB carries the reference's 1/2, alpha = opacity' * G (straight-through clamp)
            const float dinv = 1.f / det;
            const float cA = cD * dinv, cB = -b * dinv, cC = aD * dinv;
            acc_mx = -(0.5f * (float)a.W) * (cA * mom_x + cB * mom_y);
            acc_my = -(0.5f * (float)a.H) * (cC * mom_y + cB * mom_x);
            float opp = a.opac[i];
            if (a.aa) opp *= sqrtf(fmaxf(0.000025f, (e.a0 * e.c0 - b * b) / det));
            dop = opp > 0.f ? mom_0 / opp : 0.f;
        }
        const float gA = -0.5f * mom_xx, gB = -0.5f * mom_xy, gC = -0.5f * mom_yy;
        const float d2inv = 1.f / (det * det + 0.0000001f);
        float dL_da = d2inv * (-cD * cD * gA + 2.f * b * cD * gB + (det - aD * cD) * gC);
        float dL_dc = d2inv * (-aD * aD * gC + 2.f * aD * b * gB + (det - aD * cD) * gA);
        float dL_db = d2inv * 2.f * (b * cD * gA - (det + 2.f * b * b) * gB + aD * b * gC);
        if (a.aa) {
            const float det0 = e.a0 * e.c0 - b * b;
            const float ratio = det0 / det;
            const float h = sqrtf(fmaxf(0.000025f, ratio));
            const float dL_dh = dop * a.opac[i];
            dop = dop * h;
            const float dL_dr = ratio <= 0.000025f ? 0.f : dL_dh / (2.f * h);
            dL_da += dL_dr * (e.c0 / det - det0 * cD / (det * det));
            dL_dc += dL_dr * (e.a0 / det - det0 * aD / (det * det));
            dL_db += dL_dr * (-2.f * b / det + det0 * 2.f * b / (det * det));
        }
        const float *T0 = e.T0, *T1 = e.T1;
        dcov[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
        dcov[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
        dcov[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
        dcov[1] = 2.f * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2.f * T1[0] * T1[1] * dL_dc;
        dcov[2] = 2.f * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2.f * T1[0] * T1[2] * dL_dc;
        dcov[4] = 2.f * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2.f * T1[1] * T1[2] * dL_dc;
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float dT0 = 2.f * e.ST0[c] * dL_da + e.ST1[c] * dL_db;
            const float dT1 = 2.f * e.ST1[c] * dL_dc + e.ST0[c] * dL_db;
            dJ00 += dT0 * V[4 * c + 0]; dJ02 += dT0 * V[4 * c + 2];
            dJ11 += dT1 * V[4 * c + 1]; dJ12 += dT1 * V[4 * c + 2];
        }
        const float tzi = 1.f / e.tz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
        const float dtx = e.xmul * -fx * tz2 * dJ02;
        const float dty = e.ymul * -fy * tz2 * dJ12;
        float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * e.tx) * tz3 * dJ02 + (2.f * fy * e.ty) * tz3 * dJ12;
        dtz -= acc_id * tz2;
#pragma unroll
        for (int r = 0; r < 3; r++) dmean[r] += V[4 * r + 0] * dtx + V[4 * r + 1] * dty + V[4 * r + 2] * dtz;

        // pixel mean -> world mean
        const float hx = dot3p(Mx[0], px, Mx[4], py, Mx[8], pz, Mx[12]);
        const float hy = dot3p(Mx[1], px, Mx[5], py, Mx[9], pz, Mx[13]);
        const float hw = dot3p(Mx[3], px, Mx[7], py, Mx[11], pz, Mx[15]);
        const float mw = 1.f / (hw + 0.0000001f);
        const float gmx = acc_mx, gmy = acc_my;
        const float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
#pragma unroll
        for (int r = 0; r < 3; r++)
            dmean[r] += (Mx[4 * r + 0] * mw - Mx[4 * r + 3] * mul1) * gmx + (Mx[4 * r + 1] * mw - Mx[4 * r + 3] * mul2) * gmy;

        // colour
        if (use_sh) {
            const float ddx = px - a.campos[0], ddy = py - a.campos[1], ddz = pz - a.campos[2];
            const float inv = 1.f / sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
            const float x = ddx * inv, y = ddy * inv, z = ddz * inv;
            const unsigned cl = a.clamped[i];
            float dRGB[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { dRGB[c] = ((cl >> c) & 1u) ? 0.f : acc_col[c]; dcol_sh[c] = dRGB[c]; }
            float gdir[3] = {0.f, 0.f, 0.f};
            if (linear) {
                float r[48];
#pragma unroll
                for (int c = 0; c < 3; c++) r[c] = lin_dc[lane * 3 + c];
#pragma unroll
                for (int m = 0; m < RESTF; m++) r[3 + m] = wl[lane * RESTF + m];
                sh_backward_regs<3>(r, x, y, z, dRGB, gdir);
                if (!a.dL_dcolor_sh) {          // (factorised mode writes no SH gradient rows)
#pragma unroll
                    for (int c = 0; c < 3; c++) lin_dc[lane * 3 + c] = r[c];
#pragma unroll
                    for (int m = 0; m < RESTF; m++) wl[lane * RESTF + m] = r[3 + m];
                }
            } else
            switch (a.D) {
            case 0: sh_backward<0>(row, x, y, z, dRGB, gdir); break;
            case 1: sh_backward<1>(row, x, y, z, dRGB, gdir); break;
            case 2: sh_backward<2>(row, x, y, z, dRGB, gdir); break;
            default: sh_backward<3>(row, x, y, z, dRGB, gdir); break;
            }
            const float dd = x * gdir[0] + y * gdir[1] + z * gdir[2];
            dmean[0] += (gdir[0] - x * dd) * inv;
            dmean[1] += (gdir[1] - y * dd) * inv;
            dmean[2] += (gdir[2] - z * dd) * inv;
        }

        // cov3D -> scale / quaternion
        if (!a.cov3Dp) {
            const float mod = a.mod;
            const float s[3] = {mod * s_in[0], mod * s_in[1], mod * s_in[2]};
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            const float R[3][3] = {
                {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
            const float Gs[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float dR[3][3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float accs = 0.f;
#pragma unroll
                for (int a1 = 0; a1 < 3; a1++) {
                    float acc = 0.f;
#pragma unroll
                    for (int k2 = 0; k2 < 3; k2++) acc += Gs[a1][k2] * R[k2][j] * s[j];
                    const float dLm = 2.f * acc;            // d loss / d L[a1][j],  L = R diag(s)
                    accs += R[a1][j] * dLm;
                    dR[a1][j] = dLm * s[j];
                }
                dscale[j] = accs * a.scale_grad_mod;
            }
            drot[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            drot[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
            drot[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
            drot[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
    }

    // ---- SH gradient rows: zero what was not written, then stream the wave's rows out coalesced
    // (factorised mode: dL/dsh = Y(dir) x dL/dcolour is formed later, by sh_grad_expand, from the [P,3] factor written below)
    if (use_sh && a.dL_dcolor_sh) {
    } else if (linear) {
        if (!vis) {          // culled Gaussian: a zero gradient row
#pragma unroll
            for (int c = 0; c < 3; c++) lin_dc[lane * 3 + c] = 0.f;
#pragma unroll
            for (int m = 0; m < RESTF; m++) wl[lane * RESTF + m] = 0.f;
        }
        wave_sync();      // the rows are this wave's own
        if (rows > 0) {
            const int nfl = rows * RESTF, nd = rows * 3;
            float *dp = a.dL_dsh_rest + (size_t)g0 * RESTF, *dd = a.dL_dsh + (size_t)g0 * 3;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const int e4 = (lane + WAVE * j) * 4;
                // Non-temporal: the 54 MB of dL/dSH are written once and read by nobody in this call.  As plain stores the launch had two
                // speeds, 45 us in most processes and 53 in about one of four (same binary, same inputs: it goes with where the allocator
                // put the buffers); streamed past the caches it takes 44 us in every one of ten runs (profiles/r06x4_*, r06x5_*).
                if (e4 + 3 < nfl) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(wl + e4), reinterpret_cast<v4f *>(dp + e4));
                }
                else for (int t = 0; t < 4; t++) if (e4 + t < nfl) dp[e4 + t] = wl[e4 + t];
            }
            if (lane * 4 + 3 < nd) *reinterpret_cast<float4 *>(dd + lane * 4) = *reinterpret_cast<const float4 *>(lin_dc + lane * 4);
            else for (int t = 0; t < 4; t++) if (lane * 4 + t < nd) dd[lane * 4 + t] = lin_dc[lane * 4 + t];
        }
    } else if (use_sh && split) {
        for (int q = vis ? (nb * 3 + 3) / 4 : 0; q < 12; q++)
            *reinterpret_cast<float4 *>(row + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        wave_sync();      // the rows are this wave's own
        if (rows > 0) {
            for (int e = lane; e < rows * 3; e += WAVE) a.dL_dsh[(size_t)g0 * 3 + e] = wl[(e / 3) * SH_PITCH_B + (e % 3)];
            float *dp = a.dL_dsh_rest + (size_t)g0 * RESTF;
            const int nfl = rows * RESTF;
            for (int e4 = lane * 4; e4 < nfl; e4 += WAVE * 4) {
                float v4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int e = e4 + t;
                    v4[t] = e < nfl ? wl[(e / RESTF) * SH_PITCH_B + 3 + (e % RESTF)] : 0.f;
                }
                if (e4 + 3 < nfl) *reinterpret_cast<float4 *>(dp + e4) = make_float4(v4[0], v4[1], v4[2], v4[3]);
                else for (int t = 0; t < 4; t++) if (e4 + t < nfl) dp[e4 + t] = v4[t];
            }
        }
    } else if (use_sh) {
        const int used = vis ? nb * 3 : 0;
        if (vec_ok) {
            // chunks the SH backward did not write (culled Gaussian, or coefficients above the active degree)
            const int rowq = rowf / 4;
            for (int q = vis ? (nb * 3 + 3) / 4 : 0; q < rowq; q++)
                *reinterpret_cast<float4 *>(row + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            wave_sync();      // the rows are this wave's own
            if (rows > 0) {
                float4 *dst = reinterpret_cast<float4 *>(a.dL_dsh + (size_t)g0 * rowf);
                for (int idx = lane; idx < rows * rowq; idx += WAVE) {
                    int r = idx / rowq, c = idx - r * rowq;
                    dst[(size_t)r * rowq + c] = *reinterpret_cast<const float4 *>(wl + r * SH_PITCH_B + c * 4);
                }
            }
        } else if (valid) {
            for (int k = 0; k < rowf; k++) a.dL_dsh[(size_t)i * rowf + k] = k < used ? row[k] : 0.f;
        }
    }
    if (!valid) return;
    if (MESH) {
        // the frame came straight from the mesh (gmsplat.h, ABI 8): on through the face -> Gaussian parameterization from registers
        const float gs[3] = {dscale[0], dscale[1], dscale[2]}, gq[4] = {drot[0], drot[1], drot[2], drot[3]};
        splat_backward_from_registers(a.mesh, (int64_t)i, dmean, gs, gq, dop, a.mesh_dvertices, a.mesh_dalpha, a.mesh_dscale, a.mesh_dopacity);
    } else {
        a.dL_dopacity[i] = dop;
    }
    a.dL_dmean2D[3 * (size_t)i] = acc_mx; a.dL_dmean2D[3 * (size_t)i + 1] = acc_my; a.dL_dmean2D[3 * (size_t)i + 2] = 0.f;
    if (a.dL_dcolors) { a.dL_dcolors[3 * (size_t)i] = acc_col[0]; a.dL_dcolors[3 * (size_t)i + 1] = acc_col[1]; a.dL_dcolors[3 * (size_t)i + 2] = acc_col[2]; }
    if (a.dL_dcolor_sh && a.campos_row && i == 0) {
        a.dL_dcolor_sh[3 * (size_t)a.P] = a.campos[0]; a.dL_dcolor_sh[3 * (size_t)a.P + 1] = a.campos[1]; a.dL_dcolor_sh[3 * (size_t)a.P + 2] = a.campos[2];
    }
    if (a.dL_dcolor_sh) { a.dL_dcolor_sh[3 * (size_t)i] = dcol_sh[0]; a.dL_dcolor_sh[3 * (size_t)i + 1] = dcol_sh[1]; a.dL_dcolor_sh[3 * (size_t)i + 2] = dcol_sh[2]; }
    if (MESH) return;
    a.dL_dmeans3D[3 * (size_t)i] = dmean[0]; a.dL_dmeans3D[3 * (size_t)i + 1] = dmean[1]; a.dL_dmeans3D[3 * (size_t)i + 2] = dmean[2];
    if (a.cov3Dp) {
        if (a.dL_dcov3D)
#pragma unroll
            for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)i + k] = dcov[k];
    } else {
        a.dL_dscales[3 * (size_t)i] = dscale[0]; a.dL_dscales[3 * (size_t)i + 1] = dscale[1]; a.dL_dscales[3 * (size_t)i + 2] = dscale[2];
        *reinterpret_cast<float4 *>(a.dL_drots + 4 * (size_t)i) = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }
}


// ------------------------------------------------------------------------------------ fault injection
// Negative controls of the parity criterion (gmsplat.h, gms_set_fault): scale one field of every `every`-th row.  A
// separate launch between / after the production kernels, which therefore carry no fault branch.
__global__ void fault_scale_kernel(float *base, int rows, int stride, int field, int every, float factor)
{
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * every;
    if (i < rows) base[(size_t)i * stride + field] *= factor;
}

// ------------------------------------------------------------------------------------ deterministic mode: ordered reduction
// The blend kernels left one partial record per (Gaussian, tile) instance (NSUB = 1, micro-tile kernels) or per instance and 8x8
// quadrant (NSUB = 4), indexed by the instance's position in the sorted key list.  Sixteen lanes per Gaussian (lane = field of the
// 64-byte record): walk the Gaussian's tile rectangle in emission order (y outer, x inner), find the instance in the tile's sorted
// list by binary search on its unique (depth bits, id) key, and add its NSUB records in index order.  Plain store into the
// Gaussian's gradient record (all zero on entry, as always).
template <int NSUB>
__global__ void __launch_bounds__(BLOCK) det_reduce_kernel(int P, int gx, int gy, const int *radii, const SplatRec *rec, const float *depth,
                                                           const uint32_t *tile_offset, const uint64_t *keys, uint64_t capacity,
                                                           const float *part, float *accum)
{
    const int i = (int)((blockIdx.x * (unsigned)BLOCK + threadIdx.x) >> 4), f = (int)(threadIdx.x & 15u);
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float4 q0 = rec[i].q0;
    int minx, miny, maxx, maxy;
    tile_rect(q0.x, q0.y, (float)r, gx, gy, minx, miny, maxx, maxy);
    const uint64_t key = ((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i;
    float acc = 0.f;
    for (int ty = miny; ty < maxy; ty++)
        for (int tx = minx; tx < maxx; tx++) {
            const int t = ty * gx + tx;
            uint32_t lo = tile_offset[t], hi = tile_offset[t + 1];
            if ((uint64_t)hi > capacity) return;
            while (lo < hi) {                                    // first position with keys[pos] >= key
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            // (keys[lo] == key by construction: every tile of the rectangle holds this Gaussian exactly once)
#pragma unroll
            for (int sub = 0; sub < NSUB; sub++) acc += part[((size_t)lo * NSUB + sub) * GRAD_STRIDE + f];
        }
    if (f < 10) accum[(size_t)i * GRAD_STRIDE + f] = acc;
}

// ------------------------------------------------------------------------------------ factorised SH gradient
// dL/dsh[k][c] of one view is the outer product Y_k(dir) * dL/dcolour_c (clamp mask folded into dL/dcolour), and dir depends
// only on the Gaussian's position and that view's camera centre.  A multi-view step therefore exchanges the [P,3] factors of
// its views (3 floats per Gaussian per view instead of 48) and forms  sum_v Y(dir_v) (x) g_v  here, views in index order.
// The basis values are the `bv` terms of sh_backward above, same expressions: with one view this reproduces the dense
// gradient bit for bit.
struct ShExpandArgs {
    int P, D, M, V;
    const float *means3D;      // [P,3]
    const float *campos;       // [V,3]
    const float *factors;      // [V,P,3] clamp-masked dL/dcolour of every view
    size_t stride;             // floats between views
    float *dL_dsh;             // [P,M,3], or [P,1,3] with dL_dsh_rest = [P,M-1,3]
    float *dL_dsh_rest;
    int accumulate;            // add to the destination instead of overwriting it
};

template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float *Y)
{
    Y[0] = SH_C0;
    if (DEG > 0) { Y[1] = -SH_C1 * y; Y[2] = SH_C1 * z; Y[3] = -SH_C1 * x; }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        Y[4] = SH_C2[0] * x * y; Y[5] = SH_C2[1] * y * z; Y[6] = SH_C2[2] * (2.f * zz - xx - yy);
        Y[7] = SH_C2[3] * x * z; Y[8] = SH_C2[4] * (xx - yy);
    }
    if (DEG > 2) {
        const float xx = x * x, yy = y * y, zz = z * z;
        Y[9] = SH_C3[0] * y * (3.f * xx - yy); Y[10] = SH_C3[1] * x * y * z; Y[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
        Y[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); Y[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
        Y[14] = SH_C3[5] * z * (xx - yy); Y[15] = SH_C3[6] * x * (xx - 3.f * yy);
    }
}

template <int DEG>
__global__ void __launch_bounds__(BLOCK) sh_grad_expand_kernel(ShExpandArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.P) return;
    const float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
    float acc[NB * 3];
#pragma unroll
    for (int k = 0; k < NB * 3; k++) acc[k] = 0.f;
    for (int v = 0; v < a.V; v++) {
        const float *gp = a.factors + (size_t)v * a.stride + (size_t)i * 3;
        const float g[3] = {gp[0], gp[1], gp[2]};
        if (g[0] == 0.f && g[1] == 0.f && g[2] == 0.f) continue;      // not seen by this view (adds exact zeros)
        const float ddx = px - a.campos[3 * v], ddy = py - a.campos[3 * v + 1], ddz = pz - a.campos[3 * v + 2];
        const float inv = 1.f / sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
        float Y[16];
        sh_basis<DEG>(ddx * inv, ddy * inv, ddz * inv, Y);
#pragma unroll
        for (int k = 0; k < NB; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) acc[k * 3 + c] += Y[k] * g[c];
    }
    // destination rows (all indices of `acc` static); coefficients above the active degree get zero
    float *d0 = a.dL_dsh + (size_t)i * (a.dL_dsh_rest ? 3 : a.M * 3);
    float *d1 = a.dL_dsh_rest ? a.dL_dsh_rest + (size_t)i * ((a.M - 1) * 3) : d0 + 3;
    if (a.accumulate) {
#pragma unroll
        for (int c = 0; c < 3; c++) d0[c] += acc[c];
#pragma unroll
        for (int k = 3; k < NB * 3; k++) d1[k - 3] += acc[k];
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) d0[c] = acc[c];
#pragma unroll
        for (int k = 3; k < NB * 3; k++) d1[k - 3] = acc[k];
        for (int k = NB * 3; k < a.M * 3; k++) d1[k - 3] = 0.f;
    }
}

}  // namespace gms

using namespace gms;

extern "C" int32_t gms_rasterize_backward(const GmsRasterBackwardArgs *A, void *stream_)
{
    gms::TraceRange trace_range("gms_rasterize_backward");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    if (!A || A->P < 0 || A->width <= 0 || A->height <= 0) {
        set_error("gms_rasterize_backward: invalid sizes");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    const int P = A->P, W = A->width, H = A->height;
    if (P == 0) return GMS_OK;
    const bool sr = A->scales && A->rotations;
    const GmsMeshArgs *mesh = A->mesh;
    if (mesh) {
        const int32_t mrc = check_mesh_args(mesh, false);
        if (mrc != GMS_OK) return mrc;
        if (mesh->P != (int64_t)P || mesh->splats_per_face <= 0 || mesh->splats_per_face > 4 || !mesh->_opacity || !sr || A->cov3D_precomp ||
            !A->mesh_dL_dvertices || !A->mesh_dL_dalpha || !A->mesh_dL_dscale || !A->mesh_dL_d_opacity || det_mode()) {
            set_error("gms_rasterize_backward: the mesh backward inside preprocess_bwd needs a complete GmsMeshArgs (P equal, 1-4 splats per face, "
                      "_opacity), the scales + rotations path, all four mesh_dL_* outputs, and is not available in deterministic mode");
            return GMS_ERR_INVALID_ARGUMENT;
        }
    }
    if (!A->means3D || !A->opacities || !A->radii || !A->geom_buffer || !A->binning_buffer || !A->image_buffer ||
        !A->dL_dout_color || !A->dL_dmeans2D || !A->grad_accum || (!mesh && !A->dL_dopacity) || (A->colors_precomp && !A->dL_dcolors) ||
        (!mesh && !A->dL_dmeans3D) || (A->sh_factor_mode != 0 && A->sh_factor_mode != 1) || (A->sh_factor_mode && (!A->shs || !A->dL_dcolors)) ||
        (A->factor_campos_row && !A->sh_factor_mode) ||
        (A->shs && !A->sh_factor_mode && !A->dL_dsh) || (A->shs_rest && ((!A->sh_factor_mode && !A->dL_dsh_rest) || A->M != 16)) ||
        ((A->shs == nullptr) == (A->colors_precomp == nullptr)) ||
        (sr == (A->cov3D_precomp != nullptr)) || (sr && !mesh && (!A->dL_dscales || !A->dL_drotations)) ||
        (A->cov3D_precomp && !A->dL_dcov3D)) {
        set_error("gms_rasterize_backward: null or inconsistent pointer arguments");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    GeomState geom = GeomState::carve(const_cast<void *>(A->geom_buffer), (size_t)P);
    ImageState img = ImageState::carve(const_cast<void *>(A->image_buffer), (size_t)W, (size_t)H);
    const int T = gx * gy;
    const uint32_t L = seg_len_min();         // carving; the frame's own L is in img.scan_out[3]
    const uint64_t cap = (uint64_t)(A->binning_capacity > 0 ? A->binning_capacity : (A->num_rendered > 0 ? A->num_rendered : 1));
    BinningState bin = BinningState::carve(const_cast<void *>(A->binning_buffer), (size_t)cap, (size_t)T, L);

    if (A->num_rendered > 0) {
        BlendGrid g;
        g.W = W; g.H = H; g.gx = gx; g.gy = gy; g.T = T; g.scan_out = img.scan_out; g.tile_offset = img.tile_offset;
        g.unit_first = img.unit_first; g.mseg_first = img.mseg_first; g.unit_tile = bin.unit_tile; g.keys = bin.keys;
        g.seg_state = bin.seg_state; g.capacity = cap;
        g.max_units = (uint32_t)BinningState::n_units((size_t)cap, (size_t)T, L); g.dbg = 0; g.unit_run = unit_run(); g.dbg_buf = nullptr; g.tile_dead = img.tile_dead; g.tile_cmax = img.tile_cmax;
        g.mmask = bin.mmask;
        BlendBwdArgs b;
        b.rec = geom.rec; b.bg = A->background; b.final_T = img.final_T; b.n_contrib = img.n_contrib;
        b.dL_dpix = A->dL_dout_color; b.dL_dinvd = A->dL_dout_invdepth; b.accum = A->grad_accum;
        b.has_invd = A->dL_dout_invdepth != nullptr;
        b.part = nullptr;
        uint32_t mu = (uint32_t)BinningState::n_units((size_t)cap, (size_t)T, L);
        if (A->num_units > 0 && (uint64_t)A->num_units < mu) mu = (uint32_t)A->num_units;      // exact count from the forward
        const bool micro = use_micro(cap, T);
        const bool det = det_mode() != 0 && fault_mode() == 0;
        if (det) {      // deterministic mode (gmsplat.h): partial records per instance, then an ordered reduction per Gaussian
            const size_t nsub = micro ? 1 : 4, bytes = (size_t)A->num_rendered * nsub * GRAD_STRIDE * sizeof(float);
            b.part = static_cast<float *>(det_scratch(0, bytes, stream));
            if (!b.part) { set_error("deterministic mode: scratch allocation of %zu bytes failed", bytes); return GMS_ERR_ALLOC; }
            // the quadrant kernels store only the (instance, quadrant) pairs that survive their culls; the micro-tile flush writes
            // every instance (zeros included) and needs no clearing
            if (!micro) GMS_HIP_CHECK(hipMemsetAsync(b.part, 0, bytes, stream));
        }
        int32_t rc = micro ? launch_micro_backward(g, b, mu, A->debug != 0, stream) : launch_blend_backward(g, b, mu, A->debug != 0, stream);
        if (rc != GMS_OK) return rc;
        if (det) {
            const unsigned rblocks = (unsigned)(((size_t)P * 16 + BLOCK - 1) / BLOCK);
            if (micro) det_reduce_kernel<1><<<rblocks, BLOCK, 0, stream>>>(P, gx, gy, A->radii, geom.rec, geom.depth, img.tile_offset, bin.keys, cap, b.part, A->grad_accum);
            else det_reduce_kernel<4><<<rblocks, BLOCK, 0, stream>>>(P, gx, gy, A->radii, geom.rec, geom.depth, img.tile_offset, bin.keys, cap, b.part, A->grad_accum);
            GMS_KERNEL_CHECK(A->debug, stream, "det_reduce");
        }
        if (fault_mode() == 1)      // negative control: sum(q dx^2) of every 1000th Gaussian off by 2e-3
            fault_scale_kernel<<<(unsigned)((P / 1000 + 256) / 256), 256, 0, stream>>>(A->grad_accum, P, GRAD_STRIDE, GRAD_CA, 1000, 1.002f);
    }
    PreBwdArgs p{};
    p.P = P; p.D = A->D; p.M = A->M; p.W = W; p.H = H;
    p.means3D = A->means3D; p.shs = A->shs; p.shs_rest = A->shs_rest; p.colors = A->colors_precomp; p.opac = A->opacities; p.scales = A->scales;
    p.rots = A->rotations; p.cov3Dp = A->cov3D_precomp; p.view = A->viewmatrix; p.proj = A->projmatrix; p.campos = A->campos;
    p.mod = A->scale_modifier; p.scale_grad_mod = upstream_scale_mod_grad() ? 1.f : A->scale_modifier; p.tanx = A->tan_fovx; p.tany = A->tan_fovy; p.aa = A->antialiasing; p.radii = A->radii;
    p.clamped = geom.clamped; p.accum = A->grad_accum; p.rezero = fault_mode() == 3 ? 0 : A->grad_accum_rezero; p.dL_dmean2D = A->dL_dmeans2D;
    p.dL_dcolors = A->colors_precomp ? A->dL_dcolors : nullptr; p.dL_dcolor_sh = (A->shs && A->sh_factor_mode) ? A->dL_dcolors : nullptr; p.campos_row = A->factor_campos_row; p.dL_dopacity = A->dL_dopacity; p.dL_dmeans3D = A->dL_dmeans3D;
    p.dL_dcov3D = A->dL_dcov3D; p.dL_dsh = A->dL_dsh; p.dL_dsh_rest = A->dL_dsh_rest; p.dL_dscales = A->dL_dscales; p.dL_drots = A->dL_drotations;
    static int pre_linear = -1;         // GMS_PRE_BWD_LINEAR=0: the 52-dword-pitch scatter staging of rounds 1-4 for split degree-3 storage
    if (pre_linear < 0) { const char *e = getenv("GMS_PRE_BWD_LINEAR"); pre_linear = e ? (atoi(e) != 0) : 1; }
    const int lin_ok = pre_linear && A->shs && A->shs_rest && A->D == 3 && (((uintptr_t)A->shs) & 15u) == 0 && (((uintptr_t)A->shs_rest) & 15u) == 0 &&
                       (A->sh_factor_mode || ((((uintptr_t)A->dL_dsh) & 15u) == 0 && (((uintptr_t)A->dL_dsh_rest) & 15u) == 0));
    if (mesh) {
        p.mesh = *mesh; p.mesh_dvertices = A->mesh_dL_dvertices; p.mesh_dalpha = A->mesh_dL_dalpha; p.mesh_dscale = A->mesh_dL_dscale; p.mesh_dopacity = A->mesh_dL_d_opacity;
        GMS_LAUNCH(GMS_K_PREPROCESS_BWD, stream, preprocess_bwd_kernel<true><<<(unsigned)((P + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(p, lin_ok));
    } else {
        GMS_LAUNCH(GMS_K_PREPROCESS_BWD, stream, preprocess_bwd_kernel<false><<<(unsigned)((P + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(p, lin_ok));
    }
    GMS_KERNEL_CHECK(A->debug, stream, "preprocess_bwd");
    if (fault_mode() == 4 && A->dL_dscales)      // negative control: dL/dscale.x of every 1000th Gaussian off by 2e-3
        fault_scale_kernel<<<(unsigned)((P / 1000 + 256) / 256), 256, 0, stream>>>(A->dL_dscales, P, 3, 0, 1000, 1.002f);
    if (fault_mode() == 5 && A->dL_dscales)      // negative control: dL/dscale of every 100th Gaussian off by 1.3e-3 (between 1e-3 and 2e-3)
        for (int f = 0; f < 3; f++)
            fault_scale_kernel<<<(unsigned)((P / 100 + 256) / 256), 256, 0, stream>>>(A->dL_dscales, P, 3, f, 100, 1.0013f);
    return GMS_OK;
}

extern "C" int32_t gms_sh_grad_expand(const GmsShGradExpandArgs *A, void *stream_)
{
    gms::TraceRange trace_range("gms_sh_grad_expand");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    if (!A || A->P < 0 || A->V < 1 || A->D < 0 || A->D > 3 || A->M < (A->D + 1) * (A->D + 1) || !A->means3D || !A->campos || !A->factors || (A->factor_stride != 0 && A->factor_stride < (int64_t)A->P * 3) ||
        !A->dL_dsh || (A->dL_dsh_rest && A->M < 2)) {
        set_error("gms_sh_grad_expand: invalid sizes or null pointer arguments");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    if (A->P == 0) return GMS_OK;
    ShExpandArgs e;
    e.P = A->P; e.D = A->D; e.M = A->M; e.V = A->V; e.means3D = A->means3D; e.campos = A->campos; e.factors = A->factors;
    e.stride = A->factor_stride > 0 ? (size_t)A->factor_stride : (size_t)A->P * 3;
    e.dL_dsh = A->dL_dsh; e.dL_dsh_rest = A->dL_dsh_rest; e.accumulate = A->accumulate;
    const unsigned blocks = (unsigned)((A->P + BLOCK - 1) / BLOCK);
    switch (A->D) {
    case 0: GMS_LAUNCH(GMS_K_SH_EXPAND, stream, sh_grad_expand_kernel<0><<<blocks, BLOCK, 0, stream>>>(e)); break;
    case 1: GMS_LAUNCH(GMS_K_SH_EXPAND, stream, sh_grad_expand_kernel<1><<<blocks, BLOCK, 0, stream>>>(e)); break;
    case 2: GMS_LAUNCH(GMS_K_SH_EXPAND, stream, sh_grad_expand_kernel<2><<<blocks, BLOCK, 0, stream>>>(e)); break;
    default: GMS_LAUNCH(GMS_K_SH_EXPAND, stream, sh_grad_expand_kernel<3><<<blocks, BLOCK, 0, stream>>>(e)); break;
    }
    GMS_KERNEL_CHECK(A->debug, stream, "sh_grad_expand");
    return GMS_OK;
}
