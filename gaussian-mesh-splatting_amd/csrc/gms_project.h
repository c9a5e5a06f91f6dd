// gms_project.h -- per-Gaussian 3D covariance + EWA projection math shared by the forward and
// backward preprocess kernels.  The discrete decisions downstream (depth key, radius, tile
// rectangle) depend on these values, so contraction is OFF here and the few fused operations are
// explicit (dot3p): an independent float32 implementation following the same documented order
// reproduces them bit-for-bit.
#pragma once
#include "gms_common.h"

namespace gms {

struct Cov3 {
    float c[6];   // xx xy xz yy yz zz
};

// Sigma = R S S^T R^T with R from the (w,x,y,z) quaternion as given (caller normalises).
__device__ __forceinline__ void cov3d_from_scale_rot(const float s_in[3], float mod, const float q[4], Cov3 &out)
{
#pragma clang fp contract(off)
    float s0 = mod * s_in[0], s1 = mod * s_in[1], s2 = mod * s_in[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    float L00 = R00 * s0, L01 = R01 * s1, L02 = R02 * s2;
    float L10 = R10 * s0, L11 = R11 * s1, L12 = R12 * s2;
    float L20 = R20 * s0, L21 = R21 * s1, L22 = R22 * s2;
    out.c[0] = L00 * L00 + L01 * L01 + L02 * L02;
    out.c[1] = L00 * L10 + L01 * L11 + L02 * L12;
    out.c[2] = L00 * L20 + L01 * L21 + L02 * L22;
    out.c[3] = L10 * L10 + L11 * L11 + L12 * L12;
    out.c[4] = L10 * L20 + L11 * L21 + L12 * L22;
    out.c[5] = L20 * L20 + L21 * L21 + L22 * L22;
}

// Everything the EWA projection produces; the backward pass re-derives it instead of storing it.
struct Ewa {
    float tx, ty, tz;        // view-space mean with the frustum clamp applied to x, y
    float xmul, ymul;        // 1 inside the clamp, 0 outside (gradient gate)
    float T0[3], T1[3];      // rows of J * Wrot
    float ST0[3], ST1[3];    // Sigma * T0^T, Sigma * T1^T
    float a0, b, c0;         // cov2D before dilation
};

__device__ __forceinline__ void view_transform(const float *V, float px, float py, float pz, float &vx, float &vy,
                                               float &vz)
{
    vx = dot3p(V[0], px, V[4], py, V[8], pz, V[12]);
    vy = dot3p(V[1], px, V[5], py, V[9], pz, V[13]);
    vz = dot3p(V[2], px, V[6], py, V[10], pz, V[14]);
}

__device__ __forceinline__ void ewa_project(const float *V, float vx, float vy, float vz, const Cov3 &cv, float fx,
                                            float fy, float limx, float limy, Ewa &e)
{
#pragma clang fp contract(off)
    float txtz = vx / vz, tytz = vy / vz;
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    e.tz = vz;
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    float J00 = fx / e.tz, J02 = -(fx * e.tx) / (e.tz * e.tz);
    float J11 = fy / e.tz, J12 = -(fy * e.ty) / (e.tz * e.tz);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        e.T0[c] = J00 * V[4 * c + 0] + J02 * V[4 * c + 2];
        e.T1[c] = J11 * V[4 * c + 1] + J12 * V[4 * c + 2];
    }
    const float S[3][3] = {{cv.c[0], cv.c[1], cv.c[2]}, {cv.c[1], cv.c[3], cv.c[4]}, {cv.c[2], cv.c[4], cv.c[5]}};
#pragma unroll
    for (int r = 0; r < 3; r++) {
        e.ST0[r] = S[r][0] * e.T0[0] + S[r][1] * e.T0[1] + S[r][2] * e.T0[2];
        e.ST1[r] = S[r][0] * e.T1[0] + S[r][1] * e.T1[1] + S[r][2] * e.T1[2];
    }
    e.a0 = e.T0[0] * e.ST0[0] + e.T0[1] * e.ST0[1] + e.T0[2] * e.ST0[2];
    e.b = e.T0[0] * e.ST1[0] + e.T0[1] * e.ST1[1] + e.T0[2] * e.ST1[2];
    e.c0 = e.T1[0] * e.ST1[0] + e.T1[1] * e.ST1[1] + e.T1[2] * e.ST1[2];
}

}  // namespace gms
