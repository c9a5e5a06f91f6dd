// gms_mesh.h -- the face-frame arithmetic of the mesh-face -> Gaussian parameterization (K0), shared by mesh_to_gaussians.hip and
// by the fused animated forward of raster_forward.hip (which derives centre / scale / rotation inside the preprocess thread).
// games/mesh_splatting/scene/gaussian_mesh_model.py:86-169, utils/general_utils.py:43-96; contraction off throughout, so both
// users produce the same bits.
#pragma once
#include "gms_common.h"

namespace gms {

constexpr float EPS = 1e-8f;

struct V3 { float x, y, z; };
// (the pragma is lexical: an operator without it would hand its multiply / subtract to the caller WITH the `contract` flag, and
// `(v - c0 * a) - c1 * b` could then become FMAs in one kernel and not in another)
__device__ __forceinline__ V3 operator+(V3 a, V3 b)
{
#pragma clang fp contract(off)
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
__device__ __forceinline__ V3 operator-(V3 a, V3 b)
{
#pragma clang fp contract(off)
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ V3 operator*(float s, V3 a)
{
#pragma clang fp contract(off)
    return {s * a.x, s * a.y, s * a.z};
}
__device__ __forceinline__ float dot(V3 a, V3 b)
{
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
#pragma clang fp contract(off)
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 ldv(const float *p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

struct Frame {
    V3 t0, t1, t2;
    V3 N;  float nN;          // cross product and its norm
    V3 v0, v1, v2;
    V3 u1; float n1, v1n;     // t1 - mean, |u1|, |u1| + eps
    V3 v2i, w; float nw;      // t2 - mean, Gram-Schmidt residual and its norm
    float s1, s2;
};

__device__ __forceinline__ void face_frame(V3 t0, V3 t1, V3 t2, Frame &f)
{
#pragma clang fp contract(off)
    f.t0 = t0; f.t1 = t1; f.t2 = t2;
    f.N = cross(t1 - t0, t2 - t0);
    f.nN = norm(f.N);
    f.v0 = {f.N.x / (f.nN + EPS), f.N.y / (f.nN + EPS), f.N.z / (f.nN + EPS)};
    V3 sum = (t0 + t1) + t2;
    V3 mean = {sum.x / 3.f, sum.y / 3.f, sum.z / 3.f};
    f.u1 = t1 - mean;
    f.n1 = norm(f.u1);
    f.v1n = f.n1 + EPS;
    f.v1 = {f.u1.x / f.v1n, f.u1.y / f.v1n, f.u1.z / f.v1n};
    f.v2i = t2 - mean;
    float c0 = dot(f.v2i, f.v0), c1 = dot(f.v2i, f.v1);
    f.w = (f.v2i - c0 * f.v0) - c1 * f.v1;
    f.nw = norm(f.w);
    f.v2 = {f.w.x / (f.nw + EPS), f.w.y / (f.nw + EPS), f.w.z / (f.nw + EPS)};
    f.s1 = f.v1n / 2.f;
    f.s2 = dot(f.v2i, f.v2) / 2.f;
}

// rotation matrix with columns (v0,v1,v2) -> quaternion; also reports the selected candidate
struct QuatSel { int sel; float a; float sign; float cand[4]; float xsel; };   // xsel = argument of the selected sqrt

__device__ __forceinline__ void rot_to_quat(const Frame &f, float q[4], QuatSel *qs)
{
#pragma clang fp contract(off)
    const float m00 = f.v0.x, m01 = f.v1.x, m02 = f.v2.x;
    const float m10 = f.v0.y, m11 = f.v1.y, m12 = f.v2.y;
    const float m20 = f.v0.z, m21 = f.v1.z, m22 = f.v2.z;
    float x[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
    float qa[4];
    int sel = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) qa[k] = x[k] > 0.f ? sqrtf(x[k]) : 0.f;
#pragma unroll
    for (int k = 1; k < 4; k++)
        if (qa[k] > qa[sel]) sel = k;          // first maximum wins, as torch.argmax
    // select without dynamic register-array indexing (that would be demoted to LDS/scratch)
    const float a = sel == 0 ? qa[0] : sel == 1 ? qa[1] : sel == 2 ? qa[2] : qa[3];
    const float xsel = sel == 0 ? x[0] : sel == 1 ? x[1] : sel == 2 ? x[2] : x[3];
    float c[4];
    const float a2 = a * a;
    if (sel == 0) { c[0] = a2; c[1] = m21 - m12; c[2] = m02 - m20; c[3] = m10 - m01; }
    else if (sel == 1) { c[0] = m21 - m12; c[1] = a2; c[2] = m10 + m01; c[3] = m02 + m20; }
    else if (sel == 2) { c[0] = m02 - m20; c[1] = m10 + m01; c[2] = a2; c[3] = m12 + m21; }
    else { c[0] = m10 - m01; c[1] = m20 + m02; c[2] = m21 + m12; c[3] = a2; }
    const float den = 2.0f * fmaxf(a, 0.1f);
    float o[4] = {c[0] / den, c[1] / den, c[2] / den, c[3] / den};
    const float sign = o[0] < 0.f ? -1.f : 1.f;
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = o[0] < 0.f ? -o[k] : o[k];
    if (qs) {
        qs->sel = sel; qs->a = a; qs->sign = sign;
#pragma unroll
        for (int k = 0; k < 4; k++) qs->cand[k] = c[k];
        qs->xsel = xsel;
    }
}

// host: sizes / tables of a GmsMeshArgs are consistent (F * splats_per_face == P or the CSR tables present, a known alpha_mode, no
// negative size, no null input); sets the error string.  `need_face_offsets` = false: the per-splat readers (splat_from_face) alone will
// run, which use `splat_face` and not the per-face offsets.  mesh_to_gaussians.hip
int32_t check_mesh_args(const GmsMeshArgs *A, bool need_face_offsets = true);

__device__ __forceinline__ int splat_to_face(const GmsMeshArgs &a, int64_t p)
{
    return a.splats_per_face > 0 ? (int)(p / a.splats_per_face) : a.splat_face[p];
}

__device__ __forceinline__ void load_face(const GmsMeshArgs &a, int f, V3 &t0, V3 &t1, V3 &t2)
{
    const int64_t i0 = a.faces[3 * (size_t)f], i1 = a.faces[3 * (size_t)f + 1], i2 = a.faces[3 * (size_t)f + 2];
    t0 = ldv(a.vertices, (size_t)i0); t1 = ldv(a.vertices, (size_t)i1); t2 = ldv(a.vertices, (size_t)i2);
}

__device__ __forceinline__ void barycentric(int mode, const float *raw, float al[3], float &rsum)
{
#pragma clang fp contract(off)
    if (mode == GMS_ALPHA_RELU) {
        float r0 = fmaxf(raw[0], 0.f) + 1e-8f, r1 = fmaxf(raw[1], 0.f) + 1e-8f, r2 = fmaxf(raw[2], 0.f) + 1e-8f;
        rsum = (r0 + r1) + r2;
        al[0] = r0 / rsum; al[1] = r1 / rsum; al[2] = r2 / rsum;
    } else {
        float mx = fmaxf(raw[0], fmaxf(raw[1], raw[2]));
        float e0 = expf(raw[0] - mx), e1 = expf(raw[1] - mx), e2 = expf(raw[2] - mx);
        rsum = (e0 + e1) + e2;
        al[0] = e0 / rsum; al[1] = e1 / rsum; al[2] = e2 / rsum;
    }
}

// What the rasterizer is handed for one splat (the fused property getters of scene/gaussian_model.py:95-115): centre, activated
// scale, unit quaternion, sigmoid opacity.  The statements are the ones mesh_fwd_kernel executes, in its order.
struct SplatParams { float xyz[3], scale[3], q[4], opacity; };
// ... in two halves, so that a caller with other loads in flight can order them: `splat_inputs_load` issues the splat's own loads (face
// indices, raw barycentrics, scale, raw opacity: ONE round trip), `splat_from_inputs` gathers the three vertices (the second, dependent
// round trip) and computes.  preprocess_fwd's K0 instantiation puts the first half IN FRONT of its SH rows' LDS-DMA copies: loads return in
// order, and behind the DMA the vertex gather could not even be issued before 11.5 KB per wave had landed (round 6).
struct SplatInputs { int64_t i0, i1, i2; float raw[3], sc, op_raw; bool has_op; };
__device__ __forceinline__ void splat_inputs_load(const GmsMeshArgs &a, int64_t p, SplatInputs &in)
{
    in.has_op = a._opacity != nullptr;
    in.op_raw = in.has_op ? a._opacity[p] : 0.f;
    const int f = splat_to_face(a, p);
    in.i0 = a.faces[3 * (size_t)f]; in.i1 = a.faces[3 * (size_t)f + 1]; in.i2 = a.faces[3 * (size_t)f + 2];
    in.raw[0] = a._alpha[3 * p]; in.raw[1] = a._alpha[3 * p + 1]; in.raw[2] = a._alpha[3 * p + 2];
    in.sc = a._scale[p];
}
__device__ __forceinline__ void splat_from_inputs(const GmsMeshArgs &a, const SplatInputs &in, SplatParams &o);
__device__ __forceinline__ void splat_from_face(const GmsMeshArgs &a, int64_t p, SplatParams &o)
{
    SplatInputs in;
    splat_inputs_load(a, p, in);
    splat_from_inputs(a, in, o);
}
__device__ __forceinline__ void splat_from_inputs(const GmsMeshArgs &a, const SplatInputs &in, SplatParams &o)
{
#pragma clang fp contract(off)
    o.opacity = in.has_op ? 1.f / (1.f + expf(-in.op_raw)) : 0.f;
    const V3 t0 = ldv(a.vertices, (size_t)in.i0), t1 = ldv(a.vertices, (size_t)in.i1), t2 = ldv(a.vertices, (size_t)in.i2);
    const float raw[3] = {in.raw[0], in.raw[1], in.raw[2]};
    float al[3], rsum;
    barycentric(a.alpha_mode, raw, al, rsum);
    o.xyz[0] = al[0] * t0.x + al[1] * t1.x + al[2] * t2.x;
    o.xyz[1] = al[0] * t0.y + al[1] * t1.y + al[2] * t2.y;
    o.xyz[2] = al[0] * t0.z + al[1] * t1.z + al[2] * t2.z;
    Frame fr;
    face_frame(t0, t1, t2, fr);
    const float sc = in.sc;
    o.scale[0] = fmaxf(sc * EPS, 0.f) + EPS; o.scale[1] = fmaxf(sc * fr.s1, 0.f) + EPS; o.scale[2] = fmaxf(sc * fr.s2, 0.f) + EPS;
    float q[4];
    rot_to_quat(fr, q, nullptr);
    // torch.nn.functional.normalize: q / max(|q|, 1e-12)
    const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    o.q[0] = q[0] / n; o.q[1] = q[1] / n; o.q[2] = q[2] / n; o.q[3] = q[3] / n;
}

}  // namespace gms
