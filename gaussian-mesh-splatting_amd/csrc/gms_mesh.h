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

// ------------------------------------------------------------------ backward through the face frame (mesh_to_gaussians.hip, raster_backward.hip)
// what the splats of a face hand to the face: gradients w.r.t. the quaternion, the two tangent scales and (through alpha) the corners
struct FaceGrad { float dq[4]; float ds1, ds2; V3 dt0, dt1, dt2; };

// differentiate quaternion + frame once per face and scatter into the three vertices
// returns d loss / d (t0, t1, t2) of the face in out[9]
__device__ __forceinline__ void face_backward(const GmsMeshArgs &a, int f, const Frame &fr, FaceGrad &G, float out[9])
{
    // ---- quaternion -> dL/dR (columns v0, v1, v2)
    float q[4];
    QuatSel qs;
    rot_to_quat(fr, q, &qs);
    if (a.fused_activations) {   // gradient arrived w.r.t. q / |q|: project out the radial part, scale by 1/|q|
        const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
        const float u[4] = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
        const float d = u[0] * G.dq[0] + u[1] * G.dq[1] + u[2] * G.dq[2] + u[3] * G.dq[3];
#pragma unroll
        for (int k = 0; k < 4; k++) G.dq[k] = (G.dq[k] - u[k] * d) / n;
    }
    const float den = 2.0f * fmaxf(qs.a, 0.1f);
    float gc[4];
    float dden = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float gk = qs.sign * G.dq[k];
        gc[k] = gk / den;
        dden -= gk * qs.cand[k] / (den * den);
    }
    // a enters through den (if a > 0.1) and through cand[sel] = a^2
    const float gsel = qs.sel == 0 ? gc[0] : qs.sel == 1 ? gc[1] : qs.sel == 2 ? gc[2] : gc[3];
    float da = (qs.a > 0.1f ? 2.f * dden : 0.f) + 2.f * qs.a * gsel;
    const float dx = (qs.xsel > 0.f) ? da / (2.f * qs.a) : 0.f;   // a = sqrt(x), zero subgradient at x <= 0
    float d00 = 0, d01 = 0, d02 = 0, d10 = 0, d11 = 0, d12 = 0, d20 = 0, d21 = 0, d22 = 0;
    switch (qs.sel) {
    case 0:
        d00 += dx; d11 += dx; d22 += dx;
        d21 += gc[1]; d12 -= gc[1]; d02 += gc[2]; d20 -= gc[2]; d10 += gc[3]; d01 -= gc[3];
        break;
    case 1:
        d00 += dx; d11 -= dx; d22 -= dx;
        d21 += gc[0]; d12 -= gc[0]; d10 += gc[2]; d01 += gc[2]; d02 += gc[3]; d20 += gc[3];
        break;
    case 2:
        d00 -= dx; d11 += dx; d22 -= dx;
        d02 += gc[0]; d20 -= gc[0]; d10 += gc[1]; d01 += gc[1]; d12 += gc[3]; d21 += gc[3];
        break;
    default:
        d00 -= dx; d11 -= dx; d22 += dx;
        d10 += gc[0]; d01 -= gc[0]; d20 += gc[1]; d02 += gc[1]; d21 += gc[2]; d12 += gc[2];
        break;
    }
    // m[i][j] = v_j[i]
    V3 g0 = {d00, d10, d20}, g1 = {d01, d11, d21}, g2 = {d02, d12, d22};

    // ---- scales
    float g_v1n = G.ds1 * 0.5f;                       // s1 = v1n / 2
    V3 g_v2i = (G.ds2 * 0.5f) * fr.v2;                // s2 = <v2i, v2> / 2
    g2 = g2 + (G.ds2 * 0.5f) * fr.v2i;

    // ---- v2 = w / (|w| + eps)
    const float nwe = fr.nw + EPS;
    V3 gw = (1.f / nwe) * g2;
    if (fr.nw > 0.f) gw = gw - ((dot(g2, fr.w) / (nwe * nwe)) / fr.nw) * fr.w;
    // w = v2i - <v2i,v0> v0 - <v2i,v1> v1
    const float c0 = dot(fr.v2i, fr.v0), c1 = dot(fr.v2i, fr.v1);
    const float gw0 = dot(gw, fr.v0), gw1 = dot(gw, fr.v1);
    g_v2i = g_v2i + ((gw - gw0 * fr.v0) - gw1 * fr.v1);
    g0 = g0 - (c0 * gw + gw0 * fr.v2i);
    g1 = g1 - (c1 * gw + gw1 * fr.v2i);

    // ---- v1 = u1 / v1n, v1n = |u1| + eps
    V3 g_u1 = (1.f / fr.v1n) * g1;
    if (fr.n1 > 0.f) g_u1 = g_u1 + ((g_v1n - dot(g1, fr.u1) / (fr.v1n * fr.v1n)) / fr.n1) * fr.u1;

    // ---- v0 = N / (|N| + eps), N = (t1 - t0) x (t2 - t0)
    const float nNe = fr.nN + EPS;
    V3 gN = (1.f / nNe) * g0;
    if (fr.nN > 0.f) gN = gN - ((dot(g0, fr.N) / (nNe * nNe)) / fr.nN) * fr.N;
    const V3 e1 = fr.t1 - fr.t0, e2 = fr.t2 - fr.t0;
    const V3 g_e1 = cross(e2, gN), g_e2 = cross(gN, e1);

    // ---- back to the triangle
    const V3 g_mean = (-1.f / 3.f) * (g_u1 + g_v2i);
    V3 dt0 = G.dt0 + g_mean - (g_e1 + g_e2);
    V3 dt1 = G.dt1 + g_mean + g_u1 + g_e1;
    V3 dt2 = G.dt2 + g_mean + g_v2i + g_e2;
    out[0] = dt0.x; out[1] = dt0.y; out[2] = dt0.z;
    out[3] = dt1.x; out[4] = dt1.y; out[5] = dt1.z;
    out[6] = dt2.x; out[7] = dt2.y; out[8] = dt2.z;
}


// Everything the mesh backward does for ONE splat whose incoming gradients are in REGISTERS (round 6: the tail of preprocess_bwd when the
// frame was rendered straight from the mesh -- no dL/dxyz / dL/dscale / dL/drot / dL/dopacity tensors, no mesh_bwd launch).  fused_activations
// semantics (gradients w.r.t. exp / normalize / sigmoid outputs).  The face's part is LINEAR in what its splats hand it, so every splat
// backpropagates its own share through the frame and adds it to the three corners' gradients (9 float atomics per splat instead of 9 per
// face: taken for <= 4 splats per face; dL_dvertices must be all zero on entry).  The per-splat outputs are plain stores, the formulas those
// of bwd_splat_body / splat_contrib.
__device__ __forceinline__ void splat_backward_from_registers(const GmsMeshArgs &a, int64_t p, const float g_xyz[3], const float g_scale[3],
                                                              const float g_rot[4], float g_opac, float *dL_dvertices, float *dL_dalpha,
                                                              float *dL_dscale, float *dL_d_opacity)
{
    {   // sigmoid backward: g * (1 - y) * y
        const float y = 1.f / (1.f + expf(-a._opacity[p]));
        dL_d_opacity[p] = g_opac * (1.f - y) * y;
    }
    const int f = splat_to_face(a, p);
    const int64_t i0 = a.faces[3 * (size_t)f], i1 = a.faces[3 * (size_t)f + 1], i2 = a.faces[3 * (size_t)f + 2];
    const V3 t0 = ldv(a.vertices, (size_t)i0), t1 = ldv(a.vertices, (size_t)i1), t2 = ldv(a.vertices, (size_t)i2);
    const float raw[3] = {a._alpha[3 * p], a._alpha[3 * p + 1], a._alpha[3 * p + 2]};
    float al[3], rsum;
    barycentric(a.alpha_mode, raw, al, rsum);
    const V3 g = {g_xyz[0], g_xyz[1], g_xyz[2]};
    const float da[3] = {dot(g, t0), dot(g, t1), dot(g, t2)};
    const float s = da[0] * al[0] + da[1] * al[1] + da[2] * al[2];
    if (a.alpha_mode == GMS_ALPHA_RELU) {
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dalpha[3 * p + k] = raw[k] > 0.f ? (da[k] - s) / rsum : 0.f;
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dalpha[3 * p + k] = al[k] * (da[k] - s);
    }
    Frame fr;
    face_frame(t0, t1, t2, fr);
    const float sc = a._scale[p];
    const float sj[3] = {EPS, fr.s1, fr.s2};
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (sc * sj[j] > 0.f) acc += g_scale[j] * sj[j];
    dL_dscale[p] = acc;
    FaceGrad G = {};
    G.dt0 = al[0] * g; G.dt1 = al[1] * g; G.dt2 = al[2] * g;
    G.dq[0] = g_rot[0]; G.dq[1] = g_rot[1]; G.dq[2] = g_rot[2]; G.dq[3] = g_rot[3];
    if (sc * fr.s1 > 0.f) G.ds1 = g_scale[1] * sc;
    if (sc * fr.s2 > 0.f) G.ds2 = g_scale[2] * sc;
    float out[9];
    face_backward(a, f, fr, G, out);
    // The splats of a face are consecutive lanes (uniform splat count, <= 4).  The FIRST lane of a face's run inside this wave collects
    // the run's nine sums (9 atomics per splat on the same three corners in the same instruction: 146 us against 56 for the two
    // launches; one set per run: 59).  A face split between two waves leaves two partial runs: the sum is linear.  Where the run is a
    // whole face of three splats, the sums go back out so that the run's three lanes add x, y, z of ONE corner per instruction -- one
    // cache-line operation at the L2 instead of three, which is what these atomics cost (without them: 44 us).
    // (ds_bpermute: every lane of the wave takes part; the caller's only exit before this point is `i >= P`.)
    const int lane = (int)(threadIdx.x & 63);
    const int S = a.splats_per_face;                     // (the caller admits 1 .. 4 only)
    const int in_face = (int)(p % S);
    const int r = in_face < lane ? in_face : lane;       // position inside the run (a run that began in the previous wave starts at lane 0)
    int tail = S - 1 - in_face;                          // lanes of the run after this one
    if (tail > 63 - lane) tail = 63 - lane;
    if ((int64_t)tail > a.P - 1 - p) tail = (int)(a.P - 1 - p);
    float sum[9];
#pragma unroll
    for (int k = 0; k < 9; k++) sum[k] = out[k];
#pragma unroll
    for (int d = 1; d <= 3; d++) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float o = __shfl_down(out[k], d);
            if (d <= tail) sum[k] += o;                  // (only the run's first lane, which sees all of it, uses its sum)
        }
    }
    const bool whole3 = S == 3 && r + tail == 2;
    const int64_t vi[3] = {i0, i1, i2};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float v1 = __shfl_up(sum[3 * k + 1], 1), v2 = __shfl_up(sum[3 * k + 2], 2);
        const float v = r == 0 ? sum[3 * k] : r == 1 ? v1 : v2;
        if (whole3 && v != 0.f) unsafeAtomicAdd(dL_dvertices + 3 * (size_t)vi[k] + r, v);
    }
    if (whole3 || r != 0) return;
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            if (sum[3 * k + c] != 0.f) unsafeAtomicAdd(dL_dvertices + 3 * (size_t)vi[k] + c, sum[3 * k + c]);
}

}  // namespace gms
