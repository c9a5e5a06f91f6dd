// mesh_to_gaussians.hip -- fused mesh-face -> Gaussian parameterization (K0) for gfx950.
//
// Replaces the ~40 elementwise/gather PyTorch kernels (and 2-3 host syncs from boolean-mask
// indexing) of GaussianMeshModel.update_alpha + _calc_xyz + prepare_scaling_rot
// (games/mesh_splatting/scene/gaussian_mesh_model.py:86-169) and rot_to_quat_batch
// (utils/general_utils.py:43-96) with one forward kernel and two backward kernels:
//   mesh_fwd        1 thread / splat: barycentric centre, face frame (recomputed per splat -- the
//                   three vertices are L1/L2 hits for the face's other splats), log-scales,
//                   rotation-matrix -> quaternion.  HBM-bound: reads 16 B, writes 52 B per splat.
//   mesh_bwd_splat  1 thread / splat: d_alpha (through relu+L1-normalise or softmax), d_scale.
//   mesh_bwd_face   per face (1 thread, or 1 wave when the face carries >= 16 splats): sums the
//                   splats' quaternion / scale / centre gradients, differentiates quaternion
//                   selection, Gram-Schmidt frame, norms and the cross product once per face and
//                   scatters into vertices.grad with 9 float atomics per face.
// Operation order follows the reference line by line; contraction is off so the discrete
// argmax in the quaternion conversion sees the same values as a float32 CPU evaluation.
#include "gms_common.h"
#include "gms_mesh.h"

namespace gms {

__global__ void __launch_bounds__(BLOCK) mesh_fwd_kernel(GmsMeshArgs a, float *alpha_out, float *xyz, float *scaling,
                                                         float *rotation, float *scaling_act, float *rotation_unit,
                                                         float *opacity_act, unsigned splat_blocks)
{
#pragma clang fp contract(off)
    if (blockIdx.x >= splat_blocks) {             // ride-along blocks: clear the scratch the backward will accumulate into
        const int64_t i = ((int64_t)(blockIdx.x - splat_blocks) * BLOCK + threadIdx.x) * 4;
        for (int k = 0; k < 4; k++)
            if (i + k < a.prezero_count) a.prezero[i + k] = 0.f;
        return;
    }
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= a.P) return;
    if (opacity_act) opacity_act[p] = 1.f / (1.f + expf(-a._opacity[p]));      // get_opacity: torch.sigmoid
    const int f = splat_to_face(a, p);
    V3 t0, t1, t2;
    load_face(a, f, t0, t1, t2);
    const float raw[3] = {a._alpha[3 * p], a._alpha[3 * p + 1], a._alpha[3 * p + 2]};
    float al[3], rsum;
    barycentric(a.alpha_mode, raw, al, rsum);
    if (alpha_out) { alpha_out[3 * p] = al[0]; alpha_out[3 * p + 1] = al[1]; alpha_out[3 * p + 2] = al[2]; }
    xyz[3 * p] = al[0] * t0.x + al[1] * t1.x + al[2] * t2.x;
    xyz[3 * p + 1] = al[0] * t0.y + al[1] * t1.y + al[2] * t2.y;
    xyz[3 * p + 2] = al[0] * t0.z + al[1] * t1.z + al[2] * t2.z;
    Frame fr;
    face_frame(t0, t1, t2, fr);
    const float sc = a._scale[p];
    const float act0 = fmaxf(sc * EPS, 0.f) + EPS, act1 = fmaxf(sc * fr.s1, 0.f) + EPS, act2 = fmaxf(sc * fr.s2, 0.f) + EPS;
    scaling[3 * p] = logf(act0);
    scaling[3 * p + 1] = logf(act1);
    scaling[3 * p + 2] = logf(act2);
    float q[4];
    rot_to_quat(fr, q, nullptr);
    *reinterpret_cast<float4 *>(rotation + 4 * p) = make_float4(q[0], q[1], q[2], q[3]);
    if (scaling_act) { scaling_act[3 * p] = act0; scaling_act[3 * p + 1] = act1; scaling_act[3 * p + 2] = act2; }
    if (rotation_unit) {   // torch.nn.functional.normalize: q / max(|q|, 1e-12)
        const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
        *reinterpret_cast<float4 *>(rotation_unit + 4 * p) = make_float4(q[0] / n, q[1] / n, q[2] / n, q[3] / n);
    }
}

// ------------------------------------------------------------------ backward, per splat
__device__ __forceinline__ void bwd_splat_body(const GmsMeshArgs &a, unsigned block, const float *dL_dxyz, const float *dL_dscaling,
                                               float *dL_dalpha, float *dL_dscale, const float *dL_dopacity_act, float *dL_d_opacity)
{
    const int64_t p = (int64_t)block * BLOCK + threadIdx.x;
    if (p >= a.P) return;
    if (dL_d_opacity) {                             // sigmoid backward: g * (1 - y) * y
        const float y = 1.f / (1.f + expf(-a._opacity[p]));
        dL_d_opacity[p] = dL_dopacity_act[p] * (1.f - y) * y;
    }
    const int f = splat_to_face(a, p);
    V3 t0, t1, t2;
    load_face(a, f, t0, t1, t2);
    const float raw[3] = {a._alpha[3 * p], a._alpha[3 * p + 1], a._alpha[3 * p + 2]};
    float al[3], rsum;
    barycentric(a.alpha_mode, raw, al, rsum);
    const V3 g = ldv(dL_dxyz, (size_t)p);
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
    for (int j = 0; j < 3; j++) {
        const float u = sc * sj[j];
        if (u > 0.f) acc += dL_dscaling[3 * p + j] * sj[j] * (a.fused_activations ? 1.f : 1.f / (u + EPS));
    }
    dL_dscale[p] = acc;
}

__global__ void __launch_bounds__(BLOCK) mesh_bwd_splat_kernel(GmsMeshArgs a, const float *dL_dxyz, const float *dL_dscaling,
                                                               float *dL_dalpha, float *dL_dscale, float *dL_dvertices,
                                                               const float *dL_dopacity_act, float *dL_d_opacity)
{
    // the vertex gradient is accumulated with atomics by the face kernel that follows on the stream: clear it here
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < 3 * (int64_t)a.V; e += (int64_t)gridDim.x * BLOCK)
        dL_dvertices[e] = 0.f;
    bwd_splat_body(a, blockIdx.x, dL_dxyz, dL_dscaling, dL_dalpha, dL_dscale, dL_dopacity_act, dL_d_opacity);
}

// ------------------------------------------------------------------ backward, per face
__device__ __forceinline__ void splat_contrib(const GmsMeshArgs &a, int64_t p, const Frame &fr, const float *dL_dxyz,
                                              const float *dL_dscaling, const float *dL_drot, FaceGrad &G)
{
    const float raw[3] = {a._alpha[3 * p], a._alpha[3 * p + 1], a._alpha[3 * p + 2]};
    float al[3], rsum;
    barycentric(a.alpha_mode, raw, al, rsum);
    const V3 g = ldv(dL_dxyz, (size_t)p);
    G.dt0 = G.dt0 + al[0] * g; G.dt1 = G.dt1 + al[1] * g; G.dt2 = G.dt2 + al[2] * g;
    const float4 gq = *reinterpret_cast<const float4 *>(dL_drot + 4 * p);
    G.dq[0] += gq.x; G.dq[1] += gq.y; G.dq[2] += gq.z; G.dq[3] += gq.w;
    const float sc = a._scale[p];
    const float u1 = sc * fr.s1, u2 = sc * fr.s2;
    if (u1 > 0.f) G.ds1 += dL_dscaling[3 * p + 1] * sc * (a.fused_activations ? 1.f : 1.f / (u1 + EPS));
    if (u2 > 0.f) G.ds2 += dL_dscaling[3 * p + 2] * sc * (a.fused_activations ? 1.f : 1.f / (u2 + EPS));
}

__device__ __forceinline__ void face_splat_range(const GmsMeshArgs &a, int f, int64_t &b, int64_t &e)
{
    if (a.splats_per_face > 0) { b = (int64_t)f * a.splats_per_face; e = b + a.splats_per_face; }
    else { b = a.face_splat_offset[f]; e = a.face_splat_offset[f + 1]; }
}

// few splats per face: one thread per face
// `corner_grad` != NULL (deterministic mode, gmsplat.h): the nine values of a face are STORED at corner_grad[9 f ..] and summed
// per vertex, in ascending corner order, by det_vertex_gather_kernel -- no float atomics.
__device__ __forceinline__ void bwd_face_thread_body(const GmsMeshArgs &a, unsigned block, const float *dL_dxyz, const float *dL_dscaling,
                                                     const float *dL_drot, float *dL_dvertices, float *lds9, int *ldsi, float *corner_grad = nullptr)
{
    const int f = (int)(block * BLOCK + threadIdx.x);
    const bool valid = f < a.F;
    float out[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (valid) {
    V3 t0, t1, t2;
    load_face(a, f, t0, t1, t2);
    Frame fr;
    face_frame(t0, t1, t2, fr);
    FaceGrad G = {};
    int64_t b, e;
    face_splat_range(a, f, b, e);
    for (int64_t p = b; p < e; p++) splat_contrib(a, p, fr, dL_dxyz, dL_dscaling, dL_drot, G);
    face_backward(a, f, fr, G, out);
    }
    if (corner_grad) {
        if (valid) {
#pragma unroll
            for (int k = 0; k < 9; k++) corner_grad[9 * (size_t)f + k] = out[k];
        }
        return;
    }
    // Scatter into vertices.grad: the wave's 64 x 9 values are transposed through LDS so that three adjacent
    // lanes add the x, y, z of ONE vertex (same cache line -> one L2 transaction instead of three).
    float *w9 = lds9 + (threadIdx.x >> 6) * (WAVE * 9);
    int *wi = ldsi + (threadIdx.x >> 6) * (WAVE * 3);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 9; k++) w9[lane * 9 + k] = out[k];
#pragma unroll
    for (int k = 0; k < 3; k++) wi[lane * 3 + k] = valid ? (int)a.faces[3 * (size_t)f + k] : -1;
    wave_sync();                                   // w9 / wi are this wave's own
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int e = j * WAVE + lane;          // e = face_in_wave * 9 + vertex * 3 + component
        const int fl = e / 9, r = e - fl * 9;
        const int vi = wi[fl * 3 + r / 3];
        if (vi >= 0) unsafeAtomicAdd(dL_dvertices + 3 * (size_t)vi + (r % 3), w9[e]);
    }
}

__global__ void __launch_bounds__(BLOCK) mesh_bwd_face_thread_kernel(GmsMeshArgs a, const float *dL_dxyz, const float *dL_dscaling,
                                                                     const float *dL_drot, float *dL_dvertices, float *corner_grad)
{
    __shared__ float lds9[4 * WAVE * 9];
    __shared__ int ldsi[4 * WAVE * 3];
    bwd_face_thread_body(a, blockIdx.x, dL_dxyz, dL_dscaling, dL_drot, dL_dvertices, lds9, ldsi, corner_grad);
}

// One launch for the whole backward when the vertex gradient buffer was cleared ahead of time (by the forward's
// ride-along blocks): the per-splat and the per-face parts are independent then and share the grid.
__global__ void __launch_bounds__(BLOCK) mesh_bwd_fused_kernel(GmsMeshArgs a, const float *dL_dxyz, const float *dL_dscaling,
                                                               const float *dL_drot, float *dL_dvertices, float *dL_dalpha,
                                                               float *dL_dscale, const float *dL_dopacity_act, float *dL_d_opacity,
                                                               unsigned face_blocks)
{
    __shared__ float lds9[4 * WAVE * 9];
    __shared__ int ldsi[4 * WAVE * 3];
    // face blocks first: they are the longer ones
    if (blockIdx.x < face_blocks) bwd_face_thread_body(a, blockIdx.x, dL_dxyz, dL_dscaling, dL_drot, dL_dvertices, lds9, ldsi);
    else bwd_splat_body(a, blockIdx.x - face_blocks, dL_dxyz, dL_dscaling, dL_dalpha, dL_dscale, dL_dopacity_act, dL_d_opacity);
}

// many splats per face (FLAME-like, 50-100): one wave per face, lanes stride over the splats
__global__ void __launch_bounds__(BLOCK) mesh_bwd_face_wave_kernel(GmsMeshArgs a, const float *dL_dxyz, const float *dL_dscaling,
                                                                   const float *dL_drot, float *dL_dvertices, float *corner_grad)
{
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (f >= a.F) return;                         // wave-uniform
    V3 t0, t1, t2;
    load_face(a, f, t0, t1, t2);
    Frame fr;
    face_frame(t0, t1, t2, fr);
    FaceGrad G = {};
    int64_t b, e;
    face_splat_range(a, f, b, e);
    for (int64_t p = b + lane; p < e; p += WAVE) splat_contrib(a, p, fr, dL_dxyz, dL_dscaling, dL_drot, G);
    float *v = reinterpret_cast<float *>(&G);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(FaceGrad) / 4); k++) v[k] = wave_sum_to_lane63(v[k]);
    if (lane == 63) {
        float out[9];
        face_backward(a, f, fr, G, out);
        if (corner_grad) {          // deterministic mode: stored, summed per vertex by det_vertex_gather_kernel
            for (int k = 0; k < 9; k++) corner_grad[9 * (size_t)f + k] = out[k];
            return;
        }
        for (int k = 0; k < 3; k++) {
            const int64_t vi = a.faces[3 * (size_t)f + k];
            for (int c = 0; c < 3; c++) unsafeAtomicAdd(dL_dvertices + 3 * vi + c, out[3 * k + c]);
        }
    }
}

// ------------------------------------------------------------------ deterministic mode: vertex gradient without float atomics
// vertex -> incident corners (corner c = 3 f + k) as a CSR built per call: integer counts, one-block exclusive scan, fill through
// integer cursors (slot ORDER is arbitrary), then each vertex adds its corners' stored gradients in ASCENDING corner index --
// found by repeated selection of the next larger index, O(degree^2) loads (degree ~ 6; the two poles of a UV sphere: hundreds).
__global__ void __launch_bounds__(BLOCK) det_corner_count_kernel(int64_t n_corners, const int64_t *faces, uint32_t *cnt)
{
    const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (c < n_corners) atomicAdd(&cnt[faces[c]], 1u);
}

__global__ void __launch_bounds__(1024) det_scan_kernel(const uint32_t *cnt, uint32_t *off, int n)
{
    __shared__ uint32_t wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + 1023) / 1024;
    const int b = min(n, tid * per), e = min(n, b + per);
    uint32_t sum = 0;
    for (int k = b; k < e; k++) sum += cnt[k];
    uint32_t run = sum;
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)run, d);
        if (lane >= d) run += x;
    }
    if (lane == WAVE - 1) wave_tot[wave] = run;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    uint32_t pre = base + run - sum;                 // exclusive prefix of this thread's chunk
    for (int k = b; k < e; k++) { off[k] = pre; pre += cnt[k]; }
    if (tid == 1023) off[n] = pre;
}

__global__ void __launch_bounds__(BLOCK) det_corner_fill_kernel(int64_t n_corners, const int64_t *faces, const uint32_t *off, uint32_t *cur, uint32_t *adj)
{
    const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n_corners) return;
    const int64_t v = faces[c];
    adj[off[v] + atomicAdd(&cur[v], 1u)] = (uint32_t)c;
}

constexpr int DET_SMALL_DEGREE = 32;       // up to here one thread sorts-by-selection (O(d^2) loads); above, a wave ranks the list
constexpr int DET_WAVE_CAP = 2048;         // corners a wave can rank through its LDS slice; beyond: serial selection by one lane

__global__ void __launch_bounds__(BLOCK) det_vertex_gather_kernel(int V, const uint32_t *off, const uint32_t *adj, const float *corner_grad, float *dL_dvertices)
{
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= V) return;
    const uint32_t b = off[v], e = off[v + 1];
    if (e - b > (uint32_t)DET_SMALL_DEGREE) return;          // det_vertex_gather_wave_kernel's
    float gx = 0.f, gy = 0.f, gz = 0.f;
    int64_t last = -1;
    for (uint32_t n = b; n < e; n++) {
        int64_t next = INT64_MAX;
        for (uint32_t k = b; k < e; k++) { const int64_t c = adj[k]; if (c > last && c < next) next = c; }
        gx += corner_grad[3 * next]; gy += corner_grad[3 * next + 1]; gz += corner_grad[3 * next + 2];
        last = next;
    }
    dL_dvertices[3 * (size_t)v] = gx; dL_dvertices[3 * (size_t)v + 1] = gy; dL_dvertices[3 * (size_t)v + 2] = gz;
}

// High-degree vertices (the two poles of a UV sphere touch n_lon faces: 224 at the headline size -- one thread's O(d^2) selection
// took 4 ms there): one wave per vertex.  Every lane ranks its entries (rank = number of smaller corner indices: the indices of
// a vertex are distinct), the wave writes the list in rank order into its LDS slice, lane 0 adds the gradients in that order.
__global__ void __launch_bounds__(BLOCK) det_vertex_gather_wave_kernel(int V, const uint32_t *off, const uint32_t *adj, const float *corner_grad, float *dL_dvertices)
{
    __shared__ uint32_t sorted_all[4][DET_WAVE_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = blockIdx.x * (BLOCK / WAVE) + wave;
    if (v >= V) return;                                       // wave-uniform
    const uint32_t b = off[v], e = off[v + 1], d = e - b;
    if (d <= (uint32_t)DET_SMALL_DEGREE) return;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (d <= (uint32_t)DET_WAVE_CAP) {
        uint32_t *sorted = sorted_all[wave];
        for (uint32_t i = lane; i < d; i += WAVE) {
            const uint32_t c = adj[b + i];
            uint32_t rank = 0;
            for (uint32_t k = 0; k < d; k++) rank += adj[b + k] < c ? 1u : 0u;
            sorted[rank] = c;
        }
        wave_sync();
        if (lane == 0)
            for (uint32_t n = 0; n < d; n++) {
                const size_t c = sorted[n];
                gx += corner_grad[3 * c]; gy += corner_grad[3 * c + 1]; gz += corner_grad[3 * c + 2];
            }
    } else if (lane == 0) {
        int64_t last = -1;
        for (uint32_t n = b; n < e; n++) {
            int64_t next = INT64_MAX;
            for (uint32_t k = b; k < e; k++) { const int64_t c = adj[k]; if (c > last && c < next) next = c; }
            gx += corner_grad[3 * next]; gy += corner_grad[3 * next + 1]; gz += corner_grad[3 * next + 2];
            last = next;
        }
    }
    if (lane == 0) { dL_dvertices[3 * (size_t)v] = gx; dL_dvertices[3 * (size_t)v + 1] = gy; dL_dvertices[3 * (size_t)v + 2] = gz; }
}

// shared with the fused mesh input of gms_rasterize_forward (raster_forward.hip), which reads the same tables in splat_from_face
int32_t check_mesh_args(const GmsMeshArgs *A, bool need_face_offsets)
{
    if (!A || A->F < 0 || A->P < 0 || A->V < 0) { set_error("mesh args: negative size"); return GMS_ERR_INVALID_ARGUMENT; }
    if (A->P == 0) return GMS_OK;
    if (!A->vertices || !A->faces || !A->_alpha || !A->_scale) { set_error("mesh args: null input"); return GMS_ERR_INVALID_ARGUMENT; }
    if (A->splats_per_face > 0) {
        if ((int64_t)A->F * A->splats_per_face != A->P) { set_error("mesh args: P != F * splats_per_face"); return GMS_ERR_INVALID_ARGUMENT; }
    } else if ((need_face_offsets && !A->face_splat_offset) || !A->splat_face) {
        set_error("mesh args: non-uniform splat counts need face_splat_offset and splat_face");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    if (A->alpha_mode != GMS_ALPHA_RELU && A->alpha_mode != GMS_ALPHA_SOFTMAX) { set_error("mesh args: bad alpha_mode"); return GMS_ERR_INVALID_ARGUMENT; }
    return GMS_OK;
}

}  // namespace gms

using namespace gms;

extern "C" int32_t gms_mesh_to_gaussians_forward(const GmsMeshArgs *A, float *alpha, float *xyz, float *scaling,
                                                 float *rotation, float *scaling_act, float *rotation_unit,
                                                 float *opacity_act, void *stream_)
{
    gms::TraceRange trace_range("gms_mesh_to_gaussians_forward");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    int32_t rc = check_mesh_args(A);
    if (rc != GMS_OK) return rc;
    if (A->P == 0) return GMS_OK;
    if (!xyz || !scaling || !rotation) { set_error("mesh forward: null output"); return GMS_ERR_INVALID_ARGUMENT; }
    if (opacity_act && !A->_opacity) { set_error("mesh forward: opacity_activated requested without _opacity"); return GMS_ERR_INVALID_ARGUMENT; }
    const unsigned sblocks = (unsigned)((A->P + BLOCK - 1) / BLOCK);
    const unsigned zblocks = (A->prezero && A->prezero_count > 0) ? (unsigned)((A->prezero_count + 4 * BLOCK - 1) / (4 * BLOCK)) : 0u;
    GMS_LAUNCH(GMS_K_MESH_FWD, stream, mesh_fwd_kernel<<<sblocks + zblocks, BLOCK, 0, stream>>>(*A, alpha, xyz, scaling, rotation, scaling_act, rotation_unit, opacity_act, sblocks));
    GMS_KERNEL_CHECK(0, stream, "mesh_fwd");
    return GMS_OK;
}

extern "C" int32_t gms_mesh_to_gaussians_backward(const GmsMeshArgs *A, const float *dL_dxyz, const float *dL_dscaling,
                                                  const float *dL_drotation, const float *dL_dopacity_act,
                                                  float *dL_dvertices, float *dL_dalpha, float *dL_dscale,
                                                  float *dL_d_opacity, void *stream_)
{
    gms::TraceRange trace_range("gms_mesh_to_gaussians_backward");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    int32_t rc = check_mesh_args(A);
    if (rc != GMS_OK) return rc;
    if (A->P == 0) return GMS_OK;
    if (!dL_dxyz || !dL_dscaling || !dL_drotation || !dL_dvertices || !dL_dalpha || !dL_dscale) {
        set_error("mesh backward: null gradient pointer");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    if (dL_d_opacity && (!A->_opacity || !dL_dopacity_act)) {
        set_error("mesh backward: dL_d_opacity requested without _opacity / dL_dopacity_activated");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    const double avg_splats = (double)A->P / (double)(A->F > 0 ? A->F : 1);
    if (det_mode()) {
        // deterministic mode (gmsplat.h): per-corner gradients stored, then summed per vertex in ascending corner index
        const int64_t nc = 3 * (int64_t)A->F;
        float *corner_grad = static_cast<float *>(det_scratch(1, (size_t)nc * 3 * sizeof(float), stream));
        uint32_t *cnt = static_cast<uint32_t *>(det_scratch(2, ((size_t)A->V * 3 + 2) * sizeof(uint32_t), stream));
        uint32_t *adj = static_cast<uint32_t *>(det_scratch(3, (size_t)(nc > 0 ? nc : 1) * sizeof(uint32_t), stream));
        if (!corner_grad || !cnt || !adj) { set_error("deterministic mode: scratch allocation failed"); return GMS_ERR_ALLOC; }
        uint32_t *cur = cnt + A->V, *off = cnt + 2 * (size_t)A->V;
        GMS_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)A->V * 2 * sizeof(uint32_t), stream));
        GMS_LAUNCH(GMS_K_MESH_BWD_SPLAT, stream, mesh_bwd_splat_kernel<<<(unsigned)((A->P + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(*A, dL_dxyz, dL_dscaling, dL_dalpha, dL_dscale, dL_dvertices, dL_dopacity_act, dL_d_opacity));
        if (avg_splats >= 16.0) {
            const int fpb = BLOCK / WAVE;
            GMS_LAUNCH(GMS_K_MESH_BWD_FACE, stream, mesh_bwd_face_wave_kernel<<<(unsigned)((A->F + fpb - 1) / fpb), BLOCK, 0, stream>>>(*A, dL_dxyz, dL_dscaling, dL_drotation, dL_dvertices, corner_grad));
        } else {
            GMS_LAUNCH(GMS_K_MESH_BWD_FACE, stream, mesh_bwd_face_thread_kernel<<<(unsigned)((A->F + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(*A, dL_dxyz, dL_dscaling, dL_drotation, dL_dvertices, corner_grad));
        }
        if (nc > 0) {
            det_corner_count_kernel<<<(unsigned)((nc + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(nc, A->faces, cnt);
            det_scan_kernel<<<1, 1024, 0, stream>>>(cnt, off, A->V);
            det_corner_fill_kernel<<<(unsigned)((nc + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(nc, A->faces, off, cur, adj);
        } else {
            GMS_HIP_CHECK(hipMemsetAsync(off, 0, ((size_t)A->V + 1) * sizeof(uint32_t), stream));
        }
        det_vertex_gather_kernel<<<(unsigned)((A->V + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(A->V, off, adj, corner_grad, dL_dvertices);
        det_vertex_gather_wave_kernel<<<(unsigned)((A->V + 3) / 4), BLOCK, 0, stream>>>(A->V, off, adj, corner_grad, dL_dvertices);
        GMS_KERNEL_CHECK(0, stream, "mesh_bwd (deterministic)");
        return GMS_OK;
    }
    if (A->vertex_grad_prezeroed && avg_splats < 16.0) {
        const unsigned fb = (unsigned)((A->F + BLOCK - 1) / BLOCK), sb = (unsigned)((A->P + BLOCK - 1) / BLOCK);
        GMS_LAUNCH(GMS_K_MESH_BWD_FACE, stream, mesh_bwd_fused_kernel<<<fb + sb, BLOCK, 0, stream>>>(*A, dL_dxyz, dL_dscaling, dL_drotation, dL_dvertices,
                                                                                                   dL_dalpha, dL_dscale, dL_dopacity_act, dL_d_opacity, fb));
        GMS_KERNEL_CHECK(0, stream, "mesh_bwd_fused");
        return GMS_OK;
    }
    GMS_LAUNCH(GMS_K_MESH_BWD_SPLAT, stream, mesh_bwd_splat_kernel<<<(unsigned)((A->P + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(*A, dL_dxyz, dL_dscaling, dL_dalpha, dL_dscale, dL_dvertices, dL_dopacity_act, dL_d_opacity));
    GMS_KERNEL_CHECK(0, stream, "mesh_bwd_splat");
    const double avg = (double)A->P / (double)(A->F > 0 ? A->F : 1);
    if (avg >= 16.0) {
        const int fpb = BLOCK / WAVE;
        GMS_LAUNCH(GMS_K_MESH_BWD_FACE, stream, mesh_bwd_face_wave_kernel<<<(unsigned)((A->F + fpb - 1) / fpb), BLOCK, 0, stream>>>(*A, dL_dxyz, dL_dscaling, dL_drotation, dL_dvertices, nullptr));
    } else {
        GMS_LAUNCH(GMS_K_MESH_BWD_FACE, stream, mesh_bwd_face_thread_kernel<<<(unsigned)((A->F + BLOCK - 1) / BLOCK), BLOCK, 0, stream>>>(*A, dL_dxyz, dL_dscaling, dL_drotation, dL_dvertices, nullptr));
    }
    GMS_KERNEL_CHECK(0, stream, "mesh_bwd_face");
    return GMS_OK;
}
