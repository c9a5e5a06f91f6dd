// adam.hip -- multi-tensor Adam step in one launch on gfx950.
//
// Replaces `gaussians.optimizer.step()` (train.py:147) for the optimizer the models build in training_setup():
// torch.optim.Adam(param_groups, lr=0.0, eps=1e-15) with one learning rate per group
// (games/mesh_splatting/scene/gaussian_mesh_model.py:174-183, scene/gaussian_model.py:149-160).  torch's default
// (foreach) implementation issues ~10 kernels per step over all tensors; here every tensor of every group is updated
// by ONE kernel: blocks are dealt to tensors through a prefix table passed by value.  Pure streaming, HBM-bound:
// 16 B read + 12 B written per element.
//
// Arithmetic follows torch/optim/adam.py `_single_tensor_adam` (no amsgrad, no weight decay, maximize=False):
//     m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g
//     p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with the scalar factors formed in double on the host, as the Python implementation does.
#include "gms_common.h"

namespace gms {

constexpr int ADAM_MAX_TENSORS = GMS_ADAM_MAX_TENSORS;
constexpr int ADAM_CHUNK = BLOCK * 4 * 4;          // elements per block: 4 float4 per thread

struct AdamTable {
    float *param[ADAM_MAX_TENSORS];
    const float *grad[ADAM_MAX_TENSORS];
    float *exp_avg[ADAM_MAX_TENSORS];
    float *exp_avg_sq[ADAM_MAX_TENSORS];
    int64_t n[ADAM_MAX_TENSORS];
    float step_size[ADAM_MAX_TENSORS];              // lr / bias_correction1
    float inv_bc2_sqrt[ADAM_MAX_TENSORS];           // 1 / sqrt(bias_correction2)
    uint32_t first_block[ADAM_MAX_TENSORS + 1];
    int count;
};

__device__ __forceinline__ void adam_update(float &p, float g, float &m, float &v, float w1, float beta2, float w2, float step_size,
                                            float bc2_sqrt, float eps)
{
    m = m + (g - m) * w1;
    v = v * beta2 + w2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(BLOCK) adam_kernel(AdamTable t, float w1, float beta2, float w2, float eps)
{
    int k = 0;
    while (k + 1 < t.count && blockIdx.x >= t.first_block[k + 1]) k++;
    const int64_t base = (int64_t)(blockIdx.x - t.first_block[k]) * ADAM_CHUNK;
    const int64_t n = t.n[k];
    float *__restrict__ P = t.param[k];
    const float *__restrict__ G = t.grad[k];
    float *__restrict__ M = t.exp_avg[k];
    float *__restrict__ V = t.exp_avg_sq[k];
    const float step_size = t.step_size[k], bc2_sqrt = 1.f / t.inv_bc2_sqrt[k];
    const bool vec = ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t i = base + ((int64_t)j * BLOCK + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            // The moments and the gradient (6 of the 7 floats that move per element) are touched by nobody but this kernel, once per step:
            // non-temporal, so that the 420 MB of them do not push the PARAMETERS -- which the next frame's preprocess launches read -- out of
            // the 256 MB Infinity Cache.  The parameters themselves go through the caches as before.
            typedef float v4f __attribute__((ext_vector_type(4)));
            float4 p = *(const float4 *)(P + i);
            const v4f mv = __builtin_nontemporal_load((const v4f *)(M + i)), vv = __builtin_nontemporal_load((const v4f *)(V + i));
            const v4f gv = __builtin_nontemporal_load((const v4f *)(G + i));
            float4 m = make_float4(mv.x, mv.y, mv.z, mv.w), v = make_float4(vv.x, vv.y, vv.z, vv.w);
            const float4 g = make_float4(gv.x, gv.y, gv.z, gv.w);
            adam_update(p.x, g.x, m.x, v.x, w1, beta2, w2, step_size, bc2_sqrt, eps);
            adam_update(p.y, g.y, m.y, v.y, w1, beta2, w2, step_size, bc2_sqrt, eps);
            adam_update(p.z, g.z, m.z, v.z, w1, beta2, w2, step_size, bc2_sqrt, eps);
            adam_update(p.w, g.w, m.w, v.w, w1, beta2, w2, step_size, bc2_sqrt, eps);
            *(float4 *)(P + i) = p;
            v4f mo, vo;
            mo.x = m.x; mo.y = m.y; mo.z = m.z; mo.w = m.w; vo.x = v.x; vo.y = v.y; vo.z = v.z; vo.w = v.w;
            __builtin_nontemporal_store(mo, (v4f *)(M + i)); __builtin_nontemporal_store(vo, (v4f *)(V + i));
        } else {
            for (int64_t e = i; e < n && e < i + 4; e++) {
                float p = P[e], m = M[e], v = V[e];
                adam_update(p, G[e], m, v, w1, beta2, w2, step_size, bc2_sqrt, eps);
                P[e] = p; M[e] = m; V[e] = v;
            }
        }
    }
}

}  // namespace gms

using namespace gms;

extern "C" int32_t gms_adam_step(const GmsAdamTensor *tensors, int32_t count, double beta1, double beta2, double eps, void *stream_)
{
    gms::TraceRange trace_range("gms_adam_step");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    if (count < 0 || (count > 0 && !tensors)) { set_error("gms_adam_step: invalid argument"); return GMS_ERR_INVALID_ARGUMENT; }
    // 1 - beta in double, then rounded once: what the Python scalars of torch/optim/adam.py become inside its kernels
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
    for (int32_t s = 0; s < count; s += ADAM_MAX_TENSORS) {
        AdamTable t;
        t.count = 0;
        uint64_t blocks = 0;
        for (int32_t i = s; i < count && i < s + ADAM_MAX_TENSORS; i++) {
            const GmsAdamTensor &a = tensors[i];
            if (a.n < 0 || a.step < 1 || (a.n > 0 && (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq))) {
                set_error("gms_adam_step: tensor %d: null pointer, negative size or step < 1", i);
                return GMS_ERR_INVALID_ARGUMENT;
            }
            if (a.n == 0) continue;
            const int k = t.count++;
            t.param[k] = a.param; t.grad[k] = a.grad; t.exp_avg[k] = a.exp_avg; t.exp_avg_sq[k] = a.exp_avg_sq; t.n[k] = a.n;
            const double bc1 = 1.0 - pow(beta1, (double)a.step), bc2 = 1.0 - pow(beta2, (double)a.step);
            t.step_size[k] = (float)((double)a.lr / bc1);
            t.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(bc2));
            t.first_block[k] = (uint32_t)blocks;
            blocks += (uint64_t)((a.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        if (t.count == 0) continue;
        t.first_block[t.count] = (uint32_t)blocks;
        if (blocks > 0x7fffffffull) { set_error("gms_adam_step: too many elements for one launch"); return GMS_ERR_INVALID_ARGUMENT; }
        GMS_LAUNCH(GMS_K_ADAM, stream, adam_kernel<<<(unsigned)blocks, BLOCK, 0, stream>>>(t, w1, (float)beta2, w2, (float)eps));
    }
    GMS_KERNEL_CHECK(0, stream, "adam");
    return GMS_OK;
}
