// loss.hip -- fused L1 + SSIM photometric loss, forward and backward, on gfx950.
//
// Replaces the per-iteration loss of train.py:106-107,
//     loss = (1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt))
// i.e. utils/loss_utils.py:17-18 (mean |x-y|) and :33-63 (SSIM with an 11x11 Gaussian window, sigma 1.5, zero
// padding, C1 = 0.01^2, C2 = 0.03^2, mean over all pixels and channels).  The reference runs five grouped 11x11
// convolutions + ~20 elementwise kernels forward and their autograd mirror backward; here one kernel per direction:
//
//   forward   a 32x32 output tile per block: both images' 42x42 halo tiles go to LDS once, the separable window is
//             applied horizontally then vertically to the five moments (x, y, x^2, y^2, xy) out of LDS, the SSIM
//             value and its three partial derivatives w.r.t. the window moments (mu1, E[x^2], E[xy]) are formed in
//             registers; the derivative maps are stored (3 floats/pixel) and the block's L1 / SSIM sums go to a
//             partials array that a one-block kernel adds in a fixed order (deterministic).
//   backward  dSSIMsum/dx(p) = conv(dS/dmu1)(p) + 2 x(p) conv(dS/dE[x^2])(p) + y(p) conv(dS/dE[xy])(p): the same
//             tile/halo/separable structure on the three stored maps, plus the sign(x-y) term of L1.
//
// HBM-bound: forward reads 2 and writes 3 floats per pixel-channel, backward reads 5 and writes 1.
#include "gms_common.h"

namespace gms {

constexpr int LT = 32;            // output tile edge
constexpr int LR = 5;             // window radius (11 taps)
constexpr int LH = LT + 2 * LR;   // halo tile edge
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

struct SsimWindow { float w[2 * LR + 1]; };

// utils/loss_utils.py:23-25 executed with torch CPU float32: exp(-(i-5)^2/(2*1.5^2)) / sum.  The eleven values are
// pinned bit-for-bit (hex floats) because SSIM is sensitive to the window's normalisation at the 1e-8 level:
// sigma = E[x^2] - mu^2 mixes the first and second power of the weight sum, so a one-ulp difference in how the
// float32 sum is formed (torch reduces pairwise, a sequential loop does not) shifts mean SSIM by ~2e-6.
static SsimWindow make_window()
{
    static const float k[2 * LR + 1] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.10656p-2f,
                                        0x1.b43c3ep-3f, 0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d956cp-10f};
    SsimWindow win;
    for (int i = 0; i < 2 * LR + 1; i++) win.w[i] = k[i];
    return win;
}

__device__ __forceinline__ float block_sum(float v, float *red /* [4] */)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(BLOCK) l1_ssim_fwd_kernel(int H, int W, const float *__restrict__ img, const float *__restrict__ gt,
                                                            SsimWindow win, float *__restrict__ dmaps, size_t map_stride,
                                                            float *__restrict__ partials)
{
    __shared__ float sx[LH][LH + 1], sy[LH][LH + 1];
    __shared__ float hz[5][LH][LT + 1];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, plane = blockIdx.z;
    const size_t pbase = (size_t)plane * H * W;

    {
        // halo tile: every global load of this thread is issued before the first LDS store (the loop would otherwise
        // be seven dependent load -> store round trips; the block's duration is mostly this latency)
        constexpr int NIT = (LH * LH + BLOCK - 1) / BLOCK;
        float vx[NIT], vy[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = tid + it * BLOCK;
            const int r = idx / LH, c = idx - r * LH;
            const int gy = y0 + r - LR, gx = x0 + c - LR;
            const bool in = idx < LH * LH && gy >= 0 && gy < H && gx >= 0 && gx < W;
            vx[it] = in ? img[pbase + (size_t)gy * W + gx] : 0.f;
            vy[it] = in ? gt[pbase + (size_t)gy * W + gx] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = tid + it * BLOCK;
            if (idx < LH * LH) { const int r = idx / LH, c = idx - r * LH; sx[r][c] = vx[it]; sy[r][c] = vy[it]; }
        }
    }
    __syncthreads();
    // Round 6: both passes slide the window over REGISTERS -- a thread takes LQ = 4 neighbouring outputs and reads the 14 inputs they share
    // once instead of 4 x 11 times (335 LDS reads per thread -> 110).  Every output is still the same eleven products added in the same
    // order.  Horizontal: items (row, group of four columns), 42 x 8; the rows are 43 words apart and the groups four, so a wave's reads and
    // writes fall on distinct banks.  (Taking the five moment maps through ONE buffer, one after the other -- 20 KB of LDS per block, the
    // grid in one round of eight blocks per CU -- was built and is slower: it needs 145 registers, and held to fewer it spills: 38 us at
    // four waves per SIMD, 113-117 us at six / eight, against 30.)
    constexpr int LQ = 4, NW = 2 * LR + 1, NIN = LQ + NW - 1;
    static_assert(LT % LQ == 0 && BLOCK == (LT / LQ) * LT, "vertical pass: one thread per (column, group of four rows)");
    for (int item = tid; item < LH * (LT / LQ); item += BLOCK) {
        const int r = item / (LT / LQ), c0 = (item - r * (LT / LQ)) * LQ;
        float xin[NIN], yin[NIN];
#pragma unroll
        for (int j = 0; j < NIN; j++) { xin[j] = sx[r][c0 + j]; yin[j] = sy[r][c0 + j]; }
#pragma unroll
        for (int i = 0; i < LQ; i++) {
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const float x = xin[i + k], y = yin[i + k], w = win.w[k];
                const float wx = w * x, wy = w * y;
                a += wx; b += wy; aa += wx * x; bb += wy * y; ab += wx * y;
            }
            hz[0][r][c0 + i] = a; hz[1][r][c0 + i] = b; hz[2][r][c0 + i] = aa; hz[3][r][c0 + i] = bb; hz[4][r][c0 + i] = ab;
        }
    }
    __syncthreads();
    float l1_sum = 0.f, ssim_sum = 0.f;
    {
        // vertical: thread = (column c, rows r0 .. r0 + 3)
        const int c = tid & (LT - 1), r0 = (tid / LT) * LQ;
        float mom[5][LQ];
#pragma unroll
        for (int m = 0; m < 5; m++) {
            float col[NIN];
#pragma unroll
            for (int j = 0; j < NIN; j++) col[j] = hz[m][r0 + j][c];
#pragma unroll
            for (int i = 0; i < LQ; i++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NW; k++) acc += win.w[k] * col[i + k];
                mom[m][i] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < LQ; i++) {
            const int r = r0 + i;
            const int gy = y0 + r, gx = x0 + c;
            if (gy >= H || gx >= W) continue;
            const float m1 = mom[0][i], m2 = mom[1][i], e11 = mom[2][i], e22 = mom[3][i], e12 = mom[4][i];
            const float m1m2 = m1 * m2, m1sq = m1 * m1, m2sq = m2 * m2;
            const float s1 = e11 - m1sq, s2 = e22 - m2sq, s12 = e12 - m1m2;
            const float An = 2.f * m1m2 + SSIM_C1, Bn = 2.f * s12 + SSIM_C2;
            const float Ad = m1sq + m2sq + SSIM_C1, Bd = s1 + s2 + SSIM_C2;
            const float inv = 1.f / (Ad * Bd);
            const float S = An * Bn * inv;
            ssim_sum += S;
            l1_sum += fabsf(sx[r + LR][c + LR] - sy[r + LR][c + LR]);
            if (dmaps) {
                const size_t o = pbase + (size_t)gy * W + gx;
                // S = An*Bn/(Ad*Bd) with s1, s12 functions of (mu1, E[x^2], E[xy]):
                dmaps[o] = 2.f * m2 * (Bn - An) * inv - S * 2.f * m1 * (1.f / Ad - 1.f / Bd);   // dS/dmu1
                dmaps[map_stride + o] = -S / Bd;                                                 // dS/dE[x^2]
                dmaps[2 * map_stride + o] = 2.f * An * inv;                                      // dS/dE[xy]
            }
        }
    }
    const float l1_tot = block_sum(l1_sum, red);
    const float ss_tot = block_sum(ssim_sum, red);
    if (tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * b] = l1_tot;
        partials[2 * b + 1] = ss_tot;
    }
}

// value = w_l1 * mean|x-y| + w_ssim * mean(ssim) + bias; fixed summation order, double accumulators
__global__ void __launch_bounds__(BLOCK) l1_ssim_reduce_kernel(int nblocks, const float *partials, double inv_count, float w_l1,
                                                               float w_ssim, float bias, float *out)
{
    __shared__ double red[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += BLOCK) { a += (double)partials[2 * i]; b += (double)partials[2 * i + 1]; }
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double l1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv_count;
        const double ss = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv_count;
        out[0] = (float)((double)w_l1 * l1 + (double)w_ssim * ss + (double)bias);
        out[1] = (float)l1;
        out[2] = (float)ss;
    }
}

__global__ void __launch_bounds__(BLOCK) l1_ssim_bwd_kernel(int H, int W, const float *__restrict__ img, const float *__restrict__ gt,
                                                            SsimWindow win, const float *__restrict__ dmaps, size_t map_stride,
                                                            const float *__restrict__ dL_dvalue, float c_l1, float c_ssim,
                                                            float *__restrict__ dL_dimg)
{
    __shared__ float sm[3][LH][LH + 1];
    __shared__ float hz[3][LH][LT + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, plane = blockIdx.z;
    const size_t pbase = (size_t)plane * H * W;
    {
        constexpr int NIT = (LH * LH + BLOCK - 1) / BLOCK;      // all loads first, then the LDS stores (see forward)
        float v0[NIT], v1[NIT], v2[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = tid + it * BLOCK;
            const int r = idx / LH, c = idx - r * LH;
            const int gy = y0 + r - LR, gx = x0 + c - LR;
            const bool in = idx < LH * LH && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = pbase + (size_t)gy * W + gx;
            v0[it] = in ? dmaps[o] : 0.f;
            v1[it] = in ? dmaps[map_stride + o] : 0.f;
            v2[it] = in ? dmaps[2 * map_stride + o] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = tid + it * BLOCK;
            if (idx < LH * LH) { const int r = idx / LH, c = idx - r * LH; sm[0][r][c] = v0[it]; sm[1][r][c] = v1[it]; sm[2][r][c] = v2[it]; }
        }
    }
    __syncthreads();
    constexpr int LQ = 4, NW = 2 * LR + 1, NIN = LQ + NW - 1;          // (the sliding register windows of the forward)
    for (int item = tid; item < LH * (LT / LQ); item += BLOCK) {
        const int r = item / (LT / LQ), c0 = (item - r * (LT / LQ)) * LQ;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float in[NIN];
#pragma unroll
            for (int j = 0; j < NIN; j++) in[j] = sm[m][r][c0 + j];
#pragma unroll
            for (int i = 0; i < LQ; i++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NW; k++) acc += win.w[k] * in[i + k];
                hz[m][r][c0 + i] = acc;
            }
        }
    }
    __syncthreads();
    const float g = dL_dvalue ? dL_dvalue[0] : 1.f;
    const float gl1 = g * c_l1, gss = g * c_ssim;
    {
        const int c = tid & (LT - 1), r0 = (tid / LT) * LQ;
        float v[3][LQ];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float col[NIN];
#pragma unroll
            for (int j = 0; j < NIN; j++) col[j] = hz[m][r0 + j][c];
#pragma unroll
            for (int i = 0; i < LQ; i++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NW; k++) acc += win.w[k] * col[i + k];
                v[m][i] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < LQ; i++) {
            const int gy = y0 + r0 + i, gx = x0 + c;
            if (gy >= H || gx >= W) continue;
            const size_t o = pbase + (size_t)gy * W + gx;
            const float x = img[o], y = gt[o];
            const float diff = x - y;
            const float sgn = (float)(diff > 0.f) - (float)(diff < 0.f);
            dL_dimg[o] = gl1 * sgn + gss * (v[0][i] + 2.f * x * v[1][i] + y * v[2][i]);
        }
    }
}

}  // namespace gms

using namespace gms;

static bool loss_args_ok(const GmsLossArgs *a)
{
    return a && a->planes >= 0 && a->height >= 0 && a->width >= 0 &&
           ((int64_t)a->planes * a->height * a->width == 0 || (a->img && a->gt));
}

static dim3 loss_grid(const GmsLossArgs *a)
{
    return dim3((unsigned)((a->width + LT - 1) / LT), (unsigned)((a->height + LT - 1) / LT), (unsigned)a->planes);
}

extern "C" size_t gms_l1_ssim_partials(int32_t planes, int32_t height, int32_t width)
{
    const size_t nb = (size_t)((width + LT - 1) / LT) * (size_t)((height + LT - 1) / LT) * (size_t)(planes > 0 ? planes : 0);
    return 2 * (nb > 0 ? nb : 1);
}

extern "C" int32_t gms_l1_ssim_forward(const GmsLossArgs *a, float *dmaps, float *partials, float *out, void *stream_)
{
    gms::TraceRange trace_range("gms_l1_ssim_forward");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    if (!loss_args_ok(a) || !out || !partials) { set_error("gms_l1_ssim_forward: invalid argument"); return GMS_ERR_INVALID_ARGUMENT; }
    const int64_t count = (int64_t)a->planes * a->height * a->width;
    if (count == 0) { set_error("gms_l1_ssim_forward: empty image (the reference's mean() would be NaN)"); return GMS_ERR_INVALID_ARGUMENT; }
    static const SsimWindow win = make_window();
    const dim3 grid = loss_grid(a);
    if (grid.y > 65535u || grid.z > 65535u) { set_error("gms_l1_ssim_forward: image too large for one launch"); return GMS_ERR_INVALID_ARGUMENT; }
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    GMS_LAUNCH(GMS_K_LOSS_FWD, stream,
               (l1_ssim_fwd_kernel<<<grid, BLOCK, 0, stream>>>(a->height, a->width, a->img, a->gt, win, dmaps, (size_t)count, partials),
                l1_ssim_reduce_kernel<<<1, BLOCK, 0, stream>>>(nblocks, partials, 1.0 / (double)count, a->w_l1, a->w_ssim, a->bias, out)));
    GMS_KERNEL_CHECK(0, stream, "l1_ssim_fwd");
    return GMS_OK;
}

extern "C" int32_t gms_l1_ssim_backward(const GmsLossArgs *a, const float *dmaps, const float *dL_dvalue, float *dL_dimg,
                                        void *stream_)
{
    gms::TraceRange trace_range("gms_l1_ssim_backward");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    if (!loss_args_ok(a) || !dmaps || !dL_dimg) { set_error("gms_l1_ssim_backward: invalid argument"); return GMS_ERR_INVALID_ARGUMENT; }
    const int64_t count = (int64_t)a->planes * a->height * a->width;
    if (count == 0) return GMS_OK;
    static const SsimWindow win = make_window();
    const dim3 grid = loss_grid(a);
    const float inv = (float)(1.0 / (double)count);
    GMS_LAUNCH(GMS_K_LOSS_BWD, stream,
               l1_ssim_bwd_kernel<<<grid, BLOCK, 0, stream>>>(a->height, a->width, a->img, a->gt, win, dmaps, (size_t)count, dL_dvalue,
                                                             a->w_l1 * inv, a->w_ssim * inv, dL_dimg));
    GMS_KERNEL_CHECK(0, stream, "l1_ssim_bwd");
    return GMS_OK;
}
