// test_hooks.hip -- libgmsplat_testhooks.so: device functions of the production kernels exposed for tests.  NOT part of the
// product library (libgmsplat.so exports none of this); loaded only by tests/ through ctypes.
//
// gms_test_cull: the conservative culls (gms_blend.h: cull_extents, rect_hit, block_mask) against the per-pixel accept test
// (pair_power + the alpha >= 1/255 rule of the compositing kernels), all evaluated ON THE GPU by the very device functions the
// kernels inline, on caller-supplied (splat, tile) pairs.  tests/test_gpu_cull.py feeds it the adversarial generators of
// tests/test_filter_emulation.py (whose numpy emulation can only approximate the compiler's contractions and intrinsics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gms_common.h"
#include "gms_blend.h"

namespace gms {

// in [n][10] = (cov_xx, cov_yy, conic A, B, C, opacity', pixel x, pixel y, tile x0, tile y0); out [n][4] uint32 =
// {accept16 | mask16 << 16, acceptQ | hitQ << 4, bits(ex), bits(ey)}: accept = some pixel of the 4x4 block (8x8 quadrant) takes the
// splat; mask / hit = what block_mask / rect_hit keep.
__global__ void __launch_bounds__(256) test_cull_kernel(int n, const float *in, uint32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *v = in + (size_t)i * 10;
    const float a_d = v[0], c_d = v[1], A = v[2], B = v[3], C = v[4], op = v[5], px = v[6], py = v[7], tx0 = v[8], ty0 = v[9];
    float ex, ey;
    cull_extents(a_d, c_d, A, B, C, op, ex, ey);
    SplatRec r;
    r.q0 = make_float4(px, py, A, B);
    r.q1 = make_float4(C, op, 0.f, 0.f);
    r.q2 = make_float4(0.f, 0.f, ex, ey);
    const uint32_t mask = block_mask(r, tx0, ty0);
    uint32_t hitq = 0;
    for (int q = 0; q < 4; q++) {
        RectF p{tx0 + 8.f * (q & 1), ty0 + 8.f * (q >> 1), tx0 + 8.f * (q & 1) + 7.f, ty0 + 8.f * (q >> 1) + 7.f};
        if (rect_hit(r.q0, C, op, r.q2, p)) hitq |= 1u << q;
    }
    uint32_t acc16 = 0, accq = 0;
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) {
            const float dx = px - (tx0 + (float)x), dy = py - (ty0 + (float)y);
            const float pw = pair_power(A, B, C, dx, dy);
            const float al = fminf(ALPHA_MAX, op * __expf(pw));
            if (pw <= 0.f && al >= ALPHA_MIN) { acc16 |= 1u << ((y >> 2) * 4 + (x >> 2)); accq |= 1u << ((y >> 3) * 2 + (x >> 3)); }
        }
    out[4 * (size_t)i] = acc16 | (mask << 16);
    out[4 * (size_t)i + 1] = accq | (hitq << 4);
    out[4 * (size_t)i + 2] = __float_as_uint(ex);
    out[4 * (size_t)i + 3] = __float_as_uint(ey);
}

// gms_test_fixed_point: the conversions of the micro-tile backward's fixed-point gradient table (gms_blend.h: fx_from_float,
// fx_to_float) on caller-supplied (value, scale exponent) pairs
__global__ void __launch_bounds__(256) test_fx_kernel(int n, const float *y, const int *k, long long *fixed, float *back)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the walk's form (scale prepared once per lane, clamp on y, scaling inside the f64 FMA) must give the bits of the plain form
    const long long v0 = fx_from_float(y[i], k[i]);
    const long long v = fx_from_float(y[i], fx_prepare(k[i]));
    fixed[i] = v == v0 ? v : (long long)0x8000000000000000ull;          // (a sentinel no test value produces)
    back[i] = fx_to_float(v0, k[i]);
}

}  // namespace gms

extern "C" int32_t gms_test_fixed_point(int32_t n, const float *y_host, const int32_t *k_host, long long *fixed_host, float *back_host)
{
    if (n <= 0) return 0;
    float *dy = nullptr, *db = nullptr; int *dk = nullptr; long long *dv = nullptr;
    hipError_t e = hipMalloc(&dy, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&db, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&dk, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&dv, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpy(dy, y_host, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dk, k_host, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        gms::test_fx_kernel<<<(unsigned)((n + 255) / 256), 256>>>(n, dy, dk, dv, db);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(fixed_host, dv, (size_t)n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(back_host, db, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dy); (void)hipFree(db); (void)hipFree(dk); (void)hipFree(dv);
    return e == hipSuccess ? 0 : -(int32_t)e;
}

// host buffers in, host buffers out (the hook owns its device memory); returns 0 or a negative hipError
extern "C" int32_t gms_test_cull(int32_t n, const float *in_host, uint32_t *out_host)
{
    if (n <= 0) return 0;
    float *din = nullptr; uint32_t *dout = nullptr;
    hipError_t e = hipMalloc(&din, (size_t)n * 10 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n * 4 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(din, in_host, (size_t)n * 10 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        gms::test_cull_kernel<<<(unsigned)((n + 255) / 256), 256>>>(n, din, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out_host, dout, (size_t)n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(din); (void)hipFree(dout);
    return e == hipSuccess ? 0 : -(int32_t)e;
}
