// Microbenchmark (MI355X): issue cost of the VALU instruction classes the micro-tile backward's trip is made of -- plain f32, DPP row
// operations, v_cndmask, transcendentals, the f32 -> f64 -> magic-number conversion of the fixed-point table -- per SIMD, at 1, 2, 4
// and 8 waves per SIMD.  The kernel's "VALU issue fraction" (bench.py roofline.valu) prices every wave instruction at 2 cycles
// (SIMD-32, MI355X_MICROARCH.md); this measures whether that holds for the mix the walk issues.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench.bin && tools/valu_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;          // instructions per iteration (8 independent chains x 2)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(256) bench(float *out, unsigned long long *cyc, float seed)
{
    float a[8]; double d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = seed + (float)(threadIdx.x + k); d[k] = (double)a[k]; }
    const float m = 1.0000001f, c = 1e-9f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (MODE == 0) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                REP8(X)
#undef X
            } else if (MODE == 1) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                REP8(X)
#undef X
            } else if (MODE == 2) {
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (MODE == 3) {
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (MODE == 4) {
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (MODE == 5) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m) : );
                REP8(X)
#undef X
            } else if (MODE == 6) {
#define X(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (MODE == 7) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (MODE == 8) {
#define X(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
                REP8(X)
#undef X
            } else if (MODE == 9) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
                REP8(X)
#undef X
            } else if (MODE == 10) {
#define X(k) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (MODE == 11) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
                REP8(X)
#undef X
            } else if (MODE == 12) {          // a dependent chain of fma: latency, not throughput (one chain)
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
                REP8(X)
#undef X
            } else if (MODE == 13) {          // dependent DPP chain
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
                REP8(X)
#undef X
            } else if (MODE == 14) {          // dependent exp chain
#define X(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));
                REP8(X)
#undef X
            } else if (MODE == 15) {          // v_cmp + v_cndmask pairs (a select as the compiler emits it)
#define X(k) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m) : "vcc");
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += a[k] + (float)d[k];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, float *out, unsigned long long *cyc, int insts_per_x)
{
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 8}) {          // waves per SIMD = blocks per CU (a 256-thread block puts one wave on each SIMD)
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        bench<MODE><<<blocks, 256>>>(out, cyc, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        bench<MODE><<<blocks, 256>>>(out, cyc, 1.f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double n = (double)ITERS * UNROLL * insts_per_x;
        // s_memtime ticks per instruction of one wave, and per SIMD (= per wave / waves on the SIMD)
        const double per_wave = (double)h[h.size() / 2] / n;
        printf("  w/SIMD %d: %5.2f clk/inst/SIMD (%6.1f us)", wps, per_wave / wps, ms * 1e3);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8 * 4 * 256 * 8);
    printf("clock64 ticks per wave instruction, divided by the waves per SIMD (throughput view; ITERS %d x %d instructions)\n", ITERS, UNROLL);
    run<0>("v_fma_f32 (8 chains)", out, cyc, 1);
    run<1>("v_mul_f32", out, cyc, 1);
    run<2>("v_add_f32_dpp row_ror:8", out, cyc, 1);
    run<3>("v_add_f32_dpp quad_perm", out, cyc, 1);
    run<4>("v_add_f32_dpp row_half_mirror", out, cyc, 1);
    run<10>("v_mov_b32_dpp row_ror:8", out, cyc, 1);
    run<5>("v_cndmask_b32 (vcc)", out, cyc, 1);
    run<15>("v_cmp_lt_f32 + v_cndmask_b32", out, cyc, 2);
    run<6>("v_exp_f32", out, cyc, 1);
    run<7>("v_rcp_f32", out, cyc, 1);
    run<8>("v_cvt_f64_f32", out, cyc, 1);
    run<9>("v_add_f64", out, cyc, 1);
    run<11>("v_pk_fma_f32", out, cyc, 1);
    run<12>("v_fma_f32, ONE dependent chain", out, cyc, 1);
    run<13>("v_add_f32_dpp, ONE dependent chain", out, cyc, 1);
    run<14>("v_exp_f32, ONE dependent chain", out, cyc, 1);
    return 0;
}
