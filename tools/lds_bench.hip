// Microbenchmark (MI355X): what an LDS float atomic costs next to the LDS reads and plain stores of the micro-tile backward's trip.
// The walk of ru_bwd / micro_bwd takes ~1 500 cycles per trip per wave with 2-4 waves per SIMD in it (tools/micro_phases.py) although
// a trip issues ~180 cycles of VALU: this measures the LDS side of a trip in isolation, with the kernel's own access pattern
// (four 16-lane rows, each row on a different table entry, ten of its lanes active) and the kernel's occupancy (6 blocks of 4 waves
// per CU, 25 KB of LDS each).
//   hipcc -O3 --offload-arch=gfx950 tools/lds_bench.hip -o tools/lds_bench.bin && tools/lds_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

constexpr int ITERS = 4096;

template <int MODE>
__global__ void __launch_bounds__(256) bench(float *out, unsigned long long *cyc, int pad)
{
    __shared__ float table[256 * 16];          // (16 KB + 10 KB of records: 6 blocks per CU, as the kernel)
    __shared__ float4 ra[256], rb[256];
    __shared__ float2 rc[256];
    for (int k = threadIdx.x; k < 256 * 16; k += 256) table[k] = 0.f;
    ra[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); rb[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); rc[threadIdx.x] = make_float2(1.f, 2.f);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    // the lanes of a row that carry a sum (row_reduce10's layout): ten of sixteen
    const bool alane = (li & 1) == 0 ? (li != 14) : (li == 1 || li == 9);
    int field;
    switch (li) { case 0: field = 0; break; case 8: field = 5; break; case 4: field = 3; break; case 12: field = 8; break; case 2: field = 1; break;
                  case 10: field = 6; break; case 6: field = 4; break; case 14: field = 9; break; case 1: field = 2; break; default: field = 7; break; }
    uint32_t e = (uint32_t)(row * 61 + (threadIdx.x >> 6) * 17 + blockIdx.x * 7) & 255u;      // row-uniform pseudo-random entry
    float acc = 0.f;
    const float y = 1.0f + (float)li;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        e = (e * 5u + 77u + (uint32_t)row * 13u) & 255u;
        if (MODE == 1) { if (alane) atomicAdd(&table[e * 10 + field], y); }                                    // ds_add_f32, 40 lanes
        if (MODE == 2) { if (alane) atomicAdd(&table[e * 10 + field], y); __builtin_amdgcn_s_waitcnt(0xc07f); }   // ... waited for
        if (MODE == 3) { if (alane) table[e * 10 + field] = y; }                                               // ds_write_b32, 40 lanes
        if (MODE == 4) { if (alane) acc += atomicAdd(&table[e * 10 + field], y); }                              // returning atomic
        if (MODE == 5) { if (row == 0 && alane) atomicAdd(&table[e * 10 + field], y); }                        // one row: 10 lanes
        if (MODE == 6) { atomicAdd(&table[e * 10 + (li < 10 ? li : 0)], li < 10 ? y : 0.f); }                  // 64 lanes (six onto field 0)
        if (MODE == 7) { const float4 a = ra[e], b = rb[e]; const float2 c = rc[e]; acc += a.x + b.y + c.x; }  // a trip's record reads
        if (MODE == 8) { const float4 a = ra[e], b = rb[e]; const float2 c = rc[e]; acc += a.x + b.y + c.x; if (alane) atomicAdd(&table[e * 10 + field], acc); }   // reads + add: a trip
        if (MODE == 9) { const float4 a = ra[e], b = rb[e]; const float2 c = rc[e]; acc += a.x + b.y + c.x; if (alane) table[e * 10 + field] = acc; }           // reads + plain store
        if (MODE == 10) { if (alane) { const float o = table[e * 10 + field]; table[e * 10 + field] = o + y; } }  // read-add-write without atomicity
        if (MODE == 11) { if (li < 10) atomicAdd(&table[e * 10 + li], y); }                                    // 40 lanes, consecutive fields (lanes 0-9 of a row)
        if (MODE == 12) { if (li < 10) atomicAdd(&table[e * 16 + li], y); }                                    // ... 64-byte entries (would need 16 KB)
        if (MODE == 13) { if (alane) atomicAdd(reinterpret_cast<unsigned int *>(table) + e * 10 + field, (unsigned int)li + 1u); }        // ds_add_u32, 40 lanes
        if (MODE == 14) { if (alane) atomicAdd(reinterpret_cast<unsigned long long *>(table) + (e & 127u) * 10 + field, (unsigned long long)li + 1ull); }   // ds_add_u64, 40 lanes
        if (MODE == 15) { if (alane) atomicMax(reinterpret_cast<int *>(table) + e * 10 + field, (int)li + it); }                         // ds_max_i32, 40 lanes
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 123.456f || pad == 12345) out[threadIdx.x] = acc + table[threadIdx.x];
}

template <int MODE>
static void run(const char *name, float *out, unsigned long long *cyc, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MODE><<<blocks, 256>>>(out, cyc, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<MODE><<<blocks, 256>>>(out, cyc, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / ITERS;
    // per-CU instruction rate: 24 waves per CU (6 blocks x 4 waves) each issuing ITERS iterations in `ms`
    const double us = ms * 1e3, iters_per_cu = 24.0 * ITERS * ((double)blocks / (256.0 * 6.0));
    printf("%-58s %8.1f us  %7.1f cycles per iteration per wave (median, s_memtime)  %6.1f ns per iteration per CU\n", name, us, med, us * 1e3 / iters_per_cu);
}

int main()
{
    float *out; unsigned long long *cyc;
    const int blocks = 256 * 6;
    hipMalloc(&out, 4096); hipMalloc(&cyc, blocks * 4 * 8);
    printf("LDS microbenchmark, %d blocks of 256 threads (6 per CU), %d iterations per wave\n", blocks, ITERS);
    run<0>("0  loop only", out, cyc, blocks);
    run<1>("1  ds_add_f32, 4 rows x 10 lanes, no wait", out, cyc, blocks);
    run<2>("2  ds_add_f32, 4 rows x 10 lanes, lgkmcnt(0) each time", out, cyc, blocks);
    run<3>("3  ds_write_b32, 4 rows x 10 lanes", out, cyc, blocks);
    run<4>("4  ds_add_rtn_f32, 4 rows x 10 lanes, result used", out, cyc, blocks);
    run<5>("5  ds_add_f32, 1 row x 10 lanes", out, cyc, blocks);
    run<6>("6  ds_add_f32, 64 lanes", out, cyc, blocks);
    run<7>("7  record reads (b128 + b128 + b64), waited", out, cyc, blocks);
    run<8>("8  record reads + ds_add_f32 (a trip's LDS traffic)", out, cyc, blocks);
    run<9>("9  record reads + ds_write_b32", out, cyc, blocks);
    run<10>("10 ds_read + add + ds_write (no atomicity)", out, cyc, blocks);
    run<11>("11 ds_add_f32, lanes 0-9 of each row on consecutive fields", out, cyc, blocks);
    run<12>("12 ... with 64-byte table entries", out, cyc, blocks);
    run<13>("13 ds_add_u32, 4 rows x 10 lanes", out, cyc, blocks);
    run<14>("14 ds_add_u64, 4 rows x 10 lanes", out, cyc, blocks);
    run<15>("15 ds_max_i32, 4 rows x 10 lanes", out, cyc, blocks);
    return 0;
}
