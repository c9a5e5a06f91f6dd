// raster_forward.hip -- forward pass of the gfx950 Gaussian rasterizer: preprocess, binning, host orchestration.
//
// Pipeline (one HIP stream, no MFMA anywhere: there is no dense contraction on this path):
//   K1  preprocess_fwd   1 thread / Gaussian.  SH rows staged through LDS by wave-cooperative float4 loads, read back
//                        conflict-free at a 52-dword pitch.  Writes the 48-byte splat record, depth, clamp bits, radius,
//                        visibility byte; counts tile coverage in a block-level LDS tile table (one global atomic per
//                        distinct tile per block; footprints > 64 tiles are expanded by the whole wave).
//   K2  tile_scan        ONE block of 1024: eleven exclusive scans over the tiles in one pass (instances, work units,
//                        multi-segment slots, six heaviest-first unit classes, sort runs, multi-run tiles); publishes
//                        N / deepest tile / unit count to the host through the pinned slot, clears the tile counters.
//   K3  emit_instances   one 64-bit key (depth bits << 32 | id) per (Gaussian, tile): per-block LDS tile table, ONE
//                        cursor atomic per distinct tile per block; spare blocks write the (tile, segment) unit records
//                        (heaviest first) and the sort's run tables.
//   K4  tile_sort        presort of 1024-key runs (one block per run, wave-local bitonic sub-stages), then merge-path
//                        levels in LDS per multi-run tile (<= 8192 keys) or multi-block merge-path passes for deeper
//                        tiles.  Total order on (depth, id): independent of the emission order == the reference's
//                        stable radix sort.
//   K5/K6 compositing    blend.hip (segment-parallel: head / fwd / finalize).
//
// Library state: everything that outlives a call (tile counters, launch-size hints) is keyed by
// (device, stream[, W, H, P]); see FrameState / Counters below.
//
// Behavioural contract: SURVEY.md appendix A (constants, skip/stop tests, pixel-centre
// convention); reference call site renderer/gaussian_renderer/__init__.py:94-102.
#include <stdarg.h>
#include <stdio.h>

#include <stdlib.h>
#include <mutex>

#include "gms_blend.h"
#include "gms_mesh.h"
#include <atomic>
#include <vector>
#include <chrono>

#include "gms_common.h"
#include "gms_project.h"

namespace gms {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------ K1
constexpr int SMALL_AREA = 64; // tiles a lane walks itself (lock-step, aggregated atomics); larger footprints are expanded by the whole wave
constexpr int SH_PITCH = 52;   // dwords per LDS row: 48 used; 52*l mod 64 hits 16 distinct 16-B slots

struct PreArgs {
    int P, D, M, W, H, gx, gy;
    const float *means3D, *shs, *shs_rest, *colors, *opac, *scales, *rots, *cov3Dp, *view, *proj, *campos;
    float mod, tanx, tany;
    int aa;
    int *radii;
    uint8_t *visible;        // optional: radii > 0 as bytes
    GeomState geom;
    uint32_t *tile_count;
    uint32_t *zero_cursor;   // [T] emit cursors, cleared here when the frame takes the inline-scan emit (else NULL: tile_scan clears them)
    uint32_t *zero_cmax;     // [T] per-tile colour maxima (ImageState::tile_cmax), cleared here for the micro-tile launches of this frame
    int T;
    GmsMeshArgs mesh;        // K0 instantiation only: centre / scale / rotation / opacity are derived from the mesh in the thread
    float *k0_xyz, *k0_scale, *k0_rot, *k0_opac;   // ... and stored here for the backward (training frames; NULL: forward-only frame)
};

// SH -> RGB (before +0.5/clamp) for one channel from the coefficient row r[k*3+c] held in registers;
// same operation order as utils/sh_utils.py:57-112 evaluated per channel.
template <int DEG>
__device__ __forceinline__ float sh_eval_channel(const float *row, int c, float x, float y, float z)
{
#pragma clang fp contract(off)
    float res = SH_C0 * row[0 * 3 + c];
    if (DEG > 0) {
        res = res - SH_C1 * y * row[1 * 3 + c] + SH_C1 * z * row[2 * 3 + c] - SH_C1 * x * row[3 * 3 + c];
        if (DEG > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + SH_C2[0] * xy * row[4 * 3 + c] + SH_C2[1] * yz * row[5 * 3 + c] +
                  SH_C2[2] * (2.f * zz - xx - yy) * row[6 * 3 + c] + SH_C2[3] * xz * row[7 * 3 + c] +
                  SH_C2[4] * (xx - yy) * row[8 * 3 + c];
            if (DEG > 2) {
                res = res + SH_C3[0] * y * (3.f * xx - yy) * row[9 * 3 + c] + SH_C3[1] * xy * z * row[10 * 3 + c] +
                      SH_C3[2] * y * (4.f * zz - xx - yy) * row[11 * 3 + c] +
                      SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * row[12 * 3 + c] +
                      SH_C3[4] * x * (4.f * zz - xx - yy) * row[13 * 3 + c] + SH_C3[5] * z * (xx - yy) * row[14 * 3 + c] +
                      SH_C3[6] * x * (xx - 3.f * yy) * row[15 * 3 + c];
            }
        }
    }
    return res;
}

// Stage the first NQ float4 chunks of `rows` consecutive SH rows into the wave's LDS region with
// coalesced loads (each wave instruction moves 1 KiB), pitch SH_PITCH dwords per row.
template <int NQ>
__device__ __forceinline__ void stage_sh_rows(const float *shs, int g0, int rows, int rowq, float *wl, int lane)
{
    const float4 *src = reinterpret_cast<const float4 *>(shs) + (size_t)g0 * rowq;
    for (int idx = lane; idx < rows * NQ; idx += WAVE) {
        const int r = idx / NQ, c = idx - r * NQ;
        *reinterpret_cast<float4 *>(wl + r * SH_PITCH + c * 4) = src[(size_t)r * rowq + c];
    }
}

// Read the lane's row back as float4 (ds_read_b128, conflict-free at a 52-dword pitch) and evaluate.
template <int DEG>
__device__ __forceinline__ void sh_colour(const float *row_lds, float x, float y, float z, float rgb[3], unsigned &clampbits)
{
    constexpr int NQ = ((DEG + 1) * (DEG + 1) * 3 + 3) / 4;
    float r[NQ * 4];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const float4 v = *reinterpret_cast<const float4 *>(row_lds + q * 4);
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = sh_eval_channel<DEG>(r, c, x, y, z) + 0.5f;
        if (v < 0.f) { clampbits |= 1u << c; v = 0.f; }
        rgb[c] = v;
    }
}

// MODE 0: everything in one launch.  MODE 1 / 2: the same kernel cut in two for the two-stream forward -- 1 = geometry (what
// the binning needs: projection, conic, radius, extents, tile counts; 8 KB of LDS instead of 53), 2 = SH -> RGB (57 MB of
// coefficient reads that only the compositing needs: runs on a second stream beside scan / emit / sort / filter).  The two
// write disjoint bytes of the 48-byte record.
// (cull_extents -- the noise-aware half extents stored in record words 10, 11 -- lives in gms_blend.h)
// DMA (round 5; split storage, degree 3, MODE 0): the wave's coefficient block goes global -> LDS with `global_load_lds_dwordx4`
// (1 KiB per wave instruction, no staging VGPRs, no ds_write pass).  The LDS destination of that instruction is lane-linear, and
// so is the source here: the 64 rows of `_features_rest` a wave owns are 11 520 contiguous, 16-byte aligned bytes, copied as they
// lie (row pitch 45 dwords: odd, so the per-lane ds_read_b32 of a row are conflict-free), the 768 bytes of `_features_dc` behind
// them.  Before, the same block went through 12 float4 registers per lane and 48 scalar ds_write_b32 with a division by 45 each.
// K0 (round 5; forward-only frames of the animated render drivers, scripts/render_time_animated.py:68-87): the thread derives its
// Gaussian from the mesh -- barycentric centre, face frame -> activated scale and unit quaternion, sigmoid opacity, the statements
// of mesh_fwd_kernel (gms_mesh.h::splat_from_face) -- instead of loading means3D / scales / rotations / opacities: those four
// tensors (44 bytes per Gaussian written by K0 and read back here, plus K0's raw copies) never reach HBM and the K0 launch is gone.
template <int SHDEG, bool SPLIT, int MODE = 0, bool DMA = false, bool K0 = false>
__device__ __forceinline__ void preprocess_fwd_body(const PreArgs &a)
{
#pragma clang fp contract(off)
    static_assert(!DMA || (SHDEG >= 0 && SHDEG <= 3 && SPLIT && MODE == 0), "the LDS-DMA staging exists for split degree-3 STORAGE (any active degree: the rows come in whole)");
    static_assert(!K0 || DMA, "the fused mesh input rides on the LDS-DMA instantiation");
    // Round 6: the REST rows come in as TWO halves of 32 rows through the same 6 KB of LDS per wave (6 x 1 KiB DMA instructions each,
    // 1 440 of the 1 536 floats used), the 64 x 3 DC floats behind them: 27 KB per block instead of 49.
    constexpr int DMA_HALF_ROWS = WAVE / 2, DMA_HALF_Q = 6;                       // rows and DMA instructions per half
    constexpr int DMA_REST_FLOATS = DMA_HALF_Q * WAVE * 4;                        // 1 536 (>= 32 x 45 = 1 440)
    constexpr int DMA_WAVE_FLOATS = DMA_REST_FLOATS + WAVE * 3;
    __shared__ __attribute__((aligned(16))) float sh_lds[MODE == 1 ? 2 * TT_SLOTS : (DMA ? 4 * DMA_WAVE_FLOATS : 4 * WAVE * SH_PITCH)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * BLOCK + tid;
    const bool valid = i < a.P;
    if (MODE != 2 && a.zero_cursor)
        for (int q = i; q < a.T; q += (int)gridDim.x * BLOCK) a.zero_cursor[q] = 0u;
    if (MODE != 2 && a.zero_cmax)
        for (int q = i; q < a.T; q += (int)gridDim.x * BLOCK) a.zero_cmax[q] = 0u;
    if (K0 && a.mesh.prezero)          // the [V,3] buffer the mesh backward accumulates into (what spare blocks of mesh_fwd clear)
        for (int64_t q = i; q < a.mesh.prezero_count; q += (int64_t)gridDim.x * BLOCK) a.mesh.prezero[q] = 0.f;

    // Fast path: every global load of this thread is issued before anything is computed, so the
    // position / scale / rotation / opacity / SH round trips overlap instead of chaining.
    constexpr int NQ = SHDEG >= 0 ? ((SHDEG + 1) * (SHDEG + 1) * 3 + 3) / 4 : 1;
    constexpr int NB3 = SHDEG >= 0 ? (SHDEG + 1) * (SHDEG + 1) * 3 : 3;      // floats needed per row
    constexpr int RESTF = 45;                                                 // floats per row of shs_rest (M = 16)
    float4 stg[NQ];
    float dcv[3] = {0.f, 0.f, 0.f};
    float s_in[3] = {0.f, 0.f, 0.f};
    float4 q_in = make_float4(1.f, 0.f, 0.f, 0.f);
    float op_in = 0.f;
    SplatInputs k0in = {};
    if (K0 && valid) splat_inputs_load(a.mesh, (int64_t)i, k0in);          // (in front of the SH rows' DMA: see gms_mesh.h)
    if (SHDEG >= 0) {
        const int g0 = blockIdx.x * BLOCK + wave * WAVE;
        const int rows = min(WAVE, a.P - g0);
        if (MODE == 1) {
        } else if (DMA) {
            float *wl = sh_lds + wave * DMA_WAVE_FLOATS;
            const float *sp = a.shs_rest + (size_t)g0 * RESTF;       // g0 % 64 == 0: 16-byte aligned
            const int nfl = max(0, min(rows, DMA_HALF_ROWS)) * RESTF;        // first half: rows 0 .. 31
#pragma unroll
            for (int j = 0; j < DMA_HALF_Q; j++) {
                const int e4 = (lane + WAVE * j) * 4;                // (a chunk that straddles the end of the array is copied below)
                if (e4 + 3 < nfl)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + e4),
                                                     (__attribute__((address_space(3))) void *)(wl + WAVE * 4 * j), 16, 0, 0);
            }
            const float *dp = a.shs + (size_t)g0 * 3;
            if (lane * 4 + 3 < rows * 3)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dp + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(wl + DMA_REST_FLOATS), 16, 0, 0);
        } else if (!SPLIT) {
            const float4 *src = reinterpret_cast<const float4 *>(a.shs) + (size_t)g0 * 12;   // M = 16: 12 float4 per row
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                const int idx = lane + WAVE * j;
                const int r = idx / NQ, c = idx - r * NQ;
                stg[j] = idx < rows * NQ ? src[(size_t)r * 12 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            // split storage: the wave's DC block (rows*3 floats) and REST block (rows*45 floats) are each
            // contiguous; both are fetched with flat coalesced loads and re-assembled into the same
            // [k][c] LDS rows the fused layout uses
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int e = lane + WAVE * j;
                dcv[j] = e < rows * 3 ? a.shs[(size_t)g0 * 3 + e] : 0.f;
            }
            if (SHDEG == 3) {
                const float *sp = a.shs_rest + (size_t)g0 * RESTF;       // g0 % 64 == 0: 16-byte aligned
                const float4 *src = reinterpret_cast<const float4 *>(sp);
                const int nfl = rows * RESTF;
#pragma unroll
                for (int j = 0; j < NQ; j++) {
                    const int e4 = (lane + WAVE * j) * 4;
                    if (e4 + 3 < nfl) stg[j] = src[lane + WAVE * j];
                    else stg[j] = make_float4(e4 < nfl ? sp[e4] : 0.f, e4 + 1 < nfl ? sp[e4 + 1] : 0.f,
                                              e4 + 2 < nfl ? sp[e4 + 2] : 0.f, 0.f);
                }
            }
        }
        if (valid && MODE != 2 && !K0) {
            op_in = a.opac[i];
            if (!a.cov3Dp) {
                s_in[0] = a.scales[3 * (size_t)i]; s_in[1] = a.scales[3 * (size_t)i + 1]; s_in[2] = a.scales[3 * (size_t)i + 2];
                q_in = *reinterpret_cast<const float4 *>(a.rots + 4 * (size_t)i);
            }
        }
    }

    float px = 0, py = 0, pz = 0, vx = 0, vy = 0, vz = 0;
    bool vis = false;
    float pix = 0, piy = 0, cA = 0, cB = 0, cC = 0, opp = 0, a_d = 0, c_d = 0, rad = 0;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    if (K0) {
        if (valid) {
            SplatParams sp;
            splat_from_inputs(a.mesh, k0in, sp);
            px = sp.xyz[0]; py = sp.xyz[1]; pz = sp.xyz[2];
            s_in[0] = sp.scale[0]; s_in[1] = sp.scale[1]; s_in[2] = sp.scale[2];
            q_in = make_float4(sp.q[0], sp.q[1], sp.q[2], sp.q[3]);
            op_in = sp.opacity;
            if (a.k0_xyz) {          // training frame: what mesh_fwd would have written, for gms_rasterize_backward / the model's attributes
                a.k0_xyz[3 * (size_t)i] = px; a.k0_xyz[3 * (size_t)i + 1] = py; a.k0_xyz[3 * (size_t)i + 2] = pz;
                a.k0_scale[3 * (size_t)i] = s_in[0]; a.k0_scale[3 * (size_t)i + 1] = s_in[1]; a.k0_scale[3 * (size_t)i + 2] = s_in[2];
                *reinterpret_cast<float4 *>(a.k0_rot + 4 * (size_t)i) = q_in;
                a.k0_opac[i] = op_in;
            }
            view_transform(a.view, px, py, pz, vx, vy, vz); vis = vz > NEAR_Z;
        }
    } else if (valid) {
        px = a.means3D[3 * (size_t)i]; py = a.means3D[3 * (size_t)i + 1]; pz = a.means3D[3 * (size_t)i + 2];
        if (MODE == 2) vis = a.radii[i] > 0;                          // (decided by the geometry launch)
        else { view_transform(a.view, px, py, pz, vx, vy, vz); vis = vz > NEAR_Z; }
    }
    if (vis && MODE != 2) {
        const float *Mx = a.proj;
        float hx = dot3p(Mx[0], px, Mx[4], py, Mx[8], pz, Mx[12]);
        float hy = dot3p(Mx[1], px, Mx[5], py, Mx[9], pz, Mx[13]);
        float hw = dot3p(Mx[3], px, Mx[7], py, Mx[11], pz, Mx[15]);
        float pw = 1.f / (hw + 0.0000001f);
        float ndcx = hx * pw, ndcy = hy * pw;
        Cov3 cv;
        if (a.cov3Dp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cv.c[k] = a.cov3Dp[6 * (size_t)i + k];
        } else {
            if (SHDEG < 0) {
                s_in[0] = a.scales[3 * (size_t)i]; s_in[1] = a.scales[3 * (size_t)i + 1]; s_in[2] = a.scales[3 * (size_t)i + 2];
                q_in = *reinterpret_cast<const float4 *>(a.rots + 4 * (size_t)i);
            }
            float q[4] = {q_in.x, q_in.y, q_in.z, q_in.w};
            cov3d_from_scale_rot(s_in, a.mod, q, cv);
        }
        const float fx = (float)a.W / (2.f * a.tanx), fy = (float)a.H / (2.f * a.tany);
        Ewa e;
        ewa_project(a.view, vx, vy, vz, cv, fx, fy, 1.3f * a.tanx, 1.3f * a.tany, e);
        float det0 = e.a0 * e.c0 - e.b * e.b;
        a_d = e.a0 + DILATE; c_d = e.c0 + DILATE;
        float det = a_d * c_d - e.b * e.b;
        float hconv = 1.f;
        if (a.aa) hconv = sqrtf(fmaxf(0.000025f, det0 / det));
        if (det == 0.f) vis = false;
        float dinv = 1.f / det;
        cA = c_d * dinv; cB = -e.b * dinv; cC = a_d * dinv;
        float mid = 0.5f * (a_d + c_d);
        float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
        float l1 = mid + disc, l2 = mid - disc;
        rad = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        pix = ((ndcx + 1.f) * a.W - 1.f) * 0.5f;
        piy = ((ndcy + 1.f) * a.H - 1.f) * 0.5f;
        tile_rect(pix, piy, rad, a.gx, a.gy, minx, miny, maxx, maxy);
        if ((maxx - minx) * (maxy - miny) == 0) vis = false;
        opp = (SHDEG >= 0 ? op_in : a.opac[i]) * hconv;
    }

    // ---- colour
    float rgb[3] = {0, 0, 0};
    unsigned clampbits = 0;
    if (MODE == 1) {
    } else if (DMA) {
        float *wl = sh_lds + wave * DMA_WAVE_FLOATS;
        const int g0 = blockIdx.x * BLOCK + wave * WAVE;
        const int rows = max(0, min(WAVE, a.P - g0));          // (the last block's later waves may own no row at all)
        const float *sp = a.shs_rest + (size_t)g0 * RESTF;
        float dxc = 0.f, dyc = 0.f, dzc = 0.f;
        if (vis) {
            const float dx = px - a.campos[0], dy = py - a.campos[1], dz = pz - a.campos[2];
            const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            dxc = dx * inv; dyc = dy * inv; dzc = dz * inv;
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int r0 = half * DMA_HALF_ROWS;                             // first row of this half
            const int hrows = max(0, min(rows - r0, DMA_HALF_ROWS));
            if (half == 1) {
                wave_sync();                           // every lane of the first half has its coefficients in registers
                const int nfl = hrows * RESTF;
#pragma unroll
                for (int j = 0; j < DMA_HALF_Q; j++) {
                    const int e4 = (lane + WAVE * j) * 4;
                    if (e4 + 3 < nfl)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + r0 * RESTF + e4),
                                                         (__attribute__((address_space(3))) void *)(wl + WAVE * 4 * j), 16, 0, 0);
                }
            }
            if (hrows < DMA_HALF_ROWS) {   // last wave of the array: the (at most one) 16-byte chunk of each block that straddles its end
                const int nfl = hrows * RESTF;
                for (int e = (nfl & ~3) + lane; e < nfl; e += WAVE) wl[e] = sp[r0 * RESTF + e];
            }
            if (half == 0 && rows < WAVE) {
                const int nd = rows * 3;
                for (int e = (nd & ~3) + lane; e < nd; e += WAVE) wl[DMA_REST_FLOATS + e] = a.shs[(size_t)g0 * 3 + e];
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the LDS-DMA copies have landed
            wave_sync();                               // the rows are this wave's own
            if (vis && (lane >> 5) == half) {
                float r[48];
#pragma unroll
                for (int c = 0; c < 3; c++) r[c] = wl[DMA_REST_FLOATS + lane * 3 + c];
#pragma unroll
                for (int m = 0; m < RESTF; m++) r[3 + m] = wl[(lane - r0) * RESTF + m];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float v = sh_eval_channel<SHDEG>(r, c, dxc, dyc, dzc) + 0.5f;          // (the ACTIVE degree: what sh_colour<SHDEG> evaluates)
                    if (v < 0.f) { clampbits |= 1u << c; v = 0.f; }
                    rgb[c] = v;
                }
            }
        }
    } else if (SHDEG >= 0) {
        float *wl = sh_lds + wave * (WAVE * SH_PITCH);
        if (!SPLIT) {
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                const int idx = lane + WAVE * j;
                const int r = idx / NQ, c = idx - r * NQ;
                *reinterpret_cast<float4 *>(wl + r * SH_PITCH + c * 4) = stg[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int e = lane + WAVE * j;
                wl[(e / 3) * SH_PITCH + (e % 3)] = dcv[j];
            }
            if (SHDEG == 3) {
#pragma unroll
                for (int j = 0; j < NQ; j++) {
                    const float v4[4] = {stg[j].x, stg[j].y, stg[j].z, stg[j].w};
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int e = (lane + WAVE * j) * 4 + t;       // flat element of the wave's REST block
                        if (e < WAVE * RESTF) wl[(e / RESTF) * SH_PITCH + 3 + (e % RESTF)] = v4[t];
                    }
                }
            } else if (SHDEG > 0) {
                // low active degree: only the first NB3-3 floats of each REST row are needed
                const int g0 = blockIdx.x * BLOCK + wave * WAVE;
                const int rows = min(WAVE, a.P - g0);
                constexpr int need = NB3 > 3 ? NB3 - 3 : 1;
                for (int e = lane; e < rows * need; e += WAVE) {
                    const int r = e / need, c = e - r * need;
                    wl[r * SH_PITCH + 3 + c] = a.shs_rest[((size_t)g0 + r) * RESTF + c];
                }
            }
        }
        wave_sync();                               // the rows are this wave's own
        if (vis) {
            float dx = px - a.campos[0], dy = py - a.campos[1], dz = pz - a.campos[2];
            float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            sh_colour<(SHDEG >= 0 ? SHDEG : 0)>(wl + lane * SH_PITCH, dx * inv, dy * inv, dz * inv, rgb, clampbits);
        }
    } else if (a.shs) {
        const int nb = (a.D + 1) * (a.D + 1);
        const int rowf = a.M * 3;                 // floats per Gaussian in memory
        float *wl = sh_lds + wave * (WAVE * SH_PITCH);
        const bool any_vis = __any(vis);          // wave-uniform
        const bool vec_ok = (rowf % 4 == 0) && (rowf <= 48);
        if (any_vis && vec_ok) {
            const int g0 = blockIdx.x * BLOCK + wave * WAVE;
            const int rows = min(WAVE, a.P - g0);
            const int rowq = rowf / 4;
            switch (a.D) {
            case 0: stage_sh_rows<1>(a.shs, g0, rows, rowq, wl, lane); break;
            case 1: stage_sh_rows<3>(a.shs, g0, rows, rowq, wl, lane); break;
            case 2: stage_sh_rows<7>(a.shs, g0, rows, rowq, wl, lane); break;
            default: stage_sh_rows<12>(a.shs, g0, rows, rowq, wl, lane); break;
            }
        } else if (vis) {   // generic storage width (M != 16): plain per-lane loads into the lane's row
            for (int k = 0; k < nb * 3; k++) wl[lane * SH_PITCH + k] = a.shs[(size_t)i * rowf + k];
        }
        wave_sync();
        if (vis) {
            float dx = px - a.campos[0], dy = py - a.campos[1], dz = pz - a.campos[2];
            float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            float x = dx * inv, y = dy * inv, z = dz * inv;
            const float *row = wl + lane * SH_PITCH;
            switch (a.D) {
            case 0: sh_colour<0>(row, x, y, z, rgb, clampbits); break;
            case 1: sh_colour<1>(row, x, y, z, rgb, clampbits); break;
            case 2: sh_colour<2>(row, x, y, z, rgb, clampbits); break;
            default: sh_colour<3>(row, x, y, z, rgb, clampbits); break;
            }
        }
    } else if (vis) {
        rgb[0] = a.colors[3 * (size_t)i]; rgb[1] = a.colors[3 * (size_t)i + 1]; rgb[2] = a.colors[3 * (size_t)i + 2];
    }

    if (MODE == 2) {          // colour launch: the three colour words of the record and the clamp bits, nothing else
        if (vis) {
            float *rw = reinterpret_cast<float *>(a.geom.rec + i);
            *reinterpret_cast<float2 *>(rw + 6) = make_float2(rgb[0], rgb[1]);
            rw[8] = rgb[2];
            a.geom.clamped[i] = (uint8_t)clampbits;
        }
        return;
    }
    if (valid && !vis) a.radii[i] = 0;
    if (valid) a.geom.rect[i] = vis ? make_ushort4((unsigned short)minx, (unsigned short)miny, (unsigned short)maxx, (unsigned short)maxy)
                                    : make_ushort4(0, 0, 0, 0);          // (the rectangle the tile counts below were taken over)
    if (valid && a.visible) a.visible[i] = vis ? 1 : 0;
    if (vis && MODE == 1) {   // geometry launch: everything but the colour words
        float ex, ey;
        cull_extents(a_d, c_d, cA, cB, cC, opp, ex, ey);
        float *rw = reinterpret_cast<float *>(a.geom.rec + i);
        *reinterpret_cast<float4 *>(rw) = make_float4(pix, piy, cA, cB);
        *reinterpret_cast<float2 *>(rw + 4) = make_float2(cC, opp);
        rw[9] = 1.f / vz;
        *reinterpret_cast<float2 *>(rw + 10) = make_float2(ex, ey);
        a.geom.depth[i] = vz;
        a.radii[i] = (int)rad;
    }
    if (vis && MODE == 0) {
        float ex, ey;
        cull_extents(a_d, c_d, cA, cB, cC, opp, ex, ey);
        SplatRec rec;
        rec.q0 = make_float4(pix, piy, cA, cB);
        rec.q1 = make_float4(cC, opp, rgb[0], rgb[1]);
        rec.q2 = make_float4(rgb[2], 1.f / vz, ex, ey);
        a.geom.rec[i] = rec;
        a.geom.depth[i] = vz;
        a.geom.clamped[i] = (uint8_t)clampbits;
        a.radii[i] = (int)rad;
    }
    // tile coverage counts.  Footprints up to SMALL_AREA tiles are counted in the block's LDS tile table and
    // flushed with one global atomic per distinct tile; very large footprints are expanded by the whole wave,
    // a tile per lane, straight to global memory.
    const int rw = maxx - minx;
    const int area = vis ? rw * (maxy - miny) : 0;
    const bool small = area <= SMALL_AREA;
    __syncthreads();                               // SH rows no longer needed: their LDS becomes the tile table
    int *tt_key = reinterpret_cast<int *>(sh_lds);
    uint32_t *tt_cnt = reinterpret_cast<uint32_t *>(sh_lds) + TT_SLOTS;
    for (int q = tid; q < TT_SLOTS; q += BLOCK) { tt_key[q] = -1; tt_cnt[q] = 0; }
    __syncthreads();
    if (small) {
        int cx = minx, cy = miny;
        for (int k = 0; k < area; k++) {
            const int t = cy * a.gx + cx;
            const int sl = tt_insert(tt_key, t);
            if (sl >= 0) atomicAdd(&tt_cnt[sl], 1u);
            else atomicAdd(&a.tile_count[t], 1u);
            if (++cx == maxx) { cx = minx; cy++; }
        }
    }
    __syncthreads();
    for (int q = tid; q < TT_SLOTS; q += BLOCK)
        if (tt_key[q] >= 0) atomicAdd(&a.tile_count[tt_key[q]], tt_cnt[q]);
    uint64_t big = __ballot(!small);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const int bminx = __builtin_amdgcn_readlane(minx, src), bminy = __builtin_amdgcn_readlane(miny, src);
        const int brw = __builtin_amdgcn_readlane(rw, src), barea = __builtin_amdgcn_readlane(area, src);
        for (int k = lane; k < barea; k += WAVE) {
            const int ty = k / brw, tx = k - ty * brw;
            atomicAdd(&a.tile_count[(bminy + ty) * a.gx + bminx + tx], 1u);
        }
    }
}

template <int SHDEG, bool SPLIT, int MODE = 0>
__global__ void __launch_bounds__(BLOCK) preprocess_fwd_kernel(PreArgs a) { preprocess_fwd_body<SHDEG, SPLIT, MODE, false, false>(a); }
// The LDS-DMA instantiations: 27 KB of LDS per block (SH rows in two halves) admits five blocks per CU, and five blocks per CU hold the whole
// grid of the headline frame (1 171 blocks) in ONE round -- at three (49 KB) and at four (100 registers) a second, half-empty round follows:
// 35.5 / 34.9 us against 31.5 at five (profiles/r06w1_*, r06w2_*).  The compiler is held to the 96 registers that takes (two to four spilled).
// DEG: the ACTIVE SH degree (train.py:86-87 raises it every 1 000 iterations): frames rendered straight from the mesh take this kernel at
// every degree -- the storage is degree 3 whatever is active, the rows come in whole and the lower bands are evaluated out of them.
template <bool K0, int DEG = 3>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) preprocess_fwd_dma_kernel(PreArgs a)
{
    preprocess_fwd_body<DEG, true, 0, true, K0>(a);
}

// ------------------------------------------------------------------------------------ K2
// One block: exclusive scans over the tiles of (a) instance counts -> tile_offset (offset[T] = N),
// (b) segments per tile -> unit_first, (c) segments of multi-segment tiles -> mseg_first (slots of
// the per-unit pixel state).  Also resets the emit cursors.
// tile sort (K4) sizes, needed by the scan below
constexpr int SORT_RUN = 1024;
constexpr int SORT_RUNS_PER_TILE = 8;
constexpr int SORT_BIG_CHUNK = SORT_RUN * SORT_RUNS_PER_TILE;    // 8192 keys = 64 KiB LDS
constexpr int MP_CHUNK = 1024;                                   // output keys per merge-path block

constexpr int NSCAN = 11;
constexpr int NCLASS = 6;     // dispatch classes: full | partial >=3/4 L | >=1/2 L | >=1/4 L | < 1/4 L | empty
constexpr int SCAN_THREADS = 1024;

// per-tile quantities scanned: instances, segments, segments of multi-segment tiles, then the dispatch classes
__device__ __forceinline__ void tile_terms(uint32_t c, uint32_t L, uint32_t t[NSCAN], int sort_np = 0)
{
    const uint32_t ns = max(1u, (c + L - 1) / L);   // empty tiles still get a unit (background)
    t[0] = c; t[1] = ns; t[2] = ns > 1 ? ns : 0;
    const uint32_t r = c % L;
    t[3] = c / L;
    t[4] = r * 4 >= 3 * L ? 1u : 0u;
    t[5] = (r * 4 < 3 * L && r * 2 >= L) ? 1u : 0u;
    t[6] = (r * 2 < L && r * 4 >= L) ? 1u : 0u;
    t[7] = (r * 4 < L && r != 0) ? 1u : 0u;
    t[8] = c == 0 ? 1u : 0u;
    t[9] = c > 1 ? (c + MP_CHUNK - 1) / MP_CHUNK : 0u;      // 1024-key sort runs (= merge-path chunks) of the tile
    t[10] = c > (uint32_t)SORT_RUN ? 1u : 0u;               // tile needs merging
    (void)sort_np;
}

// The instance count N goes straight to the host: system-scope stores into pinned memory that the waiting host thread
// polls -- no copy-engine hop, no event wake-up latency.  Three self-tagged 64-bit words {call sequence number : value}:
// each is ONE relaxed system-scope store, so no release fence (= write-back of the XCD's dirty L2 lines) is needed to order
// a value before its flag.
__device__ __forceinline__ void publish_counts(int32_t *host_slot, int32_t seq, uint32_t n, uint32_t deepest, uint32_t units)
{
    unsigned long long *hs = reinterpret_cast<unsigned long long *>(host_slot);
    const unsigned long long tag = (unsigned long long)(uint32_t)seq << 32;
    __hip_atomic_store(&hs[2], tag | units, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&hs[1], tag | deepest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&hs[0], tag | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One block of 1024 threads: NSCAN exclusive scans over the tiles at once.  Each wave scans its 64 per-thread
// sums with DPP-free shuffles, the 16 wave totals are scanned by the first wave: two block barriers in all.
__global__ void __launch_bounds__(SCAN_THREADS) tile_scan_kernel(uint32_t *count, uint32_t *offset, uint32_t *cursor,
                                                                 uint32_t *unit_first, uint32_t *mseg_first,
                                                                 uint32_t *class_first, int T, uint32_t L_forced, int32_t *host_slot,
                                                                 int32_t seq, int sort_np, uint32_t *scan_out)
{
    __shared__ uint32_t wave_tot[NSCAN][SCAN_THREADS / WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (T + SCAN_THREADS - 1) / SCAN_THREADS;
    const int b = tid * per, e = min(T, b + per);
    // segment length of this frame: forced, or chosen from the average list length per tile (gms_blend.h)
    uint32_t L = L_forced;
    if (L == 0) {
        __shared__ uint32_t wave_n[SCAN_THREADS / WAVE];
        uint32_t n = 0;
        for (int t = b; t < e; t++) n += count[t];
        for (int d = 32; d >= 1; d >>= 1) n += (uint32_t)__shfl_xor((int)n, d);
        if (lane == 0) wave_n[wave] = n;
        __syncthreads();
        uint64_t total = 0;
        for (int w = 0; w < SCAN_THREADS / WAVE; w++) total += wave_n[w];
        L = total > (uint64_t)SEG_VERY_DEEP_PER_TILE * (uint64_t)T ? SEG_LEN_VERY_DEEP
          : total > (uint64_t)SEG_DEEP_PER_TILE * (uint64_t)T ? SEG_LEN_DEEP : SEG_LEN_SHALLOW;
    }
    uint32_t run[NSCAN], own[NSCAN];
    uint32_t deepest = 0;
#pragma unroll
    for (int k = 0; k < NSCAN; k++) run[k] = 0;
    for (int t = b; t < e; t++) {
        uint32_t q[NSCAN];
        tile_terms(count[t], L, q, sort_np);
        deepest = max(deepest, q[0]);
#pragma unroll
        for (int k = 0; k < NSCAN; k++) run[k] += q[k];
    }
    for (int d = 32; d >= 1; d >>= 1) deepest = max(deepest, (uint32_t)__shfl_xor((int)deepest, d));
    __shared__ uint32_t wave_deep[SCAN_THREADS / WAVE];
    if (lane == 0) wave_deep[wave] = deepest;
    // inclusive scan inside the wave
#pragma unroll
    for (int k = 0; k < NSCAN; k++) {
        own[k] = run[k];
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t x = (uint32_t)__shfl_up((int)run[k], d);
            if (lane >= d) run[k] += x;
        }
        if (lane == WAVE - 1) wave_tot[k][wave] = run[k];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < NSCAN; k++) {
            uint32_t v = lane < SCAN_THREADS / WAVE ? wave_tot[k][lane] : 0u;
            for (int d = 1; d < SCAN_THREADS / WAVE; d <<= 1) {
                const uint32_t x = (uint32_t)__shfl_up((int)v, d);
                if (lane >= d) v += x;
            }
            if (lane < SCAN_THREADS / WAVE) wave_tot[k][lane] = v;      // inclusive over waves
        }
    }
    __syncthreads();
    uint32_t tot[NSCAN];
#pragma unroll
    for (int k = 0; k < NSCAN; k++) {
        tot[k] = wave_tot[k][SCAN_THREADS / WAVE - 1];
        run[k] = run[k] - own[k] + (wave > 0 ? wave_tot[k][wave - 1] : 0u);   // exclusive prefix of this thread's chunk
    }
    // The instance count N goes straight to the host: two system-scope stores into pinned memory (value, then the
    // call's sequence number) that the waiting host thread polls -- no copy-engine hop, no event wake-up latency.
    if (tid == 0) {
        uint32_t dm = 0;
        for (int w = 0; w < SCAN_THREADS / WAVE; w++) dm = max(dm, wave_deep[w]);
        // {N, deepest tile, work units} go to the host right here, as early as they exist: the host thread that waits for N
        // then has the rest of the forward (emit, sort, compositing: ~220 us on the headline scene) to get the backward
        // enqueued before the GPU runs dry.  (Publishing from the emit launch instead was measured: the scan is not
        // shortened by it -- 15.7 -> 15.4 us -- and the host loses 24 us of that slack.)  The device copy serves that variant.
        scan_out[0] = tot[0]; scan_out[1] = dm; scan_out[2] = tot[1]; scan_out[3] = L;
        if (host_slot) publish_counts(host_slot, seq, tot[0], dm, tot[1]);
    }
    for (int t = b; t < e; t++) {
        uint32_t q[NSCAN];
        tile_terms(count[t], L, q, sort_np);
        offset[t] = run[0]; unit_first[t] = run[1]; mseg_first[t] = run[2];
#pragma unroll
        for (int k = 0; k < NCLASS; k++) class_first[k * (T + 1) + t] = run[3 + k];
        class_first[NCLASS * (T + 1) + t] = run[9];
        class_first[(NCLASS + 1) * (T + 1) + t] = run[10];
        cursor[t] = 0;
        count[t] = 0;          // the counter array is library-owned and stays all zero between frames (no memset per frame)
#pragma unroll
        for (int k = 0; k < NSCAN; k++) run[k] += q[k];
    }
    if (tid == 0) {
        offset[T] = tot[0]; unit_first[T] = tot[1]; mseg_first[T] = tot[2];
#pragma unroll
        for (int k = 0; k < NCLASS; k++) class_first[k * (T + 1) + T] = tot[3 + k];
        class_first[NCLASS * (T + 1) + T] = tot[9];
        class_first[(NCLASS + 1) * (T + 1) + T] = tot[10];
    }
}

// unit table, heaviest first: [all full segments | partial last segments | units of empty tiles]
struct FillUnitsArgs {
    const uint32_t *class_first, *offset, *mseg_first;
    uint4 *unit_tile;
    uint2 *deep_tab;
    uint32_t *multi_tab;
    int T, sort_np;
    const uint32_t *scan_out;      // [3] = this frame's segment length
    uint32_t max_units, max_deep, max_multi;
};

// what fill_units needs to know about one tile: its count, its exclusive prefixes `pre` of the NSCAN scanned quantities and
// the totals `tot` (tile_scan_kernel's arrays, or the inline scan of the emit launch)
__device__ __forceinline__ void fill_units_tile(const FillUnitsArgs &f, int t, uint32_t c, uint32_t L, const uint32_t pre[NSCAN], const uint32_t tot[NSCAN])
{
    const uint32_t nfull = c / L;
    uint32_t q[NSCAN];
    tile_terms(c, L, q, f.sort_np);
    const uint32_t nseg = q[1];                          // units of this tile (>= 1)
    const uint4 tail = make_uint4(pre[0], pre[0] + c, 0u, 0u);
    const uint32_t slot0 = pre[2];
    auto put = [&](uint32_t u, uint32_t seg) {
        if (u < f.max_units) {
            f.unit_tile[2 * (size_t)u] = make_uint4((uint32_t)t, seg, nseg, slot0);
            f.unit_tile[2 * (size_t)u + 1] = tail;
        }
    };
    for (uint32_t j = 0, d = pre[9]; j < q[9]; j++, d++)
        if (d < f.max_deep) f.deep_tab[d] = make_uint2((uint32_t)t, j);
    if (q[10]) { const uint32_t d = pre[10]; if (d < f.max_multi) f.multi_tab[d] = (uint32_t)t; }
    uint32_t u = pre[3];
    for (uint32_t s = 0; s < nfull; s++, u++) put(u, s);
    uint32_t base = tot[3];                             // all full units come first
    for (int k = 1; k < NCLASS; k++) {
        if (q[3 + k]) put(base + pre[3 + k], k == NCLASS - 1 ? 0u : nfull);
        base += tot[3 + k];
    }
}

__device__ __forceinline__ void fill_units(const FillUnitsArgs &f, int block)
{
    const int T = f.T;
    const int t = block * BLOCK + threadIdx.x;
    if (t >= T) return;
    const uint32_t c = f.offset[t + 1] - f.offset[t];      // (the counters were cleared by the scan)
    uint32_t pre[NSCAN], tot[NSCAN];
    pre[0] = f.offset[t]; pre[1] = 0u; pre[2] = f.mseg_first[t]; tot[0] = tot[1] = tot[2] = 0u;
#pragma unroll
    for (int k = 0; k < NCLASS + 2; k++) { pre[3 + k] = f.class_first[k * (T + 1) + t]; tot[3 + k] = f.class_first[k * (T + 1) + T]; }
    fill_units_tile(f, t, c, f.scan_out[3], pre, tot);
}

// ---- the tile scan INSIDE the emit launch (round 4).  tile_scan_kernel is one block whose 13 us sit between preprocess and emit
// with the rest of the chip idle.  On the capacity-hint path (every frame but the first of a shape) its work is done redundantly
// instead: every emit block scans the T <= 4096 counters itself (10-16 KB out of L2, ~1.5 us) and keeps the tile offsets in LDS;
// the launch's spare blocks -- one per 256 tiles, as before -- run the full NSCAN-quantity scan, write the tables later launches
// read (tile_offset, unit_first, mseg_first, the totals of class_first, scan_out) and their slice of the unit table, and the first
// of them publishes N to the host.  The counters are cleared by the presort launch, the cursors by preprocess_fwd.
constexpr int INLINE_SCAN_MAX_T = 4096;

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += x;
    }
    return v;
}

// exclusive scan of count[0..T) into s_off[0..T]; all BLOCK threads
__device__ __forceinline__ void inline_offsets(const uint32_t *count, int T, uint32_t *s_off, uint32_t *s_wave /* [4] */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (T + BLOCK - 1) / BLOCK;
    const int b = min(T, tid * per), e = min(T, b + per);
    uint32_t c[16];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { c[k] = (k < per && b + k < e) ? count[b + k] : 0u; sum += c[k]; }
    const uint32_t incl = wave_incl_scan_u32(sum, lane);
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - sum;
    for (int w = 0; w < wave; w++) pre += s_wave[w];
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < per && b + k < e) { s_off[b + k] = pre; pre += c[k]; }
    if (tid == BLOCK - 1) s_off[T] = pre;
    __syncthreads();
}

struct InlineScanLds {
    uint32_t chunk[NSCAN][BLOCK];      // exclusive prefix of every thread's chunk of tiles
    uint32_t wave_tot[NSCAN][BLOCK / WAVE];
    uint32_t total[NSCAN];
    uint32_t wave_deep[BLOCK / WAVE];
    uint32_t deepest;
};

// the spare blocks' scan of all NSCAN quantities, then tile `t`'s prefixes / totals
__device__ __forceinline__ void inline_scan_build(const uint32_t *count, int T, uint32_t L, int sort_np, InlineScanLds &S)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (T + BLOCK - 1) / BLOCK;
    const int b = min(T, tid * per), e = min(T, b + per);
    uint32_t run[NSCAN];
    uint32_t deepest = 0;
#pragma unroll
    for (int k = 0; k < NSCAN; k++) run[k] = 0;
    for (int t = b; t < e; t++) {
        uint32_t q[NSCAN];
        tile_terms(count[t], L, q, sort_np);
        deepest = max(deepest, q[0]);
#pragma unroll
        for (int k = 0; k < NSCAN; k++) run[k] += q[k];
    }
    for (int d = 32; d >= 1; d >>= 1) deepest = max(deepest, (uint32_t)__shfl_xor((int)deepest, d));
    if (lane == 0) S.wave_deep[wave] = deepest;
    uint32_t incl[NSCAN];
#pragma unroll
    for (int k = 0; k < NSCAN; k++) {
        incl[k] = wave_incl_scan_u32(run[k], lane);
        if (lane == WAVE - 1) S.wave_tot[k][wave] = incl[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NSCAN; k++) {
        uint32_t pre = incl[k] - run[k];
        for (int w = 0; w < wave; w++) pre += S.wave_tot[k][w];
        S.chunk[k][tid] = pre;
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < NSCAN; k++) S.total[k] = S.wave_tot[k][0] + S.wave_tot[k][1] + S.wave_tot[k][2] + S.wave_tot[k][3];
        S.deepest = max(max(S.wave_deep[0], S.wave_deep[1]), max(S.wave_deep[2], S.wave_deep[3]));
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------ K3
// The grid's extra blocks (beyond the Gaussians') write the unit table: it only depends on the tile scan, like this
// kernel, so it rides along instead of costing a launch of its own.
struct InlineScanArgs {
    const uint32_t *count;         // [T] per-tile instance counters (preprocess_fwd)
    uint32_t *offset, *unit_first, *mseg_first, *class_first, *scan_out;
    uint32_t L;                    // this frame's segment length (forced: the inline path is not taken when the scan has to choose it)
};

template <bool INLINE>
__global__ void __launch_bounds__(BLOCK) emit_instances_kernel(int P, int gx, int gy, const int *radii, GeomState geom,
                                                               const uint32_t *tile_offset, uint32_t *tile_cursor,
                                                               uint64_t *keys, uint64_t capacity, unsigned emit_blocks, FillUnitsArgs fu,
                                                               int32_t *host_slot, int32_t seq, const uint32_t *scan_out, InlineScanArgs is)
{
    // one LDS buffer: emit blocks = [tile offsets (inline scan) | tile table keys, counts, bases | wave sums]; spare blocks = InlineScanLds
    constexpr int OFFW = INLINE ? INLINE_SCAN_MAX_T + 4 : 0;
    __shared__ uint32_t smem[OFFW + 3 * TT_SLOTS + BLOCK / WAVE];
    static_assert(sizeof(InlineScanLds) <= sizeof(uint32_t) * (INLINE_SCAN_MAX_T + 4 + 3 * TT_SLOTS), "spare blocks alias the emit blocks' LDS");
    uint32_t *s_off = smem, *s_wave = smem + OFFW + 3 * TT_SLOTS;
    // INLINE: the spare blocks come FIRST in the grid (the first of them publishes N: the host should not wait for it behind
    // thousands of emit blocks); otherwise they follow the emit blocks as before
    const unsigned nspare = gridDim.x - emit_blocks;
    const bool spare = INLINE ? blockIdx.x < nspare : blockIdx.x >= emit_blocks;
    const unsigned spare_idx = INLINE ? blockIdx.x : blockIdx.x - emit_blocks;
    const unsigned emit_idx = INLINE ? blockIdx.x - nspare : blockIdx.x;
    if (spare) {
        if (INLINE) {
            InlineScanLds &S = *reinterpret_cast<InlineScanLds *>(smem);
            const int T = fu.T;
            inline_scan_build(is.count, T, is.L, fu.sort_np, S);
            const int t = (int)spare_idx * BLOCK + (int)threadIdx.x;
            if (spare_idx == 0 && threadIdx.x == 0) {
                is.scan_out[0] = S.total[0]; is.scan_out[1] = S.deepest; is.scan_out[2] = S.total[1]; is.scan_out[3] = is.L;
                if (host_slot) publish_counts(host_slot, seq, S.total[0], S.deepest, S.total[1]);
                is.offset[T] = S.total[0]; is.unit_first[T] = S.total[1]; is.mseg_first[T] = S.total[2];
#pragma unroll
                for (int k = 0; k < NCLASS + 2; k++) is.class_first[k * (T + 1) + T] = S.total[3 + k];
            }
            if (t < T) {
                const int per = (T + BLOCK - 1) / BLOCK;
                const int c0 = t / per;
                uint32_t pre[NSCAN], tot[NSCAN];
#pragma unroll
                for (int k = 0; k < NSCAN; k++) { pre[k] = S.chunk[k][c0]; tot[k] = S.total[k]; }
                for (int u = c0 * per; u < t; u++) {
                    uint32_t q[NSCAN];
                    tile_terms(is.count[u], is.L, q, fu.sort_np);
#pragma unroll
                    for (int k = 0; k < NSCAN; k++) pre[k] += q[k];
                }
                is.offset[t] = pre[0]; is.unit_first[t] = pre[1]; is.mseg_first[t] = pre[2];
#pragma unroll
                for (int k = 0; k < NCLASS + 2; k++) is.class_first[k * (T + 1) + t] = pre[3 + k];      // (a re-run of the tail reads them)
                fill_units_tile(fu, t, is.count[t], is.L, pre, tot);
            }
            return;
        }
        // (first spare block: this frame's counts go to the host from here when the scan left that to us)
        if (spare_idx == 0 && threadIdx.x == 0 && host_slot) publish_counts(host_slot, seq, scan_out[0], scan_out[1], scan_out[2]);
        fill_units(fu, (int)spare_idx);
        return;
    }
    if (INLINE) { inline_offsets(is.count, fu.T, s_off, s_wave); tile_offset = s_off; }
    const int i = (int)emit_idx * BLOCK + threadIdx.x;
    // (the three loads are independent: the pixel centre and the depth of a culled Gaussian are stale words that nothing reads --
    // loading them only after `r > 0` was known put a second memory round trip into a kernel that waits 85 % of its cycles)
    // (two independent loads, 12 bytes per Gaussian: the tile rectangle preprocess_fwd counted over -- all zero for a culled Gaussian --
    //  and the depth; rounds 1-5 read the radius and the pixel centre, i.e. every cache line of the 48-byte records for 8 bytes each)
    const ushort4 rc = i < P ? geom.rect[i] : make_ushort4(0, 0, 0, 0);
    const uint32_t dbits = i < P ? __float_as_uint(geom.depth[i]) : 0u;
    const int minx = rc.x, miny = rc.y, maxx = rc.z, maxy = rc.w;
    const int rw = maxx - minx;
    const int area = rw * (maxy - miny);
    const uint64_t key = area > 0 ? (((uint64_t)dbits << 32) | (uint32_t)i) : 0ull;
    (void)radii; (void)gy;
    const bool small = area <= SMALL_AREA;
    const int lane = threadIdx.x & 63;
    int cx = minx, cy = miny;
    // footprints up to SMALL_AREA tiles: count per tile in the block's LDS table, fetch one cursor range per
    // distinct tile (one global atomic each, all in flight together), then hand out ranks from LDS
    int *tt_key = reinterpret_cast<int *>(smem + OFFW);
    uint32_t *tt_cnt = smem + OFFW + TT_SLOTS, *tt_base = smem + OFFW + 2 * TT_SLOTS;
    for (int q = threadIdx.x; q < TT_SLOTS; q += BLOCK) { tt_key[q] = -1; tt_cnt[q] = 0; }
    __syncthreads();
    if (small) {
        for (int k = 0; k < area; k++) {
            const int sl = tt_insert(tt_key, cy * gx + cx);
            if (sl >= 0) atomicAdd(&tt_cnt[sl], 1u);
            if (++cx == maxx) { cx = minx; cy++; }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < TT_SLOTS; q += BLOCK)
        if (tt_key[q] >= 0) {
            tt_base[q] = tile_offset[tt_key[q]] + atomicAdd(&tile_cursor[tt_key[q]], tt_cnt[q]);
            tt_cnt[q] = 0;
        }
    __syncthreads();
    if (small) {
        cx = minx; cy = miny;
        for (int k = 0; k < area; k++) {
            const int t = cy * gx + cx;
            const int sl = tt_find(tt_key, t);
            const uint64_t slot = sl >= 0 ? (uint64_t)tt_base[sl] + atomicAdd(&tt_cnt[sl], 1u)
                                          : (uint64_t)tile_offset[t] + atomicAdd(&tile_cursor[t], 1u);
            if (slot < capacity) keys[slot] = key;
            if (++cx == maxx) { cx = minx; cy++; }
        }
    }
    // large footprints: the wave expands one splat at a time, a tile per lane
    uint64_t big = __ballot(!small);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const int bminx = __builtin_amdgcn_readlane(minx, src), bminy = __builtin_amdgcn_readlane(miny, src);
        const int brw = __builtin_amdgcn_readlane(rw, src), barea = __builtin_amdgcn_readlane(area, src);
        const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src);
        const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), src);
        const uint64_t bkey = ((uint64_t)khi << 32) | klo;
        for (int k = lane; k < barea; k += WAVE) {
            const int ty = k / brw, tx = k - ty * brw;
            const int t = (bminy + ty) * gx + bminx + tx;
            const uint64_t slot = (uint64_t)tile_offset[t] + atomicAdd(&tile_cursor[t], 1u);
            if (slot < capacity) keys[slot] = bkey;
        }
    }
}

// ------------------------------------------------------------------------------------ K4
// Per-tile sort of (depth bits << 32 | id) keys: an all-ascending bitonic network (mirror step +
// half-cleaners), so virtual +inf padding works for any length.  Compare-exchange index i always
// touches the 128-key block i/64, and thread t owns indices t, t+THREADS, ...: every sub-stage
// whose span is <= 128 keys stays inside one wave's blocks and needs only wave-level ordering
// (LDS operations of a wave execute in order) -- block barriers are paid only for the wide
// sub-stages (6 instead of 55 for 1024 keys).
__device__ __forceinline__ void ce(uint64_t &x, uint64_t &y)
{
    if (x > y) { uint64_t t = x; x = y; y = t; }
}


constexpr int WAVE_SPAN = 128;   // keys covered by one wave's 64 compare-exchanges

template <int THREADS>
__device__ __forceinline__ void stage_sync(int span, int &prev_span)
{
    if (span > WAVE_SPAN || prev_span > WAVE_SPAN) __syncthreads();
    else wave_sync();
    prev_span = span;
}

// sort m (power of two) keys in LDS: merges k = k0 .. m; with mirror=false the mirror step of the
// first merge is skipped (the caller did it and the wide strides in global memory).
template <int THREADS>
__device__ void lds_bitonic(uint64_t *s, int m, int k0, int j0, bool first_mirror, int tid)
{
    int prev = 1 << 30;
    for (int k = k0, lk = 31 - __builtin_clz(k0); k <= m; k <<= 1, lk++) {   // all sizes are powers of two:
        const int hk = k >> 1;                                                // index math with shifts/masks only
        if (k > k0 || first_mirror) {
            stage_sync<THREADS>(k, prev);
            for (int i = tid; i < (m >> 1); i += THREADS) {
                const int blk = i >> (lk - 1), off = i & (hk - 1);
                const int lo = (blk << lk) + off, hi = (blk << lk) + k - 1 - off;
                ce(s[lo], s[hi]);
            }
        }
        int j = (k > k0 || first_mirror) ? (k >> 2) : j0;
        for (int lj = j > 0 ? 31 - __builtin_clz(j) : 0; j >= 1; j >>= 1, lj--) {
            stage_sync<THREADS>(2 * j, prev);
            for (int i = tid; i < (m >> 1); i += THREADS) {
                const int lo = ((i >> lj) << (lj + 1)) + (i & (j - 1));
                ce(s[lo], s[lo + j]);
            }
        }
    }
    __syncthreads();
}

// Per-tile sort in three stages.
//  (1) presort: grid (T, 8), 256 threads: block (t, c) sorts runs c, c+8, ... of SORT_RUN keys of tile t in LDS (the
//      runs of one tile sort on different CUs).
//  (2) when some tile is deeper than SORT_BIG_CHUNK keys: merge-path passes for every multi-run tile, run width
//      doubling per pass; every block produces one 1024-key chunk of the output from the co-ranks of its two ends
//      (found by a wave-wide 64-ary search), so a deep tile's merge is spread over as many CUs as it has chunks and
//      each pass is a streaming read + write of the keys that still need merging.  Data ping-pongs between `keys` and a scratch buffer (the segment-state area, unused until
//      compositing); the presort writes a tile's runs to the side that makes its LAST pass land in `keys`.  The host
//      launches as many passes as the deepest tile of the PREVIOUS frame needs (the count travels back with N);
//  (3) merge: 1024 threads per tile with 2..8 runs: loads the tile (<= SORT_BIG_CHUNK keys) into LDS and runs only the
//      bitonic merge levels above SORT_RUN.  A tile deeper than the launched passes cover (scene changed since the last
//      frame) is sorted from scratch here by one block: slow, but correct, and the next frame launches enough passes.

enum { SORT_SINGLE = 0, SORT_LDS, SORT_MERGEPATH, SORT_FALLBACK };

__host__ __device__ __forceinline__ int sort_passes(uint64_t n)       // merge levels above SORT_RUN-key runs
{
    const uint64_t runs = (n + SORT_RUN - 1) / SORT_RUN;
    int q = 0;
    while ((1ull << q) < runs) q++;
    return q;
}

__device__ __forceinline__ int sort_class(uint64_t n, int passes_launched, int &q)
{
    q = 0;
    if (n <= (uint64_t)SORT_RUN) return SORT_SINGLE;
    q = sort_passes(n);
    if (q <= passes_launched) return SORT_MERGEPATH;      // passes are launched: every multi-run tile they cover takes them
    return n <= (uint64_t)SORT_BIG_CHUNK ? SORT_LDS : SORT_FALLBACK;
}

__global__ void __launch_bounds__(256) tile_presort_kernel(const uint32_t *tile_offset, const uint32_t *run_total, const uint2 *run_tab,
                                                           uint32_t max_runs, uint64_t *keys, uint64_t *tmp, uint64_t capacity,
                                                           int passes_launched, uint32_t *zero_count, int T)
{
    __shared__ uint64_t s[SORT_RUN];
    const int tid = threadIdx.x;
    // inline-scan frames: the per-tile counters were read by the emit launch; they are cleared here (all zero between frames)
    if (zero_count)
        for (int q = (int)blockIdx.x * 256 + tid; q < T; q += (int)gridDim.x * 256) zero_count[q] = 0u;
    if (blockIdx.x >= run_total[0] || blockIdx.x >= max_runs) return;      // one block per (tile, run) of the run table
    const uint2 tr = run_tab[blockIdx.x];
    const uint64_t beg = tile_offset[tr.x], end64 = tile_offset[tr.x + 1];
    if (end64 > capacity) return;                 // overflowed launch: results are discarded by the host
    const long n = (long)(end64 - beg);
    int q;
    const int cls = sort_class((uint64_t)n, passes_launched, q);
    if (cls == SORT_FALLBACK) return;
    uint64_t *dst_base = (cls == SORT_MERGEPATH && (q & 1)) ? tmp : keys;
    const long c0 = (long)tr.y * SORT_RUN;
    const int cn = (int)min((long)SORT_RUN, n - c0);
    const uint64_t *g = keys + beg + c0;
    uint64_t *d = dst_base + beg + c0;
    if (cn <= 1) { if (cn == 1 && tid == 0) d[0] = g[0]; return; }
    int m = 2;
    while (m < cn) m <<= 1;
    for (int i = tid; i < m; i += 256) s[i] = i < cn ? g[i] : ~0ull;
    lds_bitonic<256>(s, m, 2, 0, true, tid);
    for (int i = tid; i < cn; i += 256) d[i] = s[i];
}

// ---- register-resident presort ------------------------------------------------------------------------------------
// The same all-ascending bitonic network (mirror step + half-cleaners: every step pairs element i with i ^ M) on a run of
// <= 1024 keys, but with the keys in REGISTERS: thread t of 256 holds keys 4t .. 4t+3.  A step whose mask only touches bits
// 0-1 is a compare-exchange between the thread's own registers; bits 2-7 select the partner LANE (lane ^ (M >> 2)): DPP row
// operations for the lane masks 1, 2, 3, 4, 7, 8, 15 (full rate, no LDS), ds_bpermute for those that cross a 16-lane row; only
// the four steps with bits 8-9 (k = 512 mirror, k = 1024 mirror, strides 512 and 256) go through LDS.  The LDS version pays an
// LDS round trip and a barrier or wave sync for each of its 55 steps (a lone 1024-key block takes ~25 us: pure latency, and
// the launch is a single round of blocks); this one pays four.
template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
template <int ML>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v)
{
    if (ML == 1) return dpp64<0xB1>(v);            // quad_perm [1,0,3,2]
    if (ML == 2) return dpp64<0x4E>(v);            // quad_perm [2,3,0,1]
    if (ML == 3) return dpp64<0x1B>(v);            // quad_perm [3,2,1,0]
    if (ML == 4) return dpp64<0x1B>(dpp64<0x141>(v));      // 7 - i within the half row, then the quad reversed: i ^ 4
    if (ML == 7) return dpp64<0x141>(v);           // row_half_mirror
    if (ML == 8) return dpp64<0x128>(v);           // row_ror:8
    if (ML == 15) return dpp64<0x140>(v);          // row_mirror
    const int lo = __shfl_xor((int)(uint32_t)v, ML), hi = __shfl_xor((int)(uint32_t)(v >> 32), ML);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// one step of the network: element i = 4 t + r is paired with i ^ M; the element whose bit TOP(M) is clear keeps the minimum
template <int M>
__device__ __forceinline__ void sort_step(uint64_t (&k)[4], int t, uint64_t *lds)
{
    constexpr int MR = M & 3, ML = (M >> 2) & 63, MW = M >> 8;
    constexpr int TOP = M >= 512 ? 512 : M >= 256 ? 256 : M >= 128 ? 128 : M >= 64 ? 64 : M >= 32 ? 32 : M >= 16 ? 16 : M >= 8 ? 8 : M >= 4 ? 4 : M >= 2 ? 2 : 1;
    if (ML == 0 && MW == 0) {          // both elements in this thread
        if (MR == 1) { ce(k[0], k[1]); ce(k[2], k[3]); }
        else if (MR == 2) { ce(k[0], k[2]); ce(k[1], k[3]); }
        else { ce(k[0], k[3]); ce(k[1], k[2]); }
        return;
    }
    uint64_t p[4];
    if (MW != 0) {                     // partner in another wave: through LDS
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) lds[4 * t + r] = k[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) p[r] = lds[(4 * t + r) ^ M];
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) p[r] = lane_xor64<ML>(k[r ^ MR]);
    }
    const bool keep_min = ((4 * t) & TOP) == 0;          // (TOP >= 4 here: the same for the thread's four elements)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const bool less = p[r] < k[r];
        k[r] = (less == keep_min) ? p[r] : k[r];
    }
}
template <int J>
__device__ __forceinline__ void sort_half_cleaners(uint64_t (&k)[4], int t, uint64_t *lds)
{
    if constexpr (J >= 1) {
        sort_step<J>(k, t, lds);
        sort_half_cleaners<J / 2>(k, t, lds);
    }
}
template <int K>
__device__ __forceinline__ void sort_level(uint64_t (&k)[4], int t, uint64_t *lds)
{
    sort_step<K - 1>(k, t, lds);               // mirror: i with i ^ (K - 1)
    sort_half_cleaners<K / 4>(k, t, lds);      // strides K/4 ... 1
}

__global__ void __launch_bounds__(256) tile_presort_reg_kernel(const uint32_t *tile_offset, const uint32_t *run_total, const uint2 *run_tab,
                                                           uint32_t max_runs, uint64_t *keys, uint64_t *tmp, uint64_t capacity,
                                                           int passes_launched, uint32_t *zero_count, int T)
{
    __shared__ uint64_t s[SORT_RUN];
    const int tid = threadIdx.x;
    // inline-scan frames: the per-tile counters were read by the emit launch; they are cleared here (all zero between frames)
    if (zero_count)
        for (int q = (int)blockIdx.x * 256 + tid; q < T; q += (int)gridDim.x * 256) zero_count[q] = 0u;
    if (blockIdx.x >= run_total[0] || blockIdx.x >= max_runs) return;      // one block per (tile, run) of the run table
    const uint2 tr = run_tab[blockIdx.x];
    const uint64_t beg = tile_offset[tr.x], end64 = tile_offset[tr.x + 1];
    if (end64 > capacity) return;                 // overflowed launch: results are discarded by the host
    const long n = (long)(end64 - beg);
    int q;
    const int cls = sort_class((uint64_t)n, passes_launched, q);
    if (cls == SORT_FALLBACK) return;
    uint64_t *dst_base = (cls == SORT_MERGEPATH && (q & 1)) ? tmp : keys;
    const long c0 = (long)tr.y * SORT_RUN;
    const int cn = (int)min((long)SORT_RUN, n - c0);
    const uint64_t *g = keys + beg + c0;
    uint64_t *d = dst_base + beg + c0;
    if (cn <= 1) { if (cn == 1 && tid == 0) d[0] = g[0]; return; }
    int m = 2;
    while (m < cn) m <<= 1;
    uint64_t k[4];
#pragma unroll
    for (int r = 0; r < 4; r++) k[r] = 4 * tid + r < cn ? g[4 * tid + r] : ~0ull;      // virtual +inf padding
    sort_level<2>(k, tid, s);
    sort_level<4>(k, tid, s);
    if (m >= 8) sort_level<8>(k, tid, s);
    if (m >= 16) sort_level<16>(k, tid, s);
    if (m >= 32) sort_level<32>(k, tid, s);
    if (m >= 64) sort_level<64>(k, tid, s);
    if (m >= 128) sort_level<128>(k, tid, s);
    if (m >= 256) sort_level<256>(k, tid, s);
    if (m >= 512) sort_level<512>(k, tid, s);
    if (m >= 1024) sort_level<1024>(k, tid, s);
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (4 * tid + r < cn) d[4 * tid + r] = k[r];
}

// number of elements taken from A among the first o outputs of merge(A, B) (keys are unique)
template <typename PA, typename PB>
__device__ __forceinline__ int co_rank(int o, PA A, int la, PB B, int lb)
{
    int lo = max(0, o - lb), hi = min(o, la);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A[mid] < B[o - mid - 1]) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// The same co-rank found by a whole wave: 64 probes per round instead of one, so a 16 k-wide search takes three memory
// round trips instead of fourteen (the block cannot start loading before it knows both ends of its chunk).
__device__ __forceinline__ int co_rank_wave(int o, const uint64_t *A, int la, const uint64_t *B, int lb)
{
    const int lane = threadIdx.x & 63;
    int lo = max(0, o - lb), hi = min(o, la);
    while (lo < hi) {
        const int step = (hi - lo + 63) / 64;
        const int mid = lo + lane * step;
        const bool need_more = mid < hi && A[mid] < B[o - mid - 1];      // monotone in mid: true ... true false ... false
        const int cnt = __builtin_popcountll(__ballot(need_more));
        const int nlo = cnt > 0 ? lo + (cnt - 1) * step + 1 : lo;
        const int nhi = (cnt < 64 && lo + cnt * step < hi) ? lo + cnt * step : hi;
        lo = nlo; hi = nhi;
    }
    return lo;
}

__global__ void __launch_bounds__(256) tile_mergepath_kernel(const uint32_t *tile_offset, const uint32_t *deep_first, const uint2 *deep_tab,
                                                             uint32_t max_deep, uint64_t *keys, uint64_t *tmp, uint64_t capacity,
                                                             int passes_launched, int pass)
{
    __shared__ uint64_t s[MP_CHUNK];
    __shared__ int s_rank[2];
    const int tid = threadIdx.x;
    if (blockIdx.x >= deep_first[0] || blockIdx.x >= max_deep) return;      // deep_first[0] here = total number of chunks
    const uint2 tc = deep_tab[blockIdx.x];
    const uint64_t beg = tile_offset[tc.x], end64 = tile_offset[tc.x + 1];
    if (end64 > capacity) return;
    const long n = (long)(end64 - beg);
    int q;
    if (sort_class((uint64_t)n, passes_launched, q) != SORT_MERGEPATH || pass >= q) return;
    const bool src_is_tmp = ((q - pass) & 1) != 0;
    const uint64_t *src = (src_is_tmp ? tmp : keys) + beg;
    uint64_t *dst = (src_is_tmp ? keys : tmp) + beg;
    const long w = (long)SORT_RUN << pass;
    const long j0 = (long)tc.y * MP_CHUNK;
    const long base = (j0 / (2 * w)) * (2 * w);
    const long a1 = min(n, base + w), b1 = min(n, base + 2 * w);
    const int la = (int)(a1 - base), lb = (int)(b1 - a1);
    const uint64_t *A = src + base, *B = src + a1;
    const int o0 = (int)(j0 - base), o1 = (int)min((long)(o0 + MP_CHUNK), (long)(la + lb));
    if (tid < 64) { const int r = co_rank_wave(o0, A, la, B, lb); if (tid == 0) s_rank[0] = r; }
    else if (tid < 128) { const int r = co_rank_wave(o1, A, la, B, lb); if (tid == 64) s_rank[1] = r; }
    __syncthreads();
    const int ra0 = s_rank[0], ra1 = s_rank[1];
    const int rb0 = o0 - ra0, rb1 = o1 - ra1;
    const int na = ra1 - ra0, nb = rb1 - rb0;
    for (int i = tid; i < na; i += 256) s[i] = A[ra0 + i];
    for (int i = tid; i < nb; i += 256) s[na + i] = B[rb0 + i];
    __syncthreads();
    // each thread merges four consecutive outputs from its own co-rank inside the LDS pieces
    const int cnt = o1 - o0;
    const int lo = tid * 4;
    if (lo < cnt) {
        const uint64_t *LA = s, *LB = s + na;
        int ia = co_rank(lo, LA, na, LB, nb), ib = lo - ia;
        uint64_t out[4];
        const int m = min(4, cnt - lo);
        for (int k = 0; k < m; k++) {
            const bool takeA = ib >= nb || (ia < na && LA[ia] < LB[ib]);
            out[k] = takeA ? LA[ia++] : LB[ib++];
        }
        for (int k = 0; k < m; k++) dst[base + o0 + lo + k] = out[k];
    }
}

__global__ void __launch_bounds__(1024) tile_merge_kernel(const uint32_t *tile_offset, const uint32_t *multi_total, const uint32_t *multi_tab,
                                                          uint32_t max_multi, uint64_t *keys, uint64_t capacity, int passes_launched)
{
    constexpr int THREADS = 1024, CHUNK = SORT_BIG_CHUNK;
    __shared__ uint64_t s[CHUNK];
    const int tid = threadIdx.x;
    if (blockIdx.x >= multi_total[0] || blockIdx.x >= max_multi) return;     // one block per tile with more than one run
    const uint32_t tile = multi_tab[blockIdx.x];
    const uint64_t beg = tile_offset[tile], end64 = tile_offset[tile + 1];
    if (end64 > capacity) return;
    const long n = (long)(end64 - beg);
    int q;
    const int cls = sort_class((uint64_t)n, passes_launched, q);
    if (cls == SORT_SINGLE || cls == SORT_MERGEPATH) return;
    uint64_t *g = keys + beg;
    long np2 = 2;
    while (np2 < n) np2 <<= 1;
    const uint64_t INF = ~0ull;

    if (cls == SORT_LDS) {
        // The runs of SORT_RUN keys are sorted: merge them level by level inside LDS with merge path -- every thread
        // finds the co-rank of its K consecutive outputs by binary search, merges them into registers, and the level
        // is written back after a barrier (no second buffer).  ~30 dependent LDS reads per level instead of the
        // 11-13 compare-exchange sub-stages of a bitonic merge level.
        const int m = (int)np2;                   // 2048, 4096 or 8192: runs padded with +inf
        for (int i = tid; i < m; i += THREADS) s[i] = i < n ? g[i] : INF;
        __syncthreads();
        const int K = m / THREADS;                // 2, 4 or 8 outputs per thread
        for (int w = SORT_RUN; w < m; w <<= 1) {
            const int o = tid * K;
            const int base = (o / (2 * w)) * (2 * w);
            const uint64_t *LA = s + base, *LB = s + base + w;
            const int lo = o - base;
            int ia = co_rank(lo, LA, w, LB, w), ib = lo - ia;
            uint64_t out[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k < K) {
                    const bool takeA = ib >= w || (ia < w && LA[ia] < LB[ib]);
                    out[k] = takeA ? LA[ia++] : LB[ib++];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < K) s[o + k] = out[k];
            __syncthreads();
        }
        for (int i = tid; i < n; i += THREADS) g[i] = s[i];
        return;
    }
    // fallback (deeper than the launched merge-path passes cover): sort CHUNK-sized runs in LDS, then merge with
    // wide strides in global memory, all by this one block
    for (long c = 0; c < n; c += CHUNK) {
        for (int i = tid; i < CHUNK; i += THREADS) s[i] = (c + i) < n ? g[c + i] : INF;
        lds_bitonic<THREADS>(s, CHUNK, 2, 0, true, tid);
        for (int i = tid; i < CHUNK; i += THREADS)
            if (c + i < n) g[c + i] = s[i];
        __syncthreads();
    }
    for (long k = 2L * CHUNK; k <= np2; k <<= 1) {
        const long hk = k >> 1;
        for (long i = tid; i < (np2 >> 1); i += THREADS) {        // mirror step
            long blk = i / hk, off = i - blk * hk;
            long lo = blk * k + off, hi = blk * k + k - 1 - off;
            if (hi < n) { uint64_t x = g[lo], y = g[hi]; if (x > y) { g[lo] = y; g[hi] = x; } }
        }
        __syncthreads();
        for (long j = k >> 2; j >= CHUNK; j >>= 1) {              // strides that cross LDS chunks
            for (long i = tid; i < (np2 >> 1); i += THREADS) {
                long lo = 2 * j * (i / j) + (i % j), hi = lo + j;
                if (hi < n) { uint64_t x = g[lo], y = g[hi]; if (x > y) { g[lo] = y; g[hi] = x; } }
            }
            __syncthreads();
        }
        for (long c = 0; c < n; c += CHUNK) {                      // remaining strides inside a chunk
            for (int i = tid; i < CHUNK; i += THREADS) s[i] = (c + i) < n ? g[c + i] : INF;
            lds_bitonic<THREADS>(s, CHUNK, CHUNK, CHUNK >> 1, false, tid);
            for (int i = tid; i < CHUNK; i += THREADS)
                if (c + i < n) g[c + i] = s[i];
            __syncthreads();
        }
    }
}

__global__ void fill_background_kernel(int W, int H, const float *bg, float *out_color, float *out_invdepth)
{
    const size_t HW = (size_t)W * H;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    out_color[i] = bg[0]; out_color[HW + i] = bg[1]; out_color[2 * HW + i] = bg[2];
    out_invdepth[i] = 0.f;
}

__global__ void mark_visible_kernel(int P, const float *means3D, const float *view, uint8_t *present)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float vx, vy, vz;
    view_transform(view, means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2], vx, vy, vz);
    present[i] = vz > NEAR_Z ? 1 : 0;
}

// pinned, device-visible read-back slot, one per host thread: 64-bit words {call sequence number : N}, {.. : deepest tile},
// {.. : work units}
// (slot 0: the blocking forward; slots 1 .. DEFER_SLOTS: the ring of the deferred read-back, gms_rasterize_forward_counts)
constexpr int DEFER_SLOTS = 16, SLOT_WORDS = 16;
static int32_t *pinned_slot(int index = 0)
{
    static thread_local int32_t *slot = nullptr;
    if (!slot) {
        if (hipHostMalloc((void **)&slot, 64 * (1 + DEFER_SLOTS), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) slot = nullptr;
        else { for (int k = 0; k < SLOT_WORDS * (1 + DEFER_SLOTS); k++) slot[k] = 0; }
    }
    return slot ? slot + SLOT_WORDS * index : nullptr;
}

// Wait until the tile_scan kernel of call `seq` has published N.  The host spins on the pinned slot (the kernel
// is at most one forward + one backward away); after 10 s without an answer the stream is synchronised so a
// faulted kernel surfaces as an error instead of a hang.
static std::atomic<int64_t> g_wait_ns{0}, g_wait_calls{0};
static thread_local uint32_t t_last_deepest = 0;     // deepest tile of this thread's most recent forward

static int32_t wait_for_count(int32_t *slot, int32_t seq, hipStream_t stream, int64_t *N)
{
    const auto t0 = std::chrono::steady_clock::now();
    struct Tally {
        std::chrono::steady_clock::time_point t0;
        ~Tally() { g_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_wait_calls++; }
    } tally{t0};
    const unsigned long long *hs = reinterpret_cast<const unsigned long long *>(slot);
    const unsigned long long tag = (unsigned long long)(uint32_t)seq;
    auto ready = [&]() {                       // both self-tagged words of this call have arrived
        return (__atomic_load_n(&hs[0], __ATOMIC_ACQUIRE) >> 32) == tag && (__atomic_load_n(&hs[1], __ATOMIC_ACQUIRE) >> 32) == tag &&
               (__atomic_load_n(&hs[2], __ATOMIC_ACQUIRE) >> 32) == tag;
    };
    for (uint32_t spins = 0;; spins++) {
        if (ready()) break;
        __builtin_ia32_pause();
        if ((spins & 0xffffu) == 0xffffu &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            GMS_HIP_CHECK(hipStreamSynchronize(stream));
            if (!ready()) {
                set_error("tile_scan did not publish the instance count");
                return GMS_ERR_HIP;
            }
            break;
        }
    }
    *N = (int64_t)(uint32_t)(__atomic_load_n(&hs[0], __ATOMIC_RELAXED) & 0xffffffffull);
    return GMS_OK;
}
static uint32_t unit_count(const int32_t *slot)
{
    return (uint32_t)(__atomic_load_n(reinterpret_cast<const unsigned long long *>(slot) + 2, __ATOMIC_RELAXED) & 0xffffffffull);
}
static uint32_t deepest_tile(const int32_t *slot)
{
    return (uint32_t)(__atomic_load_n(reinterpret_cast<const unsigned long long *>(slot) + 1, __ATOMIC_RELAXED) & 0xffffffffull);
}

// GMS_MICRO: 1 = micro-tile compositing always, 0 = the quadrant-wave kernels always, unset = per frame: micro-tile for
// shallow scenes (at most 512 list entries per tile on average: mesh surfaces, translucent splats), quadrant waves for deep
// opaque ones (FLAME heads with 100 splats per face: most of a tile's list is never composited, and the two-phase
// products of blend.hip stop at the visible depth while the filter would still test every entry).
int micro_setting()
{
    static int m = -2;
    if (m == -2) { const char *e = getenv("GMS_MICRO"); m = e ? (atoi(e) != 0 ? 1 : 0) : (GMS_MICRO_DEFAULT ? -1 : 0); }
    return m;
}
bool micro_mode() { return micro_setting() != 0; }          // the micro-tile path may be taken: buffers carry its lists
bool use_micro(uint64_t capacity, int T) { const int m = micro_setting(); return m == 1 || (m < 0 && capacity <= 512ull * (uint64_t)T); }

uint32_t seg_len_forced()
{
    static int64_t L = -1;
    if (L < 0) {
        uint32_t v = 0;
        if (const char *e = getenv("GMS_SEG_LEN")) { v = (uint32_t)atoi(e); if (v < 64) v = 64; v = (v + 63u) / 64u * 64u; }
        if (micro_mode()) { if (v == 0) v = SEG_LEN_MICRO; if (v > 256u) v = 256u; }      // one L for every frame; entry indices are bytes
        L = v;
    }
    return (uint32_t)L;
}
uint32_t seg_len_min() { const uint32_t f = seg_len_forced(); return f ? f : SEG_LEN_SHALLOW; }
// The L handed to the tile scan.  Micro-tile mode pins one L (<= 256) for the frames its kernels take; a frame whose
// binning capacity says "very deep" goes to the quadrant kernels anyway (use_micro) and gets their long segments
// (capacity > SEG_VERY_DEEP_PER_TILE * T implies capacity > 512 * T: never a micro frame).  GMS_SEG_LEN overrides everything.
static uint32_t seg_len_for_frame(uint64_t capacity_hint, int T)
{
    const uint32_t f = seg_len_forced();
    if (f == 0 || getenv("GMS_SEG_LEN") || micro_setting() == 1) return f;      // (GMS_MICRO=1 forces the micro kernels on every frame)
    return capacity_hint > (uint64_t)SEG_VERY_DEEP_PER_TILE * (uint64_t)T ? SEG_LEN_VERY_DEEP : f;
}

uint32_t unit_run()
{
    static uint32_t R = 0;
    if (R == 0) {
        uint32_t v = 4;
        if (const char *e = getenv("GMS_UNIT_RUN")) v = (uint32_t)atoi(e);
        R = 1;
        while (R * 2 <= v && R < 64) R *= 2;
    }
    return R;
}

}  // namespace gms

using namespace gms;

extern "C" int32_t gms_abi_version(void) { return GMS_ABI_VERSION; }
extern "C" const char *gms_last_error(void) { return gms::g_err; }
extern "C" size_t gms_geom_bytes(int32_t P) { return GeomState::bytes((size_t)(P > 0 ? P : 1)); }
extern "C" size_t gms_image_bytes(int32_t w, int32_t h) { return ImageState::bytes((size_t)w, (size_t)h); }
extern "C" size_t gms_image_n_contrib_offset(int32_t w, int32_t h)
{
    char *const base = reinterpret_cast<char *>(uintptr_t(1) << 20);
    ImageState s = ImageState::carve(base, (size_t)w, (size_t)h);
    return (size_t)(reinterpret_cast<char *>(s.n_contrib) - base);
}
extern "C" size_t gms_image_counts_offset(int32_t w, int32_t h)
{
    char *const base = reinterpret_cast<char *>(uintptr_t(1) << 20);
    ImageState s = ImageState::carve(base, (size_t)w, (size_t)h);
    return (size_t)(reinterpret_cast<char *>(s.scan_out) - base);
}
static thread_local int64_t t_last_launched_units = 0;
// Launch-size hints learnt from earlier frames, per (host thread, device, stream, W, H, P)
struct FrameState { int device; hipStream_t stream; int W, H, P; uint32_t deepest_seen, units_hint; };
static thread_local std::vector<FrameState> t_frames;
// what a redeemed ticket learnt about a frame (gms_rasterize_forward_counts, possibly on another host thread): picked up by the next forward
// of the same (device, stream, W, H, P)
struct HintMail { int device; hipStream_t stream; int W, H, P; uint32_t deepest, units; };
static std::mutex g_mail_mu;
static std::vector<HintMail> g_mail;
static FrameState *frame_state(int device, hipStream_t stream, int W, int H, int P)
{
    for (auto &f : t_frames) if (f.device == device && f.stream == stream && f.W == W && f.H == H && f.P == P) return &f;
    if (t_frames.size() >= 64) t_frames.erase(t_frames.begin());
    t_frames.push_back({device, stream, W, H, P, 0u, 0u});
    return &t_frames.back();
}
extern "C" int64_t gms_last_launched_units(void) { return t_last_launched_units; }
static thread_local int32_t t_last_used_micro = 0;
extern "C" int32_t gms_last_used_micro(void) { return t_last_used_micro; }
extern "C" size_t gms_binning_bytes(int64_t n, int32_t w, int32_t h)
{
    const size_t T = (size_t)((w + TILE - 1) / TILE) * (size_t)((h + TILE - 1) / TILE);
    return BinningState::bytes((size_t)(n > 0 ? n : 0), T, seg_len_min(), micro_mode());
}

extern "C" int64_t gms_rasterize_forward(const GmsRasterForwardArgs *A, void *stream_)
{
    gms::TraceRange trace_range("gms_rasterize_forward");
    hipStream_t stream = (hipStream_t)stream_;
    g_err[0] = 0;
    if (!A || A->P < 0 || A->width <= 0 || A->height <= 0 || !A->out_color || !A->out_invdepth || !A->background) {
        set_error("gms_rasterize_forward: invalid sizes or null output/background pointer");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    const int P = A->P, W = A->width, H = A->height;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
    const GmsMeshArgs *mesh = A->mesh;          // forward-only frame straight from a mesh (gmsplat.h): K0 runs inside the preprocess thread
    if (P > 0 && mesh) {
        // the thread of a Gaussian reads faces / vertices / _alpha through the same tables as the K0 launch: same checks
        const int32_t mrc = check_mesh_args(mesh, false);
        if (mrc != GMS_OK) return mrc;
        const int outs = (A->mesh_out_xyz != nullptr) + (A->mesh_out_scaling_act != nullptr) + (A->mesh_out_rotation_unit != nullptr) + (A->mesh_out_opacity_act != nullptr);
        if (outs != 0 && (outs != 4 || (((uintptr_t)A->mesh_out_rotation_unit) & 15u))) {
            set_error("gms_rasterize_forward: mesh_out_* must be all NULL (forward-only frame) or all set (rotation_unit 16-byte aligned)");
            return GMS_ERR_INVALID_ARGUMENT;
        }
        if (mesh->P != (int64_t)P || !mesh->vertices || !mesh->faces || !mesh->_alpha || !mesh->_scale || !mesh->_opacity ||
            (mesh->splats_per_face <= 0 && !mesh->splat_face) || !A->shs || !A->shs_rest || A->M != 16 || A->D < 0 || A->D > 3 || A->colors_precomp ||
            A->cov3D_precomp || (((uintptr_t)A->shs) & 15u) || (((uintptr_t)A->shs_rest) & 15u)) {
            set_error("gms_rasterize_forward: the fused mesh input needs a complete GmsMeshArgs (P equal, _opacity set), split degree-3 SH "
                      "storage (shs + shs_rest, M = 16, active degree 0 .. 3, 16-byte aligned) and no precomputed colours / covariances");
            return GMS_ERR_INVALID_ARGUMENT;
        }
        if (!A->viewmatrix || !A->projmatrix || !A->campos || !A->radii || !A->geom_alloc || !A->binning_alloc || !A->image_alloc) {
            set_error("gms_rasterize_forward: null input pointer or callback");
            return GMS_ERR_INVALID_ARGUMENT;
        }
    } else if (P > 0) {
        if ((A->shs == nullptr) == (A->colors_precomp == nullptr)) {
            set_error("provide exactly one of shs / colors_precomp");
            return GMS_ERR_INVALID_ARGUMENT;
        }
        const bool sr = A->scales && A->rotations;
        if (sr == (A->cov3D_precomp != nullptr) || (!sr && (A->scales || A->rotations))) {
            set_error("provide exactly one of (scales, rotations) / cov3D_precomp");
            return GMS_ERR_INVALID_ARGUMENT;
        }
        if (!A->means3D || !A->opacities || !A->viewmatrix || !A->projmatrix || !A->campos || !A->radii ||
            !A->geom_alloc || !A->binning_alloc || !A->image_alloc) {
            set_error("gms_rasterize_forward: null input pointer or callback");
            return GMS_ERR_INVALID_ARGUMENT;
        }
        if (A->shs && (A->D < 0 || A->D > 3 || (A->D + 1) * (A->D + 1) > A->M)) {
            set_error("SH degree %d needs %d coefficients but M = %d (degrees 0..3 supported)", A->D,
                      (A->D + 1) * (A->D + 1), A->M);
            return GMS_ERR_INVALID_ARGUMENT;
        }
    }
    const size_t HW = (size_t)W * H;
    if (P == 0) {   // nothing to draw: background image, no scratch
        fill_background_kernel<<<(unsigned)((HW + 255) / 256), 256, 0, stream>>>(W, H, A->background, A->out_color,
                                                                                  A->out_invdepth);
        GMS_KERNEL_CHECK(A->debug, stream, "fill_background");
        return 0;
    }

    void *geom_mem = A->geom_alloc(A->geom_ctx, GeomState::bytes((size_t)P));
    void *img_mem = A->image_alloc(A->image_ctx, ImageState::bytes((size_t)W, (size_t)H));
    if (!geom_mem || !img_mem) { set_error("scratch allocation callback returned NULL"); return GMS_ERR_ALLOC; }
    GeomState geom = GeomState::carve(geom_mem, (size_t)P);
    ImageState img = ImageState::carve(img_mem, (size_t)W, (size_t)H);

    // Per-tile instance counters: a library-owned buffer per (host thread, DEVICE, stream) that is all zero between
    // frames -- tile_scan clears each counter after reading it -- so a frame starts without a memset.  `dirty` covers a
    // frame that failed between the preprocess launch and the scan launch.  (torch's default stream has the raw
    // handle 0 on every device, so the device is part of the key.)
    int device = 0;
    GMS_HIP_CHECK(hipGetDevice(&device));
    struct Counters { int device; hipStream_t stream; uint32_t *buf; size_t cap; bool dirty; };
    static thread_local std::vector<Counters> t_counters;
    Counters *ctr = nullptr;
    for (auto &c : t_counters) if (c.device == device && c.stream == stream) ctr = &c;
    if (!ctr) { t_counters.push_back({device, stream, nullptr, 0, true}); ctr = &t_counters.back(); }
    if (ctr->cap < (size_t)T) {
        if (ctr->buf) (void)hipFree(ctr->buf);       // `device` is current: the buffer was allocated under it
        ctr->buf = nullptr; ctr->cap = 0;
        GMS_HIP_CHECK(hipMalloc((void **)&ctr->buf, (size_t)T * 4));
        ctr->cap = (size_t)T; ctr->dirty = true;
    }
    if (ctr->dirty) GMS_HIP_CHECK(hipMemsetAsync(ctr->buf, 0, ctr->cap * 4, stream));
    ctr->dirty = true;                              // until this frame's scan has been enqueued
    img.tile_count = ctr->buf;
    // Launch-size hints learnt from earlier frames, per (device, stream, W, H, P): two scenes of different size
    // interleaved on one thread (or one scene on two devices) do not disturb each other's hints
    FrameState *fs = frame_state(device, stream, W, H, P);
    {
        std::lock_guard<std::mutex> lk(g_mail_mu);
        for (size_t k = 0; k < g_mail.size(); k++) {
            const HintMail &m = g_mail[k];
            if (m.device == device && m.stream == stream && m.W == W && m.H == H && m.P == P) {
                fs->deepest_seen = m.deepest;
                fs->units_hint = max(m.units, (uint32_t)(0.97 * fs->units_hint));
                g_mail.erase(g_mail.begin() + (long)k);
                break;
            }
        }
    }
    uint32_t &deepest_seen = fs->deepest_seen;      // deepest tile of the previous frame of this shape
    uint32_t &units_hint = fs->units_hint;          // slowly decaying maximum of its work-unit counts (views differ)

    PreArgs pa{};          // (zero-initialised: `mesh` is only filled for the K0 instantiation)
    pa.P = P; pa.D = A->D; pa.M = A->M; pa.W = W; pa.H = H; pa.gx = gx; pa.gy = gy;
    pa.means3D = A->means3D; pa.shs = A->shs; pa.shs_rest = A->shs_rest; pa.colors = A->colors_precomp; pa.opac = A->opacities;
    pa.scales = A->scales; pa.rots = A->rotations; pa.cov3Dp = A->cov3D_precomp; pa.view = A->viewmatrix;
    pa.proj = A->projmatrix; pa.campos = A->campos; pa.mod = A->scale_modifier; pa.tanx = A->tan_fovx;
    pa.tany = A->tan_fovy; pa.aa = A->antialiasing; pa.radii = A->radii; pa.visible = A->visible; pa.geom = geom; pa.tile_count = img.tile_count;
    if (mesh) {
        pa.mesh = *mesh; pa.means3D = nullptr; pa.opac = nullptr; pa.scales = nullptr; pa.rots = nullptr; pa.cov3Dp = nullptr;
        pa.k0_xyz = A->mesh_out_xyz; pa.k0_scale = A->mesh_out_scaling_act; pa.k0_rot = A->mesh_out_rotation_unit; pa.k0_opac = A->mesh_out_opacity_act;
    }
    // Inline tile scan (see emit_instances_kernel): on the capacity-hint path, when the segment length is known up front and the
    // tile table fits the emit blocks' LDS.  GMS_INLINE_SCAN=0 keeps the separate tile_scan launch.
    static int inline_env = -1;
    if (inline_env < 0) { const char *e = getenv("GMS_INLINE_SCAN"); inline_env = e ? (atoi(e) != 0) : 1; }
    const uint32_t frame_L_pre = seg_len_for_frame((uint64_t)A->binning_capacity_hint, T);
    // (every emit block repeats the scan and holds the offsets in LDS, 29 KB per block instead of 12: it pays while the emit blocks
    // are one resident round -- headline scene: 2 354 -> 2 434 it/s -- and loses beyond, config-5 size: emit 117 -> 141 us)
    const bool inline_scan = inline_env && A->binning_capacity_hint > 0 && frame_L_pre != 0 && T <= INLINE_SCAN_MAX_T &&
                             (P + BLOCK - 1) / BLOCK <= 1536;
    pa.zero_cursor = inline_scan ? img.tile_cursor : nullptr; pa.T = T;
    pa.zero_cmax = micro_mode() ? img.tile_cmax : nullptr;
    const unsigned pblocks = (unsigned)((P + BLOCK - 1) / BLOCK);
    const bool split = A->shs_rest != nullptr;
    const bool sh_fast = A->shs && A->M == 16 && (((uintptr_t)A->shs) & 15u) == 0 && (((uintptr_t)A->shs_rest) & 15u) == 0 &&
                         (mesh || A->cov3D_precomp || (((uintptr_t)A->rotations) & 15u) == 0);
    if (split && !sh_fast) {
        set_error("split SH storage (shs_rest) needs M == 16 and 16-byte aligned pointers");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    // Two-stream forward (GMS_SH_STREAM=1 enables; measured on the headline scene: no gain, 2 158 against 2 173 it/s, so off by default): the SH -> RGB half of the preprocess runs on a library-owned second stream
    // beside tile scan / emit / sort / filter, which leave most of the chip idle and need only the geometry half; the
    // compositing waits for it (fork / join with two events).  The 57 MB of coefficient reads leave the critical path.
    static int sh_stream_env = -1;
    if (sh_stream_env < 0) { const char *e = getenv("GMS_SH_STREAM"); sh_stream_env = e ? (atoi(e) != 0) : 0; }
    struct Aux { int device; hipStream_t s; hipEvent_t fork, join; };
    static thread_local std::vector<Aux> t_aux;
    Aux *aux = nullptr;
    const bool two_stream = sh_stream_env && sh_fast && A->D > 0 && !mesh;
    if (two_stream) {
        for (auto &x : t_aux) if (x.device == device) aux = &x;
        if (!aux) {
            Aux x{device, nullptr, nullptr, nullptr};
            GMS_HIP_CHECK(hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking));
            GMS_HIP_CHECK(hipEventCreateWithFlags(&x.fork, hipEventDisableTiming));
            GMS_HIP_CHECK(hipEventCreateWithFlags(&x.join, hipEventDisableTiming));
            t_aux.push_back(x); aux = &t_aux.back();
        }
    }
    static int pre_dma = -1;          // GMS_PRE_DMA=0: the register-staged SH rows instead of LDS-DMA (split degree-3 storage)
    if (pre_dma < 0) { const char *e = getenv("GMS_PRE_DMA"); pre_dma = e ? (atoi(e) != 0) : 1; }
#define GMS_PRE_M(DEG, SP, MODE, STR) GMS_LAUNCH(GMS_K_PREPROCESS_FWD, STR, (preprocess_fwd_kernel<DEG, SP, MODE><<<pblocks, BLOCK, 0, STR>>>(pa)))
#define GMS_PRE(DEG, SP)                                                              \
    do {                                                                              \
        if (aux) {                                                                    \
            GMS_PRE_M(DEG, SP, 1, stream);                                            \
            GMS_HIP_CHECK(hipEventRecord(aux->fork, stream));                         \
            GMS_HIP_CHECK(hipStreamWaitEvent(aux->s, aux->fork, 0));                  \
            GMS_PRE_M(DEG, SP, 2, aux->s);                                            \
            GMS_HIP_CHECK(hipEventRecord(aux->join, aux->s));                         \
        } else {                                                                      \
            GMS_PRE_M(DEG, SP, 0, stream);                                            \
        }                                                                             \
    } while (0)
    if (mesh) {          // (validated above: split degree-3 storage; the active degree picks the instantiation)
        switch (A->D) {
        case 0: GMS_LAUNCH(GMS_K_PREPROCESS_FWD, stream, (preprocess_fwd_dma_kernel<true, 0><<<pblocks, BLOCK, 0, stream>>>(pa))); break;
        case 1: GMS_LAUNCH(GMS_K_PREPROCESS_FWD, stream, (preprocess_fwd_dma_kernel<true, 1><<<pblocks, BLOCK, 0, stream>>>(pa))); break;
        case 2: GMS_LAUNCH(GMS_K_PREPROCESS_FWD, stream, (preprocess_fwd_dma_kernel<true, 2><<<pblocks, BLOCK, 0, stream>>>(pa))); break;
        default: GMS_LAUNCH(GMS_K_PREPROCESS_FWD, stream, (preprocess_fwd_dma_kernel<true, 3><<<pblocks, BLOCK, 0, stream>>>(pa))); break;
        }
    } else
    switch ((sh_fast ? A->D : -1) * 2 + (split ? 1 : 0)) {
    case 0: GMS_PRE_M(0, false, 0, stream); break;
    case 1: GMS_PRE_M(0, true, 0, stream); break;
    case 2: GMS_PRE(1, false); break;
    case 3: GMS_PRE(1, true); break;
    case 4: GMS_PRE(2, false); break;
    case 5: GMS_PRE(2, true); break;
    case 6: GMS_PRE(3, false); break;
    case 7:
        if (pre_dma && !aux) GMS_LAUNCH(GMS_K_PREPROCESS_FWD, stream, (preprocess_fwd_dma_kernel<false><<<pblocks, BLOCK, 0, stream>>>(pa)));
        else GMS_PRE(3, true);
        break;
    default: GMS_PRE_M(-1, false, 0, stream); break;
    }
#undef GMS_PRE
#undef GMS_PRE_M
    GMS_KERNEL_CHECK(A->debug, stream, "preprocess_fwd");
    const uint32_t L = seg_len_min();         // sizes and carving; the frame's own L is chosen by the scan (scan_out[3])
    // deferred read-back (gmsplat.h, count_ticket_out): this frame's counts go to the next slot of the thread's ring and nobody waits here
    const bool defer = A->count_ticket_out != nullptr && A->binning_capacity_hint > 0 && !A->no_host_wait;
    static thread_local uint32_t defer_counter = 0;
    const int slot_index = defer ? 1 + (int)(defer_counter++ % (uint32_t)DEFER_SLOTS) : 0;
    int32_t *slot = pinned_slot(slot_index);
    if (!slot) { set_error("hipHostMalloc for the read-back slot failed"); return GMS_ERR_HIP; }
    // The launches-only form never reads the slot: its launches publish nothing, so a captured graph does not hold this thread's
    // slot and a stale sequence number (a replay would otherwise overwrite a count an eager call on another stream has just received).
    int32_t *const pub_slot = A->no_host_wait ? nullptr : slot;
    // merge-path passes for tiles deeper than SORT_BIG_CHUNK keys: as many as the deepest tile of the previous frame
    // (+25 %) needs; a deeper tile than that falls back to the one-block sort and raises the count for the next frame
    int sort_np = 0;
    if (deepest_seen > (uint32_t)SORT_BIG_CHUNK) sort_np = sort_passes((uint64_t)deepest_seen + deepest_seen / 4);
    static thread_local int32_t seq_counter = 0;
    const int32_t seq = (seq_counter = seq_counter == 0x7fffffff ? 1 : seq_counter + 1);
    const uint32_t frame_L = seg_len_for_frame((uint64_t)A->binning_capacity_hint, T);      // 0: the scan chooses from the depth
    if (!inline_scan) {
        GMS_LAUNCH(GMS_K_TILE_SCAN, stream, tile_scan_kernel<<<1, SCAN_THREADS, 0, stream>>>(img.tile_count, img.tile_offset, img.tile_cursor,
                                                                                       img.unit_first, img.mseg_first, img.class_first, T, frame_L,
                                                                                       pub_slot, seq, sort_np, img.scan_out));
        GMS_KERNEL_CHECK(A->debug, stream, "tile_scan");
        ctr->dirty = false;
    }
    bool first_tail = true;          // the first enqueue_tail of an inline-scan frame scans inside its emit launch; a re-run reads the tables it left

    BlendFwdOut bo;
    bo.rec = geom.rec; bo.bg = A->background; bo.final_T = img.final_T; bo.n_contrib = img.n_contrib;
    bo.out_color = A->out_color; bo.out_invdepth = A->out_invdepth;

    // Blend launches are sized from the previous frame's unit count (+25 %) instead of the table's capacity (thousands of
    // blocks that only find out they have nothing to do); a frame with more units than launched is re-run like a
    // capacity overflow.
    uint32_t units_seen = 0;                          // this frame's count once known
    uint32_t launched_units = 0;
    auto enqueue_tail = [&](void *bin_mem, uint64_t capacity, bool exact_fit) -> int32_t {
        BinningState bin = BinningState::carve(bin_mem, (size_t)capacity, (size_t)T, L);
        const uint32_t mu = (uint32_t)BinningState::n_units((size_t)capacity, (size_t)T, L);
        uint32_t mu_launch = mu;
        if (!exact_fit && units_hint > 0) mu_launch = min(mu, units_hint + units_hint / 4u + 64u);
        launched_units = blend_grid_units(mu_launch);
        FillUnitsArgs fu;
        fu.class_first = img.class_first; fu.offset = img.tile_offset; fu.mseg_first = img.mseg_first;
        fu.unit_tile = bin.unit_tile; fu.deep_tab = bin.deep_tab; fu.multi_tab = bin.multi_tab;
        fu.max_multi = (uint32_t)BinningState::n_multi((size_t)capacity); fu.T = T; fu.sort_np = sort_np; fu.scan_out = img.scan_out; fu.max_units = mu;
        fu.max_deep = (uint32_t)BinningState::n_deep((size_t)capacity, (size_t)T);
        const unsigned fblocks = (unsigned)((T + BLOCK - 1) / BLOCK);
        const bool inl = inline_scan && first_tail;
        first_tail = false;
        InlineScanArgs isa{img.tile_count, img.tile_offset, img.unit_first, img.mseg_first, img.class_first, img.scan_out, frame_L};
        if (inl)
            GMS_LAUNCH(GMS_K_EMIT, stream, emit_instances_kernel<true><<<pblocks + fblocks, BLOCK, 0, stream>>>(P, gx, gy, A->radii, geom, img.tile_offset,
                                                                                                                 img.tile_cursor, bin.keys, capacity, pblocks, fu,
                                                                                                                 pub_slot, seq, img.scan_out, isa));
        else
            GMS_LAUNCH(GMS_K_EMIT, stream, emit_instances_kernel<false><<<pblocks + fblocks, BLOCK, 0, stream>>>(P, gx, gy, A->radii, geom, img.tile_offset,
                                                                                                                  img.tile_cursor, bin.keys, capacity, pblocks, fu,
                                                                                                                  nullptr, seq, img.scan_out, isa));
        GMS_KERNEL_CHECK(A->debug, stream, "emit_instances");
        uint64_t *sort_tmp = reinterpret_cast<uint64_t *>(bin.seg_state);      // free until compositing
        if ((uint64_t)BinningState::n_slots((size_t)capacity, L) * 7u * TILE_PIX * 4u < capacity * 8u) sort_np = 0;   // (very long segments)
        const uint32_t max_deep = (uint32_t)BinningState::n_deep((size_t)capacity, (size_t)T);
        const uint32_t max_multi = (uint32_t)BinningState::n_multi((size_t)capacity);
        const uint32_t *run_total = img.class_first + NCLASS * ((size_t)T + 1) + T;
        const uint32_t *multi_total = img.class_first + (NCLASS + 1) * ((size_t)T + 1) + T;
        static int presort_lds = -1;        // GMS_PRESORT=lds: the LDS bitonic presort instead of the register-resident one
        if (presort_lds < 0) { const char *e = getenv("GMS_PRESORT"); presort_lds = (e && e[0] == 'l') ? 1 : 0; }
        if (presort_lds)
            GMS_LAUNCH(GMS_K_TILE_SORT, stream, tile_presort_kernel<<<max_deep, 256, 0, stream>>>(img.tile_offset, run_total, bin.deep_tab, max_deep, bin.keys,
                                                                                                 sort_tmp, capacity, sort_np, inl ? img.tile_count : nullptr, T));
        else
            GMS_LAUNCH(GMS_K_TILE_SORT, stream, tile_presort_reg_kernel<<<max_deep, 256, 0, stream>>>(img.tile_offset, run_total, bin.deep_tab, max_deep, bin.keys,
                                                                                                     sort_tmp, capacity, sort_np, inl ? img.tile_count : nullptr, T));
        if (inl) ctr->dirty = false;          // the counters are clean again once this launch has run
        for (int pass = 0; pass < sort_np; pass++)
            GMS_LAUNCH(GMS_K_TILE_SORT, stream, tile_mergepath_kernel<<<max_deep, 256, 0, stream>>>(img.tile_offset, run_total, bin.deep_tab, max_deep, bin.keys,
                                                                                                   sort_tmp, capacity, sort_np, pass));
        // (a merge by RANK -- K interleaved binary searches per thread instead of co-rank + serial merge -- was measured in round 5 and is
        //  slower, tile_sort 32.5 -> 58 us: profiles/r05c_ab_merge.txt; the code is in the history of round 5)
        GMS_LAUNCH(GMS_K_TILE_SORT, stream, tile_merge_kernel<<<max_multi, 1024, 0, stream>>>(img.tile_offset, multi_total, bin.multi_tab, max_multi, bin.keys,
                                                                                             capacity, sort_np));
        GMS_KERNEL_CHECK(A->debug, stream, "tile_sort");
        BlendGrid g;
        g.W = W; g.H = H; g.gx = gx; g.gy = gy; g.T = T; g.scan_out = img.scan_out; g.tile_offset = img.tile_offset;
        g.unit_first = img.unit_first; g.mseg_first = img.mseg_first; g.unit_tile = bin.unit_tile; g.keys = bin.keys;
        g.seg_state = bin.seg_state; g.capacity = capacity; g.max_units = mu; g.dbg = 0; g.unit_run = unit_run(); g.dbg_buf = nullptr; g.tile_dead = img.tile_dead; g.tile_cmax = img.tile_cmax;
        g.mmask = bin.mmask;
        if (aux) GMS_HIP_CHECK(hipStreamWaitEvent(stream, aux->join, 0));      // the colours (second stream) before the compositing
        t_last_used_micro = use_micro(capacity, T) ? 1 : 0;
        if (use_micro(capacity, T)) {
            // the micro-tile kernels index a unit's entries with one byte: never launch them on a frame whose L they cannot hold
            if (frame_L == 0 || frame_L > SEG_LEN_MICRO) {
                set_error("internal: micro-tile compositing chosen for a frame with segment length %u", frame_L);
                return GMS_ERR_INVALID_ARGUMENT;
            }
            return launch_micro_forward(g, bo, mu_launch, A->debug != 0, stream);
        }
        return launch_blend_forward(g, bo, mu_launch, A->debug != 0, stream);
    };

    int64_t N;
    if (A->no_host_wait) {
        // capturable form (gmsplat.h): launches only.  The counts stay on the device (img.scan_out); overflow is the caller's to detect.
        if (A->binning_capacity_hint <= 0) { set_error("no_host_wait needs a binning capacity hint (render the shape once without it first)"); return GMS_ERR_INVALID_ARGUMENT; }
        const uint64_t cap = (uint64_t)A->binning_capacity_hint;
        void *bin_mem = A->binning_alloc(A->binning_ctx, BinningState::bytes((size_t)cap, (size_t)T, L, micro_mode()));
        if (!bin_mem) { set_error("binning allocation callback returned NULL"); return GMS_ERR_ALLOC; }
        const int32_t rc = enqueue_tail(bin_mem, cap, false);
        if (rc != GMS_OK) return rc;
        t_last_launched_units = (int64_t)launched_units;
        if (A->num_units_out) *A->num_units_out = 0;
        return (int64_t)cap;
    }
    if (defer) {
        // launches + the device's store of the counts into the ring slot; the caller redeems the ticket before its backward
        const uint64_t cap = (uint64_t)A->binning_capacity_hint;
        void *bin_mem = A->binning_alloc(A->binning_ctx, BinningState::bytes((size_t)cap, (size_t)T, L, micro_mode()));
        if (!bin_mem) { set_error("binning allocation callback returned NULL"); return GMS_ERR_ALLOC; }
        const int32_t rc = enqueue_tail(bin_mem, cap, false);
        if (rc != GMS_OK) return rc;
        t_last_launched_units = (int64_t)launched_units;
        A->count_ticket_out[0] = (int64_t)(uintptr_t)slot;          // (pinned, process-wide: any thread may poll it)
        A->count_ticket_out[1] = (int64_t)seq;
        if (A->num_units_out) *A->num_units_out = 0;
        return (int64_t)cap;
    }
    if (A->binning_capacity_hint > 0) {
        // optimistic path: enqueue the whole tail before looking at N (no pipeline bubble)
        const uint64_t cap = (uint64_t)A->binning_capacity_hint;
        void *bin_mem = A->binning_alloc(A->binning_ctx, BinningState::bytes((size_t)cap, (size_t)T, L, micro_mode()));
        if (!bin_mem) { set_error("binning allocation callback returned NULL"); return GMS_ERR_ALLOC; }
        int32_t rc = enqueue_tail(bin_mem, cap, false);
        if (rc != GMS_OK) return rc;
        t_last_launched_units = (int64_t)launched_units;
        rc = wait_for_count(slot, seq, stream, &N);
        if (rc != GMS_OK) return rc;
        deepest_seen = deepest_tile(slot);
        t_last_deepest = deepest_seen;
        units_seen = unit_count(slot);
        units_hint = max(units_seen, (uint32_t)(0.97 * units_hint));
        if ((uint64_t)N <= cap && units_seen > launched_units) {       // enough memory, too few blocks: same buffers, full-size launches
            GMS_HIP_CHECK(hipMemsetAsync(img.tile_cursor, 0, (size_t)T * 4, stream));
            rc = enqueue_tail(bin_mem, cap, true);
            if (rc != GMS_OK) return rc;
        }
        if ((uint64_t)N > cap) {                         // rare: re-run the tail at the right size
            bin_mem = A->binning_alloc(A->binning_ctx, BinningState::bytes((size_t)N, (size_t)T, L, micro_mode()));
            if (!bin_mem) { set_error("binning allocation callback returned NULL"); return GMS_ERR_ALLOC; }
            GMS_HIP_CHECK(hipMemsetAsync(img.tile_cursor, 0, (size_t)T * 4, stream));
            rc = enqueue_tail(bin_mem, (uint64_t)N, true);
            if (rc != GMS_OK) return rc;
        }
    } else {
        int32_t rc0 = wait_for_count(slot, seq, stream, &N);
        if (rc0 != GMS_OK) return rc0;
        deepest_seen = deepest_tile(slot);
        t_last_deepest = deepest_seen;
        units_seen = unit_count(slot);
        units_hint = max(units_seen, (uint32_t)(0.97 * units_hint));
        const uint64_t cap = (uint64_t)(N > 0 ? N : 1);
        void *bin_mem = A->binning_alloc(A->binning_ctx, BinningState::bytes((size_t)cap, (size_t)T, L, micro_mode()));
        if (!bin_mem) { set_error("binning allocation callback returned NULL"); return GMS_ERR_ALLOC; }
        int32_t rc = enqueue_tail(bin_mem, cap, true);         // N is exact here, so the table-sized launch is tight
        if (rc != GMS_OK) return rc;
    }
    if (A->num_units_out) *A->num_units_out = (int64_t)units_seen;
    return N;
}

extern "C" int64_t gms_rasterize_forward_counts(const int64_t *ticket, int32_t width, int32_t height, int32_t P, int64_t *num_units_out,
                                                int64_t *deepest_tile_out, void *stream_)
{
    gms::TraceRange trace_range("gms_rasterize_forward_counts");
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    if (!ticket || ticket[0] == 0 || ticket[1] <= 0 || ticket[1] > 0x7fffffffll) {
        set_error("gms_rasterize_forward_counts: not a ticket of a deferred forward");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    int32_t *slot = reinterpret_cast<int32_t *>((uintptr_t)ticket[0]);
    const int32_t seq = (int32_t)ticket[1];
    // a slot that already carries a LATER call's tag was reused: more than DEFER_SLOTS tickets of the forward thread were outstanding
    const unsigned long long tag0 = __atomic_load_n(reinterpret_cast<const unsigned long long *>(slot), __ATOMIC_ACQUIRE) >> 32;
    if (tag0 != 0ull && (int32_t)(uint32_t)tag0 - seq > 0) {
        set_error("gms_rasterize_forward_counts: ticket expired (more than %d deferred forwards outstanding on the forward's host thread)", DEFER_SLOTS);
        return GMS_ERR_INVALID_ARGUMENT;
    }
    int64_t N = 0;
    const int32_t rc = wait_for_count(slot, seq, stream, &N);
    if (rc != GMS_OK) return rc;
    const uint32_t units = unit_count(slot), deepest = deepest_tile(slot);
    int device = 0;
    GMS_HIP_CHECK(hipGetDevice(&device));
    {
        std::lock_guard<std::mutex> lk(g_mail_mu);
        bool found = false;
        for (auto &m : g_mail)
            if (m.device == device && m.stream == stream && m.W == width && m.H == height && m.P == P) { m.deepest = deepest; m.units = units; found = true; }
        if (!found) {
            if (g_mail.size() >= 64) g_mail.erase(g_mail.begin());
            g_mail.push_back({device, stream, width, height, P, deepest, units});
        }
    }
    t_last_deepest = deepest;
    if (num_units_out) *num_units_out = (int64_t)units;
    if (deepest_tile_out) *deepest_tile_out = (int64_t)deepest;
    return N;
}

extern "C" int64_t gms_last_deepest_tile(void) { return (int64_t)t_last_deepest; }

extern "C" void gms_wait_stats(double *total_ms, int64_t *calls, int32_t reset)
{
    if (total_ms) *total_ms = (double)g_wait_ns.load() * 1e-6;
    if (calls) *calls = g_wait_calls.load();
    if (reset) { g_wait_ns = 0; g_wait_calls = 0; }
}

extern "C" int32_t gms_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                                    uint8_t *present, void *stream_)
{
    (void)projmatrix;
    hipStream_t stream = (hipStream_t)stream_;
    g_err[0] = 0;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) {
        set_error("gms_mark_visible: invalid argument");
        return GMS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return GMS_OK;
    mark_visible_kernel<<<(unsigned)((P + 255) / 256), 256, 0, stream>>>(P, means3D, viewmatrix, present);
    GMS_KERNEL_CHECK(0, stream, "mark_visible");
    return GMS_OK;
}
