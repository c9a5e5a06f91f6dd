// gms_common.h -- shared device/host helpers for the gfx950 rasterizer kernels.
// Written for CDNA4 only: wave = 64 lanes, 4 waves per 16x16 pixel tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gmsplat.h"

namespace gms {

constexpr int TILE = 16;            // pixel tile edge (binning granularity)
constexpr int TILE_PIX = TILE * TILE;
constexpr int WAVE = 64;
constexpr int BLOCK = 256;          // threads per block everywhere: 4 waves
constexpr float NEAR_Z = 0.2f;
constexpr float DILATE = 0.3f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_MIN = 0.0001f;

// SH constants (utils/sh_utils.py:26-43 of the reference)
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// Per-Gaussian splat record consumed by the blend kernels: 48 bytes, gathered by id.
//   q0 = (pix.x, pix.y, conic.A, conic.B)
//   q1 = (conic.C, opacity', r, g)
//   q2 = (b, 1/depth, ext.x, ext.y)   ext = half extent of the alpha >= 1/255 ellipse's bounding box
struct __attribute__((aligned(16))) SplatRec {
    float4 q0, q1, q2;
};

// ---- scratch buffer layouts (carved from the caller's three buffers, 256-B aligned chunks)
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct GeomState {
    SplatRec *rec;      // [P]
    float *depth;       // [P]
    uint8_t *clamped;   // [P] bit c set => colour channel c was clamped at 0
    ushort4 *rect;      // [P] tile rectangle [minx, maxx) x [miny, maxy) as (minx, miny, maxx, maxy); all zero for a culled Gaussian.
                        //     What emit_instances reads (with `depth`: 12 bytes per Gaussian) instead of the pixel centre out of the
                        //     48-byte record, whose every cache line it had to fetch for 8 bytes (round 6: 25 -> 13 MB of HBM traffic)
    static __host__ __device__ size_t bytes(size_t P)
    {
        return align_up(P * sizeof(SplatRec), 256) + align_up(P * 4, 256) + align_up(P, 256) + align_up(P * sizeof(ushort4), 256);
    }
    static __host__ __device__ GeomState carve(void *base, size_t P)
    {
        GeomState g;
        char *p = (char *)base;
        g.rec = (SplatRec *)p; p += align_up(P * sizeof(SplatRec), 256);
        g.depth = (float *)p;  p += align_up(P * 4, 256);
        g.clamped = (uint8_t *)p; p += align_up(P, 256);
        g.rect = (ushort4 *)p;
        return g;
    }
};

struct ImageState {
    float *final_T;         // [H*W]
    uint32_t *n_contrib;    // [H*W]
    uint32_t *tile_count;   // [T]     instances per tile (atomically counted in preprocess); points at a library-owned,
                            //          always-zero-between-frames buffer (the slot carved here is unused)
    uint32_t *tile_cursor;  // [T]     emit cursors
    uint32_t *tile_offset;  // [T+1]   exclusive scan of tile_count; tile_offset[T] = N
    uint32_t *unit_first;   // [T+1]   exclusive scan of segments per tile; unit_first[T] = #units
    uint32_t *mseg_first;   // [T+1]   exclusive scan of segments of multi-segment tiles only
    uint32_t *class_first;  // [8][T+1] exclusive scans per dispatch class (full, 4 partial size classes, empty), of the
                            //          1024-key sort runs of every tile (row 6) and of the tiles with more than one run (row 7)
    uint32_t *tile_dead;    // [T]     all pixels of the tile finished within the first few segments
    uint32_t *tile_cmax;    // [T]     bits of the largest |colour component| among the tile's splats (cleared by preprocess_fwd, raised by
                            //          the forward micro-tile launches that stage the tile's units): bounds the colour behind any splat in
                            //          the backward's fixed-point gradient table (blend_micro.hip)
    uint32_t *scan_out;     // [4]     {N, deepest tile, work units} of this frame, for the launch that publishes them to the host
    static __host__ __device__ size_t bytes(size_t W, size_t H)
    {
        size_t T = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
        return 2 * align_up(W * H * 4, 256) + 2 * align_up(T * 4, 256) + 3 * align_up((T + 1) * 4, 256) +
               align_up(8 * (T + 1) * 4, 256) + 2 * align_up(T * 4, 256) + 256;
    }
    static __host__ __device__ ImageState carve(void *base, size_t W, size_t H)
    {
        size_t T = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
        ImageState s;
        char *p = (char *)base;
        s.final_T = (float *)p;        p += align_up(W * H * 4, 256);
        s.n_contrib = (uint32_t *)p;   p += align_up(W * H * 4, 256);
        s.tile_count = (uint32_t *)p;  p += align_up(T * 4, 256);
        s.tile_cursor = (uint32_t *)p; p += align_up(T * 4, 256);
        s.tile_offset = (uint32_t *)p; p += align_up((T + 1) * 4, 256);
        s.unit_first = (uint32_t *)p;  p += align_up((T + 1) * 4, 256);
        s.mseg_first = (uint32_t *)p;  p += align_up((T + 1) * 4, 256);
        s.class_first = (uint32_t *)p; p += align_up(8 * (T + 1) * 4, 256);
        s.tile_dead = (uint32_t *)p;   p += align_up(T * 4, 256);
        s.tile_cmax = (uint32_t *)p;   p += align_up(T * 4, 256);
        s.scan_out = (uint32_t *)p;
        return s;
    }
};

// Binning buffer: sorted keys, unit table and per-(unit, pixel) segment state.  Sized from the
// instance capacity N, the tile count T and the segment length L.
struct BinningState {
    uint64_t *keys;        // [N]  (depth_bits << 32) | gaussian id ; sorted in place per tile segment
    uint4 *unit_tile;      // [T + N/L + 1][2] unit records {tile, seg, nseg, slot0 | tile_beg, tile_end, -, -}, heaviest first
    float *seg_state;      // [2N/L + 2][7][256]
    uint2 *deep_tab;       // [N/1024 + T + 2] (tile, run): every 1024-key sort run / merge chunk of every tile with >= 2 keys
    uint32_t *multi_tab;   // [N/1024 + 2] tiles with more than one run (they need merging)
    // micro-tile compositing (blend_micro.hip): per instance (position in the sorted list) the 16-bit mask of the tile's 4x4 pixel
    // blocks its {alpha >= 1/255} ellipse touches -- written by the launch that first touches the instance's unit, read by the later
    // ones, so that every launch builds identical per-block lists.  (Until round 5: 16 bytes of list area per instance + counts.)
    uint16_t *mmask;       // [N]
    static __host__ __device__ size_t n_units(size_t N, size_t T, size_t L) { return T + N / L + 1; }
    static __host__ __device__ size_t n_slots(size_t N, size_t L) { return 2 * (N / L) + 2; }
    static __host__ __device__ size_t n_deep(size_t N, size_t T) { return N / 1024 + T + 2; }
    static __host__ __device__ size_t n_multi(size_t N) { return N / 1024 + 2; }
    static __host__ __device__ size_t bytes(size_t N, size_t T, size_t L, bool micro)
    {
        return align_up((N > 0 ? N : 1) * 8, 256) + align_up(n_units(N, T, L) * 32, 256) +
               align_up(n_slots(N, L) * 7 * TILE_PIX * 4, 256) + align_up(n_deep(N, T) * 8, 256) + align_up(n_multi(N) * 4, 256) +
               (micro ? align_up((N > 0 ? N : 1) * 2, 256) : 0);
    }
    static __host__ __device__ BinningState carve(void *base, size_t N, size_t T, size_t L)
    {
        BinningState b;
        char *p = (char *)base;
        b.keys = (uint64_t *)p;      p += align_up((N > 0 ? N : 1) * 8, 256);
        b.unit_tile = (uint4 *)p;    p += align_up(n_units(N, T, L) * 32, 256);
        b.seg_state = (float *)p;    p += align_up(n_slots(N, L) * 7 * TILE_PIX * 4, 256);
        b.deep_tab = (uint2 *)p;     p += align_up(n_deep(N, T) * 8, 256);
        b.multi_tab = (uint32_t *)p; p += align_up(n_multi(N) * 4, 256);
        b.mmask = (uint16_t *)p;      // (present only in micro mode)
        return b;
    }
};

// ---- wave64 reductions with DPP (row ops + row broadcasts; result valid in lane 63)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v)
{
    int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}

__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);       // row_half_mirror
    v = dpp_add<0x140>(v);       // row_mirror      -> every lane of a 16-row holds the row sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast31 into rows 2,3 -> lane 63 (row 3) holds the wave sum
    return v;
}

// gfx950 lane-swap instructions (the clang builtins mis-model the second result in ROCm 7.2, so
// they are issued as inline asm; `s_nop 1` covers the VALU-write -> permlane-swap-read hazard that
// hipcc does not track for asm operands).  All 64 lanes must be active.
//   v_permlane32_swap a, b: a <- [a.lo32lanes, b.lo32lanes],  b <- [a.hi32lanes, b.hi32lanes]
//   v_permlane16_swap a, b: rows (16 lanes) a=(r0,r1,r2,r3), b=(s0,s1,s2,s3) -> a=(r0,s0,r2,s2), b=(r1,s1,r3,s3)
// five / three independent swaps behind ONE hazard nop (their operands were written by earlier VALU ops;
// consecutive swaps touch different registers)
__device__ __forceinline__ void swap32x5(float &a0, float &a1, float &b0, float &b1, float &c0, float &c1, float &d0,
                                         float &d1, float &e0, float &e1)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %8, %9"
                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1), "+v"(e0), "+v"(e1));
}
__device__ __forceinline__ void swap16x3(float &a0, float &a1, float &b0, float &b1, float &c0, float &c1)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5"
                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1));
}

// Transposing wave reduction of ten values: every stage adds partner lanes AND halves the number
// of live registers (the two halves / rows / half-rows end up holding different values), ~26 VALU
// instead of 80 for ten independent butterflies.  Result layout:
//   y0, lane group (lane>>3) = 0..7 holds the wave total of v0, v4, v2, v6, v1, v5, v3, v7
//   y1, lanes 0..15 hold the total of v8, lanes 32..47 the total of v9
__device__ __forceinline__ void wave_reduce10(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                              float v7, float v8, float v9, float &y0, float &y1)
{
    swap32x5(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9);
    float w0 = v0 + v1, w1 = v2 + v3, w2 = v4 + v5, w3 = v6 + v7, w4 = v8 + v9;
    float z = 0.f;
    swap16x3(w0, w1, w2, w3, w4, z);
    const float x0 = w0 + w1;                    // rows (v0, v2, v1, v3)
    const float x1 = w2 + w3;                    // rows (v4, v6, v5, v7)
    const float x2 = w4 + z;                     // rows (v8, 0, v9, 0)
    const float a = dpp_add<0x128>(x0);          // row_ror:8 : lane l += lane l^8 (within the row)
    const float b = dpp_add<0x128>(x1);
    y0 = (threadIdx.x & 8) ? b : a;
    y1 = dpp_add<0x128>(x2);
    y0 = dpp_add<0x141>(y0); y0 = dpp_add<0xB1>(y0); y0 = dpp_add<0x4E>(y0);   // sum the 8 lanes of each group
    y1 = dpp_add<0x141>(y1); y1 = dpp_add<0xB1>(y1); y1 = dpp_add<0x4E>(y1);
}

// Two independent ten-value reductions with their stages interleaved (same layout as wave_reduce10
// for each): the second chain fills the issue slots the first one leaves while waiting.
__device__ __forceinline__ void wave_reduce10x2(float *a, float *b, float &y0a, float &y1a, float &y0b, float &y1b)
{
    swap32x5(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]);
    swap32x5(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9]);
    float wa0 = a[0] + a[1], wb0 = b[0] + b[1], wa1 = a[2] + a[3], wb1 = b[2] + b[3], wa2 = a[4] + a[5];
    float wb2 = b[4] + b[5], wa3 = a[6] + a[7], wb3 = b[6] + b[7], wa4 = a[8] + a[9], wb4 = b[8] + b[9];
    float za = 0.f, zb = 0.f;
    swap16x3(wa0, wa1, wa2, wa3, wa4, za);
    swap16x3(wb0, wb1, wb2, wb3, wb4, zb);
    const float xa0 = wa0 + wa1, xb0 = wb0 + wb1, xa1 = wa2 + wa3, xb1 = wb2 + wb3, xa2 = wa4 + za, xb2 = wb4 + zb;
    const bool hi8 = (threadIdx.x & 8) != 0;
    const float pa = dpp_add<0x128>(xa0), pb = dpp_add<0x128>(xb0), qa = dpp_add<0x128>(xa1), qb = dpp_add<0x128>(xb1);
    y0a = hi8 ? qa : pa; y0b = hi8 ? qb : pb;
    y1a = dpp_add<0x128>(xa2); y1b = dpp_add<0x128>(xb2);
    y0a = dpp_add<0x141>(y0a); y0b = dpp_add<0x141>(y0b); y1a = dpp_add<0x141>(y1a); y1b = dpp_add<0x141>(y1b);
    y0a = dpp_add<0xB1>(y0a); y0b = dpp_add<0xB1>(y0b); y1a = dpp_add<0xB1>(y1a); y1b = dpp_add<0xB1>(y1b);
    y0a = dpp_add<0x4E>(y0a); y0b = dpp_add<0x4E>(y0b); y1a = dpp_add<0x4E>(y1a); y1b = dpp_add<0x4E>(y1b);
}

// ordering point for LDS data that only ONE wave touches (its own rows / its own sort span): LDS operations of a wave
// execute in order, so no block barrier -- and no waiting for the block's other waves -- is needed
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Tile rectangle of a splat: identical code in preprocess (count) and emit, so both agree.
__device__ __forceinline__ void tile_rect(float px, float py, float radius, int gx, int gy, int &minx, int &miny,
                                          int &maxx, int &maxy)
{
    const float big = 1048576.0f;
    float q0 = (px - radius) / TILE, q1 = (py - radius) / TILE;
    float q2 = (px + radius + (TILE - 1)) / TILE, q3 = (py + radius + (TILE - 1)) / TILE;
    minx = (int)fminf(big, fmaxf(-big, q0));
    miny = (int)fminf(big, fmaxf(-big, q1));
    maxx = (int)fminf(big, fmaxf(-big, q2));
    maxy = (int)fminf(big, fmaxf(-big, q3));
    minx = min(gx, max(0, minx)); maxx = min(gx, max(0, maxx));
    miny = min(gy, max(0, miny)); maxy = min(gy, max(0, maxy));
}

// Wave-aggregated counter increment.  Neighbouring Gaussians (consecutive ids on a mesh face) land
// in the same tile, so per-lane atomics hammer a handful of addresses and serialise in L2.  Here
// the lanes that target the same counter elect a leader, the leader issues ONE atomic for the
// group and every lane derives its own rank from the ballot.  Must be called by all 64 lanes.
template <bool RETURNING>
__device__ __forceinline__ uint32_t wave_aggregated_inc(uint32_t *counters, int key, bool active)
{
    const int lane = (int)(threadIdx.x & 63);
    // group discovery: registers only
    uint32_t rank = 0, cnt = 0;
    int my_leader = lane;
    bool is_leader = false;
    uint64_t todo = __ballot(active);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int k = __builtin_amdgcn_readlane(key, leader);
        const bool mine = active && key == k;
        const uint64_t same = __ballot(mine);
        if (mine) { rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull)); my_leader = leader; }
        if (lane == leader) { is_leader = true; cnt = (uint32_t)__builtin_popcountll(same); }
        todo &= ~same;
    }
    // every group's leader issues its atomic in the same instruction: one memory round trip per call
    uint32_t base = 0;
    if (is_leader) {
        if (RETURNING) base = atomicAdd(&counters[key], cnt);
        else atomicAdd(&counters[key], cnt);
    }
    if (!RETURNING) return 0;
    base = (uint32_t)__shfl((int)base, my_leader);
    return base + rank;
}

// Block-level tile table in LDS (open addressing): the (Gaussian, tile) instances of the 256 consecutive
// Gaussians of a block fall into a few dozen distinct tiles, so counting them in LDS first turns one global
// atomic per wave-step-and-tile into one per block-and-tile.  `tt_insert` returns the slot (or -1 when 16
// probes found only other tiles: the caller then falls back to a direct global atomic).
constexpr int TT_SLOTS = 1024;
__device__ __forceinline__ uint32_t tt_hash(int tile) { return ((uint32_t)tile * 2654435761u) >> 22; }
__device__ __forceinline__ int tt_insert(int *key, int tile)
{
    uint32_t h = tt_hash(tile);
    for (int probe = 0; probe < 16; probe++) {
        const int old = atomicCAS(&key[h], -1, tile);
        if (old == -1 || old == tile) return (int)h;
        h = (h + 1) & (TT_SLOTS - 1);
    }
    return -1;
}
__device__ __forceinline__ int tt_find(const int *key, int tile)
{
    uint32_t h = tt_hash(tile);
    for (int probe = 0; probe < 16; probe++) {
        const int k = key[h];
        if (k == tile) return (int)h;
        if (k == -1) return -1;
        h = (h + 1) & (TT_SLOTS - 1);
    }
    return -1;
}

// Split form of the returning variant: `issue` elects leaders and fires the atomic (result pending in
// `base`), `finish` broadcasts the leader's base.  Lets a caller keep several independent atomics in
// flight and pay the memory round trip once.
struct AggTicket { uint32_t rank; int leader; uint32_t base; };

__device__ __forceinline__ AggTicket wave_aggregated_issue(uint32_t *counters, int key, bool active)
{
    const int lane = (int)(threadIdx.x & 63);
    AggTicket t{0u, lane, 0u};
    uint32_t cnt = 0;
    bool is_leader = false;
    uint64_t todo = __ballot(active);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int k = __builtin_amdgcn_readlane(key, leader);
        const bool mine = active && key == k;
        const uint64_t same = __ballot(mine);
        if (mine) { t.rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull)); t.leader = leader; }
        if (lane == leader) { is_leader = true; cnt = (uint32_t)__builtin_popcountll(same); }
        todo &= ~same;
    }
    if (is_leader) t.base = atomicAdd(&counters[key], cnt);
    return t;
}

__device__ __forceinline__ uint32_t wave_aggregated_finish(const AggTicket &t)
{
    return (uint32_t)__shfl((int)t.base, t.leader) + t.rank;
}

// a*b + c*d + e*f + g in the documented order ((a*b (+) c*d) (+) e*f) + g, each (+) fused.
__device__ __forceinline__ float dot3p(float a, float b, float c, float d, float e, float f, float g)
{
    float t = a * b;
    t = __fmaf_rn(c, d, t);
    t = __fmaf_rn(e, f, t);
    return t + g;
}

// fault injection for the parity criterion's negative controls (gmsplat.h, gms_set_fault); 0 in production
int fault_mode();

// Deterministic-reduction mode (gmsplat.h, gms_set_deterministic / env GAMES_HIP_DETERMINISTIC=1): every floating-point sum
// of the backward passes runs in a fixed order -- no float atomics anywhere -- so two runs give bit-identical gradients.
int det_mode();
// upstream-quirk switch (gmsplat.h, gms_set_upstream_scale_mod_grad / env GMS_UPSTREAM_SCALE_MOD_GRAD=1): dL/dscale without the scale_modifier factor
int upstream_scale_mod_grad();
// library-owned device scratch of the deterministic mode, one growing buffer per (device, stream, slot); nullptr on failure
void *det_scratch(int slot, size_t bytes, hipStream_t stream);

// thread-local error text for gms_last_error()
void set_error(const char *fmt, ...);

// ROCTX range around a C-ABI entry point (env GMS_ROCTX=1; profile.hip)
struct TraceRange { bool on; explicit TraceRange(const char *name); ~TraceRange(); };

// optional per-kernel event timing (profile.hip)
extern bool g_profile_on;
void profile_begin(int kernel_id, hipStream_t stream);
void profile_end(int kernel_id, hipStream_t stream);

}  // namespace gms

#define GMS_HIP_CHECK(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            gms::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GMS_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// launch bracket: GMS_LAUNCH(GMS_K_x, stream, kernel<<<grid, block, 0, stream>>>(args));
#define GMS_LAUNCH(kid, stream, ...)                                  \
    do {                                                              \
        if (gms::g_profile_on) gms::profile_begin(kid, stream);       \
        __VA_ARGS__;                                                  \
        if (gms::g_profile_on) gms::profile_end(kid, stream);         \
    } while (0)

#define GMS_KERNEL_CHECK(dbg, stream, name)                                                    \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e == hipSuccess && (dbg)) _e = hipStreamSynchronize(stream);                      \
        if (_e != hipSuccess) {                                                                \
            gms::set_error("kernel %s failed: %s", name, hipGetErrorString(_e));               \
            return GMS_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)
