// gms_blend.h -- work-unit (tile, segment) bookkeeping shared by the binning and blend stages.
#pragma once
#include "gms_common.h"

namespace gms {

// fields of the per-Gaussian 64-byte gradient record
enum { GRAD_MX = 0, GRAD_MY, GRAD_CA, GRAD_CB, GRAD_CC, GRAD_OP, GRAD_R, GRAD_G, GRAD_B, GRAD_ID, GRAD_STRIDE = 16 };

// per (multi-segment unit, pixel) state, SoA over the 256 pixels of the tile
enum { SEG_TLOC = 0, SEG_C0, SEG_C1, SEG_C2, SEG_D, SEG_TEND, SEG_LAST, SEG_FIELDS };
constexpr size_t SEG_FLOATS = (size_t)SEG_FIELDS * TILE_PIX;     // 7 KiB per unit

struct BlendGrid {
    int W, H, gx, gy, T;
    const uint32_t *scan_out;      // [4] {N, deepest tile, units, L} of this frame: L = entries per segment, chosen by the tile scan
    const uint32_t *tile_offset;   // [T+1]
    const uint32_t *unit_first;    // [T+1] first unit of each tile; unit_first[T] = number of units
    const uint32_t *mseg_first;    // [T+1] first segment-state slot of each multi-segment tile
    const uint4 *unit_tile;        // [units][2] {tile, seg, nseg, slot0 | tile_beg, tile_end, -, -}, heaviest first
    const uint64_t *keys;          // sorted (depth, id) keys
    float *seg_state;              // [slots][SEG_FIELDS][256]
    uint32_t *tile_dead;           // [T] set by the tloc check: every pixel finished within the first segments
    uint64_t capacity;             // instances the binning buffer can hold
    uint32_t max_units;            // entries of unit_tile
    uint32_t dbg;                  // experiment switches (env GMS_DBG; 0 in production)
    uint32_t unit_run;             // consecutive units dealt to one XCD (power of two <= 64)
    unsigned long long *dbg_buf;   // GMS_DBG&16: per block {start, end} wall clock (100 MHz), else NULL
};

struct BlendFwdOut {
    const SplatRec *rec;
    const float *bg;
    float *final_T;
    uint32_t *n_contrib;
    float *out_color;
    float *out_invdepth;
};

struct BlendBwdArgs {
    const SplatRec *rec;
    const float *bg;
    const float *final_T;
    const uint32_t *n_contrib;
    const float *dL_dpix;       // [3,H,W]
    const float *dL_dinvd;      // [H,W] or NULL
    float *accum;               // [P,16] zero-filled 64-B gradient records (see GRAD_* below)
    int has_invd;
};

// Segment length L (entries per work unit).  GMS_SEG_LEN forces it (multiple of 64); otherwise the tile scan picks it
// per frame from the scene depth -- SEG_LEN_SHALLOW up to SEG_DEEP_PER_TILE list entries per tile on average,
// SEG_LEN_DEEP above (short segments shorten the serial walk of a wave; deep scenes pay for them in per-segment state
// and in the length of the finalize / suffix chains) -- and leaves it in scan_out[3], where every later kernel of the
// frame, forward and backward, reads it.  Buffers are sized for the shortest L that can be chosen.
constexpr uint32_t SEG_LEN_SHALLOW = 128, SEG_LEN_DEEP = 256, SEG_DEEP_PER_TILE = 512;
uint32_t seg_len_forced();     // GMS_SEG_LEN, or 0
uint32_t seg_len_min();        // the forced L, or SEG_LEN_SHALLOW: what BinningState is sized and carved with
uint32_t unit_run();
inline uint32_t max_units(uint32_t T, uint64_t instances, uint32_t L) { return T + (uint32_t)(instances / L) + 1u; }
inline uint32_t max_slots(uint64_t instances, uint32_t L) { return 2u * (uint32_t)(instances / L) + 2u; }

// ---- device-side helpers shared by the blend kernel files -------------------------------------------------------
#ifndef GMS_QUEUE
#define GMS_QUEUE 256
#endif
constexpr int QUEUE = GMS_QUEUE;   // LDS splat-queue entries per batch (<= BLOCK)
constexpr uint32_t TLOC_HEAD_ENTRIES = 1024;   // list entries (per tile) whose transmittance products are always evaluated
__host__ __device__ __forceinline__ int tloc_head(uint32_t L) { return (int)(TLOC_HEAD_ENTRIES / L > 0 ? TLOC_HEAD_ENTRIES / L : 1u); }

// Exponent of the Gaussian at a pixel offset, in ONE documented operation order (discontinuity rule:
// two explicit FMAs, every other product rounded on its own, no contraction; the CPU checker evaluates the same chain).  EVERY kernel that evaluates a
// (pixel, splat) pair -- transmittance products, forward walk, backward walk -- goes through this one function, so the
// discrete skip decisions (power > 0, alpha < 1/255) are bit-identical between them: a splat can never count for a
// segment's transmittance product but not for its colour, or for the forward but not for the backward.
__device__ __forceinline__ float pair_power(float A, float B, float C, float dx, float dy)
{
#pragma clang fp contract(off)
    const float a = A * dx, c = C * dy, b = B * dx;
    const float s = __fmaf_rn(a, dx, c * dy);
    return __fmaf_rn(-0.5f, s, -(b * dy));
}

struct Unit {
    int tile, seg, nseg, tx, ty;
    uint32_t tile_beg;     // first entry of the tile in the sorted list
    uint32_t beg, end;     // this unit's entries [beg, end)
    uint32_t slot0;        // first segment-state slot of the tile (multi-segment tiles)
    uint32_t L;            // this frame's segment length
};

// Block -> unit.  The unit table lists full segments (seg_len entries) first, then the tiles' partial last
// segments, then the units of empty tiles: blocks are dispatched in index order, so the long-running units
// start first and the tail of the kernel is made of short ones.  The dispatcher places block b on XCD b % 8 (observed; speed only).  Units are dealt
// to the XCDs in runs of UNIT_RUN consecutive units: a run is a stretch of neighbouring tiles that
// gather the same splat records (L2 locality), while successive runs rotate over the eight XCDs so the
// heavy image centre and the empty border are spread over all of them (units are far from equal work).
constexpr uint32_t UNIT_RUN_MAX = 64;   // grid padding granularity; the run length itself is g.unit_run

__device__ __forceinline__ bool load_unit_at(const BlendGrid &g, Unit &u, uint32_t s, uint32_t xcd)
{
    const uint32_t nunits = g.unit_first[g.T];
    const uint32_t run = g.unit_run;
    const uint32_t idx = ((s / run) * 8u + xcd) * run + (s % run);
    if (idx >= nunits || idx >= g.max_units) return false;   // (max_units: overflowed optimistic launch)
    // one 32-byte record per unit (written by fill_units): a single round trip instead of unit -> tile tables
    const uint4 r0 = g.unit_tile[2 * (size_t)idx], r1 = g.unit_tile[2 * (size_t)idx + 1];
    u.tile = (int)r0.x;
    u.seg = (int)r0.y;
    u.nseg = (int)r0.z;
    u.slot0 = r0.w;
    u.tx = u.tile % g.gx; u.ty = u.tile / g.gx;
    u.tile_beg = r1.x;
    const uint32_t tile_end = r1.y;
    u.L = g.scan_out[3];
    u.beg = u.tile_beg + (uint32_t)u.seg * u.L;
    u.end = min(tile_end, u.beg + u.L);
    return (uint64_t)tile_end <= g.capacity;      // overflowed optimistic launch: host re-runs
}

__device__ __forceinline__ bool load_unit(const BlendGrid &g, Unit &u) { return load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u); }

// Experiment switches (env GMS_DBG) exist only in builds made with `make EXPERIMENTS=1`;
// in the default build every `dbg_on()` is a compile-time false and the branches disappear from the kernels.
#ifndef GMS_EXPERIMENTS
#define GMS_EXPERIMENTS 0
#endif
__device__ __forceinline__ bool dbg_on(const BlendGrid &g, uint32_t bit) { return GMS_EXPERIMENTS && (g.dbg & bit) != 0; }

// experiment (GMS_DBG timelines): lane 0 of every wave records [start, end] of its wave on every exit path
struct Stamp {
    unsigned long long *buf, t0; uint32_t block; bool on;
    __device__ __forceinline__ Stamp(unsigned long long *b) : buf(b), t0(b ? wall_clock64() : 0ull), block(blockIdx.x),
        on(b != nullptr && (threadIdx.x & 63) == 0) { if (on) buf = b + 8ull * 65536ull * (threadIdx.x >> 6); }
    __device__ __forceinline__ ~Stamp() { if (on) { buf[2 * (size_t)block] = t0; buf[2 * (size_t)block + 1] = wall_clock64(); } }
};

// units covered by the grid the blend launches use for `max_units` (the XCD run mapping pads to 8 * UNIT_RUN_MAX blocks)
inline uint32_t blend_grid_units(uint32_t max_units) { return 8u * UNIT_RUN_MAX * ((max_units + 8u * UNIT_RUN_MAX - 1u) / (8u * UNIT_RUN_MAX)); }

int32_t launch_blend_forward(const BlendGrid &g, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream);
int32_t launch_blend_backward(const BlendGrid &g, const BlendBwdArgs &a, uint32_t max_units, bool debug, hipStream_t stream);

}  // namespace gms
