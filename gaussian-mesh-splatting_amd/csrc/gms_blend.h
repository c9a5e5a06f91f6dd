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
    uint32_t seg_len;              // L: entries per segment (multiple of 256)
    const uint32_t *tile_offset;   // [T+1]
    const uint32_t *unit_first;    // [T+1] first unit of each tile; unit_first[T] = number of units
    const uint32_t *mseg_first;    // [T+1] first segment-state slot of each multi-segment tile
    const uint2 *unit_tile;        // [units] (tile, segment) of each unit, heaviest first
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

// segment length used by this process (env GMS_SEG_LEN, default 256; multiple of 256)
uint32_t seg_len();
uint32_t unit_run();
inline uint32_t max_units(uint32_t T, uint64_t instances, uint32_t L) { return T + (uint32_t)(instances / L) + 1u; }
inline uint32_t max_slots(uint64_t instances, uint32_t L) { return 2u * (uint32_t)(instances / L) + 2u; }

int32_t launch_blend_forward(const BlendGrid &g, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream);
int32_t launch_blend_backward(const BlendGrid &g, const BlendBwdArgs &a, uint32_t max_units, bool debug, hipStream_t stream);

}  // namespace gms
