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
    uint32_t *tile_cmax;           // [T] micro mode: bits of the largest |colour component| of the tile's splats (ImageState)
    uint64_t capacity;             // instances the binning buffer can hold
    uint32_t max_units;            // entries of unit_tile
    uint32_t dbg;                  // experiment switches (env GMS_DBG; 0 in production)
    uint32_t unit_run;             // consecutive units dealt to one XCD (power of two <= 64)
    unsigned long long *dbg_buf;   // GMS_DBG&16: per block {start, end} wall clock (100 MHz), else NULL
    uint16_t *mmask;               // micro mode: [capacity] block masks per instance, see BinningState
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
    float *part;                // deterministic mode (gmsplat.h): partial records [instances][nsub][16], indexed by the instance's
                                // position in the sorted key list; micro-tile kernels nsub = 1, quadrant kernels nsub = 4; else NULL
};

// Segment length L (entries per work unit).  GMS_SEG_LEN forces it (multiple of 64); otherwise the tile scan picks it
// per frame from the scene depth -- SEG_LEN_SHALLOW up to SEG_DEEP_PER_TILE list entries per tile on average,
// SEG_LEN_DEEP above (short segments shorten the serial walk of a wave; deep scenes pay for them in per-segment state
// and in the length of the finalize / suffix chains) -- and leaves it in scan_out[3], where every later kernel of the
// frame, forward and backward, reads it.  Buffers are sized for the shortest L that can be chosen.
constexpr uint32_t SEG_LEN_SHALLOW = 128, SEG_LEN_DEEP = 256, SEG_DEEP_PER_TILE = 512;
// Very deep scenes (config-5 size: 2 880 entries per tile, tiles up to 26 k): 512-entry segments -- measured at 997 600
// Gaussians / 1024^2, forward-only renders/s 700 / 735 / 698 / 714 / 680 / 579 at L = 256 / 512 / 768 / 1024 / 1536 / 2048,
// fwd+bwd 428 / 430 it/s at 256 / 512.
constexpr uint32_t SEG_LEN_VERY_DEEP = 512, SEG_VERY_DEEP_PER_TILE = 2048;
// Micro-tile compositing (blend_micro.hip; GMS_MICRO=0 selects the quadrant-wave kernels of blend.hip): one segment length
// for every frame (GMS_SEG_LEN, default SEG_LEN_MICRO, at most 256: a block's list holds entry indices within the unit, one byte each).
constexpr uint32_t SEG_LEN_MICRO = 256;
bool micro_mode();                                  // the micro-tile path may be taken (GMS_MICRO unset or 1)
bool use_micro(uint64_t capacity, int T);           // ... and is, for a frame whose binning capacity is `capacity`
uint32_t seg_len_forced();     // GMS_SEG_LEN, or 0
uint32_t seg_len_min();        // the forced L, or SEG_LEN_SHALLOW: what BinningState is sized and carved with
uint32_t unit_run();
inline uint32_t max_units(uint32_t T, uint64_t instances, uint32_t L) { return T + (uint32_t)(instances / L) + 1u; }
inline uint32_t max_slots(uint64_t instances, uint32_t L) { return 2u * (uint32_t)(instances / L) + 2u; }

// ---- device-side helpers shared by the blend kernel files -------------------------------------------------------
#ifndef GMS_MICRO_DEFAULT
#define GMS_MICRO_DEFAULT 1
#endif
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

// ---- The compositing step of the QUADRANT kernels' forward walks, predicated through EXEC (round 6; blend_micro.hip::fwd_step_exec has the
// reasoning: a walk costs what it issues, and the compiler's form of the nested conditions is ten instructions of mask algebra, saveexec and
// branches per entry).  A lane's state is `alive` (1 until the pixel stops, lies outside the image or was dead on entry); the entry is the
// same for the whole wave, so its validity is the caller's (scalar) branch and `pos` -- what n_contrib records -- is a scalar.  The tests
// narrow EXEC themselves, the stop rule takes its lanes out and clears `alive`, the body runs under what is left.  Same arithmetic, same order.
__device__ __forceinline__ void quad_step_exec(float pw, float al, float cr, float cg, float cb, float cd, uint32_t pos, float &T, float &C0, float &C1,
                                               float &C2, float &Dp, int &alive, uint32_t &last)
{
    uint64_t sv; float tmp, w;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_lt_i32_e32 vcc, 0, %[alive]\n\t"
                 "v_cmpx_ge_f32_e32 vcc, 0, %[pw]\n\t"
                 "v_cmpx_le_f32_e32 vcc, %[amin], %[al]\n\t"
                 "v_sub_f32_e32 %[tmp], 1.0, %[al]\n\t"
                 "v_mul_f32_e32 %[tmp], %[T], %[tmp]\n\t"
                 "v_cmp_ngt_f32_e32 vcc, %[tmin], %[tmp]\n\t"          // NOT (T' < 1e-4): the lanes that composite this entry
                 "v_cndmask_b32_e32 %[alive], 0, %[alive], vcc\n\t"
                 "s_and_b64 exec, exec, vcc\n\t"
                 "v_mul_f32_e32 %[w], %[al], %[T]\n\t"
                 "v_fmac_f32_e32 %[C0], %[cr], %[w]\n\t"
                 "v_fmac_f32_e32 %[C1], %[cg], %[w]\n\t"
                 "v_fmac_f32_e32 %[C2], %[cb], %[w]\n\t"
                 "v_fmac_f32_e32 %[Dp], %[cd], %[w]\n\t"
                 "v_mov_b32_e32 %[T], %[tmp]\n\t"
                 "v_mov_b32_e32 %[last], %[pos]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [tmp] "=&v"(tmp), [w] "=&v"(w), [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [Dp] "+v"(Dp),
                   [alive] "+v"(alive), [last] "+v"(last)
                 : [pw] "v"(pw), [al] "v"(al), [amin] "s"(ALPHA_MIN), [tmin] "s"(T_MIN), [pos] "s"(pos), [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cb), [cd] "v"(cd)
                 : "vcc");
}
// ... and of the transmittance products: Tl *= 1 - alpha where power <= 0 and alpha >= 1/255
__device__ __forceinline__ void quad_tloc_step_exec(float pw, float al, float &Tl)
{
    uint64_t sv; float tmp;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_ge_f32_e32 vcc, 0, %[pw]\n\t"
                 "v_cmpx_le_f32_e32 vcc, %[amin], %[al]\n\t"
                 "v_sub_f32_e32 %[tmp], 1.0, %[al]\n\t"
                 "v_mul_f32_e32 %[Tl], %[Tl], %[tmp]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [tmp] "=&v"(tmp), [Tl] "+v"(Tl)
                 : [pw] "v"(pw), [al] "v"(al), [amin] "s"(ALPHA_MIN)
                 : "vcc");
}

// inclusive rectangle of pixel centres
struct RectF { float wx0, wy0, wx1, wy1; };

// Does any pixel centre of the rectangle (a wave's 8x8 quadrant, a 4x4 block) see this splat with alpha >= 1/255?  Conservative (it may keep a
// pair the per-pixel test skips, never the reverse):
//   (1) bounding box of the {alpha >= 1/255} ellipse against the quadrant (rejects most far entries with 4 compares);
//   (2) EXACT ellipse-vs-rectangle: the minimum over the rectangle of Q(d) = A dx^2 + 2 B dx dy + C dy^2 (convex: the
//       centre if inside, else the best of the four edges' clamped 1-D minima) against 2 (ln(255 op) + 1e-3), recomputed
//       from the record's opacity with the SAME intrinsic the preprocess kernel used for the extents (deriving it from
//       ex^2 det(conic) / C loses it to the cancellation in A C - B^2 for long thin splats).  Mesh-bound splats are flat and
//       often diagonal on screen, where the box of the ellipse is loose.  The slack covers the float error of the edge
//       minima: 0.01 absolute + 1e-4 relative in Q, plus 4e-6 of the GROSS terms |A| mx^2 + 2 |B| mx my + |C| my^2 at the
//       far corner (the three terms cancel for a thin splat whose centre is hundreds of pixels away).
__device__ __forceinline__ bool rect_hit(const float4 q0, const float C, const float op, const float4 q2, const RectF &p, bool bbox_only = false)
{
    // (the record's extents already carry the float32 noise of the per-pixel exponent: cull_extents below)
    if (q0.x + q2.z < p.wx0 || q0.x - q2.z > p.wx1 || q0.y + q2.w < p.wy0 || q0.y - q2.w > p.wy1) return false;
    if (bbox_only) return true;                                       // experiment switch (GMS_DBG & 512)
    const float A = q0.z, B = q0.w;
    const float dx0 = p.wx0 - q0.x, dx1 = p.wx1 - q0.x, dy0 = p.wy0 - q0.y, dy1 = p.wy1 - q0.y;
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;          // centre inside the quadrant
    const float iA = __builtin_amdgcn_rcpf(A), iC = __builtin_amdgcn_rcpf(C);
    const float thr = 2.f * (__logf(255.f * op) + 1e-3f);
    const float B2 = 2.f * B;
    float qmin;
    {
        const float ya = fminf(fmaxf(-B * dx0 * iC, dy0), dy1), yb = fminf(fmaxf(-B * dx1 * iC, dy0), dy1);
        const float xa = fminf(fmaxf(-B * dy0 * iA, dx0), dx1), xb = fminf(fmaxf(-B * dy1 * iA, dx0), dx1);
        const float e0 = dx0 * (A * dx0 + B2 * ya) + C * ya * ya, e1 = dx1 * (A * dx1 + B2 * yb) + C * yb * yb;
        const float e2 = xa * (A * xa + B2 * dy0) + C * dy0 * dy0, e3 = xb * (A * xb + B2 * dy1) + C * dy1 * dy1;
        qmin = fminf(fminf(e0, e1), fminf(e2, e3));
    }
    const float mx = fmaxf(fabsf(dx0), fabsf(dx1)), my = fmaxf(fabsf(dy0), fabsf(dy1));
    const float gross = mx * (A * mx + fabsf(B2) * my) + C * my * my;
    return qmin <= thr * 1.0001f + 0.01f + 4e-6f * gross;
}
// Half extents of the bounding box the culls test (record words 10, 11).  {alpha >= 1/255} is the ellipse Q <= thr with
// thr = 2 (ln(255 op) + 1e-3), whose box is sqrt(cov_xx thr) x sqrt(cov_yy thr).  The per-pixel exponent, though, is only known
// to ~1e-7 of its GROSS terms, and beyond the tip of a long thin splat Q grows so slowly that this noise accepts pixels outside
// that box (tests/test_filter_emulation.py: 2 px at sigma_1 = 100 px and minimum width, 250 px at 1 000).  So the box stored is
// that of {Q <= thr_G}, thr_G = 1.0001 thr + 0.01 + 4e-6 * (gross terms of Q at the corner of this very box).  With
// rho = (gross at the corner) / thr_G = A cov_xx + 2 |B| sqrt(cov_xx cov_yy) + C cov_yy (a property of the shape: 2 for a
// circle, ~4 (sigma_1 / sigma_2)^2 for a thin diagonal splat) the fixed point is thr_G = (1.0001 thr + 0.01) / (1 - 4e-6 rho);
// from rho = 1.25e5 on (aspect ~180 and beyond) the exponent is noise over the whole footprint and the box is infinite.
// block_mask recovers thr_G from the stored extents (thr_G = 1.0001 thr + 0.01 + 4e-6 (A ex^2 + 2 |B| ex ey + C ey^2)).
__device__ __forceinline__ void cull_extents(float a_d, float c_d, float cA, float cB, float cC, float opp, float &ex, float &ey)
{
    const float tau = __logf(255.f * opp);
    if (tau < -1e-3f) { ex = -1e30f; ey = -1e30f; return; }
    const float thr0 = 2.f * (tau + 1e-3f) * 1.0001f + 0.01f;
    const float rho = cA * a_d + 2.f * fabsf(cB) * sqrtf(a_d * c_d) + cC * c_d;
    const float k = 1.f - 4e-6f * rho;
    if (!(k > 0.5f)) { ex = 1e30f; ey = 1e30f; return; }
    const float thrG = thr0 / k;
    ex = sqrtf(a_d * thrG); ey = sqrtf(c_d * thrG);
}

// Which of the tile's sixteen 4x4 blocks can see the splat with alpha >= 1/255 (bit by * 4 + bx).  Conservative: it may keep
// a (splat, block) pair no pixel of the block accepts, never the reverse.
//
// The region {alpha >= 1/255} is the ellipse Q(d) = A dx^2 + 2 B dx dy + C dy^2 <= thr, thr = 2 (ln(255 op) + 1e-3) (the same
// inflated threshold the record's extents were built from).  A block spans a whole band of four pixel rows, so it meets the
// (convex) ellipse iff the x-projection of (ellipse intersected with the band) overlaps the block's columns.  The slice of the
// ellipse at height dy is dx in (-B dy -+ sqrt(D)) / A with D(dy) = A thr - det dy^2; the right end is concave in dy with its
// maximum at the ellipse's rightmost point dy_R = -B ex / C, the left end convex with its minimum at -dy_R, so the band's
// projection is [left(clamp(-dy_R)), right(clamp(dy_R))] with the clamp to the band, and the band misses the ellipse iff D
// is negative at the clamped point.  Four bands x (2 square roots + 8 compares) instead of sixteen rectangle tests.
// det = A C - B^2 cancels badly for long thin splats; it is taken from the record's extent instead (ey^2 = thr A / det,
// computed from the covariance in the preprocess kernel).  Slack as in rect_hit: 0.01 + 1e-4 thr + 4e-6 x the gross terms of Q
// at the far corner of the tile (the float error of the per-pixel exponent itself), plus 1e-3 pixel on the interval ends.
__device__ __forceinline__ uint32_t block_mask(const SplatRec &r, float tx0, float ty0)
{
    const float px = r.q0.x, py = r.q0.y, A = r.q0.z, B = r.q0.w, C = r.q1.x, ex = r.q2.z, ey = r.q2.w;
    const float thr = 2.f * (__logf(255.f * r.q1.y) + 1e-3f);
    // bounding box against the tile (also rejects the ext = -1e30 records of splats below 1/255 everywhere).  The stored
    // extents are those of {Q <= thrG}, thrG = 1.0001 thr + 0.01 + 4e-6 (gross terms of Q at the box's own corner): they carry
    // the float32 noise of the per-pixel exponent (cull_extents below); 1e30 = noise over the whole footprint.
    if (px + ex < tx0 || px - ex > tx0 + 15.f || py + ey < ty0 || py - ey > ty0 + 15.f) return 0u;
    if (ex > 1e29f) return 0xffffu;
    uint32_t m = 0;
    if (!(thr > 1e-4f)) {
        // a splat that reaches 1/255 only within rounding of its centre: the bounding box alone (it carries the inflation)
#pragma unroll
        for (int by = 0; by < 4; by++) {
            const float y0 = ty0 + 4.f * by;
            const bool yhit = !(py + ey < y0 || py - ey > y0 + 3.f);
#pragma unroll
            for (int bx = 0; bx < 4; bx++) {
                const float x0 = tx0 + 4.f * bx;
                if (yhit && !(px + ex < x0 || px - ex > x0 + 3.f)) m |= 1u << (by * 4 + bx);
            }
        }
        return m;
    }
    const float thrG = thr * 1.0001f + 0.01f + 4e-6f * (ex * (A * ex + 2.f * fabsf(B) * ey) + C * ey * ey);   // what (ex, ey) belong to
    const float mx = fmaxf(fabsf(tx0 - px), fabsf(tx0 + 15.f - px)), my = fmaxf(fabsf(ty0 - py), fabsf(ty0 + 15.f - py));
    const float gross = mx * (A * mx + 2.f * fabsf(B) * my) + C * my * my;
    const float thr2 = fmaxf(thrG, thr * 1.0001f + 0.01f + 4e-6f * gross);   // inflated threshold for this tile
    const float grow = thr2 * __builtin_amdgcn_rcpf(thrG);               // ex'^2 / ex^2 = ey'^2 / ey^2
    const float iA = __builtin_amdgcn_rcpf(A), iC = __builtin_amdgcn_rcpf(C);
    const float AT = A * thr2;
    const float inv_ey2 = __builtin_amdgcn_rcpf(ey * ey * grow);        // 1 / ey'^2  (det = A thr2 / ey'^2)
    const float dyR = -B * ex * __builtin_amdgcn_sqrtf(grow) * iC;       // height of the rightmost point (the leftmost: -dyR)
#pragma unroll
    for (int by = 0; by < 4; by++) {
        const float dya = ty0 + 4.f * by - py, dyb = dya + 3.f;
        const float c1 = fminf(fmaxf(dyR, dya), dyb), c2 = fminf(fmaxf(-dyR, dya), dyb);
        const float D1 = AT * (1.f - c1 * c1 * inv_ey2), D2 = AT * (1.f - c2 * c2 * inv_ey2);
        const bool band = D1 >= 0.f;                                   // (D2 >= 0 says the same: both points lie in the band)
        const float xr = px + (__builtin_amdgcn_sqrtf(fmaxf(D1, 0.f)) - B * c1) * iA + 1e-3f;
        const float xl = px - (__builtin_amdgcn_sqrtf(fmaxf(D2, 0.f)) + B * c2) * iA - 1e-3f;
#pragma unroll
        for (int bx = 0; bx < 4; bx++) {
            const float x0 = tx0 + 4.f * bx;
            if (band && xr >= x0 && xl <= x0 + 3.f) m |= 1u << (by * 4 + bx);
        }
    }
    return m;
}

// per-pixel state of the back-to-front recurrence
struct BwdState {
    float T, acc0, acc1, acc2, accd;      // transmittance after, and colour / inverse depth composited behind, the cursor
};

// One splat against one pixel, branch-free: updates the recurrence and returns ten partial sums
// v = (q dx, q dy, q dx^2, q dx dy, q dy^2, q, w r', w g', w b', w d') with q = dL/dG * G: the geometric part is
// accumulated as MOMENTS of q -- the map to the gradients of (mean2D, conic, opacity) is linear with per-splat
// constants and is applied once per Gaussian in preprocess_bwd.  A lane for which the pair is inactive runs the
// same code with alpha = G = 0: a zero-alpha splat is transparent to the recurrence (T and the colour behind stay
// exactly as they were) and every output becomes exactly 0.
template <bool INVD>
__device__ __forceinline__ void bwd_step(BwdState &s, bool act, const float4 &r1, const float4 &r2, float dx, float dy,
                                         float G_in, float alpha_in, float dp0, float dp1, float dp2, float dinvd,
                                         float Tfinal_bgdot, float *v)
{
    const float alpha = act ? alpha_in : 0.f;
    const float Gop = act ? G_in * r1.y : 0.f;                 // alpha = min(0.99, op*G) is straight-through
    const float om = 1.f - alpha;
    const float rcp1ma = __builtin_amdgcn_rcpf(om);            // 1 - alpha >= 0.01; rcp(1) == 1
    s.T = s.T * rcp1ma;                                        // transmittance in front of this splat
    const float w = alpha * s.T;
    const float d0 = r1.z - s.acc0, d1 = r1.w - s.acc1, d2 = r2.x - s.acc2, dd = INVD ? r2.y - s.accd : 0.f;
    float dL_dalpha = d0 * dp0 + d1 * dp1 + d2 * dp2;
    if (INVD) dL_dalpha += dd * dinvd;
    dL_dalpha = dL_dalpha * s.T - Tfinal_bgdot * rcp1ma;
    // colour composited behind the NEXT (nearer) splat: alpha c + (1 - alpha) behind, evaluated as behind + alpha (c - behind) -- one FMA on
    // the difference the line above already holds instead of a multiply and an FMA per channel (round 6: the walk is bound by VALU issue;
    // the same convex combination, rounded once instead of twice)
    s.acc0 = __fmaf_rn(alpha, d0, s.acc0);
    s.acc1 = __fmaf_rn(alpha, d1, s.acc1);
    s.acc2 = __fmaf_rn(alpha, d2, s.acc2);
    if (INVD) s.accd = __fmaf_rn(alpha, dd, s.accd);
    v[6] = w * dp0; v[7] = w * dp1; v[8] = w * dp2; v[9] = INVD ? w * dinvd : 0.f;
    const float q = Gop * dL_dalpha;
    const float qx = q * dx, qy = q * dy;
    v[0] = qx; v[1] = qy; v[2] = qx * dx; v[3] = qx * dy; v[4] = qy * dy; v[5] = q;
}

// ---- 64-bit fixed-point gradient table of the micro-tile backward (blend_micro.hip, where the scheme is described)
constexpr int FX_SHIFT = 47;

// x < 2^fx_exp(x) for every finite x >= 0 (biased exponent - 126; zero and denormals: -126)
__device__ __forceinline__ int fx_exp(float x) { return (int)((__float_as_uint(x) >> 23) & 0xffu) - 126; }

// The power of two a partial sum is scaled by before it is rounded: k = FX_SHIFT - E.  `base` = FX_SHIFT minus the unit-level part of
// the field's exponent (a constant of the lane that holds the field), `op_exp` = 126 (the opacity bound is part of `base`), or -- builds
// with GMS_FX_ENTRY_OPACITY=1 -- the biased exponent of the entry's own opacity for the geometric fields (op < 2^(op_exp - 126)).
// Kept inside a float's exponent range.
__device__ __forceinline__ int fx_scale_exp(int base, uint32_t op_exp) { return min(max(base + 126 - (int)op_exp, -100), 100); }

// round(y * 2^k) as a 64-bit integer for |y| * 2^k <= 2^47: the product with a power of two is exact in float, and adding 1.5 * 2^52 in
// double leaves its nearest integer (ties to even) in the low mantissa bits for |x| < 2^51 -- no branch, two f64 instructions.
// (The first version shifted the float's mantissa by hand: 25 instructions with two divergent branches.)
__device__ __forceinline__ long long fx_from_float(float y, int k)
{
    const double MAGIC = 6755399441055744.0;
    // saturated at +-2^50 (one v_med3): a bound exceeded more than 8-fold, or a non-finite pixel gradient, clamps this one partial sum
    // instead of leaving wrong bits in the low mantissa (NaN -> -2^50 by v_med3's ordering, still a finite table entry)
    const float ys = __builtin_amdgcn_fmed3f(y * __uint_as_float((uint32_t)(k + 127) << 23), -1125899906842624.f, 1125899906842624.f);
    return __double_as_longlong((double)ys + MAGIC) - __double_as_longlong(MAGIC);
}
// The same conversion with the lane's scale prepared once (the walk of micro_bwd: k is a constant of the lane and unit): the clamp is taken
// on y itself against 2^(50 - k) and the scaling rides in the f64 FMA that adds the magic number -- v_med3, v_cvt_f64_f32, v_fma_f64 and the
// integer add on the high word: one VALU instruction fewer per table add in a kernel bound by VALU issue (round 6).  Bit-identical to
// fx_from_float: y 2^k is exact in float and in double alike, so the one rounding is the FMA's in both (tests/test_gpu_fixed_point.py).
struct FxScale { double two_k; float y_max; };
__device__ __forceinline__ FxScale fx_prepare(int k)
{
    FxScale s;
    s.two_k = __longlong_as_double((long long)(k + 1023) << 52);
    s.y_max = __uint_as_float((uint32_t)(min(50 - k, 127) + 127) << 23);          // 2^(50 - k): |y| beyond it saturates (k >= -100 -> exponent <= 150: clamped to float's range)
    return s;
}
__device__ __forceinline__ long long fx_from_float(float y, const FxScale &s)
{
    const double MAGIC = 6755399441055744.0;
    const float yc = __builtin_amdgcn_fmed3f(y, -s.y_max, s.y_max);
    return __double_as_longlong(__builtin_fma((double)yc, s.two_k, MAGIC)) - __double_as_longlong(MAGIC);
}

// (exact in double, one rounding to float)
__device__ __forceinline__ float fx_to_float(long long v, int k) { return (float)ldexp((double)v, -k); }

// unit-level exponents: the bounds on the pixel gradients, the colours and the centre-to-corner distances hold for the whole unit
struct FxTile { int eK, eX, eY, eCol, eId; };
// FX_SHIFT minus the unit-level part of a field's exponent
__device__ __forceinline__ int fx_field_base(const FxTile &t, int cx, int cy, int kind)
{
    return FX_SHIFT - (kind == 1 ? t.eCol : (kind == 2 ? t.eId : t.eK + cx * t.eX + cy * t.eY));
}

// field -> (power of dx, power of dy, kind: 0 geometry, 1 colour weight, 2 inverse-depth weight)
__device__ __forceinline__ void fx_field_kind(int f, int &cx, int &cy, int &kind)
{
    cx = f == GRAD_MX || f == GRAD_CB ? 1 : (f == GRAD_CA ? 2 : 0);
    cy = f == GRAD_MY || f == GRAD_CB ? 1 : (f == GRAD_CC ? 2 : 0);
    kind = f == GRAD_ID ? 2 : (f >= GRAD_R && f <= GRAD_B ? 1 : 0);
}

struct Unit {
    int tile, seg, nseg, tx, ty;
    uint32_t tile_beg;     // first entry of the tile in the sorted list
    uint32_t beg, end;     // this unit's entries [beg, end)
    uint32_t slot0;        // first segment-state slot of the tile (multi-segment tiles)
    uint32_t L;            // this frame's segment length
    uint32_t idx;          // index of the unit in the unit table
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
    u.idx = idx;
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

void experiment_switches(BlendGrid &g, uint32_t buf_bits, hipStream_t stream);     // blend.hip; a no-op outside make EXPERIMENTS=1
int32_t launch_blend_forward(const BlendGrid &g, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream);
int32_t launch_blend_backward(const BlendGrid &g, const BlendBwdArgs &a, uint32_t max_units, bool debug, hipStream_t stream);
int32_t launch_micro_forward(const BlendGrid &g, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream);
int32_t launch_micro_backward(const BlendGrid &g, const BlendBwdArgs &a, uint32_t max_units, bool debug, hipStream_t stream);

}  // namespace gms
