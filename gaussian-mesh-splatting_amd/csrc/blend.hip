// blend.hip -- per-tile alpha compositing (forward and backward) for gfx950, segment-parallel.
//
// A tile's depth-sorted splat list is cut into segments of at most L entries (L = seg_len,
// default 128); one work UNIT = (tile, segment).  A unit is walked by four independent waves, each owning an
// 8x8 pixel quadrant of the 16x16 tile and launched as its OWN 64-thread block (the four quadrant blocks of a unit are
// consecutive blocks of one XCD): no block barrier, no quadrant waiting at a barrier for a slower one, a finished
// quadrant frees its slot, and the dispatcher places work at quadrant granularity.  (GMS_FWD_WPB=4 / GMS_BWD_WPB=4
// select the older layout: one 256-thread block per unit, the four waves sharing one 256-entry queue.)
// Compositing is associative, so the segments of a heavy tile (thousands of splats around mesh poles /
// silhouettes) run on different CUs instead of serialising behind one block:
//
//   head     first launch, every unit that depends on nothing: the exact front-to-back walk of each tile's FIRST
//            segment (reference skip/stop tests; single-segment tiles are finished by it, and the T it ends with is
//            that segment's transmittance product) and, for the middle segments of multi-segment tiles,
//            T_loc = prod(1 - alpha) over the segment without termination.  On deep scenes the products run in two
//            phases with a per-tile check in between, so they stop where the tile is already opaque.
//   fwd      second launch, segments 1..: T_start = prod_{k<seg} T_loc_k; a pixel whose T_start < 1e-4 is finished
//            (any further contributing splat fails the T*(1-a) < 1e-4 test); exact walk from T_start.  Multi-segment
//            units write (C, D, T_end, last) partials.
//   finalize (multi-segment tiles) sums the partials in segment order and writes image / final_T /
//            n_contrib; the partials stay in place for the backward pass.
//   bwd      per unit, back-to-front from (T_end, suffix colour / T_end): the reference's recurrence
//            restarted at a segment boundary.  Ten partial sums per (wave, splat) -- moments of q = dL/dG*G and the
//            colour weights -- are reduced over the 64 lanes with a transposing DPP / permlane network and leave the
//            wave as ONE atomic instruction into the splat's 64-byte record.
//
// A wave gathers the unit's splat records 64 at a time into its LDS queue, tests the 64 entries at once against its
// quadrant (exact ellipse-vs-rectangle test on the alpha >= 1/255 ellipse: it can only remove pairs the per-pixel test
// would skip) and walks the ballot survivors, NE (= 4) entries per trip: their alpha evaluations are independent, only
// the recurrence is serial.
#include <stdlib.h>

#include "gms_common.h"
#include "gms_blend.h"

namespace gms {

struct Pix {
    int xi, yi; bool inside; float xf, yf; float wx0, wy0, wx1, wy1;
};

__device__ __forceinline__ Pix pixel_of(const BlendGrid &g, const Unit &u, int wave)
{
    const int lane = threadIdx.x & 63;
    const int qx = u.tx * TILE + (wave & 1) * 8, qy = u.ty * TILE + (wave >> 1) * 8;
    Pix p;
    p.xi = qx + (lane & 7); p.yi = qy + (lane >> 3);
    p.inside = p.xi < g.W && p.yi < g.H;
    p.xf = (float)p.xi; p.yf = (float)p.yi;
    p.wx0 = (float)qx; p.wy0 = (float)qy; p.wx1 = (float)(qx + 7); p.wy1 = (float)(qy + 7);
    return p;
}
__device__ __forceinline__ Pix pixel_of(const BlendGrid &g, const Unit &u) { return pixel_of(g, u, (int)(threadIdx.x >> 6)); }

__device__ __forceinline__ bool quadrant_hit(const SplatRec *recs, int j, int cnt, const Pix &p, bool bbox_only = false)
{
    if (j >= cnt) return false;
    return rect_hit(recs[j].q0, recs[j].q1.x, recs[j].q1.y, recs[j].q2, RectF{p.wx0, p.wy0, p.wx1, p.wy1}, bbox_only);
}

// ------------------------------------------------------------------------------------ tloc
template <int NE, int WPB>
__device__ __forceinline__ void tloc_unit(const BlendGrid &g, const SplatRec *rec, const Unit &u, SplatRec *recs, int phase, int wave)
{
    constexpr int QN = WPB == 4 ? QUEUE : WAVE;
    if (u.nseg == 1 || u.seg == u.nseg - 1) return;
    // phase 0: the first tloc_head(L) segments of every tile; phase 1: the rest, unless the head already
    // finished every pixel of the tile (then the products are irrelevant: write 0, evaluate nothing)
    if (phase >= 0 && (u.seg < tloc_head(u.L)) != (phase == 0)) return;     // phase -1: every segment in one launch
    const int qt = threadIdx.x, lane = qt & 63, tid = wave * WAVE + lane;
    if (phase == 1 && g.tile_dead[u.tile]) {
        g.seg_state[(size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + tid] = 0.f;
        return;
    }
    const Pix p = pixel_of(g, u, wave);
    float Tl = 1.f;
    for (uint32_t base = u.beg; base < u.end; base += QN) {
        // once a pixel's segment product is below 1e-4 every later segment starts dead whatever the exact
        // value: a quadrant whose pixels are all there (or outside the image) stops evaluating
        const bool quad_done = __all(Tl < T_MIN || !p.inside);
        if (WPB == 1 && quad_done) break;
        if (WPB == 4) __syncthreads(); else wave_sync();
        const uint32_t idx = base + qt;
        if (qt < QN && idx < u.end) recs[qt] = rec[(uint32_t)g.keys[idx]];
        if (WPB == 4) __syncthreads(); else wave_sync();
        const int cnt = (int)min((uint32_t)QN, u.end - base);
        if (quad_done) continue;
        for (int chunk = 0; chunk < cnt; chunk += WAVE) {
            uint64_t mask = __ballot(quadrant_hit(recs, chunk + lane, cnt, p, dbg_on(g, 512u)));
            while (mask) {
                // NE queue entries per trip: independent alpha evaluations, sequential transmittance product
                int k[NE]; bool val[NE]; float al[NE], pw[NE];
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    val[e] = mask != 0;
                    k[e] = val[e] ? chunk + __builtin_ctzll(mask) : k[e > 0 ? e - 1 : 0];
                    mask &= mask - 1;
                }
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const float4 r0 = recs[k[e]].q0, r1 = recs[k[e]].q1;
                    const float dx = r0.x - p.xf, dy = r0.y - p.yf;
                    pw[e] = pair_power(r0.z, r0.w, r1.x, dx, dy);
                    al[e] = fminf(ALPHA_MAX, r1.y * __expf(pw[e]));
                }
#pragma unroll
                for (int e = 0; e < NE; e++)
                    if (val[e]) quad_tloc_step_exec(pw[e], al[e], Tl);          // (val: the same for the whole wave)
            }
        }
    }
    g.seg_state[(size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + tid] = Tl;
}

// tile_dead[t] = 1 when the product of the first tloc_head(L) segment transmittances is < 1e-4 for every pixel
__global__ void __launch_bounds__(BLOCK) blend_tloc_check_kernel(BlendGrid g)
{
    const int tile = blockIdx.x;
    const int nseg = (int)(g.unit_first[tile + 1] - g.unit_first[tile]);
    const int nhead = tloc_head(g.scan_out[3]);
    if (nseg <= nhead + 1) return;                 // no phase-1 segment exists (the last one needs no product)
    if ((uint64_t)g.tile_offset[tile + 1] > g.capacity) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xi = (tile % g.gx) * TILE + (wave & 1) * 8 + (lane & 7), yi = (tile / g.gx) * TILE + (wave >> 1) * 8 + (lane >> 3);
    const float *st0 = g.seg_state + (size_t)g.mseg_first[tile] * SEG_FLOATS;
    float T = 1.f;
    for (int k = 0; k < nhead; k++) T *= st0[(size_t)k * SEG_FLOATS + SEG_TLOC * TILE_PIX + tid];
    const int dead = __syncthreads_and(T < T_MIN || xi >= g.W || yi >= g.H);
    if (tid == 0) g.tile_dead[tile] = dead ? 1u : 0u;
}

// ------------------------------------------------------------------------------------ fwd
template <int NE, int WPB>
__device__ __forceinline__ void fwd_unit(const BlendGrid &g, const BlendFwdOut &o, const Unit &u, SplatRec *recs, int wave)
{
    constexpr int QN = WPB == 4 ? QUEUE : WAVE;
    const int qt = threadIdx.x, lane = qt & 63, tid = wave * WAVE + lane;
    const Pix p = pixel_of(g, u, wave);

    float T = 1.f;
    {
        // prefix product of the segments in front, four independent loads per step (same left-to-right order)
        const float *tl = g.seg_state + (size_t)u.slot0 * SEG_FLOATS + SEG_TLOC * TILE_PIX + tid;
        int k = 0;
        for (; k + 4 <= u.seg; k += 4) {
            const float t0 = tl[(size_t)k * SEG_FLOATS], t1 = tl[(size_t)(k + 1) * SEG_FLOATS];
            const float t2 = tl[(size_t)(k + 2) * SEG_FLOATS], t3 = tl[(size_t)(k + 3) * SEG_FLOATS];
            T = T * t0 * t1 * t2 * t3;
        }
        for (; k < u.seg; k++) T *= tl[(size_t)k * SEG_FLOATS];
    }
    const bool dead_on_entry = T < T_MIN;          // only possible for seg > 0
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    int alive = (!p.inside || dead_on_entry) ? 0 : 1;          // (gms_blend.h::quad_step_exec clears it when the stop rule fires)

    for (uint32_t base = u.beg; base < u.end; base += QN) {
        if (WPB == 4) { if (__syncthreads_and(alive == 0)) break; }
        else { if (__all(alive == 0)) break; wave_sync(); }
        const uint32_t idx = base + qt;
        if (qt < QN && idx < u.end) recs[qt] = o.rec[(uint32_t)g.keys[idx]];
        if (WPB == 4) __syncthreads(); else wave_sync();
        const int cnt = (int)min((uint32_t)QN, u.end - base);
        if (__all(alive == 0)) continue;           // wave-uniform: this quadrant is finished
        for (int chunk = 0; chunk < cnt; chunk += WAVE) {
            uint64_t mask = __ballot(quadrant_hit(recs, chunk + lane, cnt, p, dbg_on(g, 512u)));
            while (mask) {
                // NE queue entries per trip: their alpha evaluations are independent (ILP hides the LDS and exp
                // latency when few waves are resident); only the compositing recurrence is sequential
                int k[NE]; bool val[NE]; float al[NE], pw[NE]; float4 r1[NE], r2[NE];
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    val[e] = mask != 0;
                    k[e] = val[e] ? chunk + __builtin_ctzll(mask) : k[e > 0 ? e - 1 : 0];   // wave-uniform queue slots
                    mask &= mask - 1;                                                         // no-op once mask is 0
                }
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const float4 r0 = recs[k[e]].q0;
                    r1[e] = recs[k[e]].q1; r2[e] = recs[k[e]].q2;
                    const float dx = r0.x - p.xf, dy = r0.y - p.yf;
                    pw[e] = pair_power(r0.z, r0.w, r1[e].x, dx, dy);
                    al[e] = fminf(ALPHA_MAX, r1[e].y * __expf(pw[e]));
                }
#pragma unroll
                for (int e = 0; e < NE; e++)
                    if (val[e])          // (the same for the whole wave: a scalar branch)
                        quad_step_exec(pw[e], al[e], r1[e].z, r1[e].w, r2[e].x, r2[e].y, (base - u.tile_beg) + (uint32_t)k[e] + 1u, T, C0, C1, C2, Dp, alive, last);
                if (__all(alive == 0)) break;
            }
        }
    }
    if (u.nseg == 1) {
        if (p.inside) {
            const size_t pid = (size_t)p.yi * g.W + p.xi, HW = (size_t)g.W * g.H;
            o.final_T[pid] = T;
            o.n_contrib[pid] = last;
            o.out_color[pid] = C0 + T * o.bg[0];
            o.out_color[HW + pid] = C1 + T * o.bg[1];
            o.out_color[2 * HW + pid] = C2 + T * o.bg[2];
            o.out_invdepth[pid] = Dp;
        }
    } else {
        float *st = g.seg_state + (size_t)(u.slot0 + u.seg) * SEG_FLOATS;
        st[SEG_C0 * TILE_PIX + tid] = C0; st[SEG_C1 * TILE_PIX + tid] = C1; st[SEG_C2 * TILE_PIX + tid] = C2;
        st[SEG_D * TILE_PIX + tid] = Dp;
        st[SEG_TEND * TILE_PIX + tid] = dead_on_entry ? -1.f : T;
        st[SEG_LAST * TILE_PIX + tid] = __uint_as_float(last);
        // the first segment's exact walk doubles as its transmittance product: a pixel that terminated here is
        // dead on entry to every later segment (any value < 1e-4 says so); one that did not has multiplied
        // exactly the (1 - alpha) factors the product would
        if (u.seg == 0) st[SEG_TLOC * TILE_PIX + tid] = alive == 0 ? 0.f : T;
    }
}

// First launch: every unit that depends on nothing -- the exact walk of each tile's FIRST segment (single-segment
// tiles are finished by it) and, for the middle segments of multi-segment tiles, the transmittance products.
template <int NE, int WPB>
__global__ void __launch_bounds__(WPB * WAVE) blend_head_kernel(BlendGrid g, BlendFwdOut o, int phase)
{
    __shared__ SplatRec recs[WPB == 4 ? QUEUE : WAVE];
    Unit u;
    const uint32_t bs = blockIdx.x >> 3;            // (WPB: see blend_bwd_kernel)
    if (!load_unit_at(g, u, WPB == 4 ? bs : bs >> 2, blockIdx.x & 7u)) return;
    const int wave = WPB == 4 ? (int)(threadIdx.x >> 6) : (int)(bs & 3u);
    Stamp stamp(dbg_on(g, 256u) ? g.dbg_buf : nullptr);
    if (u.seg == 0) { if (phase <= 0 && !dbg_on(g, 32u)) fwd_unit<NE, WPB>(g, o, u, recs, wave); }
    else if (!dbg_on(g, 64u)) tloc_unit<NE, WPB>(g, o.rec, u, recs, phase, wave);
}

// Second launch: segments 1.. of the multi-segment tiles, from the prefix product of the segments in front.
template <int NE, int WPB>
__global__ void __launch_bounds__(WPB * WAVE) blend_fwd_kernel(BlendGrid g, BlendFwdOut o)
{
    __shared__ SplatRec recs[WPB == 4 ? QUEUE : WAVE];
    Unit u;
    const uint32_t bs = blockIdx.x >> 3;
    if (!load_unit_at(g, u, WPB == 4 ? bs : bs >> 2, blockIdx.x & 7u)) return;
    if (u.seg == 0) return;
    const int wave = WPB == 4 ? (int)(threadIdx.x >> 6) : (int)(bs & 3u);
    Stamp stamp(dbg_on(g, 128u) ? g.dbg_buf : nullptr);
    fwd_unit<NE, WPB>(g, o, u, recs, wave);
}

// ------------------------------------------------------------------------------------ finalize
__global__ void __launch_bounds__(BLOCK) blend_finalize_kernel(BlendGrid g, BlendFwdOut o)
{
    const int tile = blockIdx.x;
    const uint32_t first = g.unit_first[tile];
    const int nseg = (int)(g.unit_first[tile + 1] - first);
    if (nseg <= 1) return;
    if ((uint64_t)g.tile_offset[tile + 1] > g.capacity) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % g.gx, ty = tile / g.gx;
    const int xi = tx * TILE + (wave & 1) * 8 + (lane & 7), yi = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    float *st0 = g.seg_state + (size_t)g.mseg_first[tile] * SEG_FLOATS;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, T = 1.f;
    uint32_t last = 0;
    // four segments per step, every load issued before the first use: a 27-segment tile is otherwise 27 dependent
    // memory round trips, and this kernel's duration is its deepest tile
    constexpr int U = 4;
    for (int k0 = 0; k0 < nseg; k0 += U) {
        float te[U], c0[U], c1[U], c2[U], dd[U], la[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const float *st = st0 + (size_t)min(k0 + j, nseg - 1) * SEG_FLOATS;
            te[j] = st[SEG_TEND * TILE_PIX + tid]; c0[j] = st[SEG_C0 * TILE_PIX + tid]; c1[j] = st[SEG_C1 * TILE_PIX + tid];
            c2[j] = st[SEG_C2 * TILE_PIX + tid]; dd[j] = st[SEG_D * TILE_PIX + tid]; la[j] = st[SEG_LAST * TILE_PIX + tid];
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            if (k0 + j < nseg && te[j] >= 0.f) {          // a segment entered dead (te < 0) contributed nothing
                C0 += c0[j]; C1 += c1[j]; C2 += c2[j]; Dp += dd[j];
                T = te[j];
                last = max(last, __float_as_uint(la[j]));
            }
        }
    }
    if (xi < g.W && yi < g.H) {
        const size_t pid = (size_t)yi * g.W + xi, HW = (size_t)g.W * g.H;
        o.final_T[pid] = T;
        o.n_contrib[pid] = last;
        o.out_color[pid] = C0 + T * o.bg[0];
        o.out_color[HW + pid] = C1 + T * o.bg[1];
        o.out_color[2 * HW + pid] = C2 + T * o.bg[2];
        o.out_invdepth[pid] = Dp;
    }
}

// ------------------------------------------------------------------------------------ bwd
// WPB = waves per block.  4: one block per unit, the four quadrant waves share one 256-entry queue and move through it
// in lockstep (two block barriers per queue).  1: one block per (unit, quadrant) -- a wave on its own 64-entry queue: no
// block barrier, no waiting for a slower quadrant, a quadrant that is done frees its slot, and the scheduler places
// work at a quarter of the granularity; the price is that each quadrant gathers the unit's records itself (from L2).
// FAULT: 0 in production; 2 = the negative control "drop the colour composited behind a segment restart" (gmsplat.h,
// gms_set_fault): a separate instantiation, so the production kernel carries no fault branch.
// DET (deterministic mode, gmsplat.h): the ten totals of a (quadrant wave, splat) pair are STORED into the partial record of
// (instance, quadrant) -- a.part[(sorted position * 4 + quadrant) * 16 + field], visited exactly once per frame, zero-filled by
// the host -- instead of being added to the Gaussian's record with atomics.
template <bool INVD, int NE, int WPB, int FAULT = 0, bool DET = false>
__global__ void __launch_bounds__(WPB * WAVE) blend_bwd_kernel(BlendGrid g, BlendBwdArgs a)
{
    constexpr int QN = WPB == 4 ? QUEUE : WAVE;
    __shared__ SplatRec recs[QN];
    __shared__ uint32_t ids[QN];
    __shared__ uint32_t wave_max[4];
    Unit u;
    // WPB == 1: the four quadrants of a unit are consecutive blocks of ONE XCD (they gather the same records)
    const uint32_t bs = blockIdx.x >> 3;
    if (!load_unit_at(g, u, WPB == 4 ? bs : bs >> 2, blockIdx.x & 7u)) return;
    if (u.end <= u.beg) return;
    const int qt = threadIdx.x, lane = qt & 63, wave = WPB == 4 ? qt >> 6 : (int)(bs & 3u);
    const int tid = wave * WAVE + lane;                  // pixel index inside the tile (quadrant-major)
    Stamp stamp(dbg_on(g, 16u) ? g.dbg_buf : nullptr);
    const Pix p = pixel_of(g, u, wave);
    const size_t HW = (size_t)g.W * g.H;
    const size_t pid = (size_t)p.yi * g.W + p.xi;
    const float Tfinal = p.inside ? a.final_T[pid] : 0.f;
    const uint32_t last = p.inside ? a.n_contrib[pid] : 0u;      // 1-based position of the last applied splat
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dinvd = 0.f;
    if (p.inside) {
        dp0 = a.dL_dpix[pid]; dp1 = a.dL_dpix[HW + pid]; dp2 = a.dL_dpix[2 * HW + pid];
        if (INVD) dinvd = a.dL_dinvd[pid];
    }
    const float Tfinal_bgdot = Tfinal * (a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2);
    const uint32_t seg_lo = u.beg - u.tile_beg, seg_hi = u.end - u.tile_beg;   // positions covered by this unit

    BwdState st8 = {Tfinal, 0.f, 0.f, 0.f, 0.f};
    if (u.nseg > 1) {
        const float *st = g.seg_state + (size_t)(u.slot0 + u.seg) * SEG_FLOATS;
        const float te = st[SEG_TEND * TILE_PIX + tid];
        if (te > 0.f) {
            // restart of the recurrence at the segment boundary: T after this segment's last applied splat and
            // the colour composited behind it (sum of the live partials of the later segments) divided by that T
            st8.T = te;
            float S0 = 0.f, S1 = 0.f, S2 = 0.f, SD = 0.f;
            // a pixel that is dead on entry to segment k is dead for every later one: stop at the first.  Four
            // segments per step with all loads issued up front (a deep tile is otherwise a chain of round trips).
            bool stop = false;
            for (int k0 = u.seg + 1; k0 < u.nseg; k0 += 4) {
                float tk[4], c0[4], c1[4], c2[4], dd[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float *sk = g.seg_state + (size_t)(u.slot0 + min(k0 + j, u.nseg - 1)) * SEG_FLOATS;
                    tk[j] = sk[SEG_TEND * TILE_PIX + tid]; c0[j] = sk[SEG_C0 * TILE_PIX + tid];
                    c1[j] = sk[SEG_C1 * TILE_PIX + tid]; c2[j] = sk[SEG_C2 * TILE_PIX + tid];
                    dd[j] = INVD ? sk[SEG_D * TILE_PIX + tid] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (k0 + j >= u.nseg || tk[j] < 0.f) stop = true;
                    if (!stop) { S0 += c0[j]; S1 += c1[j]; S2 += c2[j]; SD += dd[j]; }
                }
                if (__all(stop)) break;
            }
            const float inv = FAULT == 2 ? 0.f : 1.f / te;
            st8.acc0 = S0 * inv; st8.acc1 = S1 * inv; st8.acc2 = S2 * inv; st8.accd = SD * inv;
        }
    }

    // wave_reduce10 leaves the ten totals in lanes 0,8,...,56 (y0) and 4, 36 (y1): those ten lanes add
    // into the ten fields of the splat's 64-byte gradient record with ONE atomic instruction.
    int afield;
    switch (lane >> 3) {
    case 0: afield = GRAD_MX; break;
    case 1: afield = GRAD_CC; break;
    case 2: afield = GRAD_CA; break;
    case 3: afield = GRAD_R; break;
    case 4: afield = GRAD_MY; break;
    case 5: afield = GRAD_OP; break;
    case 6: afield = GRAD_CB; break;
    default: afield = GRAD_G; break;
    }
    if (lane == 4) afield = GRAD_B;
    if (lane == 36) afield = GRAD_ID;
    const bool alane = (lane & 7) == 0 || lane == 4 || (lane == 36 && INVD);
    float *const abase = a.accum + afield;
    const bool use_y1 = (lane & 7) != 0;

    // entries of this unit that some pixel of the tile actually composited: positions [seg_lo, top)
    uint32_t m = min(last, seg_hi);
    m = m > seg_lo ? m : 0u;                       // 0 = this pixel has nothing in this unit
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    uint32_t top = m;
    if (WPB == 4) {
        if (lane == 0) wave_max[wave] = m;
        __syncthreads();
        top = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    }
    if (top == 0) return;
    if (dbg_on(g, 2u)) return;                              // experiment: prologue only

    // the trips over one ballot mask of queue entries (slots chunk .. chunk+63 of the queue whose slot 0 is tile position hi-1)
    auto walk = [&](uint64_t mask, const int chunk, const uint32_t hi) {
    while (mask) {
        // NE queue entries per trip: loads, exp and the wave reductions of different entries are independent
        // and interleave; only the per-pixel recurrence is sequential (entry e is behind entry e+1)
        int k[NE]; bool val[NE], act[NE], any[NE]; float dx[NE], dy[NE], G[NE], al[NE]; float4 r1[NE], r2[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            val[e] = mask != 0;
            k[e] = val[e] ? chunk + __builtin_ctzll(mask) : k[e > 0 ? e - 1 : 0];
            mask &= mask - 1;
        }
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const float4 r0 = recs[k[e]].q0;
            r1[e] = recs[k[e]].q1; r2[e] = recs[k[e]].q2;
            dx[e] = r0.x - p.xf; dy[e] = r0.y - p.yf;
            const float pw = pair_power(r0.z, r0.w, r1[e].x, dx[e], dy[e]);
            G[e] = __expf(pw);
            al[e] = fminf(ALPHA_MAX, r1[e].y * G[e]);
            const uint32_t pos = hi - 1 - (uint32_t)k[e];                  // 0-based tile position
            act[e] = val[e] && pos < last && pw <= 0.f && al[e] >= ALPHA_MIN;
            any[e] = __any(act[e]);
        }
        const bool noatomics = dbg_on(g, 1u);                           // experiment switch
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
            const int f = e + 1;
            if (!(any[e] || any[f])) continue;
            float va[10], vb[10];
            if (any[e] && any[f]) {
                bwd_step<INVD>(st8, act[e], r1[e], r2[e], dx[e], dy[e], G[e], al[e], dp0, dp1, dp2, dinvd, Tfinal_bgdot, va);
                bwd_step<INVD>(st8, act[f], r1[f], r2[f], dx[f], dy[f], G[f], al[f], dp0, dp1, dp2, dinvd, Tfinal_bgdot, vb);
                if (dbg_on(g, 8u)) { float t = 0.f; for (int q = 0; q < 10; q++) t += va[q] + vb[q]; if (t == 123.456f) a.accum[0] = t; continue; }
                float y0a, y1a, y0b, y1b;
                wave_reduce10x2(va, vb, y0a, y1a, y0b, y1b);
                if (noatomics) { if (y0a == 123.456f) a.accum[0] = y1a + y0b + y1b; continue; }
                const size_t ida = ids[k[e]], idb = ids[k[f]];
                if (DET) {
                    if (alane) {
                        a.part[((size_t)(u.tile_beg + hi - 1u - (uint32_t)k[e]) * 4 + wave) * GRAD_STRIDE + afield] = use_y1 ? y1a : y0a;
                        a.part[((size_t)(u.tile_beg + hi - 1u - (uint32_t)k[f]) * 4 + wave) * GRAD_STRIDE + afield] = use_y1 ? y1b : y0b;
                    }
                } else if (alane) {
                    unsafeAtomicAdd(abase + ida * GRAD_STRIDE, use_y1 ? y1a : y0a);   // 10 lanes, one 64-B line
                    unsafeAtomicAdd(abase + idb * GRAD_STRIDE, use_y1 ? y1b : y0b);
                }
            } else {
                // (static indices only: a runtime-selected element would push the arrays to scratch)
                if (any[e]) bwd_step<INVD>(st8, act[e], r1[e], r2[e], dx[e], dy[e], G[e], al[e], dp0, dp1, dp2, dinvd, Tfinal_bgdot, va);
                else bwd_step<INVD>(st8, act[f], r1[f], r2[f], dx[f], dy[f], G[f], al[f], dp0, dp1, dp2, dinvd, Tfinal_bgdot, va);
                float y0, y1;
                wave_reduce10(va[0], va[1], va[2], va[3], va[4], va[5], va[6], va[7], va[8], va[9], y0, y1);
                if (noatomics) { if (y0 == 123.456f) a.accum[0] = y1; continue; }
                if (DET) {
                    if (alane) a.part[((size_t)(u.tile_beg + hi - 1u - (uint32_t)(any[e] ? k[e] : k[f])) * 4 + wave) * GRAD_STRIDE + afield] = use_y1 ? y1 : y0;
                } else if (alane) unsafeAtomicAdd(abase + (size_t)ids[any[e] ? k[e] : k[f]] * GRAD_STRIDE, use_y1 ? y1 : y0);
            }
        }
    }
    };

    for (uint32_t hi = top; hi > seg_lo; hi = (hi - seg_lo) > QN ? hi - QN : seg_lo) {
        const int cnt = (int)min((uint32_t)QN, hi - seg_lo);
        if (WPB == 4) __syncthreads(); else wave_sync();     // previous queue fully consumed
        if (qt < cnt) {
            const uint32_t e = hi - 1 - qt;           // queue slot 0 = backmost entry
            const uint32_t id = (uint32_t)g.keys[u.tile_beg + e];
            ids[qt] = id;
            recs[qt] = a.rec[id];
        }
        if (WPB == 4) __syncthreads(); else wave_sync();
        if (m == 0) continue;                         // wave-uniform: quadrant has nothing in this unit
        for (int chunk = 0; chunk < cnt; chunk += WAVE) {
            // m = furthest position any pixel of this quadrant composited: entries behind it are dead here
            const uint64_t mask = __ballot(quadrant_hit(recs, chunk + lane, cnt, p, dbg_on(g, 512u)) && (hi - 1 - (uint32_t)(chunk + lane)) < m);
            if (dbg_on(g, 4u)) { if (mask == 0x123456789ull) a.accum[1] = 1.f; continue; }   // experiment: queue fill + cull only
            walk(mask, chunk, hi);
        }
    }
}

// ------------------------------------------------------------------------------------ host
static unsigned long long *g_dbg_buf = nullptr;
constexpr size_t DBG_BYTES = 4 * 8ull * 65536ull * 8;     // 4 waves x 65536 blocks x 8 words ({start, end} or phase stamps)
// make EXPERIMENTS=1 builds only: GMS_DBG into g.dbg, and the stamp buffer (cleared) when any of `buf_bits` is set
void experiment_switches(BlendGrid &g, uint32_t buf_bits, hipStream_t stream)
{
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("GMS_DBG"); dbg = (GMS_EXPERIMENTS && e) ? atoi(e) : 0; }
    g.dbg = (uint32_t)dbg;
    g.dbg_buf = nullptr;
    if ((uint32_t)dbg & buf_bits) {
        if (!g_dbg_buf) (void)hipMalloc((void **)&g_dbg_buf, DBG_BYTES);
        if (g_dbg_buf) { (void)hipMemsetAsync(g_dbg_buf, 0, DBG_BYTES, stream); g.dbg_buf = g_dbg_buf; }
    }
}
int32_t launch_blend_forward(const BlendGrid &g_in, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream)
{
    // two-phase transmittance products pay off when tiles are deep on average (> 2 segments per tile over the
    // whole image); shallow scenes take one launch
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("GMS_DBG"); dbg = (GMS_EXPERIMENTS && e) ? atoi(e) : 0; }
    BlendGrid g = g_in;
    g.dbg = (uint32_t)dbg;
    g.dbg_buf = nullptr;
    if (dbg & (128 | 256)) {
        if (!g_dbg_buf) (void)hipMalloc((void **)&g_dbg_buf, DBG_BYTES);
        if (g_dbg_buf) { (void)hipMemsetAsync(g_dbg_buf, 0, DBG_BYTES, stream); g.dbg_buf = g_dbg_buf; }
    }
    static int deep_env = -2;
    if (deep_env == -2) { const char *e = getenv("GMS_DEEP"); deep_env = e ? atoi(e) : -1; }
    const bool deep = deep_env >= 0 ? deep_env != 0 : g.capacity > 512ull * (uint64_t)g.T;
    const unsigned blocks = blend_grid_units(max_units);
    static int trip = -1;
    if (trip < 0) { const char *e = getenv("GMS_TRIP"); trip = e ? atoi(e) : 4; }
    static int wpb = -1;
    if (wpb < 0) { const char *e = getenv("GMS_FWD_WPB"); wpb = (e && atoi(e) == 4) ? 4 : 1; }
    auto head = wpb == 4 ? (trip == 2 ? blend_head_kernel<2, 4> : blend_head_kernel<4, 4>)
                         : (trip == 2 ? blend_head_kernel<2, 1> : blend_head_kernel<4, 1>);
    auto fwd2 = wpb == 4 ? (trip == 2 ? blend_fwd_kernel<2, 4> : blend_fwd_kernel<4, 4>)
                         : (trip == 2 ? blend_fwd_kernel<2, 1> : blend_fwd_kernel<4, 1>);
    const unsigned wblocks = wpb == 4 ? blocks : 4u * blocks, wthreads = wpb == 4 ? BLOCK : WAVE;
    if (deep) {     // deep scene: head segments, tile-dead check, then the tail segments of the tiles still alive
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<wblocks, wthreads, 0, stream>>>(g, o, 0));
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, blend_tloc_check_kernel<<<(unsigned)g.T, BLOCK, 0, stream>>>(g));
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<wblocks, wthreads, 0, stream>>>(g, o, 1));
    } else {
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<wblocks, wthreads, 0, stream>>>(g, o, -1));
    }
    GMS_KERNEL_CHECK(debug, stream, "blend_head");
    GMS_LAUNCH(GMS_K_BLEND_FWD, stream, fwd2<<<wblocks, wthreads, 0, stream>>>(g, o));
    GMS_KERNEL_CHECK(debug, stream, "blend_fwd");
    GMS_LAUNCH(GMS_K_BLEND_FINALIZE, stream, blend_finalize_kernel<<<(unsigned)g.T, BLOCK, 0, stream>>>(g, o));
    GMS_KERNEL_CHECK(debug, stream, "blend_finalize");
    return GMS_OK;
}

int32_t launch_blend_backward(const BlendGrid &g_in, const BlendBwdArgs &a, uint32_t max_units, bool debug, hipStream_t stream)
{
    BlendGrid g = g_in;
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("GMS_DBG"); dbg = (GMS_EXPERIMENTS && e) ? atoi(e) : 0; }
    g.dbg = (uint32_t)dbg;
    g.dbg_buf = nullptr;
    if (dbg & 16) {
        if (!g_dbg_buf) (void)hipMalloc((void **)&g_dbg_buf, DBG_BYTES);
        if (g_dbg_buf) { (void)hipMemsetAsync(g_dbg_buf, 0, DBG_BYTES, stream); g.dbg_buf = g_dbg_buf; }
    }
    const unsigned blocks = blend_grid_units(max_units);
    static int trip = -1;
    if (trip < 0) { const char *e = getenv("GMS_TRIP_BWD"); trip = e ? atoi(e) : 4; }
    const bool invd = a.has_invd && a.dL_dinvd;
    static int wpb = -1;
    if (wpb < 0) { const char *e = getenv("GMS_BWD_WPB"); wpb = (e && atoi(e) == 4) ? 4 : 1; }
    if (a.part) {                           // deterministic mode (gmsplat.h): stores into per-(instance, quadrant) partial records
        if (invd) GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (blend_bwd_kernel<true, 4, 1, 0, true><<<4u * blocks, WAVE, 0, stream>>>(g, a)));
        else GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (blend_bwd_kernel<false, 4, 1, 0, true><<<4u * blocks, WAVE, 0, stream>>>(g, a)));
    } else if (fault_mode() == 2 && !invd) {       // negative control (gms_set_fault): its own instantiation
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (blend_bwd_kernel<false, 4, 1, 2><<<4u * blocks, WAVE, 0, stream>>>(g, a)));
    } else if (wpb == 4) {
        auto kern = trip == 4 ? (invd ? blend_bwd_kernel<true, 4, 4> : blend_bwd_kernel<false, 4, 4>)
                              : (invd ? blend_bwd_kernel<true, 2, 4> : blend_bwd_kernel<false, 2, 4>);
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<blocks, BLOCK, 0, stream>>>(g, a));
    } else {
        auto kern = trip == 4 ? (invd ? blend_bwd_kernel<true, 4, 1> : blend_bwd_kernel<false, 4, 1>)
                              : (invd ? blend_bwd_kernel<true, 2, 1> : blend_bwd_kernel<false, 2, 1>);
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<4u * blocks, WAVE, 0, stream>>>(g, a));
    }
    GMS_KERNEL_CHECK(debug, stream, "blend_bwd");
    return GMS_OK;
}

}  // namespace gms

extern "C" int gms_debug_read(unsigned long long *host, size_t bytes)
{
    if (!gms::g_dbg_buf) return -1;
    return (int)hipMemcpy(host, gms::g_dbg_buf, bytes < gms::DBG_BYTES ? bytes : gms::DBG_BYTES, hipMemcpyDeviceToHost);
}
