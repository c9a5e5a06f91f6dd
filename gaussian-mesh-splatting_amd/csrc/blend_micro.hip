// blend_micro.hip -- micro-tile compositing for gfx950: the wave works on four 4x4 pixel blocks at once, one per DPP row.
//
// Why.  Mesh-bound splats are small (about 14 pixels with alpha >= 1/255 per (Gaussian, tile) instance on the headline
// scene).  A wave that owns an 8x8 quadrant and walks the quadrant's culled list one splat at a time (blend.hip) has 12 of
// its 64 lanes busy and pays a 64-lane reduction per splat.  Here the unit of work is a 4x4 pixel BLOCK with its own
// pre-filtered list: a wave carries four blocks, one per 16-lane row, each row walking ITS OWN list, so one trip of the
// wave advances four (block, splat) pairs (0.6 trips per instance instead of 1.19, 39 % of the lanes busy instead of 19 %)
// and the ten gradient sums of a splat are reduced inside a 16-lane row with row-local DPP only (29 VALU for four splats
// instead of 26 for one).  Measured on the headline scene (round 3, MI355X): backward 194 -> 142 us, forward compositing
// (head + fwd + finalize) 140 -> 121 us including the filter.
//
//   filter         (unit_setup<true>, inside the first launch that touches a unit -- it used to be a kernel of its own, 25 us
//                  for a microsecond of work per block): each thread takes one entry of the unit (tile, segment of <= L <= 256
//                  list entries), gathers its splat record, finds which of the tile's sixteen 4x4 blocks its {alpha >= 1/255}
//                  ellipse touches (exact ellipse-vs-band intervals, conservative: it can only drop pairs every pixel of
//                  the block would skip) and the survivors' entry indices (one byte each) are written per block in list
//                  order -- ballot ranks, no atomics.  The cull is paid once per frame, not once per pass.
//   micro_head / micro_fwd / micro_finalize / micro_bwd
//                  the segment-parallel scheme of blend.hip unchanged -- first segments walked exactly, transmittance
//                  products of the middle segments, exact walk of segments 1.. from the prefix product, partial sums in
//                  order, backward restarted at segment boundaries -- with one 256-thread block per unit: the unit's
//                  sixteen blocks are ordered by list length and dealt four to a wave (rows of similar length), the unit's
//                  Gaussian ids sit in LDS, each row keeps a 16-entry queue of records there.  No cull, no ballot loop: a
//                  row's queue holds only entries that hit its block.
//   backward sums  go to an LDS table indexed by the splat's entry in the unit (ds_add_f32) and leave the block as ONE set of
//                  global atomics per (unit, entry): 5.3 M global float atomics per frame instead of the 16.6 M a per-(block,
//                  splat) atomic would issue (at ~120 G float atomics/s the L2 sustains, those alone would take 140 us).
//
// Positions.  n_contrib holds, per pixel, seg * L + (index in the block's list of that segment) + 1 of the last splat
// applied: monotone along the block's concatenated lists, which is all the backward needs.
//
// Measured and not kept (same scene; git history holds the code): independent rows -- (unit, block) pieces counting-sorted
// by length over the whole frame, four consecutive pieces per wave (0.58 trips per instance, 94 us of backward without its
// atomics) -- loses the unit-level LDS table (16.6 M global atomics: 144 us) and needs two planning launches; a region key
// in that sort for XCD-local gathers (slower: imbalance between XCDs); software-pipelined queue fills and deferred atomics
// (no gain: the gather latency is already covered by the resident waves); fixed-quadrant rows instead of sorted ones (+6 us);
// part of the ten sums by global atomics and part through the LDS table (+4 us); every non-first segment composited locally
// from T = 1 and walked again only where a pixel can stop inside it (no gain: on the headline scene most later segments
// hold a stopping pixel); one gradient table per WAVE with a plain LDS read-add-write instead of ds_add_f32 (same-entry rows
// inside one instruction are rare -- 2.8 % of the trips, tools/row_collisions.py -- and were sent to the atomic): 191 us
// against 177 us for the atomic kernel padded to the same 52 KB of LDS (3 blocks per CU), 143 us at its own 25 KB (6 per CU);
// 8-entry instead of 16-entry queues (18.9 KB, 8 blocks per CU): 144.7 against 142 us, the refills double.
// Round 4, the trip loop software-pipelined over its LDS traffic (reads of trip t+1 issued before the recurrence of trip t and
// consumed after it; the ds_adds deferred by one trip so that the single lgkmcnt wait of an iteration covers only operations
// issued a whole trip earlier -- the compiler branches around every exec-masked DS instruction and must then assume at the join
// that it was skipped, so any later wait for an older read is lgkmcnt(0)): 166 us at two entries per trip (92 VGPRs: 5 waves
// per SIMD instead of 6), 147 us at one entry per trip (154 unpipelined) against 143 us for this kernel; with unconditional
// 64-lane ds_adds on dummy targets (exact wait counts) 247 us.  Timing experiments of the same round (wrong results): a plain
// LDS store instead of the ds_add 118 us, no LDS write at all 136 us, no global atomics in the flush 140 us.
// The backward is bound by dependent latency, not by LDS atomic throughput: blocks per CU 3 / 4 / 5 / 6 -> 177 / 156 / 147 /
// 143 us (unused dynamic LDS as the only change), VALU issue 0.28 of peak, 57 % of the wave cycles in s_waitcnt.
#include <stdlib.h>

#include "gms_common.h"
#include "gms_blend.h"

namespace gms {

constexpr int QROW = 17;          // LDS queue slots per row (16 used): 17 x 12 dwords staggers the four rows over the banks
constexpr int QSLOTS = 4 * QROW;

struct MPix {
    int xi, yi, tid, b; bool inside; float xf, yf;
};

// pixel `li` of 4x4 block `b` of a tile; index of the pixel in the per-(unit, pixel) segment state = b * 16 + li
__device__ __forceinline__ MPix micro_pixel(const BlendGrid &g, int tx, int ty, int b, int li)
{
    MPix p;
    p.b = b;
    p.xi = tx * TILE + (b & 3) * 4 + (li & 3);
    p.yi = ty * TILE + (b >> 2) * 4 + (li >> 2);
    p.inside = p.xi < g.W && p.yi < g.H;
    p.xf = (float)p.xi; p.yf = (float)p.yi;
    p.tid = b * 16 + li;
    return p;
}

__device__ __forceinline__ uint32_t max4rows(uint32_t v)          // v is row-uniform
{
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

// What the four waves of a unit's block share: the Gaussian ids of the unit's entries (a block's list holds entry indices,
// one byte each) and the unit's sixteen blocks ordered by list length, longest first -- wave q takes blocks order[4q .. 4q+3],
// so the four rows of a wave walk lists of similar length (0.6 wave trips per instance instead of the 0.81 of a fixed
// quadrant) and the block's longest wave is known to be wave 0.
constexpr int LMAX = 256;          // micro mode: segment length <= 256 (entry index in a byte)
struct UnitShared {
    uint32_t uid[LMAX];
    uint32_t order[16], ocnt[16];
};
// ------------------------------------------------------------------------------------ filter
// (block_mask, the exact ellipse-vs-band test of a splat against the tile's sixteen 4x4 blocks, lives in gms_blend.h: the test
// hooks run the same device function on adversarial records.)
// What every unit block does first.  FILTER = this launch is the first to touch the unit: each thread takes one entry,
// gathers its splat record, finds the 4x4 blocks it can reach (block_mask) and the survivors' entry indices are written per
// block in list order -- ballot ranks inside a wave, wave bases after ONE barrier, no atomics -- to the unit's byte lists in
// global memory, where the later launches (second forward launch, backward) find them.  Otherwise the counts are read back.
// Either way: the unit's Gaussian ids into LDS and its sixteen blocks ordered by list length.
// (The cull is paid once per frame -- not once per pass -- and costs no launch of its own: the filter used to be a kernel,
// 25 us on the headline scene for work that takes a block about a microsecond.)
template <bool FILTER>
__device__ __forceinline__ void unit_setup(const BlendGrid &g, const Unit &u, UnitShared &S, const SplatRec *rec)
{
    __shared__ uint32_t wcnt[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cn = u.end - u.beg;
    uint32_t id = 0, mask = 0;
    if ((uint32_t)tid < cn) { id = (uint32_t)g.keys[u.beg + tid]; S.uid[tid] = id; }
    if (FILTER) {
        if ((uint32_t)tid < cn) mask = block_mask(rec[id], (float)(u.tx * TILE), (float)(u.ty * TILE));
        uint32_t mycnt = 0;                        // lane b < 16: hits of block b among this wave's 64 entries
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint32_t n = (uint32_t)__builtin_popcountll(__ballot((mask >> b) & 1u));
            if (lane == b) mycnt = n;
        }
        if (lane < 16) wcnt[wave][lane] = mycnt;
        __syncthreads();
        uint8_t *out = reinterpret_cast<uint8_t *>(g.mlist) + (size_t)16 * u.beg;    // entry indices within the unit, one byte each
        const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint64_t bal = __ballot((mask >> b) & 1u);
            if ((mask >> b) & 1u) {
                uint32_t base = 0;
                for (int w = 0; w < wave; w++) base += wcnt[w][b];
                out[(size_t)b * cn + base + (uint32_t)__builtin_popcountll(bal & lt)] = (uint8_t)tid;
            }
        }
    }
    if (tid < 16) {
        uint32_t c;
        if (FILTER) { c = wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid]; g.mcount[(size_t)u.idx * 16 + tid] = c; }
        else c = g.mcount[(size_t)u.idx * 16 + tid];
        uint32_t rank = 0;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0++) {
            const uint32_t cs = (uint32_t)__shfl((int)c, s0);
            rank += (cs > c || (cs == c && s0 < tid)) ? 1u : 0u;
        }
        S.order[rank] = (uint32_t)tid; S.ocnt[rank] = c;
    }
    __threadfence_block();          // the lists just written are read back by this block's other waves
    __syncthreads();
}

// ------------------------------------------------------------------------------------ queue
// Each row keeps 16 entries of its own list in LDS; lane i of row r gathers entry j0 + i of the row's list.
__device__ __forceinline__ void queue_clear(SplatRec *recs)
{
    const int lane = threadIdx.x & 63;
    SplatRec z;
    z.q0 = make_float4(0.f, 0.f, 0.f, 0.f); z.q1 = z.q0; z.q2 = z.q0;
    recs[lane] = z;
    if (lane < QSLOTS - WAVE) recs[WAVE + lane] = z;
}

// ------------------------------------------------------------------------------------ tloc
template <int NE>
__device__ __forceinline__ void micro_tloc_unit(const BlendGrid &g, const SplatRec *rec, const Unit &u, const UnitShared &S, SplatRec *recs, int phase, int q)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    float *dst = g.seg_state + (size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + p.tid;
    if (phase == 1 && g.tile_dead[u.tile]) { *dst = 0.f; return; }
    const uint32_t cn = u.end - u.beg;
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *ml = reinterpret_cast<const uint8_t *>(g.mlist) + (size_t)16 * u.beg + (size_t)p.b * cn;
    const uint32_t maxcnt = max4rows(cnt);
    queue_clear(recs);
    float Tl = 1.f;
    for (uint32_t j0 = 0; j0 < maxcnt; j0 += 16) {
        // once a pixel's segment product is below 1e-4 every later segment starts dead whatever the exact value
        if (__all(Tl < T_MIN || !p.inside || j0 >= cnt)) break;
        wave_sync();
        if (j0 + li < cnt) recs[row * QROW + li] = rec[S.uid[ml[j0 + li]]];
        wave_sync();
        const int nt = (int)min(16u, maxcnt - j0);
        for (int t = 0; t < nt; t += NE) {
            float al[NE], pw[NE]; bool val[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const SplatRec *s = recs + row * QROW + t + e;
                const float4 r0 = s->q0, r1 = s->q1;
                const float dx = r0.x - p.xf, dy = r0.y - p.yf;
                val[e] = j0 + t + e < cnt;
                pw[e] = pair_power(r0.z, r0.w, r1.x, dx, dy);
                al[e] = fminf(ALPHA_MAX, r1.y * __expf(pw[e]));
            }
#pragma unroll
            for (int e = 0; e < NE; e++)
                if (val[e] && pw[e] <= 0.f && al[e] >= ALPHA_MIN) Tl *= (1.f - al[e]);
        }
    }
    *dst = Tl;
}

// tile_dead[t] = 1 when the product of the first tloc_head(L) segment transmittances is < 1e-4 for every pixel
__global__ void __launch_bounds__(BLOCK) micro_tloc_check_kernel(BlendGrid g)
{
    const int tile = blockIdx.x;
    const int nseg = (int)(g.unit_first[tile + 1] - g.unit_first[tile]);
    const int nhead = tloc_head(g.scan_out[3]);
    if (nseg <= nhead + 1) return;                 // no phase-1 segment exists (the last one needs no product)
    if ((uint64_t)g.tile_offset[tile + 1] > g.capacity) return;
    const int tid = threadIdx.x;
    const MPix p = micro_pixel(g, tile % g.gx, tile / g.gx, tid >> 4, tid & 15);
    const float *st0 = g.seg_state + (size_t)g.mseg_first[tile] * SEG_FLOATS;
    float T = 1.f;
    for (int k = 0; k < nhead; k++) T *= st0[(size_t)k * SEG_FLOATS + SEG_TLOC * TILE_PIX + tid];
    const int dead = __syncthreads_and(T < T_MIN || !p.inside);
    if (tid == 0) g.tile_dead[tile] = dead ? 1u : 0u;
}

// ------------------------------------------------------------------------------------ fwd
template <int NE>
__device__ __forceinline__ void micro_fwd_unit(const BlendGrid &g, const BlendFwdOut &o, const Unit &u, const UnitShared &S, SplatRec *recs, int q)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    const uint32_t cn = u.end - u.beg;
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *ml = reinterpret_cast<const uint8_t *>(g.mlist) + (size_t)16 * u.beg + (size_t)p.b * cn;
    const uint32_t maxcnt = max4rows(cnt);
    const uint32_t posbase = (uint32_t)u.seg * u.L;

    float T = 1.f;
    {
        // prefix product of the segments in front, four independent loads per step (same left-to-right order)
        const float *tl = g.seg_state + (size_t)u.slot0 * SEG_FLOATS + SEG_TLOC * TILE_PIX + p.tid;
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
    bool done = !p.inside || dead_on_entry;
    queue_clear(recs);

    for (uint32_t j0 = 0; j0 < maxcnt; j0 += 16) {
        if (__all(done || j0 >= cnt)) break;
        wave_sync();
        if (j0 + li < cnt) recs[row * QROW + li] = o.rec[S.uid[ml[j0 + li]]];
        wave_sync();
        const int nt = (int)min(16u, maxcnt - j0);
        for (int t = 0; t < nt; t += NE) {
            // NE entries of every row per trip: independent alpha evaluations, sequential compositing
            float al[NE], pw[NE]; bool val[NE]; float4 r1[NE], r2[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const SplatRec *s = recs + row * QROW + t + e;
                const float4 r0 = s->q0;
                r1[e] = s->q1; r2[e] = s->q2;
                const float dx = r0.x - p.xf, dy = r0.y - p.yf;
                val[e] = j0 + t + e < cnt;
                pw[e] = pair_power(r0.z, r0.w, r1[e].x, dx, dy);
                al[e] = fminf(ALPHA_MAX, r1[e].y * __expf(pw[e]));
            }
#pragma unroll
            for (int e = 0; e < NE; e++) {
                bool act = val[e] && !done && pw[e] <= 0.f && al[e] >= ALPHA_MIN;
                const float testT = T * (1.f - al[e]);
                if (act && testT < T_MIN) { done = true; act = false; }
                if (act) {
                    const float w = al[e] * T;
                    C0 += r1[e].z * w; C1 += r1[e].w * w; C2 += r2[e].x * w;
                    Dp += r2[e].y * w;
                    T = testT;
                    last = posbase + j0 + (uint32_t)(t + e) + 1u;
                }
            }
            if (__all(done || j0 + t + NE >= cnt)) break;
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
        st[SEG_C0 * TILE_PIX + p.tid] = C0; st[SEG_C1 * TILE_PIX + p.tid] = C1; st[SEG_C2 * TILE_PIX + p.tid] = C2;
        st[SEG_D * TILE_PIX + p.tid] = Dp;
        st[SEG_TEND * TILE_PIX + p.tid] = dead_on_entry ? -1.f : T;
        st[SEG_LAST * TILE_PIX + p.tid] = __uint_as_float(last);
        // the first segment's exact walk doubles as its transmittance product (see blend.hip)
        if (u.seg == 0) st[SEG_TLOC * TILE_PIX + p.tid] = done ? 0.f : T;
    }
}

// First launch: every unit that depends on nothing -- the exact walk of each tile's FIRST segment (single-segment tiles are
// finished by it) and, for the middle segments of multi-segment tiles, the transmittance products.  One block = one unit.
template <int NE>
__global__ void __launch_bounds__(BLOCK) micro_head_kernel(BlendGrid g, BlendFwdOut o, int phase)
{
    __shared__ SplatRec recs[4][QSLOTS];
    __shared__ UnitShared S;
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    // (block-uniform decisions first: a unit with nothing to do in this launch leaves before the setup barrier)
    const bool walk = u.seg == 0 ? phase <= 0
                                 : (u.nseg > 1 && u.seg != u.nseg - 1 && (phase < 0 || (u.seg < tloc_head(u.L)) == (phase == 0)));
    if (!walk) return;
    if (u.seg > 0 && phase == 1 && g.tile_dead[u.tile]) {          // products of a dead tile: nothing to walk, empty lists on record
        if (threadIdx.x < 16) g.mcount[(size_t)u.idx * 16 + threadIdx.x] = 0u;
        g.seg_state[(size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + threadIdx.x] = 0.f;
        return;
    }
    unit_setup<true>(g, u, S, o.rec);
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);      // (rotate the sorted groups over the block's waves)
    if (u.seg == 0) micro_fwd_unit<NE>(g, o, u, S, recs[q], q);
    else micro_tloc_unit<NE>(g, o.rec, u, S, recs[q], phase, q);
}

// Second launch: segments 1.. of the multi-segment tiles, from the prefix product of the segments in front.
template <int NE>
__global__ void __launch_bounds__(BLOCK) micro_fwd_kernel(BlendGrid g, BlendFwdOut o)
{
    __shared__ SplatRec recs[4][QSLOTS];
    __shared__ UnitShared S;
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    if (u.seg == 0) return;
    if (u.seg == u.nseg - 1) unit_setup<true>(g, u, S, o.rec);          // last segments are first touched here
    else unit_setup<false>(g, u, S, o.rec);                                // middle segments: filtered by the first launch
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);      // (rotate the sorted groups over the block's waves)
    micro_fwd_unit<NE>(g, o, u, S, recs[q], q);
}

// ------------------------------------------------------------------------------------ finalize
__global__ void __launch_bounds__(BLOCK) micro_finalize_kernel(BlendGrid g, BlendFwdOut o)
{
    const int tile = blockIdx.x;
    const uint32_t first = g.unit_first[tile];
    const int nseg = (int)(g.unit_first[tile + 1] - first);
    if (nseg <= 1) return;
    if ((uint64_t)g.tile_offset[tile + 1] > g.capacity) return;
    const int tid = threadIdx.x;
    const MPix p = micro_pixel(g, tile % g.gx, tile / g.gx, tid >> 4, tid & 15);
    float *st0 = g.seg_state + (size_t)g.mseg_first[tile] * SEG_FLOATS;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, T = 1.f;
    uint32_t last = 0;
    constexpr int U = 4;          // four segments per step, every load issued before the first use
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
    if (p.inside) {
        const size_t pid = (size_t)p.yi * g.W + p.xi, HW = (size_t)g.W * g.H;
        o.final_T[pid] = T;
        o.n_contrib[pid] = last;
        o.out_color[pid] = C0 + T * o.bg[0];
        o.out_color[HW + pid] = C1 + T * o.bg[1];
        o.out_color[2 * HW + pid] = C2 + T * o.bg[2];
        o.out_invdepth[pid] = Dp;
    }
}

// ------------------------------------------------------------------------------------ bwd
// Phase stamps of one wave (make EXPERIMENTS=1, GMS_DBG & 1024; tools/micro_phases.py): wall clock (100 MHz) at
// start / after the staging / before the walk / after the walk / after the block barrier / at the end, and the wave's trips.
struct Phases {
    unsigned long long *buf; bool on;
    __device__ __forceinline__ Phases(const BlendGrid &g) : buf(nullptr), on(false)
    {
        if (dbg_on(g, 1024u) && g.dbg_buf) {
            buf = g.dbg_buf + 8ull * 65536ull * (threadIdx.x >> 6) + 8ull * (blockIdx.x & 65535u);
            on = (threadIdx.x & 63) == 0;
        }
    }
    __device__ __forceinline__ void mark(int k) { if (on) buf[k] = wall_clock64(); }
    __device__ __forceinline__ void value(int k, unsigned long long v) { if (on) buf[k] = v; }
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// Transposing reduction of ten values over the 16 lanes of every DPP row (four rows = four splats at once): each stage adds
// partner lanes AND halves the number of live registers, 29 VALU in all.  Partners: lane ^ 8 (row_ror:8), 7 - lane within
// the half row (row_half_mirror), lane ^ 2, lane ^ 1 (quad_perm).  Result: lane i of the row holds the row total of
//   i = 0: v0   8: v5   4: v3   12: v8   2: v1   10: v6   6: v4   14: v9   odd i < 8: v2   odd i > 8: v7
__device__ __forceinline__ float row_reduce10(const float *v, bool b3, bool b2, bool b1, bool b0)
{
    float u[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float keep = b3 ? v[k + 5] : v[k], send = b3 ? v[k] : v[k + 5];
        u[k] = keep + dpp_mov<0x128>(send);
    }
    const float w0 = (b2 ? u[3] : u[0]) + dpp_mov<0x141>(b2 ? u[0] : u[3]);
    const float w1 = (b2 ? u[4] : u[1]) + dpp_mov<0x141>(b2 ? u[1] : u[4]);
    const float w2 = u[2] + dpp_mov<0x141>(u[2]);
    const float x0 = (b1 ? w1 : w0) + dpp_mov<0x4E>(b1 ? w0 : w1);
    const float x1 = w2 + dpp_mov<0x4E>(w2);
    return (b0 ? x1 : x0) + dpp_mov<0xB1>(b0 ? x0 : x1);
}

// One block = one unit.  The ten sums of every (row, splat) pair go into an LDS table indexed by the splat's entry in the unit
// (ds_add_f32: 16.6 M LDS adds per frame on the headline scene); when the four waves are done the table is flushed with ONE
// set of global atomics per (unit, entry): a splat that lies in several 4x4 blocks of the tile costs ten global atomics, not
// ten per block (5.3 M instead of 16.6 M, and far fewer waves hammering the same gradient record at the same time).
// DET (deterministic mode, gmsplat.h): one table per WAVE (no cross-wave adds), the four rows of a wave add one after the other
// (two rows of one instruction can hold the same entry), and the flush sums the four tables in wave-group order into ONE partial
// record per instance, stored -- not added -- at the instance's position in the sorted list: no float atomic, fixed order.
template <bool INVD, int NE, int FAULT, bool DET = false>
__global__ void __launch_bounds__(BLOCK) micro_bwd_kernel(BlendGrid g, BlendBwdArgs a)
{
    __shared__ SplatRec recs_all[4][QSLOTS];
    __shared__ uint32_t eid_all[4][QSLOTS];            // entry index (within the unit) of every queue slot
    __shared__ float table_all[(DET ? 4 : 1) * LMAX * 10];
    __shared__ UnitShared S;
    Phases ph(g);
    ph.mark(0);
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    if (u.end <= u.beg) return;
    for (int k = threadIdx.x; k < (DET ? 4 : 1) * LMAX * 10; k += BLOCK) table_all[k] = 0.f;
    unit_setup<false>(g, u, S, a.rec);                  // (its barrier also orders the table clear)
    ph.mark(1);
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);      // (rotate the sorted groups over the block's waves)
    float *const table = table_all + (DET ? q * LMAX * 10 : 0);
    SplatRec *recs = recs_all[q];
    uint32_t *eid = eid_all[q];
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    const size_t HW = (size_t)g.W * g.H;
    const size_t pid = (size_t)p.yi * g.W + p.xi;
    const float Tfinal = p.inside ? a.final_T[pid] : 0.f;
    const uint32_t last = p.inside ? a.n_contrib[pid] : 0u;      // seg * L + index + 1 of the last splat this pixel applied
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dinvd = 0.f;
    if (p.inside) {
        dp0 = a.dL_dpix[pid]; dp1 = a.dL_dpix[HW + pid]; dp2 = a.dL_dpix[2 * HW + pid];
        if (INVD) dinvd = a.dL_dinvd[pid];
    }
    const float Tfinal_bgdot = Tfinal * (a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2);
    const uint32_t posbase = (uint32_t)u.seg * u.L;
    const uint32_t cn = u.end - u.beg;
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *ml = reinterpret_cast<const uint8_t *>(g.mlist) + (size_t)16 * u.beg + (size_t)p.b * cn;
    // entries [0, lrel) of this block's list of this segment were composited by this pixel
    const uint32_t lrel = last > posbase ? min(last - posbase, cnt) : 0u;
    uint32_t top = lrel;                                    // furthest entry any pixel of the row composited
    top = max(top, (uint32_t)__shfl_xor((int)top, 8)); top = max(top, (uint32_t)__shfl_xor((int)top, 4));
    top = max(top, (uint32_t)__shfl_xor((int)top, 2)); top = max(top, (uint32_t)__shfl_xor((int)top, 1));
    const uint32_t maxtop = max4rows(top);
    ph.value(6, maxtop);

    if (maxtop > 0) {          // (wave-uniform; a wave with nothing to walk goes straight to the flush barrier)
    BwdState st8 = {Tfinal, 0.f, 0.f, 0.f, 0.f};
    if (u.nseg > 1) {
        const float *st = g.seg_state + (size_t)(u.slot0 + u.seg) * SEG_FLOATS;
        const float te = st[SEG_TEND * TILE_PIX + p.tid];
        if (te > 0.f) {
            // restart of the recurrence at the segment boundary: T after this segment's last applied splat and the colour
            // composited behind it (sum of the live partials of the later segments) divided by that T
            st8.T = te;
            float S0 = 0.f, S1 = 0.f, S2 = 0.f, SD = 0.f;
            bool stop = false;
            for (int k0 = u.seg + 1; k0 < u.nseg; k0 += 4) {
                float tk[4], c0[4], c1[4], c2[4], dd[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float *sk = g.seg_state + (size_t)(u.slot0 + min(k0 + j, u.nseg - 1)) * SEG_FLOATS;
                    tk[j] = sk[SEG_TEND * TILE_PIX + p.tid]; c0[j] = sk[SEG_C0 * TILE_PIX + p.tid];
                    c1[j] = sk[SEG_C1 * TILE_PIX + p.tid]; c2[j] = sk[SEG_C2 * TILE_PIX + p.tid];
                    dd[j] = INVD ? sk[SEG_D * TILE_PIX + p.tid] : 0.f;
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

    // lane -> field of the 64-byte gradient record (the layout row_reduce10 leaves)
    const bool b3 = (li & 8) != 0, b2 = (li & 4) != 0, b1 = (li & 2) != 0, b0 = (li & 1) != 0;
    int afield;
    switch (li) {
    case 0: afield = GRAD_MX; break;
    case 8: afield = GRAD_OP; break;
    case 4: afield = GRAD_CB; break;
    case 12: afield = GRAD_B; break;
    case 2: afield = GRAD_MY; break;
    case 10: afield = GRAD_R; break;
    case 6: afield = GRAD_CC; break;
    case 14: afield = GRAD_ID; break;
    case 1: afield = GRAD_CA; break;
    default: afield = GRAD_G; break;       // lane 9
    }
    const bool alane = (li & 1) == 0 ? (li != 14 || INVD) : (li == 1 || li == 9);
    queue_clear(recs);
    eid[lane] = 0u;
    if (lane < QSLOTS - WAVE) eid[WAVE + lane] = 0u;
    ph.mark(2);

    // back to front: global trip t0 handles entry top - 1 - t0 of every row's list (the rows are aligned at their tops)
    for (uint32_t g0 = 0; g0 < maxtop; g0 += 16) {
        wave_sync();
        if (g0 + li < top) {
            const uint32_t e = ml[top - 1u - (g0 + li)];
            eid[row * QROW + li] = e;
            recs[row * QROW + li] = a.rec[S.uid[e]];
        }
        wave_sync();
        const int nt = (int)min(16u, maxtop - g0);
        for (int t = 0; t < nt; t += NE) {
            bool act[NE]; float dx[NE], dy[NE], G[NE], al[NE]; float4 r1[NE], r2[NE]; uint32_t se[NE];
            bool anyact = false;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const SplatRec *sr = recs + row * QROW + t + e;
                const float4 r0 = sr->q0;
                r1[e] = sr->q1; r2[e] = sr->q2;
                se[e] = eid[row * QROW + t + e];
                dx[e] = r0.x - p.xf; dy[e] = r0.y - p.yf;
                const float pw = pair_power(r0.z, r0.w, r1[e].x, dx[e], dy[e]);
                G[e] = __expf(pw);
                al[e] = fminf(ALPHA_MAX, r1[e].y * G[e]);
                const uint32_t trip = g0 + (uint32_t)(t + e);
                // entry index top - 1 - trip of the row's list; composited by this pixel iff it lies below lrel
                act[e] = trip < top && (top - 1u - trip) < lrel && pw <= 0.f && al[e] >= ALPHA_MIN;
                anyact = anyact || act[e];
            }
            if (!__any(anyact)) continue;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                float v[10];
                bwd_step<INVD>(st8, act[e], r1[e], r2[e], dx[e], dy[e], G[e], al[e], dp0, dp1, dp2, dinvd, Tfinal_bgdot, v);
                const float y = row_reduce10(v, b3, b2, b1, b0);
                // a row with no active pixel for this entry sums exact zeros: nothing to add (and its slot may be stale)
                if (DET) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (row == r && alane && y != 0.f) atomicAdd(&table[se[e] * 10u + (uint32_t)afield], y);
                        asm volatile("" ::: "memory");          // four separate LDS instructions, in row order
                    }
                } else if (alane && y != 0.f) atomicAdd(&table[se[e] * 10u + (uint32_t)afield], y);
            }
        }
    }
    }
    ph.mark(3);
    __syncthreads();
    ph.mark(4);
    // flush: sixteen entries per step, ten lanes per entry on the ten fields of its 64-byte record (one cache line)
    {
        const int f = threadIdx.x & 15;
        for (uint32_t e = threadIdx.x >> 4; e < cn; e += BLOCK / 16) {
            if (DET) {
                // every instance of the unit gets its record (zeros included: the buffer is not cleared between frames)
                const uint32_t k = e * 10u + (uint32_t)f;
                const float y = f < 10 ? ((table_all[k] + table_all[LMAX * 10 + k]) + table_all[2 * LMAX * 10 + k]) + table_all[3 * LMAX * 10 + k] : 0.f;
                a.part[(size_t)(u.beg + e) * GRAD_STRIDE + f] = y;
            } else if (f < 10) {
                const float y = table[e * 10u + (uint32_t)f];
                if (y != 0.f) unsafeAtomicAdd(a.accum + (size_t)S.uid[e] * GRAD_STRIDE + f, y);
            }
        }
    }
    ph.mark(5);
}

// ==================================================================================== resident-unit kernels (round 5)
// The same micro-tile scheme with the unit's splat records RESIDENT in LDS.  The queue kernels above re-gather a record from
// L2 once per 4x4 block that lists it (2.15 gathers per instance) through a dependent chain -- list byte (global) -> Gaussian id
// (LDS) -> 48-byte record (global) -> LDS queue, two wave_syncs -- every 16 trips of every row.  Here the block's 256 threads
// gather the unit's <= 256 records ONCE, coalesced by entry, into LDS (40 bytes each: the extents are only needed by the
// filter), the sixteen byte lists are rebuilt in LDS from a 16-bit block mask per instance (computed by the launch that first
// touches the unit, 2 bytes per instance in global memory instead of the 16-byte-per-instance list area), and a row's trip reads
// `list[block][pos]` -> `record[entry]` straight from LDS: no queue, no refill, no global access inside the walks.
struct UnitRecs {
    float4 ra[LMAX];           // pix.x, pix.y, conic A, conic B
    float4 rb[LMAX];           // conic C, opacity', r, g
    float2 rc[LMAX];           // b, 1/depth
    uint8_t list[16][LMAX];    // per 4x4 block: the entries that reach it, in list (depth) order
    uint32_t order[16], ocnt[16];
    uint32_t wcnt[4][16];
};

template <int NE> __device__ __forceinline__ uint32_t list_load(const uint8_t *lst, uint32_t pos);
template <> __device__ __forceinline__ uint32_t list_load<1>(const uint8_t *lst, uint32_t pos) { return lst[pos]; }
template <> __device__ __forceinline__ uint32_t list_load<2>(const uint8_t *lst, uint32_t pos) { return *reinterpret_cast<const uint16_t *>(lst + pos); }
template <> __device__ __forceinline__ uint32_t list_load<4>(const uint8_t *lst, uint32_t pos) { return *reinterpret_cast<const uint32_t *>(lst + pos); }

// Staging.  FILTER = this launch is the first to touch the unit: the thread of an entry computes the entry's block mask from the
// record it has just gathered and leaves it in global memory (g.mlist as uint16[instances]) for the later launches, which read
// it back instead -- every launch therefore builds IDENTICAL lists (n_contrib holds positions in them).
template <bool FILTER>
__device__ __forceinline__ void unit_stage(const BlendGrid &g, const Unit &u, UnitRecs &S, const SplatRec *rec, uint32_t *uid)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cn = u.end - u.beg;
    uint16_t *mm = reinterpret_cast<uint16_t *>(g.mlist);
    uint32_t mask = 0;
    if ((uint32_t)tid < cn) {
        const uint32_t id = reinterpret_cast<const uint32_t *>(g.keys)[2 * (size_t)(u.beg + tid)];      // low word of the (depth, id) key
        if (!FILTER) mask = mm[u.beg + tid];
        const SplatRec r = rec[id];
        if (FILTER) { mask = block_mask(r, (float)(u.tx * TILE), (float)(u.ty * TILE)); mm[u.beg + tid] = (uint16_t)mask; }
        S.ra[tid] = r.q0; S.rb[tid] = r.q1; S.rc[tid] = make_float2(r.q2.x, r.q2.y);
        if (uid) uid[tid] = id;
    } else {
        // a row that idles behind the end of its list reads whatever byte lies there: every record it can name must be finite
        S.ra[tid] = make_float4(0.f, 0.f, 0.f, 0.f); S.rb[tid] = make_float4(0.f, 0.f, 0.f, 0.f); S.rc[tid] = make_float2(0.f, 0.f);
    }
    uint32_t mycnt = 0;                        // lane b < 16: hits of block b among this wave's 64 entries
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const uint32_t n = (uint32_t)__builtin_popcountll(__ballot((mask >> b) & 1u));
        if (lane == b) mycnt = n;
    }
    if (lane < 16) S.wcnt[wave][lane] = mycnt;
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const uint64_t bal = __ballot((mask >> b) & 1u);
        if ((mask >> b) & 1u) {
            uint32_t base = 0;
            for (int w = 0; w < wave; w++) base += S.wcnt[w][b];
            S.list[b][base + (uint32_t)__builtin_popcountll(bal & lt)] = (uint8_t)tid;
        }
    }
    if (tid < 16) {
        const uint32_t c = S.wcnt[0][tid] + S.wcnt[1][tid] + S.wcnt[2][tid] + S.wcnt[3][tid];
        uint32_t rank = 0;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0++) {
            const uint32_t cs = (uint32_t)__shfl((int)c, s0);
            rank += (cs > c || (cs == c && s0 < tid)) ? 1u : 0u;
        }
        S.order[rank] = (uint32_t)tid; S.ocnt[rank] = c;
    }
    __syncthreads();
}

template <int NE>
__device__ __forceinline__ void ru_tloc_unit(const BlendGrid &g, const Unit &u, const UnitRecs &S, int phase, int q)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    float *dst = g.seg_state + (size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + p.tid;
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *lst = S.list[p.b];
    const uint32_t maxcnt = max4rows(cnt);
    float Tl = 1.f;
    (void)phase;
    for (uint32_t t = 0; t < maxcnt; t += NE) {
        // once a pixel's segment product is below 1e-4 every later segment starts dead whatever the exact value
        if (__all(Tl < T_MIN || !p.inside || t >= cnt)) break;
        const uint32_t ep = list_load<NE>(lst, min(t, (uint32_t)(LMAX - NE)));
        float al[NE], pw[NE]; bool val[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const uint32_t ent = (ep >> (8 * e)) & 0xffu;
            const float4 r0 = S.ra[ent];
            const float2 r1 = *reinterpret_cast<const float2 *>(&S.rb[ent]);
            const float dx = r0.x - p.xf, dy = r0.y - p.yf;
            val[e] = t + e < cnt;
            pw[e] = pair_power(r0.z, r0.w, r1.x, dx, dy);
            al[e] = fminf(ALPHA_MAX, r1.y * __expf(pw[e]));
        }
#pragma unroll
        for (int e = 0; e < NE; e++)
            if (val[e] && pw[e] <= 0.f && al[e] >= ALPHA_MIN) Tl *= (1.f - al[e]);
    }
    *dst = Tl;
}

template <int NE>
__device__ __forceinline__ void ru_fwd_unit(const BlendGrid &g, const BlendFwdOut &o, const Unit &u, const UnitRecs &S, int q)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *lst = S.list[p.b];
    const uint32_t maxcnt = max4rows(cnt);
    const uint32_t posbase = (uint32_t)u.seg * u.L;

    float T = 1.f;
    {
        // prefix product of the segments in front, four independent loads per step (same left-to-right order)
        const float *tl = g.seg_state + (size_t)u.slot0 * SEG_FLOATS + SEG_TLOC * TILE_PIX + p.tid;
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
    bool done = !p.inside || dead_on_entry;

    for (uint32_t t = 0; t < maxcnt; t += NE) {
        if (__all(done || t >= cnt)) break;
        // NE entries of every row per trip: independent alpha evaluations, sequential compositing
        const uint32_t ep = list_load<NE>(lst, min(t, (uint32_t)(LMAX - NE)));
        float al[NE], pw[NE]; bool val[NE]; float2 cg[NE], cb[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const uint32_t ent = (ep >> (8 * e)) & 0xffu;
            const float4 r0 = S.ra[ent], r1 = S.rb[ent];
            cb[e] = S.rc[ent];
            cg[e] = make_float2(r1.z, r1.w);
            const float dx = r0.x - p.xf, dy = r0.y - p.yf;
            val[e] = t + e < cnt;
            pw[e] = pair_power(r0.z, r0.w, r1.x, dx, dy);
            al[e] = fminf(ALPHA_MAX, r1.y * __expf(pw[e]));
        }
#pragma unroll
        for (int e = 0; e < NE; e++) {
            bool act = val[e] && !done && pw[e] <= 0.f && al[e] >= ALPHA_MIN;
            const float testT = T * (1.f - al[e]);
            if (act && testT < T_MIN) { done = true; act = false; }
            if (act) {
                const float w = al[e] * T;
                C0 += cg[e].x * w; C1 += cg[e].y * w; C2 += cb[e].x * w;
                Dp += cb[e].y * w;
                T = testT;
                last = posbase + t + (uint32_t)e + 1u;
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
        st[SEG_C0 * TILE_PIX + p.tid] = C0; st[SEG_C1 * TILE_PIX + p.tid] = C1; st[SEG_C2 * TILE_PIX + p.tid] = C2;
        st[SEG_D * TILE_PIX + p.tid] = Dp;
        st[SEG_TEND * TILE_PIX + p.tid] = dead_on_entry ? -1.f : T;
        st[SEG_LAST * TILE_PIX + p.tid] = __uint_as_float(last);
        // the first segment's exact walk doubles as its transmittance product (see blend.hip)
        if (u.seg == 0) st[SEG_TLOC * TILE_PIX + p.tid] = done ? 0.f : T;
    }
}

template <int NE>
__global__ void __launch_bounds__(BLOCK) ru_head_kernel(BlendGrid g, BlendFwdOut o, int phase)
{
    __shared__ UnitRecs S;
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    const bool walk = u.seg == 0 ? phase <= 0
                                 : (u.nseg > 1 && u.seg != u.nseg - 1 && (phase < 0 || (u.seg < tloc_head(u.L)) == (phase == 0)));
    if (!walk) return;
    if (u.seg > 0 && phase == 1 && g.tile_dead[u.tile]) {          // products of a dead tile: nothing to walk, empty lists on record
        if (threadIdx.x < u.end - u.beg) reinterpret_cast<uint16_t *>(g.mlist)[u.beg + threadIdx.x] = 0;
        g.seg_state[(size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + threadIdx.x] = 0.f;
        return;
    }
    unit_stage<true>(g, u, S, o.rec, nullptr);
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);      // (rotate the sorted groups over the block's waves)
    if (u.seg == 0) ru_fwd_unit<NE>(g, o, u, S, q);
    else ru_tloc_unit<NE>(g, u, S, phase, q);
}

template <int NE>
__global__ void __launch_bounds__(BLOCK) ru_fwd_kernel(BlendGrid g, BlendFwdOut o)
{
    __shared__ UnitRecs S;
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    if (u.seg == 0) return;
    if (u.seg == u.nseg - 1) unit_stage<true>(g, u, S, o.rec, nullptr);       // last segments are first touched here
    else unit_stage<false>(g, u, S, o.rec, nullptr);                          // middle segments: filtered by the first launch
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);
    ru_fwd_unit<NE>(g, o, u, S, q);
}

// Backward.  The rows of a wave are aligned at the BOTTOM of their lists: global trip position `pos` is the same list index
// for every row (rows whose list ends below it idle), so a trip's entry bytes are one aligned LDS read per NE entries.
template <bool INVD, int NE, int FAULT, bool DET = false>
__global__ void __launch_bounds__(BLOCK) ru_bwd_kernel(BlendGrid g, BlendBwdArgs a)
{
    __shared__ UnitRecs S;
    __shared__ uint32_t uid[LMAX];
    __shared__ float table_all[(DET ? 4 : 1) * LMAX * 10];
    Phases ph(g);
    ph.mark(0);
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    if (u.end <= u.beg) return;
    for (int k = threadIdx.x; k < (DET ? 4 : 1) * LMAX * 10; k += BLOCK) table_all[k] = 0.f;
    unit_stage<false>(g, u, S, a.rec, uid);                  // (its barriers also order the table clear)
    ph.mark(1);
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);
    float *const table = table_all + (DET ? q * LMAX * 10 : 0);
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    const size_t HW = (size_t)g.W * g.H;
    const size_t pid = (size_t)p.yi * g.W + p.xi;
    const float Tfinal = p.inside ? a.final_T[pid] : 0.f;
    const uint32_t last = p.inside ? a.n_contrib[pid] : 0u;      // seg * L + index + 1 of the last splat this pixel applied
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dinvd = 0.f;
    if (p.inside) {
        dp0 = a.dL_dpix[pid]; dp1 = a.dL_dpix[HW + pid]; dp2 = a.dL_dpix[2 * HW + pid];
        if (INVD) dinvd = a.dL_dinvd[pid];
    }
    const float Tfinal_bgdot = Tfinal * (a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2);
    const uint32_t posbase = (uint32_t)u.seg * u.L;
    const uint32_t cn = u.end - u.beg;
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *lst = S.list[p.b];
    // entries [0, lrel) of this block's list of this segment were composited by this pixel
    const uint32_t lrel = last > posbase ? min(last - posbase, cnt) : 0u;
    uint32_t top = lrel;                                    // furthest entry any pixel of the wave composited
    top = max(top, (uint32_t)__shfl_xor((int)top, 8)); top = max(top, (uint32_t)__shfl_xor((int)top, 4));
    top = max(top, (uint32_t)__shfl_xor((int)top, 2)); top = max(top, (uint32_t)__shfl_xor((int)top, 1));
    const uint32_t maxtop = max4rows(top);
    ph.value(6, maxtop);

    if (maxtop > 0) {          // (wave-uniform; a wave with nothing to walk goes straight to the flush barrier)
    BwdState st8 = {Tfinal, 0.f, 0.f, 0.f, 0.f};
    if (u.nseg > 1) {
        const float *st = g.seg_state + (size_t)(u.slot0 + u.seg) * SEG_FLOATS;
        const float te = st[SEG_TEND * TILE_PIX + p.tid];
        if (te > 0.f) {
            // restart of the recurrence at the segment boundary: T after this segment's last applied splat and the colour
            // composited behind it (sum of the live partials of the later segments) divided by that T
            st8.T = te;
            float S0 = 0.f, S1 = 0.f, S2 = 0.f, SD = 0.f;
            bool stop = false;
            for (int k0 = u.seg + 1; k0 < u.nseg; k0 += 4) {
                float tk[4], c0[4], c1[4], c2[4], dd[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float *sk = g.seg_state + (size_t)(u.slot0 + min(k0 + j, u.nseg - 1)) * SEG_FLOATS;
                    tk[j] = sk[SEG_TEND * TILE_PIX + p.tid]; c0[j] = sk[SEG_C0 * TILE_PIX + p.tid];
                    c1[j] = sk[SEG_C1 * TILE_PIX + p.tid]; c2[j] = sk[SEG_C2 * TILE_PIX + p.tid];
                    dd[j] = INVD ? sk[SEG_D * TILE_PIX + p.tid] : 0.f;
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
    ph.mark(2);

    // lane -> field of the 64-byte gradient record (the layout row_reduce10 leaves)
    const bool b3 = (li & 8) != 0, b2 = (li & 4) != 0, b1 = (li & 2) != 0, b0 = (li & 1) != 0;
    int afield;
    switch (li) {
    case 0: afield = GRAD_MX; break;
    case 8: afield = GRAD_OP; break;
    case 4: afield = GRAD_CB; break;
    case 12: afield = GRAD_B; break;
    case 2: afield = GRAD_MY; break;
    case 10: afield = GRAD_R; break;
    case 6: afield = GRAD_CC; break;
    case 14: afield = GRAD_ID; break;
    case 1: afield = GRAD_CA; break;
    default: afield = GRAD_G; break;       // lane 9
    }
    const bool alane = (li & 1) == 0 ? (li != 14 || INVD) : (li == 1 || li == 9);

    // back to front: the trip at list position pos handles entry pos of every row's list that reaches it
    for (int g0 = (int)(((maxtop + NE - 1u) / NE) * NE) - NE; g0 >= 0; g0 -= NE) {
        const uint32_t ep = list_load<NE>(lst, (uint32_t)g0);
        bool act[NE]; float dx[NE], dy[NE], G[NE], al[NE]; float4 r1[NE]; float2 r2[NE]; uint32_t se[NE];
        bool anyact = false;
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const uint32_t pos = (uint32_t)g0 + (uint32_t)(NE - 1 - e);          // descending within the trip
            se[e] = (ep >> (8 * (NE - 1 - e))) & 0xffu;
            const float4 r0 = S.ra[se[e]];
            r1[e] = S.rb[se[e]]; r2[e] = S.rc[se[e]];
            dx[e] = r0.x - p.xf; dy[e] = r0.y - p.yf;
            const float pw = pair_power(r0.z, r0.w, r1[e].x, dx[e], dy[e]);
            G[e] = __expf(pw);
            al[e] = fminf(ALPHA_MAX, r1[e].y * G[e]);
            // composited by this pixel iff it lies below lrel (lrel <= the length of the row's list)
            act[e] = pos < lrel && pw <= 0.f && al[e] >= ALPHA_MIN;
            anyact = anyact || act[e];
        }
        if (!__any(anyact)) continue;
#pragma unroll
        for (int e = 0; e < NE; e++) {
            float v[10];
            const float4 q2 = make_float4(r2[e].x, r2[e].y, 0.f, 0.f);
            bwd_step<INVD>(st8, act[e], r1[e], q2, dx[e], dy[e], G[e], al[e], dp0, dp1, dp2, dinvd, Tfinal_bgdot, v);
            const float y = row_reduce10(v, b3, b2, b1, b0);
            // a row with no active pixel for this entry sums exact zeros: nothing to add (and its entry byte may be stale)
            if (DET) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (row == r && alane && y != 0.f) atomicAdd(&table[se[e] * 10u + (uint32_t)afield], y);
                    asm volatile("" ::: "memory");          // four separate LDS instructions, in row order
                }
            } else if (alane && y != 0.f) atomicAdd(&table[se[e] * 10u + (uint32_t)afield], y);
        }
    }
    }
    ph.mark(3);
    __syncthreads();
    ph.mark(4);
    // flush: sixteen entries per step, ten lanes per entry on the ten fields of its 64-byte record (one cache line)
    {
        const int f = threadIdx.x & 15;
        for (uint32_t e = threadIdx.x >> 4; e < cn; e += BLOCK / 16) {
            if (DET) {
                // every instance of the unit gets its record (zeros included: the buffer is not cleared between frames)
                const uint32_t k = e * 10u + (uint32_t)f;
                const float y = f < 10 ? ((table_all[k] + table_all[LMAX * 10 + k]) + table_all[2 * LMAX * 10 + k]) + table_all[3 * LMAX * 10 + k] : 0.f;
                a.part[(size_t)(u.beg + e) * GRAD_STRIDE + f] = y;
            } else if (f < 10) {
                const float y = table[e * 10u + (uint32_t)f];
                if (y != 0.f) unsafeAtomicAdd(a.accum + (size_t)uid[e] * GRAD_STRIDE + f, y);
            }
        }
    }
    ph.mark(5);
}

// ------------------------------------------------------------------------------------ host
// GMS_MICRO_RU: 1 (default) = the resident-unit kernels, 0 = the row-queue kernels
static bool resident_units()
{
    static int ru = -1;
    if (ru < 0) { const char *e = getenv("GMS_MICRO_RU"); ru = e ? (atoi(e) != 0) : 1; }
    return ru != 0;
}

int32_t launch_micro_forward(const BlendGrid &g, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream)
{
    static int deep_env = -2;
    if (deep_env == -2) { const char *e = getenv("GMS_DEEP"); deep_env = e ? atoi(e) : -1; }
    const bool deep = deep_env >= 0 ? deep_env != 0 : g.capacity > 512ull * (uint64_t)g.T;
    const unsigned blocks = blend_grid_units(max_units);
    static int trip = -1;
    if (trip < 0) { const char *e = getenv("GMS_TRIP"); trip = e ? atoi(e) : 4; }      // (forward: 4 entries per trip; 2: -1.3 % it/s)
    const bool ru = resident_units();
    auto head = ru ? (trip == 4 ? ru_head_kernel<4> : (trip == 1 ? ru_head_kernel<1> : ru_head_kernel<2>))
                   : (trip == 4 ? micro_head_kernel<4> : (trip == 1 ? micro_head_kernel<1> : micro_head_kernel<2>));
    auto fwd2 = ru ? (trip == 4 ? ru_fwd_kernel<4> : (trip == 1 ? ru_fwd_kernel<1> : ru_fwd_kernel<2>))
                   : (trip == 4 ? micro_fwd_kernel<4> : (trip == 1 ? micro_fwd_kernel<1> : micro_fwd_kernel<2>));
    if (deep) {     // deep scene: head segments, tile-dead check, then the tail segments of the tiles still alive
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<blocks, BLOCK, 0, stream>>>(g, o, 0));
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, micro_tloc_check_kernel<<<(unsigned)g.T, BLOCK, 0, stream>>>(g));
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<blocks, BLOCK, 0, stream>>>(g, o, 1));
    } else {
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<blocks, BLOCK, 0, stream>>>(g, o, -1));
    }
    GMS_KERNEL_CHECK(debug, stream, "micro_head");
    GMS_LAUNCH(GMS_K_BLEND_FWD, stream, fwd2<<<blocks, BLOCK, 0, stream>>>(g, o));
    GMS_KERNEL_CHECK(debug, stream, "micro_fwd");
    GMS_LAUNCH(GMS_K_BLEND_FINALIZE, stream, micro_finalize_kernel<<<(unsigned)g.T, BLOCK, 0, stream>>>(g, o));
    GMS_KERNEL_CHECK(debug, stream, "micro_finalize");
    return GMS_OK;
}

int32_t launch_micro_backward(const BlendGrid &g_in, const BlendBwdArgs &a, uint32_t max_units, bool debug, hipStream_t stream)
{
    BlendGrid g = g_in;
    experiment_switches(g, 1024u, stream);          // (make EXPERIMENTS=1 only: GMS_DBG & 1024 = per-wave phase stamps)
    const unsigned blocks = blend_grid_units(max_units);
    static int trip = -1;
    if (trip < 0) { const char *e = getenv("GMS_TRIP_BWD"); trip = e ? atoi(e) : 2; }
    const bool invd = a.has_invd && a.dL_dinvd;
    const bool ru = resident_units();
    if (a.part) {                           // deterministic mode (gmsplat.h): per-wave tables, ordered adds, per-instance partial records
        if (ru) {
            if (invd) GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (ru_bwd_kernel<true, 2, 0, true><<<blocks, BLOCK, 0, stream>>>(g, a)));
            else GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (ru_bwd_kernel<false, 2, 0, true><<<blocks, BLOCK, 0, stream>>>(g, a)));
        } else {
            if (invd) GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<true, 2, 0, true><<<blocks, BLOCK, 0, stream>>>(g, a)));
            else GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<false, 2, 0, true><<<blocks, BLOCK, 0, stream>>>(g, a)));
        }
    } else if (fault_mode() == 2 && !invd) {       // negative control (gms_set_fault): its own instantiation
        if (ru) GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (ru_bwd_kernel<false, 2, 2><<<blocks, BLOCK, 0, stream>>>(g, a)));
        else GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<false, 2, 2><<<blocks, BLOCK, 0, stream>>>(g, a)));
    } else if (ru) {
        auto kern = trip == 1 ? (invd ? ru_bwd_kernel<true, 1, 0> : ru_bwd_kernel<false, 1, 0>)
                  : trip == 4 ? (invd ? ru_bwd_kernel<true, 4, 0> : ru_bwd_kernel<false, 4, 0>)
                              : (invd ? ru_bwd_kernel<true, 2, 0> : ru_bwd_kernel<false, 2, 0>);
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<blocks, BLOCK, 0, stream>>>(g, a));
    } else {
        auto kern = trip == 1 ? (invd ? micro_bwd_kernel<true, 1, 0> : micro_bwd_kernel<false, 1, 0>)
                  : trip == 4 ? (invd ? micro_bwd_kernel<true, 4, 0> : micro_bwd_kernel<false, 4, 0>)
                              : (invd ? micro_bwd_kernel<true, 2, 0> : micro_bwd_kernel<false, 2, 0>);
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<blocks, BLOCK, 0, stream>>>(g, a));
    }
    GMS_KERNEL_CHECK(debug, stream, "micro_bwd");
    return GMS_OK;
}

}  // namespace gms
