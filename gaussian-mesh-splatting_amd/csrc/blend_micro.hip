// blend_micro.hip -- micro-tile compositing for gfx950: the wave works on four 4x4 pixel blocks at once, one per DPP row.
//
// Why.  Mesh-bound splats are small (about 14 pixels with alpha >= 1/255 per (Gaussian, tile) instance on the headline
// scene).  A wave that owns an 8x8 quadrant and walks the quadrant's culled list one splat at a time (blend.hip) has 12 of
// its 64 lanes busy and pays a 64-lane reduction per splat.  Here the unit of work is a 4x4 pixel BLOCK with its own
// pre-filtered list: a wave carries four blocks, one per 16-lane row, each row walking ITS OWN list, so one trip of the
// wave advances four (block, splat) pairs: 0.68 trips per instance instead of 1.19, 39 % of the lanes busy instead of 19 %,
// and the ten gradient sums of a splat are reduced inside a 16-lane row with row-local DPP only (29 VALU for four splats
// instead of 26 for one).
//
//   micro_filter   one block per work unit (tile, segment of <= L <= 1024 list entries): gathers the unit's splat records
//                  once, tests each against the tile's sixteen 4x4 blocks (bounding box of the alpha >= 1/255 ellipse, then
//                  the exact ellipse-vs-rectangle test: it can only drop pairs every pixel of the block would skip) and
//                  writes, per block, the ids of the survivors in list order (ballot ranks, no atomics).  The cull is paid
//                  once per frame instead of once per pass (products, forward walk, backward walk).
//   micro_head / micro_fwd / micro_finalize / micro_bwd
//                  the segment-parallel scheme of blend.hip unchanged -- first segments walked exactly, transmittance
//                  products of the middle segments, exact walk of segments 1.. from the prefix product, partial sums in
//                  order, backward restarted at segment boundaries -- on (unit, quadrant) waves whose four rows are the
//                  quadrant's four blocks.  No cull, no ballot loop: a row's queue holds only entries that hit its block.
//
// Positions.  n_contrib holds, per pixel, seg * L + (index in the block's list of that segment) + 1 of the last splat
// applied: monotone along the block's concatenated lists, which is all the backward needs.
#include <stdlib.h>

#include "gms_common.h"
#include "gms_blend.h"

namespace gms {

constexpr int QROW = 17;          // LDS queue slots per row (16 used): 17 x 12 dwords staggers the four rows over the banks
constexpr int QSLOTS = 4 * QROW;

// Piece classes (which launches walk a (unit, block) piece) and the length buckets of the counting sort
enum { PC_FIRST = 0, PC_MID_EARLY, PC_MID_LATE, PC_LAST, PC_BWD, PC_COUNT };
constexpr int MBUCKETS = (int)BinningState::MICRO_BUCKETS;
static_assert(PC_COUNT == (int)BinningState::MICRO_CLASSES, "piece classes");
constexpr int MREGIONS = (int)BinningState::MICRO_REGIONS;
constexpr int WH_KEYS = PC_COUNT * MREGIONS * MBUCKETS;          // counters per histogram: (class, region, bucket)
constexpr int WH_CURSOR = WH_KEYS, WH_TOTAL = 2 * WH_KEYS;
// Spatial region of a tile = the XCD whose L2 should hold its splat records: 2x2-tile cells (32 px) dealt to the eight XCDs
// in a skewed pattern.  A cell's Gaussians are shared with its neighbours only along the ~5 px halo, so an XCD gathers about
// 1.3 / 8 of the scene's records (~2.4 MB at 300 k Gaussians: inside its 4 MB L2) instead of all of them.
__device__ __forceinline__ int tile_region(int tile, int gx) { const int tx = tile % gx, ty = tile / gx; return ((tx >> 1) + 3 * (ty >> 1)) & 7; }
__device__ __forceinline__ int piece_bucket(uint32_t count, uint32_t L) { return (int)min(32u, (count * 32u + L - 1u) / L); }
__device__ __forceinline__ int piece_class(int seg, int nseg, uint32_t L)
{
    if (seg == 0) return PC_FIRST;
    if (seg == nseg - 1) return PC_LAST;
    return seg < tloc_head(L) ? PC_MID_EARLY : PC_MID_LATE;
}

// pixel `li` of 4x4 block `b` of a tile; index of the pixel in the per-(unit, pixel) segment state = b * 16 + li
__device__ __forceinline__ void block_pixel(int tx, int ty, int b, int li, int &xi, int &yi)
{
    xi = tx * TILE + (b & 3) * 4 + (li & 3);
    yi = ty * TILE + (b >> 2) * 4 + (li >> 2);
}

// One row (16 lanes) of a wave = one (unit, block) piece; all fields are uniform across the row's lanes.
struct Row {
    bool on;               // the row has a piece
    int b, tile, seg, nseg, xi, yi, tid; bool inside;
    uint32_t idx, slot0, beg, cn, cnt, L;
    const uint32_t *ml;    // the piece's ids
};

// Which wave of which class of a piece table block `blockIdx.x` walks.  The blocks of XCD x (= blockIdx % 8, how the
// dispatcher deals them) take the x-th eighth of every class: the table is sorted by (class, region, length), so that eighth
// is region x's pieces up to the imbalance between regions -- the records a wave gathers are in its own XCD's L2.
__device__ __forceinline__ bool wave_of_block(const BlendGrid &g, const uint32_t *tot, const int *classes, int nclasses, uint32_t &first, uint32_t &count, uint32_t &wave)
{
    const uint32_t x = blockIdx.x & 7u;
    uint32_t w = blockIdx.x >> 3, off = 0;
    int ci = 0;
    for (int c = 0; c < PC_COUNT; c++) {
        const uint32_t n = tot[c];
        if (ci < nclasses && classes[ci] == c) {
            ci++;
            const uint32_t nw = (n + 3u) / 4u, w8 = (nw + 7u) / 8u;
            if (w < w8) {
                wave = (g.dbg & 0x10000u) ? x * w8 + w : w * 8u + x;      // regions on: XCD x walks the x-th eighth; off: interleaved
                first = off; count = n;
                return wave < nw;
            }
            w -= w8;
        }
        if (c < PC_BWD) off += n;          // (the backward table is a table of its own)
    }
    return false;
}

// rows 4 w .. 4 w + 3 of the sorted piece table `tab[first .. first + count)` for wave w
__device__ __forceinline__ Row load_row(const BlendGrid &g, const uint32_t *tab, uint32_t first, uint32_t count, uint32_t wave)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    Row r;
    const uint32_t pi = 4u * wave + (uint32_t)row;
    r.on = pi < count;
    const uint32_t desc = r.on ? tab[first + pi] : 0u;
    r.idx = desc >> 4; r.b = (int)(desc & 15u);
    const uint4 r0 = g.unit_tile[2 * (size_t)r.idx], r1 = g.unit_tile[2 * (size_t)r.idx + 1];
    r.tile = (int)r0.x; r.seg = (int)r0.y; r.nseg = (int)r0.z; r.slot0 = r0.w;
    r.L = g.scan_out[3];
    r.beg = r1.x + (uint32_t)r.seg * r.L;
    const uint32_t end = min(r1.y, r.beg + r.L);
    r.cn = end > r.beg ? end - r.beg : 0u;
    if ((uint64_t)r1.y > g.capacity) r.on = false;          // overflowed optimistic launch: the host re-runs
    r.cnt = r.on ? g.mcount[(size_t)r.idx * 16 + r.b] : 0u;
    r.ml = g.mlist + (size_t)16 * r.beg + (size_t)r.b * r.cn;
    block_pixel(r.tile % g.gx, r.tile / g.gx, r.b, li, r.xi, r.yi);
    r.inside = r.on && r.xi < g.W && r.yi < g.H;
    r.tid = r.b * 16 + li;
    return r;
}

__device__ __forceinline__ uint32_t max4rows(uint32_t v)          // v is row-uniform
{
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

// ------------------------------------------------------------------------------------ filter
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
    // bounding box of the ellipse against the tile (also rejects the ext = -1e30 records of splats below 1/255 everywhere)
    if (px + ex < tx0 || px - ex > tx0 + 15.f || py + ey < ty0 || py - ey > ty0 + 15.f) return 0u;
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
    const float mx = fmaxf(fabsf(tx0 - px), fabsf(tx0 + 15.f - px)), my = fmaxf(fabsf(ty0 - py), fabsf(ty0 + 15.f - py));
    const float gross = mx * (A * mx + 2.f * fabsf(B) * my) + C * my * my;
    const float thr2 = thr * 1.0001f + 0.01f + 4e-6f * gross;            // inflated threshold
    const float grow = thr2 * __builtin_amdgcn_rcpf(thr);                // ex'^2 / ex^2 = ey'^2 / ey^2
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

// One block per unit.  Each wave takes a contiguous quarter of the unit's entries (64 at a time), counts its hits per block,
// and after ONE barrier writes the survivors' ids behind those of the waves in front of it (ballot ranks: list order kept).
template <int MAXC>                               // 64-entry chunks per wave: L <= 256 MAXC
__global__ void __launch_bounds__(BLOCK) micro_filter_kernel(BlendGrid g, const SplatRec *rec)
{
    __shared__ uint32_t wcnt[4][16];
    __shared__ uint32_t running[16];
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cn = u.end - u.beg;
    uint32_t *out = g.mlist + (size_t)16 * u.beg;
    const float tx0 = (float)(u.tx * TILE), ty0 = (float)(u.ty * TILE);
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t per = ((cn + 255u) / 256u) * 64u;          // entries per wave, a multiple of 64
    const uint32_t w0 = (uint32_t)wave * per;
    uint32_t ids[MAXC], masks[MAXC];
    uint32_t mycnt = 0;                            // lane b < 16: hits of block b in this wave's entries
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const uint32_t e = w0 + (uint32_t)(c * 64 + lane);
        ids[c] = 0; masks[c] = 0;
        if ((uint32_t)(c * 64) < per && e < cn) {
            ids[c] = (uint32_t)g.keys[u.beg + e];
            masks[c] = block_mask(rec[ids[c]], tx0, ty0);
        }
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint32_t n = (uint32_t)__builtin_popcountll(__ballot((masks[c] >> b) & 1u));
            if (lane == b) mycnt += n;
        }
    }
    if (lane < 16) wcnt[wave][lane] = mycnt;
    __syncthreads();
    uint32_t base[16];
#pragma unroll
    for (int b = 0; b < 16; b++) {
        uint32_t s0 = 0;
        for (int w = 0; w < wave; w++) s0 += wcnt[w][b];
        base[b] = s0;
    }
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint64_t bal = __ballot((masks[c] >> b) & 1u);
            if ((masks[c] >> b) & 1u) out[(size_t)b * cn + base[b] + (uint32_t)__builtin_popcountll(bal & lt)] = ids[c];
            base[b] += (uint32_t)__builtin_popcountll(bal);
        }
    }
    if (tid < 16) running[tid] = wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
    if (tid < 16) g.mcount[(size_t)u.idx * 16 + tid] = running[tid];
}

// ------------------------------------------------------------------------------------ plan
// Counting sort of the (unit, block) pieces by list length, heaviest first, one table for the forward launches (classes
// first | middle early | middle late | last, in this order) and one for the backward (non-empty pieces): four CONSECUTIVE
// table entries form a wave, so the four rows of a wave walk lists of (nearly) the same length -- 0.58 wave trips per
// instance instead of the 0.81 the four blocks of one quadrant give -- and waves of equal weight are dispatched heaviest
// first (no long tail behind a few heavy quadrants).  Two small launches: histogram, scatter.  Each block of 1024 threads
// covers 64 units x 16 pieces and counts in LDS first: a global counter sees one atomic per block, not one per piece.
constexpr int PLAN_THREADS = 1024, PLAN_UNITS = PLAN_THREADS / 16;

__device__ __forceinline__ bool plan_piece(const BlendGrid &g, uint32_t &desc, int &cls, int &reg, int &bk, uint32_t &c)
{
    const uint32_t idx = blockIdx.x * PLAN_UNITS + (threadIdx.x >> 4);
    const int b = threadIdx.x & 15;
    const uint32_t nunits = g.unit_first[g.T];
    if (idx >= nunits || idx >= g.max_units) return false;
    const uint4 r0 = g.unit_tile[2 * (size_t)idx], r1 = g.unit_tile[2 * (size_t)idx + 1];
    if ((uint64_t)r1.y > g.capacity) return false;
    const uint32_t L = g.scan_out[3];
    c = g.mcount[(size_t)idx * 16 + b];
    bk = piece_bucket(c, L); cls = piece_class((int)r0.y, (int)r0.z, L); reg = (g.dbg & 0x10000u) ? tile_region((int)r0.x, g.gx) : 0;
    desc = idx * 16u + (uint32_t)b;
    return true;
}

__global__ void __launch_bounds__(PLAN_THREADS) micro_plan_hist_kernel(BlendGrid g)
{
    __shared__ uint32_t h[WH_KEYS];
    const int tid = threadIdx.x;
    for (int k = tid; k < WH_KEYS; k += PLAN_THREADS) h[k] = 0u;
    __syncthreads();
    uint32_t desc, c; int cls, reg, bk;
    if (plan_piece(g, desc, cls, reg, bk, c)) {
        atomicAdd(&h[(cls * MREGIONS + reg) * MBUCKETS + bk], 1u);
        if (c > 0) atomicAdd(&h[(PC_BWD * MREGIONS + reg) * MBUCKETS + bk], 1u);
    }
    __syncthreads();
    for (int k = tid; k < WH_KEYS; k += PLAN_THREADS)
        if (h[k]) atomicAdd(&g.whist[k], h[k]);
}

__global__ void __launch_bounds__(PLAN_THREADS) micro_plan_scatter_kernel(BlendGrid g)
{
    __shared__ uint32_t h[WH_KEYS];          // pieces of this block per (class, region, bucket); then the block's first slot
    __shared__ uint32_t base[WH_KEYS];       // first table slot of every key
    __shared__ uint32_t total[PC_COUNT];
    const int tid = threadIdx.x;
    for (int k = tid; k < WH_KEYS; k += PLAN_THREADS) { h[k] = 0u; base[k] = g.whist[k]; }
    __syncthreads();
    if (tid < PC_COUNT) {          // table order: class | region | heaviest bucket first (one thread per class: 264 counters in LDS)
        uint32_t run = 0;
        for (int r = 0; r < MREGIONS; r++)
            for (int k = MBUCKETS - 1; k >= 0; k--) {
                const int key = (tid * MREGIONS + r) * MBUCKETS + k;
                const uint32_t n = base[key];
                base[key] = run; run += n;
            }
        total[tid] = run;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < PC_COUNT) g.whist[WH_TOTAL + tid] = total[tid];
    uint32_t desc = 0, c = 0, rank = 0, rank_b = 0; int cls = 0, reg = 0, bk = 0;
    const bool on = plan_piece(g, desc, cls, reg, bk, c);
    const int key = (cls * MREGIONS + reg) * MBUCKETS + bk, key_b = (PC_BWD * MREGIONS + reg) * MBUCKETS + bk;
    if (on) {
        rank = atomicAdd(&h[key], 1u);
        if (c > 0) rank_b = atomicAdd(&h[key_b], 1u);
    }
    __syncthreads();
    for (int k = tid; k < WH_KEYS; k += PLAN_THREADS) {
        const uint32_t n = h[k];
        h[k] = n ? atomicAdd(&g.whist[WH_CURSOR + k], n) : 0u;
    }
    __syncthreads();
    if (!on) return;
    uint32_t off = 0;
    for (int k = 0; k < cls; k++) off += total[k];
    g.wtab_fwd[off + base[key] + h[key] + rank] = desc;
    if (c > 0) g.wtab_bwd[base[key_b] + h[key_b] + rank_b] = desc;
}

// ------------------------------------------------------------------------------------ queue
// Each row keeps 16 entries of its own list in LDS; lane i of row r gathers entry j0 + i of the row's list.
__device__ __forceinline__ SplatRec zero_rec()
{
    SplatRec z;
    z.q0 = make_float4(0.f, 0.f, 0.f, 0.f); z.q1 = z.q0; z.q2 = z.q0;
    return z;
}
__device__ __forceinline__ void queue_clear(SplatRec *recs)
{
    const int lane = threadIdx.x & 63;
    const SplatRec z = zero_rec();
    recs[lane] = z;
    if (lane < QSLOTS - WAVE) recs[WAVE + lane] = z;
}

// ------------------------------------------------------------------------------------ fwd
// Front-to-back walk of the four rows' lists.  A row is either EXACT (reference skip / stop tests from the true prefix
// transmittance; its end state is written as image pixels for a single-segment tile, else as the segment's partials) or a
// PRODUCT (a middle segment's prod(1 - alpha) without termination: the prefix of the segments behind it), chosen per row.
template <int NE>
__device__ __forceinline__ void micro_walk(const BlendGrid &g, const BlendFwdOut &o, const Row &r, SplatRec *recs, bool exact, bool product, int phase)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const float xf = (float)r.xi, yf = (float)r.yi;
    float *st = g.seg_state + (size_t)(r.slot0 + (uint32_t)r.seg) * SEG_FLOATS;
    bool skip = !(exact || product);
    if (product && phase == 1 && g.tile_dead[r.tile]) { st[SEG_TLOC * TILE_PIX + r.tid] = 0.f; skip = true; }   // products are irrelevant
    const uint32_t cnt = skip ? 0u : r.cnt;
    const uint32_t maxcnt = max4rows(cnt);
    const uint32_t posbase = (uint32_t)r.seg * r.L;

    float T = 1.f;
    if (exact && r.seg > 0) {
        // prefix product of the segments in front, four independent loads per step (same left-to-right order)
        const float *tl = g.seg_state + (size_t)r.slot0 * SEG_FLOATS + SEG_TLOC * TILE_PIX + r.tid;
        int k = 0;
        for (; k + 4 <= r.seg; k += 4) {
            const float t0 = tl[(size_t)k * SEG_FLOATS], t1 = tl[(size_t)(k + 1) * SEG_FLOATS];
            const float t2 = tl[(size_t)(k + 2) * SEG_FLOATS], t3 = tl[(size_t)(k + 3) * SEG_FLOATS];
            T = T * t0 * t1 * t2 * t3;
        }
        for (; k < r.seg; k++) T *= tl[(size_t)k * SEG_FLOATS];
    }
    const bool dead_on_entry = T < T_MIN;          // only possible for an exact row with seg > 0
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    bool done = skip || !r.inside || dead_on_entry;
    queue_clear(recs);
    // software-pipelined queue fill: the records of batch k+1 and the ids of batch k+2 are in flight while batch k is walked
    SplatRec R = zero_rec();
    uint32_t id2 = 16u + li < cnt ? r.ml[16u + li] : 0u;
    if ((uint32_t)li < cnt) R = o.rec[r.ml[li]];

    for (uint32_t j0 = 0; j0 < maxcnt; j0 += 16) {
        if (__all(done || j0 >= cnt)) break;
        wave_sync();
        if (j0 + li < cnt) recs[row * QROW + li] = R;
        if (j0 + 16u + li < cnt) R = o.rec[id2];
        id2 = j0 + 32u + li < cnt ? r.ml[j0 + 32u + li] : 0u;
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
                const float dx = r0.x - xf, dy = r0.y - yf;
                val[e] = j0 + t + e < cnt;
                pw[e] = pair_power(r0.z, r0.w, r1[e].x, dx, dy);
                al[e] = fminf(ALPHA_MAX, r1[e].y * __expf(pw[e]));
            }
#pragma unroll
            for (int e = 0; e < NE; e++) {
                bool act = val[e] && !done && pw[e] <= 0.f && al[e] >= ALPHA_MIN;
                const float testT = T * (1.f - al[e]);
                if (act && exact && testT < T_MIN) { done = true; act = false; }       // the reference's stop rule
                if (act) {
                    const float w = al[e] * T;
                    C0 += r1[e].z * w; C1 += r1[e].w * w; C2 += r2[e].x * w;
                    Dp += r2[e].y * w;
                    T = testT;
                    last = posbase + j0 + (uint32_t)(t + e) + 1u;
                    // a product below 1e-4: every later segment starts dead whatever the exact value
                    if (!exact && T < T_MIN) done = true;
                }
            }
            if (__all(done || j0 + t + NE >= cnt)) break;
        }
    }
    if (skip || !r.on) return;
    if (product) { st[SEG_TLOC * TILE_PIX + r.tid] = T; return; }
    if (r.nseg == 1) {
        if (r.inside) {
            const size_t pid = (size_t)r.yi * g.W + r.xi, HW = (size_t)g.W * g.H;
            o.final_T[pid] = T;
            o.n_contrib[pid] = last;
            o.out_color[pid] = C0 + T * o.bg[0];
            o.out_color[HW + pid] = C1 + T * o.bg[1];
            o.out_color[2 * HW + pid] = C2 + T * o.bg[2];
            o.out_invdepth[pid] = Dp;
        }
    } else {
        st[SEG_C0 * TILE_PIX + r.tid] = C0; st[SEG_C1 * TILE_PIX + r.tid] = C1; st[SEG_C2 * TILE_PIX + r.tid] = C2;
        st[SEG_D * TILE_PIX + r.tid] = Dp;
        st[SEG_TEND * TILE_PIX + r.tid] = dead_on_entry ? -1.f : T;
        st[SEG_LAST * TILE_PIX + r.tid] = __uint_as_float(last);
        // the first segment's exact walk doubles as its transmittance product (see blend.hip)
        if (r.seg == 0) st[SEG_TLOC * TILE_PIX + r.tid] = done ? 0.f : T;
    }
}

// First launch(es): every piece that depends on nothing -- first segments exactly, middle segments as products.
// phase -1: all of them; phase 0: first + early middle segments; phase 1: the late middle segments (deep scenes: after the
// tile-dead check).  The class ranges of the forward table are contiguous in exactly this order.
template <int NE>
__global__ void __launch_bounds__(WAVE) micro_head_kernel(BlendGrid g, BlendFwdOut o, int phase)
{
    __shared__ SplatRec recs[QSLOTS];
    const int all[3] = {PC_FIRST, PC_MID_EARLY, PC_MID_LATE}, late[1] = {PC_MID_LATE};
    uint32_t first, count, wave;
    if (!wave_of_block(g, g.whist + WH_TOTAL, phase == 1 ? late : all, phase == 1 ? 1 : (phase == 0 ? 2 : 3), first, count, wave)) return;
    const Row r = load_row(g, g.wtab_fwd, first, count, wave);
    const bool exact = r.on && r.seg == 0, product = r.on && r.seg > 0;
    micro_walk<NE>(g, o, r, recs, exact, product, phase);
}

// Second launch: segments 1.. of the multi-segment tiles, exactly, from the prefix product of the segments in front.
template <int NE>
__global__ void __launch_bounds__(WAVE) micro_fwd_kernel(BlendGrid g, BlendFwdOut o)
{
    __shared__ SplatRec recs[QSLOTS];
    const int later[3] = {PC_MID_EARLY, PC_MID_LATE, PC_LAST};
    uint32_t first, count, wave;
    if (!wave_of_block(g, g.whist + WH_TOTAL, later, 3, first, count, wave)) return;
    const Row r = load_row(g, g.wtab_fwd, first, count, wave);
    micro_walk<NE>(g, o, r, recs, r.on, false, -1);
}

// tile_dead[t] = 1 when the product of the first tloc_head(L) segment transmittances is < 1e-4 for every pixel
__global__ void __launch_bounds__(BLOCK) micro_tloc_check_kernel(BlendGrid g)
{
    const int tile = blockIdx.x;
    const int nseg = (int)(g.unit_first[tile + 1] - g.unit_first[tile]);
    const int nhead = tloc_head(g.scan_out[3]);
    if (nseg <= nhead + 1) return;                 // no late middle segment exists (the last one needs no product)
    if ((uint64_t)g.tile_offset[tile + 1] > g.capacity) return;
    const int tid = threadIdx.x;
    int xi, yi;
    block_pixel(tile % g.gx, tile / g.gx, tid >> 4, tid & 15, xi, yi);
    const float *st0 = g.seg_state + (size_t)g.mseg_first[tile] * SEG_FLOATS;
    float T = 1.f;
    for (int k = 0; k < nhead; k++) T *= st0[(size_t)k * SEG_FLOATS + SEG_TLOC * TILE_PIX + tid];
    const int dead = __syncthreads_and(T < T_MIN || xi >= g.W || yi >= g.H);
    if (tid == 0) g.tile_dead[tile] = dead ? 1u : 0u;
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
    int xi, yi;
    block_pixel(tile % g.gx, tile / g.gx, tid >> 4, tid & 15, xi, yi);
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

template <bool INVD, int NE, int FAULT>
__global__ void __launch_bounds__(WAVE) micro_bwd_kernel(BlendGrid g, BlendBwdArgs a)
{
    __shared__ SplatRec recs[QSLOTS];
    __shared__ uint32_t ids[2][QSLOTS];                 // ids of the batch being walked / of the batch whose sums await their atomics
    __shared__ float ystash[16][40];                    // the reduced sums of one batch: [trip][row * 10 + field lane]
    const int bwd_class[1] = {PC_BWD};
    uint32_t first, count, wave;
    if (!wave_of_block(g, g.whist + WH_TOTAL, bwd_class, 1, first, count, wave)) return;
    const Row r = load_row(g, g.wtab_bwd, 0u, count, wave);
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const float xf = (float)r.xi, yf = (float)r.yi;
    const size_t HW = (size_t)g.W * g.H;
    const size_t pid = (size_t)r.yi * g.W + r.xi;
    const float Tfinal = r.inside ? a.final_T[pid] : 0.f;
    const uint32_t last = r.inside ? a.n_contrib[pid] : 0u;      // seg * L + index + 1 of the last splat this pixel applied
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dinvd = 0.f;
    if (r.inside) {
        dp0 = a.dL_dpix[pid]; dp1 = a.dL_dpix[HW + pid]; dp2 = a.dL_dpix[2 * HW + pid];
        if (INVD) dinvd = a.dL_dinvd[pid];
    }
    const float Tfinal_bgdot = Tfinal * (a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2);
    const uint32_t posbase = (uint32_t)r.seg * r.L;
    // entries [0, lrel) of this block's list of this segment were composited by this pixel
    const uint32_t lrel = last > posbase ? min(last - posbase, r.cnt) : 0u;
    uint32_t top = lrel;                                    // furthest entry any pixel of the row composited
    top = max(top, (uint32_t)__shfl_xor((int)top, 8)); top = max(top, (uint32_t)__shfl_xor((int)top, 4));
    top = max(top, (uint32_t)__shfl_xor((int)top, 2)); top = max(top, (uint32_t)__shfl_xor((int)top, 1));
    const uint32_t maxtop = max4rows(top);
    if (maxtop == 0) return;
    // software-pipelined queue fill (see the loop below); batch 0's ids and records start travelling here, behind the
    // segment-restart loads that follow
    uint32_t idc = (uint32_t)li < top ? r.ml[top - 1u - (uint32_t)li] : 0u;             // id of this lane's entry of the current batch
    uint32_t id2 = 16u + li < top ? r.ml[top - 1u - (16u + li)] : 0u;                    // ... of the next batch
    SplatRec R = zero_rec();
    if ((uint32_t)li < top) R = a.rec[(FAULT == 10 || FAULT == 11) ? (idc & 1023u) : idc];

    BwdState st8 = {Tfinal, 0.f, 0.f, 0.f, 0.f};
    {
        // restart of the recurrence at the segment boundary: T after this segment's last applied splat and the colour
        // composited behind it (sum of the live partials of the later segments) divided by that T.  Rows of single-segment
        // tiles (and rows that are off) keep (Tfinal, 0).
        const bool multi = r.on && r.nseg > 1 && top > 0;
        const float te = multi ? g.seg_state[(size_t)(r.slot0 + (uint32_t)r.seg) * SEG_FLOATS + SEG_TEND * TILE_PIX + r.tid] : 0.f;
        const bool restart = multi && te > 0.f;
        float S0 = 0.f, S1 = 0.f, S2 = 0.f, SD = 0.f;
        bool stop = !restart;
        for (int k0 = r.seg + 1; ; k0 += 4) {
            if (k0 >= r.nseg) stop = true;
            if (__all(stop)) break;
            float tk[4], c0[4], c1[4], c2[4], dd[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float *sk = g.seg_state + (size_t)(r.slot0 + (uint32_t)max(0, min(k0 + j, r.nseg - 1))) * SEG_FLOATS;
                tk[j] = stop ? -1.f : sk[SEG_TEND * TILE_PIX + r.tid]; c0[j] = stop ? 0.f : sk[SEG_C0 * TILE_PIX + r.tid];
                c1[j] = stop ? 0.f : sk[SEG_C1 * TILE_PIX + r.tid]; c2[j] = stop ? 0.f : sk[SEG_C2 * TILE_PIX + r.tid];
                dd[j] = (INVD && !stop) ? sk[SEG_D * TILE_PIX + r.tid] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // a pixel that is dead on entry to segment k is dead for every later one: stop at the first
                if (k0 + j >= r.nseg || tk[j] < 0.f) stop = true;
                if (!stop) { S0 += c0[j]; S1 += c1[j]; S2 += c2[j]; SD += dd[j]; }
            }
        }
        if (restart) {
            st8.T = te;
            const float inv = FAULT == 2 ? 0.f : 1.f / te;
            st8.acc0 = S0 * inv; st8.acc1 = S1 * inv; st8.acc2 = S2 * inv; st8.accd = SD * inv;
        }
    }

    // lane -> field of the Gaussian's 64-byte gradient record it adds to (the layout row_reduce10 leaves)
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
    // slot of the lane's sum in a trip's stash row: the ten field lanes of a row, compacted
    const int sslot = row * 10 + ((li & 1) ? (li == 1 ? 8 : 9) : (li >> 1));      // (lane 14 -> 7; only written / read when alane)
    float *const abase = a.accum + afield;
    queue_clear(recs);
    ids[0][lane] = 0u; ids[1][lane] = 0u;
    if (lane < QSLOTS - WAVE) { ids[0][WAVE + lane] = 0u; ids[1][WAVE + lane] = 0u; }

    // Back to front: global trip t0 handles entry top - 1 - t0 of every row's list (the rows are aligned at their tops).
    // Pipeline per batch of 16 trips:  [records of batch k -> LDS]  [atomics of batch k-1 from the stash]  [loads of batch
    // k+1 / ids of batch k+2 issued]  [walk batch k: sums -> stash].  No memory operation is issued inside the walk, and
    // everything the next batch boundary waits for (vmcnt counts loads and atomics alike on gfx9) was issued a whole walk
    // earlier: neither the gather latency (two dependent hops) nor the atomics' round trip is exposed.
    uint32_t prev_trips = 0;                      // bit t: trip t of the previous batch left sums in the stash
    int buf = 0;
    auto flush = [&](int b) {
        while (prev_trips) {
            const int t = __builtin_ctz(prev_trips);
            prev_trips &= prev_trips - 1u;
            if (alane) {
                const float y = ystash[t][sslot];
                // a row with no active pixel for this entry summed exact zeros: nothing to add (and its id may be stale)
                if (FAULT == 9 || FAULT == 11) { if (y == 123.456f) a.accum[0] = y; }            // (timing experiments: no atomics)
                else if (FAULT == 12) {      // timing experiment: every XCD adds into cache lines no other XCD touches
                    const uint32_t id = ids[b][row * QROW + t];
                    if (y != 0.f) unsafeAtomicAdd(abase + (size_t)(((id >> 4) << 4) | ((blockIdx.x & 7u) << 1) | (id & 1u)) * GRAD_STRIDE, y);
                }
                else if (y != 0.f) unsafeAtomicAdd(abase + (size_t)ids[b][row * QROW + t] * GRAD_STRIDE, y);
            }
        }
    };
    for (uint32_t g0 = 0; g0 < maxtop; g0 += 16, buf ^= 1) {
        wave_sync();
        if (g0 + li < top) { ids[buf][row * QROW + li] = idc; recs[row * QROW + li] = R; }
        flush(buf ^ 1);
        idc = id2;
        if (g0 + 16u + li < top) R = a.rec[(FAULT == 10 || FAULT == 11) ? (id2 & 1023u) : id2];
        id2 = g0 + 32u + li < top ? r.ml[top - 1u - (g0 + 32u + li)] : 0u;
        wave_sync();
        const int nt = (int)min(16u, maxtop - g0);
        for (int t = 0; t < nt; t += NE) {
            bool act[NE]; float dx[NE], dy[NE], G[NE], al[NE]; float4 r1[NE], r2[NE];
            bool anyact = false;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const SplatRec *sr = recs + row * QROW + t + e;
                const float4 r0 = sr->q0;
                r1[e] = sr->q1; r2[e] = sr->q2;
                dx[e] = r0.x - xf; dy[e] = r0.y - yf;
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
                if (alane) ystash[t + e][sslot] = y;
                prev_trips |= 1u << (t + e);
            }
        }
    }
    wave_sync();
    flush(buf ^ 1);
}

// ------------------------------------------------------------------------------------ host
static uint32_t micro_flags()
{
    static int f = -1;
    if (f < 0) { const char *e = getenv("GMS_MICRO_REGIONS"); f = (e && atoi(e) != 0) ? 0x10000 : 0; }
    return (uint32_t)f;
}

int32_t launch_micro_forward(const BlendGrid &g_in, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream)
{
    BlendGrid g = g_in;
    g.dbg |= micro_flags();
    static int deep_env = -2;
    if (deep_env == -2) { const char *e = getenv("GMS_DEEP"); deep_env = e ? atoi(e) : -1; }
    const bool deep = deep_env >= 0 ? deep_env != 0 : g.capacity > 512ull * (uint64_t)g.T;
    const unsigned blocks = blend_grid_units(max_units);
    static int trip = -1;
    if (trip < 0) { const char *e = getenv("GMS_TRIP"); trip = e ? atoi(e) : 2; }
    const uint32_t Lh = seg_len_min();             // micro mode: one segment length for every frame
    auto filter = Lh <= 256u ? micro_filter_kernel<1> : (Lh <= 512u ? micro_filter_kernel<2> : micro_filter_kernel<4>);
    GMS_LAUNCH(GMS_K_MICRO_FILTER, stream, filter<<<blocks, BLOCK, 0, stream>>>(g, o.rec));
    const unsigned pblocks = (blocks + PLAN_UNITS - 1u) / PLAN_UNITS;
    GMS_LAUNCH(GMS_K_MICRO_FILTER, stream, micro_plan_hist_kernel<<<pblocks, PLAN_THREADS, 0, stream>>>(g));
    GMS_LAUNCH(GMS_K_MICRO_FILTER, stream, micro_plan_scatter_kernel<<<pblocks, PLAN_THREADS, 0, stream>>>(g));
    GMS_KERNEL_CHECK(debug, stream, "micro_filter");
    auto head = trip == 4 ? micro_head_kernel<4> : (trip == 1 ? micro_head_kernel<1> : micro_head_kernel<2>);
    auto fwd2 = trip == 4 ? micro_fwd_kernel<4> : (trip == 1 ? micro_fwd_kernel<1> : micro_fwd_kernel<2>);
    if (deep) {     // deep scene: first + early middle segments, tile-dead check, then the late middle segments of live tiles
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<4u * blocks + 64u, WAVE, 0, stream>>>(g, o, 0));
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, micro_tloc_check_kernel<<<(unsigned)g.T, BLOCK, 0, stream>>>(g));
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<4u * blocks + 64u, WAVE, 0, stream>>>(g, o, 1));
    } else {
        GMS_LAUNCH(GMS_K_BLEND_HEAD, stream, head<<<4u * blocks + 64u, WAVE, 0, stream>>>(g, o, -1));
    }
    GMS_KERNEL_CHECK(debug, stream, "micro_head");
    GMS_LAUNCH(GMS_K_BLEND_FWD, stream, fwd2<<<4u * blocks + 64u, WAVE, 0, stream>>>(g, o));
    GMS_KERNEL_CHECK(debug, stream, "micro_fwd");
    GMS_LAUNCH(GMS_K_BLEND_FINALIZE, stream, micro_finalize_kernel<<<(unsigned)g.T, BLOCK, 0, stream>>>(g, o));
    GMS_KERNEL_CHECK(debug, stream, "micro_finalize");
    return GMS_OK;
}

int32_t launch_micro_backward(const BlendGrid &g_in, const BlendBwdArgs &a, uint32_t max_units, bool debug, hipStream_t stream)
{
    BlendGrid g = g_in;
    g.dbg |= micro_flags();
    const unsigned blocks = blend_grid_units(max_units);
    static int trip = -1;
    if (trip < 0) { const char *e = getenv("GMS_TRIP_BWD"); trip = e ? atoi(e) : 2; }
    const bool invd = a.has_invd && a.dL_dinvd;
    if (fault_mode() == 2 && !invd) {       // negative control (gms_set_fault): its own instantiation
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<false, 2, 2><<<4u * blocks + 64u, WAVE, 0, stream>>>(g, a)));
    } else if (fault_mode() >= 9 && fault_mode() <= 12 && !invd) {      // timing experiments (wrong results): no atomics / L2-resident records / XCD-private atomics
        auto kern = fault_mode() == 9 ? micro_bwd_kernel<false, 2, 9> : fault_mode() == 10 ? micro_bwd_kernel<false, 2, 10>
                  : fault_mode() == 11 ? micro_bwd_kernel<false, 2, 11> : micro_bwd_kernel<false, 2, 12>;
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<4u * blocks + 64u, WAVE, 0, stream>>>(g, a));
    } else {
        auto kern = trip == 1 ? (invd ? micro_bwd_kernel<true, 1, 0> : micro_bwd_kernel<false, 1, 0>)
                  : trip == 4 ? (invd ? micro_bwd_kernel<true, 4, 0> : micro_bwd_kernel<false, 4, 0>)
                              : (invd ? micro_bwd_kernel<true, 2, 0> : micro_bwd_kernel<false, 2, 0>);
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<4u * blocks + 64u, WAVE, 0, stream>>>(g, a));
    }
    GMS_KERNEL_CHECK(debug, stream, "micro_bwd");
    return GMS_OK;
}

}  // namespace gms
