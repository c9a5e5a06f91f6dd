// blend_micro.hip -- micro-tile compositing for gfx950: the wave works on four 4x4 pixel blocks at once, one per DPP row.
//
// Why.  Mesh-bound splats are small (about 14 pixels with alpha >= 1/255 per (Gaussian, tile) instance on the headline
// scene).  A wave that owns an 8x8 quadrant and walks the quadrant's culled list one splat at a time (blend.hip) has 12 of
// its 64 lanes busy and pays a 64-lane reduction per splat.  Here the unit of work is a 4x4 pixel BLOCK with its own
// pre-filtered list: a wave carries four blocks, one per 16-lane row, each row walking ITS OWN list, so one trip of the
// wave advances four (block, splat) pairs (0.6-0.7 trips per instance instead of 1.19, 39 % of the lanes busy instead of 19 %)
// and the ten gradient sums of a splat are reduced inside a 16-lane row with row-local DPP only (29 VALU for four splats
// instead of 26 for one).
//
// One block of 256 threads = one unit (tile, segment of <= 256 list entries).  Since round 5 the unit is RESIDENT in LDS:
//   staging        (unit_stage) thread e gathers the record of entry e ONCE (key -> id -> 48-byte record, coalesced by entry) into
//                  LDS (40 bytes: the extents are only needed by the filter), the launch that first touches a unit computes the
//                  entry's 16-bit block mask -- which of the tile's sixteen 4x4 blocks its {alpha >= 1/255} ellipse touches, exact
//                  ellipse-vs-band intervals (gms_blend.h::block_mask), conservative -- and leaves it in global memory (2 bytes per
//                  instance) for the later launches, and the sixteen per-block byte lists are built in LDS from the masks by ballot
//                  ranks (list order = depth order).  The blocks are ordered by list length and dealt four to a wave.
//   walks          a row's trip reads list[block][pos] -> record[entry] straight from LDS: no queue, no refill, no global access.
//   micro_head / micro_fwd / micro_finalize / micro_bwd
//                  the segment-parallel scheme of blend.hip unchanged -- first segments walked exactly, transmittance
//                  products of the middle segments, exact walk of segments 1.. from the prefix product, partial sums in
//                  order, backward restarted at segment boundaries.
//   backward sums  go to an LDS table indexed by the splat's entry in the unit (ds_add_f32) and leave the block as ONE set of
//                  global atomics per (unit, entry): 5.3 M global float atomics per frame instead of the 16.6 M a per-(block,
//                  splat) atomic would issue.
//
// Positions.  n_contrib holds, per pixel, seg * L + (index in the block's list of that segment) + 1 of the last splat
// applied: monotone along the block's concatenated lists, which is all the backward needs.
//
// What bounds the backward (round 5, measured: DESIGN.md section 7; tools/micro_phases.py, tools/lds_bench.hip).  An LDS float
// atomic costs ~2.7 cycles PER ACTIVE LANE on MI355X -- a 4-row x 10-lane ds_add_f32 occupies the CU's LDS for 45 ns, seventeen
// times a ds_write_b32 of the same lanes -- and a frame issues 16.6 M lane-adds: ~75 us of a serial per-CU resource inside a
// 142-us kernel.  A wave's trip therefore takes ~1 500 cycles (it issues ~180 cycles of VALU) with 10-18 waves per CU in the walk:
// the lgkmcnt wait of the next trip's record reads queues behind the atomics of every other wave.  The per-wave phase stamps say
// where the wave-time goes: walk 47-51 %, waiting at the block's final barrier for the wave with the longest lists 30 %, staging
// 6-14 %, pixel state / segment restart 4-5 %, flush 5 %.  The predecessor of these kernels (rounds 3-4: a 16-entry record queue
// per row in LDS, refilled from L2 through list byte -> id -> record, the byte lists in global memory) ran at the same speed --
// 142 against 144 us, forward 101 against 96 us -- because neither its gathers nor its refills were the bound.
//
// Measured and not kept (same scene; git history holds the code): independent rows -- (unit, block) pieces counting-sorted
// by length over the whole frame, four consecutive pieces per wave (0.58 trips per instance, 94 us of backward without its
// atomics) -- loses the unit-level LDS table (16.6 M global atomics: 144 us) and needs two planning launches; a region key
// in that sort for XCD-local gathers (slower: imbalance between XCDs); fixed-quadrant rows instead of sorted ones (+6 us);
// part of the ten sums by global atomics and part through the LDS table (+4 us); every non-first segment composited locally
// from T = 1 and walked again only where a pixel can stop inside it (no gain: on the headline scene most later segments
// hold a stopping pixel); one gradient table per WAVE with a plain LDS read-add-write instead of ds_add_f32: 191 us
// against 177 us for the atomic kernel padded to the same 52 KB of LDS (3 blocks per CU: at that occupancy latency, not the
// atomic pipe, is the bound), 143 us at its own 25 KB (6 per CU); the trip loop software-pipelined over its LDS traffic: 147-166 us;
// 128-entry units (GMS_SEG_LEN=128): forward +13 us, backward +8 us (twice the segment state); plain LDS stores instead of the
// ds_add (wrong results, timing only): 118 us.
#include <stdlib.h>

#include "gms_common.h"
#include "gms_blend.h"

namespace gms {

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

constexpr int LMAX = 256;          // micro mode: segment length <= 256 (entry index in a byte)

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
    constexpr int U = 8;          // eight segments per step, every load issued before the first use (four: 10.0 us; the deepest tile of the headline frame has 27)
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
    __device__ __forceinline__ Phases(const BlendGrid &g, uint32_t bit = 1024u) : buf(nullptr), on(false)
    {
        if (dbg_on(g, bit) && g.dbg_buf) {
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
#ifndef GMS_REDUCE_BANKMASK
#define GMS_REDUCE_BANKMASK 1
#endif
// One stage of the transposing reduction for a pair of values (a, b): the lanes whose stage bit is clear end up with a + a[partner], the
// others with b + b[partner].  The stage bit is constant over a DPP BANK (four lanes) for the first two stages -- lane bit 3 (banks 2, 3)
// with row_ror:8, lane bit 2 (banks 1, 3) with row_half_mirror -- so two DPP adds with complementary bank masks write the two halves
// -- a disabled bank keeps what the register holds -- instead of two v_cndmask selects and one DPP add: the kernel is bound by VALU issue
// (tools/valu_bench.hip: plain 1.2 ns, DPP 1.76 ns per wave-instruction) and the selects were a third of the reduction.  Same sums bit for
// bit (a + b == b + a).  The s_nop covers the two wait states a DPP read needs behind the VALU write of its source.
#define GMS_BANK_PAIR(D, A, B, CTRL, LO, HI)                                                                                          \
    "v_add_f32_dpp " D ", " A ", " A " " CTRL " row_mask:0xf bank_mask:" LO "\n\tv_add_f32_dpp " D ", " B ", " B " " CTRL " row_mask:0xf bank_mask:" HI "\n\t"
// five pairs (v[k], v[k + 5]) over row_ror:8: lanes 0-7 keep the sums of v[k], lanes 8-15 those of v[k + 5].  NINE: v[9] (the
// inverse-depth weight) does not exist -- the lanes that would carry its sums (8-15 of u[4], then lane 14) are never stored, so the
// instruction that forms them is left out and those lanes hold whatever the register held.
template <bool NINE>
__device__ __forceinline__ void bank_stage_ror8(const float *v, float (&u)[5])
{
    if (NINE)
        asm volatile("s_nop 1\n\t"
                     GMS_BANK_PAIR("%0", "%5", "%10", "row_ror:8", "0x3", "0xc") GMS_BANK_PAIR("%1", "%6", "%11", "row_ror:8", "0x3", "0xc")
                     GMS_BANK_PAIR("%2", "%7", "%12", "row_ror:8", "0x3", "0xc") GMS_BANK_PAIR("%3", "%8", "%13", "row_ror:8", "0x3", "0xc")
                     "v_add_f32_dpp %4, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                     : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4])
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]));
    else
        asm volatile("s_nop 1\n\t"
                     GMS_BANK_PAIR("%0", "%5", "%10", "row_ror:8", "0x3", "0xc") GMS_BANK_PAIR("%1", "%6", "%11", "row_ror:8", "0x3", "0xc")
                     GMS_BANK_PAIR("%2", "%7", "%12", "row_ror:8", "0x3", "0xc") GMS_BANK_PAIR("%3", "%8", "%13", "row_ror:8", "0x3", "0xc")
                     GMS_BANK_PAIR("%4", "%9", "%14", "row_ror:8", "0x3", "0xc")
                     : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4])
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]));
}
// two pairs (a0, b0), (a1, b1) over row_half_mirror: lanes with bit 2 clear keep the sums of a, the others those of b
__device__ __forceinline__ void bank_stage_half_mirror(float a0, float b0, float a1, float b1, float &w0, float &w1)
{
    asm volatile("s_nop 1\n\t"
                 GMS_BANK_PAIR("%0", "%2", "%3", "row_half_mirror", "0x5", "0xa") GMS_BANK_PAIR("%1", "%4", "%5", "row_half_mirror", "0x5", "0xa")
                 : "=&v"(w0), "=&v"(w1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}

template <bool NINE = false>
__device__ __forceinline__ float row_reduce10(const float *v, bool b3, bool b2, bool b1, bool b0)
{
    float u[5];
#if GMS_REDUCE_BANKMASK
    bank_stage_ror8<NINE>(v, u);
    float w0, w1;
    bank_stage_half_mirror(u[0], u[3], u[1], u[4], w0, w1);
    const float w2 = u[2] + dpp_mov<0x141>(u[2]);
    (void)b3; (void)b2;
#else
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float keep = b3 ? v[k + 5] : v[k], send = b3 ? v[k] : v[k + 5];
        u[k] = keep + dpp_mov<0x128>(send);
    }
    const float w0 = (b2 ? u[3] : u[0]) + dpp_mov<0x141>(b2 ? u[0] : u[3]);
    const float w1 = (b2 ? u[4] : u[1]) + dpp_mov<0x141>(b2 ? u[1] : u[4]);
    const float w2 = u[2] + dpp_mov<0x141>(u[2]);
#endif
    const float x0 = (b1 ? w1 : w0) + dpp_mov<0x4E>(b1 ? w0 : w1);
    const float x1 = w2 + dpp_mov<0x4E>(w2);
    return (b0 ? x1 : x0) + dpp_mov<0xB1>(b0 ? x0 : x1);
}

// ------------------------------------------------------------------------------------ the unit, resident in LDS
// largest value of a wave's 64 lanes (non-negative inputs), valid in lane 63: DPP row operations, no LDS
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_max(float v)
{
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ float wave_max_to_lane63(float v)
{
    v = dpp_max<0xB1>(v); v = dpp_max<0x4E>(v); v = dpp_max<0x141>(v); v = dpp_max<0x140>(v);
    v = dpp_max<0x142, 0xa>(v); v = dpp_max<0x143, 0xc>(v);
    return v;
}

template <bool WITHD> struct RecTail { using type = float2; };          // b, 1/depth
template <> struct RecTail<false> { using type = float; };              // b alone (the backward without an inverse-depth gradient)
__device__ __forceinline__ void set_tail(float2 &d, float b, float invd) { d = make_float2(b, invd); }
__device__ __forceinline__ void set_tail(float &d, float b, float) { d = b; }
__device__ __forceinline__ float2 get_tail(const float2 &d) { return d; }
__device__ __forceinline__ float2 get_tail(const float &d) { return make_float2(d, 0.f); }
// WIDE (the forward launches, which have the LDS to spare): the tail at the 16-byte pitch of the other two arrays, so that one shifted
// entry byte addresses all three reads of a walk step (two address instructions fewer per entry; the walks cost what they issue)
struct __attribute__((aligned(16))) WideTail { float2 v; float2 unused; };
__device__ __forceinline__ void set_tail(WideTail &d, float b, float invd) { d.v = make_float2(b, invd); }
__device__ __forceinline__ float2 get_tail(const WideTail &d) { return d.v; }
template <bool WITHD, bool WIDE> struct RecTailSel { using type = typename RecTail<WITHD>::type; };
template <> struct RecTailSel<true, true> { using type = WideTail; };
template <bool WITHD, int LM = LMAX, bool WIDE = false>
struct UnitRecsT {
    static constexpr int CAP = LM;      // entries the image holds (the frame's segment length must not exceed it)
    float4 ra[LM];             // pix.x, pix.y, conic A, conic B
    float4 rb[LM];             // conic C, opacity', r, g
    typename RecTailSel<WITHD, WIDE>::type rc[LM];
    uint8_t list[16][LM];      // per 4x4 block: the entries that reach it, in list (depth) order
    uint16_t ocnt[16];         // list lengths, longest first
    uint8_t order[16];         // ... and whose they are
    uint8_t wcnt4[16][4];      // staging: per block, the hits of each of the four waves (<= 64): one dword per block
};
using UnitRecs = UnitRecsT<true, LMAX, true>;      // (the forward launches)

template <int NE> __device__ __forceinline__ uint32_t list_load(const uint8_t *lst, uint32_t pos);
template <> __device__ __forceinline__ uint32_t list_load<1>(const uint8_t *lst, uint32_t pos) { return lst[pos]; }
template <> __device__ __forceinline__ uint32_t list_load<2>(const uint8_t *lst, uint32_t pos) { return *reinterpret_cast<const uint16_t *>(lst + pos); }
template <> __device__ __forceinline__ uint32_t list_load<4>(const uint8_t *lst, uint32_t pos) { return *reinterpret_cast<const uint32_t *>(lst + pos); }

// Staging.  FILTER = this launch is the first to touch the unit: the thread of an entry computes the entry's block mask from the
// record it has just gathered and leaves it in global memory (g.mmask) for the later launches, which read
// it back instead -- every launch therefore builds IDENTICAL lists (n_contrib holds positions in them).
// `cmax_out` (forward launches): the largest |colour component| of the unit's splats is folded into the tile's maximum
// (ImageState::tile_cmax; one integer atomic per wave), which the backward needs to bound the colour behind a splat.
// Returns the Gaussian id of the thread's entry.
template <bool FILTER, bool WITHD, int LM, bool WIDE>
__device__ __forceinline__ uint32_t unit_stage(const BlendGrid &g, const Unit &u, UnitRecsT<WITHD, LM, WIDE> &S, const SplatRec *rec, uint32_t *cmax_out,
                                               Phases *ph = nullptr)          // (make EXPERIMENTS=1: stamps 3 = records in, 4 = counts exchanged)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cn = u.end - u.beg;
    uint16_t *mm = g.mmask;
    uint32_t mask = 0, id = 0;
    float cm = 0.f;
    if ((uint32_t)tid < cn) {
        id = reinterpret_cast<const uint32_t *>(g.keys)[2 * (size_t)(u.beg + tid)];      // low word of the (depth, id) key
        if (!FILTER) mask = mm[u.beg + tid];
        const SplatRec r = rec[id];
        if (FILTER) { mask = block_mask(r, (float)(u.tx * TILE), (float)(u.ty * TILE)); mm[u.beg + tid] = (uint16_t)mask; }
        S.ra[tid] = r.q0; S.rb[tid] = r.q1; set_tail(S.rc[tid], r.q2.x, r.q2.y);
        cm = fmaxf(fmaxf(fabsf(r.q1.z), fabsf(r.q1.w)), fabsf(r.q2.x));
    } else if (tid < LM) {
        // a row that idles behind the end of its list reads whatever byte lies there: every record it can name must be finite
        S.ra[tid] = make_float4(0.f, 0.f, 0.f, 0.f); S.rb[tid] = make_float4(0.f, 0.f, 0.f, 0.f); set_tail(S.rc[tid], 0.f, 0.f);
    }
    if (cmax_out) {
        cm = wave_max_to_lane63(cm);              // (DPP row operations: no LDS round trips)
        if (lane == 63 && cm > 0.f) atomicMax(cmax_out, __float_as_uint(cm));        // (non-negative floats order like their bits)
    }
    // Lists.  One ballot per block, kept in scalar registers across the barrier; the counts of a wave go to LDS as ONE byte per
    // (block, wave), a block's four bytes in one dword, so that after the barrier lane b reads its block's four counts with a single
    // load and forms the wave's base (the hits of the waves in front) and the block's total with two byte-sum instructions.  (Round 5
    // looped over the waves in front with a dependent LDS byte read each, inside every one of the sixteen block iterations: ~540 VALU
    // and ~180 LDS instructions of staging per wave in a kernel that is bound by VALU issue -- tools/valu_bench.hip.)
    if (GMS_EXPERIMENTS && ph) { __builtin_amdgcn_s_waitcnt(0); ph->mark(3); }
    uint64_t bal[16];
    uint32_t mycnt = 0;                        // lane b < 16: hits of block b among this wave's 64 entries
#pragma unroll
    for (int b = 0; b < 16; b++) {
        bal[b] = __ballot((mask >> b) & 1u);
        const uint32_t n = (uint32_t)__builtin_popcountll(bal[b]);          // (wave-uniform: a scalar register)
        asm("v_writelane_b32 %0, %1, %2" : "+v"(mycnt) : "s"(n), "n"(b));
    }
    if (lane < 16) S.wcnt4[lane][wave] = (uint8_t)mycnt;
    __syncthreads();
    if (GMS_EXPERIMENTS && ph) ph->mark(4);
    const uint32_t w4 = *reinterpret_cast<const uint32_t *>(&S.wcnt4[lane & 15][0]);
    // bytes of the waves in front of this one, summed (v_sad_u8 against zero adds a dword's four bytes)
    const uint32_t basev = __builtin_amdgcn_sad_u8(w4 & ((1u << (8 * wave)) - 1u), 0u, 0u);
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)basev, b);
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[b] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[b], base));
        if ((mask >> b) & 1u) S.list[b][pos] = (uint8_t)tid;
    }
    if (tid < 16) {
        const uint32_t c = __builtin_amdgcn_sad_u8(w4, 0u, 0u);
        uint32_t rank = 0;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0++) {
            const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)c, s0);
            rank += (cs > c || (cs == c && s0 < tid)) ? 1u : 0u;
        }
        S.order[rank] = (uint8_t)tid; S.ocnt[rank] = (uint16_t)c;
    }
    __syncthreads();
    return id;
}

// ---- The compositing step of the forward walks, predicated through EXEC (round 6).
// The long lists decide when these launches end, and a wave issues one instruction per ~2.7 ns whatever its kind: the compiler's form of
// `if (valid && !done && power <= 0 && alpha >= 1/255) { ... if (T' < 1e-4) done = true; else { composite } }` is 27 instructions per entry
// (v_cmp into SGPR pairs, s_and / s_or / s_xor chains, saveexec + branch + restore); this one is 17 (tloc: 7): the three tests narrow EXEC
// themselves (v_cmpx), the stop rule removes its lanes, the body runs under what is left.  The per-lane state that says "this row still
// walks" is the signed `rem` = list entries left INCLUDING this trip's (<= 0: the list is over, the pixel lies outside the image or was
// dead on entry; REM_STOPPED: the stop rule fired); entry E of the trip is live iff E < rem, and a trip ends with rem -= NE.  `last` is
// recorded as E - rem (= index - count: the caller adds the count back).  Same arithmetic in the same order as the plain form.
constexpr uint32_t LAST_NONE = 0x7fffffffu;
constexpr int REM_STOPPED = -0x40000000;
template <int E>
__device__ __forceinline__ void fwd_step_exec(float pw, float al, float cr, float cg, float cb, float cd, float &T, float &C0, float &C1, float &C2,
                                              float &Dp, int &rem, uint32_t &last)
{
    uint64_t sv; float tmp, w;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_lt_i32_e32 vcc, %[e], %[rem]\n\t"
                 "v_cmpx_ge_f32_e32 vcc, 0, %[pw]\n\t"
                 "v_cmpx_le_f32_e32 vcc, %[amin], %[al]\n\t"
                 "v_sub_f32_e32 %[tmp], 1.0, %[al]\n\t"
                 "v_mul_f32_e32 %[tmp], %[T], %[tmp]\n\t"
                 "v_cmp_ngt_f32_e32 vcc, %[tmin], %[tmp]\n\t"          // NOT (T' < 1e-4): the lanes that composite this entry
                 "v_cndmask_b32_e32 %[rem], %[stopped], %[rem], vcc\n\t"
                 "s_and_b64 exec, exec, vcc\n\t"
                 "v_mul_f32_e32 %[w], %[al], %[T]\n\t"
                 "v_fmac_f32_e32 %[C0], %[cr], %[w]\n\t"
                 "v_fmac_f32_e32 %[C1], %[cg], %[w]\n\t"
                 "v_fmac_f32_e32 %[C2], %[cb], %[w]\n\t"
                 "v_fmac_f32_e32 %[Dp], %[cd], %[w]\n\t"
                 "v_mov_b32_e32 %[T], %[tmp]\n\t"
                 "v_sub_u32_e32 %[last], %[e], %[rem]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [tmp] "=&v"(tmp), [w] "=&v"(w), [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [Dp] "+v"(Dp),
                   [rem] "+v"(rem), [last] "+v"(last)
                 : [e] "n"(E), [pw] "v"(pw), [al] "v"(al), [amin] "s"(ALPHA_MIN), [tmin] "s"(T_MIN), [stopped] "v"(REM_STOPPED), [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cb), [cd] "v"(cd)
                 : "vcc");
}
template <int E>
__device__ __forceinline__ void tloc_step_exec(float pw, float al, float &Tl, int rem)
{
    uint64_t sv; float tmp;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_lt_i32_e32 vcc, %[e], %[rem]\n\t"
                 "v_cmpx_ge_f32_e32 vcc, 0, %[pw]\n\t"
                 "v_cmpx_le_f32_e32 vcc, %[amin], %[al]\n\t"
                 "v_sub_f32_e32 %[tmp], 1.0, %[al]\n\t"
                 "v_mul_f32_e32 %[Tl], %[Tl], %[tmp]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [tmp] "=&v"(tmp), [Tl] "+v"(Tl)
                 : [e] "n"(E), [pw] "v"(pw), [al] "v"(al), [amin] "s"(ALPHA_MIN), [rem] "v"(rem)
                 : "vcc");
}
template <int NE, int E = 0> struct StepUnroll {
    template <class F> static __device__ __forceinline__ void run(F &&f) { f(std::integral_constant<int, E>{}); StepUnroll<NE, E + 1>::run(f); }
};
template <int NE> struct StepUnroll<NE, NE> { template <class F> static __device__ __forceinline__ void run(F &&) {} };

template <int NE>
__device__ __forceinline__ void micro_tloc_unit(const BlendGrid &g, const Unit &u, const UnitRecs &S, int phase, int q)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    float *dst = g.seg_state + (size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + p.tid;
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *lst = S.list[p.b];
    float Tl = 1.f;
    (void)phase;
    int rem = p.inside ? (int)cnt : 0;
    static_assert(LMAX % NE == 0, "a trip reads NE list bytes at a multiple of NE below the row's count");
    // once a pixel's segment product is below 1e-4 every later segment starts dead whatever the exact value
    if (__builtin_amdgcn_ballot_w64(rem > 0) != 0ull) do {
        const uint32_t ep = list_load<NE>(lst, 0);
        float al[NE], pw[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const uint32_t ent = (ep >> (8 * e)) & 0xffu;
            const float4 r0 = S.ra[ent];
            const float2 r1 = *reinterpret_cast<const float2 *>(&S.rb[ent]);
            const float dx = r0.x - p.xf, dy = r0.y - p.yf;
            pw[e] = pair_power(r0.z, r0.w, r1.x, dx, dy);
            al[e] = fminf(ALPHA_MAX, r1.y * __expf(pw[e]));
        }
        StepUnroll<NE>::run([&](auto e) { tloc_step_exec<decltype(e)::value>(pw[decltype(e)::value], al[decltype(e)::value], Tl, rem); });
        rem -= NE; lst += NE;
    } while (__builtin_amdgcn_ballot_w64(rem > 0 && !(Tl < T_MIN)) != 0ull);
    *dst = Tl;
}

template <int NE>
__device__ __forceinline__ void micro_fwd_unit(const BlendGrid &g, const BlendFwdOut &o, const Unit &u, const UnitRecs &S, int q)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    const uint32_t cnt = S.ocnt[4 * q + row];
    const uint8_t *lst = S.list[p.b];
    const uint32_t posbase = (uint32_t)u.seg * u.L;

    float T = 1.f;
    {
        // prefix product of the segments in front, four independent loads per step (same left-to-right order)
        const float *tl = g.seg_state + (size_t)u.slot0 * SEG_FLOATS + SEG_TLOC * TILE_PIX + p.tid;
        int k = 0;
        for (; k + 8 <= u.seg; k += 8) {          // (eight loads in flight; the product keeps its left-to-right order)
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = tl[(size_t)(k + j) * SEG_FLOATS];
#pragma unroll
            for (int j = 0; j < 8; j++) T = T * t[j];
        }
        for (; k + 4 <= u.seg; k += 4) {
            const float t0 = tl[(size_t)k * SEG_FLOATS], t1 = tl[(size_t)(k + 1) * SEG_FLOATS];
            const float t2 = tl[(size_t)(k + 2) * SEG_FLOATS], t3 = tl[(size_t)(k + 3) * SEG_FLOATS];
            T = T * t0 * t1 * t2 * t3;
        }
        for (; k < u.seg; k++) T *= tl[(size_t)k * SEG_FLOATS];
    }
    const bool dead_on_entry = T < T_MIN;          // only possible for seg > 0
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last_rel = LAST_NONE;                                        // (index - count of the last entry composited: fwd_step_exec)
    int rem = (!p.inside || dead_on_entry) ? 0 : (int)cnt;               // entries this row still has to look at (fwd_step_exec)

    static_assert(LMAX % NE == 0, "a trip reads NE list bytes at a multiple of NE below the row's count");
    // (some row has entries left => the trip's NE bytes lie inside every row's list image)
    if (__builtin_amdgcn_ballot_w64(rem > 0) != 0ull) do {
        // NE entries of every row per trip: independent alpha evaluations, sequential compositing
        const uint32_t ep = list_load<NE>(lst, 0);
        float al[NE], pw[NE]; float2 cg[NE], cb[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const uint32_t ent = (ep >> (8 * e)) & 0xffu;
            const float4 r0 = S.ra[ent], r1 = S.rb[ent];
            cb[e] = get_tail(S.rc[ent]);
            cg[e] = make_float2(r1.z, r1.w);
            const float dx = r0.x - p.xf, dy = r0.y - p.yf;
            pw[e] = pair_power(r0.z, r0.w, r1.x, dx, dy);
            al[e] = fminf(ALPHA_MAX, r1.y * __expf(pw[e]));
        }
        StepUnroll<NE>::run([&](auto e) {
            constexpr int E = decltype(e)::value;
            fwd_step_exec<E>(pw[E], al[E], cg[E].x, cg[E].y, cb[E].x, cb[E].y, T, C0, C1, C2, Dp, rem, last_rel);
        });
        rem -= NE; lst += NE;
    } while (__builtin_amdgcn_ballot_w64(rem > 0) != 0ull);
    const bool done = !p.inside || dead_on_entry || rem < REM_STOPPED / 2;      // (the plain form's `done`: outside, dead on entry or stopped)
    const uint32_t last = last_rel == LAST_NONE ? 0u : posbase + cnt + 1u + last_rel;
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
__global__ void __launch_bounds__(BLOCK) micro_head_kernel(BlendGrid g, BlendFwdOut o, int phase)
{
    __shared__ UnitRecs S;
    Phases ph(g, 2048u);          // (make EXPERIMENTS=1, GMS_DBG & 2048: start / staged / end of every wave, word 6 = 1 exact walk, 2 products)
    ph.mark(0);
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    const bool walk = u.seg == 0 ? phase <= 0
                                 : (u.nseg > 1 && u.seg != u.nseg - 1 && (phase < 0 || (u.seg < tloc_head(u.L)) == (phase == 0)));
    if (!walk) return;
    if (u.end == u.beg) {          // an empty tile (1 320 of the headline frame's 2 500): background, nothing to stage
        const MPix p = micro_pixel(g, u.tx, u.ty, (int)(threadIdx.x >> 4), (int)(threadIdx.x & 15));
        if (p.inside) {
            const size_t pid = (size_t)p.yi * g.W + p.xi, HW = (size_t)g.W * g.H;
            o.final_T[pid] = 1.f;
            o.n_contrib[pid] = 0u;
            o.out_color[pid] = 0.f + 1.f * o.bg[0];
            o.out_color[HW + pid] = 0.f + 1.f * o.bg[1];
            o.out_color[2 * HW + pid] = 0.f + 1.f * o.bg[2];
            o.out_invdepth[pid] = 0.f;
        }
        return;
    }
    if (u.seg > 0 && phase == 1 && g.tile_dead[u.tile]) {          // products of a dead tile: nothing to walk, empty lists on record
        if (threadIdx.x < u.end - u.beg) g.mmask[u.beg + threadIdx.x] = 0;
        g.seg_state[(size_t)(u.slot0 + u.seg) * SEG_FLOATS + SEG_TLOC * TILE_PIX + threadIdx.x] = 0.f;
        return;
    }
    unit_stage<true>(g, u, S, o.rec, g.tile_cmax + u.tile, &ph);
    ph.mark(1);
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);      // (rotate the sorted groups over the block's waves)
    if (u.seg == 0) micro_fwd_unit<NE>(g, o, u, S, q);
    else micro_tloc_unit<NE>(g, u, S, phase, q);
    ph.value(6, u.seg == 0 ? 1ull : 2ull);
    ph.value(7, ((unsigned long long)u.idx << 40) | ((unsigned long long)(u.end - u.beg) << 28) | ((unsigned long long)u.nseg << 14) | (unsigned long long)u.seg);
    ph.value(2, (unsigned long long)(g.tile_offset[u.tile + 1] - g.tile_offset[u.tile]));
    ph.mark(5);
}

template <int NE>
__global__ void __launch_bounds__(BLOCK) micro_fwd_kernel(BlendGrid g, BlendFwdOut o)
{
    __shared__ UnitRecs S;
    Phases ph(g, 4096u);          // (GMS_DBG & 4096)
    ph.mark(0);
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    if (u.seg == 0) return;
    if (u.seg == u.nseg - 1) unit_stage<true>(g, u, S, o.rec, g.tile_cmax + u.tile, &ph);  // last segments are first touched here
    else unit_stage<false>(g, u, S, o.rec, g.tile_cmax + u.tile, &ph);                        // middle segments: filtered by the first launch
    ph.mark(1);
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);
    micro_fwd_unit<NE>(g, o, u, S, q);
    ph.value(6, u.seg == u.nseg - 1 ? 3ull : 4ull);
    ph.value(7, ((unsigned long long)u.idx << 40) | ((unsigned long long)(u.end - u.beg) << 28) | ((unsigned long long)u.nseg << 14) | (unsigned long long)u.seg);
    ph.value(2, (unsigned long long)(g.tile_offset[u.tile + 1] - g.tile_offset[u.tile]));
    ph.mark(5);
}

// ------------------------------------------------------------------------------------ fixed-point gradient table (round 5)
// An LDS float atomic costs ~2.7 cycles PER ACTIVE LANE on MI355X; an integer one runs at the rate of a plain store (tools/lds_bench.hip:
// ds_add_f32 on 4 rows x 10 lanes 45 ns per CU, ds_add_u64 3.3 ns).  The backward issues 16.6 M lane-adds per frame, so the unit's
// gradient table holds 64-bit FIXED-POINT sums: a partial sum y of field f of entry e is added as round(y * 2^(47 - E)) with an
// exponent E fixed per (unit, field) BEFORE any add -- |y| <= 2^E -- so the sum is exact to 2^(E-48) per add, independent of the
// order of the adds (deterministic), and cannot overflow (at most 16 adds of magnitude <= 2^47).  The bound, for the sum over a 4x4
// block's 16 pixels (gms_blend.h::bwd_step): |q| = |G op dL/dalpha| <= OP ((Cmax + |bg|max) D1 + 5 Dd), where Cmax is the largest
// |colour component| of the tile's splats (the colour behind a splat is a convex combination of those: ImageState::tile_cmax, raised
// by the forward launches), D1 / Dd the tile's largest sum_c |dL/dpixel_c| / |dL/dinvdepth|, 1/depth <= 1/0.2, T <= 1 and
// Tfinal / (1 - alpha) <= T; |dx| <= X, |dy| <= Y, the largest centre-to-corner distances of the unit's splats; the colour weights
// w <= 1; OP the unit's largest opacity.  A factor 2 covers |c - behind| <= 2 Cmax for colours of either sign (colors_precomp; SH colours
// are clamped at zero and need half of it).  Rounding (T a few ulps above 1) is covered by the headroom behind the bound: the conversion
// is exact up to 2^51 and sixteen adds of 2^51 still fit 64 bits, so a bound exceeded 8-fold would be summed exactly all the same.
// Every bound is taken over the
// whole UNIT, so a field's scale 2^(47 - E) is ONE constant per unit for the lane that holds the field, and the walks pay for the
// conversion alone: a float multiply by that constant, a conversion to double, the add of the magic number 1.5 * 2^52 and one integer
// add on the high word -- four instructions per entry.  Measured on the way (HIP events): exponents from every entry's own centre,
// mantissas shifted by hand, 58 bits: 189 us (4 blocks per CU: 32.1 KB of LDS is 152 bytes too many for five); unit-level exponents,
// hand-shifted mantissas: 149 us; float scaling + magic number with the exponent of every entry's own opacity (GMS_FX_ENTRY_OPACITY=1):
// 133 us; this form: 126 us; the float table it replaces: 139-142 us.  The bounds are loose by orders of magnitude on purpose: a value
// 2^23 below its bound still carries 24 bits, and what lies far below is under the 1e-6 floor of the parity criterion.
// (fx_exp, fx_scale_exp, fx_from_float, fx_to_float, FxTile, fx_field_base, fx_field_kind live in gms_blend.h: the test hooks run the very
// conversions on adversarial values, tests/test_gpu_fixed_point.py)

// Backward.  The rows of a wave are aligned at the BOTTOM of their lists: global trip position `pos` is the same list index
// for every row (rows whose list ends below it idle), so a trip's entry bytes are one aligned LDS read per NE entries.
// FIXED (default): the gradient table is 64-bit fixed point (above), integer LDS atomics; FIXED = false keeps the float table of
// rounds 3-4 (ds_add_f32; GMS_BWD_FIXED=0).  DET (deterministic mode, gmsplat.h): the fixed-point table -- integer sums do not depend
// on the order of the adds, so the table IS deterministic (rounds 3-4 kept one float table per wave and added the rows one after
// the other: 55 KB of LDS, 235 us) -- and a flush that stores ONE partial record per instance at the instance's position in the
// sorted list instead of adding to the Gaussian's record with float atomics.
#ifndef GMS_FX_ENTRY_OPACITY
#define GMS_FX_ENTRY_OPACITY 0
#endif
constexpr bool FX_ENTRY_OPACITY = GMS_FX_ENTRY_OPACITY != 0;      // 1: the exponent of every entry's own opacity (3 more instructions per entry, up to 8 bits tighter)
template <bool INVD, int NE, int FAULT, bool DET = false, bool FIXED = true, int LM = LMAX>
__global__ void __launch_bounds__(BLOCK) micro_bwd_kernel(BlendGrid g, BlendBwdArgs a)
{
    static_assert(!DET || FIXED, "the deterministic mode is built on the fixed-point table");
    constexpr int NF = INVD ? 10 : 9;                       // fields per entry of the fixed-point table (GRAD_ID last)
    constexpr uint32_t EMASK = (uint32_t)LM - 1u;           // (LM < 256: a stale list byte behind a row's end must still name a record of the image)
    __shared__ UnitRecsT<INVD, LM> S;
    __shared__ __attribute__((aligned(8))) unsigned char table_mem[FIXED ? LM * NF * 8 : LM * 10 * 4];
    __shared__ uint32_t tile_max[5];                        // FIXED: bits of the unit's largest sum_c |dL/dpixel_c|, |dL/dinvdepth|,
                                                            //        centre-to-corner distances in x and in y
    long long *const fxt = reinterpret_cast<long long *>(table_mem);
    float *const table_all = reinterpret_cast<float *>(table_mem);
    uint32_t *const uid = reinterpret_cast<uint32_t *>(S.rc);          // (written after the walks: the tails are dead by then)
    Phases ph(g);
    ph.mark(0);
    Unit u;
    if (!load_unit_at(g, u, blockIdx.x >> 3, blockIdx.x & 7u)) return;
    if (u.end <= u.beg) return;
    if (u.end - u.beg > (uint32_t)LM) return;               // (host picks the instantiation from the frame's segment length)
    const size_t HW = (size_t)g.W * g.H;
    if (FIXED) {
        for (int k = threadIdx.x; k < LM * NF; k += BLOCK) fxt[k] = 0ll;
        if (threadIdx.x < 5) tile_max[threadIdx.x] = 0u;
    } else {
        for (int k = threadIdx.x; k < LM * 10; k += BLOCK) table_all[k] = 0.f;
    }
    const uint32_t my_id = unit_stage<false, INVD>(g, u, S, a.rec, nullptr);          // (its barriers also order the table clear)
    ph.mark(1);
    const int q = (int)(((threadIdx.x >> 6) + (blockIdx.x >> 3)) & 3u);
    float *const table = table_all;
    const int lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const MPix p = micro_pixel(g, u.tx, u.ty, (int)S.order[4 * q + row], li);
    const size_t pid = (size_t)p.yi * g.W + p.xi;
    const float Tfinal = p.inside ? a.final_T[pid] : 0.f;
    const uint32_t last = p.inside ? a.n_contrib[pid] : 0u;      // seg * L + index + 1 of the last splat this pixel applied
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dinvd = 0.f;
    if (p.inside) {
        dp0 = a.dL_dpix[pid]; dp1 = a.dL_dpix[HW + pid]; dp2 = a.dL_dpix[2 * HW + pid];
        if (INVD) dinvd = a.dL_dinvd[pid];
    }
    const float Tfinal_bgdot = Tfinal * (a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2);
    FxTile fx = {0, 0, 0, 0, 0};
    if (FIXED) {
        // the unit's bounds: largest sum_c |dL/dpixel_c| (the block's 256 lanes hold the tile's 256 pixels), |dL/dinvdepth|, and
        // centre-to-corner distances of its splats (entry threadIdx.x; the records behind the unit's end are zero)
        const float tx0 = (float)(u.tx * TILE), ty0 = (float)(u.ty * TILE);
        const float4 mine = S.ra[threadIdx.x & EMASK];
        const bool real = (uint32_t)threadIdx.x < u.end - u.beg;
        const float m_d1 = wave_max_to_lane63((fabsf(dp0) + fabsf(dp1)) + fabsf(dp2));
        const float m_x = wave_max_to_lane63(real ? fmaxf(fabsf(mine.x - tx0), fabsf(mine.x - tx0 - 15.f)) + 1.f : 1.f);
        const float m_y = wave_max_to_lane63(real ? fmaxf(fabsf(mine.y - ty0), fabsf(mine.y - ty0 - 15.f)) + 1.f : 1.f);
        const float m_dd = INVD ? wave_max_to_lane63(fabsf(dinvd)) : 0.f;
        const float m_op = FX_ENTRY_OPACITY ? 0.f : wave_max_to_lane63(real ? fabsf(S.rb[threadIdx.x & EMASK].y) : 0.f);
        if (lane == 63) {          // (non-negative floats order like their bits; integer LDS atomics run at the rate of stores)
            atomicMax(&tile_max[0], __float_as_uint(m_d1)); if (INVD) atomicMax(&tile_max[1], __float_as_uint(m_dd));
            atomicMax(&tile_max[2], __float_as_uint(m_x)); atomicMax(&tile_max[3], __float_as_uint(m_y));
            if (!FX_ENTRY_OPACITY) atomicMax(&tile_max[4], __float_as_uint(m_op));
        }
        __syncthreads();
        const float D1 = __uint_as_float(tile_max[0]), Dd = INVD ? __uint_as_float(tile_max[1]) : 0.f;
        const float cmax = __uint_as_float(g.tile_cmax[u.tile]);
        const float bgm = fmaxf(fmaxf(fabsf(a.bg[0]), fabsf(a.bg[1])), fabsf(a.bg[2]));
        // (unit-level opacity bound: folded into eK, and a field's scale is then ONE constant per lane and unit)
        fx.eK = fx_exp(32.f * ((cmax + bgm) * D1 + 5.f * Dd)) + (FX_ENTRY_OPACITY ? 0 : fx_exp(__uint_as_float(tile_max[4])));
        fx.eX = fx_exp(__uint_as_float(tile_max[2]));
        fx.eY = fx_exp(__uint_as_float(tile_max[3]));
        fx.eCol = fx_exp(32.f * D1);
        fx.eId = fx_exp(32.f * Dd);
    }
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
            constexpr int RU = 8;          // later segments per step, every load issued before the first use (4: a chain of up to seven rounds in the deepest tile)
            for (int k0 = u.seg + 1; k0 < u.nseg; k0 += RU) {
                float tk[RU], c0[RU], c1[RU], c2[RU], dd[RU];
#pragma unroll
                for (int j = 0; j < RU; j++) {
                    const float *sk = g.seg_state + (size_t)(u.slot0 + min(k0 + j, u.nseg - 1)) * SEG_FLOATS;
                    tk[j] = sk[SEG_TEND * TILE_PIX + p.tid]; c0[j] = sk[SEG_C0 * TILE_PIX + p.tid];
                    c1[j] = sk[SEG_C1 * TILE_PIX + p.tid]; c2[j] = sk[SEG_C2 * TILE_PIX + p.tid];
                    dd[j] = INVD ? sk[SEG_D * TILE_PIX + p.tid] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < RU; j++) {
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
    int fcx, fcy, fkind;
    fx_field_kind(afield, fcx, fcy, fkind);
    const int fxbase = fx_field_base(fx, fcx, fcy, fkind);          // (the lane's field: a constant of the unit)
    const FxScale fxs = fx_prepare(fx_scale_exp(fxbase, 126u));      // ... and with it the lane's scale 2^k and saturation bound

    // back to front: the trip at list position pos handles entry pos of every row's list that reaches it
    for (int g0 = (int)(((maxtop + NE - 1u) / NE) * NE) - NE; g0 >= 0; g0 -= NE) {
        const uint32_t ep = list_load<NE>(lst, (uint32_t)g0);
        bool act[NE]; float dx[NE], dy[NE], G[NE], al[NE]; float4 r1[NE]; float2 r2[NE]; uint32_t se[NE];
        bool anyact = false;
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const uint32_t pos = (uint32_t)g0 + (uint32_t)(NE - 1 - e);          // descending within the trip
            se[e] = (ep >> (8 * (NE - 1 - e))) & (0xffu & EMASK);
            const float4 r0 = S.ra[se[e]];
            r1[e] = S.rb[se[e]]; r2[e] = get_tail(S.rc[se[e]]);
            dx[e] = r0.x - p.xf; dy[e] = r0.y - p.yf;
            const float pw = pair_power(r0.z, r0.w, r1[e].x, dx[e], dy[e]);
            G[e] = __expf(pw);
            al[e] = fminf(ALPHA_MAX, r1[e].y * G[e]);
            // composited by this pixel iff it lies below lrel (lrel <= the length of the row's list)
            act[e] = pos < lrel && pw <= 0.f && al[e] >= ALPHA_MIN;
            anyact = anyact || act[e];
        }
        if (!__any(anyact)) continue;
        // the recurrence runs entry by entry; the NE row reductions that follow are independent DPP chains the compiler may interleave
        // (each stage waits two states for its operands: with one chain per branch region the slots stayed empty), and the table adds come last
        float y[NE];
        {
            float v[NE][10];
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const float4 q2 = make_float4(r2[e].x, r2[e].y, 0.f, 0.f);
                bwd_step<INVD>(st8, act[e], r1[e], q2, dx[e], dy[e], G[e], al[e], dp0, dp1, dp2, dinvd, Tfinal_bgdot, v[e]);
            }
#pragma unroll
            for (int e = 0; e < NE; e++) y[e] = row_reduce10<!INVD>(v[e], b3, b2, b1, b0);
        }
        // a row with no active pixel for an entry sums exact zeros: nothing to add (and its entry byte may be stale)
        if (FIXED) {
            long long val[NE];
#pragma unroll
            for (int e = 0; e < NE; e++)          // (a pair is only ever active on a positive opacity: its bits >> 23 are its exponent)
                val[e] = FX_ENTRY_OPACITY ? fx_from_float(y[e], fx_scale_exp(fxbase, fkind == 0 ? __float_as_uint(r1[e].y) >> 23 : 126u))
                                          : fx_from_float(y[e], fxs);
            // (adding without the test for zero was measured: 141 us against 133 -- the zeros of the idle rows are atomics too)
#pragma unroll
            for (int e = 0; e < NE; e++)
                if (alane && y[e] != 0.f)
                    atomicAdd(reinterpret_cast<unsigned long long *>(fxt) + se[e] * (uint32_t)NF + (uint32_t)afield, (unsigned long long)val[e]);
        } else {
#pragma unroll
            for (int e = 0; e < NE; e++)
                if (alane && y[e] != 0.f) atomicAdd(&table[se[e] * 10u + (uint32_t)afield], y[e]);
        }
    }
    }
    ph.mark(3);
    __syncthreads();
    ph.mark(4);
    if ((uint32_t)threadIdx.x < cn) uid[threadIdx.x] = my_id;          // (over the record tails, which no walk reads any more)
    __syncthreads();
    // flush.  Float-atomics mode: NF consecutive lanes per entry, one per field of its 64-byte record (one cache line), BLOCK / NF entries
    // per step -- 28 (27 lanes of 256 idle) instead of the 16 of a 16-lanes-per-entry layout whose lanes NF..15 had nothing to add: ten
    // steps instead of sixteen for a full unit (round 6: the launch is bound by VALU issue and this loop ran on every wave).
    if (FIXED && !DET) {
        constexpr int EPS = BLOCK / NF;                       // entries per step
        const int f = (int)threadIdx.x % NF;
        int cx, cy, kind;
        fx_field_kind(f, cx, cy, kind);
        const int kexp = fx_scale_exp(fx_field_base(fx, cx, cy, kind), 126u);
        if ((int)threadIdx.x < EPS * NF) {
            for (uint32_t e = threadIdx.x / NF; e < cn; e += EPS) {
                const long long sv = fxt[e * (uint32_t)NF + (uint32_t)f];
                if (sv != 0ll)
                    unsafeAtomicAdd(a.accum + (size_t)uid[e] * GRAD_STRIDE + f,
                                    fx_to_float(sv, FX_ENTRY_OPACITY && kind == 0 ? fx_scale_exp(fx_field_base(fx, cx, cy, kind), __float_as_uint(S.rb[e].y) >> 23) : kexp));
            }
        }
    } else {
        const int f = threadIdx.x & 15;
        int cx, cy, kind;
        fx_field_kind(f, cx, cy, kind);
        for (uint32_t e = threadIdx.x >> 4; e < cn; e += BLOCK / 16) {
            if (DET) {
                // every instance of the unit gets its record (zeros included: the buffer is not cleared between frames)
                const long long sv = f < NF ? fxt[e * (uint32_t)NF + (uint32_t)f] : 0ll;
                a.part[(size_t)(u.beg + e) * GRAD_STRIDE + f] =
                    sv != 0ll ? fx_to_float(sv, fx_scale_exp(fx_field_base(fx, cx, cy, kind), (FX_ENTRY_OPACITY && kind == 0) ? __float_as_uint(S.rb[e].y) >> 23 : 126u)) : 0.f;
            } else if (FIXED) {
                if (f < NF) {
                    const long long sv = fxt[e * (uint32_t)NF + (uint32_t)f];
                    if (sv != 0ll)
                        unsafeAtomicAdd(a.accum + (size_t)uid[e] * GRAD_STRIDE + f,
                                        fx_to_float(sv, fx_scale_exp(fx_field_base(fx, cx, cy, kind), (FX_ENTRY_OPACITY && kind == 0) ? __float_as_uint(S.rb[e].y) >> 23 : 126u)));
                }
            } else if (f < 10) {
                const float y = table[e * 10u + (uint32_t)f];
                if (y != 0.f) unsafeAtomicAdd(a.accum + (size_t)uid[e] * GRAD_STRIDE + f, y);
            }
        }
    }
    ph.value(7, ((unsigned long long)u.idx << 40) | ((unsigned long long)(u.end - u.beg) << 28) | ((unsigned long long)u.nseg << 14) | (unsigned long long)u.seg);
    ph.mark(5);
}

// ------------------------------------------------------------------------------------ host
int32_t launch_micro_forward(const BlendGrid &g_in, const BlendFwdOut &o, uint32_t max_units, bool debug, hipStream_t stream)
{
    BlendGrid g = g_in;
    experiment_switches(g, 2048u | 4096u, stream);          // (make EXPERIMENTS=1 only: per-wave stamps of micro_head / micro_fwd)
    static int deep_env = -2;
    if (deep_env == -2) { const char *e = getenv("GMS_DEEP"); deep_env = e ? atoi(e) : -1; }
    const bool deep = deep_env >= 0 ? deep_env != 0 : g.capacity > 512ull * (uint64_t)g.T;
    const unsigned blocks = blend_grid_units(max_units);
    static int trip = -1;
    if (trip < 0) { const char *e = getenv("GMS_TRIP"); trip = e ? atoi(e) : 4; }      // (forward: 4 entries per trip; 2: -2 % it/s)
    auto head = trip == 4 ? micro_head_kernel<4> : (trip == 1 ? micro_head_kernel<1> : micro_head_kernel<2>);
    auto fwd2 = trip == 4 ? micro_fwd_kernel<4> : (trip == 1 ? micro_fwd_kernel<1> : micro_fwd_kernel<2>);
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
    static int fixed = -1;              // GMS_BWD_FIXED=0: the float LDS table of rounds 3-4 (ds_add_f32) instead of the 64-bit fixed-point one
    if (fixed < 0) { const char *e = getenv("GMS_BWD_FIXED"); fixed = e ? (atoi(e) != 0) : 1; }
    if (a.part) {                           // deterministic mode (gmsplat.h): the fixed-point table, per-instance partial records
        if (invd) GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<true, 2, 0, true><<<blocks, BLOCK, 0, stream>>>(g, a)));
        else GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<false, 2, 0, true><<<blocks, BLOCK, 0, stream>>>(g, a)));
    } else if (fault_mode() == 2 && !invd) {       // negative control (gms_set_fault): its own instantiation
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<false, 2, 2><<<blocks, BLOCK, 0, stream>>>(g, a)));
    } else if (!fixed) {
        auto kern = invd ? micro_bwd_kernel<true, 2, 0, false, false> : micro_bwd_kernel<false, 2, 0, false, false>;
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<blocks, BLOCK, 0, stream>>>(g, a));
    } else {
#if GMS_EXPERIMENTS
        // Occupancy experiment (make EXPERIMENTS=1, GMS_SEG_LEN=128): a 128-entry unit image is 16 KB of LDS -> 8 blocks per CU instead
        // of 5.  Measured (round 6, profiles/r06c_*): 126.2 us against 125.7 -- 8 192 resident waves instead of 5 120, the same ~2 400 of
        // them in the walk, every phase of a wave proportionally slower.  The kernel is bound by VALU issue, not by latency or occupancy.
        if (seg_len_forced() != 0 && seg_len_forced() <= 128u && !invd) {
            GMS_LAUNCH(GMS_K_BLEND_BWD, stream, (micro_bwd_kernel<false, 2, 0, false, true, 128><<<blocks, BLOCK, 0, stream>>>(g, a)));
            GMS_KERNEL_CHECK(debug, stream, "micro_bwd");
            return GMS_OK;
        }
#endif
        auto kern = trip == 1 ? (invd ? micro_bwd_kernel<true, 1, 0> : micro_bwd_kernel<false, 1, 0>)
                  : trip == 4 ? (invd ? micro_bwd_kernel<true, 4, 0> : micro_bwd_kernel<false, 4, 0>)
                              : (invd ? micro_bwd_kernel<true, 2, 0> : micro_bwd_kernel<false, 2, 0>);
        static int lds_pad = -1;            // (make EXPERIMENTS=1 only) GMS_LDS_PAD: bytes of unused dynamic LDS per block, to time other occupancies
        if (lds_pad < 0) { const char *e = getenv("GMS_LDS_PAD"); lds_pad = (GMS_EXPERIMENTS && e) ? atoi(e) : 0; }
        GMS_LAUNCH(GMS_K_BLEND_BWD, stream, kern<<<blocks, BLOCK, (size_t)lds_pad, stream>>>(g, a));
    }
    GMS_KERNEL_CHECK(debug, stream, "micro_bwd");
    return GMS_OK;
}

}  // namespace gms
