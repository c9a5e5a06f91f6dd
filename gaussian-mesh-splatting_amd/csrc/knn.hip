// knn.hip -- exact 3-nearest-neighbour mean squared distance on gfx950 (replaces the reference's un-vendored
// `simple_knn._C.distCUDA2`, .gitmodules:1-3; call sites scene/gaussian_model.py:134 and
// games/flat_splatting/scene/flat_gaussian_model.py:47: initial Gaussian scales = sqrt(mean dist^2 to the 3 NN)).
//
// Uniform-grid search instead of the upstream Morton-order box walk: points are counting-sorted into G^3 cells
// (G from N so that a cell holds ~4 points), then every point searches Chebyshev shells of cells around its own
// cell until its third-best distance is no larger than the distance to the boundary of the searched block --
// exact, independent of the order of points inside a cell.  All phases are HBM/atomic bound and run once at
// model initialisation (not on the per-iteration hot path).
#include "gms_common.h"

namespace gms {

__device__ __forceinline__ int float_to_ordered(float f)
{
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// Lives at the head of the workspace; filled on the device so the host never waits for the bounding box.
struct KnnHeader {
    int bbox_i[6];        // ordered-int encoded min xyz, max xyz
    int G[3];             // cells per axis
    int ncell;
    float lo[3], h[3], inv_h[3];
    float slack;          // absolute rounding allowance of a cell-boundary coordinate
};

struct CellMap {
    float lo[3], inv_h[3], h[3], slack;
    int G[3];
    __device__ __forceinline__ explicit CellMap(const KnnHeader *hd)
    {
        slack = hd->slack;
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = hd->lo[k]; inv_h[k] = hd->inv_h[k]; h[k] = hd->h[k]; G[k] = hd->G[k]; }
    }
    __device__ __forceinline__ void cell_of(const float p[3], int c[3]) const
    {
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] = min(G[k] - 1, max(0, (int)((p[k] - lo[k]) * inv_h[k])));
    }
    __device__ __forceinline__ uint32_t flat(int x, int y, int z) const { return (uint32_t)((z * G[1] + y) * G[0] + x); }
};

__global__ void knn_init_kernel(KnnHeader *hd, uint32_t *cell_count, int max_cell)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3) hd->bbox_i[i] = 0x7fffffff;
    else if (i < 6) hd->bbox_i[i] = (int)0x80000000;
    for (int c = i; c <= max_cell; c += gridDim.x * blockDim.x) cell_count[c] = 0;
}

__global__ void __launch_bounds__(BLOCK) knn_bbox_kernel(int N, const float *pts, KnnHeader *hd)
{
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < N; i += gridDim.x * BLOCK)
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off >= 1; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&hd->bbox_i[k], float_to_ordered(mn[k]));
            atomicMax(&hd->bbox_i[3 + k], float_to_ordered(mx[k]));
        }
    }
}

// One thread: cubic cells of edge h so that the occupied box holds about N/4 cells; an axis thinner than h
// (planar or collinear clouds) collapses to a single cell and the budget goes to the other axes.
__global__ void knn_grid_kernel(int N, KnnHeader *hd, int max_cell)
{
    float lo[3], ext[3];
    for (int k = 0; k < 3; k++) {
        lo[k] = ordered_to_float(hd->bbox_i[k]);
        ext[k] = fmaxf(ordered_to_float(hd->bbox_i[3 + k]) - lo[k], 0.f);
    }
    const float target = fminf(fmaxf((float)N / 4.f, 1.f), (float)max_cell);
    bool flat[3] = {false, false, false};
    float h = 0.f;
    for (int round = 0; round < 3; round++) {
        float vol = 1.f; int dims = 0;
        for (int k = 0; k < 3; k++) if (!flat[k]) { vol *= ext[k]; dims++; }
        if (dims == 0) break;
        h = powf(vol / target, 1.f / dims);
        bool changed = false;
        for (int k = 0; k < 3; k++) if (!flat[k] && !(ext[k] > h)) { flat[k] = true; changed = true; }
        if (!changed) break;
    }
    int G[3];
    for (int iter = 0; iter < 64; iter++) {
        long long prod = 1;
        for (int k = 0; k < 3; k++) {
            G[k] = (flat[k] || !(h > 0.f)) ? 1 : (int)fminf(ceilf(ext[k] / h), 1024.f);
            if (G[k] < 1) G[k] = 1;
            prod *= G[k];
        }
        if (prod <= max_cell) break;
        h *= 1.1f;
    }
    if ((long long)G[0] * G[1] * G[2] > max_cell) G[0] = G[1] = G[2] = 1;
    for (int k = 0; k < 3; k++) {
        hd->G[k] = G[k];
        hd->lo[k] = lo[k];
        const float e = fmaxf(ext[k], 1e-30f);
        hd->h[k] = e / G[k];
        hd->inv_h[k] = G[k] / e;
    }
    hd->ncell = G[0] * G[1] * G[2];
    float big = 0.f;
    for (int k = 0; k < 3; k++) big = fmaxf(big, fmaxf(fabsf(lo[k]), fabsf(lo[k] + ext[k])));
    hd->slack = 4.f * 1.1920929e-7f * big;
}

__global__ void __launch_bounds__(BLOCK) knn_count_kernel(int N, const float *pts, const KnnHeader *hd, uint32_t *cell_count,
                                                          uint32_t *point_cell)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const CellMap m(hd);
    const float p[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    int c[3];
    m.cell_of(p, c);
    const uint32_t cell = m.flat(c[0], c[1], c[2]);
    point_cell[i] = cell;
    atomicAdd(&cell_count[cell], 1u);
}

// exclusive scan of cell_count[0..ncell) into cell_start[0..ncell] by one block
__global__ void __launch_bounds__(1024) knn_scan_kernel(const uint32_t *cell_count, uint32_t *cell_start, uint32_t *cell_cursor,
                                                        const KnnHeader *hd)
{
    __shared__ uint32_t wave_tot[16];
    const int ncell = hd->ncell;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (ncell + 1023) / 1024;
    const int b = min(ncell, tid * per), e = min(ncell, b + per);
    uint32_t s = 0;
    for (int c = b; c < e; c++) s += cell_count[c];
    uint32_t run = s;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)run, d); if (lane >= d) run += x; }
    if (lane == 63) wave_tot[wave] = run;
    __syncthreads();
    if (wave == 0) {
        uint32_t v = lane < 16 ? wave_tot[lane] : 0u;
        for (int d = 1; d < 16; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v += x; }
        if (lane < 16) wave_tot[lane] = v;
    }
    __syncthreads();
    uint32_t pre = run - s + (wave > 0 ? wave_tot[wave - 1] : 0u);
    for (int c = b; c < e; c++) { cell_start[c] = pre; cell_cursor[c] = pre; pre += cell_count[c]; }
    if (tid == 0) cell_start[ncell] = wave_tot[15];
}

__global__ void __launch_bounds__(BLOCK) knn_scatter_kernel(int N, const float *pts, const uint32_t *point_cell, uint32_t *cell_cursor,
                                                            float4 *sorted)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const uint32_t slot = atomicAdd(&cell_cursor[point_cell[i]], 1u);
    sorted[slot] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}

__global__ void __launch_bounds__(BLOCK) knn_query_kernel(int N, const float *pts, const KnnHeader *hd, const uint32_t *cell_start,
                                                          const float4 *sorted, float *out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const CellMap m(hd);
    const float p[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    int c[3];
    m.cell_of(p, c);
    const int rmax = max(m.G[0], max(m.G[1], m.G[2]));
    float d0 = 3.4e38f, d1 = 3.4e38f, d2 = 3.4e38f;
    for (int r = 0; r <= rmax; r++) {
        const int x0 = max(0, c[0] - r), x1 = min(m.G[0] - 1, c[0] + r);
        const int y0 = max(0, c[1] - r), y1 = min(m.G[1] - 1, c[1] + r);
        const int z0 = max(0, c[2] - r), z1 = min(m.G[2] - 1, c[2] + r);
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                const bool shell_zy = abs(z - c[2]) == r || abs(y - c[1]) == r;
                // interior cells (searched in an earlier round) are skipped: only the two end cells of the row remain
                const int step = shell_zy ? 1 : max(1, 2 * r);
                for (int x = c[0] - r; x <= c[0] + r; x += step) {
                    if (x < x0 || x > x1) continue;
                    const uint32_t cell = m.flat(x, y, z);
                    for (uint32_t s = cell_start[cell], e = cell_start[cell + 1]; s < e; s++) {
                        const float4 q = sorted[s];
                        if (__float_as_int(q.w) == i) continue;
                        const float dx = q.x - p[0], dy = q.y - p[1], dz = q.z - p[2];
                        const float d = dx * dx + dy * dy + dz * dz;
                        if (d < d2) {
                            if (d < d1) { d2 = d1; if (d < d0) { d1 = d0; d0 = d; } else d1 = d; }
                            else d2 = d;
                        }
                    }
                }
            }
        // distance from p to the boundary of the searched block; a side clipped by the grid has nothing beyond it
        float margin = 3.4e38f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (c[k] - r > 0) margin = fminf(margin, p[k] - (m.lo[k] + (c[k] - r) * m.h[k]));
            if (c[k] + r < m.G[k] - 1) margin = fminf(margin, (m.lo[k] + (c[k] + r + 1) * m.h[k]) - p[k]);
        }
        if (margin >= 3.0e38f) break;                       // whole grid searched
        margin = fmaxf(0.f, margin * 0.9999f - m.slack);     // conservative against the rounding of the cell boundaries
        if (d2 <= margin * margin) break;
    }
    const int k = min(3, N - 1);
    out[i] = k >= 3 ? (d0 + d1 + d2) / 3.f : k == 2 ? (d0 + d1) / 2.f : k == 1 ? d0 : 0.f;
}

}  // namespace gms

using namespace gms;

static size_t knn_max_cells(int N)
{
    size_t m = (size_t)(N > 0 ? N : 1) / 2 + 64;      // about twice the N/4 target: room for the ceil() per axis
    return m > (1u << 22) ? (1u << 22) : m;
}

extern "C" size_t gms_knn_workspace_bytes(int32_t N)
{
    const size_t mc = knn_max_cells(N), n = (size_t)(N > 0 ? N : 1);
    return 256 + align_up((mc + 1) * 4, 256) * 3 + align_up(n * 4, 256) + align_up(n * 16, 256);
}

extern "C" int32_t gms_knn_mean_dist2(int32_t N, const float *points, float *out, void *workspace, size_t workspace_bytes,
                                      void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    set_error("%s", "");
    if (N < 0 || (N > 0 && (!points || !out || !workspace))) { set_error("gms_knn_mean_dist2: invalid argument"); return GMS_ERR_INVALID_ARGUMENT; }
    if (N == 0) return GMS_OK;
    if (workspace_bytes < gms_knn_workspace_bytes(N)) { set_error("gms_knn_mean_dist2: workspace too small"); return GMS_ERR_CAPACITY; }
    const size_t mc = knn_max_cells(N);
    static_assert(sizeof(KnnHeader) <= 256, "header slot");
    char *w = (char *)workspace;
    KnnHeader *hd = (KnnHeader *)w;               w += 256;
    uint32_t *cell_count = (uint32_t *)w;         w += align_up((mc + 1) * 4, 256);
    uint32_t *cell_start = (uint32_t *)w;         w += align_up((mc + 1) * 4, 256);
    uint32_t *cell_cursor = (uint32_t *)w;        w += align_up((mc + 1) * 4, 256);
    uint32_t *point_cell = (uint32_t *)w;         w += align_up((size_t)N * 4, 256);
    float4 *sorted = (float4 *)w;
    const unsigned nb = (unsigned)((N + BLOCK - 1) / BLOCK);
    knn_init_kernel<<<256, 256, 0, stream>>>(hd, cell_count, (int)mc);
    knn_bbox_kernel<<<nb < 1024 ? nb : 1024, BLOCK, 0, stream>>>(N, points, hd);
    knn_grid_kernel<<<1, 1, 0, stream>>>(N, hd, (int)mc);
    knn_count_kernel<<<nb, BLOCK, 0, stream>>>(N, points, hd, cell_count, point_cell);
    knn_scan_kernel<<<1, 1024, 0, stream>>>(cell_count, cell_start, cell_cursor, hd);
    knn_scatter_kernel<<<nb, BLOCK, 0, stream>>>(N, points, point_cell, cell_cursor, sorted);
    knn_query_kernel<<<nb, BLOCK, 0, stream>>>(N, points, hd, cell_start, sorted, out);
    GMS_KERNEL_CHECK(0, stream, "knn");
    return GMS_OK;
}
