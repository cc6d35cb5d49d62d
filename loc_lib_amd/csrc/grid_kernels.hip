// loc_lib_amd/csrc/grid_kernels.hip — exact k-NN over the dense cell grid (LOCGPU_SEARCH_GRID_EXACT), gfx950.
//
// Equals KdTree::GetClosestPoint with approximate_ = false (kdtree.cpp:147-167 with the exact NeedExpand branch :227-235): the
// k leaves with the smallest float32 dist² (same Eigen reduction order, no FMA), ascending; a candidate replaces the current
// k-th only when strictly smaller (kdtree.cpp:207). Only the winner among EXACTLY equal distances can differ (visit order).
//
// One thread per query. Pass R examines the cells at Chebyshev distance ≤ R from the query's cell that earlier passes have not
// seen, row by row: x is the fastest grid axis, so a row's cells are ONE contiguous range of the cell-sorted leaf array
// (two coalescible cell_start loads, then 16-byte point loads). The search stops when the k-th distance is within the
// examined block: top ≤ (distance to the nearest unexamined face − slack)², where faces on the grid boundary do not count
// (no leaf lies beyond them). `slack` absorbs the float32 rounding of the point→cell assignment. Queries that are not
// settled by the first pass (default: the 3×3×3 block only) are appended to a list: a query that needs many rings would
// otherwise stall the 63 other lanes of its wave. A second, compacted pass continues them up to `max_ring2` rings with
// homogeneous waves; what is still open after that (queries many cells away from every leaf) is answered by the exact KD-tree
// kernel (icp_search_redo_kernel with alpha = 1). LOCGPU_GRID_RINGS / LOCGPU_GRID_RINGS2 override the two limits.
#include "grid_kernels.hpp"
#include "icp_kernels.hpp"

#include <cstdlib>

namespace locgpu {


struct GridDev {
    const uint32_t* cell_start;
    const float4* pts;
    int nx, ny, nz;
    float ox, oy, oz, cell, inv_cell, slack;
    int max_ring;   // rings examined by the first pass (every query)
    int max_ring2;  // rings examined by the second pass (the compacted list of queries the first pass left open)
};

// Returns true when the query must be answered by the tree kernel instead.
template <int K>
__device__ __forceinline__ bool grid_knn(const GridDev& g, float qx, float qy, float qz, SortedSet<K>& set, int kMaxRing) {
    set.init();
    if (!(qx == qx && qy == qy && qz == qz)) return true;  // NaN query: the tree kernel reproduces the reference's behaviour
    const float fx = floorf((qx - g.ox) * g.inv_cell), fy = floorf((qy - g.oy) * g.inv_cell), fz = floorf((qz - g.oz) * g.inv_cell);
    // far outside the grid: no pass below could reach a leaf
    if (fx < -(float)kMaxRing - 1.f || fy < -(float)kMaxRing - 1.f || fz < -(float)kMaxRing - 1.f || fx > (float)(g.nx + kMaxRing) ||
        fy > (float)(g.ny + kMaxRing) || fz > (float)(g.nz + kMaxRing))
        return true;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    int prev = -1;
    int R0 = 1;
    {
        // first ring, specialised: the 18 range bounds of the nine rows are loaded together (one memory round trip),
        // then the nine contiguous point runs are scanned.
        const int a = max(cx - 1, 0), b = min(cx + 1, g.nx - 1);
        uint32_t rs[9], re[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
            const bool ok = a <= b && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz;
            const size_t row = ok ? ((size_t)zz * g.ny + yy) * g.nx : 0;
            rs[r] = ok ? g.cell_start[row + a] : 0u;
            re[r] = ok ? g.cell_start[row + b + 1] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            for (uint32_t pi = rs[r]; pi < re[r]; ++pi) {
                const float4 p = g.pts[pi];
                const float dx = qx - p.x, dy2 = qy - p.y, dz2 = qz - p.z;
                const float dis2 = dx * dx + (dy2 * dy2 + dz2 * dz2);
                if (dis2 < set.top()) set.insert(dis2, __float_as_uint(p.w));
            }
        }
        prev = 0;  // only used from R = 2 on (after `prev = R` below)
    }
    for (int R = R0; R <= kMaxRing; ++R) {
        if (R > 1)
        for (int dz = -R; dz <= R; ++dz) {
            const int zz = cz + dz;
            if (zz < 0 || zz >= g.nz) continue;
            for (int dy = -R; dy <= R; ++dy) {
                const int yy = cy + dy;
                if (yy < 0 || yy >= g.ny) continue;
                const int rowcheb = max(abs(dy), abs(dz));
                const size_t row = ((size_t)zz * g.ny + yy) * g.nx;
                // rows already swept by earlier passes only need their two new end segments
                const int nseg = rowcheb > prev ? 1 : 2;
                for (int sgi = 0; sgi < nseg; ++sgi) {
                    int a, b;
                    if (nseg == 1) { a = cx - R; b = cx + R; }
                    else if (sgi == 0) { a = cx - R; b = cx - prev - 1; }
                    else { a = cx + prev + 1; b = cx + R; }
                    a = max(a, 0);
                    b = min(b, g.nx - 1);
                    if (a > b) continue;
                    const uint32_t s = g.cell_start[row + a], e = g.cell_start[row + b + 1];
                    for (uint32_t pi = s; pi < e; ++pi) {
                        const float4 p = g.pts[pi];
                        const float dx = qx - p.x, dy2 = qy - p.y, dz2 = qz - p.z;
                        const float dis2 = dx * dx + (dy2 * dy2 + dz2 * dz2);  // Eigen squaredNorm order, no FMA
                        if (dis2 < set.top()) set.insert(dis2, __float_as_uint(p.w));
                    }
                }
            }
        }
        // nearest face of the examined block that still has unexamined cells behind it
        float dmin = __builtin_inff();
        bool open_face = false;
        if (cx - R > 0) { dmin = fminf(dmin, qx - (g.ox + (float)(cx - R) * g.cell)); open_face = true; }
        if (cx + R < g.nx - 1) { dmin = fminf(dmin, (g.ox + (float)(cx + R + 1) * g.cell) - qx); open_face = true; }
        if (cy - R > 0) { dmin = fminf(dmin, qy - (g.oy + (float)(cy - R) * g.cell)); open_face = true; }
        if (cy + R < g.ny - 1) { dmin = fminf(dmin, (g.oy + (float)(cy + R + 1) * g.cell) - qy); open_face = true; }
        if (cz - R > 0) { dmin = fminf(dmin, qz - (g.oz + (float)(cz - R) * g.cell)); open_face = true; }
        if (cz + R < g.nz - 1) { dmin = fminf(dmin, (g.oz + (float)(cz + R + 1) * g.cell) - qz); open_face = true; }
        if (!open_face) return set.n < K ? true : false;  // every leaf was examined (fewer than K leaves: tree kernel answers like the reference)
        const float safe = dmin - g.slack;
        if (set.n == K && safe > 0.f && set.top() <= safe * safe) return false;
        prev = R;
    }
    return true;
}

template <int K>
__global__ __launch_bounds__(kBlock) void icp_search_grid_kernel(GridDev g, const float4* __restrict__ src, const int* __restrict__ counts,
                                                                 const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                 int skip_nonfinite, uint32_t* __restrict__ redo_list, unsigned int* __restrict__ redo_count,
                                                                 unsigned long long* __restrict__ search_stats) {
    const int scan = blockIdx.y;
    if (st[scan].done) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= counts[scan]) return;
    const size_t gi = (size_t)scan * max_n + i;
    const float4 p = src[gi];
    if (skip_nonfinite && !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {  // pcl::isFinite, icp cpp:64 (P2P only)
#pragma unroll
        for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = kInvalidSlot;
        return;
    }
    if (search_stats) atomicAdd(&search_stats[0], 1ull);
    const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
    SortedSet<K> set;
    if (grid_knn<K>(g, (float)qs.x, (float)qs.y, (float)qs.z, set, g.max_ring)) {
        redo_list[atomicAdd(redo_count, 1u)] = (uint32_t)gi;
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = set.id[j];
    }
}

// Second pass: the queries the first pass could not settle, compacted, so that every lane of a wave has the same kind of
// (long) search. Persistent-style 1-D grid over list_in; what is still open after max_ring2 rings goes to list_out (tree kernel).
template <int K>
__global__ __launch_bounds__(kBlock) void icp_search_grid_pass2_kernel(GridDev g, const float4* __restrict__ src, const PoseState* __restrict__ st,
                                                                       uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                       const uint32_t* __restrict__ list_in, const unsigned int* __restrict__ n_in,
                                                                       uint32_t* __restrict__ list_out, unsigned int* __restrict__ n_out) {
    const unsigned int n = *n_in;
    for (unsigned int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        const size_t gi = list_in[r];
        const int scan = (int)(gi / (size_t)max_n);
        const float4 p = src[gi];
        const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
        SortedSet<K> set;
        if (grid_knn<K>(g, (float)qs.x, (float)qs.y, (float)qs.z, set, g.max_ring2)) {
            list_out[atomicAdd(n_out, 1u)] = (uint32_t)gi;
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = set.id[j];
        }
    }
}

// Plain exact k-NN over given queries (locgpu_knn with LOCGPU_SEARCH_GRID_EXACT). out_idx[i*k] = -2 marks a query the caller
// must answer with the tree kernel.
template <int K>
__global__ __launch_bounds__(kBlock) void knn_grid_query_kernel(GridDev g, const uint2* __restrict__ tree, const float* __restrict__ queries, size_t nq,
                                                                int32_t* __restrict__ out_idx, unsigned int* __restrict__ n_flagged) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nq) return;
    SortedSet<K> set;
    if (grid_knn<K>(g, queries[3 * i], queries[3 * i + 1], queries[3 * i + 2], set, g.max_ring2)) {
        out_idx[i * K] = -2;
        atomicAdd(n_flagged, 1u);
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) out_idx[i * K + j] = (int32_t)(tree[set.id[j]].y & 0x3FFFFFFFu);
    }
}

static GridDev to_dev(const GridView& v) {
    static const int max_ring = [] { const char* e = getenv("LOCGPU_GRID_RINGS"); const int r = e ? atoi(e) : 1; return r < 1 ? 1 : (r > 8 ? 8 : r); }();
    static const int max_ring2 = [] { const char* e = getenv("LOCGPU_GRID_RINGS2"); const int r = e ? atoi(e) : 8; return r < 1 ? 1 : (r > 32 ? 32 : r); }();
    return GridDev{v.cell_start, v.pts, v.dims[0], v.dims[1], v.dims[2], v.origin[0], v.origin[1], v.origin[2], v.cell, v.inv_cell, v.slack, max_ring,
                   max_ring2};
}

bool launch_icp_search_grid(const GridView& grid, const SearchArgs& a, hipStream_t s) {
    const GridDev g = to_dev(grid);
    dim3 blocks((a.max_n + kBlock - 1) / kBlock, a.n_scans);
    // pass 1 → redo_list2 ; pass 2 (compacted) → redo_list ; exact tree kernel consumes redo_list
    (void)hipMemsetAsync(a.redo_count, 0, sizeof(unsigned int), s);
    (void)hipMemsetAsync(a.redo_count2, 0, sizeof(unsigned int), s);
    if (a.k == 1) {
        hipLaunchKernelGGL((icp_search_grid_kernel<1>), blocks, dim3(kBlock), 0, s, g, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.skip_nonfinite,
                           a.redo_list2, a.redo_count2, a.search_stats);
        hipLaunchKernelGGL((icp_search_grid_pass2_kernel<1>), dim3(1024), dim3(kBlock), 0, s, g, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.redo_list2,
                           a.redo_count2, a.redo_list, a.redo_count);
    } else if (a.k == 5) {
        hipLaunchKernelGGL((icp_search_grid_kernel<5>), blocks, dim3(kBlock), 0, s, g, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.skip_nonfinite,
                           a.redo_list2, a.redo_count2, a.search_stats);
        hipLaunchKernelGGL((icp_search_grid_pass2_kernel<5>), dim3(1024), dim3(kBlock), 0, s, g, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.redo_list2,
                           a.redo_count2, a.redo_list, a.redo_count);
    } else {
        return false;
    }
    return launch_icp_search_redo(a, s);  // exact tree traversal (alpha_eff = 1) for what is still open
}

bool launch_knn_grid_query(const GridView& grid, const uint2* tree, const float* q, size_t nq, int k, int32_t* out, unsigned int* n_flagged,
                           hipStream_t s) {
    const GridDev g = to_dev(grid);
    dim3 blocks((unsigned)((nq + kBlock - 1) / kBlock));
    if (k == 1) hipLaunchKernelGGL((knn_grid_query_kernel<1>), blocks, dim3(kBlock), 0, s, g, tree, q, nq, out, n_flagged);
    else if (k == 5) hipLaunchKernelGGL((knn_grid_query_kernel<5>), blocks, dim3(kBlock), 0, s, g, tree, q, nq, out, n_flagged);
    else return false;
    return true;
}

}  // namespace locgpu
