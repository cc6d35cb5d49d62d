// loc_lib_amd/csrc/grid_kernels.hip — exact k-NN over the cell grid (LOCGPU_SEARCH_GRID_EXACT), gfx950.
//
// Equals KdTree::GetClosestPoint with approximate_ = false (kdtree.cpp:147-167 with the exact NeedExpand branch :227-235): the
// k leaves with the smallest float32 dist² (same Eigen reduction order, no FMA), ascending; a candidate replaces the current
// k-th only when strictly smaller (kdtree.cpp:207). Where float32 distances tie exactly, WHICH of the tied leaves is kept (or
// their order) follows the tree's visit order and std::priority_queue's layout: such queries are detected and handed to the
// exact tree kernel, as are queries whose neighbours lie many cells away.
//
// One Gauss–Newton iteration's search stage:
//   1. bin     every query (transformed source point) → its tile (4×4×4 cells): per-tile counters, exclusive scan, scatter —
//              a counting sort whose sizes live on the device (no host round trip; scans that have converged drop out).
//   2. tiles   one-wave workgroups walk the tile-ordered query list in chunks of 64. For each distinct tile of a chunk the wave
//              stages the tile's candidate block — its 6×6×6 cells (the tile and one ring): 216 hash look-ups, a wave prefix sum,
//              then the cells' leaves streamed from the cell-sorted array (16-byte loads of consecutive addresses) into LDS —
//              once, and every query of the tile takes its 3×3×3 cells from LDS (9 runs: consecutive cells along x are adjacent).
//              Dense tiles fill whole chunks, so a staged block serves up to 64 queries.
//   3. walk    queries the 3×3×3 block did not settle (k-th distance beyond the block's nearest open face), outside the grid, or in
//              a block larger than the LDS stage: ring by ring through the hash, one thread per query, on the compacted list.
//   4. tree    what is still open after `max_ring2` rings, and every tie: icp_search_redo_kernel with alpha = 1.
#include "grid_kernels.hpp"
#include "icp_kernels.hpp"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

namespace locgpu {

struct GridDev {
    const uint4* cells;
    uint32_t mask;
    const float4* pts;
    int nx, ny, nz, ntx, nty, ntz;
    float ox, oy, oz, cell, inv_cell, slack;
    int max_ring2;  // rings examined by the walk kernel
};

__device__ __forceinline__ uint32_t cell_hash(uint32_t k) {
    k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
    return k;
}
// the float32 expression grid_build.hip's cell_key_kernel evaluates
__device__ __forceinline__ int cell_coord(float v, float o, float inv) { return (int)floorf((v - o) * inv); }

// {first point, count} of a cell; {0, 0} outside the grid or when no leaf lies in it
__device__ __forceinline__ uint2 cell_lookup(const GridDev& g, int cx, int cy, int cz) {
    if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz) return make_uint2(0u, 0u);
    const uint32_t key = (uint32_t)(((size_t)cz * g.ny + cy) * g.nx + cx);
    uint32_t h = cell_hash(key) & g.mask;
    for (;;) {
        const uint4 e = g.cells[h];
        if (e.x == key) return make_uint2(e.y, e.z);
        if (e.x == kEmptyCell) return make_uint2(0u, 0u);
        h = (h + 1) & g.mask;
    }
}

// One candidate. Ties that could make the reference's answer depend on its visit order raise `tie`.
template <int K>
__device__ __forceinline__ void consider(SortedSet<K>& set, bool& tie, float qx, float qy, float qz, const float4 p) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float dis2 = dx * dx + (dy * dy + dz * dz);  // Eigen squaredNorm order, no FMA (-ffp-contract=off)
    const float top = set.top();
    if (dis2 < top) tie |= set.insert(dis2, __float_as_uint(p.w));
    else if (dis2 == top && top < __builtin_inff()) tie = true;
}

// After the cells within Chebyshev distance R of (cx,cy,cz) were examined: 0 = the set is final, 1 = more rings needed,
// 2 = every leaf was examined and fewer than K exist (the tree kernel answers like the reference does).
template <int K>
__device__ __forceinline__ int settle(const GridDev& g, const SortedSet<K>& set, float qx, float qy, float qz, int cx, int cy, int cz, int R) {
    float dmin = __builtin_inff();
    bool open_face = false;
    if (cx - R > 0) { dmin = fminf(dmin, qx - (g.ox + (float)(cx - R) * g.cell)); open_face = true; }
    if (cx + R < g.nx - 1) { dmin = fminf(dmin, (g.ox + (float)(cx + R + 1) * g.cell) - qx); open_face = true; }
    if (cy - R > 0) { dmin = fminf(dmin, qy - (g.oy + (float)(cy - R) * g.cell)); open_face = true; }
    if (cy + R < g.ny - 1) { dmin = fminf(dmin, (g.oy + (float)(cy + R + 1) * g.cell) - qy); open_face = true; }
    if (cz - R > 0) { dmin = fminf(dmin, qz - (g.oz + (float)(cz - R) * g.cell)); open_face = true; }
    if (cz + R < g.nz - 1) { dmin = fminf(dmin, (g.oz + (float)(cz + R + 1) * g.cell) - qz); open_face = true; }
    if (!open_face) return set.n < K ? 2 : 0;
    const float safe = dmin - g.slack;
    return (set.n == K && safe > 0.f && set.top() <= safe * safe) ? 0 : 1;
}

// Ring walk through the hash, one thread per query. Returns true when the tree kernel must answer (tie, NaN, too far, < K leaves).
template <int K>
__device__ __forceinline__ bool grid_knn_walk(const GridDev& g, float qx, float qy, float qz, SortedSet<K>& set, int max_ring) {
    set.init();
    if (!(qx == qx && qy == qy && qz == qz)) return true;
    const float fx = floorf((qx - g.ox) * g.inv_cell), fy = floorf((qy - g.oy) * g.inv_cell), fz = floorf((qz - g.oz) * g.inv_cell);
    if (fx < -(float)max_ring - 1.f || fy < -(float)max_ring - 1.f || fz < -(float)max_ring - 1.f || fx > (float)(g.nx + max_ring) ||
        fy > (float)(g.ny + max_ring) || fz > (float)(g.nz + max_ring))
        return true;  // far outside the grid: no ring below could reach a leaf
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    bool tie = false;
    for (int R = 0; R <= max_ring; ++R) {
        for (int dz = -R; dz <= R; ++dz)
            for (int dy = -R; dy <= R; ++dy) {
                const bool shell_row = max(abs(dy), abs(dz)) == R;  // whole row is new; otherwise only its two end cells
                for (int dx = -R; dx <= R; dx += (shell_row || R == 0) ? 1 : 2 * R) {
                    const uint2 c = cell_lookup(g, cx + dx, cy + dy, cz + dz);
                    for (uint32_t pi = c.x; pi < c.x + c.y; ++pi) consider<K>(set, tie, qx, qy, qz, g.pts[pi]);
                }
            }
        if (R == 0) continue;
        const int r = settle<K>(g, set, qx, qy, qz, cx, cy, cz, R);
        if (r == 0) return tie;
        if (r == 2) return true;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ 1. binning
template <int K>
__global__ __launch_bounds__(kBlock) void grid_bin_count_kernel(GridDev g, const float4* __restrict__ src, const int* __restrict__ counts,
                                                                const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                int skip_nonfinite, uint32_t* __restrict__ qkey, uint32_t* __restrict__ tile_count,
                                                                uint32_t* __restrict__ walk_list, unsigned int* __restrict__ walk_count,
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
        qkey[gi] = kEmptyCell;
        return;
    }
    if (search_stats) atomicAdd(&search_stats[0], 1ull);
    const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
    const float qx = (float)qs.x, qy = (float)qs.y, qz = (float)qs.z;
    const int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
    const bool inside = qx == qx && qy == qy && qz == qz && (unsigned)cx < (unsigned)g.nx && (unsigned)cy < (unsigned)g.ny && (unsigned)cz < (unsigned)g.nz;
    if (!inside) {  // outside the leaves' bounding box (or NaN): the walk kernel handles it
        qkey[gi] = kEmptyCell;
        walk_list[atomicAdd(walk_count, 1u)] = (uint32_t)gi;
        return;
    }
    const uint32_t tile = (uint32_t)(((cz / kGridTile) * g.nty + (cy / kGridTile)) * g.ntx + (cx / kGridTile));
    qkey[gi] = tile;
    atomicAdd(&tile_count[tile], 1u);
}

__global__ __launch_bounds__(kBlock) void grid_bin_scatter_kernel(const int* __restrict__ counts, const PoseState* __restrict__ st, int max_n,
                                                                  const uint32_t* __restrict__ qkey, uint32_t* __restrict__ tile_offset,
                                                                  uint2* __restrict__ sorted) {
    const int scan = blockIdx.y;
    if (st[scan].done) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= counts[scan]) return;
    const size_t gi = (size_t)scan * max_n + i;
    const uint32_t key = qkey[gi];
    if (key == kEmptyCell) return;
    sorted[atomicAdd(&tile_offset[key], 1u)] = make_uint2((uint32_t)gi, key);
}

// ------------------------------------------------------------------------------------------------ 2. tiles
constexpr int kStageEdge = kGridTile + 2;                                  // 6 cells: the tile and one ring
constexpr int kStageCells = kStageEdge * kStageEdge * kStageEdge;      // 216
constexpr int kStageCap = 1024;                                        // leaves a staged block may hold (16 KB of LDS)

template <int K>
__global__ __launch_bounds__(64) void grid_tile_search_kernel(GridDev g, const uint2* __restrict__ sorted, const uint32_t* __restrict__ n_binned_ptr,
                                                              const float4* __restrict__ src, const PoseState* __restrict__ st,
                                                              uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                              uint32_t* __restrict__ walk_list, unsigned int* __restrict__ walk_count,
                                                              uint32_t* __restrict__ redo_list, unsigned int* __restrict__ redo_count) {
    __shared__ float4 s_pts[kStageCap];
    __shared__ uint32_t s_gstart[kStageCells + 8];
    __shared__ uint32_t s_lstart[kStageCells + 8];
    const int lane = threadIdx.x;
    const uint32_t n_binned = *n_binned_ptr;
    const uint32_t n_chunks = (n_binned + 63u) / 64u;
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t si = c * 64u + (uint32_t)lane;
        const bool valid = si < n_binned;
        const uint2 e = valid ? sorted[si] : make_uint2(0u, kEmptyCell);
        float qx = 0.f, qy = 0.f, qz = 0.f;
        int cx = 0, cy = 0, cz = 0;
        if (valid) {
            const int scan = (int)(e.x / (uint32_t)max_n);
            const float4 p = src[e.x];
            const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
            qx = (float)qs.x; qy = (float)qs.y; qz = (float)qs.z;
            cx = cell_coord(qx, g.ox, g.inv_cell); cy = cell_coord(qy, g.oy, g.inv_cell); cz = cell_coord(qz, g.oz, g.inv_cell);
        }
        SortedSet<K> set;
        set.init();
        bool tie = false;
        int outcome = 1;  // 0 settled, 1 needs the walk kernel, 2 needs the tree
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t cur = (uint32_t)__shfl((int)e.y, leader, 64);
            const bool mine = valid && e.y == cur;
            const int tz = (int)(cur / (uint32_t)(g.ntx * g.nty));
            const int rem = (int)(cur - (uint32_t)tz * (uint32_t)(g.ntx * g.nty));
            const int ty = rem / g.ntx, tx = rem - ty * g.ntx;
            const int bx = tx * kGridTile - 1, by = ty * kGridTile - 1, bz = tz * kGridTile - 1;
            // ---- stage: look the block's 216 cells up (4 per lane), prefix-sum their sizes across the wave
            uint32_t cnt[4], gs[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = lane * 4 + j;
                uint2 r = make_uint2(0u, 0u);
                if (id < kStageCells) r = cell_lookup(g, bx + id % kStageEdge, by + (id / kStageEdge) % kStageEdge, bz + id / (kStageEdge * kStageEdge));
                gs[j] = r.x; cnt[j] = r.y; sum += r.y;
            }
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= off) incl += t;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
            uint32_t run = incl - sum;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = lane * 4 + j;
                if (id <= kStageCells) { s_lstart[id] = run; s_gstart[id] = gs[j]; }
                run += cnt[j];
            }
            __syncthreads();
            if (total <= (uint32_t)kStageCap) {
                // ---- copy the block's leaves into LDS: position p belongs to the cell `id` with lstart[id] <= p < lstart[id + 1]
                for (uint32_t p = (uint32_t)lane; p < total; p += 64u) {
                    int lo = 0, hi = kStageCells;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int mid = (lo + hi) >> 1;
                        const bool right = s_lstart[mid] <= p;
                        lo = right ? mid : lo;
                        hi = right ? hi : mid;
                    }
                    s_pts[p] = g.pts[s_gstart[lo] + (p - s_lstart[lo])];
                }
                __syncthreads();
                if (mine) {
                    const int lx = cx - bx, ly = cy - by, lz = cz - bz;  // 1..4
#pragma unroll 1
                    for (int r = 0; r < 9; ++r) {
                        const int id0 = (lz + r / 3 - 1) * (kStageEdge * kStageEdge) + (ly + r % 3 - 1) * kStageEdge + (lx - 1);
                        const uint32_t b = s_lstart[id0], en = s_lstart[id0 + 3];  // three consecutive cells along x are one run
                        for (uint32_t pi = b; pi < en; ++pi) consider<K>(set, tie, qx, qy, qz, s_pts[pi]);
                    }
                    outcome = settle<K>(g, set, qx, qy, qz, cx, cy, cz, 1);
                }
            }
            __syncthreads();  // the next tile's staging overwrites the block
            todo &= ~__ballot(mine);
        }
        if (valid) {
            if (outcome == 0 && !tie) {
#pragma unroll
                for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + e.x] = set.id[j];
            } else if (outcome == 1 && !tie) {
                walk_list[atomicAdd(walk_count, 1u)] = e.x;
            } else {
                redo_list[atomicAdd(redo_count, 1u)] = e.x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3. walk
template <int K>
__global__ __launch_bounds__(kBlock) void grid_walk_kernel(GridDev g, const float4* __restrict__ src, const PoseState* __restrict__ st,
                                                           uint32_t* __restrict__ nn, size_t nn_pitch, int max_n, const uint32_t* __restrict__ list_in,
                                                           const unsigned int* __restrict__ n_in, uint32_t* __restrict__ list_out,
                                                           unsigned int* __restrict__ n_out) {
    const unsigned int n = *n_in;
    for (unsigned int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        const size_t gi = list_in[r];
        const int scan = (int)(gi / (size_t)max_n);
        const float4 p = src[gi];
        const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
        SortedSet<K> set;
        if (grid_knn_walk<K>(g, (float)qs.x, (float)qs.y, (float)qs.z, set, g.max_ring2)) {
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
    if (grid_knn_walk<K>(g, queries[3 * i], queries[3 * i + 1], queries[3 * i + 2], set, g.max_ring2)) {
        out_idx[i * K] = -2;
        atomicAdd(n_flagged, 1u);
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) out_idx[i * K + j] = (int32_t)(tree[set.id[j]].y & 0x3FFFFFFFu);
    }
}

static GridDev to_dev(const GridView& v) {
    static const int max_ring2 = [] { const char* e = getenv("LOCGPU_GRID_RINGS2"); const int r = e ? atoi(e) : 6; return r < 1 ? 1 : (r > 32 ? 32 : r); }();
    return GridDev{v.cells, v.cell_mask, v.pts, v.dims[0], v.dims[1], v.dims[2], v.tdims[0], v.tdims[1], v.tdims[2],
                   v.origin[0], v.origin[1], v.origin[2], v.cell, v.inv_cell, v.slack, max_ring2};
}

template <int K>
static bool search_grid_k(const GridView& grid, const GridDev& g, const SearchArgs& a, const GridSearchScratch& sc, hipStream_t s) {
    const dim3 blocks((a.max_n + kBlock - 1) / kBlock, a.n_scans);
    // work lists: redo_list2 = walk list (pass 3), redo_list = tree list (pass 4)
    (void)hipMemsetAsync(a.redo_count, 0, sizeof(unsigned int), s);
    (void)hipMemsetAsync(a.redo_count2, 0, sizeof(unsigned int), s);
    (void)hipMemsetAsync(grid.tile_count, 0, ((size_t)grid.n_tiles + 1) * sizeof(uint32_t), s);
    hipLaunchKernelGGL((grid_bin_count_kernel<K>), blocks, dim3(kBlock), 0, s, g, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.skip_nonfinite, sc.qkey,
                       grid.tile_count, a.redo_list2, a.redo_count2, a.search_stats);
    size_t tb = grid.scan_temp_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(grid.scan_temp, tb, grid.tile_count, grid.tile_count, (int)(grid.n_tiles + 1), s) != hipSuccess) return false;
    hipLaunchKernelGGL(grid_bin_scatter_kernel, blocks, dim3(kBlock), 0, s, a.counts, a.st, a.max_n, sc.qkey, grid.tile_count, sc.sorted);
    // after the scatter tile_count[t] = end of tile t; the last entry (never incremented) still holds the total
    const size_t total_q = (size_t)a.max_n * a.n_scans;
    const unsigned waves = (unsigned)std::min<size_t>((total_q + 63) / 64, 256u * 10u);
    hipLaunchKernelGGL((grid_tile_search_kernel<K>), dim3(waves), dim3(64), 0, s, g, sc.sorted, grid.tile_count + grid.n_tiles, a.src, a.st, a.nn, a.nn_pitch,
                       a.max_n, a.redo_list2, a.redo_count2, a.redo_list, a.redo_count);
    hipLaunchKernelGGL((grid_walk_kernel<K>), dim3(1024), dim3(kBlock), 0, s, g, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.redo_list2, a.redo_count2,
                       a.redo_list, a.redo_count);
    return launch_icp_search_redo(a, s);  // exact tree traversal (alpha_eff = 1) for what is still open or tied
}

bool launch_icp_search_grid(const GridView& grid, const SearchArgs& a, const GridSearchScratch& sc, hipStream_t s) {
    const GridDev g = to_dev(grid);
    if (a.k == 1) return search_grid_k<1>(grid, g, a, sc, s);
    if (a.k == 5) return search_grid_k<5>(grid, g, a, sc, s);
    return false;
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
