// loc_lib_amd/csrc/grid_kernels.hip — exact k-NN over the tile/cell grid (LOCGPU_SEARCH_GRID_EXACT), gfx950.
//
// Equals KdTree::GetClosestPoint with approximate_ = false (kdtree.cpp:147-167 with the exact NeedExpand branch :227-235): the
// k leaves with the smallest float32 dist² (same Eigen reduction order, no FMA), ascending; a candidate replaces the current
// k-th only when strictly smaller (kdtree.cpp:207). Where float32 distances tie exactly, WHICH of the tied leaves is kept (or
// their order) follows the tree's visit order and std::priority_queue's layout: such queries are detected and handed to the
// exact tree kernel, as are queries whose neighbours lie many cells away.
//
// One Gauss–Newton iteration's search stage:
//   1. bin     every query (transformed source point) → the occupied tile (4×4×4 cells) it falls in: a counting sort whose sizes
//              live on the device (no host round trip; scans that have converged drop out). Each 256-thread block first merges its
//              queries' tiles in an LDS hash, so the global counters see one atomic per distinct tile per block — hot tiles (thousands
//              of queries) would otherwise serialise on same-address atomics.
//   2. tiles   one-wave workgroups walk the tile-ordered query list in ranges of 256. For every run of queries of one tile the wave
//              stages the tile's candidate block — its 8×8×8 cells (the tile and two rings): 27 tile look-ups (L2-resident hash),
//              512 cell extents from the tile records, a wave prefix sum, then the cells' leaves streamed from the (tile, cell)-sorted
//              array into LDS — once, and every query of the run takes its 3×3×3 cells from LDS (9 runs: consecutive cells along
//              x are adjacent in the staged block) and, when the k-th distance reaches beyond that block's nearest open face, the
//              5×5×5 shell around it.
//   3. tree    what is still open after two rings (first-iteration queries far from every surface), outside every occupied tile,
//              in a block larger than the LDS stage, and every tie: the fast tree traversal with alpha = 1 over the compacted list
//              (icp_search_fast_list_kernel), then icp_search_redo_kernel for the few it hands on.
// locgpu_knn's grid mode (plain queries, no poses) walks rings through the tile records, one thread per query.
#include "grid_kernels.hpp"
#include "icp_kernels.hpp"

#include "device_prims.hpp"

#include <cstdlib>

namespace locgpu {

// Result set of the grid search: ascending sorted array with insertion; a tie raises the flag that sends the query to the exact
// tree traversal (whose libstdc++ heap decides the order among equal distances).
template <int K>
struct SortedSet {  // ascending: d[0] ≤ … ≤ d[K-1]; empty slots hold +inf / kInvalidSlot
    float d[K];
    uint32_t id[K];
    int n;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = __builtin_inff(); id[j] = kInvalidSlot; }
        n = 0;
    }
    __device__ __forceinline__ float top() const { return d[K - 1]; }
    // returns true when the insertion tied with a resident distance (⇒ heap layout would matter)
    __device__ __forceinline__ bool insert(float x, uint32_t w) {
        bool tie = false;
#pragma unroll
        for (int j = 0; j < K - 1; ++j) tie |= (x == d[j]);
        d[K - 1] = x;
        id[K - 1] = w;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool sw = d[j] < d[j - 1];
            const float lo = sw ? d[j] : d[j - 1], hi = sw ? d[j - 1] : d[j];
            const uint32_t ilo = sw ? id[j] : id[j - 1], ihi = sw ? id[j - 1] : id[j];
            d[j - 1] = lo; d[j] = hi; id[j - 1] = ilo; id[j] = ihi;
        }
        n = n < K ? n + 1 : K;
        return tie;
    }
};

struct GridDev {
    const uint2* tile_hash;
    uint32_t tile_mask;
    const TileRec* tiles;
    const float4* pts;
    int nx, ny, nz, ntx, nty, ntz;
    float ox, oy, oz, cell, inv_cell, slack;
    int max_ring2;  // rings examined by the walk kernel
};

__device__ __forceinline__ uint32_t tile_hash_fn(uint32_t k) {
    k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
    return k;
}
// the float32 expression grid_build.hip's cell_key_kernel evaluates
__device__ __forceinline__ int cell_coord(float v, float o, float inv) { return (int)floorf((v - o) * inv); }

// record index of tile (tx,ty,tz), or -1 (outside the grid, or no leaf in the tile)
__device__ __forceinline__ int tile_lookup(const GridDev& g, int tx, int ty, int tz) {
    if ((unsigned)tx >= (unsigned)g.ntx || (unsigned)ty >= (unsigned)g.nty || (unsigned)tz >= (unsigned)g.ntz) return -1;
    const uint32_t key = (uint32_t)((tz * g.nty + ty) * g.ntx + tx);
    uint32_t h = tile_hash_fn(key) & g.tile_mask;
    for (;;) {
        const uint2 e = g.tile_hash[h];
        if (e.x == key) return (int)e.y;
        if (e.x == kEmptyCell) return -1;
        h = (h + 1) & g.tile_mask;
    }
}

// {first leaf, count} of cell (ix,iy,iz) of tile record t
__device__ __forceinline__ uint2 cell_extent(const GridDev& g, int t, int ix, int iy, int iz) {
    const TileRec* r = g.tiles + t;
    const int c = (iz * kGridTile + iy) * kGridTile + ix;
    const uint32_t a = r->cstart[c], b = r->cstart[c + 1];
    return make_uint2(r->pt_start + a, b - a);
}

// {first leaf, count} of a cell given by global cell coordinates; {0, 0} when no leaf lies in it
__device__ __forceinline__ uint2 cell_lookup(const GridDev& g, int cx, int cy, int cz) {
    if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz) return make_uint2(0u, 0u);
    const int t = tile_lookup(g, cx / kGridTile, cy / kGridTile, cz / kGridTile);
    if (t < 0) return make_uint2(0u, 0u);
    return cell_extent(g, t, cx % kGridTile, cy % kGridTile, cz % kGridTile);
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

// The tile kernel's result set: ascending distances in registers, updated by a branch-free insertion network — every lane of
// a wave is at a different candidate of a different query, so a conditional insert would execute for all of them anyway and
// pay the branch on top. Two cheap tests catch every case in which the reference's answer depends on its visit order:
// a candidate equal to the current k-th distance, and an eviction while the two largest distances are equal; equal
// distances that survive to the end are found by finish().
template <int K>
struct TopK {
    float d[K];
    uint32_t id[K];
    bool tie;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = __builtin_inff(); id[j] = kInvalidSlot; }
        tie = false;
    }
    __device__ __forceinline__ float top() const { return d[K - 1]; }
    __device__ __forceinline__ void offer(float x, uint32_t w) {
        const float t = d[K - 1];
        tie |= (x == t) | ((x < t) & (K > 1 ? d[K > 1 ? K - 2 : 0] == t : false) & (t < __builtin_inff()));
        float c = x;
        uint32_t cw = w;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool sm = c < d[j];
            const float nd = sm ? c : d[j], nc = sm ? d[j] : c;
            const uint32_t nw = sm ? cw : id[j], ncw = sm ? id[j] : cw;
            d[j] = nd; id[j] = nw; c = nc; cw = ncw;
        }
    }
    __device__ __forceinline__ bool full() const { return d[K - 1] < __builtin_inff(); }
    __device__ __forceinline__ bool finish() {  // true when the tree must answer
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) tie |= (d[j] == d[j + 1]) & (d[j] < __builtin_inff());
        return tie;
    }
};

template <int K>
__device__ __forceinline__ void offer_point(TopK<K>& set, float qx, float qy, float qz, const float4 p) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    set.offer(dx * dx + (dy * dy + dz * dz), __float_as_uint(p.w));  // Eigen squaredNorm order, no FMA (-ffp-contract=off)
}

// settle() for a TopK set: see below
template <int K>
__device__ __forceinline__ int settle_top(const struct GridDev& g, const TopK<K>& set, float qx, float qy, float qz, int cx, int cy, int cz, int R);

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

template <int K>
__device__ __forceinline__ int settle_top(const GridDev& g, const TopK<K>& set, float qx, float qy, float qz, int cx, int cy, int cz, int R) {
    float dmin = __builtin_inff();
    bool open_face = false;
    if (cx - R > 0) { dmin = fminf(dmin, qx - (g.ox + (float)(cx - R) * g.cell)); open_face = true; }
    if (cx + R < g.nx - 1) { dmin = fminf(dmin, (g.ox + (float)(cx + R + 1) * g.cell) - qx); open_face = true; }
    if (cy - R > 0) { dmin = fminf(dmin, qy - (g.oy + (float)(cy - R) * g.cell)); open_face = true; }
    if (cy + R < g.ny - 1) { dmin = fminf(dmin, (g.oy + (float)(cy + R + 1) * g.cell) - qy); open_face = true; }
    if (cz - R > 0) { dmin = fminf(dmin, qz - (g.oz + (float)(cz - R) * g.cell)); open_face = true; }
    if (cz + R < g.nz - 1) { dmin = fminf(dmin, (g.oz + (float)(cz + R + 1) * g.cell) - qz); open_face = true; }
    if (!open_face) return set.full() ? 0 : 2;
    const float safe = dmin - g.slack;
    return (set.full() && safe > 0.f && set.top() <= safe * safe) ? 0 : 1;
}

// Ring walk through the tile records, one thread per query. Returns true when the tree kernel must answer (tie, NaN, too far, < K leaves).
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
constexpr int kBinSlots = 512;  // LDS hash slots per 256-thread block (≥ 2 × the block's queries)

// Block-local merge of the block's tile keys. Every thread with `has` gets the slot of its key and its rank among the block's
// queries of that key. After the call (it ends with a barrier) s_key/s_cnt hold the distinct keys and their counts.
__device__ __forceinline__ void block_merge_keys(bool has, uint32_t key, uint32_t* s_key, uint32_t* s_cnt, int& slot, uint32_t& rank) {
    for (int i = threadIdx.x; i < kBinSlots; i += kBlock) { s_key[i] = kEmptyCell; s_cnt[i] = 0u; }
    __syncthreads();
    slot = -1;
    rank = 0;
    if (has) {
        uint32_t h = tile_hash_fn(key) & (kBinSlots - 1);
        for (;;) {
            const uint32_t prev = atomicCAS(&s_key[h], kEmptyCell, key);
            if (prev == kEmptyCell || prev == key) break;
            h = (h + 1) & (kBinSlots - 1);
        }
        slot = (int)h;
        rank = atomicAdd(&s_cnt[h], 1u);
    }
    __syncthreads();
}

template <int K>
__global__ __launch_bounds__(kBlock) void grid_bin_count_kernel(GridDev g, const float4* __restrict__ src, const int* __restrict__ counts,
                                                                const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                int skip_nonfinite, uint32_t* __restrict__ qkey, uint32_t* __restrict__ tile_count,
                                                                uint32_t* __restrict__ walk_list, unsigned int* __restrict__ walk_count,
                                                                unsigned long long* __restrict__ search_stats) {
    __shared__ uint32_t s_key[kBinSlots], s_cnt[kBinSlots];
    const int scan = blockIdx.y;
    if (st[scan].done) return;  // uniform per block
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const size_t gi = (size_t)scan * max_n + i;
    bool has = false, to_tree = false, counted = false;
    uint32_t key = kEmptyCell;
    if (i < counts[scan]) {
        const float4 p = src[gi];
        if (skip_nonfinite && !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {  // pcl::isFinite, icp cpp:64 (P2P only)
#pragma unroll
            for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = kInvalidSlot;
        } else {
            counted = true;
            const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
            const float qx = (float)qs.x, qy = (float)qs.y, qz = (float)qs.z;
            const int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
            const bool inside = qx == qx && qy == qy && qz == qz && (unsigned)cx < (unsigned)g.nx && (unsigned)cy < (unsigned)g.ny && (unsigned)cz < (unsigned)g.nz;
            const int t = inside ? tile_lookup(g, cx / kGridTile, cy / kGridTile, cz / kGridTile) : -1;
            if (t >= 0) { has = true; key = (uint32_t)t; }
            else to_tree = true;  // outside every occupied tile (or NaN): the tree answers
        }
        qkey[gi] = key;
    }
    wave_append(walk_list, walk_count, to_tree, (uint32_t)gi);
    if (search_stats) {  // one add per wave, only when stats were requested
        const unsigned long long m = __ballot(counted);
        if (m && (int)__lane_id() == __ffsll((long long)m) - 1) atomicAdd(&search_stats[0], (unsigned long long)__popcll(m));
    }
    int slot;
    uint32_t rank;
    block_merge_keys(has, key, s_key, s_cnt, slot, rank);
    for (int s = threadIdx.x; s < kBinSlots; s += kBlock)
        if (s_key[s] != kEmptyCell) atomicAdd(&tile_count[s_key[s]], s_cnt[s]);
}

__global__ __launch_bounds__(kBlock) void grid_bin_scatter_kernel(const int* __restrict__ counts, const PoseState* __restrict__ st, int max_n,
                                                                  const uint32_t* __restrict__ qkey, uint32_t* __restrict__ tile_offset,
                                                                  uint2* __restrict__ sorted) {
    __shared__ uint32_t s_key[kBinSlots], s_cnt[kBinSlots], s_base[kBinSlots];
    const int scan = blockIdx.y;
    if (st[scan].done) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const size_t gi = (size_t)scan * max_n + i;
    const uint32_t key = i < counts[scan] ? qkey[gi] : kEmptyCell;
    const bool has = key != kEmptyCell;
    int slot;
    uint32_t rank;
    block_merge_keys(has, key, s_key, s_cnt, slot, rank);
    for (int s = threadIdx.x; s < kBinSlots; s += kBlock)
        if (s_key[s] != kEmptyCell) s_base[s] = atomicAdd(&tile_offset[s_key[s]], s_cnt[s]);
    __syncthreads();
    if (has) sorted[s_base[slot] + rank] = make_uint2((uint32_t)gi, key);
}

// ------------------------------------------------------------------------------------------------ 2. tiles
constexpr int kStageRing = 2;                                          // rings of cells staged around the tile
constexpr int kStageEdge = kGridTile + 2 * kStageRing;                 // 8 cells
constexpr int kStageCells = kStageEdge * kStageEdge * kStageEdge;      // 512
constexpr int kStageCap = 1024;                                        // leaves a staged block may hold (16 KB of LDS)
constexpr int kRangeQ = 256;                                           // sorted queries per work range
static_assert(kStageCells % 64 == 0 && kStageCap < 65536, "staging layout");

template <int K>
__global__ __launch_bounds__(64) void grid_tile_search_kernel(GridDev g, const uint2* __restrict__ sorted, const uint32_t* __restrict__ n_binned_ptr,
                                                              const float4* __restrict__ src, const PoseState* __restrict__ st,
                                                              uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                              uint32_t* __restrict__ tree_list, unsigned int* __restrict__ tree_count, int exp_flags) {
    __shared__ float4 s_pts[kStageCap];
    __shared__ uint16_t s_lstart[kStageCells + 8];
    __shared__ uint32_t s_keys[kRangeQ];
    __shared__ int s_nt[27];
    constexpr int kPerLane = kStageCells / 64;  // 8 consecutive cells (one x-row of the block) per lane
    const int lane = threadIdx.x;
    const uint32_t n_binned = *n_binned_ptr;
    const uint32_t n_ranges = (n_binned + kRangeQ - 1) / kRangeQ;
    for (uint32_t c = blockIdx.x; c < n_ranges; c += gridDim.x) {
        const uint32_t base = c * kRangeQ;
        const int n_in = (int)min((uint32_t)kRangeQ, n_binned - base);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kRangeQ / 64; ++k) {
            const int idx = k * 64 + lane;
            s_keys[idx] = idx < n_in ? sorted[base + idx].y : kEmptyCell;
        }
        __syncthreads();
        int pos = 0;
        while (pos < n_in) {
            const uint32_t cur = s_keys[pos];
            int end = pos;
            for (;;) {  // end of the run of `cur`
                const int idx = end + lane;
                const unsigned long long m = __ballot(idx < n_in && s_keys[idx] == cur);
                const int nz = (~m) ? __ffsll((long long)~m) - 1 : 64;
                end += nz;
                if (nz < 64) break;
            }
            // ---- stage the candidate block of tile `cur`: the tile's 4×4×4 cells and two rings around them
            const uint32_t lin = g.tiles[cur].tile_lin;
            const int tz = (int)(lin / (uint32_t)(g.ntx * g.nty));
            const int rem = (int)(lin - (uint32_t)tz * (uint32_t)(g.ntx * g.nty));
            const int ty = rem / g.ntx, tx = rem - ty * g.ntx;
            if (lane < 27) {
                const int ox = lane % 3 - 1, oy = (lane / 3) % 3 - 1, oz = lane / 9 - 1;
                s_nt[lane] = lane == 13 ? (int)cur : tile_lookup(g, tx + ox, ty + oy, tz + oz);
            }
            __syncthreads();
            uint32_t cnt[kPerLane], gs[kPerLane], sum = 0;
            {
                // lane = one x-row of the block: block cells lx = 0..7 ↔ cells −2..5 of the tile ↔ neighbour tile (lx + 2) / 4, cell (lx + 2) % 4
                const int ly = lane % kStageEdge, lz = lane / kStageEdge;
                const int ny_ = (ly + 2) / 4, nz_ = (lz + 2) / 4, iy = (ly + 2) % 4, iz = (lz + 2) % 4;
#pragma unroll
                for (int j = 0; j < kPerLane; ++j) {
                    const int t = s_nt[(nz_ * 3 + ny_) * 3 + (j + 2) / 4];
                    uint2 r = make_uint2(0u, 0u);
                    if (t >= 0) r = cell_extent(g, t, (j + 2) % 4, iy, iz);
                    gs[j] = r.x; cnt[j] = r.y; sum += r.y;
                }
            }
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= off) incl += v;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
            const bool fits = total <= (uint32_t)kStageCap;
            if (fits) {
                uint32_t run = incl - sum;
#pragma unroll
                for (int j = 0; j < kPerLane; ++j) {  // every lane copies the leaves of its own eight cells (independent 16-byte loads)
                    s_lstart[lane * kPerLane + j] = (uint16_t)run;
                    if (!(exp_flags & 2)) for (uint32_t k = 0; k < cnt[j]; ++k) s_pts[run + k] = g.pts[gs[j] + k];
                    run += cnt[j];
                }
                if (lane == 63) s_lstart[kStageCells] = (uint16_t)total;
            }
            __syncthreads();
            // ---- the run's queries, 64 at a time
            for (int j0 = pos; j0 < end; j0 += 64) {
                const int j = j0 + lane;
                bool to_tree = false;
                uint32_t gi = 0;
                if (j < end) {
                    gi = sorted[base + j].x;
                    to_tree = !fits;
                }
                if (j < end && fits && !(exp_flags & 1)) {
                    const int scan = (int)(gi / (uint32_t)max_n);
                    const float4 p = src[gi];
                    const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
                    const float qx = (float)qs.x, qy = (float)qs.y, qz = (float)qs.z;
                    const int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
                    const int lx = cx - tx * kGridTile + kStageRing, ly = cy - ty * kGridTile + kStageRing, lz = cz - tz * kGridTile + kStageRing;  // 2..5
                    TopK<K> set;
                    set.init();
                    {   // ring 1: nine rows of three consecutive cells, walked as ONE loop (a lane switches rows when its run ends)
                        int r = 0;
                        const int id00 = ((lz - 1) * kStageEdge + (ly - 1)) * kStageEdge + (lx - 1);
                        uint32_t pi = s_lstart[id00], en = s_lstart[id00 + 3];
                        for (;;) {
                            while (pi == en && r < 8) {
                                ++r;
                                const int id0 = id00 + (r / 3) * (kStageEdge * kStageEdge) + (r % 3) * kStageEdge;
                                pi = s_lstart[id0]; en = s_lstart[id0 + 3];
                            }
                            if (pi == en) break;
                            offer_point<K>(set, qx, qy, qz, s_pts[pi]);
                            ++pi;
                        }
                    }
                    int outcome = settle_top<K>(g, set, qx, qy, qz, cx, cy, cz, 1);
                    if (outcome == 1) {  // ring 2: the shell of the 5×5×5 block — whole rows where |dy| or |dz| is 2, the two end cells elsewhere
#pragma unroll 1
                        for (int r = 0; r < 25; ++r) {
                            const int dy = r % 5 - 2, dz = r / 5 - 2;
                            const int row = ((lz + dz) * kStageEdge + (ly + dy)) * kStageEdge + lx;
                            if (max(abs(dy), abs(dz)) == 2) {
                                const uint32_t b = s_lstart[row - 2], en = s_lstart[row + 3];
                                for (uint32_t pi = b; pi < en; ++pi) offer_point<K>(set, qx, qy, qz, s_pts[pi]);
                            } else {
                                uint32_t b = s_lstart[row - 2], en = s_lstart[row - 1];
                                for (uint32_t pi = b; pi < en; ++pi) offer_point<K>(set, qx, qy, qz, s_pts[pi]);
                                b = s_lstart[row + 2]; en = s_lstart[row + 3];
                                for (uint32_t pi = b; pi < en; ++pi) offer_point<K>(set, qx, qy, qz, s_pts[pi]);
                            }
                        }
                        outcome = settle_top<K>(g, set, qx, qy, qz, cx, cy, cz, 2);
                    }
                    const bool tie = set.finish();
                    if (outcome == 0 && !tie) {
#pragma unroll
                        for (int jj = 0; jj < K; ++jj) nn[(size_t)jj * nn_pitch + gi] = set.id[jj];
                    } else {
                        to_tree = true;
                    }
                }
                wave_append(tree_list, tree_count, to_tree, gi);
            }
            __syncthreads();  // the next run's staging overwrites the block
            pos = end;
        }
    }
}

// ------------------------------------------------------------------------------------------------ ring walk (locgpu_knn only)
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
    return GridDev{v.tile_hash, v.tile_mask, v.tiles, v.pts, v.dims[0], v.dims[1], v.dims[2], v.tdims[0], v.tdims[1], v.tdims[2],
                   v.origin[0], v.origin[1], v.origin[2], v.cell, v.inv_cell, v.slack, max_ring2};
}

template <int K>
static bool search_grid_k(const GridView& grid, const GridDev& g, const SearchArgs& a, const GridSearchScratch& sc, hipStream_t s) {
    const dim3 blocks((a.max_n + kBlock - 1) / kBlock, a.n_scans);
    // work lists: redo_list2 = queries for the fast tree traversal, redo_list = what that hands to the exact redo kernel
    (void)hipMemsetAsync(a.redo_count, 0, sizeof(unsigned int), s);
    (void)hipMemsetAsync(a.redo_count2, 0, sizeof(unsigned int), s);
    (void)hipMemsetAsync(sc.tile_count, 0, ((size_t)grid.n_tocc + 1) * sizeof(uint32_t), s);
    hipLaunchKernelGGL((grid_bin_count_kernel<K>), blocks, dim3(kBlock), 0, s, g, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.skip_nonfinite, sc.qkey,
                       sc.tile_count, a.redo_list2, a.redo_count2, a.search_stats);
    size_t tb = grid.scan_temp_bytes;
    if (prim::exclusive_sum(sc.scan_temp, tb, sc.tile_count, sc.tile_count, (int)(grid.n_tocc + 1), s) != hipSuccess) return false;
    hipLaunchKernelGGL(grid_bin_scatter_kernel, blocks, dim3(kBlock), 0, s, a.counts, a.st, a.max_n, sc.qkey, sc.tile_count, sc.sorted);
    // after the scatter tile_count[t] = end of tile t's queries; the last entry (never incremented) still holds the total
    const size_t total_q = (size_t)a.max_n * a.n_scans;
    const unsigned waves = (unsigned)std::min<size_t>((total_q + kRangeQ - 1) / kRangeQ, 256u * 8u);
    static const int exp_flags = [] { const char* e = getenv("LOCGPU_GRID_EXP"); return e ? atoi(e) : 0; }();  // timing experiments only (results wrong)
    hipLaunchKernelGGL((grid_tile_search_kernel<K>), dim3(waves), dim3(64), 0, s, g, sc.sorted, sc.tile_count + grid.n_tocc, a.src, a.st, a.nn, a.nn_pitch,
                       a.max_n, a.redo_list2, a.redo_count2, exp_flags);
    return launch_icp_search_list(a, a.redo_list2, a.redo_count2, s);  // a.alpha_eff = 1: exact pruning
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
