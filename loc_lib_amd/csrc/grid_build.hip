// loc_lib_amd/csrc/grid_build.hip — device-side ingest of the exact-search grid (LOCGPU_SEARCH_GRID_EXACT), gfx950.
//
// The grid answers the reference's EXACT k-NN (KdTree with approximate_ = false, reachable through
// KdtreeRegistration::SetEnableANN(false), kdtree.cpp:285-288 / :227-235). To return what that tree would, it indexes exactly
// the points the tree holds — its leaves; duplicate points that the reference's degenerate-split rule drops
// (kdtree.cpp:76-81,118-120) are not in the grid either — and every grid point carries its leaf's slot in the packed tree, so
// the fit/accumulate kernels gather neighbours the same way in both search modes.
//
// Everything runs on the GPU from the tree that is already in HBM (no copy of the map back to the host):
//   gather the leaves → bounding box (block partials) → cell edge: the smallest of a geometric ladder for which an occupied
//   cell holds ≥ the target occupancy on average (64-bit cell keys, radix sort, count of distinct keys) → 32-bit keys
//   (tile · 64 + cell inside the tile) → stable radix sort of (key, leaf) → leaves gathered in that order → run-length encode of
//   the tile part (occupied tiles, leaf counts) → exclusive scan (first leaf of every tile) → one wave per occupied tile writes
//   its record (65 prefix sums of the cells' counts) → open-addressing hash table tile → record.
#include "grid_kernels.hpp"

#include "device_prims.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace locgpu {

namespace {

constexpr int kGB = 256;
constexpr int kBoxBlocks = 1024;

__global__ __launch_bounds__(kGB) void gather_leaves_kernel(const uint2* __restrict__ tree, const uint32_t* __restrict__ leaf_slots, size_t n,
                                                            float4* __restrict__ out, unsigned int* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * kGB + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = leaf_slots[i];
    uint4 w;
    __builtin_memcpy(&w, tree + slot, 16);
    const float x = __uint_as_float(w.x), y = __uint_as_float(w.z), z = __uint_as_float(w.w);
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) atomicOr(bad, 1u);
    out[i] = make_float4(x, y, z, __uint_as_float(slot));
}

__global__ __launch_bounds__(kGB) void bbox_partial_kernel(const float4* __restrict__ pts, size_t n, float* __restrict__ part) {
    __shared__ float s[kGB / 64][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = (size_t)blockIdx.x * kGB + threadIdx.x; i < n; i += (size_t)gridDim.x * kGB) {
        const float4 p = pts[i];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int a = 0; a < 3; ++a) { s[wave][a] = lo[a]; s[wave][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[0][threadIdx.x];
        for (int w = 1; w < kGB / 64; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// ladder probe: a 64-bit key of the cell a point falls in for edge 1/inv
__global__ __launch_bounds__(kGB) void probe_key_kernel(const float4* __restrict__ pts, size_t n, float ox, float oy, float oz, float inv,
                                                        unsigned long long* __restrict__ keys) {
    const size_t i = (size_t)blockIdx.x * kGB + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const unsigned long long ix = (unsigned long long)fmaxf((p.x - ox) * inv, 0.f), iy = (unsigned long long)fmaxf((p.y - oy) * inv, 0.f),
                             iz = (unsigned long long)fmaxf((p.z - oz) * inv, 0.f);
    keys[i] = (ix << 42) | (iy << 21) | iz;
}

__global__ __launch_bounds__(kGB) void count_distinct_kernel(const unsigned long long* __restrict__ keys, size_t n, unsigned int* __restrict__ cnt) {
    const size_t i = (size_t)blockIdx.x * kGB + threadIdx.x;
    const bool head = i < n && (i == 0 || keys[i] != keys[i - 1]);
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(cnt, (unsigned int)__popcll(m));
}

__global__ __launch_bounds__(kGB) void cell_key_kernel(const float4* __restrict__ pts, size_t n, float ox, float oy, float oz, float inv, int nx, int ny,
                                                       int nz, int ntx, int nty, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const size_t i = (size_t)blockIdx.x * kGB + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    // the SAME float32 expression the query kernels evaluate (grid_kernels.hip: cell_coord)
    int cx = (int)floorf((p.x - ox) * inv), cy = (int)floorf((p.y - oy) * inv), cz = (int)floorf((p.z - oz) * inv);
    cx = min(max(cx, 0), nx - 1); cy = min(max(cy, 0), ny - 1); cz = min(max(cz, 0), nz - 1);
    const uint32_t tile = (uint32_t)(((cz / kGridTile) * nty + (cy / kGridTile)) * ntx + (cx / kGridTile));
    const uint32_t in_tile = (uint32_t)(((cz % kGridTile) * kGridTile + (cy % kGridTile)) * kGridTile + (cx % kGridTile));
    keys[i] = tile * (uint32_t)kTileCells + in_tile;
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kGB) void gather_sorted_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ key_s,
                                                            size_t n, float4* __restrict__ out, uint32_t* __restrict__ tile_of) {
    const size_t i = (size_t)blockIdx.x * kGB + threadIdx.x;
    if (i < n) { out[i] = in[idx[i]]; tile_of[i] = key_s[i] / (uint32_t)kTileCells; }
}

__device__ __forceinline__ uint32_t tile_hash_fn(uint32_t k) {
    k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
    return k;
}

// One wave per occupied tile: lane c counts the tile's leaves in cell c, a wave prefix sum gives cstart[].
__global__ __launch_bounds__(64) void tile_record_kernel(const uint32_t* __restrict__ key_s, const uint32_t* __restrict__ tile_lin,
                                                         const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_cnt, uint32_t n_tocc,
                                                         TileRec* __restrict__ tiles, uint2* __restrict__ hash, uint32_t mask, unsigned int* __restrict__ flags) {
    const uint32_t t = blockIdx.x;
    if (t >= n_tocc) return;
    const int lane = threadIdx.x;
    const uint32_t s0 = tile_start[t], cnt = tile_cnt[t];
    uint32_t mine = 0;
    for (uint32_t i = 0; i < cnt; ++i) mine += ((key_s[s0 + i] & (uint32_t)(kTileCells - 1)) == (uint32_t)lane) ? 1u : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += v;
    }
    TileRec& r = tiles[t];
    r.cstart[lane] = (uint16_t)(incl - mine);
    const unsigned long long occ = __ballot(mine != 0);
    if (lane == 63) {
        r.cstart[64] = (uint16_t)incl;
        r.cstart[65] = 0;
        r.pt_start = s0;
        r.tile_lin = tile_lin[t];
        if (cnt > 65535u) atomicOr(&flags[0], 2u);  // a tile must fit 16-bit prefix sums
        atomicAdd(&flags[1], (unsigned int)__popcll(occ));
        uint32_t h = tile_hash_fn(tile_lin[t]) & mask;
        for (;;) {
            const uint32_t prev = atomicCAS(&hash[h].x, kEmptyCell, tile_lin[t]);
            if (prev == kEmptyCell) { hash[h].y = t; break; }
            h = (h + 1) & mask;
        }
    }
}

struct Tmp {  // frees its device buffers on every exit path
    std::vector<void*> ptrs;
    ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t bytes) {
        const hipError_t e = hipMalloc((void**)p, bytes ? bytes : 1);
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

#define GB_TRY(expr)                      \
    do {                                  \
        const hipError_t e__ = (expr);    \
        if (e__ != hipSuccess) return e__; \
    } while (0)

inline unsigned blocks_for(size_t n) { return (unsigned)((n + kGB - 1) / kGB); }

}  // namespace

void grid_free(GridBuffers& b) {
    if (b.tile_hash) (void)hipFree(b.tile_hash);
    if (b.tiles) (void)hipFree(b.tiles);
    if (b.pts) (void)hipFree(b.pts);
    if (b.tile_count) (void)hipFree(b.tile_count);
    if (b.scan_temp) (void)hipFree(b.scan_temp);
    b = GridBuffers();
}

hipError_t grid_build_device(const uint2* d_tree, const uint32_t* d_leaf_slots, size_t n, hipStream_t s, GridBuffers& buf, GridView& view,
                             std::string& msg) {
    grid_free(buf);
    view = GridView();
    if (n == 0) { msg = "empty tree"; return hipErrorInvalidValue; }
    if (n >= 0xFFFFFFF0ull) { msg = "too many leaves for 32-bit point indices"; return hipErrorInvalidValue; }
    static const double target_occ = [] { const char* e = getenv("LOCGPU_GRID_OCC"); const double v = e ? atof(e) : 4.0; return v >= 1.0 ? v : 4.0; }();
    Tmp tmp;
    float4* d_leaves = nullptr;
    unsigned int* d_flag = nullptr;
    float* d_part = nullptr;
    GB_TRY(tmp.alloc(&d_leaves, n * sizeof(float4)));
    GB_TRY(tmp.alloc(&d_flag, 2 * sizeof(unsigned int)));
    GB_TRY(tmp.alloc(&d_part, kBoxBlocks * 6 * sizeof(float)));
    GB_TRY(hipMemsetAsync(d_flag, 0, 2 * sizeof(unsigned int), s));
    hipLaunchKernelGGL(gather_leaves_kernel, dim3(blocks_for(n)), dim3(kGB), 0, s, d_tree, d_leaf_slots, n, d_leaves, d_flag);
    const int nb = (int)std::min<size_t>(blocks_for(n), kBoxBlocks);
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(kGB), 0, s, d_leaves, n, d_part);
    std::vector<float> part((size_t)nb * 6);
    unsigned int flag[2] = {0, 0};
    GB_TRY(hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    GB_TRY(hipMemcpyAsync(flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, s));
    GB_TRY(hipStreamSynchronize(s));
    if (flag[0]) { msg = "grid search needs finite target coordinates"; return hipErrorInvalidValue; }
    float lo[3] = {part[0], part[1], part[2]}, hi[3] = {part[3], part[4], part[5]};
    for (int b = 1; b < nb; ++b)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], part[6 * b + a]); hi[a] = std::max(hi[a], part[6 * b + 3 + a]); }

    // ---- cell edge
    unsigned long long *d_k64 = nullptr, *d_k64s = nullptr;
    GB_TRY(tmp.alloc(&d_k64, n * sizeof(unsigned long long)));
    GB_TRY(tmp.alloc(&d_k64s, n * sizeof(unsigned long long)));
    size_t sort_bytes = 0;
    GB_TRY(prim::sort_keys(nullptr, sort_bytes, d_k64, d_k64s, (int)n, 0, 64, s));
    {
        size_t b2 = 0;
        uint32_t* z = nullptr;
        GB_TRY(prim::sort_pairs(nullptr, b2, z, z, z, z, (int)n, 0, 32, s));
        sort_bytes = std::max(sort_bytes, b2);
        GB_TRY(prim::run_length_encode(nullptr, b2, z, z, z, z, (int)n, s));
        sort_bytes = std::max(sort_bytes, b2);
        GB_TRY(prim::exclusive_sum(nullptr, b2, z, z, (int)n, s));
        sort_bytes = std::max(sort_bytes, b2);
    }
    void* d_sort_tmp = nullptr;
    GB_TRY(tmp.alloc(&d_sort_tmp, sort_bytes));
    const double ext = std::max({(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2], 1e-3});
    double cell = ext;
    for (double c = ext / 8192.0; c < ext; c *= 1.3) {
        hipLaunchKernelGGL(probe_key_kernel, dim3(blocks_for(n)), dim3(kGB), 0, s, d_leaves, n, lo[0], lo[1], lo[2], (float)(1.0 / c), d_k64);
        size_t tb = sort_bytes;
        GB_TRY(prim::sort_keys(d_sort_tmp, tb, d_k64, d_k64s, (int)n, 0, 63, s));
        GB_TRY(hipMemsetAsync(d_flag + 1, 0, sizeof(unsigned int), s));
        hipLaunchKernelGGL(count_distinct_kernel, dim3(blocks_for(n)), dim3(kGB), 0, s, d_k64s, n, d_flag + 1);
        unsigned int occ = 0;
        GB_TRY(hipMemcpyAsync(&occ, d_flag + 1, sizeof(occ), hipMemcpyDeviceToHost, s));
        GB_TRY(hipStreamSynchronize(s));
        cell = c;
        if (occ == 0 || (double)n / (double)occ >= target_occ) break;
    }
    // the 32-bit sort key is tile·64 + cell-in-tile: at most 2^26 tiles
    int dims[3], tdims[3];
    float cellf, inv;
    for (;;) {
        cellf = (float)cell;
        inv = 1.0f / cellf;
        double ncell = 1.0, ntile = 1.0;
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            const double d = std::floor((double)((hi[a] - lo[a]) * inv)) + 1.0;
            if (d >= 2097152.0) ok = false;
            dims[a] = (int)std::min(d, 2097151.0);
            tdims[a] = (dims[a] + kGridTile - 1) / kGridTile;
            ncell *= (double)dims[a];
            ntile *= (double)tdims[a];
        }
        if (ok && ncell < 4.0e9 && ntile <= (double)(1u << 26)) break;
        cell *= 1.26;
    }
    // dims from the float32 expression the kernels use
    for (int a = 0; a < 3; ++a) {
        dims[a] = (int)std::floor((hi[a] - lo[a]) * inv) + 1;
        tdims[a] = (dims[a] + kGridTile - 1) / kGridTile;
    }

    // ---- sort the leaves by (tile, cell inside the tile)
    uint32_t *d_key = nullptr, *d_idx = nullptr, *d_key_s = nullptr, *d_idx_s = nullptr, *d_tile_of = nullptr, *d_utile = nullptr, *d_cnt = nullptr,
             *d_start = nullptr, *d_nocc = nullptr;
    GB_TRY(tmp.alloc(&d_key, n * 4)); GB_TRY(tmp.alloc(&d_idx, n * 4)); GB_TRY(tmp.alloc(&d_key_s, n * 4)); GB_TRY(tmp.alloc(&d_idx_s, n * 4));
    GB_TRY(tmp.alloc(&d_tile_of, n * 4)); GB_TRY(tmp.alloc(&d_utile, n * 4)); GB_TRY(tmp.alloc(&d_cnt, n * 4)); GB_TRY(tmp.alloc(&d_start, n * 4));
    GB_TRY(tmp.alloc(&d_nocc, 4));
    hipLaunchKernelGGL(cell_key_kernel, dim3(blocks_for(n)), dim3(kGB), 0, s, d_leaves, n, lo[0], lo[1], lo[2], inv, dims[0], dims[1], dims[2], tdims[0],
                       tdims[1], d_key, d_idx);
    size_t tb = sort_bytes;
    GB_TRY(prim::sort_pairs(d_sort_tmp, tb, d_key, d_key_s, d_idx, d_idx_s, (int)n, 0, 32, s));  // stable: tree order inside a cell
    GB_TRY(hipMalloc((void**)&buf.pts, n * sizeof(float4)));
    hipLaunchKernelGGL(gather_sorted_kernel, dim3(blocks_for(n)), dim3(kGB), 0, s, d_leaves, d_idx_s, d_key_s, n, buf.pts, d_tile_of);
    tb = sort_bytes;
    GB_TRY(prim::run_length_encode(d_sort_tmp, tb, d_tile_of, d_utile, d_cnt, d_nocc, (int)n, s));
    uint32_t n_tocc = 0;
    GB_TRY(hipMemcpyAsync(&n_tocc, d_nocc, 4, hipMemcpyDeviceToHost, s));
    GB_TRY(hipStreamSynchronize(s));
    tb = sort_bytes;
    GB_TRY(prim::exclusive_sum(d_sort_tmp, tb, d_cnt, d_start, (int)n_tocc, s));
    uint32_t cap = 1024;
    while (cap < 2u * n_tocc) cap <<= 1;
    GB_TRY(hipMalloc((void**)&buf.tile_hash, (size_t)cap * sizeof(uint2)));
    GB_TRY(hipMemsetAsync(buf.tile_hash, 0xFF, (size_t)cap * sizeof(uint2), s));
    GB_TRY(hipMalloc((void**)&buf.tiles, (size_t)n_tocc * sizeof(TileRec)));
    GB_TRY(hipMemsetAsync(d_flag, 0, 2 * sizeof(unsigned int), s));
    hipLaunchKernelGGL(tile_record_kernel, dim3(n_tocc), dim3(64), 0, s, d_key_s, d_utile, d_start, d_cnt, n_tocc, buf.tiles, buf.tile_hash, cap - 1, d_flag);
    GB_TRY(hipMemcpyAsync(flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, s));
    GB_TRY(hipStreamSynchronize(s));
    if (flag[0] & 2u) { msg = "a tile holds more than 65535 leaves (degenerate point distribution)"; return hipErrorInvalidValue; }
    const uint32_t n_occ = flag[1];

    // ---- per-iteration binning scratch: one counter per occupied tile
    GB_TRY(hipMalloc((void**)&buf.tile_count, ((size_t)n_tocc + 2) * sizeof(uint32_t)));
    size_t scan_bytes = 0;
    GB_TRY(prim::exclusive_sum(nullptr, scan_bytes, buf.tile_count, buf.tile_count, (int)(n_tocc + 1), s));
    GB_TRY(hipMalloc(&buf.scan_temp, scan_bytes ? scan_bytes : 1));
    GB_TRY(hipGetLastError());
    GB_TRY(hipStreamSynchronize(s));

    view.tile_hash = buf.tile_hash;
    view.tile_mask = cap - 1;
    view.tiles = buf.tiles;
    view.n_tocc = n_tocc;
    view.pts = buf.pts;
    float max_abs = 0.f;
    for (int a = 0; a < 3; ++a) {
        view.dims[a] = dims[a];
        view.tdims[a] = tdims[a];
        view.origin[a] = lo[a];
        max_abs = std::max(max_abs, std::max(std::fabs(lo[a]), std::fabs(lo[a] + dims[a] * cellf)));
    }
    view.cell = cellf;
    view.inv_cell = inv;
    view.slack = 1e-3f * cellf + 16.f * 1.2e-7f * max_abs;  // float32 rounding of the point→cell assignment and of the face positions
    view.num_points = n;
    view.num_cells = n_occ;
    view.tile_count = buf.tile_count;
    view.scan_temp = buf.scan_temp;
    view.scan_temp_bytes = scan_bytes;
    view.bytes = (size_t)cap * sizeof(uint2) + (size_t)n_tocc * sizeof(TileRec) + n * sizeof(float4) + ((size_t)n_tocc + 2) * sizeof(uint32_t);
    return hipSuccess;
}

}  // namespace locgpu
