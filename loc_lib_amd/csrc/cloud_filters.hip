// loc_lib_amd/csrc/cloud_filters.hip — the cloud filters either side of the matcher, on device-resident clouds.
//
// What the reference does here is glue around PCL 1.8 (a dependency that is not under the reference tree):
//   VoxelFilter::Filter  LocUtils/src/model/cloud_filter/voxel_filter.cpp:19-25  → pcl::VoxelGrid<PointXYZI>::filter
//   BoxFilter::Filter    LocUtils/src/model/cloud_filter/box_filter.cpp:25-32    → pcl::CropBox<PointXYZI>::filter
//   RemoveNanPoint       LocUtils/include/LocUtils/common/point_cloud_utils.h:13-20 → pcl::removeNaNFromPointCloud
//   pcl::transformPointCloud(scan, kf, pose.matrix())                            lio.cpp:244,279
// called on every scan before ScanMatch (loc.cpp:217-218, lio.cpp:236) and on the local map after every keyframe
// (lio.cpp:300, loc.cpp:187-194). The kernels below follow PCL 1.8's published algorithms (voxel_grid.hpp, crop_box.hpp,
// filter.hpp, transforms.hpp) and produce the same clouds: same points, same order — VoxelGrid centroids in ascending voxel index, CropBox /
// removeNaN survivors in input order. The one freedom PCL leaves is the order in which a voxel's points are summed
// (it sorts with an unstable std::sort on the voxel index alone); here the radix sort is stable, so the float32 sums run
// in input order.
//
// All of it is HBM-bound byte shuffling (16 B per point in, a key/value radix sort, 16 B per survivor out):
//   voxel filter   algorithmic bytes per input point: 16 (bounding box) + 16 (keys) + 8·2·passes (sort) + 8 + 16 (gather)
//   crop / NaN     16 in + 1 flag + 16·kept out
#include "cloud_filters.hpp"

#include "device_prims.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>

#include "context.hpp"
#include "launch.hpp"

namespace locgpu {

namespace {

constexpr int kFB = 256;

__device__ __forceinline__ bool finite3(const float4& p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// getMinMax3D: bounding box of the (finite, unless the cloud is flagged dense) points. Every block writes one partial box;
// the set-up kernel folds them (same-address atomics from thousands of waves serialise at ~10 ns each).
constexpr int kMinMaxBlocks = 1024;

__global__ __launch_bounds__(kFB) void minmax_kernel(const float4* __restrict__ pts, size_t n, int dense, float* __restrict__ partial) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const size_t stride = (size_t)gridDim.x * kFB;
    size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    auto take = [&](const float4& p) {
        if (!dense && !finite3(p)) return;
        const float c[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = c[a] < mn[a] ? c[a] : mn[a];
            mx[a] = c[a] > mx[a] ? c[a] : mx[a];
        }
    };
    for (; i + 3 * stride < n; i += 4 * stride) {  // four independent loads in flight per thread
        const float4 p0 = pts[i], p1 = pts[i + stride], p2 = pts[i + 2 * stride], p3 = pts[i + 3 * stride];
        take(p0); take(p1); take(p2); take(p3);
    }
    for (; i < n; i += stride) take(pts[i]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            const float o1 = __shfl_xor(mn[a], off, 64), o2 = __shfl_xor(mx[a], off, 64);
            mn[a] = o1 < mn[a] ? o1 : mn[a];
            mx[a] = o2 > mx[a] ? o2 : mx[a];
        }
    }
    __shared__ float s_box[kFB / 64][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_box[wave][a] = mn[a]; s_box[wave][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_box[0][threadIdx.x];
        for (int w = 1; w < kFB / 64; ++w) {
            const float o = s_box[w][threadIdx.x];
            v = threadIdx.x < 3 ? (o < v ? o : v) : (o > v ? o : v);
        }
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// VoxelGrid::applyFilter's set-up: overflow test, min_b_, div_b_, divb_mul_ (voxel_grid.hpp), all in its float32/int arithmetic.
// One 64-thread block: lanes fold the per-block boxes, lane 0 does the scalar part.
__global__ __launch_bounds__(64) void voxel_setup_kernel(VoxelParams* P, const float* __restrict__ partial, int n_partial, float inv_leaf) {
    float box[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int b = threadIdx.x; b < n_partial; b += 64) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = partial[b * 6 + a], hi = partial[b * 6 + 3 + a];
            box[a] = lo < box[a] ? lo : box[a];
            box[3 + a] = hi > box[3 + a] ? hi : box[3 + a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            const float o1 = __shfl_xor(box[a], off, 64), o2 = __shfl_xor(box[3 + a], off, 64);
            box[a] = o1 < box[a] ? o1 : box[a];
            box[3 + a] = o2 > box[3 + a] ? o2 : box[3 + a];
        }
    }
    if (threadIdx.x != 0) return;
    const float inv = inv_leaf;
    const float* mn = box;
    const float* mx = box + 3;
    P->inv_leaf = inv;
    P->n_out = 0;
    P->status = 0;
    if (mn[0] > mx[0]) { P->status = 2; return; }  // no finite point at all
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    if ((double)dx * (double)dy * (double)dz > 2147483647.0) { P->status = 1; return; }
    for (int a = 0; a < 3; ++a) {
        P->min_b[a] = (int)floorf(mn[a] * inv);
        P->div_b[a] = (int)floorf(mx[a] * inv) - P->min_b[a] + 1;
    }
    P->mul[0] = 1;
    P->mul[1] = P->div_b[0];
    P->mul[2] = P->div_b[0] * P->div_b[1];
    const unsigned long long cells = (unsigned long long)P->div_b[0] * (unsigned long long)P->div_b[1] * (unsigned long long)P->div_b[2];
    P->invalid_key = cells < 0xFFFFFFFFull ? (uint32_t)cells : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(kFB) void voxel_key_kernel(const float4* __restrict__ pts, size_t n, int dense, const VoxelParams* __restrict__ P,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const float inv = P->inv_leaf;
    const int ijk0 = (int)(floorf(p.x * inv) - (float)P->min_b[0]);
    const int ijk1 = (int)(floorf(p.y * inv) - (float)P->min_b[1]);
    const int ijk2 = (int)(floorf(p.z * inv) - (float)P->min_b[2]);
    const uint32_t idx = (uint32_t)(ijk0 * P->mul[0] + ijk1 * P->mul[1] + ijk2 * P->mul[2]);
    keys[i] = (dense || finite3(p)) ? idx : P->invalid_key;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kFB) void voxel_head_kernel(const uint32_t* __restrict__ keys, size_t n, const VoxelParams* __restrict__ P, int dense,
                                                         uint32_t* __restrict__ head) {
    const size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    const bool valid = dense || k != P->invalid_key;  // invalid keys sort behind every cell
    head[i] = (valid && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

// start[v] = sorted position of voxel v's first point; start[n_voxels] = number of valid points.
__global__ __launch_bounds__(kFB) void voxel_starts_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ head,
                                                           const uint32_t* __restrict__ rank, size_t n, int dense, uint32_t* __restrict__ start,
                                                           VoxelParams* P) {
    const size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = head[i], r = rank[i];
    if (h) start[r] = (uint32_t)i;
    const uint32_t inv = P->invalid_key;
    const bool valid = dense || keys[i] != inv;
    const bool next_valid = i + 1 < n && (dense || keys[i + 1] != inv);
    if (valid && !next_valid) {  // last valid point: closes the last run
        start[r + h] = (uint32_t)(i + 1);
        P->n_out = r + h;
    }
}

// One thread per voxel: CentroidPoint's float32 running sums in sorted (= input) order, then one division each. The gathers
// of a run do not depend on the sums, so they are issued four at a time.
// n_voxels_dev (optional): the number of voxels as the device knows it — the launch is then sized by an upper bound and the host
// learns the count only from the single read-back at the end of the filter.
__global__ __launch_bounds__(kFB) void voxel_centroid_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ vals,
                                                             const uint32_t* __restrict__ start, uint32_t n_voxels, float4* __restrict__ out,
                                                             const VoxelParams* __restrict__ n_voxels_dev) {
    const uint32_t v = blockIdx.x * kFB + threadIdx.x;
    if (n_voxels_dev) n_voxels = n_voxels_dev->status == 0 ? n_voxels_dev->n_out : 0u;
    if (v >= n_voxels) return;
    const uint32_t b = start[v], e = start[v + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    uint32_t j = b;
    for (; j + 4 <= e; j += 4) {
        const uint32_t i0 = vals[j], i1 = vals[j + 1], i2 = vals[j + 2], i3 = vals[j + 3];
        const float4 p0 = pts[i0], p1 = pts[i1], p2 = pts[i2], p3 = pts[i3];
        sx += p0.x; sy += p0.y; sz += p0.z; si += p0.w;
        sx += p1.x; sy += p1.y; sz += p1.z; si += p1.w;
        sx += p2.x; sy += p2.y; sz += p2.z; si += p2.w;
        sx += p3.x; sy += p3.y; sz += p3.z; si += p3.w;
    }
    for (; j < e; ++j) {
        const float4 p = pts[vals[j]];
        sx += p.x; sy += p.y; sz += p.z; si += p.w;
    }
    const float cnt = (float)(e - b);
    out[v] = float4{sx / cnt, sy / cnt, sz / cnt, si / cnt};
}

// ---- order-preserving stream compaction (CropBox, removeNaN): count per tile → exclusive scan of the tile counts →
// scatter with the predicate re-evaluated. 16 B read twice + 16 B written per survivor; no flag array, no atomics.
struct CropPred {  // CropBox::applyFilter (identity transform): inclusive bounds, written as PCL's "outside" test so NaN behaves alike
    float mnx, mny, mnz, mxx, mxy, mxz;
    int dense;
    __device__ __forceinline__ bool operator()(const float4& p) const {
        const bool skip = !dense && !finite3(p);
        const bool outside = (p.x < mnx || p.y < mny || p.z < mnz) || (p.x > mxx || p.y > mxy || p.z > mxz);
        return !skip && !outside;
    }
};
struct FinitePred {
    __device__ __forceinline__ bool operator()(const float4& p) const { return finite3(p); }
};

constexpr int kTileItems = 4;                  // points per thread
constexpr int kTile = kFB * kTileItems;        // points per block; item k of thread t is tile_base + k·kFB + t (coalesced)

template <class Pred>
__global__ __launch_bounds__(kFB) void compact_count_kernel(const float4* __restrict__ pts, size_t n, Pred pred, uint32_t* __restrict__ tile_count) {
    const size_t base = (size_t)blockIdx.x * kTile;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        const size_t i = base + (size_t)k * kFB + threadIdx.x;
        if (i < n) c += pred(pts[i]) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    __shared__ uint32_t s_c[kFB / 64];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kFB / 64; ++w) t += s_c[w];
        tile_count[blockIdx.x] = t;
    }
}

template <class Pred>
__global__ __launch_bounds__(kFB) void compact_scatter_kernel(const float4* __restrict__ pts, size_t n, Pred pred, const uint32_t* __restrict__ tile_offset,
                                                              const uint32_t* __restrict__ tile_count, uint32_t n_tiles, float4* __restrict__ out,
                                                              VoxelParams* P) {
    const size_t base = (size_t)blockIdx.x * kTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t s_seg[kTileItems * (kFB / 64)];
    float4 p[kTileItems];
    bool keep[kTileItems];
    unsigned long long ballot[kTileItems];
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        const size_t i = base + (size_t)k * kFB + threadIdx.x;
        keep[k] = false;
        if (i < n) { p[k] = pts[i]; keep[k] = pred(p[k]); }
        ballot[k] = __ballot(keep[k]);
        if (lane == 0) s_seg[k * (kFB / 64) + wave] = (uint32_t)__popcll(ballot[k]);
    }
    __syncthreads();
    const uint32_t tile_base = tile_offset[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        uint32_t seg = 0;
        const int me = k * (kFB / 64) + wave;
        for (int q = 0; q < kTileItems * (kFB / 64); ++q) seg += q < me ? s_seg[q] : 0u;
        const uint32_t within = (uint32_t)__popcll(ballot[k] & ((1ull << lane) - 1ull));
        if (keep[k]) out[tile_base + seg + within] = p[k];
    }
    if (blockIdx.x == n_tiles - 1 && threadIdx.x == 0) P->n_out = tile_base + tile_count[blockIdx.x];
}

struct M34 { double v[12]; };  // row-major 3×4, passed by value

__global__ __launch_bounds__(kFB) void transform_cloud_f64_kernel(const float4* __restrict__ src, size_t n, int dense, M34 m, float4* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    float4 o = p;
    if (dense || finite3(p)) {
        const double x = p.x, y = p.y, z = p.z;
        o.x = (float)(((m.v[0] * x + m.v[1] * y) + m.v[2] * z) + m.v[3]);
        o.y = (float)(((m.v[4] * x + m.v[5] * y) + m.v[6] * z) + m.v[7]);
        o.z = (float)(((m.v[8] * x + m.v[9] * y) + m.v[10] * z) + m.v[11]);
    }
    dst[i] = o;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + kFB - 1) / kFB); }

FilterScratch* scratch(locgpu_ctx* ctx) {
    if (!ctx->filt) ctx->filt = new FilterScratch();
    return ctx->filt;
}

#define LOCGPU_TRY(expr)                   \
    do {                                   \
        const hipError_t e__ = (expr);     \
        if (e__ != hipSuccess) return e__; \
    } while (0)

hipError_t ensure_scratch(locgpu_ctx* ctx, size_t n) {
    FilterScratch* S = scratch(ctx);
    if (!S->d_params) {
        LOCGPU_TRY(hipMalloc((void**)&S->d_params, sizeof(VoxelParams)));
        LOCGPU_TRY(hipHostMalloc((void**)&S->h_params, sizeof(VoxelParams)));
        LOCGPU_TRY(hipMalloc((void**)&S->d_partial, kMinMaxBlocks * 6 * sizeof(float)));
    }
    if (n <= S->cap) return hipSuccess;
    const size_t cap = n + n / 4 + 1024;
    for (int j = 0; j < 2; ++j) {
        if (S->keys[j]) (void)hipFree(S->keys[j]);
        if (S->vals[j]) (void)hipFree(S->vals[j]);
        S->keys[j] = S->vals[j] = nullptr;
    }
    if (S->head) (void)hipFree(S->head);
    if (S->rank) (void)hipFree(S->rank);
    if (S->temp) (void)hipFree(S->temp);
    S->head = S->rank = nullptr; S->temp = nullptr; S->cap = 0;
    for (int j = 0; j < 2; ++j) {
        LOCGPU_TRY(hipMalloc((void**)&S->keys[j], cap * sizeof(uint32_t)));
        LOCGPU_TRY(hipMalloc((void**)&S->vals[j], cap * sizeof(uint32_t)));
    }
    LOCGPU_TRY(hipMalloc((void**)&S->head, cap * sizeof(uint32_t)));
    LOCGPU_TRY(hipMalloc((void**)&S->rank, cap * sizeof(uint32_t)));
    size_t b1 = 0, b2 = 0;
    LOCGPU_TRY(prim::sort_pairs(nullptr, b1, S->keys[0], S->keys[1], S->vals[0], S->vals[1], (int)cap, 0, 32, ctx->stream));
    LOCGPU_TRY(prim::exclusive_sum(nullptr, b2, S->head, S->rank, (int)cap, ctx->stream));
    S->temp_bytes = std::max(b1, b2) + 256;
    LOCGPU_TRY(hipMalloc(&S->temp, S->temp_bytes));
    S->cap = cap;
    return hipSuccess;
}

// Scratch cloud buffer that results are produced into before being swapped into `out` (which may own the input).
hipError_t ensure_tmp(locgpu_ctx* ctx, size_t n) {
    FilterScratch* S = scratch(ctx);
    if (n <= S->tmp_cap && S->d_tmp) return hipSuccess;
    if (S->d_tmp) (void)hipFree(S->d_tmp);
    S->d_tmp = nullptr; S->tmp_cap = 0;
    const size_t cap = n + n / 4 + 1024;
    LOCGPU_TRY(hipMalloc((void**)&S->d_tmp, cap * sizeof(float4)));
    S->tmp_cap = cap;
    return hipSuccess;
}

// Hands the scratch result buffer (holding n points) to `out` and takes out's old storage as the new scratch.
void swap_in(locgpu_ctx* ctx, locgpu_cloud* out, size_t n, int dense) {
    FilterScratch* S = scratch(ctx);
    std::swap(out->d, S->d_tmp);
    std::swap(out->cap, S->tmp_cap);
    out->n = n;
    out->is_dense = dense;
}

hipError_t read_params(locgpu_ctx* ctx) {
    FilterScratch* S = scratch(ctx);
    LOCGPU_TRY(hipMemcpyAsync(S->h_params, S->d_params, sizeof(VoxelParams), hipMemcpyDeviceToHost, ctx->stream));
    return hipStreamSynchronize(ctx->stream);
}

hipError_t copy_through(locgpu_ctx* ctx, const locgpu_cloud* in, locgpu_cloud* out) {
    if (in == out) return hipSuccess;
    LOCGPU_TRY(cloud_reserve(out, in->n, false));
    if (in->n) LOCGPU_TRY(hipMemcpyAsync(out->d, in->d, in->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    out->n = in->n;
    out->is_dense = in->is_dense;
    return hipSuccess;
}

}  // namespace

void filters_free(locgpu_ctx* ctx) {
    FilterScratch* S = ctx->filt;
    if (!S) return;
    for (int j = 0; j < 2; ++j) {
        if (S->keys[j]) (void)hipFree(S->keys[j]);
        if (S->vals[j]) (void)hipFree(S->vals[j]);
    }
    if (S->head) (void)hipFree(S->head);
    if (S->rank) (void)hipFree(S->rank);
    if (S->temp) (void)hipFree(S->temp);
    if (S->d_params) (void)hipFree(S->d_params);
    if (S->h_params) (void)hipHostFree(S->h_params);
    if (S->d_partial) (void)hipFree(S->d_partial);
    if (S->d_tmp) (void)hipFree(S->d_tmp);
    if (S->stage_ev) { (void)hipEventSynchronize(S->stage_ev); (void)hipEventDestroy(S->stage_ev); }
    if (S->h_stage) (void)hipHostFree(S->h_stage);
    delete S;
    ctx->filt = nullptr;
}

hipError_t cloud_reserve(locgpu_cloud* c, size_t n, bool keep) {
    if (n <= c->cap && c->d) return hipSuccess;
    const size_t cap = n + n / 4 + 1024;
    float4* d = nullptr;
    LOCGPU_TRY(hipMalloc((void**)&d, cap * sizeof(float4)));
    if (keep && c->d && c->n) {
        const hipError_t e = hipMemcpyAsync(d, c->d, c->n * sizeof(float4), hipMemcpyDeviceToDevice, c->ctx->stream);
        if (e != hipSuccess) { (void)hipFree(d); return e; }
        LOCGPU_TRY(hipStreamSynchronize(c->ctx->stream));
    }
    if (c->d) (void)hipFree(c->d);
    c->d = d;
    c->cap = cap;
    return hipSuccess;
}

hipError_t cloud_mark_ready(locgpu_cloud* c) {
    if (!c || !c->ctx) return hipErrorInvalidValue;
    if (!c->ready) LOCGPU_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    return hipEventRecord(c->ready, c->ctx->stream);
}

// hipErrorInvalidDevice: the cloud lives on another GPU (the caller's refusal); any other error is HIP's own.
// Consumers only READ the cloud, and every entry point that takes a foreign cloud returns with its stream synchronised (the align
// calls and locgpu_submap_add_keyframe are blocking), so the owner's later writes cannot overtake them; a non-blocking consumer would
// have to record an event here for the owner to wait on.
hipError_t cloud_input_ready(locgpu_ctx* ctx, const locgpu_cloud* c) {
    if (!c || !c->ctx) return hipErrorInvalidValue;
    if (c->ctx == ctx) return hipSuccess;
    if (c->ctx->device != ctx->device) return hipErrorInvalidDevice;
    LOCGPU_TRY(hipSetDevice(ctx->device));  // the fallback event below must be created on the consumer's device
    // behind the call that produced the cloud — not behind whatever its owner has enqueued since (a filter stage running a scan
    // ahead would otherwise hold the matcher back: the two stages would take turns instead of overlapping)
    if (c->ready) return hipStreamWaitEvent(ctx->stream, c->ready, 0);
    if (!ctx->foreign_ev) LOCGPU_TRY(hipEventCreateWithFlags(&ctx->foreign_ev, hipEventDisableTiming));
    LOCGPU_TRY(hipEventRecord(ctx->foreign_ev, c->ctx->stream));
    return hipStreamWaitEvent(ctx->stream, ctx->foreign_ev, 0);
}

hipError_t cloud_stage(locgpu_ctx* ctx, size_t n, float4** out) {
    FilterScratch* S = scratch(ctx);
    if (S->stage_busy) {  // an upload returned with its copy still in flight (cloud_stage_release)
        LOCGPU_TRY(hipEventSynchronize(S->stage_ev));
        S->stage_busy = false;
    }
    if (n > S->stage_cap || !S->h_stage) {
        if (S->h_stage) (void)hipHostFree(S->h_stage);
        S->h_stage = nullptr; S->stage_cap = 0;
        const size_t cap = n + n / 4 + 1024;
        LOCGPU_TRY(hipHostMalloc((void**)&S->h_stage, cap * sizeof(float4)));
        S->stage_cap = cap;
    }
    *out = S->h_stage;
    return hipSuccess;
}

// The staging buffer has been handed to an asynchronous copy on the context's stream: the next cloud_stage() waits for it.
hipError_t cloud_stage_release(locgpu_ctx* ctx) {
    FilterScratch* S = scratch(ctx);
    if (!S->stage_ev) LOCGPU_TRY(hipEventCreateWithFlags(&S->stage_ev, hipEventDisableTiming));
    LOCGPU_TRY(hipEventRecord(S->stage_ev, ctx->stream));
    S->stage_busy = true;
    return hipSuccess;
}

hipError_t voxel_filter_dev(locgpu_ctx* ctx, const locgpu_cloud* in, float leaf, locgpu_cloud* out, int* status) {
    const size_t n = in->n;
    *status = 0;
    if (n == 0) { *status = 2; out->n = 0; out->is_dense = 1; return hipSuccess; }
    LOCGPU_TRY(ensure_scratch(ctx, n + 1));  // + 1: keys[0] is reused below as `start`, which has n_voxels + 1 ≤ n + 1 entries
    FilterScratch* S = scratch(ctx);
    hipStream_t s = ctx->stream;
    const int dense = in->is_dense;
    const float inv = 1.0f / leaf;  // inverse_leaf_size_
    const int n_partial = (int)std::min<size_t>(blocks_for(n), (size_t)kMinMaxBlocks);
    hipLaunchKernelGGL(minmax_kernel, dim3(n_partial), dim3(kFB), 0, s, in->d, n, dense, S->d_partial);
    hipLaunchKernelGGL(voxel_setup_kernel, dim3(1), dim3(64), 0, s, S->d_params, S->d_partial, n_partial, inv);
    LOCGPU_TRY(hipGetLastError());
    // A scan-sized cloud is filtered with ONE host read-back (the output size, at the end) instead of two: the per-scan path of the
    // front-ends is latency-bound, and each read-back is a stream synchronisation. The sort then covers all 32 key bits (the grid
    // size is only known on the device), the centroid launch is sized by the input, and the two rare outcomes the set-up kernel
    // reports — no finite point, leaf too small — are acted on after the fact (the kernels in between ran on stale parameters,
    // within their buffers). A map-sized cloud keeps the early read-back: there the sort's key width matters.
    const bool one_readback = n < (1u << 20);
    int end_bit = 32;
    if (!one_readback) {
        LOCGPU_TRY(read_params(ctx));
        const VoxelParams hp = *S->h_params;
        if (hp.status == 2) { *status = 2; out->n = 0; out->is_dense = 1; return hipSuccess; }
        if (hp.status == 1) { *status = 1; return copy_through(ctx, in, out); }  // "Leaf size is too small…": output = *input_
        end_bit = 1;
        while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)hp.invalid_key) ++end_bit;
    }
    hipLaunchKernelGGL(voxel_key_kernel, dim3(blocks_for(n)), dim3(kFB), 0, s, in->d, n, dense, S->d_params, S->keys[0], S->vals[0]);
    size_t tb = S->temp_bytes;
    LOCGPU_TRY(prim::sort_pairs(S->temp, tb, S->keys[0], S->keys[1], S->vals[0], S->vals[1], (int)n, 0, end_bit, s));
    hipLaunchKernelGGL(voxel_head_kernel, dim3(blocks_for(n)), dim3(kFB), 0, s, S->keys[1], n, S->d_params, dense, S->head);
    tb = S->temp_bytes;
    LOCGPU_TRY(prim::exclusive_sum(S->temp, tb, S->head, S->rank, (int)n, s));
    uint32_t* start = S->keys[0];  // free again after the sort; capacity ≥ n + 1 (ensure_scratch above)
    hipLaunchKernelGGL(voxel_starts_kernel, dim3(blocks_for(n)), dim3(kFB), 0, s, S->keys[1], S->head, S->rank, n, dense, start, S->d_params);
    LOCGPU_TRY(hipGetLastError());
    if (one_readback) {
        LOCGPU_TRY(ensure_tmp(ctx, n));
        hipLaunchKernelGGL(voxel_centroid_kernel, dim3(blocks_for(n)), dim3(kFB), 0, s, in->d, S->vals[1], start, 0u, S->d_tmp, S->d_params);
        LOCGPU_TRY(hipGetLastError());
        LOCGPU_TRY(read_params(ctx));
        const VoxelParams hp = *S->h_params;
        if (hp.status == 2) { *status = 2; out->n = 0; out->is_dense = 1; return hipSuccess; }
        if (hp.status == 1) { *status = 1; return copy_through(ctx, in, out); }  // "Leaf size is too small…": output = *input_
        swap_in(ctx, out, hp.n_out, 1);  // applyFilter: output.is_dense = true
        return hipSuccess;
    }
    LOCGPU_TRY(read_params(ctx));
    const uint32_t n_voxels = S->h_params->n_out;
    LOCGPU_TRY(ensure_tmp(ctx, n_voxels));
    if (n_voxels)
        hipLaunchKernelGGL(voxel_centroid_kernel, dim3(blocks_for(n_voxels)), dim3(kFB), 0, s, in->d, S->vals[1], start, n_voxels, S->d_tmp, (const VoxelParams*)nullptr);
    LOCGPU_TRY(hipGetLastError());
    swap_in(ctx, out, S->h_params->n_out, 1);  // applyFilter: output.is_dense = true
    return hipSuccess;
}

template <class Pred>
static hipError_t compact(locgpu_ctx* ctx, const locgpu_cloud* in, Pred pred, locgpu_cloud* out) {
    const size_t n = in->n;
    LOCGPU_TRY(ensure_scratch(ctx, n));
    LOCGPU_TRY(ensure_tmp(ctx, n));
    FilterScratch* S = scratch(ctx);
    hipStream_t s = ctx->stream;
    const unsigned n_tiles = (unsigned)((n + kTile - 1) / kTile);
    uint32_t *tile_count = S->head, *tile_offset = S->rank;
    hipLaunchKernelGGL((compact_count_kernel<Pred>), dim3(n_tiles), dim3(kFB), 0, s, in->d, n, pred, tile_count);
    size_t tb = S->temp_bytes;
    LOCGPU_TRY(prim::exclusive_sum(S->temp, tb, tile_count, tile_offset, (int)n_tiles, s));
    hipLaunchKernelGGL((compact_scatter_kernel<Pred>), dim3(n_tiles), dim3(kFB), 0, s, in->d, n, pred, tile_offset, tile_count, n_tiles, S->d_tmp, S->d_params);
    LOCGPU_TRY(hipGetLastError());
    LOCGPU_TRY(read_params(ctx));
    swap_in(ctx, out, S->h_params->n_out, 1);  // Filter<PointT>::filter copies the survivors; CropBox / removeNaN mark the result dense
    return hipSuccess;
}

hipError_t crop_box_dev(locgpu_ctx* ctx, const locgpu_cloud* in, const float mn[3], const float mx[3], locgpu_cloud* out) {
    if (in->n == 0) { out->n = 0; out->is_dense = 1; return hipSuccess; }
    return compact(ctx, in, CropPred{mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], in->is_dense}, out);
}

hipError_t remove_nan_dev(locgpu_ctx* ctx, const locgpu_cloud* in, locgpu_cloud* out) {
    if (in->is_dense || in->n == 0) {  // removeNaNFromPointCloud trusts the flag: a dense cloud is copied as it is
        const hipError_t e = copy_through(ctx, in, out);
        if (in->n == 0) { out->n = 0; out->is_dense = 1; }
        return e;
    }
    return compact(ctx, in, FinitePred{}, out);
}

hipError_t transform_dev(locgpu_ctx* ctx, const locgpu_cloud* in, const double pose[7], locgpu_cloud* out) {
    // Lio::AddCloud calls pcl::transformPointCloud(*scan, *key_frame_scan, pose.matrix()) with a DOUBLE 4×4 (lio.cpp:244,279):
    // PCL 1.8's templated overload then evaluates m00·x + m01·y + m02·z + m03 in double, left to right, and stores float;
    // on a cloud that is not flagged dense it leaves non-finite points as they are. (ScanMatch's output cloud uses the
    // float overload instead — pose.matrix().cast<float>(), icp cpp:241 — that one is transform_cloud_kernel.)
    LOCGPU_TRY(ensure_scratch(ctx, 1));
    FilterScratch* S = scratch(ctx);
    double R[9];
    quat_to_R(pose, R);
    M34 m;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m.v[4 * r + c] = R[3 * r + c];
        m.v[4 * r + 3] = pose[4 + r];
    }
    const size_t n = in->n;
    if (in == out) {
        LOCGPU_TRY(ensure_tmp(ctx, n));
    } else {
        LOCGPU_TRY(cloud_reserve(out, n, false));
    }
    float4* dst = in == out ? S->d_tmp : out->d;
    if (n) hipLaunchKernelGGL(transform_cloud_f64_kernel, dim3(blocks_for(n)), dim3(kFB), 0, ctx->stream, in->d, n, in->is_dense, m, dst);
    LOCGPU_TRY(hipGetLastError());
    const int dense = in->is_dense;
    if (in == out) swap_in(ctx, out, n, dense);
    else { out->n = n; out->is_dense = dense; }
    return hipSuccess;
}

hipError_t append_dev(locgpu_ctx* ctx, locgpu_cloud* dst, const locgpu_cloud* src) {
    // pcl::PointCloud::operator+= : points appended, is_dense = both dense
    const size_t n0 = dst->n, n1 = src->n;
    LOCGPU_TRY(cloud_reserve(dst, n0 + n1, true));
    if (n1) LOCGPU_TRY(hipMemcpyAsync(dst->d + n0, src->d, n1 * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    dst->n = n0 + n1;
    dst->is_dense = (dst->is_dense && src->is_dense) ? 1 : 0;
    return hipSuccess;
}

}  // namespace locgpu
