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

#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>
#include <cstring>

#include "context.hpp"
#include "launch.hpp"

namespace locgpu {

namespace {

constexpr int kFB = 256;

__device__ __forceinline__ uint32_t f2ord(float f) {  // monotone float → uint map (for atomicMin/atomicMax)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
__device__ __forceinline__ bool finite3(const float4& p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

__global__ void voxel_init_kernel(VoxelParams* P, float inv_leaf) {
    for (int a = 0; a < 3; ++a) { P->min_enc[a] = f2ord(FLT_MAX); P->max_enc[a] = f2ord(-FLT_MAX); }
    P->status = 0;
    P->n_out = 0;
    P->inv_leaf = inv_leaf;
}

// getMinMax3D: bounding box of the (finite, unless the cloud is flagged dense) points.
__global__ __launch_bounds__(kFB) void minmax_kernel(const float4* __restrict__ pts, size_t n, int dense, VoxelParams* P) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = (size_t)blockIdx.x * kFB + threadIdx.x; i < n; i += (size_t)gridDim.x * kFB) {
        const float4 p = pts[i];
        if (!dense && !finite3(p)) continue;
        const float c[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = c[a] < mn[a] ? c[a] : mn[a];
            mx[a] = c[a] > mx[a] ? c[a] : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            const float o1 = __shfl_xor(mn[a], off, 64), o2 = __shfl_xor(mx[a], off, 64);
            mn[a] = o1 < mn[a] ? o1 : mn[a];
            mx[a] = o2 > mx[a] ? o2 : mx[a];
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&P->min_enc[a], f2ord(mn[a]));
            atomicMax(&P->max_enc[a], f2ord(mx[a]));
        }
    }
}

// VoxelGrid::applyFilter's set-up: overflow test, min_b_, div_b_, divb_mul_ (voxel_grid.hpp), all in its float32/int arithmetic.
__global__ void voxel_setup_kernel(VoxelParams* P) {
    const float inv = P->inv_leaf;
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(P->min_enc[a]); mx[a] = ord2f(P->max_enc[a]); }
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

// One thread per run of equal keys: CentroidPoint's float32 running sums in sorted (= input) order, then one division each.
__global__ __launch_bounds__(kFB) void voxel_centroid_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ vals, const uint32_t* __restrict__ head,
                                                             const uint32_t* __restrict__ rank, size_t n, float4* __restrict__ out, VoxelParams* P) {
    const size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) P->n_out = rank[i] + head[i];
    if (!head[i]) return;
    const uint32_t k = keys[i];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    size_t j = i;
    do {
        const float4 p = pts[vals[j]];
        sx += p.x; sy += p.y; sz += p.z; si += p.w;
        ++j;
    } while (j < n && keys[j] == k);
    const float cnt = (float)(j - i);
    out[rank[i]] = float4{sx / cnt, sy / cnt, sz / cnt, si / cnt};
}

// CropBox::applyFilter (identity transform): inclusive bounds, written as PCL's "outside" test so that NaN behaves alike.
__global__ __launch_bounds__(kFB) void crop_flag_kernel(const float4* __restrict__ pts, size_t n, int dense, float mnx, float mny, float mnz, float mxx,
                                                        float mxy, float mxz, unsigned char* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const bool skip = !dense && !finite3(p);
    const bool outside = (p.x < mnx || p.y < mny || p.z < mnz) || (p.x > mxx || p.y > mxy || p.z > mxz);
    flags[i] = (!skip && !outside) ? 1 : 0;
}

__global__ __launch_bounds__(kFB) void finite_flag_kernel(const float4* __restrict__ pts, size_t n, unsigned char* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * kFB + threadIdx.x;
    if (i >= n) return;
    flags[i] = finite3(pts[i]) ? 1 : 0;
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
        LOCGPU_TRY(hipMalloc((void**)&S->d_m12, 12 * sizeof(float)));
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
    if (S->flags) (void)hipFree(S->flags);
    if (S->temp) (void)hipFree(S->temp);
    S->head = S->rank = nullptr; S->flags = nullptr; S->temp = nullptr; S->cap = 0;
    for (int j = 0; j < 2; ++j) {
        LOCGPU_TRY(hipMalloc((void**)&S->keys[j], cap * sizeof(uint32_t)));
        LOCGPU_TRY(hipMalloc((void**)&S->vals[j], cap * sizeof(uint32_t)));
    }
    LOCGPU_TRY(hipMalloc((void**)&S->head, cap * sizeof(uint32_t)));
    LOCGPU_TRY(hipMalloc((void**)&S->rank, cap * sizeof(uint32_t)));
    LOCGPU_TRY(hipMalloc((void**)&S->flags, cap));
    size_t b1 = 0, b2 = 0, b3 = 0;
    LOCGPU_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, S->keys[0], S->keys[1], S->vals[0], S->vals[1], (int)cap, 0, 32, ctx->stream));
    LOCGPU_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, S->head, S->rank, (int)cap, ctx->stream));
    LOCGPU_TRY(hipcub::DeviceSelect::Flagged(nullptr, b3, (float4*)nullptr, S->flags, (float4*)nullptr, &S->d_params->n_out, (int)cap, ctx->stream));
    S->temp_bytes = std::max(b1, std::max(b2, b3)) + 256;
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
    if (S->flags) (void)hipFree(S->flags);
    if (S->temp) (void)hipFree(S->temp);
    if (S->d_params) (void)hipFree(S->d_params);
    if (S->h_params) (void)hipHostFree(S->h_params);
    if (S->d_m12) (void)hipFree(S->d_m12);
    if (S->d_tmp) (void)hipFree(S->d_tmp);
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

hipError_t cloud_stage(locgpu_ctx* ctx, size_t n, float4** out) {
    FilterScratch* S = scratch(ctx);
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

hipError_t voxel_filter_dev(locgpu_ctx* ctx, const locgpu_cloud* in, float leaf, locgpu_cloud* out, int* status) {
    const size_t n = in->n;
    *status = 0;
    if (n == 0) { *status = 2; out->n = 0; out->is_dense = 1; return hipSuccess; }
    LOCGPU_TRY(ensure_scratch(ctx, n));
    FilterScratch* S = scratch(ctx);
    hipStream_t s = ctx->stream;
    const int dense = in->is_dense;
    const float inv = 1.0f / leaf;  // inverse_leaf_size_
    hipLaunchKernelGGL(voxel_init_kernel, dim3(1), dim3(1), 0, s, S->d_params, inv);
    hipLaunchKernelGGL(minmax_kernel, dim3(std::min(blocks_for(n), 2048u)), dim3(kFB), 0, s, in->d, n, dense, S->d_params);
    hipLaunchKernelGGL(voxel_setup_kernel, dim3(1), dim3(1), 0, s, S->d_params);
    LOCGPU_TRY(hipGetLastError());
    LOCGPU_TRY(read_params(ctx));
    const VoxelParams hp = *S->h_params;
    if (hp.status == 2) { *status = 2; out->n = 0; out->is_dense = 1; return hipSuccess; }
    if (hp.status == 1) { *status = 1; return copy_through(ctx, in, out); }  // "Leaf size is too small…": output = *input_
    int end_bit = 1;
    while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)hp.invalid_key) ++end_bit;
    hipLaunchKernelGGL(voxel_key_kernel, dim3(blocks_for(n)), dim3(kFB), 0, s, in->d, n, dense, S->d_params, S->keys[0], S->vals[0]);
    size_t tb = S->temp_bytes;
    LOCGPU_TRY(hipcub::DeviceRadixSort::SortPairs(S->temp, tb, S->keys[0], S->keys[1], S->vals[0], S->vals[1], (int)n, 0, end_bit, s));
    hipLaunchKernelGGL(voxel_head_kernel, dim3(blocks_for(n)), dim3(kFB), 0, s, S->keys[1], n, S->d_params, dense, S->head);
    tb = S->temp_bytes;
    LOCGPU_TRY(hipcub::DeviceScan::ExclusiveSum(S->temp, tb, S->head, S->rank, (int)n, s));
    LOCGPU_TRY(ensure_tmp(ctx, n));
    hipLaunchKernelGGL(voxel_centroid_kernel, dim3(blocks_for(n)), dim3(kFB), 0, s, in->d, S->keys[1], S->vals[1], S->head, S->rank, n, S->d_tmp, S->d_params);
    LOCGPU_TRY(hipGetLastError());
    LOCGPU_TRY(read_params(ctx));
    swap_in(ctx, out, S->h_params->n_out, 1);  // applyFilter: output.is_dense = true
    return hipSuccess;
}

static hipError_t select_flagged(locgpu_ctx* ctx, const locgpu_cloud* in, locgpu_cloud* out) {
    FilterScratch* S = scratch(ctx);
    const size_t n = in->n;
    LOCGPU_TRY(ensure_tmp(ctx, n));
    size_t tb = S->temp_bytes;
    LOCGPU_TRY(hipcub::DeviceSelect::Flagged(S->temp, tb, in->d, S->flags, S->d_tmp, &S->d_params->n_out, (int)n, ctx->stream));
    LOCGPU_TRY(read_params(ctx));
    swap_in(ctx, out, S->h_params->n_out, 1);  // Filter<PointT>::filter → copyPointCloud by indices; CropBox / removeNaN mark the result dense
    return hipSuccess;
}

hipError_t crop_box_dev(locgpu_ctx* ctx, const locgpu_cloud* in, const float mn[3], const float mx[3], locgpu_cloud* out) {
    const size_t n = in->n;
    if (n == 0) { out->n = 0; out->is_dense = 1; return hipSuccess; }
    LOCGPU_TRY(ensure_scratch(ctx, n));
    FilterScratch* S = scratch(ctx);
    hipLaunchKernelGGL(crop_flag_kernel, dim3(blocks_for(n)), dim3(kFB), 0, ctx->stream, in->d, n, in->is_dense, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], S->flags);
    LOCGPU_TRY(hipGetLastError());
    return select_flagged(ctx, in, out);
}

hipError_t remove_nan_dev(locgpu_ctx* ctx, const locgpu_cloud* in, locgpu_cloud* out) {
    if (in->is_dense || in->n == 0) {  // removeNaNFromPointCloud trusts the flag: a dense cloud is copied as it is
        const hipError_t e = copy_through(ctx, in, out);
        if (in->n == 0) { out->n = 0; out->is_dense = 1; }
        return e;
    }
    const size_t n = in->n;
    LOCGPU_TRY(ensure_scratch(ctx, n));
    FilterScratch* S = scratch(ctx);
    hipLaunchKernelGGL(finite_flag_kernel, dim3(blocks_for(n)), dim3(kFB), 0, ctx->stream, in->d, n, S->flags);
    LOCGPU_TRY(hipGetLastError());
    return select_flagged(ctx, in, out);
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
