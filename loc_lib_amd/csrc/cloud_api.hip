// loc_lib_amd/csrc/cloud_api.hip — C ABI of the device-resident clouds, the filters and the keyframe local map
// (include/locgpu.h, "Clouds resident in HBM …"). Kernels and device-side steps live in cloud_filters.hip.
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "cloud_filters.hpp"
#include "context.hpp"

using namespace locgpu;

// Lio's keyframe bookkeeping (lio.cpp:268-306) on resident clouds.
struct locgpu_submap {
    locgpu_ctx* ctx = nullptr;
    size_t num_kfs = 0;
    float leaf = 0.f;
    std::deque<locgpu_cloud*> scans;  // scans_in_local_map_ (world frame, unfiltered)
    locgpu_cloud* map = nullptr;      // local_map_
};

namespace {

int hip_fail(locgpu_ctx* ctx, hipError_t e, const char* what) {
    hip_ok(ctx, e, what);
    return e == hipErrorOutOfMemory ? LOCGPU_ERR_OOM : LOCGPU_ERR_NO_DEVICE;
}

bool same_ctx(const locgpu_cloud* a, const locgpu_cloud* b) { return a && b && a->ctx && a->ctx == b->ctx; }

locgpu_cloud* new_cloud(locgpu_ctx* ctx) {
    auto* c = new locgpu_cloud();
    c->ctx = ctx;
    return c;
}
void free_cloud(locgpu_cloud* c) {
    if (!c) return;
    if (c->d) (void)hipFree(c->d);
    if (c->ready) (void)hipEventDestroy(c->ready);
    delete c;
}

int upload(locgpu_cloud* c, const void* pts, size_t n, size_t stride, size_t ioff, int is_dense) {
    locgpu_ctx* ctx = c->ctx;
    if (n > 0x7FFFFF00u) return fail(ctx, LOCGPU_ERR_INVALID, "cloud_upload: more than 2^31 points");
    hipError_t e = cloud_reserve(c, n, false);
    if (e != hipSuccess) return hip_fail(ctx, e, "cloud_upload: hipMalloc");
    float4* stage = nullptr;
    e = cloud_stage(ctx, n, &stage);
    if (e != hipSuccess) return hip_fail(ctx, e, "cloud_upload: hipHostMalloc");
    const char* base = (const char*)pts;
    const bool has_i = ioff != LOCGPU_NO_INTENSITY;
    if (stride == sizeof(float4) && ioff == 12) {
        std::memcpy(stage, base, n * sizeof(float4));  // already {x, y, z, intensity} records (pcl::PointXYZI without its padding)
    } else {
        for (size_t i = 0; i < n; ++i) {
            float4 p{0.f, 0.f, 0.f, 0.f};
            std::memcpy(&p, base + i * stride, 12);
            if (has_i) std::memcpy(&p.w, base + i * stride + ioff, 4);
            stage[i] = p;
        }
    }
    if (n) {
        // The caller's points have been deep-copied (they may be freed now); the copy to HBM is only enqueued — whatever uses the
        // cloud next is ordered behind it on the same stream, and the next user of the staging buffer waits for it.
        e = hipMemcpyAsync(c->d, stage, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = cloud_stage_release(ctx);
        if (e != hipSuccess) return hip_fail(ctx, e, "cloud_upload: H2D");
    }
    c->n = n;
    c->is_dense = is_dense ? 1 : 0;
    return LOCGPU_OK;
}

int download(const locgpu_cloud* c, void* out, size_t stride, size_t ioff) {
    locgpu_ctx* ctx = c->ctx;
    const size_t n = c->n;
    if (n == 0) return LOCGPU_OK;
    float4* stage = nullptr;
    hipError_t e = cloud_stage(ctx, n, &stage);
    if (e != hipSuccess) return hip_fail(ctx, e, "cloud_download: hipHostMalloc");
    e = hipMemcpyAsync(stage, c->d, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "cloud_download: D2H");
    char* base = (char*)out;
    const bool has_i = ioff != LOCGPU_NO_INTENSITY;
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(base + i * stride, &stage[i], 12);
        if (has_i) std::memcpy(base + i * stride + ioff, &stage[i].w, 4);
    }
    return LOCGPU_OK;
}

bool layout_ok(size_t stride, size_t ioff) { return stride >= 12 && (ioff == LOCGPU_NO_INTENSITY || (ioff >= 12 && ioff + 4 <= stride)); }

// Shared body of the host-pointer one-shots: upload into the context's scratch cloud, run `step`, download.
template <class Step>
int one_shot(locgpu_ctx* ctx, const char* name, const void* pts, size_t n, size_t stride, size_t ioff, int is_dense, void* out, size_t* out_n,
             int* out_dense, Step step) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if ((n && (!pts || !out)) || !out_n || !layout_ok(stride, ioff)) return fail(ctx, LOCGPU_ERR_INVALID, std::string(name) + ": bad arguments");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    locgpu_cloud c;
    c.ctx = ctx;
    int rc = upload(&c, pts, n, stride, ioff, is_dense);
    if (rc == LOCGPU_OK) {
        e = step(&c);
        if (e != hipSuccess) rc = hip_fail(ctx, e, name);
    }
    if (rc == LOCGPU_OK) rc = download(&c, out, stride, ioff);
    if (rc == LOCGPU_OK) {
        *out_n = c.n;
        if (out_dense) *out_dense = c.is_dense;
    }
    if (c.d) (void)hipFree(c.d);
    return rc;
}

}  // namespace

extern "C" {

int locgpu_cloud_create(locgpu_ctx* ctx, locgpu_cloud** out) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    *out = new_cloud(ctx);
    return LOCGPU_OK;
}

void locgpu_cloud_destroy(locgpu_cloud* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    free_cloud(c);
}

int locgpu_cloud_upload(locgpu_cloud* c, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense) {
    if (!c) return LOCGPU_ERR_INVALID;
    if ((n && !pts) || !layout_ok(stride_bytes, intensity_offset)) return fail(c->ctx, LOCGPU_ERR_INVALID, "cloud_upload: bad arguments");
    LOCGPU_HIP(c->ctx, hipSetDevice(c->ctx->device));
    const int rc = upload(c, pts, n, stride_bytes, intensity_offset, is_dense);
    if (rc == LOCGPU_OK) (void)cloud_mark_ready(c);
    return rc;
}

int locgpu_cloud_info(const locgpu_cloud* c, size_t* n, int* is_dense) {
    if (!c) return LOCGPU_ERR_INVALID;
    if (n) *n = c->n;
    if (is_dense) *is_dense = c->is_dense;
    return LOCGPU_OK;
}

int locgpu_cloud_download(const locgpu_cloud* c, void* out, size_t capacity, size_t stride_bytes, size_t intensity_offset) {
    if (!c) return LOCGPU_ERR_INVALID;
    if ((c->n && !out) || capacity < c->n || !layout_ok(stride_bytes, intensity_offset)) return fail(c->ctx, LOCGPU_ERR_INVALID, "cloud_download: bad arguments or capacity < size");
    LOCGPU_HIP(c->ctx, hipSetDevice(c->ctx->device));
    return download(c, out, stride_bytes, intensity_offset);
}

int locgpu_cloud_copy(const locgpu_cloud* in, locgpu_cloud* out) {
    if (!in) return LOCGPU_ERR_INVALID;
    if (!same_ctx(in, out)) return fail(in->ctx, LOCGPU_ERR_INVALID, "cloud_copy: clouds of different contexts");
    if (in == out) return LOCGPU_OK;
    locgpu_ctx* ctx = in->ctx;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = cloud_reserve(out, in->n, false);
    if (e == hipSuccess && in->n) e = hipMemcpyAsync(out->d, in->d, in->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "cloud_copy");
    out->n = in->n;
    out->is_dense = in->is_dense;
    (void)cloud_mark_ready(out);
    return LOCGPU_OK;
}

int locgpu_cloud_remove_nan(const locgpu_cloud* in, locgpu_cloud* out) {
    if (!in) return LOCGPU_ERR_INVALID;
    if (!same_ctx(in, out)) return fail(in->ctx, LOCGPU_ERR_INVALID, "cloud_remove_nan: clouds of different contexts");
    LOCGPU_HIP(in->ctx, hipSetDevice(in->ctx->device));
    const hipError_t e = remove_nan_dev(in->ctx, in, out);
    if (e == hipSuccess) (void)cloud_mark_ready(out);
    return e == hipSuccess ? LOCGPU_OK : hip_fail(in->ctx, e, "cloud_remove_nan");
}

int locgpu_cloud_voxel_filter(const locgpu_cloud* in, float leaf, locgpu_cloud* out, int* passthrough) {
    if (!in) return LOCGPU_ERR_INVALID;
    if (!same_ctx(in, out)) return fail(in->ctx, LOCGPU_ERR_INVALID, "cloud_voxel_filter: clouds of different contexts");
    if (!(leaf > 0.f) || !std::isfinite(leaf)) return fail(in->ctx, LOCGPU_ERR_INVALID, "cloud_voxel_filter: leaf size must be positive");
    LOCGPU_HIP(in->ctx, hipSetDevice(in->ctx->device));
    int status = 0;
    const hipError_t e = voxel_filter_dev(in->ctx, in, leaf, out, &status);
    if (e != hipSuccess) return hip_fail(in->ctx, e, "cloud_voxel_filter");
    if (passthrough) *passthrough = status == 1 ? 1 : 0;
    (void)cloud_mark_ready(out);
    return LOCGPU_OK;
}

int locgpu_cloud_crop_box(const locgpu_cloud* in, const float min_xyz[3], const float max_xyz[3], locgpu_cloud* out) {
    if (!in) return LOCGPU_ERR_INVALID;
    if (!same_ctx(in, out) || !min_xyz || !max_xyz) return fail(in->ctx, LOCGPU_ERR_INVALID, "cloud_crop_box: bad arguments");
    LOCGPU_HIP(in->ctx, hipSetDevice(in->ctx->device));
    const hipError_t e = crop_box_dev(in->ctx, in, min_xyz, max_xyz, out);
    if (e == hipSuccess) (void)cloud_mark_ready(out);
    return e == hipSuccess ? LOCGPU_OK : hip_fail(in->ctx, e, "cloud_crop_box");
}

int locgpu_cloud_transform(const locgpu_cloud* in, const double pose[7], locgpu_cloud* out) {
    if (!in) return LOCGPU_ERR_INVALID;
    if (!same_ctx(in, out) || !pose) return fail(in->ctx, LOCGPU_ERR_INVALID, "cloud_transform: bad arguments");
    LOCGPU_HIP(in->ctx, hipSetDevice(in->ctx->device));
    const hipError_t e = transform_dev(in->ctx, in, pose, out);
    if (e == hipSuccess) (void)cloud_mark_ready(out);
    return e == hipSuccess ? LOCGPU_OK : hip_fail(in->ctx, e, "cloud_transform");
}

int locgpu_cloud_append(locgpu_cloud* dst, const locgpu_cloud* src) {
    if (!dst) return LOCGPU_ERR_INVALID;
    if (!same_ctx(dst, src) || dst == src) return fail(dst->ctx, LOCGPU_ERR_INVALID, "cloud_append: bad arguments");
    if (dst->n + src->n > 0x7FFFFF00u) return fail(dst->ctx, LOCGPU_ERR_INVALID, "cloud_append: more than 2^31 points");
    LOCGPU_HIP(dst->ctx, hipSetDevice(dst->ctx->device));
    const hipError_t e = append_dev(dst->ctx, dst, src);
    if (e == hipSuccess) (void)cloud_mark_ready(dst);
    return e == hipSuccess ? LOCGPU_OK : hip_fail(dst->ctx, e, "cloud_append");
}

int locgpu_voxel_filter(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense, float leaf, void* out,
                        size_t* out_n, int* out_is_dense) {
    if (ctx && (!(leaf > 0.f) || !std::isfinite(leaf))) return fail(ctx, LOCGPU_ERR_INVALID, "voxel_filter: leaf size must be positive");
    return one_shot(ctx, "voxel_filter", pts, n, stride_bytes, intensity_offset, is_dense, out, out_n, out_is_dense, [&](locgpu_cloud* c) {
        int status = 0;
        return voxel_filter_dev(ctx, c, leaf, c, &status);
    });
}

int locgpu_crop_box(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense, const float min_xyz[3],
                    const float max_xyz[3], void* out, size_t* out_n, int* out_is_dense) {
    if (ctx && (!min_xyz || !max_xyz)) return fail(ctx, LOCGPU_ERR_INVALID, "crop_box: bad arguments");
    return one_shot(ctx, "crop_box", pts, n, stride_bytes, intensity_offset, is_dense, out, out_n, out_is_dense,
                    [&](locgpu_cloud* c) { return crop_box_dev(ctx, c, min_xyz, max_xyz, c); });
}

int locgpu_remove_nan(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense, void* out, size_t* out_n,
                      int* out_is_dense) {
    return one_shot(ctx, "remove_nan", pts, n, stride_bytes, intensity_offset, is_dense, out, out_n, out_is_dense,
                    [&](locgpu_cloud* c) { return remove_nan_dev(ctx, c, c); });
}

// ---- keyframe local map ----
int locgpu_submap_create(locgpu_ctx* ctx, int num_kfs, float leaf, locgpu_submap** out) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    *out = nullptr;
    if (num_kfs < 1 || !(leaf > 0.f) || !std::isfinite(leaf)) return fail(ctx, LOCGPU_ERR_INVALID, "submap_create: num_kfs >= 1 and leaf > 0 required");
    auto* m = new locgpu_submap();
    m->ctx = ctx;
    m->num_kfs = (size_t)num_kfs;
    m->leaf = leaf;
    m->map = new_cloud(ctx);
    *out = m;
    return LOCGPU_OK;
}

void locgpu_submap_destroy(locgpu_submap* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    for (locgpu_cloud* c : m->scans) free_cloud(c);
    free_cloud(m->map);
    delete m;
}

int locgpu_submap_add_keyframe(locgpu_submap* m, const locgpu_cloud* scan, const double pose[7]) {
    if (!m) return LOCGPU_ERR_INVALID;
    locgpu_ctx* ctx = m->ctx;
    if (!scan || !scan->ctx) return fail(ctx, LOCGPU_ERR_INVALID, "submap_add_keyframe: bad cloud");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    { const hipError_t ce = cloud_input_ready(ctx, scan); if (ce == hipErrorInvalidDevice) return fail(ctx, LOCGPU_ERR_INVALID, "submap_add_keyframe: the cloud belongs to a context on another GPU"); if (!hip_ok(ctx, ce, "submap_add_keyframe: ordering behind the cloud's context")) return LOCGPU_ERR_NO_DEVICE; }
    // key_frame_scan = transformPointCloud(scan, pose.matrix())   lio.cpp:278-279
    locgpu_cloud* kf = new_cloud(ctx);
    hipError_t e;
    if (pose) {
        e = transform_dev(ctx, scan, pose, kf);
    } else {
        e = cloud_reserve(kf, scan->n, false);
        if (e == hipSuccess && scan->n) e = hipMemcpyAsync(kf->d, scan->d, scan->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream);
        kf->n = scan->n;
        kf->is_dense = scan->is_dense;
    }
    if (e != hipSuccess) { free_cloud(kf); return hip_fail(ctx, e, "submap_add_keyframe: transform"); }
    m->scans.push_back(kf);  // :281
    if (m->scans.size() > m->num_kfs) {  // :283-294 drop the oldest, rebuild from the retained keyframes
        (void)hipStreamSynchronize(ctx->stream);
        free_cloud(m->scans.front());
        m->scans.pop_front();
        m->map->n = 0;
        m->map->is_dense = 1;  // local_map_.reset(new PointCloudType)
        size_t total = 0;
        for (locgpu_cloud* c : m->scans) total += c->n;
        if (total > 0x7FFFFF00u) return fail(ctx, LOCGPU_ERR_INVALID, "submap_add_keyframe: local map exceeds 2^31 points");
        e = cloud_reserve(m->map, total, false);
        for (locgpu_cloud* c : m->scans)
            if (e == hipSuccess) e = append_dev(ctx, m->map, c);
    } else {  // :295-298 append to the already filtered map
        if (m->map->n + kf->n > 0x7FFFFF00u) return fail(ctx, LOCGPU_ERR_INVALID, "submap_add_keyframe: local map exceeds 2^31 points");
        e = append_dev(ctx, m->map, kf);
    }
    if (e != hipSuccess) return hip_fail(ctx, e, "submap_add_keyframe: append");
    int status = 0;
    e = voxel_filter_dev(ctx, m->map, m->leaf, m->map, &status);  // :300 local_map_filter_ptr_->Filter(local_map_, local_map_)
    if (e != hipSuccess) return hip_fail(ctx, e, "submap_add_keyframe: voxel filter");
    return LOCGPU_OK;
}

int locgpu_submap_cloud(locgpu_submap* m, locgpu_cloud** map) {
    if (!m || !map) return LOCGPU_ERR_INVALID;
    *map = m->map;
    return LOCGPU_OK;
}

int locgpu_submap_last_keyframe(locgpu_submap* m, locgpu_cloud** kf) {
    if (!m || !kf) return LOCGPU_ERR_INVALID;
    if (m->scans.empty()) return fail(m->ctx, LOCGPU_ERR_INVALID, "submap_last_keyframe: no keyframe yet");
    *kf = m->scans.back();
    return LOCGPU_OK;
}

int locgpu_submap_info(const locgpu_submap* m, int* n_keyframes, size_t* map_points) {
    if (!m) return LOCGPU_ERR_INVALID;
    if (n_keyframes) *n_keyframes = (int)m->scans.size();
    if (map_points) *map_points = m->map->n;
    return LOCGPU_OK;
}

// ---- LOAM feature picker (loam_features.hip) ----
int locgpu_cloud_loam_extract(const locgpu_cloud* in, const uint8_t* ring, int num_scan, locgpu_cloud* edge, locgpu_cloud* surf) {
    if (!in) return LOCGPU_ERR_INVALID;
    locgpu_ctx* ctx = in->ctx;
    if (!same_ctx(in, edge) || !same_ctx(in, surf) || edge == surf || edge == in || surf == in || (in->n && !ring))
        return fail(ctx, LOCGPU_ERR_INVALID, "cloud_loam_extract: bad arguments");
    if (num_scan < 1 || num_scan > 256) return fail(ctx, LOCGPU_ERR_INVALID, "cloud_loam_extract: num_scan must be in 1..256");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    bool too_long = false;
    const hipError_t e = loam_extract_dev(ctx, in, ring, num_scan, edge, surf, &too_long);
    if (e != hipSuccess) return hip_fail(ctx, e, "cloud_loam_extract");
    if (too_long) return fail(ctx, LOCGPU_ERR_INVALID, "cloud_loam_extract: a ring has more than 6 x 2048 points");
    (void)cloud_mark_ready(edge);
    (void)cloud_mark_ready(surf);
    return LOCGPU_OK;
}

int locgpu_loam_extract(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int intensity_is_u8, size_t ring_offset,
                        int num_scan, void* edge_out, size_t* n_edge, void* surf_out, size_t* n_surf, size_t out_stride_bytes,
                        size_t out_intensity_offset) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    const size_t isz = intensity_is_u8 ? 1 : 4;
    if ((n && (!pts || !edge_out || !surf_out)) || !n_edge || !n_surf || stride_bytes < 12 || ring_offset < 12 || ring_offset + 1 > stride_bytes ||
        (intensity_offset != LOCGPU_NO_INTENSITY && (intensity_offset < 12 || intensity_offset + isz > stride_bytes)) ||
        !layout_ok(out_stride_bytes, out_intensity_offset) || n > 0x7FFFFF00u)
        return fail(ctx, LOCGPU_ERR_INVALID, "loam_extract: bad arguments");
    if (num_scan < 1 || num_scan > 256) return fail(ctx, LOCGPU_ERR_INVALID, "loam_extract: num_scan must be in 1..256");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    locgpu_cloud in, edge, surf;
    in.ctx = edge.ctx = surf.ctx = ctx;
    int rc = LOCGPU_OK;
    hipError_t e = cloud_reserve(&in, n, false);
    float4* stage = nullptr;
    if (e == hipSuccess) e = cloud_stage(ctx, n, &stage);
    std::vector<unsigned char> ring(n);
    if (e == hipSuccess) {
        const char* base = (const char*)pts;
        for (size_t i = 0; i < n; ++i) {
            float4 p{0.f, 0.f, 0.f, 0.f};
            std::memcpy(&p, base + i * stride_bytes, 12);
            if (intensity_offset != LOCGPU_NO_INTENSITY) {
                if (intensity_is_u8) p.w = (float)*(const unsigned char*)(base + i * stride_bytes + intensity_offset);  // p.intensity = pt.intensity (:33)
                else std::memcpy(&p.w, base + i * stride_bytes + intensity_offset, 4);
            }
            stage[i] = p;
            ring[i] = *(const unsigned char*)(base + i * stride_bytes + ring_offset);
        }
        if (n) {
            e = hipMemcpyAsync(in.d, stage, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        in.n = n;
    }
    bool too_long = false;
    if (e == hipSuccess) e = loam_extract_dev(ctx, &in, ring.data(), num_scan, &edge, &surf, &too_long);
    if (e != hipSuccess) rc = hip_fail(ctx, e, "loam_extract");
    else if (too_long) rc = fail(ctx, LOCGPU_ERR_INVALID, "loam_extract: a ring has more than 6 x 2048 points");
    if (rc == LOCGPU_OK) rc = download(&edge, edge_out, out_stride_bytes, out_intensity_offset);
    if (rc == LOCGPU_OK) rc = download(&surf, surf_out, out_stride_bytes, out_intensity_offset);
    if (rc == LOCGPU_OK) { *n_edge = edge.n; *n_surf = surf.n; }
    if (in.d) (void)hipFree(in.d);
    if (edge.d) (void)hipFree(edge.d);
    if (surf.d) (void)hipFree(surf.d);
    return rc;
}

}  // extern "C"

