// loc_lib_amd/csrc/cloud_filters.hpp — device-resident clouds and the filters either side of the matcher.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

struct locgpu_ctx;

// A cloud resident in HBM: one float4 {x, y, z, intensity} per point. `is_dense` is pcl::PointCloud::is_dense — the
// PCL filters trust it (a dense cloud is never tested for NaN), so it travels with the data.
struct locgpu_cloud {
    locgpu_ctx* ctx = nullptr;
    float4* d = nullptr;
    size_t n = 0, cap = 0;
    int is_dense = 1;
    hipEvent_t ready = nullptr;  // recorded on the owner's stream behind the last call that wrote the cloud (cloud_mark_ready): what ANOTHER context waits for
};

namespace locgpu {

// Written by the setup kernel, read back by the host once per voxel filter.
struct VoxelParams {
    int32_t min_b[3], div_b[3], mul[3];
    uint32_t invalid_key;             // key given to non-finite points of a non-dense cloud (= number of cells)
    int32_t status;                   // 0 ok, 1 leaf too small (pass-through), 2 no finite point
    uint32_t n_out;                   // number of output points of the last filter
    float inv_leaf;
};

struct FilterScratch {
    size_t cap = 0;          // points
    uint32_t* keys[2] = {nullptr, nullptr};
    uint32_t* vals[2] = {nullptr, nullptr};
    uint32_t* head = nullptr;
    uint32_t* rank = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    VoxelParams* d_params = nullptr;
    VoxelParams* h_params = nullptr;  // pinned
    float* d_partial = nullptr;       // per-block bounding boxes of the min/max pass
    float4* d_tmp = nullptr;          // staging for in-place filters
    size_t tmp_cap = 0;
    float4* h_stage = nullptr;        // pinned host staging for upload/download
    hipEvent_t stage_ev = nullptr;  // recorded behind an upload's H2D: the staging buffer is free again once it has fired
    bool stage_busy = false;
    size_t stage_cap = 0;
};

// A cloud as a READ-ONLY input of context `ctx` (matcher source, keyframe, target): its own context's clouds need nothing (one
// stream); a cloud of ANOTHER context on the same GPU is accepted too — a front-end that uploads and filters scan i+1 on one
// context while another matches scan i — and `ctx`'s stream is then ordered behind everything the owning context has enqueued so
// far. The caller must not let the owner modify the cloud while it is in use here. Returns hipErrorInvalidValue for another device.
hipError_t cloud_input_ready(locgpu_ctx* ctx, const locgpu_cloud* c);
// Called by the owner's entry points behind every call that wrote `c`: records c->ready on the owner's stream.
hipError_t cloud_mark_ready(locgpu_cloud* c);
void filters_free(locgpu_ctx* ctx);
hipError_t cloud_reserve(locgpu_cloud* c, size_t n, bool keep);
hipError_t cloud_stage(locgpu_ctx* ctx, size_t n, float4** out);
hipError_t cloud_stage_release(locgpu_ctx* ctx);  // pinned staging of at least n points

// All run on ctx->stream and return after the result size is known (one small D2H + sync each).
// `out` may alias `in`'s owner (in-place): results are produced in scratch and swapped in.
hipError_t voxel_filter_dev(locgpu_ctx* ctx, const locgpu_cloud* in, float leaf, locgpu_cloud* out, int* status);
hipError_t crop_box_dev(locgpu_ctx* ctx, const locgpu_cloud* in, const float mn[3], const float mx[3], locgpu_cloud* out);
hipError_t remove_nan_dev(locgpu_ctx* ctx, const locgpu_cloud* in, locgpu_cloud* out);
hipError_t transform_dev(locgpu_ctx* ctx, const locgpu_cloud* in, const double pose[7], locgpu_cloud* out);
hipError_t append_dev(locgpu_ctx* ctx, locgpu_cloud* dst, const locgpu_cloud* src);

// loam_features.hip
void loam_free(locgpu_ctx* ctx);
hipError_t loam_extract_dev(locgpu_ctx* ctx, const locgpu_cloud* in, const unsigned char* ring, int num_scan, locgpu_cloud* edge, locgpu_cloud* surf,
                            bool* too_long);

}  // namespace locgpu
