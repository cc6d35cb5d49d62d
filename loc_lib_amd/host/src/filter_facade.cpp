// loc_lib_amd/host/src/filter_facade.cpp — the reference's cloud filters as thin hosts of liblocgpu.so.
//
// Behaviour kept from the reference (file:line = reference):
//  * VoxelFilter::Filter returns true and may filter a cloud into itself (voxel_filter.cpp:19-25; lio.cpp:300);
//    the result is dense, height 1, and every point is a fresh PointXYZI with x, y, z, intensity set (PCL's
//    CentroidPoint::get) — unless PCL's "leaf size is too small" rule applies, then the output is a copy of the input;
//  * BoxFilter::Filter clears the output FIRST (box_filter.cpp:27), so filtering a cloud into itself yields an empty
//    cloud exactly as in the reference; the reference's function falls off its end without a return value — this one
//    returns true;
//  * edges are float32 sums of size and origin (box_filter.cpp:59-66).
#include <cstring>

#include "../../../include/locgpu.h"
#include "LocUtils/model/cloud_filter/box_filter.hpp"
#include "LocUtils/model/cloud_filter/voxel_filter.hpp"
#include "LocUtils/model/feature_extract/loam_feature_extract.hpp"
#include "locgpu_facade/cloud_ops.hpp"

namespace LocUtils {

namespace {
constexpr size_t kIntensityOffset = offsetof(PointType, intensity);
bool ensure(locgpu_ctx*& ctx, int device_id) { return ctx || locgpu_create(device_id, &ctx) == LOCGPU_OK; }

// Filter<PointT>::filter writes into a temporary when output aliases input. `run` fills `tmp`, pre-sized to the input
// with default-constructed points (data[3] = 1, padding zero): the library writes x, y, z and intensity of each result.
template <class Run>
bool filter_into(const CloudPtr& in, CloudPtr& out, Run run) {
    decltype(in->points) tmp;
    tmp.resize(in->points.size());
    size_t m = 0;
    int dense = 1;
    if (!run(tmp.data(), &m, &dense)) return false;
    tmp.resize(m);
#ifndef LOCGPU_FACADE_STANDALONE
    // pcl::Filter::filter copies these from the input (and VoxelGrid's "leaf size is too small" branch copies the whole cloud,
    // whose points carry nothing beyond x, y, z, intensity); the stand-in cloud type of the standalone build has no such members.
    out->header = in->header;
    out->sensor_origin_ = in->sensor_origin_;
    out->sensor_orientation_ = in->sensor_orientation_;
#endif
    out->points.swap(tmp);
    out->width = (unsigned)m;
    out->height = 1;
    out->is_dense = dense != 0;
    return true;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ VoxelFilter
VoxelFilter::VoxelFilter(float voxel_size) : leaf_(voxel_size) {}
VoxelFilter::~VoxelFilter() { locgpu_destroy(ctx_); }
const char* VoxelFilter::LastError() const { return locgpu_last_error(ctx_); }

bool VoxelFilter::Filter(const CloudPtr& input_cloud_ptr, CloudPtr& filtered_cloud_ptr) {
    if (!input_cloud_ptr || !filtered_cloud_ptr || !ensure(ctx_, device_id_)) return true;  // the reference returns true whatever happened
    const CloudPtr in = input_cloud_ptr;  // keeps the input alive when the output pointer is the same object
    filter_into(in, filtered_cloud_ptr, [&](PointType* out, size_t* m, int* dense) {
        return locgpu_voxel_filter(ctx_, in->points.data(), in->points.size(), sizeof(PointType), kIntensityOffset, in->is_dense ? 1 : 0, leaf_, out, m,
                                   dense) == LOCGPU_OK;
    });
    return true;
}

// ------------------------------------------------------------------------------------------------ BoxFilter
BoxFilter::BoxFilter(float step_x, float step_y, float step_z) {
    size_.resize(6);
    edge_.resize(6);
    origin_.resize(3);
    size_ = {-step_x, step_x, -step_y, step_y, -step_z, step_z};
    SetSize(size_);
}
BoxFilter::~BoxFilter() { locgpu_destroy(ctx_); }
const char* BoxFilter::LastError() const { return locgpu_last_error(ctx_); }

bool BoxFilter::Filter(const CloudPtr& input_cloud_ptr, CloudPtr& filtered_cloud_ptr) {
    if (!input_cloud_ptr || !filtered_cloud_ptr) return true;
    filtered_cloud_ptr->clear();  // box_filter.cpp:27 — before the input is read
    if (!ensure(ctx_, device_id_)) return true;
    const float mn[3] = {edge_.at(0), edge_.at(2), edge_.at(4)}, mx[3] = {edge_.at(1), edge_.at(3), edge_.at(5)};
    const CloudPtr in = input_cloud_ptr;
    filter_into(in, filtered_cloud_ptr, [&](PointType* out, size_t* m, int* dense) {
        return locgpu_crop_box(ctx_, in->points.data(), in->points.size(), sizeof(PointType), kIntensityOffset, in->is_dense ? 1 : 0, mn, mx, out, m,
                               dense) == LOCGPU_OK;
    });
    return true;
}

void BoxFilter::SetSize(std::vector<float> size) {
    size_ = size;
    CalculateEdge();
}

void BoxFilter::SetOrigin(std::vector<float> origin) {
    origin_ = origin;
    CalculateEdge();
}

void BoxFilter::CalculateEdge() {
    for (size_t i = 0; i < origin_.size(); ++i) {
        edge_.at(2 * i) = size_.at(2 * i) + origin_.at(i);
        edge_.at(2 * i + 1) = size_.at(2 * i + 1) + origin_.at(i);
    }
}

std::vector<float> BoxFilter::GetEdge() { return edge_; }

// ------------------------------------------------------------------------------------------------ LoamFeatureExtract
LoamFeatureExtract::LoamFeatureExtract(LoamFeatureOptions option) : option_(option) {}
LoamFeatureExtract::~LoamFeatureExtract() { locgpu_destroy(ctx_); }
const char* LoamFeatureExtract::LastError() const { return locgpu_last_error(ctx_); }

void LoamFeatureExtract::Extract(FullCloudPtr& pc_in, CloudPtr& pc_out_edge, CloudPtr& pc_out_surf) {
    if (!pc_in || !pc_out_edge || !pc_out_surf || pc_in->points.empty() || !ensure(ctx_, device_id_)) return;
    const size_t n = pc_in->points.size();
    decltype(pc_out_edge->points) edge(n), surf(n);  // default-constructed PointXYZI, x/y/z/intensity filled by the library
    size_t ne = 0, ns = 0;
    if (locgpu_loam_extract(ctx_, pc_in->points.data(), n, sizeof(FullPointType), offsetof(FullPointType, intensity), 1, offsetof(FullPointType, ring),
                            (int)option_.num_scan_, edge.data(), &ne, surf.data(), &ns, sizeof(PointType), kIntensityOffset) != LOCGPU_OK)
        return;
    pc_out_edge->points.insert(pc_out_edge->points.end(), edge.begin(), edge.begin() + ne);  // the reference push_back()s into whatever is there
    pc_out_surf->points.insert(pc_out_surf->points.end(), surf.begin(), surf.begin() + ns);
    pc_out_edge->width = (unsigned)pc_out_edge->points.size(); pc_out_edge->height = 1;
    pc_out_surf->width = (unsigned)pc_out_surf->points.size(); pc_out_surf->height = 1;
}

// ------------------------------------------------------------------------------------------------ RemoveNanPoint
namespace gpu {
CloudPtr RemoveNanPoint(const CloudPtr& input, int device_id) {
    CloudPtr output(new PointCloudType);
    if (!input) return output;
    static thread_local locgpu_ctx* ctx = nullptr;  // one context per calling thread, kept for the process lifetime
    static thread_local int ctx_device = -1;
    if (ctx && ctx_device != device_id) { locgpu_destroy(ctx); ctx = nullptr; }
    if (!ensure(ctx, device_id)) return output;
    ctx_device = device_id;
    filter_into(input, output, [&](PointType* out, size_t* m, int* dense) {
        return locgpu_remove_nan(ctx, input->points.data(), input->points.size(), sizeof(PointType), kIntensityOffset, input->is_dense ? 1 : 0, out, m,
                                 dense) == LOCGPU_OK;
    });
    return output;
}
}  // namespace gpu

}  // namespace LocUtils
