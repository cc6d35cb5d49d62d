// Drop-in for LocUtils/include/LocUtils/model/cloud_filter/voxel_filter.hpp:11-20 (+ src voxel_filter.cpp:9-25): the
// pcl::VoxelGrid<PointXYZI> wrapper every scan and every local map goes through, here on the GPU (locgpu_voxel_filter).
#pragma once
#include "LocUtils/model/cloud_filter/cloud_filter_interface.hpp"

struct locgpu_ctx;

namespace LocUtils {

class VoxelFilter : public CloudFilterInterface {
public:
    VoxelFilter(float voxel_size = 0.5);
    ~VoxelFilter() override;
    VoxelFilter(const VoxelFilter&) = delete;
    VoxelFilter& operator=(const VoxelFilter&) = delete;

    // input and output may be the same cloud (lio.cpp:300 filters local_map_ in place, loc.cpp:218 the current scan)
    bool Filter(const CloudPtr& input_cloud_ptr, CloudPtr& filtered_cloud_ptr) override;

    // additions to the reference's surface
    void SetDevice(int device_id) { device_id_ = device_id; }
    const char* LastError() const;

private:
    float leaf_;
    int device_id_ = 0;
    locgpu_ctx* ctx_ = nullptr;
};

}  // namespace LocUtils
