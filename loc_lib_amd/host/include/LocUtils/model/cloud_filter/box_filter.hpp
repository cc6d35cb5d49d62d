// Drop-in for LocUtils/include/LocUtils/model/cloud_filter/box_filter.hpp:11-56 (+ src box_filter.cpp:14-72): the
// pcl::CropBox<PointXYZI> wrapper that cuts the local map out of the global map (loc.cpp:187-194), here on the GPU
// (locgpu_crop_box). The reference also declares `BoxFilter() = default`, which is ambiguous with the all-defaulted
// constructor below and therefore unusable; it is left out.
#pragma once
#include <vector>

#include "LocUtils/model/cloud_filter/cloud_filter_interface.hpp"

struct locgpu_ctx;

namespace LocUtils {

class BoxFilter : public CloudFilterInterface {
public:
    BoxFilter(float step_x = 150.f, float step_y = 150.f, float step_z = 150.f);
    ~BoxFilter() override;
    BoxFilter(const BoxFilter&) = delete;
    BoxFilter& operator=(const BoxFilter&) = delete;

    bool Filter(const CloudPtr& input_cloud_ptr, CloudPtr& filtered_cloud_ptr) override;
    void SetSize(std::vector<float> size);
    void SetOrigin(std::vector<float> origin);
    std::vector<float> GetEdge();

    void SetDevice(int device_id) { device_id_ = device_id; }
    const char* LastError() const;

private:
    void CalculateEdge();

    std::vector<float> origin_;
    std::vector<float> size_;
    std::vector<float> edge_;
    int device_id_ = 0;
    locgpu_ctx* ctx_ = nullptr;
};

}  // namespace LocUtils
