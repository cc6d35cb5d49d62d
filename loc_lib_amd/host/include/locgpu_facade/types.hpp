// loc_lib_amd/host/include/locgpu_facade/types.hpp
//
// Types the façade headers speak in.
//
// * Inside the reference's tree (slam_demo / LocUtils build, Eigen + Sophus + PCL present) compile WITHOUT
//   LOCGPU_FACADE_STANDALONE: the reference's own headers provide them —
//   LocUtils/common/eigen_types.h (SE3 = Sophus::SE3d :66, Mat6d :24, Vec6d :40, Vec3f) and
//   LocUtils/common/point_types.h (PointType = pcl::PointXYZI :18, PointCloudType :19, CloudPtr :20).
// * In this repository's image none of those libraries exist, so the façade is compiled and tested with
//   -DLOCGPU_FACADE_STANDALONE against the layout-compatible minimal types below: the façade only ever touches
//   `cloud->points.data()/size()/resize()`, `is_dense`, `width`, `height`, `sizeof(PointType)`, `pose.data()` (7 doubles: quaternion xyzw + translation, the
//   Sophus::SE3d::data() order), `H.data()` and `B.data()`.
#pragma once

#ifndef LOCGPU_FACADE_STANDALONE
#include "LocUtils/common/eigen_types.h"
#include "LocUtils/common/point_types.h"
#else
#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace LocUtils {

struct alignas(16) PointType {  // pcl::PointXYZI: 32 bytes, x y z at 0/4/8, intensity at 16
    float x = 0, y = 0, z = 0, pad0 = 1.f;
    float intensity = 0, pad1[3] = {0, 0, 0};
};
static_assert(sizeof(PointType) == 32, "pcl::PointXYZI layout");

struct PointCloudType {  // the members of pcl::PointCloud the façade touches
    std::vector<PointType> points;
    unsigned width = 0, height = 0;
    bool is_dense = true;
    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); width = 0; height = 0; }
    using Ptr = std::shared_ptr<PointCloudType>;
};
using CloudPtr = PointCloudType::Ptr;

struct alignas(16) FullPointType {  // LocUtils::FullPointType, point_types.h:65-78 (64 bytes)
    float x = 0, y = 0, z = 0, pad0 = 1.f;
    float range = 0, radius = 0;
    std::uint8_t intensity = 0, ring = 0, angle = 0;
    double time_span = 0, time_intervel = 0;
    float height = 0;
};
static_assert(sizeof(FullPointType) == 64, "FullPointType layout");
struct FullPointCloudType {
    std::vector<FullPointType> points;
    bool is_dense = true;
    using Ptr = std::shared_ptr<FullPointCloudType>;
};
using FullCloudPtr = FullPointCloudType::Ptr;

struct SE3 {  // Sophus::SE3d parameter layout
    double p[7] = {0, 0, 0, 1, 0, 0, 0};
    double* data() { return p; }
    const double* data() const { return p; }
};
struct Mat6d {  // symmetric here, so row- vs column-major does not matter
    double m[36] = {0};
    double* data() { return m; }
    const double* data() const { return m; }
    double& operator()(int r, int c) { return m[6 * r + c]; }
};
struct Vec6d {
    double v[6] = {0};
    double* data() { return v; }
    const double* data() const { return v; }
    double& operator[](int i) { return v[i]; }
};
struct Vec3f {
    float v[3] = {0, 0, 0};
    const float* data() const { return v; }
    float operator[](int i) const { return v[i]; }
};

}  // namespace LocUtils
#endif
