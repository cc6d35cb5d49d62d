// loc_lib_amd/host/include/locgpu_facade/cloud_ops.hpp — GPU versions of the free functions of
// LocUtils/include/LocUtils/common/point_cloud_utils.h that sit on the per-scan path. That header also holds PCD I/O and is
// not shadowed; a maintainer replaces the body of RemoveNanPoint (:13-20) with `return gpu::RemoveNanPoint(input);`.
#pragma once
#include "locgpu_facade/types.hpp"

namespace LocUtils {
namespace gpu {

// pcl::removeNaNFromPointCloud as RemoveNanPoint calls it: a new cloud; a cloud flagged dense is copied unchanged.
CloudPtr RemoveNanPoint(const CloudPtr& input, int device_id = 0);

}  // namespace gpu
}  // namespace LocUtils
