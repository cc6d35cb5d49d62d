// Drop-in for LocUtils/include/LocUtils/model/cloud_filter/cloud_filter_interface.hpp:9-18 — the interface Loc and Lio hold
// their filters through (std::shared_ptr<CloudFilterInterface>, slam/3d/loc.hpp, lio.hpp; built at loc.cpp:112-117,
// lio.cpp:111-113).
#pragma once
#include "locgpu_facade/types.hpp"
#ifndef LOCGPU_FACADE_STANDALONE
#include "LocUtils/common/point_cloud_utils.h"  // the reference's header pulls this in for its users
#endif

namespace LocUtils {

class CloudFilterInterface {
public:
    virtual ~CloudFilterInterface() = default;
    virtual bool Filter(const CloudPtr& input_cloud_ptr, CloudPtr& filtered_cloud_ptr) = 0;
};

}  // namespace LocUtils
