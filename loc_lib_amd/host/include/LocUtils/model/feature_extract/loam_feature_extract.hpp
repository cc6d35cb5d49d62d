// Drop-in for LocUtils/include/LocUtils/model/feature_extract/loam_feature_extract.hpp:12-44 (+ src loam_feature_extract.cpp):
// the per-ring curvature feature picker Lio::AddCloud(FullCloudPtr) runs on every scan (lio.cpp:52,323), here on the GPU
// (locgpu_loam_extract). Same constructor, options and Extract signature; ExtractFromSector — a helper the reference exposes
// but only calls from Extract — is not offered separately.
#pragma once
#include <cstddef>

#include "locgpu_facade/types.hpp"

struct locgpu_ctx;

namespace LocUtils {

struct IdAndValue {
    IdAndValue() {}
    IdAndValue(int id, double value) : id_(id), value_(value) {}
    int id_ = 0;
    double value_ = 0;
};

struct LoamFeatureOptions {
    size_t num_scan_{16};
};

class LoamFeatureExtract {
public:
    LoamFeatureExtract(LoamFeatureOptions option);
    ~LoamFeatureExtract();
    LoamFeatureExtract(const LoamFeatureExtract&) = delete;
    LoamFeatureExtract& operator=(const LoamFeatureExtract&) = delete;

    // appends to pc_out_edge / pc_out_surf like the reference (push_back, :116,147)
    void Extract(FullCloudPtr& pc_in, CloudPtr& pc_out_edge, CloudPtr& pc_out_surf);

    void SetDevice(int device_id) { device_id_ = device_id; }
    const char* LastError() const;

private:
    LoamFeatureOptions option_;
    int device_id_ = 0;
    locgpu_ctx* ctx_ = nullptr;
};

}  // namespace LocUtils
