// Drop-in for LocUtils/include/LocUtils/model/matching/3d/ndt/ndt_registration.hpp: same enums, NdtOptions
// (field for field) and virtuals. DIRECT_NDT and INCREMENTAL_NDT run in liblocgpu.so. PCL_NDT is a no-op in the
// reference too (ndt_registration.cpp:69-70,246-247).
#pragma once
#include <cstddef>
#include <memory>

#include "LocUtils/model/matching/3d/matching_interface.h"

struct locgpu_ctx;

namespace LocUtils {

enum class NdtNearbyType { CENTER, NEARBY6 };                      // reference hpp:16-20
enum class NdtMethod { PCL_NDT, DIRECT_NDT, INCREMENTAL_NDT };    // reference hpp:21-26

struct NdtOptions {  // reference hpp:27-42
    int max_iteration_ = 20;
    double voxel_size_ = 1.0;
    double inv_voxel_size_ = 1.0;  // recomputed from voxel_size_ by the constructors, like the reference (cpp:15,25)
    int min_effective_pts_ = 10;
    int min_pts_in_voxel_ = 3;
    int max_pts_in_voxel_ = 50;
    double eps_ = 1e-2;
    double res_outlier_th_ = 20.0;
    bool remove_centroid_ = false;
    std::size_t capacity_ = 100000;
    NdtNearbyType nearby_type_ = NdtNearbyType::NEARBY6;
    NdtMethod method_{NdtMethod::DIRECT_NDT};
};

class NdtRegistration : public MatchingInterface {
public:
    NdtRegistration();
    explicit NdtRegistration(NdtOptions options);
    ~NdtRegistration() override;
    NdtRegistration(const NdtRegistration&) = delete;
    NdtRegistration& operator=(const NdtRegistration&) = delete;

    bool SetInputTarget(const CloudPtr& input_target) override;
    bool CaculateMatrixHAndB(const CloudPtr& input_source, const SE3& predict_pose, Mat6d& H, Vec6d& B) override;
    bool ScanMatch(const CloudPtr& input_source, const SE3& predict_pose, CloudPtr& result_cloud_ptr, SE3& result_pose) override;
    float GetFitnessScore() override;

    void SetDevice(int device_id);
    // Text of the last liblocgpu error — or the refusal when the options name a branch that is not on the GPU path (NdtMethod::PCL_NDT,
    // ndt_registration.cpp:69,246; remove_centroid_ = true, :380-384): SetInputTarget and ScanMatch then return false and touch nothing.
    const char* LastError() const;

private:
    bool EnsureContext();
    const char* Unsupported() const;  // nullptr when the options are on the GPU path
    NdtOptions options_;
    locgpu_ctx* ctx_ = nullptr;
    int device_id_ = 0;
    bool has_target_ = false;
};

}  // namespace LocUtils
