// Drop-in for LocUtils/include/LocUtils/model/matching/3d/loam/loam_registration.hpp (:22-60): LoamOption and
// LoamRegistration, the matcher slam_demo selects with `matching_method: 0`. It owns one P2Line matcher for edge points and one
// P2Plane matcher for surface points and runs its own Gauss–Newton loop on the SUM of their normal equations
// (loam_registration.cpp:38-99). Here both evaluations run on the GPU over resident batches; the 6×6 solve stays on the host.
#pragma once
#include <memory>

#include "LocUtils/model/matching/3d/icp/icp_registration.hpp"
#include "LocUtils/model/matching/3d/matching_interface.h"

#include "LocUtils/model/feature_extract/loam_feature_extract.hpp"  // LoamFeatureOptions

struct locgpu_ctx;
struct locgpu_batch;

namespace LocUtils {

struct LoamOption {  // reference hpp:22-36
    LoamFeatureOptions feature_option_;
    IcpOptions surf_icp_option_{IcpMethod::P2PLANE};
    IcpOptions edge_icp_option_{IcpMethod::P2LINE};
    int min_edge_pts_{20};
    int min_surf_pts_{20};
    int max_iteration_{20};
    bool use_edge_points_{true};
    bool use_surf_points_{true};
    double eps_{1e-3};
};

class LoamRegistration : public MatchingInterface {
public:
    LoamRegistration();
    explicit LoamRegistration(LoamOption option);
    ~LoamRegistration() override;
    LoamRegistration(const LoamRegistration&) = delete;
    LoamRegistration& operator=(const LoamRegistration&) = delete;

    using MatchingInterface::ScanMatch;
    using MatchingInterface::SetInputTarget;
    bool SetInputTarget(const CloudPtr& edge_input, const CloudPtr& surf_input) override;
    bool ScanMatch(const CloudPtr& edge_input, const CloudPtr& surf_input, const SE3& predict_pose, CloudPtr& result_cloud_ptr,
                   SE3& result_pose) override;
    float GetFitnessScore() override;
    void SetDevice(int device_id);

private:
    LoamOption options_;
    locgpu_ctx* edge_ctx_ = nullptr;  // icp_edge_ptr_ (P2Line)
    locgpu_ctx* surf_ctx_ = nullptr;  // icp_surf_ptr_ (P2Plane)
    // one-scan source batches, kept across ScanMatch calls (device buffers of a batch are a dozen hipMalloc/hipFree pairs)
    locgpu_batch* edge_batch_ = nullptr;
    locgpu_batch* surf_batch_ = nullptr;
    size_t edge_cap_ = 0, surf_cap_ = 0;
    int device_id_ = 0;
    bool has_edge_ = false, has_surf_ = false;
};

}  // namespace LocUtils
