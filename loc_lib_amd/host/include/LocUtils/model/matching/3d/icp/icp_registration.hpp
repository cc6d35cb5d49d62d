// Drop-in for LocUtils/include/LocUtils/model/matching/3d/icp/icp_registration.hpp: same enum, option struct
// (field for field, same defaults — slam_demo only overwrites some of them, lio_matching_flow.cpp:32-38), constructors
// and virtuals. The work happens in liblocgpu.so (include/locgpu.h); nothing is computed on the CPU here.
// IcpMethod::PCLICP (a pass-through to pcl::IterativeClosestPoint in the reference, :385-399) is not on the
// accelerated path: constructing with it is accepted, matching calls report failure through the bool.
#pragma once
#include <memory>

#include "LocUtils/model/matching/3d/matching_interface.h"

struct locgpu_ctx;

namespace LocUtils {

enum class IcpMethod { P2P, P2LINE, P2PLANE, PCLICP };  // reference hpp:15-20

struct IcpOptions {  // reference hpp:22-39
    IcpOptions() {}
    IcpOptions(IcpMethod method) { method_ = method; }
    int max_iteration_ = 20;
    double max_nn_distance_ = 1.0;
    double max_plane_distance_ = 0.1;
    double max_line_distance_ = 0.5;
    int min_effective_pts_ = 10;
    double eps_ = 1e-2;
    double euc_fitness_eps_ = 0.36;
    bool use_initial_translation_ = true;
    bool use_ann{false};
    IcpMethod method_{IcpMethod::P2P};
};

class IcpRegistration : public MatchingInterface {
public:
    IcpRegistration();
    explicit IcpRegistration(IcpOptions options);
    ~IcpRegistration() override;
    IcpRegistration(const IcpRegistration&) = delete;
    IcpRegistration& operator=(const IcpRegistration&) = delete;

    bool SetInputTarget(const CloudPtr& input_target) override;
    bool CaculateMatrixHAndB(const CloudPtr& input_source, const SE3& predict_pose, Mat6d& H, Vec6d& B) override;
    bool ScanMatch(const CloudPtr& input_source, const SE3& predict_pose, CloudPtr& result_cloud_ptr, SE3& result_pose) override;
    float GetFitnessScore() override;

    // Which GPU the matcher lives on (default 0). Not in the reference; must be called before SetInputTarget.
    void SetDevice(int device_id);
    // Text of the last liblocgpu error (the reference only logs through glog) — or, when the options name a branch of the reference
    // that is not on the GPU path (IcpMethod::PCLICP, icp_registration.cpp:385-399; use_initial_translation_ = false, :273,311,351),
    // the refusal: SetInputTarget, CaculateMatrixHAndB and ScanMatch then return false and touch nothing, instead of quietly running
    // something else.
    const char* LastError() const;

private:
    bool EnsureContext();
    const char* Unsupported() const;  // nullptr when the options are on the GPU path
    IcpOptions options_;
    locgpu_ctx* ctx_ = nullptr;
    int device_id_ = 0;
    bool has_target_ = false;
};

}  // namespace LocUtils
