// Drop-in for LocUtils/include/LocUtils/model/matching/3d/matching_interface.h:13-54 — the plugin boundary the
// front-ends hold as std::shared_ptr<MatchingInterface> (slam/3d/loc.hpp:85, lio.hpp:112). Same virtuals, same
// defaults (every non-pure virtual answers `true`), so Loc / Lio / LoamRegistration compile against it unchanged.
#pragma once
#include "locgpu_facade/types.hpp"

namespace LocUtils {

class MatchingInterface {
public:
    virtual ~MatchingInterface() = default;

    virtual bool SetInputTarget(const CloudPtr& /*input_target*/) { return true; }
    // H and B only — the hook LoamRegistration uses (loam_registration.cpp:56,66)
    virtual bool CaculateMatrixHAndB(const CloudPtr& /*input_source*/, const SE3& /*predict_pose*/, Mat6d& /*H*/, Vec6d& /*B*/) { return true; }
    virtual bool ScanMatch(const CloudPtr& /*input_source*/, const SE3& /*predict_pose*/, CloudPtr& /*result_cloud_ptr*/, SE3& /*result_pose*/) {
        return true;
    }
    // two-cloud (edge + surf) overloads used by the LOAM matcher
    virtual bool SetInputTarget(const CloudPtr& /*edge_input*/, const CloudPtr& /*surf_input*/) { return true; }
    virtual bool ScanMatch(const CloudPtr& /*edge_input*/, const CloudPtr& /*surf_input*/, const SE3& /*predict_pose*/,
                           CloudPtr& /*result_cloud_ptr*/, SE3& /*result_pose*/) {
        return true;
    }
    virtual float GetFitnessScore() = 0;
};

}  // namespace LocUtils
