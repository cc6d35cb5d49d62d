// Drop-in for the brute-force search plug-in: BfnnRegistration (LocUtils/include/LocUtils/model/search_point/bfnn/bfnn.h:11-37,
// bfnn.cpp:7-50) over the C ABI (locgpu_bfnn_set_target / locgpu_bfnn_knn).
#pragma once
#include "LocUtils/model/search_point/kdtree/kdtree.h"

namespace LocUtils {

class BfnnRegistration : public SearchPointInterface {
public:
    explicit BfnnRegistration(bool use_multi = false);
    ~BfnnRegistration() override;
    bool SetTargetCloud(const CloudPtr& cloud) override;
    std::vector<int> FindNearstPoints(const Vec3f& point, int k) override;
    // many queries at once (packed xyz); out = n*k indices. Not in the reference.
    bool FindNearstPointsBatch(const float* xyz, size_t n, int k, std::vector<int>& out);
    void FindCloud(const CloudPtr& cloud2, std::vector<std::pair<size_t, size_t>>& matches) override;  // empty, like bfnn.cpp:52-56

private:
    locgpu_ctx* ctx_ = nullptr;
};

}  // namespace LocUtils
