// Drop-in for the search plug-in the reference's matcher owns: SearchPointInterface
// (LocUtils/include/LocUtils/model/search_point/search_point_interface.h:9-24) and KdtreeRegistration
// (kdtree/kdtree.h:134-156, kdtree.cpp:252-293). FindNearstPoints answers one query through the GPU tree — fine for
// tools and tests; the matchers themselves never call it per point (the whole cloud is searched in one launch).
#pragma once
#include <utility>
#include <vector>

#include "locgpu_facade/types.hpp"

struct locgpu_ctx;

namespace LocUtils {

class SearchPointInterface {
public:
    virtual ~SearchPointInterface() = default;
    virtual bool SetTargetCloud(const CloudPtr& cloud) = 0;
    virtual std::vector<int> FindNearstPoints(const Vec3f& point, int k) = 0;
    virtual void FindCloud(const CloudPtr& cloud2, std::vector<std::pair<size_t, size_t>>& matches) = 0;
    virtual void SetEnableANN(bool /*use_ann*/ = true, float /*alpha*/ = 0.1) {}
};

class KdtreeRegistration : public SearchPointInterface {
public:
    explicit KdtreeRegistration(bool use_multi = false);
    ~KdtreeRegistration() override;
    bool SetTargetCloud(const CloudPtr& cloud) override;
    std::vector<int> FindNearstPoints(const Vec3f& point, int k) override;
    // many queries at once (packed xyz); out = n*k indices. Not in the reference.
    bool FindNearstPointsBatch(const float* xyz, size_t n, int k, std::vector<int>& out);
    void FindCloud(const CloudPtr& cloud2, std::vector<std::pair<size_t, size_t>>& matches) override;  // empty, like kdtree.cpp:290-293
    void SetEnableANN(bool use_ann = true, float alpha = 0.1) override;

private:
    locgpu_ctx* ctx_ = nullptr;
    bool approximate_ = true;  // KdTree::approximate_ default (kdtree.h:128)
    float alpha_ = 0.1f;       // KdTree::alpha_ default (kdtree.h:129)
};

}  // namespace LocUtils
