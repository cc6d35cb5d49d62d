// oracle/locref_kdtree.hpp
//
// TEST INFRASTRUCTURE ONLY (see locref_math.hpp header). PARITY UNPINNED.
//
// CPU restatement of the reference's mean-split KD-tree and its alpha-pruned k-NN DFS:
//   build : LocUtils/src/model/search_point/kdtree/kdtree.cpp:10-31 (BuildTree), :58-94 (Insert),
//           :96-123 (FindSplitAxisAndThresh) + common/math_utils.h:35-47 (ComputeMeanAndCovDiag)
//   query : kdtree.cpp:147-167 (GetClosestPoint), :169-195 (Knn), :197-212 (ComputeDisForLeaf),
//           :214-236 (NeedExpand); defaults approximate_=true, alpha_=0.1f (kdtree.h:128-129)
// Written in the reference's own style on purpose (pointer nodes, std::priority_queue,
// a fresh std::vector per query) so that timing it is a fair stand-in for the reference CPU path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <queue>
#include <vector>

namespace locref {

struct F3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct KdNode {
    int id = -1;
    int point_idx = 0;
    int axis = 0;
    float thresh = 0.0f;
    KdNode* left = nullptr;
    KdNode* right = nullptr;
    bool IsLeaf() const { return left == nullptr && right == nullptr; }
};

struct NodeAndDistance {
    NodeAndDistance(const KdNode* n, float d2) : node(n), dist2(d2) {}
    const KdNode* node;
    float dist2;
    bool operator<(const NodeAndDistance& o) const { return dist2 < o.dist2; }
};

struct KnnStats {
    uint64_t nodes_visited = 0;   // every Knn() call (internal + leaf)
    uint64_t leaves_visited = 0;  // ComputeDisForLeaf calls
};

class KdTree {
public:
    // kdtree.cpp:10-31. Points are kept as f32 xyz (point_types.h:28-38).
    bool Build(const float* xyz, size_t n, size_t stride_floats) {
        if (n == 0) return false;
        cloud_.resize(n);
        for (size_t i = 0; i < n; ++i) cloud_[i] = {xyz[i * stride_floats], xyz[i * stride_floats + 1], xyz[i * stride_floats + 2]};
        pool_.clear();
        pool_.reserve(2 * n);  // the reference calls `new` per node; a pool keeps build time sane at 1e7
        size_ = 0;
        next_id_ = 0;
        depth_ = 0;
        root_ = NewNode();
        std::vector<int> idx(n);
        for (size_t i = 0; i < n; ++i) idx[i] = (int)i;
        Insert(idx, root_, 1);
        return true;
    }

    size_t size() const { return size_; }
    size_t num_nodes() const { return pool_.size(); }
    int depth() const { return depth_; }
    const KdNode* root() const { return root_; }
    const std::vector<F3>& cloud() const { return cloud_; }
    void SetEnableANN(bool use_ann, float alpha) { approximate_ = use_ann; alpha_ = alpha; }

    // kdtree.cpp:147-167. Returns false (empty result) when k > number of leaves.
    bool GetClosestPoint(const F3& pt, std::vector<int>& closest_idx, int k, KnnStats* st = nullptr) const {
        if ((size_t)k > size_) { closest_idx.clear(); return false; }
        std::priority_queue<NodeAndDistance> knn_result;
        Knn(pt, root_, knn_result, k, st);
        closest_idx.resize(knn_result.size());
        for (int i = (int)closest_idx.size() - 1; i >= 0; --i) {
            closest_idx[i] = knn_result.top().node->point_idx;
            knn_result.pop();
        }
        return true;
    }

    // Eigen (p1 - p2).squaredNorm() on Vector3f: redux order x0 + (x1 + x2), f32, no FMA.
    static inline float Dis2(const F3& a, const F3& b) {
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        return dx * dx + (dy * dy + dz * dz);
    }

private:
    KdNode* NewNode() {
        pool_.emplace_back();
        KdNode* n = &pool_.back();
        n->id = next_id_++;
        return n;
    }

    // kdtree.cpp:58-94
    void Insert(const std::vector<int>& points, KdNode* node, int level) {
        if (level > depth_) depth_ = level;
        if (points.empty()) return;
        if (points.size() == 1) {
            size_++;
            node->point_idx = points[0];
            return;
        }
        std::vector<int> left, right;
        if (!FindSplitAxisAndThresh(points, node->axis, node->thresh, left, right)) {
            size_++;
            node->point_idx = points[0];
            return;
        }
        if (!left.empty()) { node->left = NewNode(); Insert(left, node->left, level + 1); }
        if (!right.empty()) { node->right = NewNode(); Insert(right, node->right, level + 1); }
    }

    // kdtree.cpp:96-123 + math_utils.h:35-47: f32 sequential sums, /len, /(len-1), first arg-max.
    bool FindSplitAxisAndThresh(const std::vector<int>& point_idx, int& axis, float& th, std::vector<int>& left,
                                std::vector<int>& right) const {
        const size_t len = point_idx.size();
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int idx : point_idx) { const F3& p = cloud_[idx]; sx = sx + p.x; sy = sy + p.y; sz = sz + p.z; }
        const float flen = (float)len;
        const float mx = sx / flen, my = sy / flen, mz = sz / flen;
        float vx = 0.f, vy = 0.f, vz = 0.f;
        for (int idx : point_idx) {
            const F3& p = cloud_[idx];
            const float dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
            vx = vx + dx * dx; vy = vy + dy * dy; vz = vz + dz * dz;
        }
        const float flen1 = (float)(len - 1);
        vx = vx / flen1; vy = vy / flen1; vz = vz / flen1;
        // Eigen maxCoeff visitor: strictly-greater replaces, so the first maximum wins.
        axis = 0;
        float best = vx;
        if (vy > best) { best = vy; axis = 1; }
        if (vz > best) { best = vz; axis = 2; }
        th = axis == 0 ? mx : (axis == 1 ? my : mz);
        left.reserve(len / 2 + 1);
        right.reserve(len / 2 + 1);
        for (int idx : point_idx) {
            if (cloud_[idx][axis] < th) left.emplace_back(idx);
            else right.emplace_back(idx);
        }
        if (point_idx.size() > 1 && (left.empty() || right.empty())) return false;
        return true;
    }

    // kdtree.cpp:169-195
    void Knn(const F3& pt, const KdNode* node, std::priority_queue<NodeAndDistance>& res, int k, KnnStats* st) const {
        if (st) st->nodes_visited++;
        if (node->IsLeaf()) {
            ComputeDisForLeaf(pt, node, res, k, st);
            return;
        }
        const KdNode *this_side, *that_side;
        if (pt[node->axis] < node->thresh) { this_side = node->left; that_side = node->right; }
        else { this_side = node->right; that_side = node->left; }
        Knn(pt, this_side, res, k, st);
        if (NeedExpand(pt, node, res, k)) Knn(pt, that_side, res, k, st);
    }

    // kdtree.cpp:197-212
    void ComputeDisForLeaf(const F3& pt, const KdNode* node, std::priority_queue<NodeAndDistance>& res, int k,
                           KnnStats* st) const {
        if (st) st->leaves_visited++;
        const float dis2 = Dis2(pt, cloud_[node->point_idx]);
        if ((int)res.size() < k) {
            res.emplace(node, dis2);
        } else if (dis2 < res.top().dist2) {
            res.emplace(node, dis2);
            res.pop();
        }
    }

    // kdtree.cpp:214-236
    bool NeedExpand(const F3& pt, const KdNode* node, const std::priority_queue<NodeAndDistance>& res, int k) const {
        if ((int)res.size() < k) return true;
        const float d = pt[node->axis] - node->thresh;
        if (approximate_) return (d * d) < res.top().dist2 * alpha_;
        return (d * d) < res.top().dist2;
    }

    std::vector<F3> cloud_;
    std::vector<KdNode> pool_;  // reserve(2n) up front: pointers stay valid
    KdNode* root_ = nullptr;
    size_t size_ = 0;
    int next_id_ = 0;
    int depth_ = 0;
    bool approximate_ = true;
    float alpha_ = 0.1f;
};

}  // namespace locref
