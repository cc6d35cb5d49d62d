// oracle/locref_flat.hpp
//
// TEST INFRASTRUCTURE ONLY (see locref_math.hpp header). PARITY UNPINNED.
//
// "R2" of BASELINE.md: the SAME point-to-plane path as locref.cpp / locref_kdtree.hpp (R1, written in the reference's own
// allocation-heavy style) restated the way a CPU-minded engineer would ship it — the tree flattened into one array of 16-byte
// nodes in preorder, the result set a fixed array driven by std::push_heap / std::pop_heap (what std::priority_queue calls, so
// equal distances resolve identically), no per-query std::vector, plane fit on fixed arrays. Results are bit-identical to R1
// (tests/test_oracle_kat.py::test_flat_port_equals_reference_style); only the time differs. It exists so that the CPU baseline
// beside the GPU number is not flattered by the reference's mallocs: bench.py reports R1 (authoritative: the reference is
// single-threaded and allocates like this), R2, and R3 = R2 scan-parallel over native threads.
// Follows: kdtree.cpp:147-236 (query), icp_registration.cpp:161-213 (H, B), :345-381 (loop), math_utils.h:112-136 (FitPlane).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "locref_kdtree.hpp"

namespace locref {

struct FlatNode {  // leaf: axis = -1, left = point index
    float thresh;
    int32_t axis;
    int32_t left, right;
};

struct FlatHeapEntry {
    float dist2;
    int32_t point;
    bool operator<(const FlatHeapEntry& o) const { return dist2 < o.dist2; }
};

class FlatKdTree {
public:
    void FromTree(const KdTree& t) {
        nodes_.clear();
        nodes_.reserve(t.num_nodes());
        cloud_ = &t.cloud();
        size_ = t.size();
        Copy(t.root());
    }
    size_t size() const { return size_; }

    // kdtree.cpp:147-167: k nearest in ascending distance into out[0..k); returns how many (0 when k > number of leaves)
    int Knn(const F3& pt, int k, bool approximate, float alpha, int* out) const {
        if ((size_t)k > size_ || k > kMaxK) return 0;
        FlatHeapEntry heap[kMaxK + 1];
        int n = 0;
        Visit(0, pt, k, approximate, alpha, heap, n);
        const int cnt = n;
        for (int i = cnt - 1; i >= 0; --i) {
            out[i] = heap[0].point;
            std::pop_heap(heap, heap + n);
            --n;
        }
        return cnt;
    }
    static constexpr int kMaxK = 8;

private:
    int32_t Copy(const KdNode* nd) {
        const int32_t me = (int32_t)nodes_.size();
        nodes_.push_back({0.f, -1, 0, 0});
        if (nd->IsLeaf()) {
            nodes_[me].left = nd->point_idx;
            return me;
        }
        nodes_[me].thresh = nd->thresh;
        nodes_[me].axis = nd->axis;
        const int32_t l = Copy(nd->left);
        const int32_t r = Copy(nd->right);
        nodes_[me].left = l;
        nodes_[me].right = r;
        return me;
    }

    // kdtree.cpp:169-236 (Knn, ComputeDisForLeaf, NeedExpand) on the flat array
    void Visit(int32_t ni, const F3& pt, int k, bool approximate, float alpha, FlatHeapEntry* heap, int& n) const {
        const FlatNode& nd = nodes_[ni];
        if (nd.axis < 0) {
            const float dis2 = KdTree::Dis2(pt, (*cloud_)[nd.left]);
            if (n < k) {
                heap[n++] = {dis2, nd.left};
                std::push_heap(heap, heap + n);
            } else if (dis2 < heap[0].dist2) {
                heap[n++] = {dis2, nd.left};
                std::push_heap(heap, heap + n);
                std::pop_heap(heap, heap + n);
                --n;
            }
            return;
        }
        const float q = pt[nd.axis];
        const bool go_left = q < nd.thresh;
        Visit(go_left ? nd.left : nd.right, pt, k, approximate, alpha, heap, n);
        bool expand = n < k;
        if (!expand) {
            const float d = q - nd.thresh;
            expand = approximate ? (d * d) < heap[0].dist2 * alpha : (d * d) < heap[0].dist2;
        }
        if (expand) Visit(go_left ? nd.right : nd.left, pt, k, approximate, alpha, heap, n);
    }

    std::vector<FlatNode> nodes_;
    const std::vector<F3>* cloud_ = nullptr;
    size_t size_ = 0;
};

}  // namespace locref
