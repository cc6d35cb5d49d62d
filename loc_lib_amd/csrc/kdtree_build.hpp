// loc_lib_amd/csrc/kdtree_build.hpp — host-side packed KD-tree ingest (see kdtree_build.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace locgpu {

struct PackedKdTree {
    std::vector<uint64_t> slots;  // 8-byte slots, preorder (layout in kdtree_build.cpp)
    std::vector<uint32_t> leaf_slots;  // slot index of every leaf, in preorder (what the exact-search grid is built from on the device)
    size_t num_leaves = 0;        // KdTree::size_ (kdtree.h:124)
    size_t num_nodes = 0;         // internal + leaf nodes
    size_t num_points = 0;
    int depth = 0;                // root = level 1
    bool bounded = true;          // every coordinate and split threshold is finite and below 1e18 in magnitude (squared distances cannot overflow)
};

// xyz: n packed float32 triples. Returns false and sets err on failure.
bool build_packed_kdtree(const float* xyz, size_t n, PackedKdTree& out, std::string& err);

}  // namespace locgpu
