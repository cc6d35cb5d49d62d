// loc_lib_amd/csrc/grid_build.hpp — host-side exact-search grid ingest (see grid_build.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace locgpu {

struct SearchGrid {
    std::vector<uint32_t> cell_start;  // [nx*ny*nz + 1], x fastest
    std::vector<float> points;         // [num_points][4] = x, y, z, bits(tree leaf slot)
    size_t num_points = 0;
    int dims[3] = {0, 0, 0};
    float origin[3] = {0, 0, 0};
    float cell = 1.f, inv_cell = 1.f;
};

// slots: the packed KD-tree (kdtree_build.cpp layout). Returns false and sets err on failure.
bool build_search_grid(const uint64_t* slots, size_t n_slots, SearchGrid& g, std::string& err);

}  // namespace locgpu
