// loc_lib_amd/csrc/grid_kernels.hpp — device view of the exact-search grid and its launchers (see grid_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.hpp"

namespace locgpu {

struct GridView {
    const uint32_t* cell_start = nullptr;  // [nx*ny*nz + 1]
    const float4* pts = nullptr;           // leaves sorted by cell: x, y, z, bits(tree leaf slot)
    int dims[3] = {0, 0, 0};
    float origin[3] = {0, 0, 0};
    float cell = 1.f, inv_cell = 1.f, slack = 0.f;
    size_t num_points = 0, bytes = 0;
};

// Search stage of one GN iteration in grid mode: grid kernel + exact tree kernel for the queries it could not settle.
bool launch_icp_search_grid(const GridView& grid, const SearchArgs& a, hipStream_t s);
bool launch_knn_grid_query(const GridView& grid, const uint2* tree, const float* q, size_t nq, int k, int32_t* out, unsigned int* n_flagged,
                           hipStream_t s);

}  // namespace locgpu
