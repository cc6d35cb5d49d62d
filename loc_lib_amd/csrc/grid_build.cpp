// loc_lib_amd/csrc/grid_build.cpp — host-side ingest of the exact-search grid (LOCGPU_SEARCH_GRID_EXACT).
//
// The grid answers the reference's EXACT k-NN (KdTree with approximate_ = false, reachable through
// KdtreeRegistration::SetEnableANN(false), kdtree.cpp:285-288 / :227-235). To return what that tree would, it indexes exactly
// the points the tree holds: its leaves — duplicate points that the reference's degenerate-split rule drops
// (kdtree.cpp:76-81,118-120) are not in the grid either — and every grid point carries its leaf's slot in the packed tree,
// so the fit/accumulate kernels gather neighbours the same way in both search modes.
//
// Layout: a dense array of cell start offsets (x fastest, then y, then z) over the leaves' bounding box, and the leaves as
// float4 {x, y, z, slot} sorted by cell. Because x is the fastest axis, the points of a run of consecutive cells of one row
// are one contiguous range: a query's 3×3×3 neighbourhood is 9 contiguous ranges.
// The cell edge is chosen so that an occupied cell holds ≈3 points: for points on surfaces the 3×3×3 block then holds the
// 5 nearest neighbours of almost every query (≈30 candidates) and larger rings are rare.
#include "grid_build.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_set>

namespace locgpu {

bool build_search_grid(const uint64_t* slots, size_t n_slots, SearchGrid& g, std::string& err) {
    g = SearchGrid();
    // 1. leaves of the packed tree
    std::vector<float> xyz;
    std::vector<uint32_t> slot_of;
    for (size_t i = 0; i < n_slots;) {
        const uint32_t meta = (uint32_t)(slots[i] >> 32);
        if ((meta >> 30) == 3u) {
            float x, y, z;
            const uint32_t xb = (uint32_t)slots[i], yb = (uint32_t)slots[i + 1], zb = (uint32_t)(slots[i + 1] >> 32);
            std::memcpy(&x, &xb, 4); std::memcpy(&y, &yb, 4); std::memcpy(&z, &zb, 4);
            if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) { err = "grid search needs finite target coordinates"; return false; }
            xyz.push_back(x); xyz.push_back(y); xyz.push_back(z);
            slot_of.push_back((uint32_t)i);
            i += 2;
        } else {
            i += 1;
        }
    }
    const size_t n = slot_of.size();
    if (n == 0) { err = "empty tree"; return false; }
    float lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], xyz[3 * i + a]); hi[a] = std::max(hi[a], xyz[3 * i + a]); }

    // 2. cell edge: the smallest of a geometric ladder for which an occupied cell holds ≥3 points on average. Occupancy is
    //    measured exactly on a 1/64 sample of the CELLS (chosen by key hash), one pass over all points per candidate.
    const double ext = std::max({(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2], 1e-3});
    double cell = ext;
    for (double c = ext / 8192.0; c < ext; c *= 1.3) {
        std::unordered_set<uint64_t> occ;
        size_t pts_in_sampled = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t ix = (uint64_t)((xyz[3 * i] - lo[0]) / c), iy = (uint64_t)((xyz[3 * i + 1] - lo[1]) / c), iz = (uint64_t)((xyz[3 * i + 2] - lo[2]) / c);
            uint64_t k = (ix << 42) | (iy << 21) | iz, h = k;
            h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
            if ((h & 63) != 0 && n > 100000) continue;
            occ.insert(k);
            ++pts_in_sampled;
        }
        cell = c;
        if (occ.empty() || (double)pts_in_sampled / (double)occ.size() >= 3.0) break;
    }
    for (;;) {
        const double nx = std::floor(((double)hi[0] - lo[0]) / cell) + 1, ny = std::floor(((double)hi[1] - lo[1]) / cell) + 1, nz = std::floor(((double)hi[2] - lo[2]) / cell) + 1;
        if (nx * ny * nz <= (double)(1u << 28) && nx < 2097152 && ny < 2097152 && nz < 2097152) break;
        cell *= 1.26;
    }
    g.cell = (float)cell;
    g.inv_cell = 1.0f / g.cell;
    for (int a = 0; a < 3; ++a) g.origin[a] = lo[a];
    // cell of a coordinate, float32 arithmetic shared with the kernel: floor((v − origin) · inv_cell), clamped
    auto cell_of = [&](float v, int a, int dim) {
        int c = (int)std::floor((v - g.origin[a]) * g.inv_cell);
        return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
    };
    for (int a = 0; a < 3; ++a) g.dims[a] = (int)std::floor((hi[a] - g.origin[a]) * g.inv_cell) + 1;
    const size_t n_cells = (size_t)g.dims[0] * g.dims[1] * g.dims[2];

    // 3. counting sort by linear cell index (x fastest)
    g.cell_start.assign(n_cells + 1, 0);
    std::vector<uint32_t> cell_idx(n);
    for (size_t i = 0; i < n; ++i) {
        const size_t cx = cell_of(xyz[3 * i], 0, g.dims[0]), cy = cell_of(xyz[3 * i + 1], 1, g.dims[1]), cz = cell_of(xyz[3 * i + 2], 2, g.dims[2]);
        const size_t c = (cz * g.dims[1] + cy) * g.dims[0] + cx;
        cell_idx[i] = (uint32_t)c;
        g.cell_start[c + 1]++;
    }
    for (size_t c = 0; c < n_cells; ++c) g.cell_start[c + 1] += g.cell_start[c];
    g.points.resize(4 * n);
    std::vector<uint32_t> cursor(g.cell_start.begin(), g.cell_start.end() - 1);
    for (size_t i = 0; i < n; ++i) {  // stable: points keep their tree (preorder) order inside a cell
        const size_t dst = cursor[cell_idx[i]]++;
        g.points[4 * dst] = xyz[3 * i]; g.points[4 * dst + 1] = xyz[3 * i + 1]; g.points[4 * dst + 2] = xyz[3 * i + 2];
        std::memcpy(&g.points[4 * dst + 3], &slot_of[i], 4);
    }
    g.num_points = n;
    return true;
}

}  // namespace locgpu

// Host-only test hook (not part of include/locgpu.h): tree + grid ingest of a cloud, copied out for the CPU test-suite.
// params = {origin x,y,z, cell, inv_cell}; returns the number of grid points (= tree leaves) or 0.
#include "kdtree_build.hpp"
extern "C" __attribute__((visibility("default"))) size_t locgpu_debug_build_grid(const float* xyz, size_t n, float* out_pts, size_t pts_cap,
                                                                                  uint32_t* out_cell_start, size_t cells_cap, int32_t dims[3],
                                                                                  float params[5]) {
    locgpu::PackedKdTree t;
    std::string err;
    if (!locgpu::build_packed_kdtree(xyz, n, t, err)) return 0;
    locgpu::SearchGrid g;
    if (!locgpu::build_search_grid(t.slots.data(), t.slots.size(), g, err)) return 0;
    for (int a = 0; a < 3; ++a) { dims[a] = g.dims[a]; params[a] = g.origin[a]; }
    params[3] = g.cell; params[4] = g.inv_cell;
    if (out_pts) std::memcpy(out_pts, g.points.data(), std::min(pts_cap, g.points.size()) * sizeof(float));
    if (out_cell_start) std::memcpy(out_cell_start, g.cell_start.data(), std::min(cells_cap, g.cell_start.size()) * sizeof(uint32_t));
    return g.num_points;
}
