// loc_lib_amd/csrc/grid_kernels.hpp — the exact-search grid in HBM (LOCGPU_SEARCH_GRID_EXACT): device view, build, launchers.
//
// Layout (built on the device by grid_build.hip from the tree's leaves):
//   pts        float4 {x, y, z, bits(tree leaf slot)} sorted by (tile, cell inside the tile): a tile's leaves are one run, and so
//              are each of its cells'
//   tiles      one 140-byte record per OCCUPIED tile (4×4×4 cells): first leaf, and the 65 exclusive prefix sums of its cells'
//              leaf counts — only occupied tiles exist, so the cell edge is set by the data (≈4 leaves per occupied cell), not by
//              the size of a dense array over the map's bounding box
//   tile_hash  open-addressing hash, tile (linear index, x fastest) → record; a few MB, L2-resident
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "launch.hpp"

namespace locgpu {

constexpr int kGridTile = 4;                    // cells per tile edge
constexpr int kTileCells = kGridTile * kGridTile * kGridTile;
constexpr uint32_t kEmptyCell = 0xFFFFFFFFu;

struct TileRec {
    uint32_t pt_start;               // first leaf of the tile in `pts`
    uint32_t tile_lin;               // (tz·nty + ty)·ntx + tx
    uint16_t cstart[kTileCells + 2]; // cstart[c] = leaves of the tile in cells < c (cell c = z·16 + y·4 + x); [64] = the tile's leaf count
};
static_assert(sizeof(TileRec) == 140, "TileRec layout");

struct GridView {
    const uint2* tile_hash = nullptr;  // [tile_mask + 1] {tile_lin, record index}; key kEmptyCell = free
    uint32_t tile_mask = 0;
    const TileRec* tiles = nullptr;    // [n_tocc]
    uint32_t n_tocc = 0;               // occupied tiles
    const float4* pts = nullptr;       // leaves sorted by (tile, cell)
    int dims[3] = {0, 0, 0};           // cells per axis
    int tdims[3] = {0, 0, 0};          // tiles per axis
    float origin[3] = {0, 0, 0};
    float cell = 1.f, inv_cell = 1.f, slack = 0.f;
    size_t num_points = 0, num_cells = 0, bytes = 0;
    // per-iteration query binning: what a batch's scratch needs (GridSearchScratch); the context's own copy serves locgpu_knn
    uint32_t* tile_count = nullptr;  // [n_tocc + 1]
    void* scan_temp = nullptr;
    size_t scan_temp_bytes = 0;
};

struct GridBuffers {  // device allocations behind a GridView
    uint2* tile_hash = nullptr;
    TileRec* tiles = nullptr;
    float4* pts = nullptr;
    uint32_t* tile_count = nullptr;
    void* scan_temp = nullptr;
};

// Builds the grid from the packed tree's leaves (d_tree, d_leaf_slots). Returns hipSuccess or the failing call's error;
// `msg` explains a logical failure (e.g. non-finite coordinates), reported with hipErrorInvalidValue.
hipError_t grid_build_device(const uint2* d_tree, const uint32_t* d_leaf_slots, size_t n_leaves, hipStream_t s, GridBuffers& buf, GridView& view,
                             std::string& msg);
void grid_free(GridBuffers& buf);

struct GridSearchScratch {  // per batch: the queries of one iteration, binned by tile
    uint32_t* qkey;     // [pitch] tile key of query gi (kEmptyCell: not binned)
    uint2* sorted;      // [pitch] {gi, tile key} in tile order
    // per-tile query counts and the scan's workspace: per BATCH as well — alignments of several batches run at once on different
    // streams (a context-wide array here was overwritten by the batch next door: a memory fault with three alignments in flight)
    uint32_t* tile_count;  // [n_tocc + 1]
    void* scan_temp;       // [GridView::scan_temp_bytes]
};

// Search stage of one GN iteration in grid mode: bin by tile → tile kernel (LDS-staged candidate blocks) → ring walk for the
// queries the 3×3×3 block did not settle → exact tree kernel for what is still open / tied.
bool launch_icp_search_grid(const GridView& grid, const SearchArgs& a, const GridSearchScratch& scratch, hipStream_t s);
bool launch_knn_grid_query(const GridView& grid, const uint2* tree, const float* q, size_t nq, int k, int32_t* out, unsigned int* n_flagged,
                           hipStream_t s);

}  // namespace locgpu
