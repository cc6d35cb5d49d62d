// loc_lib_amd/csrc/ndt_kernels.hpp — direct-NDT voxel table in HBM and its kernels' launchers.
//
// Replaces std::unordered_map<Eigen::Vector3i, NdtVoxelData> grids_ (ndt_registration.hpp:130) with an
// open-addressing hash table: 64-bit packed keys (21 bits per axis, biased) → dense voxel id → μ (3 f64) and
// info (9 f64, row-major). Only voxels the reference keeps (count > min_pts_in_voxel, ndt cpp:111,137) get an id.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hpp"

struct NdtTable {
    unsigned long long* d_keys = nullptr;  // [cap], kNdtEmpty when free
    int* d_vid = nullptr;                  // [cap] dense voxel id or -1
    double* d_mu = nullptr;                // [n_vox][3]
    double* d_info = nullptr;              // [n_vox][9]
    int* d_vox_key = nullptr;              // [n_vox][3] for dumps
    size_t cap = 0, n_vox = 0;
    double inv_voxel = 1.0;
    double res_outlier_th = 20.0;
    int n_nearby = 7;
};

namespace locgpu {

constexpr unsigned long long kNdtEmpty = ~0ull;
constexpr int kNdtBias = 1 << 20;

__host__ __device__ inline bool ndt_key_in_range(int x, int y, int z) {
    return x > -kNdtBias && x < kNdtBias && y > -kNdtBias && y < kNdtBias && z > -kNdtBias && z < kNdtBias;
}
__host__ __device__ inline unsigned long long ndt_pack(int x, int y, int z) {
    return ((unsigned long long)(unsigned)(x + kNdtBias) << 42) | ((unsigned long long)(unsigned)(y + kNdtBias) << 21) |
           (unsigned long long)(unsigned)(z + kNdtBias);
}
__host__ __device__ inline size_t ndt_hash(unsigned long long k, size_t cap_mask) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (size_t)k & cap_mask;
}

// Build (SetDirectNdtTargetCloud, ndt_registration.cpp:87-148). Returns hipError; *bad_key is set when a point falls
// outside the ±2^20-voxel key range.
hipError_t ndt_build(NdtTable& t, const float4* d_pts, size_t n, double voxel_size, int min_pts_in_voxel, hipStream_t s, bool* bad_key);
void ndt_table_free(NdtTable& t);

// K5: per-point 7-voxel probe + χ² gate + un-weighted JᵀJ / Jᵀe sums (AlignNdt inner loop, ndt cpp:399-433).
// active / n_active (optional): the scans to launch (SearchArgs::active); split_scans > 0: split the partial sums as a plain batch of
// that many scans would (AccumArgs::split_scans).
int launch_ndt_accum(const NdtTable* t, const float4* src, const int* counts, const PoseState* st, int max_n, int n_scans, double* partials,
                     hipStream_t s, const int* active = nullptr, int n_active = 0, int split_scans = 0, const int* src_of = nullptr);

}  // namespace locgpu
