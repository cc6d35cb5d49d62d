// loc_lib_amd/csrc/ndt_kernels.hpp — direct-NDT voxel table in HBM and its kernels' launchers.
//
// Replaces std::unordered_map<Eigen::Vector3i, NdtVoxelData> grids_ (ndt_registration.hpp:130) with an open-addressing hash
// table of 16-byte slots {64-bit packed key (21 bits per axis, biased), dense voxel index} — four to a cache line, so a collision walk
// mostly stays in the line it started in — and a DENSE array of 128-byte records {μ (3 f64), info (9 f64, row-major), key}: one more
// cache line once the key has been found (rounds 1-4: key → voxel index → μ → info, three dependent gathers into four arrays; a
// record per table slot instead of a dense array was measured too: a four times larger footprint, 13 % slower). Only voxels the
// reference keeps (count > min_pts_in_voxel, ndt cpp:111,137) are in the table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hpp"

struct alignas(128) NdtRecord {
    double mu[3];
    double info[9];
    unsigned long long key;
    double pad[3];
};
static_assert(sizeof(NdtRecord) == 128, "one cache line per voxel");
struct alignas(16) NdtSlot {
    unsigned long long key;  // kNdtEmpty when free
    unsigned int vid;        // index of the voxel's record
    unsigned int pad;
};

struct NdtTable {
    NdtSlot* d_slots = nullptr;  // [cap]
    NdtRecord* d_rec = nullptr;  // [n_vox (allocated: runs)], dense
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

// Slot of a voxel in the DIRECT table (the incremental one keeps ndt_hash): the two halves of the packed key folded with one
// multiply, then the 32-bit murmur finaliser — the accumulate kernel hashes seven voxels per point, and the 64-bit finaliser seven
// times over was a good part of its integer instructions. On the bench map's 614 k voxels it collides no more often than the 64-bit
// one (11 k keys share a first slot at load 0.02; x·A ^ y·B ^ z·C, the classic spatial hash, ten times as many: walls and ground are
// lattices of keys, and its products cancel on them).
__host__ __device__ inline size_t ndt_hash32(unsigned long long key, size_t cap_mask) {
    uint32_t h = (uint32_t)key + (uint32_t)(key >> 32) * 0x9E3779B1u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return (size_t)h & cap_mask;
}

// Build (SetDirectNdtTargetCloud, ndt_registration.cpp:87-148). Returns hipError; *bad_key is set when a point falls
// outside the ±2^20-voxel key range.
hipError_t ndt_build(NdtTable& t, const float4* d_pts, size_t n, double voxel_size, int min_pts_in_voxel, hipStream_t s, bool* bad_key);
void ndt_table_free(NdtTable& t);
// Test read-back: up to out_cap voxels as dense arrays (keys n×3 int32, mu n×3, info n×9), in table order.
hipError_t ndt_dump(const NdtTable& t, int* keys, double* mu, double* info, size_t out_cap, hipStream_t s);

// K5: per-point 7-voxel probe + χ² gate + un-weighted JᵀJ / Jᵀe sums (AlignNdt inner loop, ndt cpp:399-433).
// active / n_active (optional): the scans to launch (SearchArgs::active); split_scans > 0: split the partial sums as a plain batch of
// that many scans would (AccumArgs::split_scans).
int launch_ndt_accum(const NdtTable* t, const float4* src, const int* counts, const PoseState* st, int max_n, int n_scans, double* partials,
                     hipStream_t s, const int* active = nullptr, int n_active = 0, int split_scans = 0, const int* src_of = nullptr);

}  // namespace locgpu
