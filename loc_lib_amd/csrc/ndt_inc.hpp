// loc_lib_amd/csrc/ndt_inc.hpp — incremental NDT state and launchers (see ndt_inc.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "device_math.hpp"

namespace locgpu {

struct IncNdtState;

IncNdtState* inc_ndt_create(size_t capacity, double voxel_size);
void inc_ndt_destroy(IncNdtState* st);
size_t inc_ndt_num_voxels(const IncNdtState* st);
hipError_t inc_ndt_ingest(IncNdtState& st, const float4* host_pts, const float4* d_pts, size_t n, hipStream_t s, bool* bad_key);
void launch_inc_accum(const IncNdtState* st, double res_th, int n_nearby, const float4* src, const int* counts, const PoseState* ps, int max_n,
                      int n_scans, double* partials, hipStream_t s, const int* active = nullptr, int n_active = 0, const int* src_of = nullptr);
size_t inc_ndt_dump(const IncNdtState* st, int32_t* keys, double* mu, double* info, size_t cap);
const void* inc_ndt_table_ptr(const IncNdtState* st);

}  // namespace locgpu
