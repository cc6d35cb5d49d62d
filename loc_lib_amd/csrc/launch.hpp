// loc_lib_amd/csrc/launch.hpp — host-callable launchers of the kernels in icp_kernels.hip / ndt_kernels.hip.
#pragma once
#include "gn_post.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hpp"

namespace locgpu {

struct GnParams;

struct SearchArgs {
    const uint2* tree;
    size_t tree_bytes;
    int depth;
    const float4* src;
    const int* counts;
    const PoseState* st;
    uint32_t* nn;
    size_t nn_pitch;
    int max_n, n_scans, k;
    float alpha_eff;
    int skip_nonfinite;
    unsigned long long* visit_totals;  // null unless visit counting is on
    uint32_t* redo_list;               // [n_scans*max_n] queries the fast kernel hands to the exact kernel
    unsigned int* redo_count;
    uint32_t* redo_list2;              // grid mode: first pass → second pass
    unsigned int* redo_count2;
    unsigned long long* search_stats;  // [2] cumulative: queries searched, queries redone exactly (may be null)
    // Later chunks of an alignment: the indices of the scans still open (device array) and their number. The hot search kernel and
    // the accumulate kernels then run grid.y = n_active instead of n_scans blocks of early exits (a 256-scan step's last eight
    // iterations hold a handful of scans: 460 k empty workgroups cost ≈96 µs per launch). nullptr = every scan.
    const int* active = nullptr;
    int n_active = 0;
    // Scan pools: the points of scan slot s are region src_of[s] of `src` (an arena of max_n-point regions); nullptr = region s.
    const int* src_of = nullptr;
    // instrumented pass only: bitmap over the tree's 8-byte slots (one bit each, zeroed by the caller, counted and cleared again
    // by launch_count_touched) — which slots this search launch reads at all
    uint32_t* touched = nullptr;
};

struct AccumArgs {
    const uint2* tree;
    const float4* src;
    const int* counts;
    const PoseState* st;
    const uint32_t* nn;
    size_t nn_pitch;
    int max_n, n_scans;
    double gate;        // max_plane / max_line / max_nn distance
    double* partials;   // [n_scans][blocks_per_scan][kAccW]
    const int* active = nullptr;  // see SearchArgs
    int n_active = 0;
    const int* src_of = nullptr;  // see SearchArgs
    // > 0: split the sums as a plain batch of this many scans would (a scan pool of many slots serving jobs of that size), so that
    // a pooled scan's partial sums — hence its pose — are the plain batch's bit for bit. 0: by n_scans.
    int split_scans = 0;
};

bool launch_icp_search(const SearchArgs& a, hipStream_t s);
// exact tree traversal over a.redo_list only (the list is filled by a preceding fast / grid kernel)
bool launch_icp_search_redo(const SearchArgs& a, hipStream_t s);
// the walk traversal with every level stored (a.alpha_eff) over the queries in `list`, then the exact redo kernel for its ties
bool launch_icp_search_list(const SearchArgs& a, const uint32_t* list, const unsigned int* n_list, hipStream_t s);
bool launch_knn_query(const uint2* tree, int depth, const float* q, size_t nq, int k, float alpha_eff, int32_t* out, uint32_t* visits,
                      hipStream_t s);
// returns the number of partial blocks per scan the kernel wrote (what gn_solve must sum)
int launch_icp_accum(int method, const AccumArgs& a, hipStream_t s);
// list_counts (optional): the two search work-list counters, zeroed for the next iteration
// scans (optional): n_scans indices — block i solves scan scans[i] instead of scan i
// post (optional, one scan): the solve also writes an iteration word — and, once the scan is done, its result — to pinned host memory
void launch_gn_solve(const double* partials, int blocks_per_scan, PoseState* st, int n_scans, const GnParams& prm, int do_update, double* hb_out,
                     unsigned int* list_counts, hipStream_t s, const int* scans = nullptr, const GnPost* post = nullptr);
int icp_accum_split(int method, int max_n, int n_scans);  // points per thread of the accumulate kernels (the split of the partial sums)
// Sharded batches: acc[g][0..28) = sum of the block partials of global scan g when this rank holds it (local index g - first),
// zeros otherwise — in exactly the order gn_solve_kernel sums them, so that all-reduce(acc) followed by gn_solve on acc
// (blocks_per_scan = 1) gives bit for bit the single-GPU result.
// owned (optional, scan pools): per global scan, whether this rank holds its points (then first = 0, n_local = n_total)
void launch_sum_partials(const double* partials, int blocks_per_scan, const PoseState* st_all, int first, int n_local, int n_total, double* acc,
                         hipStream_t s, const unsigned char* owned = nullptr);
struct M12f { float v[12]; };  // rows of pose.matrix().cast<float>() (icp_registration.cpp:241), a kernel argument
void launch_transform_cloud(const float4* src, size_t n, const M12f& m12, float* dst_xyz, hipStream_t s);
// Code-object self-test (once per process): no walk kernel owns static LDS, so every traversal stack starts at LDS address 0 —
// the precondition of search_walk.hpp's out-of-range rows (tests/test_gpu_lds_semantics.py pins the hardware side).
bool search_kernels_lds_ok();
// instrumented pass: totals[3] += number of set bits of touched[0..n_words), then touched := 0
void launch_count_touched(uint32_t* touched, size_t n_words, unsigned long long* totals, hipStream_t s);
// test hook: slot lists [k][pitch] → original point indices out[query * k + j] (-1 = none)
void launch_nn_to_index(const uint2* tree, const uint32_t* nn, size_t nn_pitch, size_t n_queries, int k, int32_t* out, hipStream_t s);

}  // namespace locgpu
