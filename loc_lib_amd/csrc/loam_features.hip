// loc_lib_amd/csrc/loam_features.hip — LOAM edge / surface feature picker on the GPU.
//
// Replaces LoamFeatureExtract::Extract + ExtractFromSector
// (LocUtils/src/model/feature_extract/loam_feature_extract.cpp:19-91, :93-151), which Lio::AddCloud(FullCloudPtr) runs on every
// scan before the LOAM matcher (lio.cpp:323). Same results, quirks included:
//   * points are bucketed per ring in input order (:27-36); rings with fewer than 131 points are skipped (:40-43);
//   * curvature = squared norm of (sum of the 10 ring neighbours − 10·p), the sums in float32 left to right (:47-69);
//   * six sectors per ring, each WITHOUT its last element (end iterator = begin + sector_end, :73-85);
//   * per sector: sort by curvature, walk from the largest: stop at value ≤ 0.1, at most 20 edges, the 21st pick is marked
//     but emitted nowhere, ±5 ring neighbours are marked while consecutive gaps² ≤ 0.05 (:100-139); every unmarked point of
//     the sector becomes a surface point in ascending curvature (:143-149);
//   * outputs are appended ring by ring, sector by sector.
// The reference's std::sort leaves the order of EQUAL curvatures open; here ties are ordered by ascending ring index.
//
// Mapping: one 256-thread workgroup per (sector, ring): bitonic sort of the sector's (curvature, id) pairs in LDS, the
// inherently sequential pick loop on one lane (≤ 21 picks), ordered compaction of the surface points by ballot/popcount.
#include "device_prims.hpp"

#include <cstring>
#include <string>
#include <vector>

#include "cloud_filters.hpp"
#include "context.hpp"

namespace locgpu {

namespace {

constexpr int kLB = 256;
constexpr int kMaxSector = 2048;   // longest sector (ring length / 6) the LDS sort holds
constexpr int kMaxEdges = 20;

struct LoamParams {
    uint32_t n_edge, n_surf;
    int32_t too_long;  // a sector exceeded kMaxSector
};

__global__ __launch_bounds__(kLB) void ring_key_kernel(const unsigned char* __restrict__ ring, size_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const size_t i = (size_t)blockIdx.x * kLB + threadIdx.x;
    if (i >= n) return;
    keys[i] = ring[i];
    vals[i] = (uint32_t)i;
}

// start[r] = first sorted position with key ≥ r, r = 0..num_scan
__global__ void ring_start_kernel(const uint32_t* __restrict__ keys, uint32_t n, int num_scan, uint32_t* __restrict__ start) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > num_scan) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < (uint32_t)r) lo = mid + 1;
        else hi = mid;
    }
    start[r] = lo;
}

__global__ __launch_bounds__(kLB) void ring_gather_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ vals, size_t n, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kLB + threadIdx.x;
    if (i >= n) return;
    out[i] = pts[vals[i]];
}

// curvature of ring-local index j ∈ [5, size−5) (loam_feature_extract.cpp:47-69); L is the ring-ordered cloud.
__global__ __launch_bounds__(kLB) void curvature_kernel(const float4* __restrict__ L, const uint32_t* __restrict__ ring_start, int num_scan,
                                                        double* __restrict__ curv) {
    const int r = blockIdx.y;
    const uint32_t base = ring_start[r], size = ring_start[r + 1] - base;
    if (size < 131) return;
    const uint32_t j = blockIdx.x * kLB + threadIdx.x + 5;
    if (j + 5 >= size) return;
    const float4* P = L + base + j;
    const float fx = P[-5].x + P[-4].x + P[-3].x + P[-2].x + P[-1].x - 10 * P[0].x + P[1].x + P[2].x + P[3].x + P[4].x + P[5].x;
    const float fy = P[-5].y + P[-4].y + P[-3].y + P[-2].y + P[-1].y - 10 * P[0].y + P[1].y + P[2].y + P[3].y + P[4].y + P[5].y;
    const float fz = P[-5].z + P[-4].z + P[-3].z + P[-2].z + P[-1].z - 10 * P[0].z + P[1].z + P[2].z + P[3].z + P[4].z + P[5].z;
    const double dx = fx, dy = fy, dz = fz;
    curv[base + j] = dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ bool gap_too_large(const float4& a, const float4& b) {
    const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;  // float differences, widened (:123-126)
    return dx * dx + dy * dy + dz * dz > 0.05;
}

// One workgroup per (sector, ring). Writes the sector's edges to edge_slot[task·20 …] and its surface points to
// surf_slot[ring_base + 5 + sector_start …] (sectors do not overlap there), plus the two counts.
__global__ __launch_bounds__(kLB) void sector_kernel(const float4* __restrict__ L, const double* __restrict__ curv, const uint32_t* __restrict__ ring_start,
                                                     float4* __restrict__ edge_slot, float4* __restrict__ surf_slot, uint32_t* __restrict__ edge_cnt,
                                                     uint32_t* __restrict__ surf_cnt, LoamParams* P) {
    __shared__ double s_val[kMaxSector];
    __shared__ int s_id[kMaxSector];
    __shared__ unsigned char s_picked[kMaxSector + 16];
    __shared__ int s_edges[kMaxEdges];
    __shared__ int s_n_edge;
    __shared__ uint32_t s_wave[kLB / 64];
    const int sec = blockIdx.x, r = blockIdx.y, task = r * 6 + sec, tid = threadIdx.x;
    const uint32_t base = ring_start[r], size = ring_start[r + 1] - base;
    if (size < 131) {
        if (tid == 0) { edge_cnt[task] = 0; surf_cnt[task] = 0; }
        return;
    }
    const int total = (int)size - 10;
    const int len = total / 6;
    const int s_start = len * sec;
    const int s_end = sec == 5 ? total - 1 : len * (sec + 1) - 1;
    const int m = s_end - s_start;  // the sub-vector excludes element `sector_end`
    if (m > kMaxSector) {
        if (tid == 0) { edge_cnt[task] = 0; surf_cnt[task] = 0; P->too_long = 1; }
        return;
    }
    if (m <= 0) {
        if (tid == 0) { edge_cnt[task] = 0; surf_cnt[task] = 0; }
        return;
    }
    int pow2 = 1;
    while (pow2 < m) pow2 <<= 1;
    for (int t = tid; t < pow2; t += kLB) {
        const int id = 5 + s_start + t;  // cloud_curvature[k].id_ = k + 5
        s_val[t] = t < m ? curv[base + id] : __builtin_inf();
        s_id[t] = t < m ? id : 0x7FFFFFFF;
    }
    for (int t = tid; t < m + 16; t += kLB) s_picked[t] = 0;
    __syncthreads();
    // bitonic sort ascending by (value, id)
    for (int k = 2; k <= pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < pow2; t += kLB) {
                const int x = t ^ j;
                if (x > t) {
                    const double va = s_val[t], vb = s_val[x];
                    const int ia = s_id[t], ib = s_id[x];
                    const bool a_gt_b = va > vb || (va == vb && ia > ib);
                    const bool up = (t & k) == 0;
                    if (a_gt_b == up) { s_val[t] = vb; s_val[x] = va; s_id[t] = ib; s_id[x] = ia; }
                }
            }
            __syncthreads();
        }
    }
    // the pick loop (:100-139), sequential by definition; picked flags are indexed by id − s_start (ids reach 5 beyond either end)
    if (tid == 0) {
        int n_picked = 0, n_edge = 0;
        const float4* R = L + base;
        for (int i = m - 1; i >= 0; --i) {
            const int ind = s_id[i];
            if (s_picked[ind - s_start]) continue;
            if (s_val[i] <= 0.1) break;
            n_picked++;
            s_picked[ind - s_start] = 1;
            if (n_picked <= kMaxEdges) s_edges[n_edge++] = ind;
            else break;
            for (int k = 1; k <= 5; k++) {
                if (gap_too_large(R[ind + k], R[ind + k - 1])) break;
                s_picked[ind + k - s_start] = 1;
            }
            for (int k = -1; k >= -5; k--) {
                if (gap_too_large(R[ind + k], R[ind + k + 1])) break;
                s_picked[ind + k - s_start] = 1;
            }
        }
        s_n_edge = n_edge;
        edge_cnt[task] = (uint32_t)n_edge;
    }
    __syncthreads();
    for (int e = tid; e < s_n_edge; e += kLB) edge_slot[(size_t)task * kMaxEdges + e] = L[base + s_edges[e]];
    // surface points: unpicked elements in ascending sorted order (:143-149) — ordered compaction, 256 positions per round
    uint32_t running = 0;
    const int lane = tid & 63, wave = tid >> 6;
    for (int t0 = 0; t0 < m; t0 += kLB) {
        const int t = t0 + tid;
        int ind = 0;
        bool keep = false;
        if (t < m) { ind = s_id[t]; keep = !s_picked[ind - s_start]; }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < kLB / 64; ++w) { before += w < wave ? s_wave[w] : 0u; all += s_wave[w]; }
        if (keep) surf_slot[base + 5 + s_start + running + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = L[base + ind];
        running += all;
        __syncthreads();
    }
    if (tid == 0) surf_cnt[task] = running;
}

// exclusive scan of the per-task counts (≤ 256·6 tasks) by one workgroup; totals into P
__global__ __launch_bounds__(kLB) void task_scan_kernel(const uint32_t* __restrict__ edge_cnt, const uint32_t* __restrict__ surf_cnt, int n_tasks,
                                                        uint32_t* __restrict__ edge_off, uint32_t* __restrict__ surf_off, LoamParams* P) {
    if (threadIdx.x != 0) return;
    uint32_t e = 0, s = 0;
    for (int t = 0; t < n_tasks; ++t) {
        edge_off[t] = e; surf_off[t] = s;
        e += edge_cnt[t]; s += surf_cnt[t];
    }
    P->n_edge = e;
    P->n_surf = s;
}

__global__ __launch_bounds__(kLB) void sector_scatter_kernel(const float4* __restrict__ edge_slot, const float4* __restrict__ surf_slot,
                                                             const uint32_t* __restrict__ ring_start, const uint32_t* __restrict__ edge_cnt,
                                                             const uint32_t* __restrict__ surf_cnt, const uint32_t* __restrict__ edge_off,
                                                             const uint32_t* __restrict__ surf_off, float4* __restrict__ edge_out, float4* __restrict__ surf_out) {
    const int sec = blockIdx.x, r = blockIdx.y, task = r * 6 + sec;
    const uint32_t ne = edge_cnt[task], ns = surf_cnt[task];
    if (ne == 0 && ns == 0) return;
    const uint32_t base = ring_start[r], size = ring_start[r + 1] - base;
    const int len = ((int)size - 10) / 6;
    const uint32_t src = base + 5 + (uint32_t)(len * sec);
    for (uint32_t e = threadIdx.x; e < ne; e += kLB) edge_out[edge_off[task] + e] = edge_slot[(size_t)task * kMaxEdges + e];
    for (uint32_t s = threadIdx.x; s < ns; s += kLB) surf_out[surf_off[task] + s] = surf_slot[src + s];
}

struct LoamScratch {
    size_t cap = 0;
    int tasks_cap = 0;
    unsigned char* d_ring = nullptr;
    uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr};
    float4 *ring_pts = nullptr, *surf_slot = nullptr, *edge_slot = nullptr;
    double* curv = nullptr;
    uint32_t *ring_start = nullptr, *edge_cnt = nullptr, *surf_cnt = nullptr, *edge_off = nullptr, *surf_off = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    LoamParams *d_params = nullptr, *h_params = nullptr;
};

#define LOCGPU_TRY(expr)                   \
    do {                                   \
        const hipError_t e__ = (expr);     \
        if (e__ != hipSuccess) return e__; \
    } while (0)

void free_scratch(LoamScratch* S) {
    if (!S) return;
    void* ptrs[] = {S->d_ring, S->keys[0], S->keys[1], S->vals[0], S->vals[1], S->ring_pts, S->surf_slot, S->edge_slot, S->curv, S->ring_start,
                    S->edge_cnt, S->surf_cnt, S->edge_off, S->surf_off, S->temp, S->d_params};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (S->h_params) (void)hipHostFree(S->h_params);
    delete S;
}

hipError_t ensure(locgpu_ctx* ctx, size_t n, int num_scan) {
    if (!ctx->loam) ctx->loam = new LoamScratch();
    LoamScratch* S = (LoamScratch*)ctx->loam;
    if (!S->d_params) {
        LOCGPU_TRY(hipMalloc((void**)&S->d_params, sizeof(LoamParams)));
        LOCGPU_TRY(hipHostMalloc((void**)&S->h_params, sizeof(LoamParams)));
    }
    const int tasks = num_scan * 6;
    if (tasks > S->tasks_cap) {
        void* ptrs[] = {S->ring_start, S->edge_cnt, S->surf_cnt, S->edge_off, S->surf_off, S->edge_slot};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        S->ring_start = S->edge_cnt = S->surf_cnt = S->edge_off = S->surf_off = nullptr; S->edge_slot = nullptr; S->tasks_cap = 0;
        LOCGPU_TRY(hipMalloc((void**)&S->ring_start, (num_scan + 1) * sizeof(uint32_t)));
        LOCGPU_TRY(hipMalloc((void**)&S->edge_cnt, tasks * sizeof(uint32_t)));
        LOCGPU_TRY(hipMalloc((void**)&S->surf_cnt, tasks * sizeof(uint32_t)));
        LOCGPU_TRY(hipMalloc((void**)&S->edge_off, tasks * sizeof(uint32_t)));
        LOCGPU_TRY(hipMalloc((void**)&S->surf_off, tasks * sizeof(uint32_t)));
        LOCGPU_TRY(hipMalloc((void**)&S->edge_slot, (size_t)tasks * kMaxEdges * sizeof(float4)));
        S->tasks_cap = tasks;
    }
    if (n > S->cap) {
        void* ptrs[] = {S->d_ring, S->keys[0], S->keys[1], S->vals[0], S->vals[1], S->ring_pts, S->surf_slot, S->curv, S->temp};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        S->d_ring = nullptr; S->keys[0] = S->keys[1] = S->vals[0] = S->vals[1] = nullptr; S->ring_pts = S->surf_slot = nullptr; S->curv = nullptr;
        S->temp = nullptr; S->cap = 0;
        const size_t cap = n + n / 4 + 1024;
        LOCGPU_TRY(hipMalloc((void**)&S->d_ring, cap));
        for (int j = 0; j < 2; ++j) {
            LOCGPU_TRY(hipMalloc((void**)&S->keys[j], cap * sizeof(uint32_t)));
            LOCGPU_TRY(hipMalloc((void**)&S->vals[j], cap * sizeof(uint32_t)));
        }
        LOCGPU_TRY(hipMalloc((void**)&S->ring_pts, cap * sizeof(float4)));
        LOCGPU_TRY(hipMalloc((void**)&S->surf_slot, cap * sizeof(float4)));
        LOCGPU_TRY(hipMalloc((void**)&S->curv, cap * sizeof(double)));
        size_t tb = 0;
        LOCGPU_TRY(prim::sort_pairs(nullptr, tb, S->keys[0], S->keys[1], S->vals[0], S->vals[1], (int)cap, 0, 8, ctx->stream));
        S->temp_bytes = tb + 256;
        LOCGPU_TRY(hipMalloc(&S->temp, S->temp_bytes));
        S->cap = cap;
    }
    return hipSuccess;
}

}  // namespace

void loam_free(locgpu_ctx* ctx) {
    free_scratch((LoamScratch*)ctx->loam);
    ctx->loam = nullptr;
}

// in: cloud resident in HBM; ring: host bytes, one per point. *too_long is set when a sector exceeds the LDS sort capacity.
hipError_t loam_extract_dev(locgpu_ctx* ctx, const locgpu_cloud* in, const unsigned char* ring, int num_scan, locgpu_cloud* edge, locgpu_cloud* surf,
                            bool* too_long) {
    const size_t n = in->n;
    *too_long = false;
    edge->n = 0; edge->is_dense = 1;
    surf->n = 0; surf->is_dense = 1;
    if (n == 0) return hipSuccess;
    LOCGPU_TRY(ensure(ctx, n, num_scan));
    LoamScratch* S = (LoamScratch*)ctx->loam;
    hipStream_t s = ctx->stream;
    const unsigned nb = (unsigned)((n + kLB - 1) / kLB);
    LOCGPU_TRY(hipMemcpyAsync(S->d_ring, ring, n, hipMemcpyHostToDevice, s));
    LOCGPU_TRY(hipMemsetAsync(S->d_params, 0, sizeof(LoamParams), s));
    hipLaunchKernelGGL(ring_key_kernel, dim3(nb), dim3(kLB), 0, s, S->d_ring, n, S->keys[0], S->vals[0]);
    size_t tb = S->temp_bytes;
    LOCGPU_TRY(prim::sort_pairs(S->temp, tb, S->keys[0], S->keys[1], S->vals[0], S->vals[1], (int)n, 0, 8, s));  // stable: input order per ring
    hipLaunchKernelGGL(ring_start_kernel, dim3((num_scan + 1 + 63) / 64), dim3(64), 0, s, S->keys[1], (uint32_t)n, num_scan, S->ring_start);
    hipLaunchKernelGGL(ring_gather_kernel, dim3(nb), dim3(kLB), 0, s, in->d, S->vals[1], n, S->ring_pts);
    hipLaunchKernelGGL(curvature_kernel, dim3(nb, num_scan), dim3(kLB), 0, s, S->ring_pts, S->ring_start, num_scan, S->curv);
    hipLaunchKernelGGL(sector_kernel, dim3(6, num_scan), dim3(kLB), 0, s, S->ring_pts, S->curv, S->ring_start, S->edge_slot, S->surf_slot, S->edge_cnt,
                       S->surf_cnt, S->d_params);
    hipLaunchKernelGGL(task_scan_kernel, dim3(1), dim3(kLB), 0, s, S->edge_cnt, S->surf_cnt, num_scan * 6, S->edge_off, S->surf_off, S->d_params);
    LOCGPU_TRY(hipGetLastError());
    LOCGPU_TRY(hipMemcpyAsync(S->h_params, S->d_params, sizeof(LoamParams), hipMemcpyDeviceToHost, s));
    LOCGPU_TRY(hipStreamSynchronize(s));
    if (S->h_params->too_long) { *too_long = true; return hipSuccess; }
    const uint32_t ne = S->h_params->n_edge, ns = S->h_params->n_surf;
    LOCGPU_TRY(cloud_reserve(edge, ne, false));
    LOCGPU_TRY(cloud_reserve(surf, ns, false));
    hipLaunchKernelGGL(sector_scatter_kernel, dim3(6, num_scan), dim3(kLB), 0, s, S->edge_slot, S->surf_slot, S->ring_start, S->edge_cnt, S->surf_cnt,
                       S->edge_off, S->surf_off, edge->d, surf->d);
    LOCGPU_TRY(hipGetLastError());
    LOCGPU_TRY(hipStreamSynchronize(s));
    edge->n = ne;
    surf->n = ns;
    return hipSuccess;
}

}  // namespace locgpu
