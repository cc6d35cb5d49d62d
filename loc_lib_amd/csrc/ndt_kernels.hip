// loc_lib_amd/csrc/ndt_kernels.hip — direct NDT on gfx950: voxel-table build (K4) and per-iteration accumulation (K5).
//
// K4 follows NdtRegistration::SetDirectNdtTargetCloud (ndt_registration.cpp:87-148): key = trunc-toward-zero of
// pt/voxel (`(pt * inv_voxel_size_).cast<int>()`, :100), per voxel with more than min_pts_in_voxel points the mean and
// the (n−1)-normalised covariance (math_utils.h:55-72), SVD, λ1,λ2 clamped to ≥ 1e-3·λ0, info = V·diag(1/λ)·Uᵀ (:118-130).
// Sums are FP64 atomics, so μ/Σ differ from the reference's index-ordered sums by summation order only (~1e-16 rel.).
// K5 follows AlignNdt's inner loop (:399-433): 7 probes in the reference's offset order (:57-58), χ² gate with info,
// sums NOT weighted by info, effective_num once per source point.
#include "icp_kernels.hpp"
#include "ndt_kernels.hpp"

namespace locgpu {

__device__ __forceinline__ void ndt_key_of(const D3& p, double inv, int& kx, int& ky, int& kz) {
    kx = (int)(p.x * inv); ky = (int)(p.y * inv); kz = (int)(p.z * inv);  // C++ double→int: truncation toward zero
}

__global__ __launch_bounds__(kBlock) void ndt_insert_kernel(const float4* __restrict__ pts, size_t n, double inv, unsigned long long* keys,
                                                            int* counts, size_t cap_mask, int* pt_slot, int* bad) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    int kx, ky, kz;
    ndt_key_of(D3{(double)p.x, (double)p.y, (double)p.z}, inv, kx, ky, kz);
    if (!ndt_key_in_range(kx, ky, kz)) { atomicExch(bad, 1); pt_slot[i] = -1; return; }
    const unsigned long long key = ndt_pack(kx, ky, kz);
    size_t h = ndt_hash(key, cap_mask);
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[h], kNdtEmpty, key);
        if (prev == kNdtEmpty || prev == key) break;
        h = (h + 1) & cap_mask;
    }
    atomicAdd(&counts[h], 1);
    pt_slot[i] = (int)h;
}

__global__ __launch_bounds__(kBlock) void ndt_assign_kernel(const unsigned long long* keys, const int* counts, size_t cap, int min_pts, int* vid,
                                                            int* n_vox) {
    const size_t h = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (h >= cap) return;
    int id = -1;
    if (keys[h] != kNdtEmpty && counts[h] > min_pts) id = atomicAdd(n_vox, 1);
    vid[h] = id;
}

__global__ __launch_bounds__(kBlock) void ndt_sum_kernel(const float4* __restrict__ pts, size_t n, const int* pt_slot, const int* vid, double* sums) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int h = pt_slot[i];
    if (h < 0) return;
    const int v = vid[h];
    if (v < 0) return;
    const float4 p = pts[i];
    atomicAdd(&sums[3 * (size_t)v + 0], (double)p.x);
    atomicAdd(&sums[3 * (size_t)v + 1], (double)p.y);
    atomicAdd(&sums[3 * (size_t)v + 2], (double)p.z);
}

__global__ __launch_bounds__(kBlock) void ndt_mean_kernel(const unsigned long long* keys, const int* counts, const int* vid, size_t cap, const double* sums,
                                                          double* mu, int* vox_key, int* vox_cnt) {
    const size_t h = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (h >= cap) return;
    const int v = vid[h];
    if (v < 0) return;
    const double len = (double)counts[h];
    for (int c = 0; c < 3; ++c) mu[3 * (size_t)v + c] = sums[3 * (size_t)v + c] / len;
    const unsigned long long k = keys[h];
    vox_key[3 * (size_t)v + 0] = (int)((k >> 42) & 0x1FFFFF) - kNdtBias;
    vox_key[3 * (size_t)v + 1] = (int)((k >> 21) & 0x1FFFFF) - kNdtBias;
    vox_key[3 * (size_t)v + 2] = (int)(k & 0x1FFFFF) - kNdtBias;
    vox_cnt[v] = counts[h];
}

__global__ __launch_bounds__(kBlock) void ndt_cov_kernel(const float4* __restrict__ pts, size_t n, const int* pt_slot, const int* vid, const double* mu,
                                                         double* cov6) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int h = pt_slot[i];
    if (h < 0) return;
    const int v = vid[h];
    if (v < 0) return;
    const float4 p = pts[i];
    const double dx = (double)p.x - mu[3 * (size_t)v], dy = (double)p.y - mu[3 * (size_t)v + 1], dz = (double)p.z - mu[3 * (size_t)v + 2];
    double* c = cov6 + 6 * (size_t)v;
    atomicAdd(&c[0], dx * dx); atomicAdd(&c[1], dx * dy); atomicAdd(&c[2], dx * dz);
    atomicAdd(&c[3], dy * dy); atomicAdd(&c[4], dy * dz); atomicAdd(&c[5], dz * dz);
}

// Per voxel: Σ/(n−1), one-sided Jacobi SVD, clamp, info = V diag(1/λ) Uᵀ.
__global__ __launch_bounds__(kBlock) void ndt_info_kernel(const double* cov6, const int* vox_cnt, size_t n_vox, double* info) {
    const size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= n_vox) return;
    const double len1 = (double)(vox_cnt[v] - 1);
    const double* c = cov6 + 6 * v;
    const double sxx = c[0] / len1, sxy = c[1] / len1, sxz = c[2] / len1, syy = c[3] / len1, syz = c[4] / len1, szz = c[5] / len1;
    double a[3][3] = {{sxx, sxy, sxz}, {sxy, syy, syz}, {sxz, syz, szz}};  // columns of the symmetric Σ
    double vv[3][3];
    jacobi_svd_onesided<3, 3>(a, vv);
    double sv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) sv[k] = sqrt(a[k][0] * a[k][0] + a[k][1] * a[k][1] + a[k][2] * a[k][2]);
    // order by descending singular value (stable, like std::sort on three keys with > comparator)
    int o0 = 0, o1 = 1, o2 = 2;
    if (sv[o1] > sv[o0]) { const int t = o0; o0 = o1; o1 = t; }
    if (sv[o2] > sv[o1]) { const int t = o1; o1 = o2; o2 = t; }
    if (sv[o1] > sv[o0]) { const int t = o0; o0 = o1; o1 = t; }
    const int ord[3] = {o0, o1, o2};
    double lam[3], U[3][3], V[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int cidx = ord[k];
        lam[k] = cidx == 0 ? sv[0] : (cidx == 1 ? sv[1] : sv[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double vr = cidx == 0 ? vv[0][r] : (cidx == 1 ? vv[1][r] : vv[2][r]);
            const double ar = cidx == 0 ? a[0][r] : (cidx == 1 ? a[1][r] : a[2][r]);
            V[k][r] = vr;
            U[k][r] = lam[k] > 0.0 ? ar / lam[k] : vr;
        }
    }
    if (lam[1] < lam[0] * 1e-3) lam[1] = lam[0] * 1e-3;
    if (lam[2] < lam[0] * 1e-3) lam[2] = lam[0] * 1e-3;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += V[k][r] * (1.0 / lam[k]) * U[k][cc];
            info[9 * v + 3 * r + cc] = s;
        }
}

__global__ void fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long val) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = val;
}

void ndt_table_free(NdtTable& t) {
    if (t.d_keys) (void)hipFree(t.d_keys);
    if (t.d_vid) (void)hipFree(t.d_vid);
    if (t.d_mu) (void)hipFree(t.d_mu);
    if (t.d_info) (void)hipFree(t.d_info);
    if (t.d_vox_key) (void)hipFree(t.d_vox_key);
    t = NdtTable();
}

#define NDT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

hipError_t ndt_build(NdtTable& t, const float4* d_pts, size_t n, double voxel_size, int min_pts_in_voxel, hipStream_t s, bool* bad_key) {
    ndt_table_free(t);
    *bad_key = false;
    t.inv_voxel = 1.0 / voxel_size;  // the reference's constructors recompute it (ndt cpp:15,25)
    size_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    int *d_counts = nullptr, *d_pt_slot = nullptr, *d_scalar = nullptr, *d_vox_cnt = nullptr;
    double *d_sums = nullptr, *d_cov = nullptr;
    auto cleanup = [&]() {
        if (d_counts) (void)hipFree(d_counts);
        if (d_pt_slot) (void)hipFree(d_pt_slot);
        if (d_scalar) (void)hipFree(d_scalar);
        if (d_vox_cnt) (void)hipFree(d_vox_cnt);
        if (d_sums) (void)hipFree(d_sums);
        if (d_cov) (void)hipFree(d_cov);
    };
    NDT_TRY(hipMalloc((void**)&t.d_keys, cap * sizeof(unsigned long long)));
    NDT_TRY(hipMalloc((void**)&t.d_vid, cap * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_counts, cap * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_pt_slot, n * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_scalar, 2 * sizeof(int)));
    NDT_TRY(hipMemsetAsync(d_counts, 0, cap * sizeof(int), s));
    NDT_TRY(hipMemsetAsync(d_scalar, 0, 2 * sizeof(int), s));
    const unsigned gcap = (unsigned)((cap + kBlock - 1) / kBlock), gn = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(fill_u64_kernel, dim3(gcap), dim3(kBlock), 0, s, t.d_keys, cap, kNdtEmpty);
    hipLaunchKernelGGL(ndt_insert_kernel, dim3(gn), dim3(kBlock), 0, s, d_pts, n, t.inv_voxel, t.d_keys, d_counts, cap - 1, d_pt_slot, d_scalar + 1);
    hipLaunchKernelGGL(ndt_assign_kernel, dim3(gcap), dim3(kBlock), 0, s, t.d_keys, d_counts, cap, min_pts_in_voxel, t.d_vid, d_scalar);
    int h_scalar[2] = {0, 0};
    NDT_TRY(hipMemcpyAsync(h_scalar, d_scalar, sizeof(h_scalar), hipMemcpyDeviceToHost, s));
    NDT_TRY(hipStreamSynchronize(s));
    *bad_key = h_scalar[1] != 0;
    t.cap = cap;
    t.n_vox = (size_t)h_scalar[0];
    const size_t nv = t.n_vox ? t.n_vox : 1;
    NDT_TRY(hipMalloc((void**)&t.d_mu, nv * 3 * sizeof(double)));
    NDT_TRY(hipMalloc((void**)&t.d_info, nv * 9 * sizeof(double)));
    NDT_TRY(hipMalloc((void**)&t.d_vox_key, nv * 3 * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_vox_cnt, nv * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_sums, nv * 3 * sizeof(double)));
    NDT_TRY(hipMalloc((void**)&d_cov, nv * 6 * sizeof(double)));
    NDT_TRY(hipMemsetAsync(d_sums, 0, nv * 3 * sizeof(double), s));
    NDT_TRY(hipMemsetAsync(d_cov, 0, nv * 6 * sizeof(double), s));
    if (t.n_vox) {
        hipLaunchKernelGGL(ndt_sum_kernel, dim3(gn), dim3(kBlock), 0, s, d_pts, n, d_pt_slot, t.d_vid, d_sums);
        hipLaunchKernelGGL(ndt_mean_kernel, dim3(gcap), dim3(kBlock), 0, s, t.d_keys, d_counts, t.d_vid, cap, d_sums, t.d_mu, t.d_vox_key, d_vox_cnt);
        hipLaunchKernelGGL(ndt_cov_kernel, dim3(gn), dim3(kBlock), 0, s, d_pts, n, d_pt_slot, t.d_vid, t.d_mu, d_cov);
        hipLaunchKernelGGL(ndt_info_kernel, dim3((unsigned)((t.n_vox + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_cov, d_vox_cnt, t.n_vox, t.d_info);
    }
    NDT_TRY(hipGetLastError());
    NDT_TRY(hipStreamSynchronize(s));
    cleanup();
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// K5. Grid (ceil(max_n/256), n_scans).
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void ndt_accum_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ vid,
                                                           const double* __restrict__ mu, const double* __restrict__ info, size_t cap_mask,
                                                           double inv_voxel, double res_th, int n_nearby, const float4* __restrict__ src,
                                                           const int* __restrict__ counts, const PoseState* __restrict__ st, int max_n,
                                                           double* __restrict__ partials, int pts, const int* __restrict__ active,
                                                           const int* __restrict__ src_of) {
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (st[scan].done) return;
    double acc[28];
#pragma unroll
    for (int v = 0; v < 28; ++v) acc[v] = 0.0;
#pragma unroll 1
    for (int pp = 0; pp < pts; ++pp) {  // several points per thread before the 28-value block reduction (≈ as costly as a point)
    const int i = (blockIdx.x * pts + pp) * kBlock + threadIdx.x;
    if (i < counts[scan]) {
        const float4 p = src[(size_t)(src_of ? src_of[scan] : scan) * max_n + i];
        const D3 q{(double)p.x, (double)p.y, (double)p.z};
        const D3 qs = se3_apply(st[scan].q, st[scan].t, q);
        int kx, ky, kz;
        ndt_key_of(qs, inv_voxel, kx, ky, kz);
        // nearby_grids_ order (ndt cpp:57-58): (0,0,0) (-1,0,0) (1,0,0) (0,1,0) (0,-1,0) (0,0,-1) (0,0,1)
        const int ox[7] = {0, -1, 1, 0, 0, 0, 0}, oy[7] = {0, 0, 0, 1, -1, 0, 0}, oz[7] = {0, 0, 0, 0, 0, -1, 1};
        double n_acc = 0.0;
        D3 esum{0.0, 0.0, 0.0};
        // The seven voxels of a point are independent look-ups, each a chain of dependent gathers (hash slot → voxel index → μ, info).
        // Written as a loop with `continue`, they ran one after the other — 21 memory latencies per point at four waves per SIMD.
        // Here every level is issued for all seven before the next one is needed (first probe of the open-addressing table for all,
        // then the few collisions one by one, then the indices, then the records), and the acceptance test is a select; the sums
        // are formed in the reference's order j = 0..6 from the same numbers, so the result is the same bits.
        unsigned long long key[7], kk[7];
        size_t hs[7];
        bool found[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int x = kx + ox[j], y = ky + oy[j], z = kz + oz[j];
            found[j] = j < n_nearby && ndt_key_in_range(x, y, z);
            key[j] = ndt_pack(found[j] ? x : kx, found[j] ? y : ky, found[j] ? z : kz);
            hs[j] = ndt_hash(key[j], cap_mask);
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) kk[j] = keys[hs[j]];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (found[j] && kk[j] != key[j] && kk[j] != kNdtEmpty) {  // collision on the first probe: walk on (load factor ≤ 0.5)
                size_t h = (hs[j] + 1) & cap_mask;
                for (;;) {
                    const unsigned long long k2 = keys[h];
                    if (k2 == key[j]) { kk[j] = k2; hs[j] = h; break; }
                    if (k2 == kNdtEmpty) { kk[j] = k2; break; }
                    h = (h + 1) & cap_mask;
                }
            }
            found[j] = found[j] && kk[j] == key[j];
        }
        int vx[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) vx[j] = vid[found[j] ? hs[j] : 0];
#pragma unroll
        for (int j = 0; j < 7; ++j) found[j] = found[j] && vx[j] >= 0;  // −1: a voxel that was dropped for having too few points (ndt cpp:136-142)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int v = found[j] ? vx[j] : 0;  // a voxel that is not there reads record 0 and is not accepted
            const double* m = mu + 3 * (size_t)v;
            const double* I = info + 9 * (size_t)v;
            const D3 e{qs.x - m[0], qs.y - m[1], qs.z - m[2]};
            const double ev[3] = {e.x, e.y, e.z};
            double res = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) res += ev[r] * I[3 * r + c] * ev[c];
            const bool accept = found[j] && !(isnan(res) || res > res_th);
            n_acc = accept ? n_acc + 1.0 : n_acc;
            esum.x = accept ? esum.x + e.x : esum.x;
            esum.y = accept ? esum.y + e.y : esum.y;
            esum.z = accept ? esum.z + e.z : esum.z;
        }
        acc[27] += 1.0;  // effective_num++ once per source point (ndt cpp:432)
        if (n_acc > 0.0) {
            double Rh[3][3];  // R·hat(q); the zeros of hat() drop out exactly
            const double* R = st[scan].R;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                Rh[r][0] = R[3 * r + 1] * q.z - R[3 * r + 2] * q.y;
                Rh[r][1] = R[3 * r + 2] * q.x - R[3 * r + 0] * q.z;
                Rh[r][2] = R[3 * r + 0] * q.y - R[3 * r + 1] * q.x;
            }
            // J = [A | I3] with A = −R·hat(q) (ndt cpp:413-416). The reference forms JᵀJ and Jᵀe entry by entry; with the identity block
            // written out, ((J0a·J0b) + J1a·J1b) + J2a·J2b is A's column product for a, b < 3, the element A[b−3][a] for a < 3 ≤ b (the
            // two products with 0.0 add nothing), 1 on the rest of the diagonal and 0 elsewhere — the same sums bit for bit (up to the
            // sign of a zero), a third of the FP64 instructions: x·0.0 is not something the compiler may drop on its own.
            double A[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) A[r][c] = -Rh[r][c];
            const double ev[3] = {esum.x, esum.y, esum.z};
            int o = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) {
                    if (a < 3 && b < 3) {
                        double s = A[0][a] * A[0][b];
                        s += A[1][a] * A[1][b];
                        s += A[2][a] * A[2][b];
                        acc[o] += n_acc * s;  // the same J for every accepted voxel of this point
                    } else if (a < 3) {
                        acc[o] += n_acc * A[b - 3][a];
                    } else if (a == b) {
                        acc[o] += n_acc;
                    }
                    ++o;
                }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double s = -A[0][a] * ev[0];
                s += -A[1][a] * ev[1];
                s += -A[2][a] * ev[2];
                acc[21 + a] += s;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[24 + c] += -ev[c];
        }
    }
    }
    // block reduce (same scheme as the ICP accumulators)
    __shared__ double s_part[kBlock / 64][kAccW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < 28; ++v) {
        const double s = wave_sum(acc[v]);
        if (lane == 0) s_part[wave][v] = s;
    }
    __syncthreads();
    if (threadIdx.x < 28) {
        double s = s_part[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) s += s_part[w][threadIdx.x];
        partials[((size_t)scan * gridDim.x + blockIdx.x) * kAccW + threadIdx.x] = s;
    }
}

int launch_ndt_accum(const NdtTable* t, const float4* src, const int* counts, const PoseState* st, int max_n, int n_scans, double* partials,
                     hipStream_t s, const int* active, int n_active, int split_scans, const int* src_of) {
    const int blocks = (max_n + kBlock - 1) / kBlock;
    const long total_blocks = (long)blocks * (split_scans > 0 ? split_scans : n_scans);  // the split — the order of the sums — never depends on `active`
    const int pts = total_blocks >= 8192 ? 8 : (total_blocks >= 4096 ? 4 : (total_blocks >= 2048 ? 2 : 1));
    dim3 grid((blocks + pts - 1) / pts, active ? n_active : n_scans);
    hipLaunchKernelGGL(ndt_accum_kernel, grid, dim3(kBlock), 0, s, t->d_keys, t->d_vid, t->d_mu, t->d_info, t->cap - 1, t->inv_voxel, t->res_outlier_th,
                       t->n_nearby, src, counts, st, max_n, partials, pts, active, src_of);
    return (int)grid.x;  // partial blocks per scan
}

}  // namespace locgpu
