// loc_lib_amd/csrc/ndt_kernels.hip — direct NDT on gfx950: voxel-table build (K4) and per-iteration accumulation (K5).
//
// K4 follows NdtRegistration::SetDirectNdtTargetCloud (ndt_registration.cpp:87-148): key = trunc-toward-zero of
// pt/voxel (`(pt * inv_voxel_size_).cast<int>()`, :100), per voxel with more than min_pts_in_voxel points the mean and
// the (n−1)-normalised covariance (math_utils.h:55-72), SVD, λ1,λ2 clamped to ≥ 1e-3·λ0, info = V·diag(1/λ)·Uᵀ (:118-130).
// Round 5: the sums are the REFERENCE's sums — keys → stable radix sort (a voxel's points consecutive, in input order) → one
// thread per voxel adding sequentially, products not fused — so μ and Σ equal the reference's bit for bit and two ingests of one
// map give the same bits (rounds 1-4 summed with FP64 atomics: 1e-10 / 1e-7 from the oracle, different last bits run to run).
// The table is open addressing over 16-byte {key, voxel index} slots plus a dense array of 128-byte records {μ, info}
// (ndt_kernels.hpp): two dependent gathers per voxel instead of three. Voxels the reference drops (:136-142) are simply not inserted.
// K5 follows AlignNdt's inner loop (:399-433): 7 probes in the reference's offset order (:57-58), χ² gate with info,
// sums NOT weighted by info, effective_num once per source point.
#include "device_prims.hpp"
#include "icp_kernels.hpp"
#include "ndt_kernels.hpp"

#ifndef LOCGPU_NDT_WAVES
#define LOCGPU_NDT_WAVES 4
#endif
namespace locgpu {

__device__ __forceinline__ void ndt_key_of(const D3& p, double inv, int& kx, int& ky, int& kz) {
    kx = (int)(p.x * inv); ky = (int)(p.y * inv); kz = (int)(p.z * inv);  // C++ double→int: truncation toward zero
}

// key of every point; a point outside the ±2^20-voxel range raises *bad (the ingest is refused)
__global__ __launch_bounds__(kBlock) void ndt_key_kernel(const float4* __restrict__ pts, size_t n, double inv, unsigned long long* __restrict__ pkey,
                                                         uint32_t* __restrict__ pidx, int* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    int kx, ky, kz;
    ndt_key_of(D3{(double)p.x, (double)p.y, (double)p.z}, inv, kx, ky, kz);
    unsigned long long key = kNdtEmpty;
    if (!ndt_key_in_range(kx, ky, kz)) *bad = 1;
    else key = ndt_pack(kx, ky, kz);
    pkey[i] = key;
    pidx[i] = (uint32_t)i;
}

// sorted keys → 1 at the first point of every run
__global__ __launch_bounds__(kBlock) void ndt_head_kernel(const unsigned long long* __restrict__ skey, size_t n, int* __restrict__ head) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}

// run u: first sorted position; ustart[runs] = n; *n_runs = runs; the points in voxel order (a voxel's points consecutive, in
// input order — the order the reference's per-voxel index list has, ndt cpp:97-103)
__global__ __launch_bounds__(kBlock) void ndt_runs_kernel(const uint32_t* __restrict__ sidx, const int* __restrict__ head, const int* __restrict__ uid, size_t n,
                                                          const float4* __restrict__ pts, uint32_t* __restrict__ ustart, float4* __restrict__ psorted, int* __restrict__ n_runs) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    psorted[i] = pts[sidx[i]];
    if (head[i]) ustart[uid[i]] = (uint32_t)i;
    if (i == n - 1) {
        const int runs = uid[i] + head[i];
        ustart[runs] = (uint32_t)n;
        *n_runs = runs;
    }
}

__global__ void ndt_clear_kernel(NdtSlot* slots, size_t cap) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) slots[i] = NdtSlot{kNdtEmpty, 0u, 0u};
}

// One thread per voxel (run of the sorted keys) with more than min_pts points: math::ComputeMeanAndCov (math_utils.h:55-72) summing
// in input order, Σ/(n−1), one-sided Jacobi SVD, clamp, info = V diag(1/λ) Uᵀ (ndt cpp:111-130) — then the record goes into the table.
__global__ __launch_bounds__(kBlock) void ndt_voxel_kernel(const unsigned long long* __restrict__ skey, const float4* __restrict__ psorted, const uint32_t* __restrict__ ustart,
                                                           const int* __restrict__ n_runs, int min_pts, NdtSlot* __restrict__ slots, NdtRecord* __restrict__ rec, size_t cap_mask,
                                                           int* __restrict__ n_vox) {
    const int u = blockIdx.x * kBlock + threadIdx.x;
    if (u >= *n_runs) return;
    const uint32_t b = ustart[u], e = ustart[u + 1];
    const uint32_t len = e - b;
    const unsigned long long key = skey[b];
    if (key == kNdtEmpty || (int)len <= min_pts) return;  // count > min_pts_in_voxel_ (ndt cpp:111,137)
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (uint32_t j = b; j < e; ++j) { const float4 p = psorted[j]; sx = sx + (double)p.x; sy = sy + (double)p.y; sz = sz + (double)p.z; }
    const double mx = sx / (double)len, my = sy / (double)len, mz = sz / (double)len;
    double c00 = 0.0, c01 = 0.0, c02 = 0.0, c11 = 0.0, c12 = 0.0, c22 = 0.0;
    for (uint32_t j = b; j < e; ++j) {
        const float4 p = psorted[j];
        const double dx = (double)p.x - mx, dy = (double)p.y - my, dz = (double)p.z - mz;
        c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;  // -ffp-contract=off: product, then sum
    }
    const double len1 = (double)(len - 1);
    const double sxx = c00 / len1, sxy = c01 / len1, sxz = c02 / len1, syy = c11 / len1, syz = c12 / len1, szz = c22 / len1;
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
    // claim a table slot (keys are distinct: a first probe that finds the slot taken walks on)
    size_t h = ndt_hash32(key, cap_mask);
    for (;;) {
        const unsigned long long prev = atomicCAS(&slots[h].key, kNdtEmpty, key);
        if (prev == kNdtEmpty) break;
        h = (h + 1) & cap_mask;
    }
    const unsigned int vid = (unsigned int)atomicAdd(n_vox, 1);  // which record a voxel gets varies from run to run; what is in it does not
    slots[h].vid = vid;
    NdtRecord& R = rec[vid];
    R.key = key;
    R.mu[0] = mx; R.mu[1] = my; R.mu[2] = mz;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += V[k][r] * (1.0 / lam[k]) * U[k][cc];
            R.info[3 * r + cc] = s;
        }
}

// Read-back for tests: the records as dense arrays.
__global__ __launch_bounds__(kBlock) void ndt_dump_kernel(const NdtRecord* __restrict__ rec, size_t n, int* __restrict__ keys, double* __restrict__ mu, double* __restrict__ info) {
    const size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= n) return;
    const NdtRecord& R = rec[o];
    keys[3 * o + 0] = (int)((R.key >> 42) & 0x1FFFFF) - kNdtBias;
    keys[3 * o + 1] = (int)((R.key >> 21) & 0x1FFFFF) - kNdtBias;
    keys[3 * o + 2] = (int)(R.key & 0x1FFFFF) - kNdtBias;
    for (int k = 0; k < 3; ++k) mu[3 * o + k] = R.mu[k];
    for (int k = 0; k < 9; ++k) info[9 * o + k] = R.info[k];
}

void ndt_table_free(NdtTable& t) {
    if (t.d_slots) (void)hipFree(t.d_slots);
    if (t.d_rec) (void)hipFree(t.d_rec);
    t = NdtTable();
}

#define NDT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

hipError_t ndt_build(NdtTable& t, const float4* d_pts, size_t n, double voxel_size, int min_pts_in_voxel, hipStream_t s, bool* bad_key) {
    ndt_table_free(t);
    *bad_key = false;
    t.inv_voxel = 1.0 / voxel_size;  // the reference's constructors recompute it (ndt cpp:15,25)
    unsigned long long *d_pkey = nullptr, *d_skey = nullptr;
    uint32_t *d_pidx = nullptr, *d_sidx = nullptr, *d_ustart = nullptr;
    int *d_head = nullptr, *d_uid = nullptr, *d_scalar = nullptr;  // scalars: [0] runs, [1] bad key, [2] voxels kept
    float4* d_psorted = nullptr;
    void* d_temp = nullptr;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_pkey, (void*)d_skey, (void*)d_pidx, (void*)d_sidx, (void*)d_ustart, (void*)d_head, (void*)d_uid, (void*)d_scalar, (void*)d_psorted, d_temp})
            if (p) (void)hipFree(p);
    };
    if (n > 0xFFFFFFF0ull) return hipErrorInvalidValue;
    NDT_TRY(hipMalloc((void**)&d_pkey, n * sizeof(unsigned long long)));
    NDT_TRY(hipMalloc((void**)&d_skey, n * sizeof(unsigned long long)));
    NDT_TRY(hipMalloc((void**)&d_pidx, n * sizeof(uint32_t)));
    NDT_TRY(hipMalloc((void**)&d_sidx, n * sizeof(uint32_t)));
    NDT_TRY(hipMalloc((void**)&d_ustart, (n + 1) * sizeof(uint32_t)));
    NDT_TRY(hipMalloc((void**)&d_head, n * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_uid, n * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_psorted, n * sizeof(float4)));
    NDT_TRY(hipMalloc((void**)&d_scalar, 4 * sizeof(int)));
    NDT_TRY(hipMemsetAsync(d_scalar, 0, 4 * sizeof(int), s));
    size_t b1 = 0, b2 = 0;
    NDT_TRY(prim::sort_pairs(nullptr, b1, d_pkey, d_skey, d_pidx, d_sidx, n, 0, 64, s));
    NDT_TRY(prim::exclusive_sum(nullptr, b2, d_head, d_uid, n, s));
    size_t tb = std::max(b1, b2);
    NDT_TRY(hipMalloc(&d_temp, std::max<size_t>(tb, 16)));
    const unsigned gn = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(ndt_key_kernel, dim3(gn), dim3(kBlock), 0, s, d_pts, n, t.inv_voxel, d_pkey, d_pidx, d_scalar + 1);
    tb = b1;
    NDT_TRY(prim::sort_pairs(d_temp, tb, d_pkey, d_skey, d_pidx, d_sidx, n, 0, 64, s));  // stable: a voxel's points keep their input order
    hipLaunchKernelGGL(ndt_head_kernel, dim3(gn), dim3(kBlock), 0, s, d_skey, n, d_head);
    tb = b2;
    NDT_TRY(prim::exclusive_sum(d_temp, tb, d_head, d_uid, n, s));
    hipLaunchKernelGGL(ndt_runs_kernel, dim3(gn), dim3(kBlock), 0, s, d_sidx, d_head, d_uid, n, d_pts, d_ustart, d_psorted, d_scalar);
    int h_scalar[4] = {0, 0, 0, 0};
    NDT_TRY(hipMemcpyAsync(h_scalar, d_scalar, sizeof(h_scalar), hipMemcpyDeviceToHost, s));
    NDT_TRY(hipStreamSynchronize(s));
    *bad_key = h_scalar[1] != 0;
    const size_t runs = (size_t)h_scalar[0];
    // A SPARSE table: at load 0.03 nearly every look-up — hit or miss — ends at its first probe, and a collision walk is what costs
    // (a wave walks as long as its unluckiest lane, seven times per point); the lines a launch touches are as many as the voxels it
    // looks up, whatever the table's size (measured: load <= 0.5, 15.7 ms per 256-scan step; load <= 0.03, see profiles/experiments.md).
    size_t cap = 1024;
    while (cap < 32 * runs && cap < ((size_t)1 << 26)) cap <<= 1;  // at most 1 GB of slots ...
    while (cap < 2 * runs) cap <<= 1;                              // ... but never above load 0.5
    NDT_TRY(hipMalloc((void**)&t.d_slots, cap * sizeof(NdtSlot)));
    NDT_TRY(hipMalloc((void**)&t.d_rec, std::max<size_t>(runs, 1) * sizeof(NdtRecord)));  // room for every run; the kept voxels fill a prefix
    t.cap = cap;
    hipLaunchKernelGGL(ndt_clear_kernel, dim3((unsigned)((cap + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, t.d_slots, cap);
    NDT_TRY(hipMemsetAsync(t.d_rec, 0, std::max<size_t>(runs, 1) * sizeof(NdtRecord), s));  // record 0 is read for voxels that are not there: finite numbers, never accepted
    if (runs > 0 && !*bad_key)
        hipLaunchKernelGGL(ndt_voxel_kernel, dim3((unsigned)((runs + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_skey, d_psorted, d_ustart, d_scalar, min_pts_in_voxel, t.d_slots,
                           t.d_rec, cap - 1, d_scalar + 2);
    NDT_TRY(hipGetLastError());
    NDT_TRY(hipMemcpyAsync(h_scalar, d_scalar, sizeof(h_scalar), hipMemcpyDeviceToHost, s));
    NDT_TRY(hipStreamSynchronize(s));
    t.n_vox = (size_t)h_scalar[2];
    cleanup();
    return hipSuccess;
}

hipError_t ndt_dump(const NdtTable& t, int* keys, double* mu, double* info, size_t out_cap, hipStream_t s) {
    const size_t n = std::min(out_cap, t.n_vox);
    if (!t.d_rec || n == 0) return hipSuccess;
    int* d_keys = nullptr;
    double *d_mu = nullptr, *d_info = nullptr;
    auto cleanup = [&]() { for (void* p : {(void*)d_keys, (void*)d_mu, (void*)d_info}) if (p) (void)hipFree(p); };
    NDT_TRY(hipMalloc((void**)&d_keys, n * 3 * sizeof(int)));
    NDT_TRY(hipMalloc((void**)&d_mu, n * 3 * sizeof(double)));
    NDT_TRY(hipMalloc((void**)&d_info, n * 9 * sizeof(double)));
    hipLaunchKernelGGL(ndt_dump_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, t.d_rec, n, d_keys, d_mu, d_info);
    if (keys) NDT_TRY(hipMemcpyAsync(keys, d_keys, n * 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (mu) NDT_TRY(hipMemcpyAsync(mu, d_mu, n * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (info) NDT_TRY(hipMemcpyAsync(info, d_info, n * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    NDT_TRY(hipStreamSynchronize(s));
    cleanup();
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// K5. Grid (ceil(max_n/256), n_scans).
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LOCGPU_NDT_WAVES, LOCGPU_NDT_WAVES))) void ndt_accum_kernel(const NdtSlot* __restrict__ slots, const NdtRecord* __restrict__ rec, size_t cap_mask,
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
        // The seven voxels of a point are independent look-ups. Written as a loop with `continue`, they ran one after the other. Here
        // the first probe of the open-addressing table is issued for all seven (one 16-byte load returns key and voxel index), then
        // the few collisions one by one (four slots to a cache line: the walk rarely leaves the line), then the seven records (one
        // 128-byte line each); the acceptance test is a select and the sums are formed in the reference's order j = 0..6 from the
        // same numbers.
        unsigned long long key[7], kk[7];
        unsigned int vx[7];
        size_t hs[7];
        bool found[7];
        // the packed key is linear in the coordinates (ndt_pack): a face neighbour's key is the centre's ± one constant, and it is in
        // range when the centre is and the one coordinate that moved still is
        const bool centre_ok = ndt_key_in_range(kx, ky, kz);
        const unsigned long long key0 = ndt_pack(centre_ok ? kx : 0, centre_ok ? ky : 0, centre_ok ? kz : 0);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int moved = ox[j] != 0 ? kx + ox[j] : (oy[j] != 0 ? ky + oy[j] : kz + oz[j]);  // j = 0: kz, in range with the centre
            found[j] = j < n_nearby && centre_ok && moved > -kNdtBias && moved < kNdtBias;
            const long long delta = (long long)ox[j] * (1ll << 42) + (long long)oy[j] * (1ll << 21) + (long long)oz[j];  // a constant after unrolling
            key[j] = found[j] ? key0 + (unsigned long long)delta : key0;
            hs[j] = ndt_hash32(key[j], cap_mask);
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) { const NdtSlot sl = slots[hs[j]]; kk[j] = sl.key; vx[j] = sl.vid; }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (found[j] && kk[j] != key[j] && kk[j] != kNdtEmpty) {  // collision on the first probe: walk on (load factor ≤ 0.5)
                size_t h = (hs[j] + 1) & cap_mask;
                for (;;) {
                    const NdtSlot sl = slots[h];
                    if (sl.key == key[j]) { kk[j] = sl.key; vx[j] = sl.vid; break; }
                    if (sl.key == kNdtEmpty) { kk[j] = sl.key; break; }
                    h = (h + 1) & cap_mask;
                }
            }
            found[j] = found[j] && kk[j] == key[j];
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const NdtRecord& R = rec[found[j] ? vx[j] : 0u];  // a voxel that is not there (never seen, or dropped for having too few points, ndt cpp:136-142) reads record 0 and is not accepted
            const double* m = R.mu;
            const double* I = R.info;
            const D3 e{qs.x - m[0], qs.y - m[1], qs.z - m[2]};
            // e.transpose() * v.info_ * e (ndt cpp:416) = (eᵀ·info)·e: the row vector first, then its dot product with e — 12 + 8 FP64
            // operations in three short chains (a sum over the nine e_r·info_rc·e_c terms is 27 in one long chain)
            const double t0 = (e.x * I[0] + e.y * I[3]) + e.z * I[6];
            const double t1 = (e.x * I[1] + e.y * I[4]) + e.z * I[7];
            const double t2 = (e.x * I[2] + e.y * I[5]) + e.z * I[8];
            const double res = (t0 * e.x + t1 * e.y) + t2 * e.z;
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
    hipLaunchKernelGGL(ndt_accum_kernel, grid, dim3(kBlock), 0, s, t->d_slots, t->d_rec, t->cap - 1, t->inv_voxel, t->res_outlier_th,
                       t->n_nearby, src, counts, st, max_n, partials, pts, active, src_of);
    return (int)grid.x;  // partial blocks per scan
}

}  // namespace locgpu
