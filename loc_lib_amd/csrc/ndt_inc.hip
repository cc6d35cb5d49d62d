// loc_lib_amd/csrc/ndt_inc.hip — incremental NDT (NdtMethod::INCREMENTAL_NDT), the mapping flow's default
// (slam_demo/config/slam.yaml:53).
//
// Reference: NdtRegistration::SetIncNdtTargetCloud (ndt_registration.cpp:150-183), UpdateVoxel (:185-236),
// AlignIncNdt (:262-372). Split of work:
//   * host (this file, IncNdtState::ingest): the reference's LRU bookkeeping, point by point in input order — a
//     std::list of voxels with move-to-front on touch and eviction from the tail once `capacity_` is reached (:158-171).
//     This is sequential control logic (which voxels exist), not arithmetic; each live voxel owns a dense slot.
//   * device: per-voxel statistics of the points added by THIS call and the per-iteration sums. `flag_first_scan_` is set
//     to true at the end of every SetIncNdtTargetCloud (:181), so UpdateVoxel always takes its first branch (:186-198):
//     more than one point ⇒ mean, (n−1)-covariance, info = (Σ + 1e-3·I)⁻¹; a single point ⇒ μ = the point, info = 100·I.
//     Voxels not touched by the call keep their previous statistics.
//   * align: AlignIncNdt differs from the direct variant: sums ARE info-weighted (H += Jᵀ·info·J, err += −Jᵀ·info·e,
//     :345-346), effective_num counts accepted (point, voxel) pairs (:343), too few ⇒ `return false` with result = current
//     pose (:349-353), and there is no det(H) test.
#include <list>
#include <unordered_map>
#include <vector>

#include "icp_kernels.hpp"
#include "ndt_inc.hpp"
#include "ndt_kernels.hpp"

namespace locgpu {

// ---------------------------------------------------------------------------------------------- statistics kernels
__global__ __launch_bounds__(kBlock) void inc_sum_kernel(const float4* __restrict__ pts, const int* __restrict__ pt_slot, size_t n,
                                                         const unsigned char* __restrict__ slot_dead, double* sums, int* counts) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0 || slot_dead[s]) return;
    const float4 p = pts[i];
    atomicAdd(&counts[s], 1);
    atomicAdd(&sums[3 * (size_t)s + 0], (double)p.x);
    atomicAdd(&sums[3 * (size_t)s + 1], (double)p.y);
    atomicAdd(&sums[3 * (size_t)s + 2], (double)p.z);
}

__global__ __launch_bounds__(kBlock) void inc_mean_kernel(const int* __restrict__ active, int n_active, const double* sums, const int* counts,
                                                          double* mu) {
    const int a = blockIdx.x * kBlock + threadIdx.x;
    if (a >= n_active) return;
    const int s = active[a];
    const double len = (double)counts[s];
    if (len > 0)
        for (int c = 0; c < 3; ++c) mu[3 * (size_t)s + c] = sums[3 * (size_t)s + c] / len;
}

__global__ __launch_bounds__(kBlock) void inc_cov_kernel(const float4* __restrict__ pts, const int* __restrict__ pt_slot, size_t n,
                                                         const unsigned char* __restrict__ slot_dead, const double* mu, double* cov6) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0 || slot_dead[s]) return;
    const float4 p = pts[i];
    const double dx = (double)p.x - mu[3 * (size_t)s], dy = (double)p.y - mu[3 * (size_t)s + 1], dz = (double)p.z - mu[3 * (size_t)s + 2];
    double* c = cov6 + 6 * (size_t)s;
    atomicAdd(&c[0], dx * dx); atomicAdd(&c[1], dx * dy); atomicAdd(&c[2], dx * dz);
    atomicAdd(&c[3], dy * dy); atomicAdd(&c[4], dy * dz); atomicAdd(&c[5], dz * dz);
}

// UpdateVoxel, first-scan branch (ndt cpp:186-198).
__global__ __launch_bounds__(kBlock) void inc_info_kernel(const int* __restrict__ active, int n_active, const int* counts, const double* cov6,
                                                          double* info) {
    const int a = blockIdx.x * kBlock + threadIdx.x;
    if (a >= n_active) return;
    const int s = active[a];
    const int n = counts[s];
    double* I = info + 9 * (size_t)s;
    if (n > 1) {
        const double l1 = (double)(n - 1);
        const double* c = cov6 + 6 * (size_t)s;
        const double m00 = c[0] / l1 + 1e-3, m01 = c[1] / l1, m02 = c[2] / l1, m11 = c[3] / l1 + 1e-3, m12 = c[4] / l1, m22 = c[5] / l1 + 1e-3;
        const double det = m00 * (m11 * m22 - m12 * m12) - m01 * (m01 * m22 - m12 * m02) + m02 * (m01 * m12 - m11 * m02);
        const double id = 1.0 / det;
        I[0] = (m11 * m22 - m12 * m12) * id; I[1] = (m02 * m12 - m01 * m22) * id; I[2] = (m01 * m12 - m02 * m11) * id;
        I[3] = I[1];                         I[4] = (m00 * m22 - m02 * m02) * id; I[5] = (m02 * m01 - m00 * m12) * id;
        I[6] = I[2];                         I[7] = I[5];                         I[8] = (m00 * m11 - m01 * m01) * id;
    } else if (n == 1) {
        for (int k = 0; k < 9; ++k) I[k] = (k % 4 == 0) ? 1e2 : 0.0;
    }
}

__global__ __launch_bounds__(kBlock) void inc_table_kernel(const unsigned long long* __restrict__ keys_in, const int* __restrict__ slots_in, int n,
                                                           unsigned long long* keys, int* vid, size_t cap_mask) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys_in[i];
    size_t h = ndt_hash(key, cap_mask);
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[h], kNdtEmpty, key);
        if (prev == kNdtEmpty) break;
        h = (h + 1) & cap_mask;
    }
    vid[h] = slots_in[i];
}

__global__ void inc_fill_kernel(unsigned long long* p, size_t n, unsigned long long v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------- accumulate kernel
// Grid (ceil(max_n/256), n_scans). acc[27] = number of accepted (point, voxel) residuals.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void inc_accum_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ vid,
                                                           const double* __restrict__ mu, const double* __restrict__ info, size_t cap_mask,
                                                           double inv_voxel, double res_th, int n_nearby, const float4* __restrict__ src,
                                                           const int* __restrict__ counts, const PoseState* __restrict__ st, int max_n,
                                                           double* __restrict__ partials) {
#pragma clang fp contract(fast)
    const int scan = blockIdx.y;
    if (st[scan].done) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double acc[28];
#pragma unroll
    for (int v = 0; v < 28; ++v) acc[v] = 0.0;
    if (i < counts[scan]) {
        const float4 p = src[(size_t)scan * max_n + i];
        const D3 q{(double)p.x, (double)p.y, (double)p.z};
        const D3 qs = se3_apply(st[scan].q, st[scan].t, q);
        const int kx = (int)(qs.x * inv_voxel), ky = (int)(qs.y * inv_voxel), kz = (int)(qs.z * inv_voxel);
        const int ox[7] = {0, -1, 1, 0, 0, 0, 0}, oy[7] = {0, 0, 0, 1, -1, 0, 0}, oz[7] = {0, 0, 0, 0, 0, -1, 1};
        double Is[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Ie[3] = {0, 0, 0}, n_acc = 0.0;
        // the seven look-ups level by level instead of one chain of dependent gathers after the other (see ndt_accum_kernel); the sums
        // are formed in the order j = 0..6 from the same numbers
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
            if (found[j] && kk[j] != key[j] && kk[j] != kNdtEmpty) {  // collision on the first probe: walk on
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
        for (int j = 0; j < 7; ++j) found[j] = found[j] && vx[j] >= 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int v = found[j] ? vx[j] : 0;  // a voxel that is not there reads record 0 and is not accepted
            const double* m = mu + 3 * (size_t)v;
            const double* I = info + 9 * (size_t)v;
            const double e[3] = {qs.x - m[0], qs.y - m[1], qs.z - m[2]};
            double ie[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) ie[r] = (I[3 * r] * e[0] + I[3 * r + 1] * e[1]) + I[3 * r + 2] * e[2];
            const double res = (e[0] * ie[0] + e[1] * ie[1]) + e[2] * ie[2];
            const bool accept = found[j] && !(isnan(res) || res > res_th);
            n_acc = accept ? n_acc + 1.0 : n_acc;
#pragma unroll
            for (int k = 0; k < 9; ++k) Is[k] = accept ? Is[k] + I[k] : Is[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) Ie[k] = accept ? Ie[k] + ie[k] : Ie[k];
        }
        acc[27] = n_acc;
        if (n_acc > 0.0) {
            // J = [A | I3], A = −R·hat(q) (the same for every voxel of this point) ⇒ Σ_v Jᵀ·info_v·J = Jᵀ·(Σ info_v)·J
            const double* R = st[scan].R;
            double A[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                A[r][0] = -(R[3 * r + 1] * q.z - R[3 * r + 2] * q.y);
                A[r][1] = -(R[3 * r + 2] * q.x - R[3 * r + 0] * q.z);
                A[r][2] = -(R[3 * r + 0] * q.y - R[3 * r + 1] * q.x);
            }
            double J[3][6], IJ[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) { J[r][c] = A[r][c]; J[r][3 + c] = (r == c) ? 1.0 : 0.0; }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) IJ[r][c] = (Is[3 * r] * J[0][c] + Is[3 * r + 1] * J[1][c]) + Is[3 * r + 2] * J[2][c];
            int o = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[o++] = (J[0][a] * IJ[0][b] + J[1][a] * IJ[1][b]) + J[2][a] * IJ[2][b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] = -((J[0][a] * Ie[0] + J[1][a] * Ie[1]) + J[2][a] * Ie[2]);
        }
    }
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

// ---------------------------------------------------------------------------------------------- host state
struct KeyHash {
    size_t operator()(unsigned long long k) const { return (size_t)ndt_hash(k, ~(size_t)0); }
};

struct IncNdtState {
    size_t capacity = 100000;
    double inv_voxel = 1.0;
    std::list<unsigned long long> lru;  // front = most recently touched (data_, ndt_registration.hpp:126)
    struct Entry { std::list<unsigned long long>::iterator it; int slot; };
    std::unordered_map<unsigned long long, Entry, KeyHash> map;  // inc_grids_ (:127)
    std::vector<int> free_slots;
    int n_slots = 0;  // slots ever handed out
    // device
    double *d_mu = nullptr, *d_info = nullptr, *d_sums = nullptr, *d_cov = nullptr;
    int* d_counts = nullptr;
    unsigned char* d_dead = nullptr;
    size_t slot_cap = 0;
    unsigned long long* d_keys = nullptr;
    int* d_vid = nullptr;
    size_t table_cap = 0;
};

#define INC_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

static hipError_t grow(IncNdtState& st, size_t need, hipStream_t s) {
    if (need <= st.slot_cap) return hipSuccess;
    size_t cap = st.slot_cap ? st.slot_cap : 4096;
    while (cap < need) cap *= 2;
    double *mu = nullptr, *info = nullptr, *sums = nullptr, *cov = nullptr;
    int* counts = nullptr;
    unsigned char* dead = nullptr;
    INC_TRY(hipMalloc((void**)&mu, cap * 3 * sizeof(double)));
    INC_TRY(hipMalloc((void**)&info, cap * 9 * sizeof(double)));
    INC_TRY(hipMalloc((void**)&sums, cap * 3 * sizeof(double)));
    INC_TRY(hipMalloc((void**)&cov, cap * 6 * sizeof(double)));
    INC_TRY(hipMalloc((void**)&counts, cap * sizeof(int)));
    INC_TRY(hipMalloc((void**)&dead, cap));
    if (st.slot_cap) {
        INC_TRY(hipMemcpyAsync(mu, st.d_mu, st.slot_cap * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
        INC_TRY(hipMemcpyAsync(info, st.d_info, st.slot_cap * 9 * sizeof(double), hipMemcpyDeviceToDevice, s));
        INC_TRY(hipStreamSynchronize(s));
        (void)hipFree(st.d_mu); (void)hipFree(st.d_info); (void)hipFree(st.d_sums); (void)hipFree(st.d_cov); (void)hipFree(st.d_counts); (void)hipFree(st.d_dead);
    }
    st.d_mu = mu; st.d_info = info; st.d_sums = sums; st.d_cov = cov; st.d_counts = counts; st.d_dead = dead;
    st.slot_cap = cap;
    return hipSuccess;
}

IncNdtState* inc_ndt_create(size_t capacity, double voxel_size) {
    auto* st = new IncNdtState();
    st->capacity = capacity;
    st->inv_voxel = 1.0 / voxel_size;
    return st;
}

void inc_ndt_destroy(IncNdtState* st) {
    if (!st) return;
    (void)hipFree(st->d_mu); (void)hipFree(st->d_info); (void)hipFree(st->d_sums); (void)hipFree(st->d_cov); (void)hipFree(st->d_counts);
    (void)hipFree(st->d_dead); (void)hipFree(st->d_keys); (void)hipFree(st->d_vid);
    delete st;
}

size_t inc_ndt_num_voxels(const IncNdtState* st) { return st ? st->map.size() : 0; }

// SetIncNdtTargetCloud. `host_pts` = the cloud as float4 on the host (for the key pass), `d_pts` = the same on the device.
hipError_t inc_ndt_ingest(IncNdtState& st, const float4* host_pts, const float4* d_pts, size_t n, hipStream_t s, bool* bad_key) {
    *bad_key = false;
    std::vector<int> pt_slot(n);
    std::vector<int> active;            // slots touched by this call, in first-touch order
    std::vector<char> touched;          // per slot
    std::vector<int> died;              // slots evicted during this call: recycled only afterwards
    std::vector<unsigned char> dead;    // per slot, for the device
    auto mark = [&](int slot) {
        if ((size_t)slot >= touched.size()) touched.resize(slot + 1, 0);
        if (!touched[slot]) { touched[slot] = 1; active.push_back(slot); }
    };
    for (size_t i = 0; i < n; ++i) {
        const double x = (double)host_pts[i].x * st.inv_voxel, y = (double)host_pts[i].y * st.inv_voxel, z = (double)host_pts[i].z * st.inv_voxel;
        const int kx = (int)x, ky = (int)y, kz = (int)z;  // (pt * inv_voxel_size_).cast<int>(): truncation toward zero
        if (!ndt_key_in_range(kx, ky, kz)) { *bad_key = true; pt_slot[i] = -1; continue; }
        const unsigned long long key = ndt_pack(kx, ky, kz);
        auto it = st.map.find(key);
        if (it == st.map.end()) {
            int slot;
            if (!st.free_slots.empty()) { slot = st.free_slots.back(); st.free_slots.pop_back(); }
            else slot = st.n_slots++;
            st.lru.push_front(key);
            st.map.emplace(key, IncNdtState::Entry{st.lru.begin(), slot});
            pt_slot[i] = slot;
            mark(slot);
            if (st.lru.size() >= st.capacity) {  // ndt cpp:161-165: drop the least recently used voxel
                const unsigned long long old = st.lru.back();
                auto oit = st.map.find(old);
                died.push_back(oit->second.slot);
                st.map.erase(oit);
                st.lru.pop_back();
            }
        } else {
            st.lru.splice(st.lru.begin(), st.lru, it->second.it);  // touched ⇒ most recent (ndt cpp:169-170)
            it->second.it = st.lru.begin();
            pt_slot[i] = it->second.slot;
            mark(it->second.slot);
        }
    }
    INC_TRY(grow(st, (size_t)std::max(st.n_slots, 1), s));
    dead.assign(st.slot_cap, 0);
    for (int sl : died) dead[sl] = 1;
    // active voxels that were evicted again within this very call are not updated (the reference would dereference a
    // default-constructed iterator there, ndt cpp:178)
    std::vector<int> live_active;
    for (int sl : active) if (!dead[sl]) live_active.push_back(sl);

    int *d_pt_slot = nullptr, *d_active = nullptr;
    hipError_t rc = hipSuccess;
    auto run = [&]() -> hipError_t {
        INC_TRY(hipMalloc((void**)&d_pt_slot, std::max<size_t>(n, 1) * sizeof(int)));
        INC_TRY(hipMalloc((void**)&d_active, std::max<size_t>(live_active.size(), 1) * sizeof(int)));
        INC_TRY(hipMemcpyAsync(d_pt_slot, pt_slot.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
        INC_TRY(hipMemcpyAsync(d_active, live_active.data(), live_active.size() * sizeof(int), hipMemcpyHostToDevice, s));
        INC_TRY(hipMemcpyAsync(st.d_dead, dead.data(), st.slot_cap, hipMemcpyHostToDevice, s));
        INC_TRY(hipMemsetAsync(st.d_counts, 0, st.slot_cap * sizeof(int), s));
        INC_TRY(hipMemsetAsync(st.d_sums, 0, st.slot_cap * 3 * sizeof(double), s));
        INC_TRY(hipMemsetAsync(st.d_cov, 0, st.slot_cap * 6 * sizeof(double), s));
        const unsigned gn = (unsigned)((n + kBlock - 1) / kBlock), ga = (unsigned)((live_active.size() + kBlock - 1) / kBlock);
        if (n && !live_active.empty()) {
            hipLaunchKernelGGL(inc_sum_kernel, dim3(gn), dim3(kBlock), 0, s, d_pts, d_pt_slot, n, st.d_dead, st.d_sums, st.d_counts);
            hipLaunchKernelGGL(inc_mean_kernel, dim3(ga), dim3(kBlock), 0, s, d_active, (int)live_active.size(), st.d_sums, st.d_counts, st.d_mu);
            hipLaunchKernelGGL(inc_cov_kernel, dim3(gn), dim3(kBlock), 0, s, d_pts, d_pt_slot, n, st.d_dead, st.d_mu, st.d_cov);
            hipLaunchKernelGGL(inc_info_kernel, dim3(ga), dim3(kBlock), 0, s, d_active, (int)live_active.size(), st.d_counts, st.d_cov, st.d_info);
        }
        // rebuild the key → slot table from the live map
        std::vector<unsigned long long> keys;
        std::vector<int> slots;
        keys.reserve(st.map.size());
        slots.reserve(st.map.size());
        for (const auto& kv : st.map) { keys.push_back(kv.first); slots.push_back(kv.second.slot); }
        size_t cap = 1024;
        while (cap < 2 * keys.size()) cap <<= 1;
        if (cap != st.table_cap) {
            (void)hipFree(st.d_keys); (void)hipFree(st.d_vid);
            st.d_keys = nullptr; st.d_vid = nullptr;
            INC_TRY(hipMalloc((void**)&st.d_keys, cap * sizeof(unsigned long long)));
            INC_TRY(hipMalloc((void**)&st.d_vid, cap * sizeof(int)));
            st.table_cap = cap;
        }
        hipLaunchKernelGGL(inc_fill_kernel, dim3((unsigned)((cap + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, st.d_keys, cap, kNdtEmpty);
        unsigned long long* d_kin = nullptr;
        int* d_sin = nullptr;
        if (!keys.empty()) {
            INC_TRY(hipMalloc((void**)&d_kin, keys.size() * sizeof(unsigned long long)));
            hipError_t e = hipMalloc((void**)&d_sin, slots.size() * sizeof(int));
            if (e != hipSuccess) { (void)hipFree(d_kin); return e; }
            (void)hipMemcpyAsync(d_kin, keys.data(), keys.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
            (void)hipMemcpyAsync(d_sin, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice, s);
            hipLaunchKernelGGL(inc_table_kernel, dim3((unsigned)((keys.size() + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_kin, d_sin, (int)keys.size(),
                               st.d_keys, st.d_vid, cap - 1);
        }
        hipError_t e = hipStreamSynchronize(s);
        if (d_kin) (void)hipFree(d_kin);
        if (d_sin) (void)hipFree(d_sin);
        if (e != hipSuccess) return e;
        return hipGetLastError();
    };
    rc = run();
    if (d_pt_slot) (void)hipFree(d_pt_slot);
    if (d_active) (void)hipFree(d_active);
    for (int sl : died) st.free_slots.push_back(sl);  // recycle only now: no slot is reused within the call that freed it
    return rc;
}

void launch_inc_accum(const IncNdtState* st, double res_th, int n_nearby, const float4* src, const int* counts, const PoseState* ps, int max_n,
                      int n_scans, double* partials, hipStream_t s) {
    dim3 grid((max_n + kBlock - 1) / kBlock, n_scans);
    hipLaunchKernelGGL(inc_accum_kernel, grid, dim3(kBlock), 0, s, st->d_keys, st->d_vid, st->d_mu, st->d_info, st->table_cap - 1, st->inv_voxel, res_th,
                       n_nearby, src, counts, ps, max_n, partials);
}

size_t inc_ndt_dump(const IncNdtState* st, int32_t* keys, double* mu, double* info, size_t cap) {
    std::vector<double> h_mu((size_t)st->n_slots * 3), h_info((size_t)st->n_slots * 9);
    if (st->n_slots) {
        (void)hipMemcpy(h_mu.data(), st->d_mu, h_mu.size() * sizeof(double), hipMemcpyDeviceToHost);
        (void)hipMemcpy(h_info.data(), st->d_info, h_info.size() * sizeof(double), hipMemcpyDeviceToHost);
    }
    size_t n = 0;
    for (const auto& kv : st->map) {
        if (n < cap) {
            const unsigned long long k = kv.first;
            const size_t sl = (size_t)kv.second.slot;
            if (keys) {
                keys[3 * n + 0] = (int)((k >> 42) & 0x1FFFFF) - kNdtBias;
                keys[3 * n + 1] = (int)((k >> 21) & 0x1FFFFF) - kNdtBias;
                keys[3 * n + 2] = (int)(k & 0x1FFFFF) - kNdtBias;
            }
            if (mu) for (int c = 0; c < 3; ++c) mu[3 * n + c] = h_mu[3 * sl + c];
            if (info) for (int c = 0; c < 9; ++c) info[9 * n + c] = h_info[9 * sl + c];
        }
        ++n;
    }
    return n;
}

const void* inc_ndt_table_ptr(const IncNdtState* st) { return st ? (const void*)st->d_keys : nullptr; }

}  // namespace locgpu
