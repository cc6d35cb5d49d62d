// loc_lib_amd/csrc/icp_kernels.hip — kernel bodies + launchers of the ICP hot path (see icp_kernels.hpp).
#include "icp_kernels.hpp"
#include "launch.hpp"
#include "search_walk.hpp"

#include <algorithm>
#include <cstdlib>

namespace locgpu {

// ---------------------------------------------------------------------------------------------
// K1, instrumented / unbounded-tree form: one thread per source point, the exact one-pass traversal with the libstdc++ heap.
// Grid (ceil(max_n/256), n_scans). Runs only for visit counting (bench.py's algorithmic bytes) and for trees with non-finite or huge
// coordinates; the hot kernel is icp_search_walk_kernel below.
template <int KMAX, int D, bool COUNT>
__global__ __launch_bounds__(kBlock) void icp_search_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                            const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                            uint32_t* __restrict__ nn, size_t nn_pitch, int max_n, int k, float alpha_eff,
                                                            int skip_nonfinite, unsigned long long* __restrict__ visit_totals, uint32_t* __restrict__ touched,
                                                            const int* __restrict__ active, const int* __restrict__ src_of) {
    __shared__ uint32_t s_far[D][kBlock];
    __shared__ float s_d2[D][kBlock];
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (st[scan].done) return;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    if (i >= counts[scan]) return;
    const size_t gi = (size_t)scan * max_n + i;
    const float4 p = src[(size_t)(src_of ? src_of[scan] : scan) * max_n + i];
    uint32_t out[KMAX];
    int cnt = 0;
    uint32_t nvis = 0, lvis = 0;
    const bool finite = !skip_nonfinite || (isfinite(p.x) && isfinite(p.y) && isfinite(p.z));  // pcl::isFinite, icp cpp:64 (P2P only)
    if (finite) {
        const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
        KnnHeap<KMAX> heap;
        tree_knn_flat<KMAX, D, COUNT>(tree, (float)qs.x, (float)qs.y, (float)qs.z, k, alpha_eff, s_far, s_d2, tid, heap, nvis, lvis, touched);
        heap_to_sorted<KMAX>(heap, out, cnt);
    } else {
#pragma unroll
        for (int j = 0; j < KMAX; ++j) out[j] = kInvalidSlot;
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j < k) nn[(size_t)j * nn_pitch + gi] = out[j];
    if (COUNT) {
        atomicAdd(&visit_totals[0], (unsigned long long)nvis);
        atomicAdd(&visit_totals[1], (unsigned long long)lvis);
        atomicAdd(&visit_totals[2], 1ull);
    }
}

// Streaming accesses (elements touched once per launch) are marked non-temporal so that they do not evict tree nodes from L2 —
// measured one by one on the bench workload (search ms per 256-scan step): the search kernel's neighbour-list stores and source loads
// 25.0 -> 24.3; the fit kernel's neighbour-list loads -> 24.0. NOT the fit kernel's source loads (-> 24.7: the next search wants them cached).
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_once(const float4* __restrict__ p) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    return float4{v.x, v.y, v.z, v.w};
}

// K1: the traversal of search_walk.hpp. One-wave workgroups; dynamic LDS = DF rows x LANES x 8 B (the stack is the ONLY LDS of the
// kernel: a row below its bottom must lie outside the allocation). `dummy` = slot of the sentinel leaf.
// Every lane of the wave must call walk_query (wave-wide ballots inside); a lane without a query passes valid = false.
template <int K, int ROWB, bool STAMP = false, int WIN = 0>
__device__ __forceinline__ void walk_query(__amdgpu_buffer_rsrc_t rsrc, const uint2* __restrict__ tree, Walk<K>& w, bool valid, float alpha_eff, int T, uint32_t dummy,
                                           uint32_t col_addr, int cap) {
#pragma unroll
    for (int j = 0; j < K; ++j) { w.d[j] = __builtin_inff(); w.id[j] = kInvalidSlot; }
    w.slow = 0;
    // finite-arithmetic precondition of the traversal (the tree is `bounded`): anything else goes to the exact kernel
    const bool sane = fabsf(w.qx) < 1e18f && fabsf(w.qy) < 1e18f && fabsf(w.qz) < 1e18f;
    if (valid && sane) {
        walk_descend<K, ROWB, WIN>(rsrc, tree, w, T, col_addr);
    } else {
        w.cur = dummy; w.avail = 0; w.c3n = 0; w.slow = valid ? 1u : 0u;
    }
#ifndef LOCGPU_K1_C
#define LOCGPU_K1_C 2  // internal steps per round (A/B builds: -DLOCGPU_K1_C=1|3)
#endif
    walk_rounds_capped<K, ROWB, LOCGPU_K1_C, STAMP>(rsrc, w, alpha_eff, dummy, col_addr, cap);
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) w.slow |= w.d[j] == w.d[j + 1] ? 1u : 0u;  // equal distances in the final set: heap pop order is layout-dependent
}

// One or two queries of the wave that need the exact traversal (ties): answered here, the wave's LDS (its stacks are dead now)
// serving as their two stack columns. Returns false when there are more (the caller appends them to the redo list).
template <int K>
__device__ __forceinline__ bool walk_exact_in_wave(const uint2* __restrict__ tree, bool slow, float qx, float qy, float qz, float alpha_eff, uint2* lds,
                                                   uint32_t* __restrict__ nn, size_t nn_pitch, size_t gi, unsigned long long* __restrict__ search_stats) {
    const unsigned long long slow_mask = __ballot(slow);
    if (slow_mask == 0ull || __popcll(slow_mask) > 2) return slow_mask == 0ull;
    if (slow) {
        uint32_t(*s_far)[2] = reinterpret_cast<uint32_t(*)[2]>(lds);
        float(*s_d2)[2] = reinterpret_cast<float(*)[2]>(reinterpret_cast<char*>(lds) + 64 * 2 * sizeof(uint32_t));
        const int col = __popcll(slow_mask & ((1ull << (threadIdx.x & 63)) - 1ull));
        KnnHeap<K> heap;
        uint32_t nvis = 0, lvis = 0, out[K];
        int cnt;
        tree_knn_flat<K, 64, false, 2>(tree, qx, qy, qz, K, alpha_eff, s_far, s_d2, col, heap, nvis, lvis);
        heap_to_sorted<K>(heap, out, cnt);
#pragma unroll
        for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = out[j];
        if (search_stats) atomicAdd(&search_stats[1], 1ull);
    }
    return true;
}

// LANES = active lanes per wave = stack columns (64; 16 for launches that cannot fill the chip anyway: a wave's time is its
// longest traversal, and with 128-byte rows every level fits in LDS — T = 0 — so that no query needs the deep pass).
// WIN > 0: of the levels ≥ T only the deepest WIN = DF − 2 of each query's first descent are stored (search_walk.hpp, walk_descend).
template <int K, int DF, int LANES = 64, bool STAMP = false, int WIN = 0>
__global__ __launch_bounds__(64) void icp_search_walk_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                             const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                             uint32_t* __restrict__ nn, size_t nn_pitch, int max_n, float alpha_eff, int T,
                                                             unsigned int tree_bytes, uint32_t dummy, int skip_nonfinite, uint32_t* __restrict__ redo_list,
                                                             unsigned int* __restrict__ redo_count, uint32_t* __restrict__ deep_list,
                                                             unsigned int* __restrict__ deep_count, unsigned long long* __restrict__ search_stats,
                                                             const int* __restrict__ active, const int* __restrict__ src_of) {
    extern __shared__ uint2 s_dyn[];
    constexpr int ROWB = LANES * 8;
    static_assert(DF * ROWB >= 64 * 2 * 8, "exact stack columns do not fit");
    // The traversal reads rows BELOW the stack's bottom (they must lie outside the workgroup's LDS allocation and read as 0): the
    // stack has to start at LDS address 0, i.e. the kernel must own no other LDS. The launcher checks the code object
    // (search_kernels_lds_ok); this is the last line of defence — wrong neighbours or an endless loop otherwise.
    if ((uint32_t)(size_t)s_dyn != 0u) __builtin_trap();
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;  // later chunks of an alignment launch only the scans still open
    const int tid = threadIdx.x;
    if (st[scan].done) return;
    const int i = blockIdx.x * LANES + tid;
    if (tid >= LANES || i >= counts[scan]) return;
    const size_t gi = (size_t)scan * max_n + i;
    // a scan pool keeps the points in an arena of source regions and says which one a slot reads (scan_pool.hip); else slot = region
    const float4 p = load_once(&src[(size_t)(src_of ? src_of[scan] : scan) * max_n + i]);
    // pcl::isFinite, icp cpp:64 (P2P only): such a point has no neighbours; its lane stays in the wave with nothing to do (the list
    // it stores at the end is the empty one)
    const bool finite = !skip_nonfinite || (isfinite(p.x) && isfinite(p.y) && isfinite(p.z));
    if (search_stats && finite) atomicAdd(&search_stats[0], 1ull);
    const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tree, 0, (int)tree_bytes, 0x00020000);
    Walk<K> w;
    w.qx = (float)qs.x; w.qy = (float)qs.y; w.qz = (float)qs.z;
    static_assert(WIN <= DF - 2, "the window lies in the rows above the two candidate rows");
    walk_query<K, ROWB, STAMP, WIN>(rsrc, tree, w, finite, alpha_eff, T, dummy, (uint32_t)(size_t)(&s_dyn[tid]), DF);  // DF rows: a stack that outgrows them → deep pass
    if (STAMP && search_stats) {
        // diagnostic build (LOCGPU_STAMP=1): rounds each lane needed against the rounds its wave ran — the kernel's lane
        // efficiency — and the per-query round counts behind the neighbour-list area of redo_list (2 x pitch entries in this build)
        atomicAdd(&search_stats[4], (unsigned long long)w.rounds);
        if ((tid & 63) == 0) { atomicAdd(&search_stats[9], (unsigned long long)w.wave_rounds); atomicAdd(&search_stats[12], 1ull); }
        atomicAdd(&search_stats[13], (unsigned long long)w.wave_rounds);  // rounds paid by this lane's wave, summed over lanes
        redo_list[nn_pitch + gi] = w.rounds;
    }
    const bool deep = w.c3n == 1u;
    const bool slow = !deep && w.slow != 0u;
    if (!deep && !slow) {
#pragma unroll
        for (int j = 0; j < K; ++j) __builtin_nontemporal_store(w.id[j], &nn[(size_t)j * nn_pitch + gi]);
    }
    wave_append(deep_list, deep_count, deep, (uint32_t)gi);
    if (LANES == 16) {
        // the one-scan launch: every tied query is answered here, two at a time on the wave's dead stacks, so that the latency path
        // has no redo launch behind it (a wave holds 16 queries; more than two ties in one is a lattice map)
        bool pending = slow;
        for (;;) {
            const unsigned long long m = __ballot(pending);
            if (m == 0ull) break;
            const bool mine = pending && __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull)) < 2;
            (void)walk_exact_in_wave<K>(tree, mine, w.qx, w.qy, w.qz, alpha_eff, s_dyn, nn, nn_pitch, gi, search_stats);
            pending = pending && !mine;
        }
    } else if (!walk_exact_in_wave<K>(tree, slow, w.qx, w.qy, w.qz, alpha_eff, s_dyn, nn, nn_pitch, gi, search_stats)) {
        wave_append(redo_list, redo_count, slow, (uint32_t)gi);
    }
}

// The deep pass: the same traversal with EVERY level stored (T = 0, D + 2 rows) over the list of queries whose un-stored top
// levels could not be resolved from two candidates (≈1e-3 of them at 15 rows). One-wave workgroups, grid-stride over the list
// whose length lives on the device.
template <int K, int D, int LANES = 64>
__global__ __launch_bounds__(64) void icp_search_walk_list_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                  const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                  float alpha_eff, unsigned int tree_bytes, uint32_t dummy, const uint32_t* __restrict__ list,
                                                                  const unsigned int* __restrict__ n_list, uint32_t* __restrict__ redo_list,
                                                                  unsigned int* __restrict__ redo_count, unsigned long long* __restrict__ search_stats,
                                                                  const int* __restrict__ src_of) {
    extern __shared__ uint2 s_dyn[];
    constexpr int ROWB = LANES * 8;
    static_assert((D + 2) * ROWB >= 64 * 2 * 8, "exact stack columns do not fit");
    if ((uint32_t)(size_t)s_dyn != 0u) __builtin_trap();  // see icp_search_walk_kernel
    const unsigned int n = *n_list;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && search_stats) atomicAdd(&search_stats[2], (unsigned long long)n);
    if (tid >= LANES) return;  // LANES stack columns per wave (see icp_search_walk_kernel)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tree, 0, (int)tree_bytes, 0x00020000);
    // a short list is spread thin: these are the long traversals, and a wave's time is the longest among its lanes
    const unsigned int per_wave = min((unsigned int)LANES, max(1u, (n + gridDim.x - 1) / gridDim.x));
    for (unsigned int r0 = blockIdx.x * per_wave; r0 < n; r0 += gridDim.x * per_wave) {
        const unsigned int r = r0 + (unsigned int)tid;
        const bool valid = (unsigned int)tid < per_wave && r < n;
        const uint32_t gi = valid ? list[r] : 0u;
        Walk<K> w;
        w.qx = w.qy = w.qz = 0.f;
        if (valid) {
            const int scan = (int)(gi / (uint32_t)max_n);
            const float4 p = src_of ? src[(size_t)src_of[scan] * max_n + (gi - (uint32_t)scan * (uint32_t)max_n)] : src[gi];
            const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
            w.qx = (float)qs.x; w.qy = (float)qs.y; w.qz = (float)qs.z;
        }
        walk_query<K, ROWB>(rsrc, tree, w, valid, alpha_eff, 0, dummy, (uint32_t)(size_t)(&s_dyn[tid]), 0x7fffffff);  // D + 2 rows ≥ depth: cannot overflow
        const bool slow = valid && w.slow != 0u;
        if (valid && !slow) {
#pragma unroll
            for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = w.id[j];
        }
        if (!walk_exact_in_wave<K>(tree, slow, w.qx, w.qy, w.qz, alpha_eff, s_dyn, nn, nn_pitch, gi, search_stats))
            wave_append(redo_list, redo_count, slow, gi);
    }
}

// Exact recomputation of the queries the fast kernel could not finish (≈1e-6 of them on real maps: ≈30 per 256-scan launch).
// One-wave workgroups over the list. A short list is dealt one query per WAVE (lane 0): 30 queries sharing a wave would each
// pay the others' heap-emulation branches and the longest traversal; alone in its wave a query costs its own ≈40 dependent loads.
// A long list (a lattice map: every distance ties) fills all 64 lanes of every wave.
constexpr int kRedoWaves = 2048;
// The deep pass: full waves, 2048 of them (16 lanes to a wave on 8192 waves — a wave's time is the longest traversal among its lanes,
// and these are the long ones — was measured: search 17.25 → 17.47 ms per 256-scan step; the pass is issue-bound like the walk kernel)
constexpr int kDeepLanes = 64, kDeepWaves = 2048;
template <int KMAX, int D>
__global__ __launch_bounds__(64) void icp_search_redo_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                             const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch,
                                                             int max_n, int k, float alpha_eff, const uint32_t* __restrict__ redo_list,
                                                             const unsigned int* __restrict__ redo_count, unsigned long long* __restrict__ search_stats,
                                                             const int* __restrict__ src_of) {
    __shared__ uint32_t s_far[D][64];
    __shared__ float s_d2[D][64];
    const unsigned int n = *redo_count;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && search_stats) atomicAdd(&search_stats[1], (unsigned long long)n);
    const unsigned int per_wave = min(64u, (n + gridDim.x - 1) / gridDim.x);  // queries a wave takes per round
    if ((unsigned)tid >= per_wave) return;
    for (unsigned int r = blockIdx.x * per_wave + tid; r < n; r += gridDim.x * per_wave) {
        const size_t gi = redo_list[r];
        const int scan = (int)(gi / (size_t)max_n);
        const float4 p = src_of ? src[(size_t)src_of[scan] * max_n + (gi - (size_t)scan * max_n)] : src[gi];
        const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
        KnnHeap<KMAX> heap;
        uint32_t nvis = 0, lvis = 0, out[KMAX];
        int cnt;
        tree_knn_flat<KMAX, D, false, 64>(tree, (float)qs.x, (float)qs.y, (float)qs.z, k, alpha_eff, s_far, s_d2, tid, heap, nvis, lvis);
        heap_to_sorted<KMAX>(heap, out, cnt);
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j < k) nn[(size_t)j * nn_pitch + gi] = out[j];
    }
}

// Plain k-NN over given queries (SearchPointInterface::FindNearstPoints). 1-D grid.
template <int KMAX, int D>
__global__ __launch_bounds__(kBlock) void knn_query_kernel(const uint2* __restrict__ tree, const float* __restrict__ queries, size_t nq,
                                                           int k, float alpha_eff, int32_t* __restrict__ out_idx,
                                                           uint32_t* __restrict__ visits) {
    __shared__ uint32_t s_far[D][kBlock];
    __shared__ float s_d2[D][kBlock];
    const int tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * kBlock + tid;
    if (i >= nq) return;
    KnnHeap<KMAX> heap;
    uint32_t nvis = 0, lvis = 0;
    tree_knn<KMAX, D, true>(tree, queries[3 * i], queries[3 * i + 1], queries[3 * i + 2], k, alpha_eff, s_far, s_d2, tid, heap, nvis, lvis);
    uint32_t out[KMAX];
    int cnt;
    heap_to_sorted<KMAX>(heap, out, cnt);
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j < k) out_idx[i * k + j] = (out[j] == kInvalidSlot) ? -1 : (int32_t)(tree[out[j]].y & 0x3FFFFFFFu);  // original point index
    if (visits) { visits[2 * i] = nvis; visits[2 * i + 1] = lvis; }
}

// ---------------------------------------------------------------------------------------------
// Block reduction of `acc[0..NV)` → partials[block][0..NV). Wave butterfly, then LDS across the 4 waves.
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NV], double* __restrict__ dst) {
    __shared__ double s_part[kBlock / 64][kAccW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const double s = wave_sum(acc[v]);
        if (lane == 0) s_part[wave][v] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = s_part[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) s += s_part[w][threadIdx.x];
        dst[threadIdx.x] = s;
    }
}

// acc layout: [0..20] upper triangle of H row by row (00 01 .. 05 11 12 .. 55), [21..26] B, [27] effective_num.
template <int ROWS>
__device__ __forceinline__ void add_rows(double (&acc)[28], const double (&J)[ROWS][6], const double (&e)[ROWS]) {
#pragma clang fp contract(fast)
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
            double s = J[0][i] * J[0][j];
#pragma unroll
            for (int r = 1; r < ROWS; ++r) s += J[r][i] * J[r][j];
            acc[o++] += s;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = -J[0][i] * e[0];
#pragma unroll
        for (int r = 1; r < ROWS; ++r) s += -J[r][i] * e[r];
        acc[21 + i] += s;
    }
}

// Block-cooperative form of add_rows (the plane and line kernels). Keeping the 28 sums per thread costs 56 VGPRs that are live across
// the whole plane/line fit and cap those kernels at three waves per SIMD. Instead every thread leaves its point's row
// {J0..J5, -e, fit} in LDS, and thread (entry, slice) adds the products of ITS entry over its slice of the block's 256 rows: the
// same 28 FMAs per point and thread, one accumulator. Entry → the two row components it multiplies: 0..20 the upper triangle of
// JᵀJ, 21..26 J·(−e), 27 fit·fit (a count), 28..31 idle. Every thread of the block must call add()/store() (barriers inside).
constexpr int kAccPad = kBlock + 2;  // LDS row stride: consecutive rows four banks apart
struct RowAccum {
    int ent, slice, ra, rb;
    double sum;
    bool used;
    __device__ __forceinline__ void init() {
        ent = threadIdx.x & 31;
        slice = threadIdx.x >> 5;
        ra = 7; rb = 7;
        int o = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {
                if (o == ent) { ra = i; rb = j; }
                ++o;
            }
        if (ent >= 21 && ent < 27) { ra = ent - 21; rb = 6; }
        sum = 0.0;
        used = false;
    }
    template <int ROWS>
    __device__ __forceinline__ void add(double (&s_row)[8][kAccPad], const double (&J)[ROWS][6], const double (&neg_e)[ROWS], double fitted) {
#pragma clang fp contract(fast)
        const int tid = threadIdx.x;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (used) __syncthreads();  // the previous rows have been consumed
            used = true;
#pragma unroll
            for (int c = 0; c < 6; ++c) s_row[c][tid] = J[r][c];
            s_row[6][tid] = neg_e[r];
            s_row[7][tid] = r == 0 ? fitted : 0.0;
            __syncthreads();
            if (ent < 28) {
#pragma unroll 8
                for (int k = 0; k < 32; ++k) {  // rows slice, slice + 8, …: neighbouring slices read neighbouring LDS banks
                    const int col = k * (kBlock / 32) + slice;
                    sum += s_row[ra][col] * s_row[rb][col];
                }
            }
        }
    }
    __device__ __forceinline__ void store(double (&s_slice)[kBlock / 32][32], double* __restrict__ dst) {
        s_slice[slice][ent] = sum;
        __syncthreads();
        if (threadIdx.x < 28) {
            double t = s_slice[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < kBlock / 32; ++w) t += s_slice[w][threadIdx.x];
            dst[threadIdx.x] = t;
        }
    }
};

// R·hat(q), coefficient order of the oracle's left-to-right 3×3 product (zeros of hat() drop out exactly).
__device__ __forceinline__ void R_hat(const double* R, const D3& q, double (&Rh)[3][3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Rh[r][0] = R[3 * r + 1] * q.z - R[3 * r + 2] * q.y;
        Rh[r][1] = R[3 * r + 2] * q.x - R[3 * r + 0] * q.z;
        Rh[r][2] = R[3 * r + 0] * q.y - R[3 * r + 1] * q.x;
    }
}


// K2, P2Plane: IcpRegistration::CaculateMatrixHAndBP2Plane (icp_registration.cpp:161-213) + math::FitPlane (math_utils.h:112-136).
// FIT: 0 = plane_null_vector (4-column one-sided Jacobi), 1 = plane_null_vector_secular with the former as its fall-back.
template <int FIT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void icp_plane_accum_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                 const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                                 const uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                 double max_plane_distance, double* __restrict__ partials, int kPlanePts,
                                                                 const int* __restrict__ active, const int* __restrict__ src_of) {
    __shared__ double s_row[8][kAccPad];
    __shared__ double s_slice[kBlock / 32][32];
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (st[scan].done) return;  // uniform per block
    const int tid = threadIdx.x;
    RowAccum ra;  // see there: the 28 sums are not kept per thread
    ra.init();
#pragma unroll 1
    for (int pp = 0; pp < kPlanePts; ++pp) {
        const int i = (blockIdx.x * kPlanePts + pp) * kBlock + tid;
        double J[1][6] = {{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
        double neg_e[1] = {0.0};
        double fitted = 0.0;
        if (i < counts[scan]) {
            const size_t gi = (size_t)scan * max_n + i;
            // all five indices and the point in one round trip (not: the fifth, then the rest behind its test)
            uint32_t slot[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) slot[j] = __builtin_nontemporal_load(&nn[(size_t)j * nn_pitch + gi]);
            const float4 p = src[(size_t)(src_of ? src_of[scan] : scan) * max_n + i];
            if (slot[4] != kInvalidSlot) {  // nn.size() > 3: k=5 yields 5 or (k > size_) none
                D3 nb[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) nb[j] = leaf_point(tree, slot[j]);
                double n4[4];
                if constexpr (FIT == 1) {
                    if (!plane_null_vector_secular(nb, n4)) plane_null_vector(nb, n4);
                } else {
                    plane_null_vector(nb, n4);
                }
                const D3 n3{n4[0], n4[1], n4[2]};
                bool fit = true;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const double err = dot3(n3, nb[j]) + n4[3];
                    if (err * err > 1e-2) fit = false;
                }
                if (fit) {
                    fitted = 1.0;  // effective_num++ before the residual gate (icp cpp:184)
                    const D3 q{(double)p.x, (double)p.y, (double)p.z};  // the transformed point is only needed from here on
                    const D3 qs = se3_apply(st[scan].q, st[scan].t, q);
                    const double dis = dot3(n3, qs) + n4[3];
                    if (!(fabs(dis) > max_plane_distance)) {
                        const double* R = st[scan].R;
                        double nR[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) nR[c] = -n3.x * R[c] + (-n3.y * R[3 + c] + -n3.z * R[6 + c]);  // the reference binary's order (libLocUtils.so 0x58745-0x58801)
                        J[0][0] = nR[1] * q.z - nR[2] * q.y;
                        J[0][1] = nR[2] * q.x - nR[0] * q.z;
                        J[0][2] = nR[0] * q.y - nR[1] * q.x;
                        J[0][3] = n3.x; J[0][4] = n3.y; J[0][5] = n3.z;
                        neg_e[0] = -dis;
                    }
                }
            }
        }
        ra.add<1>(s_row, J, neg_e, fitted);
    }
    ra.store(s_slice, partials + ((size_t)scan * gridDim.x + blockIdx.x) * kAccW);
}

// K2', P2P: CaculateMatrixHAndBP2P (icp_registration.cpp:57-103), including the /16 on the rotation block.
__global__ __launch_bounds__(kBlock) void icp_point_accum_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                 const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                                 const uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                 double max_nn_distance, double* __restrict__ partials, int pts,
                                                                 const int* __restrict__ active, const int* __restrict__ src_of) {
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (st[scan].done) return;
    double acc[28];
#pragma unroll
    for (int v = 0; v < 28; ++v) acc[v] = 0.0;
#pragma unroll 1
    for (int pp = 0; pp < pts; ++pp) {  // several points per thread before the 28-value wave reduction (see the plane kernel)
    const int i = (blockIdx.x * pts + pp) * kBlock + threadIdx.x;
    if (i < counts[scan]) {
        const size_t gi = (size_t)scan * max_n + i;
        const uint32_t s0 = nn[gi];
        if (s0 != kInvalidSlot) {
            const float4 p = src[(size_t)(src_of ? src_of[scan] : scan) * max_n + i];
            const D3 q{(double)p.x, (double)p.y, (double)p.z};
            const D3 qs = se3_apply(st[scan].q, st[scan].t, q);
            const D3 e3 = leaf_point(tree, s0) - qs;
            const double dis2 = dot3(e3, e3);
            if (!(dis2 > max_nn_distance)) {
                acc[27] += 1.0;
                double Rh[3][3];
                R_hat(st[scan].R, q, Rh);
                double J[3][6];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { J[r][c] = Rh[r][c] / 16; J[r][3 + c] = (r == c) ? -1.0 : 0.0; }
                const double e[3] = {e3.x, e3.y, e3.z};
                add_rows<3>(acc, J, e);
            }
        }
    }
    }
    block_reduce_store<28>(acc, partials + ((size_t)scan * gridDim.x + blockIdx.x) * kAccW);
}

// K2', P2Line: CaculateMatrixHAndBP2Line (icp_registration.cpp:105-159) + math::FitLine (math_utils.h:138-163).
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void icp_line_accum_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                                const uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                double max_line_distance, double* __restrict__ partials, int pts,
                                                                const int* __restrict__ active, const int* __restrict__ src_of) {
    __shared__ double s_row[8][kAccPad];
    __shared__ double s_slice[kBlock / 32][32];
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (st[scan].done) return;
    RowAccum ra;
    ra.init();
#pragma unroll 1
    for (int pp = 0; pp < pts; ++pp) {
    const int i = (blockIdx.x * pts + pp) * kBlock + threadIdx.x;
    double J[3][6], neg_e[3] = {0.0, 0.0, 0.0};
    double fitted = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) J[r][c] = 0.0;
    if (i < counts[scan]) {
        const size_t gi = (size_t)scan * max_n + i;
        uint32_t slot[5];  // one round trip for the five indices and the point (see the plane kernel)
#pragma unroll
        for (int j = 0; j < 5; ++j) slot[j] = nn[(size_t)j * nn_pitch + gi];
        const float4 p = src[(size_t)(src_of ? src_of[scan] : scan) * max_n + i];
        if (slot[4] != kInvalidSlot) {  // nn.size() == 5
            const D3 q{(double)p.x, (double)p.y, (double)p.z};
            const D3 qs = se3_apply(st[scan].q, st[scan].t, q);
            D3 nb[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) nb[j] = leaf_point(tree, slot[j]);
            D3 sum{0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 5; ++j) sum = sum + nb[j];
            const D3 p0{sum.x / 5.0, sum.y / 5.0, sum.z / 5.0};
            D3 dl[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) dl[j] = nb[j] - p0;
            const D3 d = line_direction(dl);
            bool fit = true;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const D3 c = cross3(d, nb[j] - p0);
                if (dot3(c, c) > max_line_distance) fit = false;
            }
            if (fit) {
                fitted = 1.0;
                const D3 e3 = cross3(d, qs - p0);  // SO3::hat(d) * (qs - p0)
                if (!(sqrt(dot3(e3, e3)) > max_line_distance)) {
                    const double hd[3][3] = {{0.0, -d.z, d.y}, {d.z, 0.0, -d.x}, {-d.y, d.x, 0.0}};
                    const double hq[3][3] = {{0.0, -q.z, q.y}, {q.z, 0.0, -q.x}, {-q.y, q.x, 0.0}};
                    const double* R = st[scan].R;
                    double hR[3][3], A[3][3];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            double s = hd[r][0] * R[c];
                            s += hd[r][1] * R[3 + c];
                            s += hd[r][2] * R[6 + c];
                            hR[r][c] = s;
                        }
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            double s = hR[r][0] * hq[0][c];
                            s += hR[r][1] * hq[1][c];
                            s += hR[r][2] * hq[2][c];
                            A[r][c] = s;
                        }
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) { J[r][c] = -A[r][c]; J[r][3 + c] = hd[r][c]; }
                    neg_e[0] = -e3.x; neg_e[1] = -e3.y; neg_e[2] = -e3.z;
                }
            }
        }
    }
    ra.add<3>(s_row, J, neg_e, fitted);
    }
    ra.store(s_slice, partials + ((size_t)scan * gridDim.x + blockIdx.x) * kAccW);
}

// Sum of the block partials of one scan, per column, in a fixed order (chunk c takes rows c, c + 8, …; the chunks are then added in
// order). Many loads are in flight per thread: a single-scan alignment has 450 rows, and a row-by-row load → add chain made this the
// longest part of the solve kernel. Returns the column total in threads 0..kAccW-1 (0 elsewhere). Ends with the block synchronised.
__device__ __forceinline__ double reduce_partials(const double* __restrict__ rows, int blocks_per_scan, bool mine, double (*s_sum)[kAccW]) {
    const int col = threadIdx.x & (kAccW - 1), chunk = threadIdx.x / kAccW;
    constexpr int kChunks = kBlock / kAccW;
    double s = 0.0;
    if (mine && col < 28) {
        constexpr int kFlight = 32;  // loads in flight per thread (round 5: 8 → 32: a one-scan alignment's 450 rows are two rounds instead of seven; the order of the sums — row after row within a chunk — is unchanged, so are the bits)
        for (int b = chunk; b < blocks_per_scan; b += kChunks * kFlight) {
            double v[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; ++u) {
                const int idx = b + u * kChunks;
                v[u] = idx < blocks_per_scan ? rows[(size_t)idx * kAccW + col] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kFlight; ++u) s += v[u];
        }
    }
    s_sum[chunk][col] = s;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x < kAccW) {
        t = s_sum[0][threadIdx.x];
#pragma unroll
        for (int c = 1; c < kChunks; ++c) t += s_sum[c][threadIdx.x];
    }
    __syncthreads();
    return t;
}

// ---------------------------------------------------------------------------------------------
// K3: one 256-thread block per scan. Sums the block partials in a fixed order, then thread 0 runs the
// reference's checks and update (icp_registration.cpp:204-211 + 362-375; ndt_registration.cpp:435-459).
// hb_out (optional): per scan 44 doubles = H (36, row-major), B (6), effective_num, ok.
// scans (optional): the scans to solve — block i takes scan scans[i] (a scan pool solves its open slots only, scan_pool.hip).
__device__ __forceinline__ void gn_update(const double* tot, PoseState& ps, const GnParams& prm, int do_update, double* __restrict__ hb) {
    double H[36], B[6], dx[6] = {0, 0, 0, 0, 0, 0};
    int o = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { H[6 * i + j] = tot[o]; H[6 * j + i] = tot[o]; ++o; }
    for (int i = 0; i < 6; ++i) B[i] = tot[21 + i];
    const long long eff = (long long)tot[27];
    bool ok;
    const double det = lu6_det_solve_reg(H, B, dx);  // in registers: unrolled, pivot rows swapped in with selects (device_math.hpp)
    auto write_hb = [&](double okv) {
        if (!hb) return;
        for (int i = 0; i < 36; ++i) hb[i] = H[i];
        for (int i = 0; i < 6; ++i) hb[36 + i] = B[i];
        hb[42] = (double)eff;
        hb[43] = okv;
    };
    if (prm.method == 3) {
        // direct NDT: det(H)==0 is tested FIRST and aborts the whole alignment (ndt cpp:435-436)
        if (det == 0.0) {
            ps.status = 1; ps.done = 1; ps.iterations += 1; ps.last_eff = eff;
            write_hb(0.0);
            return;
        }
        ok = eff >= prm.min_effective_pts;
    } else if (prm.method == 4) {
        // incremental NDT: too few accepted residuals ⇒ `result_pose = pose; return false` (ndt cpp:349-353); no det(H) test
        ok = eff >= prm.min_effective_pts;
        if (!ok) {
            ps.status = 2; ps.done = 1; ps.iterations += 1; ps.last_eff = eff;
            write_hb(0.0);
            return;
        }
    } else {
        ok = (eff >= prm.min_effective_pts) && !(det == 0.0);
    }
    write_hb(ok ? 1.0 : 0.0);
    ps.last_eff = eff;
    if (!do_update) return;
    ps.iterations += 1;
    if (ok) {
        if (prm.method == 0)
            for (int i = 0; i < 6; ++i) dx[i] = dx[i] / 16;  // dx = H.inverse()/16 * err (icp cpp:287)
        se3_apply_update(ps.q, ps.t, dx);
        quat_to_R(ps.q, ps.R);
        // dx.norm() as the reference's binary sums it: three packets p0 + (p1 + p2), then low + high (libLocUtils.so 0x5b113-0x5b185)
        const double nrm = sqrt((dx[0] * dx[0] + (dx[2] * dx[2] + dx[4] * dx[4])) + (dx[1] * dx[1] + (dx[3] * dx[3] + dx[5] * dx[5])));
        ps.last_dx_norm = nrm;
        if (nrm < prm.eps) { ps.converged = 1; ps.done = 1; }
    }
    if (ps.iterations >= prm.max_iteration) ps.done = 1;
}

__global__ __launch_bounds__(kBlock) void gn_solve_kernel(const double* __restrict__ partials, int blocks_per_scan, PoseState* __restrict__ st,
                                                          GnParams prm, int do_update, double* __restrict__ hb_out, unsigned int* __restrict__ list_counts,
                                                          const int* __restrict__ scans, GnPost post) {
    __shared__ double s_sum[kBlock / kAccW][kAccW];
    const int scan = scans ? scans[blockIdx.x] : (int)blockIdx.x;
    // The search stage's work-list counters (walk kernel → deep pass → redo kernel) are consumed by now: zero them for the next iteration's
    // search instead of paying two fill launches per iteration (a single-scan alignment is launch-latency bound).
    if (list_counts && blockIdx.x == 0 && threadIdx.x < 4) list_counts[threadIdx.x] = 0u;
    if (st[scan].done) return;
    const double col_total = reduce_partials(partials + (size_t)scan * blocks_per_scan * kAccW, blocks_per_scan, true, s_sum);
    if (threadIdx.x < kAccW) s_sum[0][threadIdx.x] = col_total;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot[28];
    for (int v = 0; v < 28; ++v) tot[v] = s_sum[0][v];
    PoseState& ps = st[scan];
    gn_update(tot, ps, prm, do_update, hb_out ? hb_out + 44 * (size_t)scan : nullptr);
    if (post.word) {
        // A one-scan alignment paced from the host (locgpu_api.hip, align_finish): a word in pinned host memory says which iteration
        // this was — the host launches the next iteration but one when it sees it — and a finished scan's result goes there too: no
        // chunk of idle launches, no copy, no stream synchronisation on the latency path. Plain stores to fine-grained host memory
        // (they write through); a system-scope RELEASE here writes the whole L2 back — ≈12 µs per iteration, measured — so the
        // record is sealed by a checksum instead of a fence: the host takes it when the sum over what it reads matches.
        const unsigned long long tag = ((unsigned long long)post.call << 32) | ((unsigned long long)(unsigned int)ps.iterations << 1) | (ps.done ? 1ull : 0ull);
        if (ps.done) {
            GnPostRecord r;
            for (int i = 0; i < 4; ++i) r.w[i] = __double_as_longlong(ps.q[i]);
            for (int i = 0; i < 3; ++i) r.w[4 + i] = __double_as_longlong(ps.t[i]);
            r.w[7] = __double_as_longlong(ps.last_dx_norm);
            r.w[8] = (unsigned long long)ps.last_eff;
            r.w[9] = ((unsigned long long)(unsigned int)ps.converged << 32) | (unsigned int)ps.status;
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            volatile u64x2* to = reinterpret_cast<volatile u64x2*>(post.record);
            for (int i = 0; i < GnPostRecord::kWords / 2; ++i) to[i] = u64x2{r.w[2 * i], r.w[2 * i + 1]};
            *reinterpret_cast<volatile u64x2*>(post.word) = u64x2{tag, gn_post_sum(tag, r)};
        } else {
            __hip_atomic_store(post.word, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// First half of gn_solve_kernel for sharded batches (see launch.hpp): one block per GLOBAL scan.
// owned (optional, scan pools): owned[g] != 0 where this rank holds the points of slot g; then first = 0 and n_local = all slots.
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double* __restrict__ partials, int blocks_per_scan, const PoseState* __restrict__ st_all,
                                                              int first, int n_local, double* __restrict__ acc, const unsigned char* __restrict__ owned) {
    __shared__ double s_sum[kBlock / kAccW][kAccW];
    const int g = blockIdx.x;
    const int scan = g - first;
    const bool mine = scan >= 0 && scan < n_local && !st_all[g].done && (!owned || owned[g]);  // a finished scan's partials are stale: contribute zeros (nobody reads them)
    const double t = reduce_partials(partials + (size_t)(mine ? scan : 0) * blocks_per_scan * kAccW, blocks_per_scan, mine, s_sum);
    if (threadIdx.x < kAccW) acc[(size_t)g * kAccW + threadIdx.x] = threadIdx.x < 28 ? t : 0.0;
}

// Instrumented pass: number of tree slots a search launch read at all (bitmap of tree_knn_flat), added to totals[3]; the bitmap is cleared.
__global__ __launch_bounds__(kBlock) void count_touched_kernel(uint32_t* __restrict__ touched, size_t n_words, unsigned long long* __restrict__ totals) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_words; i += (size_t)gridDim.x * kBlock) {
        c += (unsigned long long)__popc(touched[i]);
        touched[i] = 0u;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&totals[3], c);
}

// pcl::transformPointCloud with the float32 4×4 (icp_registration.cpp:241): ((m0·x + m1·y) + m2·z) + m3 per row.
// The 3×4 matrix travels as a kernel argument (no upload in front of the launch); the output is packed x, y, z — what goes back to the
// caller's cloud, whose other fields are the source's (locgpu_api.hip, write_output_cloud).
__global__ __launch_bounds__(kBlock) void transform_cloud_kernel(const float4* __restrict__ src, size_t n, M12f m, float* __restrict__ dst_xyz) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    const float* m12 = m.v;
    dst_xyz[3 * i + 0] = ((m12[0] * p.x + m12[1] * p.y) + m12[2] * p.z) + m12[3];
    dst_xyz[3 * i + 1] = ((m12[4] * p.x + m12[5] * p.y) + m12[6] * p.z) + m12[7];
    dst_xyz[3 * i + 2] = ((m12[8] * p.x + m12[9] * p.y) + m12[10] * p.z) + m12[11];
}

// Test hook: the search stage's slot lists as original point indices (what knn_query_kernel reports), out[(gi * k) + j].
__global__ __launch_bounds__(kBlock) void nn_to_index_kernel(const uint2* __restrict__ tree, const uint32_t* __restrict__ nn, size_t nn_pitch, size_t n_queries,
                                                             int k, int32_t* __restrict__ out) {
    const size_t gi = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (gi >= n_queries) return;
    for (int j = 0; j < k; ++j) {
        const uint32_t slot = nn[(size_t)j * nn_pitch + gi];
        out[gi * k + j] = slot == kInvalidSlot ? -1 : (int32_t)(tree[slot].y & 0x3FFFFFFFu);
    }
}

// ---------------------------------------------------------------------------------------------
// Launchers (declared in launch.hpp).
void launch_nn_to_index(const uint2* tree, const uint32_t* nn, size_t nn_pitch, size_t n_queries, int k, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(nn_to_index_kernel, dim3((unsigned)((n_queries + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, tree, nn, nn_pitch, n_queries, k, out);
}

template <int KMAX, int D>
static void launch_search_kd(const SearchArgs& a, hipStream_t s) {
    dim3 grid((a.max_n + kBlock - 1) / kBlock, a.active ? a.n_active : a.n_scans);
    if (a.visit_totals)
        hipLaunchKernelGGL((icp_search_kernel<KMAX, D, true>), grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.k, a.alpha_eff, a.skip_nonfinite, a.visit_totals, a.touched, a.active, a.src_of);
    else
        hipLaunchKernelGGL((icp_search_kernel<KMAX, D, false>), grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.k, a.alpha_eff, a.skip_nonfinite, a.visit_totals, nullptr, a.active, a.src_of);
}

template <int KMAX>
static bool launch_search_k(const SearchArgs& a, hipStream_t s) {
    if (a.depth <= 32) launch_search_kd<KMAX, 32>(a, s);
    else if (a.depth <= 40) launch_search_kd<KMAX, 40>(a, s);
    else if (a.depth <= 64) launch_search_kd<KMAX, 64>(a, s);
    else return false;
    return true;
}
// Stored LDS stack rows per lane in the 64-lane walk kernel (8 B each; the levels above them are kept as candidates, search_walk.hpp).
// One-wave workgroups (waves retire independently), so the rows set the waves a CU holds: 15 → 7.5 KB → 21 waves per CU. The balance
// is between occupancy and the queries that go to the deep pass — measured on the bench workload (search ms per 256-scan step, round 2):
// 12 → 30.0, 13 → 28.2, 14 → 26.2, 15 → 25.5, 16 → 27.2, 20 → 29.0, 24 → 35.9.
// LOCGPU_FAST_STACK=12|24 selects other depths: the parity tests run the deep-pass hand-over at stack depths where it is common.
static int fast_stack_depth() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LOCGPU_FAST_STACK");
        v = e ? atoi(e) : 10;
        if (v != 12 && v != 15 && v != 24) v = 10;
    }
    return v;
}

// The walk kernels' LDS stack must start at LDS address 0 (search_walk.hpp: rows below the bottom are read and must fall outside
// the allocation). That holds exactly when the kernel has no static LDS of its own — checked on the code object for every
// instantiation the launchers below can select, once per process, when the first context is created (not at launch time: a launch
// may happen under stream capture).
template <auto Kernel>
static bool no_static_lds() {
    hipFuncAttributes attr{};
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(Kernel)) == hipSuccess && attr.sharedSizeBytes == 0;
}
template <int K, int D>
static bool walk_kernels_ok_kd() {
    return no_static_lds<icp_search_walk_kernel<K, 10, 64, false, 8>>() && no_static_lds<icp_search_walk_kernel<K, 12>>() && no_static_lds<icp_search_walk_kernel<K, 15>>() && no_static_lds<icp_search_walk_kernel<K, 24>>() &&
           no_static_lds<icp_search_walk_kernel<K, D + 2, 16>>() && no_static_lds<icp_search_walk_list_kernel<K, D, kDeepLanes>>();
}
bool search_kernels_lds_ok() {
    static const bool ok = walk_kernels_ok_kd<1, 32>() && walk_kernels_ok_kd<1, 40>() && walk_kernels_ok_kd<1, 64>() &&
                           walk_kernels_ok_kd<5, 32>() && walk_kernels_ok_kd<5, 40>() && walk_kernels_ok_kd<5, 64>();
    return ok;
}

// The search stage of one Gauss–Newton iteration: walk kernel → deep pass → exact redo (the last two over device-side lists that are
// almost always short). a.redo_count / a.redo_count2 are zero here: the caller clears them before an alignment's first iteration,
// gn_solve_kernel after every search.
template <int K, int D, int DF, int WIN = 0>
static bool launch_walk_kd(const SearchArgs& a, hipStream_t s) {
    static const bool stamp = [] { const char* e = getenv("LOCGPU_STAMP"); return e && atoi(e) != 0; }();
    const uint32_t dummy = (uint32_t)(a.tree_bytes / 8);  // the sentinel leaf behind the tree
    const unsigned int rsrc_bytes = (unsigned int)a.tree_bytes + 16u;
    const int n_launch = a.active ? a.n_active : a.n_scans;  // grid.y: the scans still open (a.active) or all of them
    if (!stamp && (size_t)((a.max_n + 63) / 64) * n_launch <= 2048) {
        // fewer than 2048 full waves (one or two scans): quarter-filled waves with every level stored — no deep pass, and the
        // 16-lane kernel answers its ties itself: no redo launch on the latency path
        dim3 g1((a.max_n + 15) / 16, n_launch);
        hipLaunchKernelGGL((icp_search_walk_kernel<K, D + 2, 16>), g1, dim3(64), (D + 2) * 16 * 8, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.alpha_eff, 0, rsrc_bytes, dummy, a.skip_nonfinite, a.redo_list, a.redo_count, a.redo_list2, a.redo_count2, a.search_stats, a.active, a.src_of);
        return true;
    }
    // rows 0/1 of the DF stored rows hold the candidates of the un-stored levels, the other DF-2 rows one level each. Queries whose
    // un-stored levels need more than two candidates, or whose candidate descent outgrows the rows, go through a.redo_list2 to the
    // deep pass (every level stored), ties to the exact redo kernel.
    // DF = 10: the sliding window (WIN = 8); T as for 15 rows — the un-stored top levels are sized by what a TYPICAL descent can store,
    // the window takes care of the deeper ones (LOCGPU_K1_T_OFFSET: measurement knob)
    static const int t_off = [] { const char* e = getenv("LOCGPU_K1_T_OFFSET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 13; }();
    const int stored = WIN ? t_off : DF - 2;
    const int Tw = a.depth > stored ? a.depth - stored : 0;
    dim3 g2((a.max_n + 63) / 64, n_launch);
    // LOCGPU_K1_LDS_BYTES (measurement only, profiles/r06_k1_occupancy.md): a larger dynamic LDS allocation than the DF rows need, i.e.
    // fewer waves per CU. The extra bytes sit ABOVE the stack (a push beyond row DF lands in them instead of off the allocation:
    // harmless, `cap` sends such a query to the deep pass either way).
    static const bool lds_knob = getenv("LOCGPU_K1_LDS_BYTES") != nullptr;  // present at the first launch: re-read at every launch (tools/k1_occupancy.py sweeps it in one process)
    unsigned lds_bytes = DF * 64 * 8;
    if (lds_knob) { const char* e = getenv("LOCGPU_K1_LDS_BYTES"); lds_bytes = (unsigned)std::max(e ? atoi(e) : 0, DF * 64 * 8); }
    if (stamp)  // diagnostic build: counts rounds per lane and per wave (search_stats[4], [9], [12], [13]; per query at redo_list[pitch + gi]); timing meaningless
        hipLaunchKernelGGL((icp_search_walk_kernel<K, DF, 64, true, WIN>), g2, dim3(64), DF * 64 * 8, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.alpha_eff, Tw, rsrc_bytes, dummy, a.skip_nonfinite, a.redo_list, a.redo_count, a.redo_list2, a.redo_count2, a.search_stats, a.active, a.src_of);
    else
        hipLaunchKernelGGL((icp_search_walk_kernel<K, DF, 64, false, WIN>), g2, dim3(64), lds_bytes, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.alpha_eff, Tw, rsrc_bytes, dummy, a.skip_nonfinite, a.redo_list, a.redo_count, a.redo_list2, a.redo_count2, a.search_stats, a.active, a.src_of);
    // 2048 one-wave blocks of 17 KB LDS are all resident at once (nine fit a CU): the first iteration's ≈140 k deep queries take one
    // traversal per wave instead of two or three in sequence (search 18.07 → 17.80 ms per 256-scan step; 4608: 17.96)
    hipLaunchKernelGGL((icp_search_walk_list_kernel<K, D, kDeepLanes>), dim3(kDeepWaves), dim3(64), (D + 2) * kDeepLanes * 8, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n,
                       a.alpha_eff, rsrc_bytes, dummy, a.redo_list2, a.redo_count2, a.redo_list, a.redo_count, a.search_stats, a.src_of);
    hipLaunchKernelGGL((icp_search_redo_kernel<K, D>), dim3(kRedoWaves), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.k,
                       a.alpha_eff, a.redo_list, a.redo_count, a.search_stats, a.src_of);
    return true;
}
template <int K, int D>
static bool launch_walk_d(const SearchArgs& a, hipStream_t s) {
    switch (fast_stack_depth()) {
        default: {  // 10 rows = 5 120 B: 32 waves per CU (profiles/r06_k1_occupancy.md)
            static const int win = [] { const char* e = getenv("LOCGPU_K1_WIN"); return e ? atoi(e) : 8; }();
            if (win == 6) return launch_walk_kd<K, D, 10, 6>(a, s);
            if (win == 7) return launch_walk_kd<K, D, 10, 7>(a, s);
            return launch_walk_kd<K, D, 10, 8>(a, s);
        }
        case 12: {
            static const int win = [] { const char* e = getenv("LOCGPU_K1_WIN"); return e ? atoi(e) : 0; }();
            if (win == 8) return launch_walk_kd<K, D, 12, 8>(a, s);
            if (win == 9) return launch_walk_kd<K, D, 12, 9>(a, s);
            return launch_walk_kd<K, D, 12>(a, s);
        }
        case 24: return launch_walk_kd<K, D, 24>(a, s);
        case 15: return launch_walk_kd<K, D, 15>(a, s);  // round 2-5's shape: 13 fixed levels, 21 waves per CU
    }
}
template <int K>
static bool launch_walk_k(const SearchArgs& a, hipStream_t s) {
    if (a.depth <= 32) return launch_walk_d<K, 32>(a, s);
    if (a.depth <= 40) return launch_walk_d<K, 40>(a, s);
    if (a.depth <= 64) return launch_walk_d<K, 64>(a, s);
    return false;
}

template <int K>
static bool launch_redo_k(const SearchArgs& a, hipStream_t s) {
#define LOCGPU_REDO(D) hipLaunchKernelGGL((icp_search_redo_kernel<K, D>), dim3(kRedoWaves), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.k, \
                                          a.alpha_eff, a.redo_list, a.redo_count, a.search_stats, a.src_of)
    if (a.depth <= 32) LOCGPU_REDO(32);
    else if (a.depth <= 40) LOCGPU_REDO(40);
    else if (a.depth <= 64) LOCGPU_REDO(64);
    else return false;
#undef LOCGPU_REDO
    return true;
}
bool launch_icp_search_redo(const SearchArgs& a, hipStream_t s) {
    if (a.k == 1) return launch_redo_k<1>(a, s);
    if (a.k == 5) return launch_redo_k<5>(a, s);
    return false;
}

// The walk traversal with every level stored over `list` (length *n_list on the device) — what the grid search could not settle,
// with a.alpha_eff = 1 — then the exact redo kernel for its ties. a.redo_count must have been zeroed on the stream.
template <int K>
static bool launch_walk_list_k(const SearchArgs& a, const uint32_t* list, const unsigned int* n_list, hipStream_t s) {
    const uint32_t dummy = (uint32_t)(a.tree_bytes / 8);
    const unsigned int rsrc_bytes = (unsigned int)a.tree_bytes + 16u;
#define LOCGPU_LIST(D) hipLaunchKernelGGL((icp_search_walk_list_kernel<K, D, kDeepLanes>), dim3(kDeepWaves), dim3(64), (D + 2) * kDeepLanes * 8, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, \
                                          a.alpha_eff, rsrc_bytes, dummy, list, n_list, a.redo_list, a.redo_count, a.search_stats, a.src_of)
    if (a.depth <= 32) LOCGPU_LIST(32);
    else if (a.depth <= 40) LOCGPU_LIST(40);
    else if (a.depth <= 64) LOCGPU_LIST(64);
    else return false;
#undef LOCGPU_LIST
    return launch_redo_k<K>(a, s);
}
bool launch_icp_search_list(const SearchArgs& a, const uint32_t* list, const unsigned int* n_list, hipStream_t s) {
    if (a.k == 1) return launch_walk_list_k<1>(a, list, n_list, s);
    if (a.k == 5) return launch_walk_list_k<5>(a, list, n_list, s);
    return false;
}

// LOCGPU_PLANE_FIT (read once): 1 = secular-equation fit (plane_null_vector_secular), 0 = the 4-column Jacobi fit.
int plane_fit_mode() {
    static const int mode = [] {
        const char* e = getenv("LOCGPU_PLANE_FIT");
        return e && *e ? atoi(e) : 1;
    }();
    return mode;
}

bool launch_icp_search(const SearchArgs& a, hipStream_t s) {
    // Instrumented (visit-counting) runs and trees with non-finite / huge coordinates (no redo list) use the exact one-pass kernel.
    const bool exact_only = a.visit_totals != nullptr || !a.redo_list || !a.redo_list2;
    if (!exact_only) {
        if (a.k == 1) return launch_walk_k<1>(a, s);
        if (a.k == 5) return launch_walk_k<5>(a, s);
        return false;
    }
    if (a.k == 1) return launch_search_k<1>(a, s);
    if (a.k == 5) return launch_search_k<5>(a, s);
    return false;
}

template <int KMAX>
static bool launch_knn_k(const uint2* tree, int depth, const float* q, size_t nq, int k, float alpha_eff, int32_t* out, uint32_t* visits,
                         hipStream_t s) {
    dim3 grid((unsigned)((nq + kBlock - 1) / kBlock));
    if (depth <= 32) hipLaunchKernelGGL((knn_query_kernel<KMAX, 32>), grid, dim3(kBlock), 0, s, tree, q, nq, k, alpha_eff, out, visits);
    else if (depth <= 40) hipLaunchKernelGGL((knn_query_kernel<KMAX, 40>), grid, dim3(kBlock), 0, s, tree, q, nq, k, alpha_eff, out, visits);
    else if (depth <= 64) hipLaunchKernelGGL((knn_query_kernel<KMAX, 64>), grid, dim3(kBlock), 0, s, tree, q, nq, k, alpha_eff, out, visits);
    else return false;
    return true;
}
bool launch_knn_query(const uint2* tree, int depth, const float* q, size_t nq, int k, float alpha_eff, int32_t* out, uint32_t* visits,
                      hipStream_t s) {
    if (k == 1) return launch_knn_k<1>(tree, depth, q, nq, k, alpha_eff, out, visits, s);
    if (k <= 5) return launch_knn_k<5>(tree, depth, q, nq, k, alpha_eff, out, visits, s);
    if (k <= 8) return launch_knn_k<8>(tree, depth, q, nq, k, alpha_eff, out, visits, s);
    return false;
}

// Points per thread of the accumulate kernels for a batch of n_scans scans of at most max_n points: amortise the block reduction
// when the batch already fills the chip; 1 for small launches (latency). The plane kernel's reduction is cheap (LDS rows, see
// there): 4 is as good as 8 and leaves a finer tail; the line and point kernels still pay a 28-value wave reduction per block.
int icp_accum_split(int method, int max_n, int n_scans) {
    const long total_blocks = (long)((max_n + kBlock - 1) / kBlock) * n_scans;
    return total_blocks >= 8192 ? (method == 2 ? 4 : 8) : (total_blocks >= 4096 ? 4 : (total_blocks >= 2048 ? 2 : 1));
}

int launch_icp_accum(int method, const AccumArgs& a, hipStream_t s) {
    const int blocks = (a.max_n + kBlock - 1) / kBlock;
    int pts = icp_accum_split(method, a.max_n, a.n_scans);  // ALL scans of the batch, open or not: the split — hence the order of the sums — must not depend on a.active
    if (a.split_scans > 0) pts = icp_accum_split(method, a.max_n, a.split_scans);
    const dim3 grid((blocks + pts - 1) / pts, a.active ? a.n_active : a.n_scans);
    if (method == 2) {
        if (plane_fit_mode() == 1)
            hipLaunchKernelGGL(icp_plane_accum_kernel<1>, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active, a.src_of);
        else
            hipLaunchKernelGGL(icp_plane_accum_kernel<0>, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active, a.src_of);
    } else if (method == 1)
        hipLaunchKernelGGL(icp_line_accum_kernel, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active, a.src_of);
    else
        hipLaunchKernelGGL(icp_point_accum_kernel, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active, a.src_of);
    return (int)grid.x;
}

void launch_gn_solve(const double* partials, int blocks_per_scan, PoseState* st, int n_scans, const GnParams& prm, int do_update, double* hb_out,
                     unsigned int* list_counts, hipStream_t s, const int* scans, const GnPost* post) {
    hipLaunchKernelGGL(gn_solve_kernel, dim3(n_scans), dim3(kBlock), 0, s, partials, blocks_per_scan, st, prm, do_update, hb_out, list_counts, scans, post ? *post : GnPost{});
}

void launch_sum_partials(const double* partials, int blocks_per_scan, const PoseState* st_all, int first, int n_local, int n_total, double* acc,
                         hipStream_t s, const unsigned char* owned) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(n_total), dim3(kBlock), 0, s, partials, blocks_per_scan, st_all, first, n_local, acc, owned);
}

void launch_count_touched(uint32_t* touched, size_t n_words, unsigned long long* totals, hipStream_t s) {
    hipLaunchKernelGGL(count_touched_kernel, dim3(1024), dim3(kBlock), 0, s, touched, n_words, totals);
}

void launch_transform_cloud(const float4* src, size_t n, const M12f& m12, float* dst_xyz, hipStream_t s) {
    hipLaunchKernelGGL(transform_cloud_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, src, n, m12, dst_xyz);
}

}  // namespace locgpu
