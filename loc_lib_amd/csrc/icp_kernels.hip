// loc_lib_amd/csrc/icp_kernels.hip — kernel bodies + launchers of the ICP hot path (see icp_kernels.hpp).
#include "icp_kernels.hpp"
#include "launch.hpp"
#include "search_walk.hpp"

#include <cstdlib>

namespace locgpu {

// ---------------------------------------------------------------------------------------------
// K1: one thread per source point. Grid (ceil(max_n/256), n_scans).
template <int KMAX, int D, bool COUNT, int VARIANT = 0>
__global__ __launch_bounds__(kBlock) void icp_search_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                            const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                            uint32_t* __restrict__ nn, size_t nn_pitch, int max_n, int k, float alpha_eff,
                                                            int skip_nonfinite, unsigned long long* __restrict__ visit_totals, uint32_t* __restrict__ touched) {
    __shared__ uint32_t s_far[D][kBlock];
    __shared__ float s_d2[D][kBlock];
    const int scan = blockIdx.y;
    if (st[scan].done) return;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    if (i >= counts[scan]) return;
    const size_t gi = (size_t)scan * max_n + i;
    const float4 p = src[gi];
    uint32_t out[KMAX];
    int cnt = 0;
    uint32_t nvis = 0, lvis = 0;
    const bool finite = !skip_nonfinite || (isfinite(p.x) && isfinite(p.y) && isfinite(p.z));  // pcl::isFinite, icp cpp:64 (P2P only)
    if (finite) {
        const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
        KnnHeap<KMAX> heap;
        if (VARIANT != 9) tree_knn_flat<KMAX, D, COUNT>(tree, (float)qs.x, (float)qs.y, (float)qs.z, k, alpha_eff, s_far, s_d2, tid, heap, nvis, lvis, touched);
        else tree_knn<KMAX, D, COUNT>(tree, (float)qs.x, (float)qs.y, (float)qs.z, k, alpha_eff, s_far, s_d2, tid, heap, nvis, lvis);
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

// K1 fast path (see tree_knn_fast): exact for every query it completes; the others go to redo_list.
// search_stats[0] += queries handled here, search_stats[1] += queries handed to the exact redo kernel.
template <int K, int DF, int BLK, bool STAMP = false>
__global__ __launch_bounds__(BLK) void icp_search_fast_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                 const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                                 uint32_t* __restrict__ nn, size_t nn_pitch, int max_n, float alpha_eff, int T,
                                                                 unsigned int tree_bytes, int skip_nonfinite, uint32_t* __restrict__ redo_list,
                                                                 unsigned int* __restrict__ redo_count,
                                                                 unsigned long long* __restrict__ search_stats, int lanes) {
    __shared__ uint2 s_stack[DF][BLK];
    const int scan = blockIdx.y;
    if (st[scan].done) return;
    const int tid = threadIdx.x;
    // `lanes` < BLK: a small launch (one scan) spread over twice as many, half-filled waves — it cannot fill the chip anyway, and
    // a wave's time is the longest traversal among its lanes
    const int i = blockIdx.x * lanes + tid;
    if (tid >= lanes || i >= counts[scan]) return;
    const size_t gi = (size_t)scan * max_n + i;
    const float4 p = load_once(&src[gi]);
    if (skip_nonfinite && !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {  // pcl::isFinite, icp cpp:64 (P2P only)
#pragma unroll
        for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = kInvalidSlot;
        return;
    }
    if (search_stats) atomicAdd(&search_stats[0], 1ull);  // one aggregated add per wave; only when stats were requested
    const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
    SortedSet<K> set;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tree, 0, (int)tree_bytes, 0x00020000);
    unsigned long long diag[6] = {0, 0, 0, 0, 0, 0};
    // The fast traversal assumes finite arithmetic (the tree is `bounded`, kdtree_build.cpp): a query that is NaN, infinite or
    // astronomically far goes straight to the exact kernel.
    const float fqx = (float)qs.x, fqy = (float)qs.y, fqz = (float)qs.z;
    const bool sane = fabsf(fqx) < 1e18f && fabsf(fqy) < 1e18f && fabsf(fqz) < 1e18f;
    const bool slow = !sane || tree_knn_fast<K, DF, BLK, STAMP>(rsrc, fqx, fqy, fqz, alpha_eff, T, s_stack, tid, set, diag, (STAMP && search_stats) ? search_stats + 15 : nullptr);
    if (STAMP && search_stats) {
        // per-wave maxima (what the wave pays) and per-lane sums (useful work); search_stats[2..] are diagnostic slots
        unsigned long long wmax[5];
        for (int j = 0; j < 5; ++j) {
            unsigned long long v = diag[j];
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
            wmax[j] = v;
        }
        for (int j = 0; j < 5; ++j) atomicAdd(&search_stats[2 + j], diag[j]);                 // lane sums
        if ((tid & 63) == 0) { for (int j = 0; j < 5; ++j) atomicAdd(&search_stats[7 + j], wmax[j]); atomicAdd(&search_stats[12], 1ull); }  // wave maxima, waves
        atomicAdd(&search_stats[16 + min((int)(diag[2] >> 1), 63)], 1ull);
        redo_list[nn_pitch + gi] = (uint32_t)(diag[5] << 16) | (uint32_t)min(diag[2], 65535ull);  // the diagnostic build allocates 2 x pitch entries
        if ((tid & 63) == 0) atomicAdd(&search_stats[80 + min((int)(wmax[2] >> 1), 63)], 1ull);
    }
    if (BLK == 64 && !STAMP) {
        // One or two unfinished queries in this wave (the usual case: ≈30 of 29.5 M): answer them here with the exact traversal, the
        // wave's LDS (all of its fast stacks are dead now) serving as their two stack columns — instead of a separate kernel whose
        // ≈55 µs the whole iteration waits for. More than two (a lattice map: every distance ties) go to the list as before.
        const unsigned long long slow_mask = __ballot(slow);
        if (slow_mask != 0ull && __popcll(slow_mask) <= 2) {
            if (slow) {
                uint32_t(*s_far)[2] = reinterpret_cast<uint32_t(*)[2]>(&s_stack[0][0]);
                float(*s_d2)[2] = reinterpret_cast<float(*)[2]>(reinterpret_cast<char*>(&s_stack[0][0]) + 64 * 2 * sizeof(uint32_t));
                static_assert(sizeof(s_stack) >= 64 * 2 * 8, "exact stack columns do not fit");
                const int col = __popcll(slow_mask & ((1ull << tid) - 1ull));
                KnnHeap<K> heap;
                uint32_t nvis = 0, lvis = 0, out[K];
                int cnt;
                tree_knn_flat<K, 64, false, 2>(tree, fqx, fqy, fqz, K, alpha_eff, s_far, s_d2, col, heap, nvis, lvis);
                heap_to_sorted<K>(heap, out, cnt);
#pragma unroll
                for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = out[j];
                if (search_stats) atomicAdd(&search_stats[1], 1ull);
            } else {
#pragma unroll
                for (int j = 0; j < K; ++j) __builtin_nontemporal_store(set.id[j], &nn[(size_t)j * nn_pitch + gi]);
            }
            return;
        }
    }
    if (slow) {
        redo_list[atomicAdd(redo_count, 1u)] = (uint32_t)gi;
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) __builtin_nontemporal_store(set.id[j], &nn[(size_t)j * nn_pitch + gi]);
    }
}

// K1, round 3: the traversal of search_walk.hpp. One-wave workgroups; dynamic LDS = DF rows x 64 lanes x 8 B (the stack is the
// ONLY LDS of the kernel: a row below its bottom must lie outside the allocation). `dummy` = slot of the sentinel leaf.
// Every lane of the wave must call walk_query (wave-wide ballots inside); a lane without a query passes valid = false.
template <int K, int ROWB, int MODE, bool STAMP = false>  // MODE 0: flat trips, 2: rounds, 12: rounds capped at two internal steps (default)
__device__ __forceinline__ void walk_query(__amdgpu_buffer_rsrc_t rsrc, const uint2* __restrict__ tree, Walk<K>& w, bool valid, float alpha_eff, int T, uint32_t dummy,
                                           uint32_t col_addr, int cap, int stop_at = 0) {
#pragma unroll
    for (int j = 0; j < K; ++j) { w.d[j] = __builtin_inff(); w.id[j] = kInvalidSlot; }
    w.slow = 0;
    // finite-arithmetic precondition of the traversal (the tree is `bounded`): anything else goes to the exact kernel
    const bool sane = fabsf(w.qx) < 1e18f && fabsf(w.qy) < 1e18f && fabsf(w.qz) < 1e18f;
    if (valid && sane) {
        walk_descend<K, ROWB>(rsrc, tree, w, T, col_addr);
    } else {
        w.cur = dummy; w.avail = 0; w.c3n = 0; w.slow = valid ? 1u : 0u;
    }
    if (MODE == 2) walk_rounds<K, ROWB>(rsrc, w, alpha_eff, dummy, col_addr, cap);
    else if (MODE == 12) walk_rounds_capped<K, ROWB, 2, STAMP>(rsrc, w, alpha_eff, dummy, col_addr, cap, stop_at);
    else do walk_trip<K, ROWB>(rsrc, w, alpha_eff, dummy, col_addr, cap); while (__ballot(w.cur != dummy || w.avail > 0) != 0ull);
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

// ---- straggler hand-over (round 4; OPT-IN, LOCGPU_WALK_STOP=8 — see walk_stop_lanes() for why it is off). A wave of the walk kernel stops once at most `stop_at` of its lanes still have work; those lanes
// write what the traversal needs to go on — query, result set, position, flags and the live rows of their LDS stack — to a spill
// record, and icp_search_walk_cont_kernel continues them, 64 to a wave. Same traversal, same order, same results; the rounds a wave
// pays for its last few lanes are paid by full waves instead (paid rounds −20 % by simulation on per-query round counts of the bench
// workload, tools/sim_wave_binning.py; restarting the stragglers from scratch instead of continuing them: −8 %).
// Record of entry i of `cap`: hdr[i] = {query index, next slot, avail | slow << 16, c3n}, q[i] = query, set[j·cap + i] = {d_j, id_j},
// stack[r·cap + i] = stack row r (r < avail). Wave g of a launch owns the entries [g·stop_at, (g + 1)·stop_at) and writes how many
// it used to n[g] (SpillBuf: launch.hpp): no counter is shared, and the continuation's waves are ≈80 % full instead of packed.

template <int K, int ROWB>
__device__ __forceinline__ void walk_spill(const SpillBuf& sp, const Walk<K>& w, uint32_t gi, uint32_t col_addr, bool active, unsigned int region, int per_region) {
    // No shared counter: one atomic per wave on one address serialises (≈3.5 ns each: +1.6 ms on a 460 k-wave launch, measured).
    // Region `region` of the buffer belongs to this wave alone; it holds at most per_region (= stop_at) stragglers.
    const unsigned long long m = __ballot(active);
    const int lane = (int)__lane_id();
    if (lane == (int)__builtin_amdgcn_readfirstlane(lane)) sp.n[region] = (unsigned int)__popcll(m);  // 0 for a wave that finished: the continuation skips it
    if (active) {
        const size_t i = (size_t)region * (size_t)per_region + (size_t)__popcll(m & ((1ull << lane) - 1ull));
        sp.hdr[i] = uint4{gi, w.cur, (uint32_t)w.avail | (w.slow << 16), w.c3n};
        sp.q[i] = float4{w.qx, w.qy, w.qz, 0.f};
#pragma unroll
        for (int j = 0; j < K; ++j) sp.set[(size_t)j * sp.cap + i] = uint2{__float_as_uint(w.d[j]), w.id[j]};
        for (int r = 0; r < w.avail; ++r) {
            const u32x2 row = *reinterpret_cast<lds_u32x2*>(col_addr + (uint32_t)r * ROWB);
            sp.stack[(size_t)r * sp.cap + i] = uint2{row.x, row.y};
        }
    }
}

// LANES = active lanes per wave = stack columns (64; 16 for launches that cannot fill the chip anyway: a wave's time is its
// longest traversal, and with 128-byte rows every level fits in LDS — T = 0 — so that no query needs the deep pass).
template <int K, int DF, int MODE, int LANES = 64, bool STAMP = false>
__global__ __launch_bounds__(64) void icp_search_walk_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                             const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                             uint32_t* __restrict__ nn, size_t nn_pitch, int max_n, float alpha_eff, int T,
                                                             unsigned int tree_bytes, uint32_t dummy, int skip_nonfinite, uint32_t* __restrict__ redo_list,
                                                             unsigned int* __restrict__ redo_count, uint32_t* __restrict__ deep_list,
                                                             unsigned int* __restrict__ deep_count, unsigned long long* __restrict__ search_stats,
                                                             const int* __restrict__ active, unsigned long long* __restrict__ same_mask, int have_previous,
                                                             SpillBuf spill, int stop_at) {
    extern __shared__ uint2 s_dyn[];
    constexpr int ROWB = LANES * 8;
    static_assert(DF * ROWB >= 64 * 2 * 8, "exact stack columns do not fit");
    // The traversal reads rows BELOW the stack's bottom (they must lie outside the workgroup's LDS allocation and read as 0): the
    // stack has to start at LDS address 0, i.e. the kernel must own no other LDS. The launcher checks the code object
    // (lds_stack_starts_at_zero); this is the last line of defence — wrong neighbours or an endless loop otherwise.
    if ((uint32_t)(size_t)s_dyn != 0u) __builtin_trap();
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;  // later chunks of an alignment launch only the scans still open
    const int tid = threadIdx.x;
    // straggler hand-over: this wave's count is written by lane 0 — here for a wave that leaves at once, again in walk_spill (the
    // live lanes of a wave are a prefix of it, so lane 0 is among them)
    if (stop_at > 0 && tid == 0) spill.n[blockIdx.y * gridDim.x + blockIdx.x] = 0u;
    if (st[scan].done) return;
    const int i = blockIdx.x * LANES + tid;
    if (tid >= LANES || i >= counts[scan]) return;
    const size_t gi = (size_t)scan * max_n + i;
    const float4 p = load_once(&src[gi]);
    // pcl::isFinite, icp cpp:64 (P2P only): such a point has no neighbours; its lane stays in the wave with nothing to do (the list
    // it stores at the end is the empty one)
    const bool finite = !skip_nonfinite || (isfinite(p.x) && isfinite(p.y) && isfinite(p.z));
    if (search_stats && finite) atomicAdd(&search_stats[0], 1ull);
    const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tree, 0, (int)tree_bytes, 0x00020000);
    Walk<K> w;
    w.qx = (float)qs.x; w.qy = (float)qs.y; w.qz = (float)qs.z;
    walk_query<K, ROWB, MODE, STAMP>(rsrc, tree, w, finite, alpha_eff, T, dummy, (uint32_t)(size_t)(&s_dyn[tid]), DF, stop_at);  // DF rows: a stack that outgrows them → deep pass
    // lanes the loop left unfinished (stop_at > 0): to the continuation kernel
    bool handed_over = false;
    if (stop_at > 0) {
        handed_over = w.cur != dummy || w.avail > 0;
        walk_spill<K, ROWB>(spill, w, (uint32_t)gi, (uint32_t)(size_t)(&s_dyn[tid]), handed_over, blockIdx.y * gridDim.x + blockIdx.x, stop_at);
    }
    if (STAMP && search_stats) {
        // diagnostic build (LOCGPU_STAMP=1, MODE 12): rounds each lane needed against the rounds its wave ran — the kernel's lane
        // efficiency — and the per-query round counts behind the neighbour-list area of redo_list (2 x pitch entries in this build)
        atomicAdd(&search_stats[4], (unsigned long long)w.rounds);
        if ((tid & 63) == 0) { atomicAdd(&search_stats[9], (unsigned long long)w.wave_rounds); atomicAdd(&search_stats[12], 1ull); }
        atomicAdd(&search_stats[13], (unsigned long long)w.wave_rounds);  // rounds paid by this lane's wave, summed over lanes
        redo_list[nn_pitch + gi] = w.rounds;
    }
    const bool deep = !handed_over && w.c3n == 1u;
    const bool slow = !handed_over && !deep && w.slow != 0u;
    if (LANES == 64 && same_mask != nullptr) {
        // Plane cache: one word per wave — the lanes whose list is, index for index, the previous iteration's (the one the fit kernel's
        // cache was filled for). The old list is read HERE, at the end of the wave's life and right before it is overwritten (K
        // coalesced dwords; read at the start, the loads sat in front of every tree load in the wave's in-order memory counter: +15 %
        // search time). A query finished by the deep pass, in the wave or by the redo kernel counts as changed: it is simply refitted.
        bool same = have_previous != 0 && !deep && !slow && !handed_over;
        if (have_previous != 0) {
            uint32_t prev_id[K];
#pragma unroll
            for (int j = 0; j < K; ++j) prev_id[j] = __builtin_nontemporal_load(&nn[(size_t)j * nn_pitch + gi]);
#pragma unroll
            for (int j = 0; j < K; ++j) same = same && w.id[j] == prev_id[j];
        }
        const unsigned long long m = __ballot(same);
        if (__builtin_amdgcn_readfirstlane(tid) == tid)  // the first live lane of the wave
            same_mask[(size_t)scan * (size_t)((max_n + 63) >> 6) + blockIdx.x] = m;
    }
    if (!deep && !slow && !handed_over) {
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
template <int K, int D>
__global__ __launch_bounds__(64) void icp_search_walk_list_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                  const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                  float alpha_eff, unsigned int tree_bytes, uint32_t dummy, const uint32_t* __restrict__ list,
                                                                  const unsigned int* __restrict__ n_list, uint32_t* __restrict__ redo_list,
                                                                  unsigned int* __restrict__ redo_count, unsigned long long* __restrict__ search_stats) {
    extern __shared__ uint2 s_dyn[];
    constexpr int ROWB = 64 * 8;
    if ((uint32_t)(size_t)s_dyn != 0u) __builtin_trap();  // see icp_search_walk_kernel
    const unsigned int n = *n_list;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && search_stats) atomicAdd(&search_stats[2], (unsigned long long)n);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tree, 0, (int)tree_bytes, 0x00020000);
    // a short list is spread thin: these are the long traversals, and a wave's time is the longest among its lanes
    const unsigned int per_wave = min(64u, max(1u, (n + gridDim.x - 1) / gridDim.x));
    for (unsigned int r0 = blockIdx.x * per_wave; r0 < n; r0 += gridDim.x * per_wave) {
        const unsigned int r = r0 + (unsigned int)tid;
        const bool valid = (unsigned int)tid < per_wave && r < n;
        const uint32_t gi = valid ? list[r] : 0u;
        Walk<K> w;
        w.qx = w.qy = w.qz = 0.f;
        if (valid) {
            const int scan = (int)(gi / (uint32_t)max_n);
            const float4 p = src[gi];
            const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
            w.qx = (float)qs.x; w.qy = (float)qs.y; w.qz = (float)qs.z;
        }
        walk_query<K, ROWB, 12>(rsrc, tree, w, valid, alpha_eff, 0, dummy, (uint32_t)(size_t)(&s_dyn[tid]), 0x7fffffff);  // D + 2 rows ≥ depth: cannot overflow
        const bool slow = valid && w.slow != 0u;
        if (valid && !slow) {
#pragma unroll
            for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = w.id[j];
        }
        if (!walk_exact_in_wave<K>(tree, slow, w.qx, w.qy, w.qz, alpha_eff, s_dyn, nn, nn_pitch, gi, search_stats))
            wave_append(redo_list, redo_count, slow, gi);
    }
}

// Continuation of the walk kernel's stragglers (see SpillBuf): one-wave workgroups, grid-stride over the spill records, 64 to a wave;
// every lane restores its query's registers and stack rows and goes on with the same loop to the end. Finished queries are handled
// exactly as at the end of the walk kernel (list stored | deep pass | exact traversal for ties).
template <int K, int DF, bool STAMP = false>
__global__ __launch_bounds__(64) void icp_search_walk_cont_kernel(const uint2* __restrict__ tree, uint32_t* __restrict__ nn, size_t nn_pitch, float alpha_eff,
                                                                  unsigned int tree_bytes, uint32_t dummy, SpillBuf spill, unsigned int n_regions, int per_region,
                                                                  uint32_t* __restrict__ redo_list, unsigned int* __restrict__ redo_count,
                                                                  uint32_t* __restrict__ deep_list, unsigned int* __restrict__ deep_count,
                                                                  unsigned long long* __restrict__ search_stats) {
    extern __shared__ uint2 s_dyn[];
    constexpr int ROWB = 64 * 8;
    if ((uint32_t)(size_t)s_dyn != 0u) __builtin_trap();  // see icp_search_walk_kernel
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tree, 0, (int)tree_bytes, 0x00020000);
    const uint32_t col_addr = (uint32_t)(size_t)(&s_dyn[tid]);
    // entry i of the buffer: region i / per_region (one walk-kernel wave), its (i % per_region)-th straggler
    const size_t i = (size_t)blockIdx.x * 64u + (size_t)tid;
    const unsigned int region = (unsigned int)(i / (unsigned int)per_region);
    const bool valid = region < n_regions && (unsigned int)(i % (unsigned int)per_region) < spill.n[region];
    if (__ballot(valid) == 0ull) return;
    Walk<K> w;
    uint32_t gi = 0;
    w.qx = w.qy = w.qz = 0.f;
    w.cur = dummy; w.avail = 0; w.c3n = 0; w.slow = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) { w.d[j] = __builtin_inff(); w.id[j] = kInvalidSlot; }
    if (valid) {
        const uint4 h = spill.hdr[i];
        const float4 q = spill.q[i];
        gi = h.x; w.cur = h.y; w.avail = (int)(h.z & 0xFFFFu); w.slow = h.z >> 16; w.c3n = h.w;
        w.qx = q.x; w.qy = q.y; w.qz = q.z;
#pragma unroll
        for (int j = 0; j < K; ++j) { const uint2 e = spill.set[(size_t)j * spill.cap + i]; w.d[j] = as_f32(e.x); w.id[j] = e.y; }
        for (int r = 0; r < w.avail; ++r) {
            const uint2 row = spill.stack[(size_t)r * spill.cap + i];
            *reinterpret_cast<lds_u32x2*>(col_addr + (uint32_t)r * ROWB) = u32x2{row.x, row.y};
        }
    }
    walk_rounds_capped<K, ROWB, 2, STAMP>(rsrc, w, alpha_eff, dummy, col_addr, DF, 0);
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) w.slow |= w.d[j] == w.d[j + 1] ? 1u : 0u;
    if (STAMP && search_stats && valid) {
        atomicAdd(&search_stats[4], (unsigned long long)w.rounds);
        atomicAdd(&search_stats[13], (unsigned long long)w.wave_rounds);
    }
    const bool deep = valid && w.c3n == 1u;
    const bool slow = valid && !deep && w.slow != 0u;
    if (valid && !deep && !slow) {
#pragma unroll
        for (int j = 0; j < K; ++j) __builtin_nontemporal_store(w.id[j], &nn[(size_t)j * nn_pitch + gi]);
    }
    wave_append(deep_list, deep_count, deep, gi);
    if (!walk_exact_in_wave<K>(tree, slow, w.qx, w.qy, w.qz, alpha_eff, s_dyn, nn, nn_pitch, gi, search_stats))
        wave_append(redo_list, redo_count, slow, gi);
}

// The fast traversal over a LIST of queries (grid mode: what the tile kernel could not settle, with alpha_eff = 1). One-wave
// workgroups, grid-stride over the list whose length lives on the device. Queries it cannot finish go to redo_list as usual.
template <int K, int DF>
__global__ __launch_bounds__(64) void icp_search_fast_list_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                  const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                  float alpha_eff, int T, unsigned int tree_bytes, const uint32_t* __restrict__ list,
                                                                  const unsigned int* __restrict__ n_list, uint32_t* __restrict__ redo_list,
                                                                  unsigned int* __restrict__ redo_count, unsigned long long* __restrict__ search_stats) {
    __shared__ uint2 s_stack[DF][64];
    const unsigned int n = *n_list;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && search_stats) atomicAdd(&search_stats[2], (unsigned long long)n);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tree, 0, (int)tree_bytes, 0x00020000);
    for (unsigned int r0 = blockIdx.x * 64u; r0 < n; r0 += gridDim.x * 64u) {
        const unsigned int r = r0 + (unsigned int)tid;
        bool slow = false;
        uint32_t gi = 0;
        if (r < n) {
            gi = list[r];
            const int scan = (int)(gi / (uint32_t)max_n);
            const float4 p = src[gi];
            const D3 qs = se3_apply(st[scan].q, st[scan].t, D3{(double)p.x, (double)p.y, (double)p.z});
            const float fqx = (float)qs.x, fqy = (float)qs.y, fqz = (float)qs.z;
            const bool sane = fabsf(fqx) < 1e18f && fabsf(fqy) < 1e18f && fabsf(fqz) < 1e18f;
            SortedSet<K> set;
            slow = !sane || tree_knn_fast<K, DF, 64>(rsrc, fqx, fqy, fqz, alpha_eff, T, s_stack, tid, set);
            if (!slow) {
#pragma unroll
                for (int j = 0; j < K; ++j) nn[(size_t)j * nn_pitch + gi] = set.id[j];
            }
        }
        wave_append(redo_list, redo_count, slow, gi);
    }
}

// Exact recomputation of the queries the fast kernel could not finish (≈1e-6 of them on real maps: ≈30 per 256-scan launch).
// One-wave workgroups over the list. A short list is dealt one query per WAVE (lane 0): 30 queries sharing a wave would each
// pay the others' heap-emulation branches and the longest traversal; alone in its wave a query costs its own ≈40 dependent loads.
// A long list (a lattice map: every distance ties) fills all 64 lanes of every wave.
constexpr int kRedoWaves = 2048;
template <int KMAX, int D>
__global__ __launch_bounds__(64) void icp_search_redo_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                             const PoseState* __restrict__ st, uint32_t* __restrict__ nn, size_t nn_pitch,
                                                             int max_n, int k, float alpha_eff, const uint32_t* __restrict__ redo_list,
                                                             const unsigned int* __restrict__ redo_count, unsigned long long* __restrict__ search_stats) {
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
        const float4 p = src[gi];
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
                                                                 const int* __restrict__ active) {
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
            const float4 p = src[gi];
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
                        for (int c = 0; c < 3; ++c) nR[c] = (-n3.x * R[c] + -n3.y * R[3 + c]) + -n3.z * R[6 + c];
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

// K2 with the PLANE CACHE (round 4; OPT-IN with LOCGPU_PLANE_CACHE=1 — built, bit-identical, and a net LOSS on the
// bench workload: per 256-scan step the fit kernel gains 0.15 ms (9.20 → 9.05: the fit is two thirds of the kernel, 30 % of the
// point-iterations keep their list, and the compaction's own cost — barriers, scattered list loads, 32 B of cache traffic per point —
// eats most of that), while the search kernel pays 2.0-2.4 ms for reading the previous lists back at the end of every wave
// (profiles/experiments.md). Kept for the measurement, not selected by default.)
// FitPlane's 4-vector depends only on the five
// neighbour indices, and from the second iteration on a growing share of the points keeps its list (5 % / 13 % / 22 % / 36 % / 49 % / 60 % /
// 69 % / 75 % in iterations 1…8 of the bench workload) — never a whole wave of them, so skipping the fit per lane saves nothing. Here
// a block compacts the points that need a fit: the search kernel has left one bit per query ("same five indices as last time", same_mask),
//   1. every thread looks up the bits of its `pts` points; the ones to (re)fit are appended to a list in LDS;
//   2. the block's threads walk that list — full waves except the last: gather the five leaves, fit, residual check of the five
//      (math_utils.h:112-136), and leave the 4-vector in the per-point cache in HBM (all zero = "no plane": k > size_, or the check failed;
//      a fitted vector has unit length);
//   3. every thread takes its points' vectors — the fresh ones from LDS, the kept ones from the cache — forms residual and Jacobian
//      (icp cpp:184-201) and the rows are summed exactly as in icp_plane_accum_kernel — same order, same bits.
// A cached vector is the one the same code computed from the same five leaves: results equal the uncached kernel's bit for bit.
constexpr int kPlaneCachePts = 2;  // points per thread: 512 plane vectors (16 KB) stay in LDS between the fit and the residual stage
template <int FIT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void icp_plane_cached_accum_kernel(const uint2* __restrict__ tree, const float4* __restrict__ src,
                                                                 const int* __restrict__ counts, const PoseState* __restrict__ st,
                                                                 const uint32_t* __restrict__ nn, size_t nn_pitch, int max_n,
                                                                 double max_plane_distance, double* __restrict__ partials, int kPlanePts,
                                                                 const int* __restrict__ active, double* __restrict__ plane_cache,
                                                                 const unsigned long long* __restrict__ same_mask, int use_cache) {
    __shared__ double s_row[8][kAccPad];
    __shared__ double s_slice[kBlock / 32][32];
    __shared__ double s_n4[kPlaneCachePts * kBlock][4];  // the vectors fitted by this block, by point (a refit point is read back from here, not from HBM)
    __shared__ unsigned short s_todo[kPlaneCachePts * kBlock];
    __shared__ int s_ntodo;
    const int scan = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (st[scan].done) return;  // uniform per block
    const int tid = threadIdx.x;
    const int n_pts = counts[scan];
    const int base_i = blockIdx.x * kPlanePts * kBlock;
    const size_t scan_off = (size_t)scan * max_n;
    if (tid == 0) s_ntodo = 0;
    __syncthreads();
    // ---- 1. which of the block's points need a fit
    unsigned int refit_bits = 0;  // bit pp: this thread's pp-th point is (re)fitted by the block in stage 2
    const size_t mask_row = (size_t)scan * (size_t)((max_n + 63) >> 6);
#pragma unroll 1
    for (int pp = 0; pp < kPlanePts; ++pp) {
        const int off = pp * kBlock + tid;
        const int i = base_i + off;
        bool todo = i < n_pts;
        if (todo && use_cache) todo = ((same_mask[mask_row + (size_t)(i >> 6)] >> (i & 63)) & 1ull) == 0ull;
        refit_bits |= todo ? (1u << pp) : 0u;
        const unsigned long long m = __ballot(todo);
        if (m != 0ull) {
            const int lane = tid & 63;
            const int leader = __ffsll((long long)m) - 1;
            int b0 = 0;
            if (lane == leader) b0 = atomicAdd(&s_ntodo, __popcll(m));
            b0 = __shfl(b0, leader, 64);
            if (todo) s_todo[b0 + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)off;
        }
    }
    __syncthreads();
    // ---- 2. fit them (which thread fits which point varies from run to run; what is written for a point does not)
    const int n_todo = s_ntodo;
#pragma unroll 1
    for (int t = tid; t < n_todo; t += kBlock) {
        const int off = (int)s_todo[t];
        const size_t gi = scan_off + (size_t)(base_i + off);
        uint32_t slot[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) slot[j] = __builtin_nontemporal_load(&nn[(size_t)j * nn_pitch + gi]);
        double n4[4] = {0.0, 0.0, 0.0, 0.0};
        if (slot[4] != kInvalidSlot) {  // nn.size() > 3: k=5 yields 5 or (k > size_) none
            D3 nb[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) nb[j] = leaf_point(tree, slot[j]);
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
            if (!fit) { n4[0] = 0.0; n4[1] = 0.0; n4[2] = 0.0; n4[3] = 0.0; }
        }
        double* c = plane_cache + 4 * gi;
        c[0] = n4[0]; c[1] = n4[1]; c[2] = n4[2]; c[3] = n4[3];
        s_n4[off][0] = n4[0]; s_n4[off][1] = n4[1]; s_n4[off][2] = n4[2]; s_n4[off][3] = n4[3];
    }
    __syncthreads();
    // ---- 3. residual, Jacobian, sums — as icp_plane_accum_kernel
    RowAccum ra;
    ra.init();
#pragma unroll 1
    for (int pp = 0; pp < kPlanePts; ++pp) {
        const int i = base_i + pp * kBlock + tid;
        double J[1][6] = {{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
        double neg_e[1] = {0.0};
        double fitted = 0.0;
        if (i < n_pts) {
            const size_t gi = scan_off + (size_t)i;
            double n40, n41, n42, n43;
            if ((refit_bits >> pp) & 1u) {  // fitted a moment ago by this block: from LDS
                const double* c = &s_n4[pp * kBlock + tid][0];
                n40 = c[0]; n41 = c[1]; n42 = c[2]; n43 = c[3];
            } else {                        // kept from an earlier iteration: from the cache
                const double* c = plane_cache + 4 * gi;
                n40 = c[0]; n41 = c[1]; n42 = c[2]; n43 = c[3];
            }
            const float4 p = src[gi];
            if (!(n40 == 0.0 && n41 == 0.0 && n42 == 0.0 && n43 == 0.0)) {
                fitted = 1.0;  // effective_num++ before the residual gate (icp cpp:184)
                const D3 n3{n40, n41, n42};
                const D3 q{(double)p.x, (double)p.y, (double)p.z};
                const D3 qs = se3_apply(st[scan].q, st[scan].t, q);
                const double dis = dot3(n3, qs) + n43;
                if (!(fabs(dis) > max_plane_distance)) {
                    const double* R = st[scan].R;
                    double nR[3];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) nR[cc] = (-n3.x * R[cc] + -n3.y * R[3 + cc]) + -n3.z * R[6 + cc];
                    J[0][0] = nR[1] * q.z - nR[2] * q.y;
                    J[0][1] = nR[2] * q.x - nR[0] * q.z;
                    J[0][2] = nR[0] * q.y - nR[1] * q.x;
                    J[0][3] = n3.x; J[0][4] = n3.y; J[0][5] = n3.z;
                    neg_e[0] = -dis;
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
                                                                 const int* __restrict__ active) {
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
            const float4 p = src[gi];
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
                                                                const int* __restrict__ active) {
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
        const float4 p = src[gi];
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
// order). Eight loads are in flight per thread: a single-scan alignment has 450 rows, and a row-by-row load → add chain made this the
// longest part of the solve kernel. Returns the column total in threads 0..kAccW-1 (0 elsewhere). Ends with the block synchronised.
__device__ __forceinline__ double reduce_partials(const double* __restrict__ rows, int blocks_per_scan, bool mine, double (*s_sum)[kAccW]) {
    const int col = threadIdx.x & (kAccW - 1), chunk = threadIdx.x / kAccW;
    constexpr int kChunks = kBlock / kAccW;
    double s = 0.0;
    if (mine && col < 28) {
        for (int b = chunk; b < blocks_per_scan; b += kChunks * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = b + u * kChunks;
                v[u] = idx < blocks_per_scan ? rows[(size_t)idx * kAccW + col] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
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
__global__ __launch_bounds__(kBlock) void gn_solve_kernel(const double* __restrict__ partials, int blocks_per_scan, PoseState* __restrict__ st,
                                                          GnParams prm, int do_update, double* __restrict__ hb_out, unsigned int* __restrict__ list_counts) {
    __shared__ double s_sum[kBlock / kAccW][kAccW];
    __shared__ double s_lu[36 + 6];
    const int scan = blockIdx.x;
    // The search stage's work-list counters (fast kernel → redo kernel) are consumed by now: zero them for the next iteration's
    // search instead of paying two fill launches per iteration (a single-scan alignment is launch-latency bound).
    if (list_counts && scan == 0 && threadIdx.x < 4) list_counts[threadIdx.x] = 0u;
    if (st[scan].done) return;
    const double col_total = reduce_partials(partials + (size_t)scan * blocks_per_scan * kAccW, blocks_per_scan, true, s_sum);
    if (threadIdx.x < kAccW) s_sum[0][threadIdx.x] = col_total;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot[28];
    for (int v = 0; v < 28; ++v) tot[v] = s_sum[0][v];
    double H[36], B[6], dx[6] = {0, 0, 0, 0, 0, 0};
    int o = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { H[6 * i + j] = tot[o]; H[6 * j + i] = tot[o]; ++o; }
    for (int i = 0; i < 6; ++i) B[i] = tot[21 + i];
    const long long eff = (long long)tot[27];
    PoseState& ps = st[scan];
    bool ok;
    const double det = lu6_det_solve(H, B, dx, s_lu);  // LU workspace in LDS: its pivoting indexes rows dynamically
    if (prm.method == 3) {
        // direct NDT: det(H)==0 is tested FIRST and aborts the whole alignment (ndt cpp:435-436)
        if (det == 0.0) {
            ps.status = 1; ps.done = 1; ps.iterations += 1; ps.last_eff = eff;
            if (hb_out) { for (int i = 0; i < 36; ++i) hb_out[44 * scan + i] = H[i]; for (int i = 0; i < 6; ++i) hb_out[44 * scan + 36 + i] = B[i]; hb_out[44 * scan + 42] = (double)eff; hb_out[44 * scan + 43] = 0.0; }
            return;
        }
        ok = eff >= prm.min_effective_pts;
    } else if (prm.method == 4) {
        // incremental NDT: too few accepted residuals ⇒ `result_pose = pose; return false` (ndt cpp:349-353); no det(H) test
        ok = eff >= prm.min_effective_pts;
        if (!ok) {
            ps.status = 2; ps.done = 1; ps.iterations += 1; ps.last_eff = eff;
            if (hb_out) { for (int i = 0; i < 36; ++i) hb_out[44 * scan + i] = H[i]; for (int i = 0; i < 6; ++i) hb_out[44 * scan + 36 + i] = B[i]; hb_out[44 * scan + 42] = (double)eff; hb_out[44 * scan + 43] = 0.0; }
            return;
        }
    } else {
        ok = (eff >= prm.min_effective_pts) && !(det == 0.0);
    }
    if (hb_out) {
        for (int i = 0; i < 36; ++i) hb_out[44 * scan + i] = H[i];
        for (int i = 0; i < 6; ++i) hb_out[44 * scan + 36 + i] = B[i];
        hb_out[44 * scan + 42] = (double)eff;
        hb_out[44 * scan + 43] = ok ? 1.0 : 0.0;
    }
    ps.last_eff = eff;
    if (!do_update) return;
    ps.iterations += 1;
    if (ok) {
        if (prm.method == 0)
            for (int i = 0; i < 6; ++i) dx[i] = dx[i] / 16;  // dx = H.inverse()/16 * err (icp cpp:287)
        se3_apply_update(ps.q, ps.t, dx);
        quat_to_R(ps.q, ps.R);
        double n2 = 0.0;
        for (int i = 0; i < 6; ++i) n2 += dx[i] * dx[i];
        const double nrm = sqrt(n2);
        ps.last_dx_norm = nrm;
        if (nrm < prm.eps) { ps.converged = 1; ps.done = 1; }
    }
    if (ps.iterations >= prm.max_iteration) ps.done = 1;
}

// First half of gn_solve_kernel for sharded batches (see launch.hpp): one block per GLOBAL scan.
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double* __restrict__ partials, int blocks_per_scan, const PoseState* __restrict__ st_all,
                                                              int first, int n_local, double* __restrict__ acc) {
    __shared__ double s_sum[kBlock / kAccW][kAccW];
    const int g = blockIdx.x;
    const int scan = g - first;
    const bool mine = scan >= 0 && scan < n_local && !st_all[g].done;  // a finished scan's partials are stale: contribute zeros (nobody reads them)
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
__global__ __launch_bounds__(kBlock) void transform_cloud_kernel(const float4* __restrict__ src, size_t n, const float* __restrict__ m12,
                                                                 float4* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    float4 o;
    o.x = ((m12[0] * p.x + m12[1] * p.y) + m12[2] * p.z) + m12[3];
    o.y = ((m12[4] * p.x + m12[5] * p.y) + m12[6] * p.z) + m12[7];
    o.z = ((m12[8] * p.x + m12[9] * p.y) + m12[10] * p.z) + m12[11];
    o.w = p.w;
    dst[i] = o;
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
    dim3 grid((a.max_n + kBlock - 1) / kBlock, a.n_scans);
    if (a.visit_totals)
        hipLaunchKernelGGL((icp_search_kernel<KMAX, D, true>), grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.k, a.alpha_eff, a.skip_nonfinite, a.visit_totals, a.touched);
    else
        hipLaunchKernelGGL((icp_search_kernel<KMAX, D, false>), grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.k, a.alpha_eff, a.skip_nonfinite, a.visit_totals, nullptr);
}
// EXPERIMENT hook (timing only; variants 2-4 truncate the stack and give wrong neighbours): LOCGPU_SEARCH_VARIANT
template <int KMAX, int D, int V>
static void launch_search_exp(const SearchArgs& a, hipStream_t s) {
    dim3 grid((a.max_n + kBlock - 1) / kBlock, a.n_scans);
    hipLaunchKernelGGL((icp_search_kernel<KMAX, D, false, V>), grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                       a.k, a.alpha_eff, a.skip_nonfinite, a.visit_totals, nullptr);
}
static int search_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LOCGPU_SEARCH_VARIANT"); v = e ? atoi(e) : 0; }
    return v;
}

template <int KMAX>
static bool launch_search_k(const SearchArgs& a, hipStream_t s) {
    if (KMAX == 5 && !a.visit_totals && a.depth <= 32) {
        switch (search_variant()) {
            case 1: launch_search_exp<5, 32, 1>(a, s); return true;
            case 9: launch_search_exp<5, 32, 9>(a, s); return true;
            case 2: launch_search_exp<5, 16, 1>(a, s); return true;
            case 3: launch_search_exp<5, 8, 1>(a, s); return true;
            case 4: launch_search_exp<5, 16, 0>(a, s); return true;
            case 5: launch_search_exp<5, 8, 0>(a, s); return true;
            default: break;
        }
    }
    if (a.depth <= 32) launch_search_kd<KMAX, 32>(a, s);
    else if (a.depth <= 40) launch_search_kd<KMAX, 40>(a, s);
    else if (a.depth <= 64) launch_search_kd<KMAX, 64>(a, s);
    else return false;
    return true;
}
// Stored LDS stack entries per thread in the fast kernel (8 B each; the levels above them are not stored, see tree_knn_fast).
// One-wave workgroups (waves retire independently), so DF sets the waves a CU holds: 15 → 7.5 KB → 21 waves per CU. Since a single
// un-stored entry that can still pass is expanded without a replay, the un-stored levels are cheap and the balance is between
// occupancy and the (rare) paths deeper than DF that go to the exact traversal — measured on the bench workload (search ms per
// 256-scan step): 12 → 30.0, 13 → 28.2, 14 → 26.2, 15 → 25.5, 16 → 27.2, 20 → 29.0, 24 → 35.9.
// LOCGPU_FAST_STACK=12|24 selects other depths (parity tests); LOCGPU_FAST_BLOCK=128|256 other block shapes of the round-2 kernel.
static int fast_stack_depth() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LOCGPU_FAST_STACK");
        v = e ? atoi(e) : 15;
        if (v != 12 && v != 24) v = 15;  // 12 and 24 exist for the parity tests at other stack depths
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
template <int K, int D, int DF>
static bool walk_kernels_ok_kdf() {
    return no_static_lds<icp_search_walk_kernel<K, DF, 0>>() && no_static_lds<icp_search_walk_kernel<K, DF, 2>>() && no_static_lds<icp_search_walk_kernel<K, DF, 12>>() &&
           no_static_lds<icp_search_walk_cont_kernel<K, DF>>();
}
template <int K, int D>
static bool walk_kernels_ok_kd() {
    return walk_kernels_ok_kdf<K, D, 12>() && walk_kernels_ok_kdf<K, D, 15>() && walk_kernels_ok_kdf<K, D, 24>() &&
           no_static_lds<icp_search_walk_kernel<K, D + 2, 12, 16>>() && no_static_lds<icp_search_walk_list_kernel<K, D>>();
}
bool search_kernels_lds_ok() {
    static const bool ok = walk_kernels_ok_kd<1, 32>() && walk_kernels_ok_kd<1, 40>() && walk_kernels_ok_kd<1, 64>() &&
                           walk_kernels_ok_kd<5, 32>() && walk_kernels_ok_kd<5, 40>() && walk_kernels_ok_kd<5, 64>();
    return ok;
}

template <int K, int D, int DF>
static bool launch_fast_kd(const SearchArgs& a, hipStream_t s) {
    const int T = a.depth > DF ? a.depth - DF : 0;  // leading stack positions the fast kernel does not store
    // a.redo_count is zero here: the caller clears it before an alignment's first iteration, gn_solve_kernel after every search
    static const bool stamp = [] { const char* e = getenv("LOCGPU_STAMP"); return e && atoi(e) != 0; }();
    if (stamp && a.redo_list2) {
        // diagnostic build of the default shape (capped rounds, 64 lanes, DF rows) that counts rounds per lane and per wave:
        // results unchanged, timing meaningless. search_stats[4] = Σ lane rounds, [13] = Σ over lanes of their wave's rounds,
        // [9] = Σ wave rounds, [12] = waves; per-query rounds at redo_list[pitch + gi] (locgpu_debug_stamp_trips).
        const uint32_t dummy = (uint32_t)(a.tree_bytes / 8);
        const int Tw = a.depth > DF - 2 ? a.depth - (DF - 2) : 0;
        const int n_launch = a.active ? a.n_active : a.n_scans;
        dim3 g2((a.max_n + 63) / 64, n_launch);
        static const int stop_env_s = walk_stop_lanes();
        const int stop_at_s = (stop_env_s > 0 && stop_env_s < 32 && a.spill.hdr != nullptr && (size_t)g2.x * g2.y >= walk_stop_min_waves() &&
                               (size_t)g2.x * g2.y * stop_env_s <= a.spill.cap) ? stop_env_s : 0;  // as the shipped launch below
        hipLaunchKernelGGL((icp_search_walk_kernel<K, DF, 12, 64, true>), g2, dim3(64), DF * 64 * 8, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.alpha_eff, Tw, (unsigned int)a.tree_bytes + 16u, dummy, a.skip_nonfinite, a.redo_list, a.redo_count, a.redo_list2, a.redo_count2,
                           a.search_stats, a.active, a.same_mask, a.have_previous, a.spill, stop_at_s);
        if (stop_at_s > 0) {
            const unsigned int n_regions = g2.x * g2.y;
            hipLaunchKernelGGL((icp_search_walk_cont_kernel<K, DF, true>), dim3((unsigned int)(((size_t)n_regions * stop_at_s + 63) / 64)), dim3(64), DF * 64 * 8, s, a.tree, a.nn, a.nn_pitch,
                               a.alpha_eff, (unsigned int)a.tree_bytes + 16u, dummy, a.spill, n_regions, stop_at_s, a.redo_list, a.redo_count, a.redo_list2, a.redo_count2, a.search_stats);
        }
        hipLaunchKernelGGL((icp_search_walk_list_kernel<K, D>), dim3(2048), dim3(64), (D + 2) * 64 * 8, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.alpha_eff, (unsigned int)a.tree_bytes + 16u, dummy, a.redo_list2, a.redo_count2, a.redo_list, a.redo_count, a.search_stats);
        hipLaunchKernelGGL((icp_search_redo_kernel<K, D>), dim3(kRedoWaves), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.k,
                           a.alpha_eff, a.redo_list, a.redo_count, a.search_stats);
        return true;
    }
    static const int blk = [] { const char* e = getenv("LOCGPU_FAST_BLOCK"); const int v = e ? atoi(e) : 64; return (v == 256 || v == 128) ? v : 64; }();
    static const int walk = [] { const char* e = getenv("LOCGPU_WALK"); return e ? atoi(e) : 1; }();  // 0 = the round-2 kernel (A/B runs)
    if (walk == 1 && a.redo_list2) {
        // round-3 traversal (search_walk.hpp): rows 0/1 of the DF stored rows hold the candidates of the un-stored levels, the other
        // DF-2 rows one level each. Queries whose un-stored levels need more than two candidates, or whose candidate descent outgrows
        // the rows, go through a.redo_list2 to the deep pass (every level stored), ties to the exact redo kernel as before.
        const uint32_t dummy = (uint32_t)(a.tree_bytes / 8);
        static const int wpad = [] { const char* e = getenv("LOCGPU_LDS_PAD"); return e ? atoi(e) : 0; }();  // experiment: extra dynamic LDS lowers occupancy
        static const int mode = [] { const char* e = getenv("LOCGPU_WALK_MODE"); return e ? atoi(e) : 12; }();
        static const int small = [] { const char* e = getenv("LOCGPU_SMALL_LANES"); return e ? atoi(e) : 16; }();
        const int n_launch = a.active ? a.n_active : a.n_scans;  // grid.y: the scans still open (a.active) or all of them
        if ((size_t)((a.max_n + 63) / 64) * n_launch <= 2048 && small == 16) {
            // fewer than 2048 full waves (one or two scans): quarter-filled waves with every level stored — no deep pass
            dim3 g1((a.max_n + 15) / 16, n_launch);
            hipLaunchKernelGGL((icp_search_walk_kernel<K, D + 2, 12, 16>), g1, dim3(64), (D + 2) * 16 * 8, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                               a.alpha_eff, 0, (unsigned int)a.tree_bytes + 16u, dummy, a.skip_nonfinite, a.redo_list, a.redo_count, a.redo_list2, a.redo_count2,
                               a.search_stats, a.active, (unsigned long long*)nullptr, 0, SpillBuf{}, 0);
            return true;  // no redo launch: the 16-lane kernel answers its ties itself
        }
        const int Tw = a.depth > DF - 2 ? a.depth - (DF - 2) : 0;
        dim3 g2((a.max_n + 63) / 64, n_launch);
        // straggler hand-over (SpillBuf): only where the launch is long enough for one more kernel behind it to pay
        static const int stop_env = walk_stop_lanes();  // lanes left when a wave stops; 0 = off (the default)
        const bool hand_over = mode == 12 && stop_env > 0 && stop_env < 32 && a.spill.hdr != nullptr && (size_t)g2.x * g2.y >= walk_stop_min_waves() &&
                               (size_t)g2.x * g2.y * stop_env <= a.spill.cap;
        const int stop_at = hand_over ? stop_env : 0;
#define LOCGPU_WALK_LAUNCH(M) hipLaunchKernelGGL((icp_search_walk_kernel<K, DF, M>), g2, dim3(64), DF * 64 * 8 + wpad, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, \
                                                 a.max_n, a.alpha_eff, Tw, (unsigned int)a.tree_bytes + 16u, dummy, a.skip_nonfinite, a.redo_list, a.redo_count,       \
                                                 a.redo_list2, a.redo_count2, a.search_stats, a.active, a.same_mask, a.have_previous, a.spill, stop_at)
        if (mode == 2) LOCGPU_WALK_LAUNCH(2);
        else if (mode == 12) LOCGPU_WALK_LAUNCH(12);
        else LOCGPU_WALK_LAUNCH(0);
#undef LOCGPU_WALK_LAUNCH
        if (hand_over) {
            const unsigned int n_regions = g2.x * g2.y;
            const unsigned int cont_waves = (unsigned int)(((size_t)n_regions * stop_at + 63) / 64);
            hipLaunchKernelGGL((icp_search_walk_cont_kernel<K, DF>), dim3(cont_waves), dim3(64), DF * 64 * 8, s, a.tree, a.nn, a.nn_pitch, a.alpha_eff, (unsigned int)a.tree_bytes + 16u,
                               dummy, a.spill, n_regions, stop_at, a.redo_list, a.redo_count, a.redo_list2, a.redo_count2, a.search_stats);
        }
        // 2048 one-wave blocks of 17 KB LDS are all resident at once (nine fit a CU): the first iteration's ≈140 k deep queries take one
        // traversal per wave instead of two or three in sequence (search 18.07 → 17.80 ms per 256-scan step; 4608: 17.96)
        static const int deep_grid = [] { const char* e = getenv("LOCGPU_DEEP_GRID"); return e ? atoi(e) : 2048; }();
        hipLaunchKernelGGL((icp_search_walk_list_kernel<K, D>), dim3(deep_grid), dim3(64), (D + 2) * 64 * 8, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n,
                           a.alpha_eff, (unsigned int)a.tree_bytes + 16u, dummy, a.redo_list2, a.redo_count2, a.redo_list, a.redo_count, a.search_stats);
        hipLaunchKernelGGL((icp_search_redo_kernel<K, D>), dim3(kRedoWaves), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.k,
                           a.alpha_eff, a.redo_list, a.redo_count, a.search_stats);
        return true;
    }
    if (blk != 256) {
        static const int lds_pad = [] { const char* e = getenv("LOCGPU_LDS_PAD"); return e ? atoi(e) : 0; }();  // experiment: extra dynamic LDS lowers occupancy
        static const int small_lanes = [] { const char* e = getenv("LOCGPU_SMALL_LANES"); const int v = e ? atoi(e) : 16; return (v == 16 || v == 32) ? v : 64; }();
        // fewer than 2048 full waves (one or two scans): half-filled waves, see the kernel
        const int lanes = (blk == 64 && (size_t)((a.max_n + 63) / 64) * a.n_scans <= 2048) ? small_lanes : blk;
        dim3 g2((a.max_n + lanes - 1) / lanes, a.n_scans);
        if (blk == 64)
            hipLaunchKernelGGL((icp_search_fast_kernel<K, DF, 64>), g2, dim3(64), lds_pad, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                               a.alpha_eff, T, (unsigned int)a.tree_bytes, a.skip_nonfinite, a.redo_list, a.redo_count, a.search_stats, lanes);
        else
            hipLaunchKernelGGL((icp_search_fast_kernel<K, DF, 128>), g2, dim3(128), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                               a.alpha_eff, T, (unsigned int)a.tree_bytes, a.skip_nonfinite, a.redo_list, a.redo_count, a.search_stats, 128);
        hipLaunchKernelGGL((icp_search_redo_kernel<K, D>), dim3(kRedoWaves), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.k,
                           a.alpha_eff, a.redo_list, a.redo_count, a.search_stats);
        return true;
    }
    dim3 grid((a.max_n + kBlock - 1) / kBlock, a.n_scans);
    hipLaunchKernelGGL((icp_search_fast_kernel<K, DF, kBlock>), grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n,
                       a.alpha_eff, T, (unsigned int)a.tree_bytes, a.skip_nonfinite, a.redo_list, a.redo_count, a.search_stats, kBlock);
    hipLaunchKernelGGL((icp_search_redo_kernel<K, D>), dim3(kRedoWaves), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.k,
                       a.alpha_eff, a.redo_list, a.redo_count, a.search_stats);
    return true;
}
template <int K, int D>
static bool launch_fast_d(const SearchArgs& a, hipStream_t s) {
    switch (fast_stack_depth()) {
        case 12: return launch_fast_kd<K, D, 12>(a, s);
        case 24: return launch_fast_kd<K, D, 24>(a, s);
        default: return launch_fast_kd<K, D, 15>(a, s);
    }
}
template <int K>
static bool launch_fast_k(const SearchArgs& a, hipStream_t s) {
    if (a.depth <= 32) return launch_fast_d<K, 32>(a, s);
    if (a.depth <= 40) return launch_fast_d<K, 40>(a, s);
    if (a.depth <= 64) return launch_fast_d<K, 64>(a, s);
    return false;
}

template <int K>
static bool launch_redo_k(const SearchArgs& a, hipStream_t s) {
#define LOCGPU_REDO(D) hipLaunchKernelGGL((icp_search_redo_kernel<K, D>), dim3(kRedoWaves), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.k, \
                                          a.alpha_eff, a.redo_list, a.redo_count, a.search_stats)
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

// Fast traversal over `list` (length *n_list on the device), then the exact redo kernel for what it hands on. a.redo_count must
// have been zeroed on the stream; a.alpha_eff is the pruning factor (1 = exact).
template <int K>
static bool launch_fast_list_k(const SearchArgs& a, const uint32_t* list, const unsigned int* n_list, hipStream_t s) {
    constexpr int DF = 15;
    const int T = a.depth > DF ? a.depth - DF : 0;
    hipLaunchKernelGGL((icp_search_fast_list_kernel<K, DF>), dim3(4096), dim3(64), 0, s, a.tree, a.src, a.st, a.nn, a.nn_pitch, a.max_n, a.alpha_eff, T,
                       (unsigned int)a.tree_bytes, list, n_list, a.redo_list, a.redo_count, a.search_stats);
    return launch_redo_k<K>(a, s);
}
bool launch_icp_search_list(const SearchArgs& a, const uint32_t* list, const unsigned int* n_list, hipStream_t s) {
    if (a.depth > 64) return false;
    if (a.k == 1) return launch_fast_list_k<1>(a, list, n_list, s);
    if (a.k == 5) return launch_fast_list_k<5>(a, list, n_list, s);
    return false;
}

// LOCGPU_WALK_STOP = lanes a wave of the 64-lane search kernel leaves to the continuation kernel (straggler hand-over, see SpillBuf).
// DEFAULT 0 = off: built for round 4 and measured a net LOSS on the bench workload — the walk kernel gains nothing (14.3 vs 14.5 ms
// per 256-scan step at 8 lanes: what the saved rounds are worth, the spill costs) and the continuation adds 1.9 ms
// (profiles/experiments.md). Kept opt-in with its parity tests so that the measurement can be repeated.
int walk_stop_lanes() {
    static const int v = [] { const char* e = getenv("LOCGPU_WALK_STOP"); const int x = e ? atoi(e) : 0; return (x > 0 && x < 32) ? x : 0; }();
    return v;
}

// Waves of queries from which a search launch hands its stragglers over to the continuation kernel (one more launch must pay for
// itself); LOCGPU_WALK_STOP_MIN_WAVES lowers it for the parity tests.
size_t walk_stop_min_waves() {
    static const size_t v = [] { const char* e = getenv("LOCGPU_WALK_STOP_MIN_WAVES"); return e ? (size_t)atoll(e) : (size_t)16384; }();
    return v;
}

// LOCGPU_PLANE_FIT (read once): 1 = secular-equation fit (plane_null_vector_secular), 0 = the 4-column Jacobi fit.
int plane_fit_mode() {
    static const int mode = [] {
        const char* e = getenv("LOCGPU_PLANE_FIT");
        return e && *e ? atoi(e) : 1;
    }();
    return mode;
}

// LOCGPU_PLANE_CACHE (read once): 0 = off (DEFAULT: measured a net loss, see icp_plane_cached_accum_kernel), 1 = on; experiments: 2 = the
// cached kernel's structure with every point refitted, 3 = the search kernel marks unchanged lists but the plain fit kernel runs
int plane_cache_mode() {
    static const int m = [] { const char* e = getenv("LOCGPU_PLANE_CACHE"); const int v = e ? atoi(e) : 0; return (v >= 0 && v <= 3) ? v : 0; }();
    return m;
}

// Does launch_icp_search(a) run the 64-lane walk kernel, the one that fills a.same_mask? (The fit/accumulate launcher must take the
// same decision: the plane cache is only usable behind it.) Mirrors the choices of launch_icp_search / launch_fast_kd.
bool icp_search_writes_same_mask(const SearchArgs& a) {
    static const int walk = [] { const char* e = getenv("LOCGPU_WALK"); return e ? atoi(e) : 1; }();
    static const int small = [] { const char* e = getenv("LOCGPU_SMALL_LANES"); return e ? atoi(e) : 16; }();
    if (plane_cache_mode() == 0 || !a.same_mask || a.k != 5 || a.depth > 64) return false;
    if (a.visit_totals != nullptr || search_variant() != 0 || !a.redo_list || !a.redo_list2 || walk != 1) return false;
    const int n_launch = a.active ? a.n_active : a.n_scans;
    if ((size_t)((a.max_n + 63) / 64) * n_launch <= 2048 && small == 16) return false;  // the 16-lane one-scan kernel
    return true;
}

bool launch_icp_search(const SearchArgs& a, hipStream_t s) {
    // Instrumented (visit-counting) runs and LOCGPU_SEARCH_VARIANT experiments use the exact one-pass kernel.
    const bool exact_only = a.visit_totals != nullptr || search_variant() != 0 || !a.redo_list;
    if (!exact_only) {
        if (a.k == 1) return launch_fast_k<1>(a, s);
        if (a.k == 5) return launch_fast_k<5>(a, s);
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

int launch_icp_accum(int method, const AccumArgs& a, hipStream_t s) {
    const int blocks = (a.max_n + kBlock - 1) / kBlock;
    // points per thread: amortise the block reduction when the batch already fills the chip; 1 for small launches (latency).
    // The plane kernel's reduction is cheap (LDS rows, see there): 4 is as good as 8 and leaves a finer tail; the line and point
    // kernels still pay a 28-value wave reduction per block.
    static const int forced = [] { const char* e = getenv("LOCGPU_PLANE_PTS"); return e ? atoi(e) : 0; }();
    const long total_blocks = (long)blocks * a.n_scans;  // ALL scans of the batch, open or not: the split — hence the order of the sums — must not depend on a.active
    int pts = forced > 0 ? forced : (total_blocks >= 8192 ? (method == 2 ? 4 : 8) : (total_blocks >= 4096 ? 4 : (total_blocks >= 2048 ? 2 : 1)));
    if (pts > 8) pts = 8;
    if (method == 2 && a.plane_cache && a.same_mask) pts = kPlaneCachePts;
    const dim3 grid((blocks + pts - 1) / pts, a.active ? a.n_active : a.n_scans);
    if (method == 2 && a.plane_cache && a.same_mask) {
        if (plane_fit_mode() == 1)
            hipLaunchKernelGGL(icp_plane_cached_accum_kernel<1>, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts,
                               a.active, a.plane_cache, a.same_mask, a.use_cache);
        else
            hipLaunchKernelGGL(icp_plane_cached_accum_kernel<0>, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts,
                               a.active, a.plane_cache, a.same_mask, a.use_cache);
    } else if (method == 2) {
        if (plane_fit_mode() == 1)
            hipLaunchKernelGGL(icp_plane_accum_kernel<1>, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active);
        else
            hipLaunchKernelGGL(icp_plane_accum_kernel<0>, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active);
    } else if (method == 1)
        hipLaunchKernelGGL(icp_line_accum_kernel, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active);
    else
        hipLaunchKernelGGL(icp_point_accum_kernel, grid, dim3(kBlock), 0, s, a.tree, a.src, a.counts, a.st, a.nn, a.nn_pitch, a.max_n, a.gate, a.partials, pts, a.active);
    return (int)grid.x;
}

void launch_gn_solve(const double* partials, int blocks_per_scan, PoseState* st, int n_scans, const GnParams& prm, int do_update, double* hb_out,
                     unsigned int* list_counts, hipStream_t s) {
    hipLaunchKernelGGL(gn_solve_kernel, dim3(n_scans), dim3(kBlock), 0, s, partials, blocks_per_scan, st, prm, do_update, hb_out, list_counts);
}

void launch_sum_partials(const double* partials, int blocks_per_scan, const PoseState* st_all, int first, int n_local, int n_total, double* acc,
                         hipStream_t s) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(n_total), dim3(kBlock), 0, s, partials, blocks_per_scan, st_all, first, n_local, acc);
}

void launch_count_touched(uint32_t* touched, size_t n_words, unsigned long long* totals, hipStream_t s) {
    hipLaunchKernelGGL(count_touched_kernel, dim3(1024), dim3(kBlock), 0, s, touched, n_words, totals);
}

void launch_transform_cloud(const float4* src, size_t n, const float* m12, float4* dst, hipStream_t s) {
    hipLaunchKernelGGL(transform_cloud_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, src, n, m12, dst);
}

}  // namespace locgpu
