// loc_lib_amd/csrc/icp_kernels.hpp
//
// HIP kernels of the ICP hot path for gfx950 (MI355X). One thread per source point, 256-thread
// workgroups, 2-D grids (x: 256-point chunk of a scan, y: scan of the batch).
//
//   K1  icp_search_kernel   transform (FP64) → float32 query → reference-faithful KD-tree DFS
//                           (kdtree.cpp:169-236) with the per-thread DFS stack in LDS and the
//                           libstdc++ binary-heap result set in VGPRs. HBM-latency/transaction bound.
//   K2  icp_*_accum_kernel  gather the k neighbours, fit plane / line (math_utils.h:112-163) with an
//                           in-register FP64 one-sided Jacobi SVD, gate, build J, accumulate the 21+6
//                           unique normal-equation sums; wave-64 butterfly + LDS → one partial per block
//                           (icp_registration.cpp:57-213).
//   K3  gn_solve_kernel     deterministic reduction of the block partials, 6×6 LU, SE3 update,
//                           convergence flags kept on the device (icp_registration.cpp:267-381).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hpp"

namespace locgpu {

constexpr int kBlock = 256;
constexpr int kAccW = 32;          // doubles per block partial: 21 H (upper) + 6 B + effective_num + 4 spare
constexpr uint32_t kInvalidSlot = 0xFFFFFFFFu;
constexpr int kSearchStatSlots = 16 + 2 * 64;  // [0..15] counters, then two 64-bin histograms of the LOCGPU_STAMP diagnostic build

struct GnParams {
    int method;              // locgpu_icp_method, 3 = direct NDT, 4 = incremental NDT
    int max_iteration;
    int min_effective_pts;
    double eps;
    double max_nn_distance, max_plane_distance, max_line_distance;
    bool operator==(const GnParams& o) const {  // field by field: the struct has padding bytes
        return method == o.method && max_iteration == o.max_iteration && min_effective_pts == o.min_effective_pts && eps == o.eps &&
               max_nn_distance == o.max_nn_distance && max_plane_distance == o.max_plane_distance && max_line_distance == o.max_line_distance;
    }
};

__device__ __forceinline__ float as_f32(uint32_t u) { return __uint_as_float(u); }

// list[(*count)++] = value for every lane with `pred`, one atomic per wave (same-address atomics cost ≈10 ns each on gfx950: a
// per-lane append from millions of queries serialises for milliseconds). Every active lane of the wave must reach the call.
__device__ __forceinline__ void wave_append(uint32_t* __restrict__ list, unsigned int* __restrict__ count, bool pred, uint32_t value) {
    const unsigned long long mask = __ballot(pred);
    if (!mask) return;
    const int lane = (int)__lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(mask));
    base = (unsigned int)__shfl((int)base, leader, 64);
    if (pred) list[base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// ---------------------------------------------------------------------------------------------
// Result set: std::priority_queue<NodeAndDistance> (kdtree.h:33-39, kdtree.cpp:156-165,197-212)
// restated as libstdc++'s __push_heap / __adjust_heap on a register-resident array. Dynamic
// positions are resolved with select chains so nothing spills to scratch.
template <int KMAX>
struct KnnHeap {
    float d[KMAX + 1];
    uint32_t id[KMAX + 1];
    int n;

    __device__ __forceinline__ float getd(int i) const {
        float r = d[0];
#pragma unroll
        for (int j = 1; j <= KMAX; ++j) {
            const float t = d[j];  // unconditional static-index read: keeps the array promotable to VGPRs
            r = (i == j) ? t : r;
        }
        return r;
    }
    __device__ __forceinline__ uint32_t geti(int i) const {
        uint32_t r = id[0];
#pragma unroll
        for (int j = 1; j <= KMAX; ++j) {
            const uint32_t t = id[j];
            r = (i == j) ? t : r;
        }
        return r;
    }
    __device__ __forceinline__ void set(int i, float v, uint32_t w) {
#pragma unroll
        for (int j = 0; j <= KMAX; ++j) {
            const float od = d[j];
            const uint32_t oi = id[j];
            d[j] = (i == j) ? v : od;
            id[j] = (i == j) ? w : oi;
        }
    }
    __device__ __forceinline__ float top() const { return d[0]; }

    // __push_heap(first, hole, 0, value): sift the hole up while parent < value.
    __device__ __forceinline__ void sift_up(int hole, float val, uint32_t w) {
        while (hole > 0) {
            const int parent = (hole - 1) / 2;
            const float pd = getd(parent);
            if (!(pd < val)) break;
            set(hole, pd, geti(parent));
            hole = parent;
        }
        set(hole, val, w);
    }
    __device__ __forceinline__ void push(float val, uint32_t w) {  // emplace = push_back + push_heap
        const int hole = n;
        n++;
        sift_up(hole, val, w);
    }
    __device__ __forceinline__ void pop() {  // pop_heap + pop_back
        const float value = getd(n - 1);
        const uint32_t vid = geti(n - 1);
        const int len = n - 1;
        int hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (getd(second) < getd(second - 1)) second--;
            set(hole, getd(second), geti(second));
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            set(hole, getd(second - 1), geti(second - 1));
            hole = second - 1;
        }
        sift_up(hole, value, vid);
        n--;
    }
};

template <>
struct KnnHeap<1> {  // k = 1 (P2P): push-then-pop of a strictly smaller distance just replaces the element
    float d[1];
    uint32_t id[1];
    int n;
    __device__ __forceinline__ float top() const { return d[0]; }
    __device__ __forceinline__ uint32_t geti(int) const { return id[0]; }
    __device__ __forceinline__ void push(float val, uint32_t w) {
        if (n == 0 || val < d[0]) { d[0] = val; id[0] = w; }  // with n == 1 the larger of the two is what pop() removes
        n++;
    }
    __device__ __forceinline__ void pop() { n--; }
};

// ---------------------------------------------------------------------------------------------
// KdTree::Knn / ComputeDisForLeaf / NeedExpand (kdtree.cpp:169-236), iterative. The DFS stack holds,
// for each internal node on the current path whose far side may still be expanded, the far child's
// slot and d² = (q[axis] − thresh)²; NeedExpand is evaluated when the entry is popped, i.e. after the
// near side returned, exactly like the recursion. An entry is not pushed when the result set is already
// full and d² ≥ top·alpha: top only shrinks, so that NeedExpand could never come true later.
// alpha_eff = alpha_ in approximate mode, 1.0f in exact mode (x·1.0f == x).
template <int KMAX, int D, bool COUNT>
__device__ __forceinline__ void tree_knn(const uint2* __restrict__ tree, float qx, float qy, float qz, int k, float alpha_eff,
                                         uint32_t (*s_far)[kBlock], float (*s_d2)[kBlock], int tid, KnnHeap<KMAX>& heap,
                                         uint32_t& nvis, uint32_t& lvis) {
    heap.n = 0;
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        for (;;) {  // descend to a leaf
            uint4 w;
            __builtin_memcpy(&w, tree + cur, 16);  // node + following slot in one 16-byte load
            if (COUNT) nvis++;
            const uint32_t meta = w.y;
            const uint32_t tag = meta >> 30;
            if (tag == 3u) {
                if (COUNT) lvis++;
                const float dx = qx - as_f32(w.x), dy = qy - as_f32(w.z), dz = qz - as_f32(w.w);
                const float dis2 = dx * dx + (dy * dy + dz * dz);  // Eigen squaredNorm order, no FMA (-ffp-contract=off)
                if (heap.n < k) {
                    heap.push(dis2, cur);
                } else if (dis2 < heap.top()) {
                    heap.push(dis2, cur);
                    heap.pop();
                }
                break;
            }
            const float th = as_f32(w.x);
            const float qa = tag == 0u ? qx : (tag == 1u ? qy : qz);
            const float dd = qa - th;
            const float d2 = dd * dd;
            const uint32_t right = meta & 0x3FFFFFFFu;
            const bool go_left = qa < th;
            const uint32_t far_slot = go_left ? right : cur + 1u;
            if (heap.n < k || d2 < heap.top() * alpha_eff) {
                s_far[sp][tid] = far_slot;
                s_d2[sp][tid] = d2;
                sp = sp + 1 < D ? sp + 1 : D - 1;  // host guarantees depth-1 <= D; clamp keeps LDS in bounds regardless
            }
            cur = go_left ? cur + 1u : right;
        }
        bool found = false;
        while (sp > 0) {  // backtrack: NeedExpand at the youngest pending node
            sp--;
            const float d2 = s_d2[sp][tid];
            if (heap.n < k || d2 < heap.top() * alpha_eff) {
                cur = s_far[sp][tid];
                found = true;
                break;
            }
        }
        if (!found) break;
    }
}

// Same traversal as tree_knn, written as ONE loop in which every live lane visits exactly one node per trip
// (load → leaf: result-set update + backtrack | internal: push far side, step to the near side). All lanes of a wave
// issue their node load at the same point, so a wave keeps up to 64 loads in flight instead of serialising the
// descend / backtrack phases of different lanes.
template <int KMAX, int D, bool COUNT, int BLK = kBlock>
__device__ __forceinline__ void tree_knn_flat(const uint2* __restrict__ tree, float qx, float qy, float qz, int k, float alpha_eff,
                                              uint32_t (*s_far)[BLK], float (*s_d2)[BLK], int tid, KnnHeap<KMAX>& heap,
                                              uint32_t& nvis, uint32_t& lvis, uint32_t* __restrict__ touched = nullptr) {
    heap.n = 0;
    int sp = 0;
    uint32_t cur = 0;
    bool live = true;
    while (live) {
        uint4 w;
        __builtin_memcpy(&w, tree + cur, 16);
        if (COUNT) nvis++;
        const uint32_t meta = w.y;
        const uint32_t tag = meta >> 30;
        // instrumented pass: one bit per 8-byte tree slot this launch reads at all (a leaf is two slots) — the compulsory tree bytes
        if (COUNT && touched) {
            atomicOr(&touched[cur >> 5], 1u << (cur & 31u));
            if (tag == 3u) atomicOr(&touched[(cur + 1u) >> 5], 1u << ((cur + 1u) & 31u));
        }
        if (tag == 3u) {
            if (COUNT) lvis++;
            const float dx = qx - as_f32(w.x), dy = qy - as_f32(w.z), dz = qz - as_f32(w.w);
            const float dis2 = dx * dx + (dy * dy + dz * dz);
            if (heap.n < k) {
                heap.push(dis2, cur);
            } else if (dis2 < heap.top()) {
                heap.push(dis2, cur);
                heap.pop();
            }
            live = false;
            const float bound = heap.top() * alpha_eff;
            const bool open = heap.n < k;
            while (sp > 0) {
                sp--;
                if (open || s_d2[sp][tid] < bound) {
                    cur = s_far[sp][tid];
                    live = true;
                    break;
                }
            }
        } else {
            const float th = as_f32(w.x);
            const float qa = tag == 0u ? qx : (tag == 1u ? qy : qz);
            const float dd = qa - th;
            const float d2 = dd * dd;
            const uint32_t right = meta & 0x3FFFFFFFu;
            const bool go_left = qa < th;
            if (heap.n < k || d2 < heap.top() * alpha_eff) {
                s_far[sp][tid] = go_left ? right : cur + 1u;
                s_d2[sp][tid] = d2;
                sp = sp + 1 < D ? sp + 1 : D - 1;
            }
            cur = go_left ? cur + 1u : right;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path of the same traversal. Two things make the exact kernel above slow: the libstdc++ heap emulation
// (hundreds of select instructions whenever any lane of the wave sits on a leaf) and the 8-byte × depth LDS stack
// that caps occupancy at two workgroups per CU. Both are only needed in rare situations, which this kernel detects:
//   * result set: a sorted array with insertion (≈40 instructions). It keeps exactly the elements and the final
//     order std::priority_queue would unless an eviction happens while the maximum is tied (which element leaves is then a
//     matter of heap layout) or two distances of the final set are equal (their pop order is); both raise `slow`.
//   * stack: the first T pushes of a query are always the top T tree levels of its first descent (the result set is
//     empty, NeedExpand is unconditionally true). They are popped last, against the final bound, and are almost never
//     expanded — so they are not stored at all, only the minimum of their d². When the stack drains down to them:
//     if min d² ≥ top·alpha (and the set is full) every one of them would be rejected and the query is finished,
//     exactly; otherwise `slow`.
// A query that raised `slow` is appended to a redo list and recomputed by the exact kernel.
template <int K>
struct SortedSet {  // ascending: d[0] ≤ … ≤ d[K-1]; empty slots hold +inf / kInvalidSlot
    float d[K];
    uint32_t id[K];
    int n;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = __builtin_inff(); id[j] = kInvalidSlot; }
        n = 0;
    }
    __device__ __forceinline__ float top() const { return d[K - 1]; }
    // returns true when the insertion tied with a resident distance (⇒ heap layout would matter)
    __device__ __forceinline__ bool insert(float x, uint32_t w) {
        bool tie = false;
#pragma unroll
        for (int j = 0; j < K - 1; ++j) tie |= (x == d[j]);
        d[K - 1] = x;
        id[K - 1] = w;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool sw = d[j] < d[j - 1];
            const float lo = sw ? d[j] : d[j - 1], hi = sw ? d[j - 1] : d[j];
            const uint32_t ilo = sw ? id[j] : id[j - 1], ihi = sw ? id[j - 1] : id[j];
            d[j - 1] = lo; d[j] = hi; id[j - 1] = ilo; id[j] = ihi;
        }
        n = n < K ? n + 1 : K;
        return tie;
    }
};

// Returns true when the query must be redone by the exact kernel. T = number of un-stored leading stack positions.
//
// Replay: when the stack drains down to the un-stored entries and one of them could still pass NeedExpand
// (min d² < top·alpha, or the set is not full), the top levels are walked again from the root — the same `<`
// decisions, hence the same path — pushing entries under the pruning rule with the CURRENT bound (valid because the
// bound only shrinks). `sp` such entries were pending, all from levels 0..sp-1 of the first descent, so exactly `sp`
// levels are replayed; from then on every position is stored (T = 0). Decisions and visit order stay those of the
// recursion.
//
// The loop body is written branch-free (selects) except for the LDS push and the backtrack block: the kernel is
// bound by instruction issue — 64 lanes sit in different phases, so a branchy body executes every side anyway and
// pays the exec-mask bookkeeping on top. `tree_rsrc` is a buffer descriptor over the packed tree: one
// buffer_load_dwordx4 (32-bit offset) returns a node together with the slot behind it (a whole leaf).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// STAMP = diagnostic build only (LOCGPU_STAMP=1): diag[0..4] receive this lane's first-descent cycles, total cycles,
// main-loop trips, VISIT trips and POP rounds. Never instantiated in the timed path.
template <int K, int DF, int BLK, bool STAMP = false>
__device__ __forceinline__ bool tree_knn_fast(__amdgpu_buffer_rsrc_t tree_rsrc, float qx, float qy, float qz, float alpha_eff, int T,
                                              uint2 (*s_stack)[BLK], int tid, SortedSet<K>& set, unsigned long long* diag = nullptr,
                                              unsigned long long* replay_counter = nullptr) {
    unsigned long long t_begin = 0, t_mid = 0;
    unsigned int n_trips = 0, n_visit = 0, n_pop = 0, n_desc = 0;
    if (STAMP) t_begin = __builtin_amdgcn_s_memtime();
    set.init();
    int sp = 0;
    uint32_t cur = 0;
    // The un-stored levels: the two smallest d² among them with their far slots (c1 ≤ c2, and which of the two is the deeper,
    // i.e. younger, level) and the third smallest d². When the stack has drained, the entries that can still pass NeedExpand are
    // among these — the bound only shrinks, so an entry that fails once fails for good. If at most the two candidates can pass,
    // they are expanded directly, the younger first and the other re-tested after it, exactly as the recursion would (entries
    // younger than a candidate are tested before it and fail; older ones after it, against a smaller bound). Only when the third
    // could pass as well are the top levels replayed from the root.
    float c1_d2 = __builtin_inff(), c2_d2 = __builtin_inff(), c3_d2 = __builtin_inff();
    uint32_t c1_far = 0, c2_far = 0, c1_younger = 0;
    // Loop-carried flags live in VGPRs as integers: a divergent `bool` is a lane mask in SGPRs, and every region that
    // assigns it costs three scalar mask instructions at its merge point — the scalar unit is shared by the CU's four SIMDs.
    uint32_t slow = 0, live = 1, need_pop = 0;

    // ---- first descent: the result set is empty, so every internal node pushes its far side (NeedExpand is true while
    // size < k) and no lane meets a leaf or pops. A minimal loop for these ≈depth trips; lanes leave it at their first leaf.
    for (;;) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(tree_rsrc, (int)(cur << 3), 0, 0);
        const uint32_t meta = w.y;
        const uint32_t tag = meta >> 30;
        if (tag == 3u) break;
        if (STAMP) n_desc++;
        const float th = as_f32(w.x);
        const float qa = tag == 0u ? qx : (tag == 1u ? qy : qz);
        const float dd = qa - th;
        const float d2 = dd * dd;
        const uint32_t right = meta & 0x3FFFFFFFu;
        const bool go_left = qa < th;
        // T ≥ depth − DF (launch_fast_kd), and a descent pushes at most depth − 1 entries: position sp − T never reaches DF here
        const uint32_t far_slot = go_left ? right : cur + 1u;
        if (sp < T) {  // levels come in increasing depth: the new entry is younger than both candidates
            const bool lt1 = d2 < c1_d2, lt2 = d2 < c2_d2;
            c3_d2 = lt2 ? c2_d2 : (d2 < c3_d2 ? d2 : c3_d2);
            c2_d2 = lt1 ? c1_d2 : (lt2 ? d2 : c2_d2);
            c2_far = lt1 ? c1_far : (lt2 ? far_slot : c2_far);
            c1_younger = lt1 ? 1u : (lt2 ? 0u : c1_younger);
            c1_far = lt1 ? far_slot : c1_far;
            c1_d2 = lt1 ? d2 : c1_d2;
        } else {
            s_stack[sp - T][tid] = make_uint2(far_slot, __float_as_uint(d2));
        }
        sp++;
        cur = go_left ? cur + 1u : right;
    }
    live = slow ^ 1u;
    if (STAMP) t_mid = __builtin_amdgcn_s_memtime();

    while (live) {
        if (STAMP) { n_trips++; n_visit += need_pop ? 0 : 1; n_pop += need_pop ? 1 : 0; }
        if (!need_pop) {  // ------------------------------------------------ VISIT one node
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(tree_rsrc, (int)(cur << 3), 0, 0);
            const uint32_t meta = w.y;
            const uint32_t tag = meta >> 30;
            const bool is_leaf = tag == 3u;
            const float top = set.top();  // +inf until the set holds K points: `x < top` and `x < top·alpha` are then true for every
                                          // finite x, which is exactly the reference's "size < k" rule (kdtree.cpp:203,222)

            // leaf side (ComputeDisForLeaf, kdtree.cpp:197-212). A real branch: during the first descent no lane of the
            // wave sits on a leaf and the whole block is skipped; inside it is select-only.
            if (is_leaf) {
                const float dx = qx - as_f32(w.x), dy = qy - as_f32(w.z), dz = qz - as_f32(w.w);
                const float dis2 = dx * dx + (dy * dy + dz * dz);
                const bool ins = dis2 < top;  // n<K: top is +inf; n==K: strict `<` (kdtree.cpp:207)
                set.d[K - 1] = ins ? dis2 : set.d[K - 1];
                set.id[K - 1] = ins ? cur : set.id[K - 1];
#pragma unroll
                for (int j = K - 1; j > 0; --j) {  // one bubble pass; a no-op when nothing was inserted
                    const bool sw = set.d[j] < set.d[j - 1];
                    const float lo = sw ? set.d[j] : set.d[j - 1], hi = sw ? set.d[j - 1] : set.d[j];
                    const uint32_t ilo = sw ? set.id[j] : set.id[j - 1], ihi = sw ? set.id[j - 1] : set.id[j];
                    set.d[j - 1] = lo; set.d[j] = hi; set.id[j - 1] = ilo; set.id[j] = ihi;
                }
                // The only moments the heap's layout decides WHICH elements stay: an eviction while the maximum is tied — the
                // evicted distance then equals the new maximum. (Ties that survive to the end are caught after the loop.)
                slow |= (ins && top == set.d[K - 1] && top < __builtin_inff()) ? 1u : 0u;
            }

            // internal side (Knn, kdtree.cpp:177-194), predicated on !is_leaf
            const float th = as_f32(w.x);
            const float qa = tag == 0u ? qx : (tag == 1u ? qy : qz);
            const float dd = qa - th;
            const float d2 = dd * dd;
            const uint32_t right = meta & 0x3FFFFFFFu;
            const bool go_left = qa < th;
            const uint32_t far_slot = go_left ? right : cur + 1u;
            const bool push = !is_leaf && d2 < top * alpha_eff;  // else NeedExpand can never come true later
            // Position in the stored part of the stack. Never negative here: the main loop only visits a node after popping a
            // stored entry (sp ≥ T afterwards) or after the replay (T = 0), so the un-stored levels are first-descent business only.
            // It can exceed the stored part only after a replay in a tree deeper than DF.
            const int idx = sp - T;
            const bool store = push && idx < DF;
            slow |= (push && idx >= DF) ? 1u : 0u;  // deeper than the fast stack
            if (store) s_stack[idx][tid] = make_uint2(far_slot, __float_as_uint(d2));
            sp += push ? 1 : 0;
            need_pop = is_leaf ? 1u : 0u;
            cur = go_left ? cur + 1u : right;
        }
        if (need_pop) {  // ------------------------------------------------- POP: NeedExpand (kdtree.cpp:214-236), youngest first
            const float bound = set.top() * alpha_eff;  // +inf while the set is not full: everything passes
            const int avail = sp - T;
            if (avail <= 0) {
                // nothing stored is left: only un-stored first-descent entries remain, of which the candidates are what matters
                const bool p1 = c1_d2 < bound;  // false also when there is none (+inf)
                live = p1 ? 1u : 0u;            // otherwise finished: every un-stored entry is rejected by the final bound
                if (p1 && !(c3_d2 < bound)) {
                    const bool p2 = c2_d2 < bound;
                    const bool take1 = !p2 || c1_younger != 0u;
                    cur = take1 ? c1_far : c2_far;
                    // the other candidate stays on for a later test if it passes now; else it (like everything else) is out for good
                    c1_d2 = p2 ? (take1 ? c2_d2 : c1_d2) : __builtin_inff();
                    c1_far = take1 ? c2_far : c1_far;
                    c2_d2 = __builtin_inff();
                    c3_d2 = __builtin_inff();
                    sp = 0;
                    T = 0;
                    need_pop = 0;
                } else if (p1) {
                    if (STAMP && replay_counter) atomicAdd(replay_counter, 1ull);  // diagnostic build only: a live pointer across the loop costs 1.6 %
                    // Rare: three or more could pass. Walk the un-stored levels 0..sp-1 again from the root (same `<` decisions, hence
                    // the same internal nodes) and push them under the pruning rule with the CURRENT bound; from now on every position
                    // is stored. (No direct expansion has happened before: it leaves c3 = +inf.)
                    const int levels = sp;
                    uint32_t c = 0;
                    sp = 0;
                    T = 0;
                    for (int l = 0; l < levels; ++l) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(tree_rsrc, (int)(c << 3), 0, 0);
                        const uint32_t tg = v.y >> 30;
                        const float th = as_f32(v.x);
                        const float qa = tg == 0u ? qx : (tg == 1u ? qy : qz);
                        const float dd = qa - th;
                        const float d2 = dd * dd;
                        const uint32_t right = v.y & 0x3FFFFFFFu;
                        const bool go_left = qa < th;
                        if (d2 < bound) {
                            if (sp < DF) s_stack[sp][tid] = make_uint2(go_left ? right : c + 1u, __float_as_uint(d2));
                            else slow = 1;
                            sp++;
                        }
                        c = go_left ? c + 1u : right;
                    }
                    c1_d2 = __builtin_inff();
                    c2_d2 = __builtin_inff();
                    c3_d2 = __builtin_inff();
                }
                // need_pop stays set: the re-pushed entries are popped like any others (an empty stack ends the query next trip)
            } else {
                // up to four entries per LDS round trip, examined youngest first: their d² first, then the far slot of the one
                // that passes, read with a computed row (a select over four loaded slots makes the compiler branch per case)
                const uint32_t* s32 = reinterpret_cast<const uint32_t*>(&s_stack[0][0]);
                float ed2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = avail - 1 - j >= 0 ? avail - 1 - j : 0;
                    ed2[j] = __uint_as_float(s32[(row * BLK + tid) * 2 + 1]);
                }
                int hit = 4;
#pragma unroll
                for (int j = 3; j >= 0; --j) hit = (j < avail && ed2[j] < bound) ? j : hit;
                const int row_hit = avail - 1 - hit >= 0 ? avail - 1 - hit : 0;
                const uint32_t far_hit = s32[(row_hit * BLK + tid) * 2];
                const bool found = hit < 4;
                const int used = found ? hit + 1 : (avail < 4 ? avail : 4);
                sp -= used;
                cur = found ? far_hit : cur;
                need_pop = found ? 0u : 1u;
            }
        }
        live = slow ? 0u : live;
    }
    // equal distances in the final set: std::priority_queue would pop them in a layout-dependent order
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) slow |= set.d[j] == set.d[j + 1] ? 1u : 0u;
    if (STAMP) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        diag[0] = t_mid - t_begin; diag[1] = t_end - t_begin; diag[2] = n_trips; diag[3] = n_visit; diag[4] = n_pop; diag[5] = n_desc;
    }
    return slow != 0;
}

// Pops the heap into ascending-distance order (kdtree.cpp:160-165).
template <int KMAX>
__device__ __forceinline__ void heap_to_sorted(KnnHeap<KMAX>& heap, uint32_t (&out)[KMAX], int& count) {
    count = heap.n;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) out[j] = kInvalidSlot;
    for (int i = count - 1; i >= 0; --i) {
        const uint32_t w = heap.geti(0);
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const uint32_t o = out[j];
            out[j] = (i == j) ? w : o;
        }
        heap.pop();
    }
}

// Leaf point at `slot` as doubles (ToVec3d, point_types.h:26).
__device__ __forceinline__ D3 leaf_point(const uint2* __restrict__ tree, uint32_t slot) {
    uint4 w;
    __builtin_memcpy(&w, tree + slot, 16);
    return {(double)as_f32(w.x), (double)as_f32(w.z), (double)as_f32(w.w)};
}

}  // namespace locgpu
