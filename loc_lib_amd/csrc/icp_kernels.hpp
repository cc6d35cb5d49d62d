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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
