// loc_lib_amd/csrc/search_walk.hpp — round-3 form of the hot search traversal (K1), see DESIGN.md §3.
//
// Same recursion as tree_knn_fast (kdtree.cpp:169-236 replayed node for node), reorganised so that ONE straight-line, fully
// predicated region is executed per loop trip — no per-lane branches, hence no exec-mask bookkeeping on the CU's shared scalar unit:
//
//   * every lane visits exactly one node per trip. A lane that has nothing to visit (it is still popping, or it is finished) visits
//     the SENTINEL LEAF behind the packed tree (coordinates 3e38: its squared distance overflows to +inf and is never inserted), so
//     "leaf" is the only predicate of the trip: internal node → push the far side, step to the near side; leaf → result-set update,
//     then POP up to four stack entries;
//   * the stack holds {far slot, −d²}. With the sign flipped, `d² < bound` is a SIGNED INTEGER compare of the raw bits against the
//     bits of −bound (non-positive floats order like their magnitudes), a row below the stack's bottom reads as 0 from outside the
//     workgroup's LDS allocation (never "less than" a non-positive bound) and INT_MIN (−0.0f) as the bound switches a test off:
//     the `j < avail` and `is this lane popping` predicates cost nothing;
//   * the push is an unconditional LDS store to the row above the top (garbage there is harmless; beyond the last row it falls off
//     the allocation) followed by `avail += pushed`;
//   * the two candidates of the un-stored top levels (tree_knn_fast) are materialised as rows 0 and 1 of the stack after the first
//     descent — older below younger, which is the recursion's order — so the drain to them is an ordinary pop; only the third
//     smallest d² stays in a register. A query for which it could pass as well (≈1e-3 of them) is marked and recomputed by the
//     DEEP PASS: the same traversal with every level stored, over a device-side list. The other DF−2 rows hold one level each and
//     T (the number of un-stored levels) is chosen so that the first descent and everything below a stored level fit; only the
//     descent from a candidate, which starts on an un-stored level, can outgrow the rows — such a query goes to the deep pass too;
//   * a pop only happens on a leaf, and a lane on a leaf pushes nothing: the four youngest rows are read at the top of the trip,
//     next to the node load, not behind it;
//   * the result set is updated with v_med3_f32: inserting x into ascending d[0..K) and dropping the largest is
//     d'[j] = med3(d[j-1], x, d[j]), d'[0] = min(d[0], x).
//
// A query is flagged `slow` (→ exact recomputation with the libstdc++ heap, as before) when an eviction happens while the maximum
// is tied or when two distances of its final set are equal.
#pragma once
#include "icp_kernels.hpp"

namespace locgpu {

constexpr float kSentinelCoord = 3.0e38f;  // the sentinel leaf behind the tree: (q − 3e38)² = +inf for every sane query

template <int K>
struct Walk {
    float qx, qy, qz;
    float d[K];        // ascending; +inf = empty
    uint32_t id[K];
    uint32_t cur;      // slot to visit next (`dummy` = the sentinel leaf: nothing to visit)
    int avail;         // rows on the stack
    uint32_t c3n;      // bits of −(third smallest d² of the un-stored levels); 0 = none; 1 = the query needs the deep pass
    uint32_t slow;
    uint32_t rounds, wave_rounds;  // LOCGPU_STAMP diagnostic build only: main-loop rounds this lane needed / its wave ran
};

typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((address_space(3))) lds_u32x2;

__device__ __forceinline__ lds_u32* lds_ptr(uint32_t byte_addr) { return reinterpret_cast<lds_u32*>(byte_addr); }

// First descent (result set empty ⇒ every level pushes). Levels 0..T-1 are not stored: their two smallest d² become rows 0/1, the
// third smallest goes to w.c3n. Levels ≥ T are stored from row 2 on. Leaves w.cur at the first leaf (not yet visited).
// col_addr = LDS byte address of this lane's stack column (row stride ROWB bytes).
// Every lane still in the loop is on the same level (one level per trip from the root), so "is this level stored" is a scalar
// question: two loops — candidates only, then stores only — instead of one body that carries both (the kernel is VALU-bound).
template <int K, int ROWB>
__device__ __forceinline__ void walk_descend(__amdgpu_buffer_rsrc_t rsrc, const uint2* __restrict__ tree, Walk<K>& w, int T, uint32_t col_addr) {
    float c1 = __builtin_inff(), c2 = __builtin_inff(), c3 = __builtin_inff();
    uint32_t f1 = 0, f2 = 0, c1_younger = 0;
    uint32_t cur = 0;
    const float qx = w.qx, qy = w.qy, qz = w.qz;  // values, not lvalues: `c ? w.qx : w.qy` is a select of ADDRESSES and pins w in scratch
    bool at_leaf = false;
    int l = 0;
    // The lanes of a wave are neighbours on a scan ring: they share the top ≈10–14 levels of the tree. While every lane takes the
    // same turn, the node is read ONCE with a scalar load and its axis, threshold and children are wave-uniform — per lane only
    // d² and the candidate update remain (the kernel is VALU-bound, the scalar unit idles).
    {
        uint32_t cur_s = 0;
        for (; l < T; ++l) {
            const uint2 hd = tree[__builtin_amdgcn_readfirstlane(cur_s)];
            const uint32_t meta = __builtin_amdgcn_readfirstlane(hd.y);
            if (meta >= 0xC0000000u) break;  // a leaf above level T: the per-lane loop below sees it
            const float th = as_f32(__builtin_amdgcn_readfirstlane(hd.x));
            const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
            const bool go_left = qa < th;
            const unsigned long long m = __ballot(go_left);
            if (m != 0ull && m != __ballot(true)) break;  // the lanes part here: this level is handled per lane
            const uint32_t right = meta & 0x3FFFFFFFu, cur1 = cur_s + 1u;
            const uint32_t far_slot = m != 0ull ? right : cur1;
            const float dd = qa - th;
            const float d2 = dd * dd;
            const bool lt1 = d2 < c1, lt2 = d2 < c2;
            c3 = __builtin_amdgcn_fmed3f(c2, d2, c3);
            c2 = __builtin_amdgcn_fmed3f(c1, d2, c2);
            c1 = __builtin_fminf(c1, d2);
            const uint32_t f2n = lt2 ? far_slot : f2;
            f2 = lt1 ? f1 : f2n;
            f1 = lt1 ? far_slot : f1;
            const uint32_t yn = lt2 ? 0u : c1_younger;
            c1_younger = lt1 ? 1u : yn;
            cur_s = m != 0ull ? cur1 : right;
        }
        cur = cur_s;
    }
    for (; l < T; ++l) {  // un-stored levels: sorted insertion of d² into (c1 ≤ c2 ≤ c3), far slots of the first two
        const u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
        const uint32_t meta = n.y;
        if (meta >= 0xC0000000u) { at_leaf = true; break; }
        const float th = as_f32(n.x);
        const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
        const float dd = qa - th;
        const float d2 = dd * dd;
        const uint32_t right = meta & 0x3FFFFFFFu;
        const bool go_left = qa < th;
        const uint32_t cur1 = cur + 1u;
        const uint32_t far_slot = go_left ? right : cur1;
        const bool lt1 = d2 < c1, lt2 = d2 < c2;
        c3 = __builtin_amdgcn_fmed3f(c2, d2, c3);
        c2 = __builtin_amdgcn_fmed3f(c1, d2, c2);
        c1 = __builtin_fminf(c1, d2);
        const uint32_t f2n = lt2 ? far_slot : f2;
        f2 = lt1 ? f1 : f2n;
        f1 = lt1 ? far_slot : f1;
        const uint32_t yn = lt2 ? 0u : c1_younger;  // levels come in increasing depth: the new entry is younger than both candidates
        c1_younger = lt1 ? 1u : yn;
        cur = go_left ? cur1 : right;
    }
    int rows = 2;
    if (!at_leaf) {
        uint32_t row = col_addr + 2u * ROWB;
        for (;;) {  // stored levels
            const u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
            const uint32_t meta = n.y;
            if (meta >= 0xC0000000u) break;
            const float th = as_f32(n.x);
            const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
            const float dd = qa - th;
            const uint32_t right = meta & 0x3FFFFFFFu;
            const bool go_left = qa < th;
            const uint32_t cur1 = cur + 1u;
            *reinterpret_cast<lds_u32x2*>(row) = u32x2{go_left ? right : cur1, __float_as_uint(-(dd * dd))};
            row += ROWB;
            rows++;
            cur = go_left ? cur1 : right;
        }
    }
    // rows 0 (older) and 1 (younger): popped younger first, each against the bound of its own moment — the recursion's order
    const bool y1 = c1_younger != 0u;
    lds_u32* p0 = lds_ptr(col_addr);
    const uint32_t f_old = y1 ? f2 : f1, f_young = y1 ? f1 : f2;
    const float c_old = y1 ? c2 : c1, c_young = y1 ? c1 : c2;
    p0[0] = f_old;
    p0[1] = __float_as_uint(-c_old);
    p0[ROWB / 4] = f_young;
    p0[ROWB / 4 + 1] = __float_as_uint(-c_young);
    w.cur = cur;
    w.avail = rows;
    w.c3n = c3 < __builtin_inff() ? __float_as_uint(-c3) : 0u;
}

// One trip of the main loop for every lane of the wave (see the header comment).
// A lane is finished when w.cur == dummy && w.avail == 0. A lane whose stack drains to rows 0/1 while the THIRD un-stored entry
// could pass as well cannot be continued from two candidates: it is marked (w.c3n = 1, sticky) and recomputed by the deep pass
// (the same traversal with every level stored); what it computes from there on is discarded.
//
// Latency: a pop only happens on a leaf, and a lane on a leaf pushes nothing — so the four youngest stack rows are read at the TOP
// of the trip, next to the node load, instead of behind it (two LDS round trips less on the dependent chain).
template <int K, int ROWB>
__device__ __forceinline__ void walk_trip(__amdgpu_buffer_rsrc_t rsrc, Walk<K>& w, float alpha, uint32_t dummy, uint32_t col_addr, int cap) {
    // every field is copied to a value first: a conditional between two members is an lvalue (a select of addresses) and would pin w in scratch
    const float qx = w.qx, qy = w.qy, qz = w.qz;
    const uint32_t cur = w.cur;
    const int avail = w.avail;
    float d[K];
    uint32_t id[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { d[j] = w.d[j]; id[j] = w.id[j]; }
    const u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
    // rows avail-4 .. avail-1 as {far slot, −d²}; a row below 0 lies outside the allocation and reads {0, 0} (plain integer
    // address arithmetic: it wraps by definition)
    const uint32_t a4 = col_addr + (uint32_t)(avail - 4) * ROWB;
    const u32x2 r3 = *reinterpret_cast<lds_u32x2*>(a4), r2 = *reinterpret_cast<lds_u32x2*>(a4 + ROWB),
                r1 = *reinterpret_cast<lds_u32x2*>(a4 + 2u * ROWB), r0 = *reinterpret_cast<lds_u32x2*>(a4 + 3u * ROWB);
    const uint32_t meta = n.y;
    const bool is_leaf = meta >= 0xC0000000u;

    // ---- leaf: ComputeDisForLeaf (kdtree.cpp:197-212). x = +inf for internal nodes and for the sentinel: nothing is inserted.
    const float dx = qx - as_f32(n.x), dy = qy - as_f32(n.z), dz = qz - as_f32(n.w);
    const float dis2 = dx * dx + (dy * dy + dz * dz);  // Eigen squaredNorm order, no FMA
    const float x = is_leaf ? dis2 : __builtin_inff();
    bool c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = x < d[j];  // strict: an equal distance is not inserted before its equals (kdtree.cpp:207)
    if (K >= 2) {
        // eviction while the maximum is tied: which of the tied elements leaves is a matter of heap layout ⇒ slow.
        // d[K-1] − d[K-2] is NaN while the set is not full (inf − inf), 0 exactly when a finite maximum is tied.
        const float gap = d[K - 1] - d[K - 2];
        const float xt = gap == 0.0f ? x : __builtin_inff();
        const uint32_t slow0 = w.slow;
        w.slow = xt < d[K - 1] ? 1u : slow0;
    }
#pragma unroll
    for (int j = K - 1; j >= 1; --j) {
        const uint32_t below = id[j - 1], here = id[j];
        const uint32_t t = c[j] ? cur : here;
        id[j] = c[j - 1] ? below : t;
        d[j] = __builtin_amdgcn_fmed3f(d[j - 1], x, d[j]);
    }
    {
        const uint32_t here = id[0];
        id[0] = c[0] ? cur : here;
        d[0] = __builtin_fminf(d[0], x);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) { w.d[j] = d[j]; w.id[j] = id[j]; }

    // −bound of this trip: for an internal node the set is unchanged, so one product serves the push test and the pop
    const uint32_t nbound = __float_as_uint(-(d[K - 1] * alpha));  // −inf while the set is not full: everything passes

    // ---- internal: Knn (kdtree.cpp:177-194)
    const float th = as_f32(n.x);
    const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
    const float dd = qa - th;
    const uint32_t nd2 = __float_as_uint(-(dd * dd));
    const uint32_t right = meta & 0x3FFFFFFFu;
    const bool go_left = qa < th;
    const uint32_t cur1 = cur + 1u;
    const uint32_t next = go_left ? cur1 : right;
    const uint32_t far_slot = go_left ? right : cur1;
    // unconditional store to the row above the top; it counts only if the entry can still pass NeedExpand (else it never will)
    const uint32_t top = col_addr + (uint32_t)avail * ROWB;
    *reinterpret_cast<lds_u32x2*>(top) = u32x2{far_slot, nd2};
    const uint32_t nd2g = is_leaf ? 0u : nd2;
    const bool push = (int)nd2g < (int)nbound;
    const int pushed = push ? 1 : 0;
    const uint32_t next_after = next;

    // ---- pop (NeedExpand, kdtree.cpp:214-236), youngest first, up to four rows; only lanes on a leaf (real or sentinel)
    const int nb = (int)(is_leaf ? nbound : 0x80000000u);  // INT_MIN: nothing passes
    int hit = 4;
    uint32_t far_hit = is_leaf ? dummy : next_after;
    hit = (int)r3.y < nb ? 3 : hit; far_hit = (int)r3.y < nb ? r3.x : far_hit;
    hit = (int)r2.y < nb ? 2 : hit; far_hit = (int)r2.y < nb ? r2.x : far_hit;
    hit = (int)r1.y < nb ? 1 : hit; far_hit = (int)r1.y < nb ? r1.x : far_hit;
    hit = (int)r0.y < nb ? 0 : hit; far_hit = (int)r0.y < nb ? r0.x : far_hit;
    const int avail_eff = is_leaf ? avail : 0;
    const int used = min(min(hit + 1, 4), avail_eff);
    // did the scan reach rows 0/1 while the third un-stored entry could pass as well? (lowest row examined: avail-1-min(hit,3))
    const int low = avail_eff - 1 - min(hit, 3);
    const uint32_t c3n = w.c3n;
    const uint32_t c3sel = low < 2 ? c3n : 0u;
    const bool deep = ((int)c3sel < nb) | (avail > cap);  // the third candidate could pass, or a push fell off the stack (see walk_rounds_capped)
    w.c3n = deep ? 1u : c3n;  // 1 = "deep pass" (a positive value never passes the first test again)
    w.cur = far_hit;
    w.avail = avail + pushed - used;
}

// ---------------------------------------------------------------------------------------------------------------------------
// "Rounds" form of the main loop: the kernel is bound by VALU issue (87 % busy, profiles/r03_pmc_kernels.md) and in the flat loop
// above every trip pays for the leaf block (≈45 instructions) AND the internal block (≈20) although a lane uses only one of them.
// Here a round is: leaf stage for the whole wave (every live lane sits on a leaf: result-set update + pop, to the next node), then
// a per-lane loop of internal steps down to the next leaf. A wave pays per round the longest descent of its lanes, but at the price
// of the small internal body only: search 21.9 -> 20.0 ms per 256-scan step. (Tried on top: an inner loop that keeps popping when
// four rows failed instead of a sentinel round — 20.5 ms, not kept.)
template <int K, int ROWB>
__device__ __forceinline__ void walk_rounds(__amdgpu_buffer_rsrc_t rsrc, Walk<K>& w, float alpha, uint32_t dummy, uint32_t col_addr, int cap) {
    const float qx = w.qx, qy = w.qy, qz = w.qz;
    uint32_t cur = w.cur;
    int avail = w.avail;
    float d[K];
    uint32_t id[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { d[j] = w.d[j]; id[j] = w.id[j]; }
    uint32_t slow = w.slow, c3n = w.c3n;
    u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);  // the first leaf (or the sentinel)
    do {
        // ---- leaf stage: ComputeDisForLeaf (kdtree.cpp:197-212), then NeedExpand over the four youngest rows (kdtree.cpp:214-236)
        const uint32_t a4 = col_addr + (uint32_t)(avail - 4) * ROWB;
        const u32x2 r3 = *reinterpret_cast<lds_u32x2*>(a4), r2 = *reinterpret_cast<lds_u32x2*>(a4 + ROWB),
                    r1 = *reinterpret_cast<lds_u32x2*>(a4 + 2u * ROWB), r0 = *reinterpret_cast<lds_u32x2*>(a4 + 3u * ROWB);
        const float dx = qx - as_f32(n.x), dy = qy - as_f32(n.z), dz = qz - as_f32(n.w);
        const float x = dx * dx + (dy * dy + dz * dz);  // Eigen squaredNorm order, no FMA; +inf for the sentinel
        bool c[K];
#pragma unroll
        for (int j = 0; j < K; ++j) c[j] = x < d[j];
        if (K >= 2) {
            const float gap = d[K - 1] - d[K - 2];
            const float xt = gap == 0.0f ? x : __builtin_inff();
            slow = xt < d[K - 1] ? 1u : slow;
        }
#pragma unroll
        for (int j = K - 1; j >= 1; --j) {
            const uint32_t below = id[j - 1], here = id[j];
            const uint32_t t = c[j] ? cur : here;
            id[j] = c[j - 1] ? below : t;
            d[j] = __builtin_amdgcn_fmed3f(d[j - 1], x, d[j]);
        }
        {
            const uint32_t here = id[0];
            id[0] = c[0] ? cur : here;
            d[0] = __builtin_fminf(d[0], x);
        }
        const int nb = (int)__float_as_uint(-(d[K - 1] * alpha));  // −inf while the set is not full: everything passes
        int hit = 4;
        uint32_t nxt = dummy;
        hit = (int)r3.y < nb ? 3 : hit; nxt = (int)r3.y < nb ? r3.x : nxt;
        hit = (int)r2.y < nb ? 2 : hit; nxt = (int)r2.y < nb ? r2.x : nxt;
        hit = (int)r1.y < nb ? 1 : hit; nxt = (int)r1.y < nb ? r1.x : nxt;
        hit = (int)r0.y < nb ? 0 : hit; nxt = (int)r0.y < nb ? r0.x : nxt;
        const int used = min(min(hit + 1, 4), avail);
        const int low = avail - 1 - min(hit, 3);
        const uint32_t c3sel = low < 2 ? c3n : 0u;
        const bool deep = ((int)c3sel < nb) | (avail > cap);
        c3n = deep ? 1u : c3n;
        cur = nxt;
        avail -= used;
        // ---- internal steps (Knn, kdtree.cpp:177-194) down to the next leaf; the bound does not change on the way
        n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
        while (n.y < 0xC0000000u) {
            const uint32_t meta = n.y;
            const float th = as_f32(n.x);
            const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
            const float dd = qa - th;
            const uint32_t nd2 = __float_as_uint(-(dd * dd));
            const uint32_t right = meta & 0x3FFFFFFFu;
            const bool go_left = qa < th;
            const uint32_t cur1 = cur + 1u;
            *reinterpret_cast<lds_u32x2*>(col_addr + (uint32_t)avail * ROWB) = u32x2{go_left ? right : cur1, nd2};
            avail += (int)nd2 < nb ? 1 : 0;
            cur = go_left ? cur1 : right;
            n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
        }
    } while (__ballot(cur != dummy || avail > 0) != 0ull);
#pragma unroll
    for (int j = 0; j < K; ++j) { w.d[j] = d[j]; w.id[j] = id[j]; }
    w.slow = slow; w.c3n = c3n; w.cur = cur; w.avail = avail;
}

// Capped rounds: like walk_rounds, but a round runs at most C internal steps; a lane that has not reached its leaf by then sits
// out the next leaf stage (x = +inf, pop switched off) and keeps descending after it. Between the flat loop (C = 1, both blocks
// every trip) and the rounds loop (C = ∞: every round waits for the wave's longest descent).
// stop_at (round 4): the loop ends once at most `stop_at` lanes of the wave still have work — a wave runs until its slowest lane is done,
// and the last few lanes of nearly every wave are what half of the rounds are paid for (lane efficiency 0.53). The lanes left over
// keep their state in w and in their stack rows: the caller spills it and a continuation kernel picks them up, 64 stragglers to a
// wave, with this very function (it reloads the node at w.cur first). 0 = run to the end.
template <int K, int ROWB, int C, bool STAMP = false>
__device__ __forceinline__ void walk_rounds_capped(__amdgpu_buffer_rsrc_t rsrc, Walk<K>& w, float alpha, uint32_t dummy, uint32_t col_addr, int cap, int stop_at = 0) {
    const float qx = w.qx, qy = w.qy, qz = w.qz;
    uint32_t cur = w.cur;
    int avail = w.avail;
    uint32_t my_rounds = 0, all_rounds = 0;  // STAMP: rounds in which this lane still had work / rounds the wave ran
    float d[K];
    uint32_t id[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { d[j] = w.d[j]; id[j] = w.id[j]; }
    uint32_t slow = w.slow, c3n = w.c3n;
    u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
    do {
        if (STAMP) { my_rounds += (cur != dummy || avail > 0) ? 1u : 0u; all_rounds++; }
        const bool is_leaf = n.y >= 0xC0000000u;
        const uint32_t a4 = col_addr + (uint32_t)(avail - 4) * ROWB;
        const u32x2 r3 = *reinterpret_cast<lds_u32x2*>(a4), r2 = *reinterpret_cast<lds_u32x2*>(a4 + ROWB),
                    r1 = *reinterpret_cast<lds_u32x2*>(a4 + 2u * ROWB), r0 = *reinterpret_cast<lds_u32x2*>(a4 + 3u * ROWB);
        const float dx = qx - as_f32(n.x), dy = qy - as_f32(n.z), dz = qz - as_f32(n.w);
        const float dis2 = dx * dx + (dy * dy + dz * dz);
        const float x = is_leaf ? dis2 : __builtin_inff();
        bool c[K];
#pragma unroll
        for (int j = 0; j < K; ++j) c[j] = x < d[j];
        if (K >= 2) {
            const float gap = d[K - 1] - d[K - 2];
            const float xt = gap == 0.0f ? x : __builtin_inff();
            slow = xt < d[K - 1] ? 1u : slow;
        }
#pragma unroll
        for (int j = K - 1; j >= 1; --j) {
            const uint32_t below = id[j - 1], here = id[j];
            const uint32_t t = c[j] ? cur : here;
            id[j] = c[j - 1] ? below : t;
            d[j] = __builtin_amdgcn_fmed3f(d[j - 1], x, d[j]);
        }
        {
            const uint32_t here = id[0];
            id[0] = c[0] ? cur : here;
            d[0] = __builtin_fminf(d[0], x);
        }
        const int nbound = (int)__float_as_uint(-(d[K - 1] * alpha));
        const int nb = is_leaf ? nbound : (int)0x80000000u;
        int hit = 4;
        uint32_t nxt = is_leaf ? dummy : cur;
        hit = (int)r3.y < nb ? 3 : hit; nxt = (int)r3.y < nb ? r3.x : nxt;
        hit = (int)r2.y < nb ? 2 : hit; nxt = (int)r2.y < nb ? r2.x : nxt;
        hit = (int)r1.y < nb ? 1 : hit; nxt = (int)r1.y < nb ? r1.x : nxt;
        hit = (int)r0.y < nb ? 0 : hit; nxt = (int)r0.y < nb ? r0.x : nxt;
        const int avail_eff = is_leaf ? avail : 0;
        const int used = min(min(hit + 1, 4), avail_eff);
        const int low = avail_eff - 1 - min(hit, 3);
        const uint32_t c3sel = low < 2 ? c3n : 0u;
        // Rows 2.. hold one stored level each on the first descent and cannot overflow there; but a CANDIDATE (rows 0/1) sits on an
        // un-stored level, and the descent from it may push more survivors than there are rows (deep, unbalanced trees whose split
        // planes hug the query: a map made of straight lines does it). A push beyond the last row falls off the LDS allocation and
        // is lost, so a stack that stands higher than `cap` rows at a leaf stage — it only grows between two of them — sends the
        // query to the deep pass as well. One compare and a scalar OR per stage.
        const bool deep = ((int)c3sel < nb) | (avail > cap);
        c3n = deep ? 1u : c3n;
        // A query that goes to the deep pass is recomputed there from scratch: its lane RETIRES here (round 4) instead of finishing a
        // traversal whose result is thrown away — these are the long traversals, the ones a wave waits for (0.5-3 % of the queries,
        // but 30-85 % of the waves hold one). After an overflow the rows above the stack's top are not the lane's own either.
        nxt = deep ? dummy : nxt;
        const bool moved = nxt != cur;
        cur = nxt;
        avail = deep ? 0 : avail - used;
        // a lane that popped loads its new node; one that was not on a leaf still holds its internal node in n
        if (moved) n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
#pragma unroll
        for (int step = 0; step < C; ++step) {
            if (n.y < 0xC0000000u) {
                const uint32_t meta = n.y;
                const float th = as_f32(n.x);
                const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
                const float dd = qa - th;
                const uint32_t nd2 = __float_as_uint(-(dd * dd));
                const uint32_t right = meta & 0x3FFFFFFFu;
                const bool go_left = qa < th;
                const uint32_t cur1 = cur + 1u;
                *reinterpret_cast<lds_u32x2*>(col_addr + (uint32_t)avail * ROWB) = u32x2{go_left ? right : cur1, nd2};
                avail += (int)nd2 < nbound ? 1 : 0;
                cur = go_left ? cur1 : right;
                n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
            }
        }
    } while (__popcll(__ballot(cur != dummy || avail > 0)) > stop_at);
#pragma unroll
    for (int j = 0; j < K; ++j) { w.d[j] = d[j]; w.id[j] = id[j]; }
    w.slow = slow; w.c3n = c3n; w.cur = cur; w.avail = avail;
    if (STAMP) { w.rounds = my_rounds; w.wave_rounds = all_rounds; }
}

}  // namespace locgpu
